"""Data-parallel execution of the hash-grid path: one process per GPU, the sample batch (pixels / ray points)
sharded across ranks, parameters replicated, and ONE all-reduce (sum) of a flat gradient buffer per step
(RCCL over xGMI on MI355X: ``torch.distributed`` backend "nccl"; "gloo" on CPU for tests).

The reference has no distributed code at all; this is the scaling axis BASELINE.json asks
for. Samples are independent, so there is no data-path collective: the only exchange is the gradient sum.

Design for xGMI: the codebook gradient (48.8 MB for the 16-level 3-D grid) plus the few-hundred-byte decoder /
entropy-model gradients live in ONE contiguous fp32 buffer (``FlatGradients``); ``param.grad`` are views into it,
so autograd accumulates straight into the communication buffer and a step issues exactly one collective with no
packing copies. Loss is a global mean: each rank scales its local-mean loss by 1/world (equal shards).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    use_gpu = torch.cuda.is_available()
    # SHACIRA_TEST_SINGLE_GPU=1: every rank on cuda:0 with the gloo backend -- only to exercise the multi-rank control
    # flow of bench.py / the harness on a one-GPU box (RCCL refuses two ranks on one device)
    single = use_gpu and os.environ.get("SHACIRA_TEST_SINGLE_GPU") == "1"
    if single:
        local, backend = 0, backend or "gloo"
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # torchrun: join even a 1-rank group
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if use_gpu and not single:
            kwargs["device_id"] = device
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shard_bounds(num_samples, rank, world):
    """Contiguous, near-equal shard [lo, hi) of a batch of ``num_samples`` for ``rank``."""
    base, rem = divmod(num_samples, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensor, rank, world):
    lo, hi = shard_bounds(tensor.shape[0], rank, world)
    return tensor[lo:hi]


class FlatGradients:
    """One contiguous gradient buffer for a list of parameters; ``p.grad`` become views into it.

    ``allreduce()`` sums the buffer over all ranks with a single collective (async handle optional so the caller
    can overlap it with independent work, e.g. the MLP backward or the next batch's sample generation).
    """

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("parameters must share one device and dtype")
        self.group = group
        sizes = [p.numel() for p in self.params]
        # 128-byte aligned segments so every view starts on a cache line
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 31) // 32 * 32
        self.flat = torch.zeros(off, dtype=dt, device=dev)
        self._touched = set()
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            # which parameters took part in this step's loss: the views are never None, so without this Adam would
            # apply weight decay / moment decay to parameters that received no gradient (torch skips grad=None ones,
            # and so does the 1-GPU path that zeroes with set_to_none) and 1-GPU / N-GPU runs would diverge
            p.register_post_accumulate_grad_hook(lambda q, s=self: s._touched.add(id(q)))

    def zero_(self):
        self.flat.zero_()
        self._touched.clear()

    def hide_untouched(self):
        """Before optimizer.step(): parameters that received no gradient this step get ``.grad = None`` (the optimiser
        skips them, as with one GPU). Returns the list to hand to ``restore`` afterwards. The loss graph is the same on
        every rank, so every rank hides the same parameters."""
        hidden = [p for p in self.params if id(p) not in self._touched]
        for p in hidden:
            p.grad = None
        return hidden

    def restore(self, hidden):
        for p in hidden:
            p.grad = self.view_of(p)

    def view_of(self, param):
        for p, o in zip(self.params, self.offsets):
            if p is param:
                return self.flat[o:o + p.numel()].view_as(p)
        raise KeyError("parameter not in this bucket")

    def allreduce(self, async_op=False):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


def global_mean_loss(local_sum, global_count):
    """Loss whose gradient, summed over ranks, equals the gradient of the mean over the GLOBAL batch: the local sum
    over the GLOBAL element count (shards need not be equal: ``shard_bounds`` hands out near-equal ones)."""
    return local_sum / global_count
