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


def visible_gpu_count():
    """Number of GPUs a child process would see, WITHOUT touching the HIP runtime (a launcher parent that has
    initialised the GPU must never fork / exec its ranks): HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES if set, else the KFD topology in sysfs (nodes whose ``simd_count`` > 0 are GPUs)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    root, n = "/sys/class/kfd/kfd/topology/nodes", 0
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    return n


def level_groups(num_lods, chunks):
    """Level ranges [(lb, le), ...] for building the gradient in ``chunks`` groups (coarse levels first: they are the
    small rows; the fine, 4 MiB-per-level rows are spread over the later groups so that each all-reduce moves a similar
    number of bytes while the next group is computed)."""
    chunks = max(1, min(int(chunks), num_lods))
    if chunks == 1:
        return [(0, num_lods)]
    cuts = [0] + [max(1, min(num_lods - 1, round(num_lods * (5 + 3 * k / (chunks - 1)) / 8))) for k in range(chunks - 1)]
    cuts = sorted(set(cuts + [num_lods]))
    return list(zip(cuts[:-1], cuts[1:]))


class GradientReducer:
    """Sum of a gradient tensor over the ranks, in place. ``collective``:

    * ``"allreduce"``: one ``all_reduce`` (RCCL picks ring / tree / direct: ``NCCL_ALGO``);
    * ``"rs_ag"``: ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` on the flat buffer -- on xGMI's full mesh every
      rank owns 1/world of the rows and all 7 links carry a share of each phase (48.8 MB is
      ~0.56 ms single-link-bound as a ring vs ~0.08 ms ideal as direct reduce-scatter + all-gather).

    The flat view must be padded to a multiple of ``world`` elements: allocate gradient buffers with
    ``padded_numel`` and hand the padded flat tensor in (the pad stays zero)."""

    def __init__(self, collective="allreduce", group=None):
        if collective not in ("allreduce", "rs_ag"):
            raise ValueError(f"unknown collective {collective!r}")
        self.collective, self.group = collective, group
        self._shard = None

    @staticmethod
    def padded_numel(numel, world):
        return (numel + world - 1) // world * world

    def reduce(self, flat, async_op=False):
        """``flat``: contiguous 1-D (or any contiguous) tensor; summed over ranks in place. Returns a list of work
        handles when ``async_op`` (wait on all of them), else None."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return [] if async_op else None
        world = dist.get_world_size(self.group)
        if self.collective == "allreduce":
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            return [w] if async_op else None
        v = flat.view(-1)
        if v.numel() % world:
            raise ValueError("rs_ag needs a buffer padded to a multiple of the world size (GradientReducer.padded_numel)")
        n = v.numel() // world
        if async_op:
            # the all-gather reads the shard the reduce-scatter fills: two async ops are ordered only on backends that keep
            # one stream per process group (RCCL), not on gloo's worker threads, and the single shard buffer cannot be shared
            # by two calls in flight -- so the asynchronous form gets a shard of its own and waits between the two phases
            shard = torch.empty(n, dtype=v.dtype, device=v.device)
            dist.reduce_scatter_tensor(shard, v, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            return [dist.all_gather_into_tensor(v, shard, group=self.group, async_op=True)]
        if self._shard is None or self._shard.numel() != n or self._shard.device != v.device or self._shard.dtype != v.dtype:
            self._shard = torch.empty(n, dtype=v.dtype, device=v.device)
        dist.reduce_scatter_tensor(self._shard, v, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_gather_into_tensor(v, self._shard, group=self.group)
        return None


def backward_in_groups(backward_levels, grad, groups, row_of, reducer):
    """The data-parallel step's gradient phase: ``backward_levels(lb, le, first)`` fills the rows of levels [lb, le) of
    ``grad`` ([T, F], contiguous); each finished group's row range starts its reduction asynchronously while the next
    group is computed. With one group (the default) this is one backward + one collective over the whole buffer.
    ``rs_ag`` reduces whole-buffer only (row ranges are not padded): it requires a single group."""
    if len(groups) == 1:
        backward_levels(groups[0][0], groups[0][1], True)
        reducer.reduce(grad)
        return
    if reducer.collective != "allreduce":
        raise ValueError("level groups overlap all-reduces of row ranges; use --collective allreduce with --ar-chunks > 1")
    pending = []
    for gi, (lb, le) in enumerate(groups):
        backward_levels(lb, le, gi == 0)
        pending += reducer.reduce(grad[row_of(lb):row_of(le)], async_op=True)
    for wk in pending:
        wk.wait()


def global_mean_loss(local_sum, global_count):
    """Loss whose gradient, summed over ranks, equals the gradient of the mean over the GLOBAL batch: the local sum
    over the GLOBAL element count (shards need not be equal: ``shard_bounds`` hands out near-equal ones)."""
    return local_sum / global_count
