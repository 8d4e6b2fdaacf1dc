"""Fused Adam for the hash-grid path's parameters ("next" row f1 of SURVEY.md section 8).

Drop-in for ``torch.optim.Adam`` (the optimiser the reference's trainers build, wisp/trainers/base_trainer.py:206-266)
restricted to what those configs use: amsgrad=False, maximize=False, L2 ``weight_decay``, per-group ``lr``. All fp32
GPU parameters that share (betas, eps) are stepped by ONE multi-tensor HIP launch (``shacira_adam_step_multi``, up to
32 tensors per launch) instead of torch's ~10-kernel foreach chain; with ``zero_grad_in_step=True`` the same pass
clears ``.grad`` (useful with ``dist.FlatGradients``, whose gradient buffer is persistent); ``capturable=True`` keeps
the step count on the device so ``step()`` can be recorded into a HIP graph. State dict keys (``step``, ``exp_avg``,
``exp_avg_sq``) match torch's. Parameters that are not fp32-contiguous-on-GPU are stepped with the same formula in
torch ops.
"""
import ctypes
import math

import torch

from . import _lib
from .hip_ops import _on_device

_MAX = 32


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, zero_grad_in_step=False,
                 capturable=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.zero_grad_in_step = zero_grad_in_step
        self.capturable = capturable

    # Bias correction follows every parameter's OWN ``state['step']`` (as torch.optim.Adam does): a parameter that gets its
    # first gradient late starts at t = 1, and a restored checkpoint continues at the saved count. Parameters that share a
    # count are stepped by one launch. capturable: the count is an int32 DEVICE tensor shared by the parameters that were
    # born in the same step (the kernel reads it, so the launch can be replayed from a HIP graph while it advances).
    def _launch(self, batch, b1, b2, eps, device, step, step_dev):
        k = len(batch)
        PtrArr, I64Arr, FArr = ctypes.c_void_p * k, ctypes.c_int64 * k, ctypes.c_float * k
        ps = PtrArr(*[p.data_ptr() for p, _, _, _, _, _ in batch])
        gs = PtrArr(*[g.data_ptr() for _, g, _, _, _, _ in batch])
        ms = PtrArr(*[m.data_ptr() for _, _, m, _, _, _ in batch])
        vs = PtrArr(*[v.data_ptr() for _, _, _, v, _, _ in batch])
        ns = I64Arr(*[p.numel() for p, _, _, _, _, _ in batch])
        lrs = FArr(*[float(lr) for _, _, _, _, lr, _ in batch])
        wds = FArr(*[float(wd) for _, _, _, _, _, wd in batch])
        sd = ctypes.c_void_p(step_dev.data_ptr()) if step_dev is not None else None
        with _on_device(device):
            rc = _lib.lib().shacira_adam_step_multi(k, ns, ps, gs, ms, vs, lrs, wds, float(b1), float(b2), float(eps),
                                                    int(step), sd, int(self.zero_grad_in_step),
                                                    ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
        _lib.check(rc, "adam_step_multi")

    def _new_state(self, p, born):
        st = self.state[p]
        if self.capturable:
            key = p.device
            if key not in born:                      # one shared device counter for everything born in this step
                born[key] = torch.zeros((1,), dtype=torch.int32, device=p.device)
            st["step"] = born[key]
        else:
            st["step"] = torch.tensor(0.0)
        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        pending = {}      # (device, b1, b2, eps, step key) -> (host step, device step tensor, [tensors])
        bumped = set()    # device counters already advanced in this call
        born = {}
        if self.capturable:
            # A device counter is shared by the parameters that have been stepped TOGETHER so far. When only some of its
            # sharers have a gradient in this step (the others were skipped: grad None), the stepping ones move to a copy of
            # the counter first, so that a skipped parameter's count -- and its later bias correction -- stays its own
            # (torch.optim.Adam semantics). Eager bookkeeping: under graph replay the set of stepped parameters is fixed.
            by_counter = {}
            for group in self.param_groups:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st and "step" in st and torch.is_tensor(st["step"]) and st["step"].is_cuda:
                        by_counter.setdefault(st["step"].data_ptr(), (st["step"], [], []))[1 if p.grad is not None else 2].append(p)
            for sd, stepping, skipped in by_counter.values():
                if stepping and skipped:
                    fresh = sd.clone()
                    for p in stepping:
                        self.state[p]["step"] = fresh
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st = self._new_state(p, born)
                g, m, v = p.grad, st["exp_avg"], st["exp_avg_sq"]
                fused = (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
                         and g.dtype == torch.float32 and not g.is_sparse)
                if self.capturable:
                    if not fused:
                        raise RuntimeError("FusedAdam(capturable=True) handles fp32 contiguous GPU parameters only")
                    sd = st["step"]
                    if sd.data_ptr() not in bumped:
                        sd += 1                      # on the stream: part of the captured graph
                        bumped.add(sd.data_ptr())
                    key = (p.device, b1, b2, group["eps"], sd.data_ptr())
                    pending.setdefault(key, (1, sd, []))[2].append((p, g, m, v, group["lr"], group["weight_decay"]))
                    continue
                st["step"] += 1
                t = int(st["step"])
                if fused:
                    key = (p.device, b1, b2, group["eps"], t)
                    pending.setdefault(key, (t, None, []))[2].append((p, g, m, v, group["lr"], group["weight_decay"]))
                    continue
                gr = g.add(p, alpha=group["weight_decay"]) if group["weight_decay"] else g
                m.mul_(b1).add_(gr, alpha=1 - b1)
                v.mul_(b2).addcmul_(gr, gr, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(group["eps"])
                p.addcdiv_(m, denom, value=-group["lr"] / (1 - b1 ** t))
                if self.zero_grad_in_step:
                    g.zero_()
        for (device, b1, b2, eps, _), (t, sd, items) in pending.items():
            for i in range(0, len(items), _MAX):
                self._launch(items[i:i + _MAX], b1, b2, eps, device, t, sd)
        return loss

    def load_state_dict(self, state_dict):
        """torch.optim.Adam-compatible: restores the moments AND the per-parameter step counts (a checkpoint written by
        torch.optim.Adam or by this class). capturable: equal counts on one device are re-shared as one device counter."""
        super().load_state_dict(state_dict)
        shared = {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if not st or "step" not in st:
                    continue
                t = int(torch.as_tensor(st["step"]).item())
                if self.capturable:
                    key = (p.device, t)
                    if key not in shared:
                        shared[key] = torch.full((1,), t, dtype=torch.int32, device=p.device)
                    st["step"] = shared[key]
                else:
                    st["step"] = torch.tensor(float(t))
