"""Fused Adam for the hash-grid path's parameters ("next" row f1 of SURVEY.md section 8).

Drop-in for ``torch.optim.Adam`` (the optimiser the reference's trainers build, wisp/trainers/base_trainer.py:206-266)
restricted to what those configs use: amsgrad=False, maximize=False, L2 ``weight_decay``, per-group ``lr``. Every
fp32 GPU parameter is stepped by ONE HIP kernel (``shacira_adam_step``) instead of torch's ~10-kernel foreach chain;
with ``zero_grad_in_step=True`` the same pass clears ``.grad`` (useful with ``dist.FlatGradients``, whose gradient
buffer is persistent). State dict keys (``step``, ``exp_avg``, ``exp_avg_sq``) match torch's, so checkpoints
interchange. Parameters that are not fp32-contiguous-on-GPU are stepped with the same formula in torch ops.
"""
import ctypes
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, zero_grad_in_step=False,
                 capturable=False):
        """capturable=True keeps the step count in device memory (like torch's ``capturable``) so that ``step()`` can
        be recorded into a HIP graph; GPU fp32 parameters only."""
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.zero_grad_in_step = zero_grad_in_step
        self.capturable = capturable
        self._step_dev = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = None
        if self.capturable:
            if self._step_dev is None:
                dev = next(p for g in self.param_groups for p in g["params"]).device
                self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._step_dev += 1          # on the stream: part of the captured graph
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g, m, v = p.grad, st["exp_avg"], st["exp_avg_sq"]
                if self.capturable:
                    L = L or _lib.lib()
                    with torch.cuda.device(p.device):
                        rc = L.shacira_adam_step_capturable(
                            p.numel(), ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(g.contiguous().data_ptr()),
                            ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(v.data_ptr()), float(group["lr"]), float(b1),
                            float(b2), float(group["eps"]), float(group["weight_decay"]),
                            ctypes.c_void_p(self._step_dev.data_ptr()), int(self.zero_grad_in_step),
                            ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
                    _lib.check(rc, "adam_step_capturable")
                    continue
                st["step"] += 1
                t = int(st["step"])
                if (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
                        and g.dtype == torch.float32 and not g.is_sparse):
                    L = L or _lib.lib()
                    with torch.cuda.device(p.device):
                        rc = L.shacira_adam_step(p.numel(), ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(g.data_ptr()),
                                                 ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(v.data_ptr()),
                                                 float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                 float(group["weight_decay"]), t, int(self.zero_grad_in_step),
                                                 ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
                    _lib.check(rc, "adam_step")
                else:
                    gr = g.add(p, alpha=group["weight_decay"]) if group["weight_decay"] else g
                    m.mul_(b1).add_(gr, alpha=1 - b1)
                    v.mul_(b2).addcmul_(gr, gr, value=1 - b2)
                    denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(group["eps"])
                    p.addcdiv_(m, denom, value=-group["lr"] / (1 - b1 ** t))
                    if self.zero_grad_in_step:
                        g.zero_()
        return loss
