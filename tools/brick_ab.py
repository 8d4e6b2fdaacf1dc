#!/usr/bin/env python3
"""Round-6 A/B: S1 (or another r3_ab workload) forward + backward with and without the batch's plan / the brick pass; checks the
planned backward against the plain one and the oracle on a slice.  usage: brick_ab.py [workload] [lo:hi ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops, _lib


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


W = {"S1": (3, 19, 1 << 20, 2048, 16, 2), "S1h": (3, 19, 1 << 19, 2048, 16, 2), "S1q": (3, 19, 1 << 18, 2048, 16, 2),
     "LEGO": (3, 19, 409600, 512, 24, 4)}
name = sys.argv[1] if len(sys.argv) > 1 else "S1"
ranges = sys.argv[2:]
DT = torch.float16 if os.environ.get("R3_DTYPE") == "f16" else torch.float32
dim, bw, N, mx, L, F = W[name]
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
first = torch.from_numpy(first_np).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1)
coords[0] = 1.0
coords[1] = -1.0
coords = coords.cuda()
go = torch.randn(N, L * F, generator=g).cuda().to(DT)
table = (torch.randn(T, F, generator=g) * 0.01).cuda().to(DT)
fwd = hip_ops.hashgrid_interpolate_cuda
plan = hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)
print("plan bytes", None if plan is None else plan.numel())
f_plain = fwd(coords, table, first, res, bw)
f_plan = fwd(coords, table, first, res, bw, plan=plan)
print("forward planned == plain:", torch.equal(f_plain, f_plan))
g_plain = hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F)
t_f = timed(lambda: fwd(coords, table, first, res, bw, plan=plan))
t_b0 = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F))
print(f"{name}: fwd(plan) {t_f:.4f} ms   bwd plain {t_b0:.4f} ms")


def level_err(a, b):
    out = []
    for l in range(L):
        lo, hi = int(first_np[l]), int(first_np[l]) + sizes[l]
        ref = b[lo:hi].double()
        out.append(float((a[lo:hi].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
    return out


for rg in ["auto"] + ranges:
    if rg == "auto":
        _lib.set_option("bwd_brick_lo", -1)
        _lib.set_option("bwd_brick_hi", -1)
    else:
        lo, hi = rg.split(":")
        _lib.set_option("bwd_brick_lo", int(lo))
        _lib.set_option("bwd_brick_hi", int(hi))
    g_b = hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F, plan=plan)
    errs = level_err(g_b, g_plain)
    t_b = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F, plan=plan))
    print(f"brick {rg:>6s}: bwd {t_b:.4f} ms   max err vs plain per level (of level max): {max(errs):.2e}  "
          + " ".join(f"{e:.0e}" for e in errs))
