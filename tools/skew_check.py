#!/usr/bin/env python3
"""Planned vs plain operator times on non-uniform batches of 2^20 samples (config D's table): ray points, a Gaussian blob, everything
in one block, a thin slab.  A robustness check: the sort's coarse bins and the brick units are sized for uniform batches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops, harness


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dim, bw, N, L, F = 3, 19, 1 << 20, 16, 2
res = geo(16, 2048, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
go = torch.randn(N, L * F, generator=g).cuda()
kinds = {
    "uniform": torch.rand(N, 3, generator=g) * 2 - 1,
    "ray_points": harness.ray_points(N // 16, 16, g),
    "gaussian_blob": (torch.randn(N, 3, generator=g) * 0.15).clamp(-1, 1),
    "thin_slab": torch.cat([torch.rand(N, 2, generator=g) * 2 - 1, torch.rand(N, 1, generator=g) * 0.02], 1),
    "one_block": 0.3 + torch.rand(N, 3, generator=g) * 0.02,
}
for name, c in kinds.items():
    coords = c.float().contiguous().cuda()
    plan = hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)
    tf0 = timed(lambda: hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw))
    tf1 = timed(lambda: hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw, plan=plan))
    tb0 = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F))
    tb1 = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F, plan=plan))
    g0 = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
    g1 = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F, plan=plan)
    err = float((g0 - g1).abs().max() / g0.abs().max())
    print(f"{name:14s} fwd {tf0:.3f} / planned {tf1:.3f} ms   bwd plain {tb0:.3f} / planned {tb1:.3f} ms   planned-vs-plain {err:.1e} of max")
