#!/usr/bin/env python3
"""Cost of running the S1 backward as G sequential level groups that could share one item array (workspace diet).
usage: level_groups_ab.py [workload=S1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


dim, bw, N, mx, L, F = 3, 19, 1 << 20, 2048, 16, 2
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes)); g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, L * F, generator=g).cuda()
ws = hip_ops.backward_workspace(dim, N, T, torch.float32, res, bw, F, coords.device)
out = torch.empty(T, F, device="cuda")


def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def grouped(bounds):
    def run():
        for k, (lb, le) in enumerate(zip(bounds[:-1], bounds[1:])):
            hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F, levels=(lb, le), out=out,
                                      workspace=ws, flags=_lib.BWD_STAGE_ALL_LEVELS if k == 0 else _lib.BWD_REUSE_STAGED)
    return run


ref = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F).clone()
print("one call        %.4f ms" % timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)))
for bounds in ([0, 16], [0, 10, 16], [0, 9, 16], [0, 8, 12, 16], [0, 7, 10, 13, 16]):
    t = timed(grouped(bounds))
    grouped(bounds)(); torch.cuda.synchronize()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"groups {bounds}: {t:.4f} ms   max diff vs one call {err:.1e}")
