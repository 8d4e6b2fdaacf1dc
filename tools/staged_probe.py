#!/usr/bin/env python3
"""How much of the S1 backward is the transpose pass? (full call vs a call that reuses the staged gradients)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
def timed(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
dim, res, bw, F, N = 3, geo(16, 2048, 16), 19, 2, 1 << 20
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
ws = hip_ops.backward_workspace(dim, N, T, torch.float32, res, bw, F, coords.device)
out = torch.empty((T, F), device=coords.device)
call = lambda flags: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F, levels=(0, 16), out=out, workspace=ws, flags=flags)
call(_lib.BWD_STAGE_ALL_LEVELS)
print("full (stage):", round(timed(lambda: call(_lib.BWD_STAGE_ALL_LEVELS)), 3), "ms")
print("reuse staged:", round(timed(lambda: call(_lib.BWD_REUSE_STAGED)), 3), "ms")
for fork in (0, 1):
    _lib.set_option("bwd_fork", fork)
    print(f"fork={fork} full:", round(timed(lambda: call(_lib.BWD_STAGE_ALL_LEVELS)), 3), " reuse:", round(timed(lambda: call(_lib.BWD_REUSE_STAGED)), 3))
