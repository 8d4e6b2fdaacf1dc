#!/usr/bin/env python3
"""Backward time at NeRF-sized batches for the tunables that shift fixed costs (developer A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
def timed(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
dim, res, bw, F = 3, geo(16, 2048, 16), 19, 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
for N in (16384, 65536, 131072, 262144, 524288, 1048576):
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
    row = []
    for acc in (128, 64, 0):
        _lib.set_option("bin_acc_kib", acc)
        row.append(timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)))
    _lib.set_option("bin_acc_kib", 0)
    print(f"N={N}: bwd acc128 {row[0]*1e3:.1f} us, acc64 {row[1]*1e3:.1f} us, auto {row[2]*1e3:.1f} us")
