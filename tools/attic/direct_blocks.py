#!/usr/bin/env python3
"""All-direct Kodak tables: backward time (the direct-level kernel runs 256 workgroups in total)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
dim, res, bw, F = 2, geo(16, 512, 16), 11, 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
for N in (49152, 393216, 24 * 393216):
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
    t = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F))
    print(f"N={N}: backward {t * 1e3:.1f} us   (history: 512 workgroups in total 77.3 / 1178.8 us at 393216 / 9437184 samples, "
          f"256: 65.3 / 1152.8, 128: 104.3 / 2287.5, 1024: 86.9 / 1338.9)")
