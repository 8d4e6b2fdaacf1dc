#!/usr/bin/env python3
"""Backward: transpose and count as two kernels on two streams (bwd_fuse=0) vs fused into one kernel (bwd_fuse=1)."""
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
for dim in (3, 2):
    res, bw, F = geo(16, 2048, 16), 19, 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
    T = sum(sizes); g = torch.Generator().manual_seed(0)
    for N in (262144, 524288, 786432, 1048576, 2097152):
        coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
        row = []
        for fuse in (0, 2):
            _lib.set_option("bwd_fuse", fuse)
            row.append(timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)) * 1e3)
        _lib.set_option("bwd_fuse", 1)
        print(f"dim{dim} N={N}: two kernels {row[0]:.1f} us, fused {row[1]:.1f} us")
