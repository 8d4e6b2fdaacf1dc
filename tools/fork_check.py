#!/usr/bin/env python3
"""A/B of the backward with and without the side-stream fork (count+scans || transpose+direct levels)."""
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

for dim, bw, N in ((3, 19, 1 << 20), (2, 19, 1 << 20), (3, 19, 65536)):
    res, F = geo(16, 2048, 16), 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
    T = sum(sizes)
    g = torch.Generator().manual_seed(0)
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
    go = torch.randn(N, 32, generator=g).cuda()
    out = {}
    for fork in (0, 1):
        _lib.set_option("bwd_fork", fork)
        out[fork] = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F).clone()
        t = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F))
        print(f"dim{dim} N={N} fork={fork}: {t:.3f} ms")
    print("  identical:", torch.equal(out[0], out[1]))
