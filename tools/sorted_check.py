#!/usr/bin/env python3
"""A/B: forward variant 6 (samples in caller order) vs variant 7 (cell-sorted order). Bit-exact check + timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib

def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]

def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it

for dim, bw, N in ((3, 19, 1 << 20), (2, 19, 1 << 20), (3, 19, 1 << 18), (3, 19, (1 << 20) + 77)):
    res, F = geo(16, 2048, 16), 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
    T = sum(sizes)
    g = torch.Generator().manual_seed(0)
    table = (torch.randn(T, F, generator=g) * 0.01).cuda()
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
    f = hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda
    out = {}
    for v in (6, 7):
        _lib.set_option("fwd_variant", v)
        out[v] = f(coords, table, first, res, bw).clone()
        t = timed(lambda: f(coords, table, first, res, bw))
        print(f"dim{dim} N={N} variant {v}: {t:.3f} ms")
    print("  bit-exact:", torch.equal(out[6], out[7]))
    _lib.set_option("fwd_variant", -1)
