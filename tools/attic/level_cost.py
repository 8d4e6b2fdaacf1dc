#!/usr/bin/env python3
"""Forward cost of a 16-level grid made of 16 copies of one resolution (dense vs hashed levels). Dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
def run(dim, res1, bw=19, N=1 << 20, L=16, variant=-1):
    res = [res1] * L
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
    T = sum(sizes); g = torch.Generator().manual_seed(0)
    table = (torch.randn(T, 2, generator=g) * 0.01).cuda()
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
    _lib.set_option("fwd_variant", variant)
    f = (hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda)
    for _ in range(3): f(coords, table, first, res, bw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f(coords, table, first, res, bw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10
for r in (17, 31, 59, 81, 154, 562, 2049):
    print(f"3-D 16 x res {r:5d} ({'dense' if r**3 < 2**19 else 'hashed'}, {min(2**19, r**3)*8/1e6:.2f} MB/level): fwd {run(3, r):.3f} ms", flush=True)
