#!/usr/bin/env python3
"""A/B of backward tunables on workload S1 (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
def bench(dim, N, opts):
    res, bw, F = geo(16, 2048, 16), 19, 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
    T = sum(sizes); g = torch.Generator().manual_seed(0)
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
    for k, v in opts.items(): _lib.set_option(k, v)
    f = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20
for dbg in (0, 4, 8, 12):
    print(f"dim=3 bin_debug={dbg}: bwd {bench(3, 1 << 20, {'bin_debug': dbg}):.3f} ms", flush=True)
_lib.set_option("bin_debug", 0)
