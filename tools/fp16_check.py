#!/usr/bin/env python3
"""S1 with an fp16 codebook (what the reference runs under AMP): forward / backward operator time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops
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
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
for dt in (torch.float32, torch.float16):
    table = (torch.randn(T, F, generator=g) * 0.01).cuda().to(dt)
    go = torch.randn(N, 32, generator=g).cuda().to(dt)
    tf = timed(lambda: hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw))
    tb = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, dt, first, res, bw, F))
    print(f"{dt}: forward {tf:.3f} ms, backward {tb:.3f} ms -> {N / (tf + tb) / 1e6:.2f} G samples/s")
