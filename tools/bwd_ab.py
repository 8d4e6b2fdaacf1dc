#!/usr/bin/env python3
"""S1 backward (and forward) timing for A/B builds: SHACIRA_HIP_LIB=<variant.so> python3 tools/bwd_ab.py [iters]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops

def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]

def timed(fn, it):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it

it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(os.environ.get("N", 1 << 20))
dim, bw = int(os.environ.get("DIM", 3)), 19
res, F = geo(16, 2048, 16), 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
first = torch.from_numpy(first_np).cuda()
T = sum(sizes)
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, 32, generator=g).cuda()
table = torch.randn(T, F, generator=g).cuda()
b = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
tb = timed(b, it)
full = b()
chk = float(full.double().abs().sum())
print(f"{os.environ.get('SHACIRA_HIP_LIB', 'default')}: dim{dim} N={N} bwd {tb:.4f} ms  |sum| {chk:.9e}")
