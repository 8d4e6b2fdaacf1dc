#!/usr/bin/env python3
"""Option sweeps of the tiled forward / backward on S1 (timing only)."""
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

dim, bw, N = int(os.environ.get("DIM", 3)), 19, int(os.environ.get("N", 1 << 20))
res, F = geo(16, 2048, 16), 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes)
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, 32, generator=g).cuda()
f = lambda: hip_ops._hashgrid_forward(dim, coords, table, first, res, bw)
_lib.set_option("tiled", 0)
ref = f().clone()
print(f"old path fwd {timed(f):.3f} ms")
_lib.set_option("tiled", 1)
for lc in (-1, 16, 12, 10, 9, 8, 7, 6, 5, 4, 0):
    _lib.set_option("tiled_lc_fwd", lc)
    ok = torch.equal(f(), ref)
    print(f"tiled fwd lc={lc}: {timed(f):.3f} ms  bit-exact={ok}")
_lib.set_option("tiled_lc_fwd", -1)
