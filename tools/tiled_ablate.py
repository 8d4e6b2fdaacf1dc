#!/usr/bin/env python3
"""Timing-only ablation of the unit kernels (option tiled_dbg; results are wrong with a non-zero mask)."""
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

dim, bw, N = 3, 19, 1 << 20
res, F = geo(16, 2048, 16), 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes)
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, 32, generator=g).cuda()
_lib.set_option("tiled", 1)
feats, ctx = hip_ops._hashgrid_forward(dim, coords, table, first, res, bw, want_context=True)
f = lambda: hip_ops._hashgrid_forward(dim, coords, table, first, res, bw)
b = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, context=ctx)
for name, masks in (("fwd", [0, 1, 2, 3, 4, 8, 12, 15]), ("bwd", [0, 16, 32, 64, 96, 112])):
    for m in masks:
        _lib.set_option("tiled_dbg", m)
        print(f"{name} dbg={m:3d}: {timed(f if name == 'fwd' else b):.3f} ms")
_lib.set_option("tiled_dbg", 0)
for lc in (8, 7, 6, 5, 0):
    _lib.set_option("tiled_lc_bwd", lc)
    ctx = hip_ops._hashgrid_forward(dim, coords, table, first, res, bw, want_context=True)[1]
    print(f"bwd lc={lc}: {timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, context=ctx)):.3f} ms")
_lib.set_option("tiled_lc_bwd", -1)
