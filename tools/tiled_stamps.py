#!/usr/bin/env python3
"""Diagnostic build only (-DSHACIRA_TILED_STAMPS): per-phase cycle shares of the unit kernels (wave 0 of each workgroup)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib

def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]

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
L = _lib.lib()
fn = L.shacira_debug_tiled_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_uint64 * 32)()
for _ in range(3):
    feats, ctx = hip_ops._hashgrid_forward(dim, coords, table, first, res, bw, want_context=True)
    hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, context=ctx)
torch.cuda.synchronize()
fn(buf, 1)
feats, ctx = hip_ops._hashgrid_forward(dim, coords, table, first, res, bw, want_context=True)
hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, context=ctx)
torch.cuda.synchronize()
fn(buf, 1)
names = {1: "fwd region load (+barrier)", 2: "fwd compute (+barrier)", 3: "fwd epilogue assemble", 4: "fwd epilogue store",
         8: "bwd prologue row loads", 9: "bwd prologue distribute", 10: "bwd gmax", 11: "bwd region+zero", 12: "bwd accumulate",
         13: "bwd flush"}
v = list(buf)
for grp in ((1, 2, 3, 4), (8, 9, 10, 11, 12, 13)):
    tot = sum(v[k] for k in grp)
    for k in grp:
        print(f"  {names[k]:28s} {v[k]/1e6:10.2f} Mclk  {100*v[k]/max(tot,1):5.1f} %")
    print(f"  total {tot/1e6:.1f} Mclk over all workgroups (wave 0) -> per unit {tot/3025:.0f} clk")
