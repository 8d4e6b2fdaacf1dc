#!/usr/bin/env python3
"""Backward: staged transpose (bwd_rows=0) vs fused rows scatter (bwd_rows=1), with / without a sample context."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
from oracle import hashgrid_c as oc

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

cases = [(3, 19, 1 << 20, 2048), (3, 19, 65536, 2048), (3, 19, (1 << 18) + 77, 2048), (2, 19, 1 << 20, 2048), (2, 11, 393216, 512)]
for dim, bw, N, mx in cases:
    res, F = geo(16, mx, 16), 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    first = torch.from_numpy(first_np).cuda()
    T = sum(sizes)
    g = torch.Generator().manual_seed(0)
    table = (torch.randn(T, F, generator=g) * 0.01).cuda()
    coords = (torch.rand(N, dim, generator=g) * 2 - 1)
    coords[0] = 1.0; coords[1] = -1.0; coords[2] = float("nan"); coords[3] = 2.5
    coords = coords.cuda()
    go = torch.randn(N, 32, generator=g).cuda()
    f = lambda ctx=False: hip_ops._hashgrid_forward(dim, coords, table, first, res, bw, want_context=ctx)
    b = lambda ctx=None: hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, context=ctx)
    out = {}
    for rows in (0, 1):
        _lib.set_option("bwd_rows", rows)
        out[rows] = b().clone()
        tf, tb = timed(f), timed(b)
        feats, ctx = f(True)
        line = f"dim{dim} bw{bw} N={N} bwd_rows={rows}: fwd {tf:.3f} bwd {tb:.3f} sum {tf+tb:.3f}"
        if ctx is not None:
            gc = b(ctx).clone()
            tf2, tb2 = timed(lambda: f(True)), timed(lambda: b(ctx))
            line += f" | with context: fwd {tf2:.3f} bwd {tb2:.3f} sum {tf2+tb2:.3f} (diff {float((gc-out[rows]).abs().max()):.2e})"
        print(line)
    _lib.set_option("bwd_rows", 1)
    print(f"   max|rows - staged| = {float((out[0]-out[1]).abs().max()):.3e} (scale {float(out[0].abs().max()):.3e})")
    n_or = min(N, 1 << 16)
    cs, gs = coords[:n_or].contiguous(), go[:n_or].contiguous()
    gr = hip_ops.hashgrid_backward(dim, cs, gs, T, table.dtype, first, res, bw, F)
    ref_g = oc.backward(cs.cpu().numpy(), gs.cpu().numpy(), (T, F), first_np, res, bw)
    err = 0.0
    for l in range(16):
        lo = first_np[l]; hi = lo + sizes[l]
        sc = np.abs(ref_g[lo:hi]).max()
        err = max(err, float(np.abs(gr[lo:hi].cpu().numpy() - ref_g[lo:hi]).max() / max(sc, 1e-30)))
    print(f"   slice (N={n_or}) backward max per-level relative error vs oracle: {err:.3e}")
