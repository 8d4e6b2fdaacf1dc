#!/usr/bin/env python3
"""Cell-sorted ("tiled") path vs the previous kernels and the oracle: bit-exact forward, backward tolerance, timing."""
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

cases = [(3, 19, 1 << 20, 16, 2048), (3, 19, (1 << 18) + 77, 16, 2048), (2, 19, 1 << 20, 16, 2048), (2, 11, 393216, 16, 512)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for dim, bw, N, mn, mx in cases:
    res, F = geo(mn, mx, 16), 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    first = torch.from_numpy(first_np).cuda()
    T = sum(sizes)
    g = torch.Generator().manual_seed(0)
    table = (torch.randn(T, F, generator=g) * 0.01).cuda()
    coords = (torch.rand(N, dim, generator=g) * 2 - 1)
    coords[0] = 1.0; coords[1] = -1.0; coords[2] = float("nan"); coords[3] = 2.5; coords[4] = -9.0
    coords = coords.cuda()
    go = torch.randn(N, 32, generator=g).cuda()
    f = lambda ctx=False: hip_ops._hashgrid_forward(dim, coords, table, first, res, bw, want_context=ctx)
    b = lambda ctx=None: hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, context=ctx)
    out = {}
    for tiled in (0, 1):
        _lib.set_option("tiled", tiled)
        feats = f().clone(); grad = b().clone()
        tf, tb = timed(f), timed(b)
        out[tiled] = (feats, grad)
        print(f"dim{dim} bw{bw} N={N} tiled={tiled}: fwd {tf:.3f} ms  bwd {tb:.3f} ms  sum {tf+tb:.3f}")
    feats, ctx = f(True)
    if ctx is not None:
        gctx = b(ctx).clone()
        tf2 = timed(lambda: f(True))
        tb = timed(lambda: b(ctx))
        print(f"   with a sample context: fwd {tf2:.3f} ms  bwd {tb:.3f} ms  sum {tf2+tb:.3f}; grad equal-ish: {float((gctx - out[1][1]).abs().max()):.3e}")
    _lib.set_option("tiled", -1)
    print("  forward bit-exact vs old path:", torch.equal(out[0][0], out[1][0]))
    d = (out[0][1] - out[1][1]).abs().max().item(); s = out[0][1].abs().max().item()
    print(f"  backward max|diff| vs old path {d:.3e} (scale {s:.3e})")
    n_or = min(N, 1 << 16)
    # oracle on a slice: run the tiled path on the slice itself (forced)
    _lib.set_option("tiled", 1)
    cs, gs = coords[:n_or].contiguous(), go[:n_or].contiguous()
    fo, c2 = hip_ops._hashgrid_forward(dim, cs, table, first, res, bw, want_context=True)
    gr = hip_ops.hashgrid_backward(dim, cs, gs, T, table.dtype, first, res, bw, F, context=c2)
    _lib.set_option("tiled", -1)
    ref_f = oc.forward(cs.cpu().numpy(), table.cpu().numpy(), first_np, res, bw)
    print("  slice forward bit-exact vs oracle:", np.array_equal(fo.cpu().numpy(), ref_f))
    ref_g = oc.backward(cs.cpu().numpy(), gs.cpu().numpy(), (T, F), first_np, res, bw)
    err = 0.0
    for l in range(16):
        lo = first_np[l]; hi = lo + sizes[l]
        sc = np.abs(ref_g[lo:hi]).max()
        err = max(err, float(np.abs(gr[lo:hi].cpu().numpy() - ref_g[lo:hi]).max() / max(sc, 1e-30)))
    print(f"  slice backward max per-level relative error vs oracle: {err:.3e}")
