#!/usr/bin/env python3
"""Backward timing + oracle slice on S1-like problems (A/B of backward changes)."""
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

for dim, bw, N, mx in [(3, 19, 1 << 20, 2048), (3, 19, 65536, 2048), (3, 19, (1 << 18) + 77, 2048), (2, 19, 1 << 20, 2048), (2, 11, 393216, 512)]:
    res, F = geo(16, mx, 16), 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    first = torch.from_numpy(first_np).cuda()
    T = sum(sizes)
    g = torch.Generator().manual_seed(0)
    coords = (torch.rand(N, dim, generator=g) * 2 - 1)
    coords[0] = 1.0; coords[1] = -1.0; coords[2] = float("nan")
    coords = coords.cuda()
    go = torch.randn(N, 32, generator=g).cuda()
    b = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
    tb = timed(b)
    n_or = min(N, 1 << 16)
    cs, gs = coords[:n_or].contiguous(), go[:n_or].contiguous()
    gr = hip_ops.hashgrid_backward(dim, cs, gs, T, torch.float32, first, res, bw, F)
    ref_g = oc.backward(cs.cpu().numpy(), gs.cpu().numpy(), (T, F), first_np, res, bw)
    err = 0.0
    for l in range(16):
        lo = first_np[l]; hi = lo + sizes[l]
        err = max(err, float(np.abs(gr[lo:hi].cpu().numpy() - ref_g[lo:hi]).max() / max(np.abs(ref_g[lo:hi]).max(), 1e-30)))
    full = b()
    lhs = float((full.double().sum(0)).abs().sum()); 
    print(f"dim{dim} bw{bw} N={N}: bwd {tb:.3f} ms | slice (N={n_or}) max per-level rel err vs oracle {err:.2e} | sum check {float(full.double().sum()):.6e} vs {float(go.double()[~torch.isnan(coords).any(1)].sum()):.6e}")
