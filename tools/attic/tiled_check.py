#!/usr/bin/env python3
"""Cell-sorted ("tiled") forward vs the other variants and the oracle: bit-exactness + timing over batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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

cases = []
for dim in (3, 2):
    for e in (16, 17, 18, 19, 20, 21):
        cases.append((dim, 19, (1 << e) + (77 if e == 18 else 0), 16, 2048))
cases += [(2, 11, 393216, 16, 512), (2, 11, 24 * 393216, 16, 512)]
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
    f = lambda: hip_ops._hashgrid_forward(dim, coords, table, first, res, bw)
    out, tm = {}, {}
    for tiled in (0, 1):
        _lib.set_option("tiled", tiled)
        out[tiled] = f().clone()
        tm[tiled] = timed(f)
    _lib.set_option("tiled", -1)
    ta = timed(f)
    n_or = min(N, 1 << 15)
    _lib.set_option("tiled", 1)
    cs = coords[:n_or].contiguous()
    fo = hip_ops._hashgrid_forward(dim, cs, table, first, res, bw)
    _lib.set_option("tiled", -1)
    ok_or = np.array_equal(fo.cpu().numpy(), oc.forward(cs.cpu().numpy(), table.cpu().numpy(), first_np, res, bw))
    print(f"dim{dim} bw{bw} N={N:9d}: other variants {tm[0]:.3f} ms  tiled {tm[1]:.3f} ms  auto {ta:.3f} ms | "
          f"tiled == others bit-exact: {torch.equal(out[0], out[1])}, tiled slice == oracle: {ok_or}")
