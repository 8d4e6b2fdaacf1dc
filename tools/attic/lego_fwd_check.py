#!/usr/bin/env python3
"""nerf_lego.yaml grid (3-D, 24 levels, F=4, bw 19): forward, automatic path vs cell-sorted path forced, by batch size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for dim, F, L, mx in ((3, 4, 24, 512), (3, 2, 16, 2048), (2, 2, 16, 2048), (2, 4, 16, 2048)):
    res, bw = geo(16, mx, L), 19
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
    T = sum(sizes); g = torch.Generator().manual_seed(0)
    table = (torch.randn(T, F, generator=g) * 0.01).cuda()
    for N in (1 << 15, 1 << 16, 98304, 1 << 17, 196608, 1 << 18, 327680, 400_000, 1 << 19):
        coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
        f = lambda: hip_ops._hashgrid_forward(dim, coords, table, first, res, bw)
        row = []
        for tiled in (0, 1, -1):
            _lib.set_option("tiled", tiled)
            if os.environ.get("TRACE"):
                print(f"  dim={dim} F={F} N={N} tiled={tiled} ...", flush=True)
            row.append(timed(f))
            if os.environ.get("TRACE"):
                torch.cuda.synchronize()
        _lib.set_option("tiled", -1)
        print(f"dim={dim} F={F} L={L} N={N}: unsorted {row[0]:.3f} ms, cell-sorted {row[1]:.3f} ms, automatic {row[2]:.3f} ms", flush=True)
