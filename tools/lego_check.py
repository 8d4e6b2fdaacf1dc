#!/usr/bin/env python3
"""nerf_lego.yaml-shaped grid (3-D, 24 levels, F=4, bw 19, res 16..512): fwd/bwd timing (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, res, bw, F = 3, geo(16, 512, 24), 19, 4
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
for N in (65536, 1 << 20):
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 24 * F, generator=g).cuda()
    f = lambda: hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw)
    b = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
    for _ in range(3): f(); b()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0
    for _ in range(10):
        ev[0].record(); f(); ev[1].record(); b(); ev[2].record(); torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]) / 10; tb += ev[1].elapsed_time(ev[2]) / 10
    byt = 12 + 24 * 8 * F * 4 + 24 * F * 4
    print(f"nerf_lego grid T={T} N={N}: fwd {tf:.3f} ms ({N*byt/tf/1e6:.0f} GB/s) bwd {tb:.3f} ms ({N*byt/tb/1e6:.0f} GB/s) -> {N/(tf+tb)/1e3:.0f} Msamples/s", flush=True)
