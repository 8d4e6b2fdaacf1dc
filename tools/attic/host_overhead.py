#!/usr/bin/env python3
"""Host cost of one forward + backward call pair: batches so small that the GPU is idle between launches (the S1 table,
N = 2^18 takes the forked large-batch path with all its launches and events but ~0.3 ms of GPU time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, res, bw, F = 3, geo(16, 2048, 16), 19, 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
for N in (1 << 20, 1 << 18, 4096):
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
    def step():
        f = hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw)
        gr = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # host-only: time the python + C launch path with the GPU never the bottleneck is not separable here; report both
    print(f"N={N}: enqueue loop {t_enq / 200 * 1e3:.3f} ms/step, with final sync {t_all / 200 * 1e3:.3f} ms/step", flush=True)
import cProfile, pstats
N = 4096
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    f = hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw)
    gr = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
