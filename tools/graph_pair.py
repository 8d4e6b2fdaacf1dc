#!/usr/bin/env python3
"""Operator pair (forward + backward) eager vs replayed from one HIP graph, S1 table at several batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, res, bw, F = 3, geo(16, 2048, 16), 19, 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
for N in (4096, 65536, 1 << 18, 1 << 20):
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
    out = torch.empty((T, F), device="cuda")
    ws = hip_ops.backward_workspace(dim, N, T, torch.float32, res, bw, F, coords.device)
    def pair():
        f = hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw)
        hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F, out=out, workspace=ws)
        return f
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): pair()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        pair()
    def timed(fn, it=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
    print(f"N={N}: eager {timed(pair):.4f} ms/pair, graph replay {timed(graph.replay):.4f} ms/pair", flush=True)
