#!/usr/bin/env python3
"""S1 step (forward + backward operators) issued eagerly vs replayed from a HIP graph: ms per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, res, bw, F, N = 3, geo(16, 2048, 16), 19, 2, 1 << 20
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
def step():
    f = hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw)
    gr = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
    return f, gr
def wall(fn, it=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
print("eager  :", round(wall(step), 4), "ms/step")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = step()
print("graphed:", round(wall(graph.replay), 4), "ms/step")
ref = step()
graph.replay(); torch.cuda.synchronize()
print("same forward:", torch.equal(out[0], ref[0]), " backward max rel diff:", float((out[1] - ref[1]).abs().max() / ref[1].abs().max()))
