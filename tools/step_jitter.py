#!/usr/bin/env python3
"""Per-step times of the bench's planned S1 step (HIP events around every step of one back-to-back loop) + per-operator times:
where the mean's distance from the median comes from.   usage: step_jitter.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dim, bw, N, L, F = 3, 19, 1 << 20, 16, 2
res = geo(16, 2048, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, L * F, generator=g).cuda()
plan = hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)


def step(ev=None):
    if ev:
        ev[0].record()
    hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw, plan=plan)
    if ev:
        ev[1].record()
    hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F, plan=plan)
    if ev:
        ev[2].record()


for _ in range(20):
    step()
torch.cuda.synchronize()
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
import time
t0 = time.perf_counter()
for k in range(K):
    step(evs[k])
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K * 1e3
f = np.array([e[0].elapsed_time(e[1]) for e in evs])
b = np.array([e[1].elapsed_time(e[2]) for e in evs])
gap = np.array([evs[k][2].elapsed_time(evs[k + 1][0]) for k in range(K - 1)])
print(f"wall/step {wall:.4f} ms; fwd mean {f.mean():.4f} p50 {np.median(f):.4f}; bwd mean {b.mean():.4f} p50 {np.median(b):.4f}; "
      f"gap mean {gap.mean() * 1e3:.1f} us max {gap.max() * 1e3:.1f}")
print("fwd :", " ".join(f"{x * 1e3:.0f}" for x in f[:80]))
print("bwd :", " ".join(f"{x * 1e3:.0f}" for x in b[:80]))

# the same step captured once into a HIP graph and replayed (what a training loop with static buffers does)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
for _ in range(10):
    graph.replay()
torch.cuda.synchronize()
a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
for _ in range(K):
    graph.replay()
b2.record()
torch.cuda.synchronize()
print(f"graph replay: wall/step {(time.perf_counter() - t0) / K * 1e3:.4f} ms, events {a.elapsed_time(b2) / K:.4f} ms")
