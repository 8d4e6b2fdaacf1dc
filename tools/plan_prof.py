#!/usr/bin/env python3
"""rocprof target: S1 forward (building the plan) + backward (reading it) a few times.  usage: plan_prof.py [iters] [workload]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


W = {"S1": (3, 19, 1 << 20, 2048, 16, 2), "LEGO": (3, 19, 409600, 512, 24, 4)}
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dim, bw, N, mx, L, F = W[sys.argv[2] if len(sys.argv) > 2 else "S1"]
DT = torch.float16 if os.environ.get("R3_DTYPE") == "f16" else torch.float32
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, L * F, generator=g).cuda().to(DT)
table = (torch.randn(T, F, generator=g) * 0.01).cuda().to(DT)
plan = hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)
for _ in range(iters):
    hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw, plan=plan)
    hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F, plan=plan)
torch.cuda.synchronize()
