#!/usr/bin/env python3
"""rocprof target: S1 backward restricted to levels [lo, hi) a few times.  usage: levels_prof.py lo hi [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


lo, hi = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dim, bw, N, mx, L, F = 3, 19, 1 << 20, 2048, 16, 2
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, L * F, generator=g).cuda()
out = torch.zeros(T, F, device="cuda")
for _ in range(iters):
    hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F, levels=(lo, hi), out=out)
torch.cuda.synchronize()
