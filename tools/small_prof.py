#!/usr/bin/env python3
"""rocprof target: config D (4096 rays x 16 ray points) or E shard (512 x 16) forward + backward a few times.
usage: small_prof.py [D|E] [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops, harness


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


which = sys.argv[1] if len(sys.argv) > 1 else "D"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 65536 if which == "D" else 8192
dim, bw, L, F = 3, 19, 16, 2
res = geo(16, 2048, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(7)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = harness.ray_points(n // 16, 16, g).contiguous().cuda()
go = torch.randn(n, L * F, generator=g).cuda()
for _ in range(iters):
    hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw)
    hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
torch.cuda.synchronize()
