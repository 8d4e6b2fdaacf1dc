#!/usr/bin/env python3
"""nerf_lego.yaml grid backward at one batch size, for rocprofv3 (N from the environment, default 262144)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, res, bw, F = 3, geo(16, 512, 24), 19, 4
N = int(os.environ.get("N", 262144))
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 24 * F, generator=g).cuda()
for _ in range(12):
    hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
torch.cuda.synchronize()
