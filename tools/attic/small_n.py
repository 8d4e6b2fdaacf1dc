#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dim, res, bw, F = 3, geo(16, 2048, 16), 19, 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
for _ in range(20):
    hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw)
    hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
torch.cuda.synchronize()
