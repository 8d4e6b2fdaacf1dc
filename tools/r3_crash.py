#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, bw, L, F = 3, 19, 16, 2
N = int(sys.argv[1]); kind = sys.argv[2]
res = geo(16, 2048, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
first = torch.from_numpy(first_np).cuda(); T = int(sum(sizes))
rng = np.random.default_rng(72)
c = rng.uniform(-1, 1, (N, dim))
if kind == "cube": c = c ** 3
coords = torch.from_numpy(c.astype(np.float32)).cuda()
go = torch.randn(N, L * F).cuda()
g = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
torch.cuda.synchronize()
print("ok", N, kind, float(g.double().sum()), float(go.double().sum()), flush=True)
