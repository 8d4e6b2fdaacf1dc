#!/usr/bin/env python3
"""S1 (or DIM / N from the environment) fwd + bwd a few times -- run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib

def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]

dim = int(os.environ.get("DIM", 3)); bw = int(os.environ.get("BW", 19)); N = int(os.environ.get("N", 1 << 20))
res, F = geo(16, int(os.environ.get("MAXRES", 2048)), int(os.environ.get("LEVELS", 16))), 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes)
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
if os.environ.get("RAYS"):     # config D's batch: ray points, N / 16 rays x 16 samples
    from shacira_amd import harness
    coords = harness.ray_points(N // 16, 16, g).contiguous().cuda()
else:
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, len(res) * F, generator=g).cuda()
for _ in range(int(os.environ.get("ITERS", 10))):
    if not os.environ.get("BWD_ONLY"):
        hip_ops._hashgrid_forward(dim, coords, table, first, res, bw)
    if not os.environ.get("FWD_ONLY"):
        hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F)
torch.cuda.synchronize()
