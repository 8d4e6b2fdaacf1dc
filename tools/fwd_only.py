#!/usr/bin/env python3
"""Run only the forward (or backward) operator a few times: target for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else -1
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 3
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
res, bw, F, N = geo(16, 2048, 16), 19, 2, 1 << 20
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes)
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, 32, generator=g).cuda()
_lib.set_option("fwd_variant" if which == "fwd" else "bwd_variant", variant)
for _ in range(iters):
    if which == "fwd":
        (hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda)(coords, table, first, res, bw)
    else:
        hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F)
torch.cuda.synchronize()
