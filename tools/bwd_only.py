#!/usr/bin/env python3
"""Backward operator of one named workload a few times: target for rocprofv3 --pmc runs.  usage: bwd_only.py <workload> [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops

W = {"S1": (3, 19, 1 << 20, 2048, 16, 2), "D": (3, 19, 65536, 2048, 16, 2), "LEGO": (3, 19, 1 << 18, 512, 24, 4),
     "LEGO400": (3, 19, 409600, 512, 24, 4), "S2": (2, 19, 1 << 20, 2048, 16, 2)}


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


dim, bw, N, mx, L, F = W[sys.argv[1]]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DT = torch.float16 if os.environ.get("R3_DTYPE") == "f16" else torch.float32
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, L * F, generator=g).cuda().to(DT)
for kv in filter(None, os.environ.get("SHACIRA_OPTS", "").split(",")):   # e.g. SHACIRA_OPTS=bwd_item12=1,bwd_run_pad=0
    from shacira_amd import _lib
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
for _ in range(iters):
    hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F)
torch.cuda.synchronize()
