#!/usr/bin/env python3
"""Forward S1 under option sets, interleaved medians.  usage: [WORKLOAD=S1|S1h|S1q|S1x|LEGO|LEGOq] [R3_DTYPE=f16] fwd_lc_ab.py "<optset>" ...   ("-" = defaults)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops, _lib


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


W = {"S1": (3, 19, 1 << 20, 2048, 16, 2), "S1h": (3, 19, 1 << 19, 2048, 16, 2), "S1q": (3, 19, 1 << 18, 2048, 16, 2),
     "S1x": (3, 19, 3 << 19, 2048, 16, 2), "S1e": (3, 19, 1 << 17, 2048, 16, 2), "S1s": (3, 19, 1 << 16, 2048, 16, 2),
     "S1m": (3, 19, 3 << 16, 2048, 16, 2), "LEGOs": (3, 19, 51200, 512, 24, 4), "LEGOt": (3, 19, 25600, 512, 24, 4),
     "S2": (2, 19, 1 << 20, 2048, 16, 2), "S2q": (2, 19, 1 << 18, 2048, 16, 2), "S2e": (2, 19, 1 << 17, 2048, 16, 2), "LEGO": (3, 19, 409600, 512, 24, 4), "LEGOq": (3, 19, 102400, 512, 24, 4)}
name = os.environ.get("WORKLOAD", "S1")
dim, bw, N, mx, L, F = W[name]
DT = torch.float16 if os.environ.get("R3_DTYPE") == "f16" else torch.float32
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
table = (torch.randn(T, F, generator=g) * 0.01).cuda().to(DT)
plan = hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)


def run(optset, it=40):
    saved = []
    if optset != "-":
        for kv in optset.split(","):
            k, v = kv.split("=")
            saved.append((k, _lib.get_option(k)))
            _lib.set_option(k, int(v))
    fn = lambda: hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw, plan=plan)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    for k, v in saved:
        _lib.set_option(k, v)
    return a.elapsed_time(b) / it


cfgs = sys.argv[1:] or ["-"]
times = {c: [] for c in cfgs}
for rep in range(5):
    for c in cfgs:
        times[c].append(run(c))
for c in cfgs:
    print(f"fwd {name} {c:40s} median {np.median(times[c]):.4f} ms  (min {min(times[c]):.4f} max {max(times[c]):.4f})")
