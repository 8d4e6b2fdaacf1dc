#!/usr/bin/env python3
"""Interleaved A/B of backward configurations on S1 (median of repetitions): plain call vs planned calls under option sets.
usage: brick_cfg_ab.py [workload] "<optset>" ...   optset = "name=value,..." ; always includes the plain (plan-less) call"""
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
     "LEGO": (3, 19, 409600, 512, 24, 4)}
name = sys.argv[1]
optsets = sys.argv[2:]
DT = torch.float16 if os.environ.get("R3_DTYPE") == "f16" else torch.float32
dim, bw, N, mx, L, F = W[name]
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, L * F, generator=g).cuda().to(DT)
table = (torch.randn(T, F, generator=g) * 0.01).cuda().to(DT)
plan = hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)
hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw, plan=plan)
DEFAULTS = {"bwd_brick": -1, "bwd_brick_lo": -1, "bwd_brick_hi": -1, "bwd_brick_fork": 2, "bwd_brick_span": 0, "bwd_item12": -1}


def run(optset, it=40):
    for k, v in DEFAULTS.items():
        _lib.set_option(k, v)
    use_plan = optset != "plain"
    saved = []
    if use_plan and optset != "-":
        for kv in optset.split(","):
            k, v = kv.split("=")
            saved.append((k, _lib.get_option(k)))
            _lib.set_option(k, int(v))
    fn = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F, plan=plan if use_plan else None)
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


cfgs = ["plain"] + optsets
times = {c: [] for c in cfgs}
for rep in range(5):
    for c in cfgs:
        times[c].append(run(c))
for c in cfgs:
    print(f"{name} {c:60s} median {np.median(times[c]):.4f} ms  (min {min(times[c]):.4f} max {max(times[c]):.4f})")
