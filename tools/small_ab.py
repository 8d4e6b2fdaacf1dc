#!/usr/bin/env python3
"""Interleaved A/B of option sets on the small NeRF batches (config D: 4096 rays x 16 ray points; E shard: 512 x 16).
usage: small_ab.py D|E "<optset>" ...   optset = "name=value,..." ("-" = defaults); prints forward / backward medians"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops, harness, _lib


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


which = sys.argv[1]
optsets = sys.argv[2:] or ["-"]
n = {"D": 65536, "E": 8192, "H": 131072, "Q": 262144 - 4096, "X": 16384, "Y": 32768}[which]
dim, bw, L, F = 3, 19, 16, 2
res = geo(16, 2048, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(7)
table = (torch.randn(T, F, generator=g) * 0.01).cuda()
coords = harness.ray_points(n // 16, 16, g).contiguous().cuda()
go = torch.randn(n, L * F, generator=g).cuda()


def timed(fn, it=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def run(optset):
    saved = []
    if optset != "-":
        for kv in optset.split(","):
            k, v = kv.split("=")
            saved.append((k, _lib.get_option(k)))
            _lib.set_option(k, int(v))
    same = torch.equal(hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw), ref_f)
    if not same:
        print("!! forward differs under", optset)
    f = timed(lambda: hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw))
    b = timed(lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F))
    for k, v in saved:
        _lib.set_option(k, v)
    return f, b


ref_f = hip_ops.hashgrid_interpolate_cuda(coords, table, first, res, bw).clone()
ref = hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F).clone()
times = {c: [] for c in optsets}
for rep in range(5):
    for c in optsets:
        times[c].append(run(c))
for c in optsets:
    f = np.median([t[0] for t in times[c]])
    b = np.median([t[1] for t in times[c]])
    print(f"{which} {c:44s} fwd {f:6.1f} us   bwd {b:6.1f} us   sum {f + b:6.1f}")
