#!/usr/bin/env python3
"""S1 table backward: fork / image-size rules around 2^18 ... 2^20 samples (30 calls each, two rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
def timed(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
dim = int(os.environ.get("DIM", 3))
F, L, mx = 2, 16, 2048
res, bw = geo(16, mx, L), 19
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
combos = (("auto", {}), ("fork0", {"bwd_fork": 0}), ("acc128", {"bin_acc_kib": 128}), ("acc64", {"bin_acc_kib": 64}),
          ("fork0+acc128", {"bwd_fork": 0, "bin_acc_kib": 128}), ("fork0+acc64", {"bwd_fork": 0, "bin_acc_kib": 64}))
for N in (196608, 262144, 327680, 400000, 458752, 524288, 655360, 786432, 1048576):
    coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, L * F, generator=g).cuda()
    b = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
    best = {}
    for rnd in range(2):
        for name, opts in combos:
            saved = {k: _lib.get_option(k) for k in opts}
            for k, v in opts.items(): _lib.set_option(k, v)
            t = timed(b)
            best[name] = min(best.get(name, 1e9), t)
            for k, v in saved.items(): _lib.set_option(k, v)
    print(f"dim={dim} N={N}: " + "  ".join(f"{k} {v:.3f}" for k, v in best.items()), flush=True)
