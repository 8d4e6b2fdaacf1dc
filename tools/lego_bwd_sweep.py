#!/usr/bin/env python3
"""nerf_lego.yaml grid (3-D, 24 levels, F=4, bw 19) backward: the batch-size rules (image size, fused count) by batch size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
def timed(fn, it=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for F, L, mx in ((4, 24, 512), (2, 16, 2048)):
    dim, res, bw = 3, geo(16, mx, L), 19
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
    T = sum(sizes); g = torch.Generator().manual_seed(0)
    for N in (1 << 16, 1 << 17, 1 << 18, 400_000, 1 << 19):
        coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, L * F, generator=g).cuda()
        b = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
        out = [f"auto {timed(b):.3f}"]
        for name, opts in (("acc64", {"bin_acc_kib": 64}), ("acc128", {"bin_acc_kib": 128}), ("fuse0", {"bwd_fuse": 0}),
                           ("fuse2", {"bwd_fuse": 2}), ("fork0", {"bwd_fork": 0})):
            saved = {k: _lib.get_option(k) for k in opts}
            for k, v in opts.items(): _lib.set_option(k, v)
            out.append(f"{name} {timed(b):.3f}")
            for k, v in saved.items(): _lib.set_option(k, v)
        print(f"F={F} L={L} N={N}: " + "  ".join(out), flush=True)
