#!/usr/bin/env python3
"""Round-3 A/B driver: times forward / backward of one workload under several option sets and checks each against the
oracle on a slice.   usage: r3_ab.py <workload> <optset> [<optset> ...]   optset = "name=value,name=value" or "-"
workloads: S1 (3-D bw19 2^20), S2 (2-D bw19 2^20), D (3-D ray points 65536), B (2-D bw11 393216), LEGO (3-D L24 F4 2^18)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shacira_amd import hip_ops, _lib
from oracle import hashgrid_c as oc


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


W = {"S1": (3, 19, 1 << 20, 2048, 16, 2), "S1b18": (3, 18, 1 << 20, 2048, 16, 2), "S2": (2, 19, 1 << 20, 2048, 16, 2), "D": (3, 19, 65536, 2048, 16, 2),
     "B": (2, 11, 393216, 512, 16, 2), "LEGO": (3, 19, 1 << 18, 512, 24, 4), "S1h": (3, 19, 1 << 19, 2048, 16, 2),
     "S1q": (3, 19, 1 << 18, 2048, 16, 2), "S1e": (3, 19, 1 << 17, 2048, 16, 2), "S2q": (2, 19, 1 << 18, 2048, 16, 2), "S2e": (2, 19, 1 << 17, 2048, 16, 2), "KY": (2, 11, 393216, 512, 24, 2), "B64k": (2, 11, 65536, 512, 16, 2), "B16k": (2, 11, 16384, 512, 16, 2), "B4k": (2, 11, 4096, 512, 16, 2), "B1k": (2, 11, 1024, 512, 16, 2), "B256": (2, 11, 256, 512, 16, 2), "Dh": (3, 19, 32768, 2048, 16, 2), "D8k": (3, 19, 8192, 2048, 16, 2), "D4k": (3, 19, 4096, 2048, 16, 2), "D2k": (3, 19, 2048, 2048, 16, 2), "Dq": (3, 19, 16384, 2048, 16, 2), "C": (2, 11, 9437184, 512, 16, 2),
     "S2_4k": (2, 19, 4096, 2048, 16, 2), "S2_8k": (2, 19, 8192, 2048, 16, 2), "S2_16k": (2, 19, 16384, 2048, 16, 2),
     "S2_32k": (2, 19, 32768, 2048, 16, 2), "L4k": (3, 19, 4096, 512, 24, 4), "L8k": (3, 19, 8192, 512, 24, 4),
     "L16k": (3, 19, 16384, 512, 24, 4), "D12k": (3, 19, 12288, 2048, 16, 2)}
name = sys.argv[1]
DT = torch.float16 if os.environ.get("R3_DTYPE") == "f16" else torch.float32
dim, bw, N, mx, L, F = W[name]
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
first = torch.from_numpy(first_np).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
if name == "D":
    from shacira_amd import harness
    coords = harness.ray_points(N // 16, 16, g).contiguous()
else:
    coords = torch.rand(N, dim, generator=g) * 2 - 1
coords[0] = 1.0
coords[1] = -1.0
coords = coords.cuda()
go = torch.randn(N, L * F, generator=g).cuda().to(DT)
table = (torch.randn(T, F, generator=g) * 0.01).cuda().to(DT)
fwd_op = hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda
n_or = min(N, 1 << 15)
cs, gs = coords[:n_or].contiguous(), go[:n_or].contiguous()
ref_g = oc.backward(cs.cpu().numpy(), gs.float().cpu().numpy(), (T, F), first_np, res, bw)
ref_f = oc.forward(cs.cpu().numpy(), table.float().cpu().numpy(), first_np, res, bw)
if DT == torch.float16:
    ref_f = ref_f.astype(np.float16)
full_ref = None
lib = _lib.lib()
for optset in sys.argv[2:]:
    opts = [] if optset == "-" else [kv.split("=") for kv in optset.split(",")]
    saved = [(k, lib.shacira_get_option(k.encode())) for k, _ in opts]
    for k, v in opts:
        _lib.set_option(k, int(v))
    try:
        f = lambda: fwd_op(coords, table, first, res, bw)
        b = lambda: hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F)
        tf, tb = timed(f), timed(b)
        # interleaved like the bench step
        both = timed(lambda: (f(), b()))
        gr = hip_ops.hashgrid_backward(dim, cs, gs, T, DT, first, res, bw, F).float().cpu().numpy()
        err = 0.0
        for l in range(L):
            lo, hi = first_np[l], first_np[l] + sizes[l]
            err = max(err, float(np.abs(gr[lo:hi] - ref_g[lo:hi]).max() / max(np.abs(ref_g[lo:hi]).max(), 1e-30)))
        full = b()
        if full_ref is None:
            full_ref = full.clone()
        dfull = float((full.float() - full_ref.float()).abs().max() / full_ref.float().abs().max())
        ff = fwd_op(cs, table, first, res, bw).cpu().numpy()
        fwd_ok = np.array_equal(ff, ref_f)
        print(f"{name} [{optset}] fwd {tf:.4f} bwd {tb:.4f} pair {both:.4f} ms | bwd slice err {err:.1e} full-vs-first {dfull:.1e} "
              f"| fwd slice bit-exact {fwd_ok}", flush=True)
    finally:
        for k, v in saved:
            _lib.set_option(k, v)
