#!/usr/bin/env python3
"""Per-unit phase timeline of the consume pass (instrumented variant build only):
    make -C shacira_amd/csrc variant NAME=ctrace FILE=hashgrid_bwd_bin.hip EXTRA=-DCONSUME_TRACE
    SHACIRA_HIP_LIB=$GRAFT_REPO_ROOT/shacira_amd/lib/variants/ctrace.so python tools/consume_trace.py <workload>
Wave 0's 100 MHz clock at: unit start, first round of item loads issued, image zeroed, first round added, all rounds added,
unit end (flush issued). Prints per-level averages and the busy fraction of the workgroups over the kernel's span."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from shacira_amd import hip_ops, _lib

W = {"S1": (3, 19, 1 << 20, 2048, 16, 2), "D": (3, 19, 65536, 2048, 16, 2), "LEGO": (3, 19, 1 << 18, 512, 24, 4),
     "LEGO400": (3, 19, 409600, 512, 24, 4), "D8k": (3, 19, 8192, 2048, 16, 2), "S2": (2, 19, 1 << 20, 2048, 16, 2)}


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


name = sys.argv[1]
DT = torch.float16 if os.environ.get("R3_DTYPE") == "f16" else torch.float32
dim, bw, N, mx, L, F = W[name]
res = geo(16, mx, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = int(sum(sizes))
g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda()
go = torch.randn(N, L * F, generator=g).cuda().to(DT)
lib = _lib.lib()
fn = lib.shacira_debug_consume_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
buf = np.zeros((16384, 8), np.uint64)
cnt = ctypes.c_uint(0)
for _ in range(3):
    hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F)
torch.cuda.synchronize()
assert fn(buf.ctypes.data, 16384, ctypes.byref(cnt)) == 0
hip_ops.hashgrid_backward(dim, coords, go, T, DT, first, res, bw, F)
torch.cuda.synchronize()
assert fn(buf.ctypes.data, 16384, ctypes.byref(cnt)) == 0
n = min(cnt.value, 16384)
r = buf[:n]
lvl = (r[:, 0] >> np.uint64(32)).astype(np.int64)
items = r[:, 1].astype(np.int64)
t = r[:, 2:8].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0          # 100 MHz -> us
span = us[:, 5].max()
dur = us[:, 5] - us[:, 0]
print(f"{name} {DT}: {n} units, kernel span {span:.1f} us, sum of unit times {dur.sum():.0f} us "
      f"= {dur.sum() / span:.1f} workgroups busy on average")
print("level  units  items/unit |  issue   zero   1st-add  rest-add  flush | total us per unit")
for l in sorted(set(lvl.tolist())):
    m = lvl == l
    d = us[m]
    ph = [d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 5] - d[:, 4]]
    print(f"{l:5d} {m.sum():6d} {items[m].mean():10.0f} | " + " ".join(f"{p.mean():7.2f}" for p in ph) +
          f" | {dur[m].mean():7.2f}   (res {res[l]})")
# when do units start / end: concurrency profile in 10 slices
edges = np.linspace(0, span, 11)
act = [int(((us[:, 0] < b) & (us[:, 5] > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("units overlapping each tenth of the span:", act)
