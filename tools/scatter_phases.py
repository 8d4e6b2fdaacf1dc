import os, sys, ctypes
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
os.environ["SHACIRA_HIP_LIB"] = os.environ["GRAFT_REPO_ROOT"] + "/shacira_amd/lib/variants/sct.so"
import numpy as np, torch
from shacira_amd import hip_ops, _lib
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, bw, N = 3, 19, 1 << 20
res, F = geo(16, 2048, 16), 2
sizes = [min(2 ** bw, r ** dim) for r in res]
first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).cuda()
T = sum(sizes); g = torch.Generator().manual_seed(0)
coords = (torch.rand(N, dim, generator=g) * 2 - 1).cuda(); go = torch.randn(N, 32, generator=g).cuda()
lib = ctypes.CDLL(os.environ["SHACIRA_HIP_LIB"])
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3): hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
torch.cuda.synchronize(); lib.shacira_debug_scatter_times(buf, 1)
it = 10
for _ in range(it): hip_ops.hashgrid_backward(dim, coords, go, T, torch.float32, first, res, bw, F)
torch.cuda.synchronize(); lib.shacira_debug_scatter_times(buf, 0)
n = buf[7]
names = ["loads+enumerate+rank", "barrier1", "scan+barrier2", "stage", "syncthreads(vmcnt0)", "write-out issue", "store drain"]
print("workgroups", n, "per call", n / it)
tot = 0
for k in range(7):
    us = buf[k] / n / 100.0   # wall_clock64: 100 MHz
    tot += us
    print(f"{names[k]:24s} {us:7.2f} us per workgroup")
print("sum", round(tot, 2), "us; 16384 WGs / 512 slots x sum =", round(tot * 32, 1), "us")
