#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops as ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, bw, L, F = 3, 19, 16, 2
N = int(sys.argv[1]); what = sys.argv[2]; newdata = sys.argv[3]
res = geo(16, 2048, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
tf = torch.from_numpy(first_np).cuda(); T = int(sum(sizes))
rng = np.random.default_rng(71)
tc = torch.from_numpy(rng.uniform(-1, 1, (N, dim)).astype(np.float32)).cuda()
tg = torch.randn(N, L * F).cuda()
tt = (torch.randn(T, F) * 0.01).cuda()
out = torch.empty((T, 2), device="cuda")
ws = ops.backward_workspace(dim, N, T, torch.float32, res, bw, 2, torch.device("cuda"))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    if "f" in what: ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw)
    if "b" in what: ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    if "f" in what: feats = ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw)
    if "b" in what: ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws)
rng = np.random.default_rng(72)
if newdata == "cube":
    tc.copy_(torch.from_numpy((rng.uniform(-1, 1, (N, dim)) ** 3).astype(np.float32)))
elif newdata == "uni":
    tc.copy_(torch.from_numpy((rng.uniform(-1, 1, (N, dim))).astype(np.float32)))
out.fill_(-3.0)
if len(sys.argv) > 4 and sys.argv[4] == "zerows": ws.zero_()
if len(sys.argv) > 4 and sys.argv[4] == "garbage": ws.fill_(0x7f)
torch.cuda.synchronize()
graph.replay()
torch.cuda.synchronize()
print("ok", N, what, newdata, float(out.double().sum()) if "b" in what else 0.0, float(tg.double().sum()), flush=True)
