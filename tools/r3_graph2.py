#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shacira_amd import hip_ops as ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dim, bw, L, F = 3, 19, 16, 2
N = int(sys.argv[1])
res = geo(16, 2048, L)
sizes = [min(2 ** bw, r ** dim) for r in res]
first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
tf = torch.from_numpy(first_np).cuda(); T = int(sum(sizes))
rng = np.random.default_rng(71)
tc = torch.from_numpy(rng.uniform(-1, 1, (N, dim)).astype(np.float32)).cuda()
tg = torch.randn(N, L * F).cuda()
out = torch.empty((T, 2), device="cuda")
ws = ops.backward_workspace(dim, N, T, torch.float32, res, bw, 2, torch.device("cuda"))
flags = sys.argv[2] if len(sys.argv) > 2 else ""
if "s" in flags:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws)
    torch.cuda.current_stream().wait_stream(side)
else:
    ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws)
rng = np.random.default_rng(72)
tc.copy_(torch.from_numpy((rng.uniform(-1, 1, (N, dim))).astype(np.float32)))
if "z" in flags: ws.zero_()
out.fill_(-3.0)
torch.cuda.synchronize()
graph.replay()
torch.cuda.synchronize()
wg = ws.cpu().numpy().copy(); og = out.cpu().numpy().copy()
ws.zero_(); out.fill_(-3.0)
ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws)
torch.cuda.synchronize()
we = ws.cpu().numpy(); oe = out.cpu().numpy()
gT_bytes = ((N + 1) // 2 * 2) * L * F * 4
print("gT bytes", gT_bytes, "ws bytes", wg.size)
d = np.nonzero(wg != we)[0]
print("differing bytes", d.size, "first", d[:5], "last", d[-5:] if d.size else None)
# region summary in 1 MiB chunks after gT
if d.size and "v" in flags:
    h, edges = np.histogram(d, bins=40)
    for c, e in zip(h, edges): 
        if c: print(f"  offset ~{int(e)}: {c} bytes differ")
print("out equal", np.array_equal(og, oe), float(np.abs(og - oe).max()))
