#!/usr/bin/env python3
"""Config C (24 Kodak images batched: N = 9 437 184 lattice samples, 2-D 16-level grids): timing + properties."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shacira_amd import hip_ops
def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1)); return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]
dev = torch.device("cuda:0")
H, W, IM = 512, 768, 24
rows = (torch.arange(H, dtype=torch.float32) / H - 0.5) * 2
cols = (torch.arange(W, dtype=torch.float32) / W - 0.5) * 2
rr, cc = torch.meshgrid(rows, cols, indexing="ij")
lattice = torch.stack([rr, cc], -1).reshape(-1, 2)
g = torch.Generator().manual_seed(0)
coords = torch.cat([lattice[torch.randperm(H * W, generator=g)] for _ in range(IM)]).to(dev)
N = coords.shape[0]
for name, mx, bw in (("B  (bw 11, res 16..512)", 512, 11), ("B' (bw 19, res 16..2048)", 2048, 19)):
    res = geo(16, mx, 16)
    sizes = [min(2 ** bw, r ** 2) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).to(dev)
    T = sum(sizes)
    table = (torch.randn(T, 2, generator=g) * 0.01).to(dev)
    go = torch.randn(N, 32, generator=g).to(dev)
    f = lambda: hip_ops.hashgrid_interpolate2d_cuda(coords, table, first, res, bw)
    b = lambda: hip_ops.hashgrid_backward(2, coords, go, T, torch.float32, first, res, bw, 2)
    feats = f(); grad = b(); torch.cuda.synchronize()
    const = hip_ops.hashgrid_interpolate2d_cuda(coords, torch.full((T, 2), 0.5, device=dev), first, res, bw)
    lhs = float((feats.double() * go.double()).sum()); rhs = float((table.double() * grad.double()).sum())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0
    for _ in range(5):
        ev[0].record(); f(); ev[1].record(); b(); ev[2].record(); torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]) / 5; tb += ev[1].elapsed_time(ev[2]) / 5
    print(f"config C, table {name}: N={N} fwd {tf:.3f} ms bwd {tb:.3f} ms -> {N/(tf+tb)/1e3:.0f} Msamples/s fwd+bwd "
          f"({N*1296/(tf+tb)/1e6:.0f} GB/s algorithmic); partition of unity err {float((const-0.5).abs().max()):.1e}; "
          f"adjoint <f,go>={lhs:.6e} <t,g>={rhs:.6e}", flush=True)
