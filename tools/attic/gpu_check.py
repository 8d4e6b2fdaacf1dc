#!/usr/bin/env python3
"""Quick on-GPU parity + timing check through the C-ABI (dev tool; the real tests live in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shacira_amd import hip_ops, _lib
from oracle import hashgrid_c as oc

def table_layout(res, bw, dim):
    s = [min(2 ** bw, r ** dim) for r in res]
    f = np.concatenate([[0], np.cumsum(s)[:-1]]).astype(np.int32)
    return s, f, int(sum(s))

def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]

def run(dim, res, bw, N, F=2, seed=0, check=True, iters=20):
    dev = torch.device("cuda:0")
    s, f, T = table_layout(res, bw, dim)
    rng = np.random.default_rng(seed)
    coords = rng.uniform(-1, 1, (N, dim)).astype(np.float32)
    if check:
        coords[:4] = 1.0; coords[4:8] = -1.0
    table = (rng.standard_normal((T, F)) * 0.01).astype(np.float32)
    go = rng.standard_normal((N, len(res) * F)).astype(np.float32)
    tc, tt, tg = (torch.from_numpy(a).to(dev) for a in (coords, table, go))
    tf = torch.from_numpy(f).to(dev)
    fwd = hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda
    bwd = hip_ops.hashgrid_interpolate_backward_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_backward_cuda
    feats = fwd(tc, tt, tf, res, bw)
    gcb = bwd(tc, tg, tt, tf, res, bw, F, False)
    torch.cuda.synchronize()
    msg = f"dim={dim} L={len(res)} bw={bw} N={N} T={T}"
    if check:
        rf = oc.forward(coords, table, f, res, bw)
        rg = oc.backward(coords, go, (T, F), f, res, bw)
        ef = np.abs(feats.cpu().numpy() - rf).max() / np.abs(rf).max()
        eg = np.abs(gcb.cpu().numpy() - rg).max() / np.abs(rg).max()
        exact = (feats.cpu().numpy() == rf).mean()
        msg += f"  fwd relerr={ef:.2e} (bit-equal frac {exact:.4f})  bwd relerr={eg:.2e}"
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(3):
        fwd(tc, tt, tf, res, bw); bwd(tc, tg, tt, tf, res, bw, F, False)
    torch.cuda.synchronize()
    tfw = tbw = 0.0
    for _ in range(iters):
        ev[0].record(); fwd(tc, tt, tf, res, bw); ev[1].record(); bwd(tc, tg, tt, tf, res, bw, F, False); ev[2].record()
        torch.cuda.synchronize()
        tfw += ev[0].elapsed_time(ev[1]); tbw += ev[1].elapsed_time(ev[2])
    tfw /= iters; tbw /= iters
    nc = 2 ** dim
    bytes_f = 4 * dim + len(res) * nc * F * 4 + len(res) * F * 4
    msg += f"  fwd {tfw:.3f} ms ({N*bytes_f/tfw/1e6:.0f} GB/s)  bwd {tbw:.3f} ms ({N*bytes_f/tbw/1e6:.0f} GB/s)  fwd+bwd {N/(tfw+tbw)/1e3:.1f} Msamples/s"
    print(msg, flush=True)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), "abi", _lib.lib().shacira_abi_version())
    for fv in (3, 6, -1):
        _lib.set_option("fwd_variant", fv)
        print("fwd_variant", fv)
        run(3, geo(16, 2048, 16), 19, 20000)
        run(3, geo(16, 2048, 16), 19, 1 << 20, check=False)
        run(2, geo(16, 2048, 16), 19, 1 << 20, check=False)
        run(2, geo(16, 512, 16), 11, 393216, check=False)
        run(3, geo(16, 2048, 16), 19, 65536, check=False)
    _lib.set_option("fwd_variant", -1)
    print("auto")
    run(2, geo(16, 512, 8), 11, 50000)
    run(2, geo(16, 512, 16), 11, 50000)
    run(2, geo(16, 2048, 16), 19, 50000)
    run(3, geo(16, 2048, 16), 19, 50000)
    run(3, geo(16, 512, 24), 19, 20000, F=4)
    run(3, geo(16, 2048, 16), 19, 1 << 20, check=False)
    run(2, geo(16, 2048, 16), 19, 1 << 20, check=False)
    run(2, geo(16, 512, 16), 11, 393216, check=False)
    run(3, geo(16, 2048, 16), 19, 65536, check=False)
