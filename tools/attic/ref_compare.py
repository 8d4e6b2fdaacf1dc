"""HIP path and C oracle against the reference's own operators (oracle/_ref, built by oracle/ref_build.py) on the GPU box:
prints, per configuration, whether the forward is bit-identical and how far the gradients are apart (relative to each
level's largest entry), next to the run-to-run spread of the reference's own atomics.
    python tools/ref_compare.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CONFIGS, geo, table_layout  # noqa: E402
from oracle import hashgrid_c as oc  # noqa: E402
from oracle import ref_build  # noqa: E402
from shacira_amd import hip_ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 17) + 3
ref = ref_build.load()
dev = torch.device("cuda:0")
cases = {k: v + (2,) for k, v in CONFIGS.items()}
cases["lego_F4"] = (3, geo(16, 512, 24), 19, 4)
cases["small3d"] = (3, [4, 7, 12, 33, 80, 81], 19, 2)


def level_rel(a, b, first, sizes):
    worst = 0.0
    for l in range(len(sizes)):
        lo, hi = int(first[l]), int(first[l]) + int(sizes[l])
        s = np.abs(b[lo:hi]).max()
        if s > 0:
            worst = max(worst, float(np.abs(a[lo:hi] - b[lo:hi]).max() / s))
    return worst


for name, (dim, res, bw, F) in cases.items():
    sizes, first, T = table_layout(res, bw, dim)
    rng = np.random.default_rng(1)
    coords = rng.uniform(-1, 1, (N, dim)).astype(np.float32)
    coords[0] = 1.0; coords[1] = -1.0; coords[2] = np.nan; coords[3] = 2.5; coords[4] = -9.0
    coords[5] = np.float32(1.0) - np.float32(2.0 ** -24); coords[6] = np.float32(-1.0) + np.float32(2.0 ** -24)
    coords[7, 0] = 1.0
    table = (rng.standard_normal((T, F)) * 0.01).astype(np.float32)
    go = rng.standard_normal((N, len(res) * F)).astype(np.float32)
    for dt in (torch.float32, torch.float16, torch.float64):
        tc = torch.from_numpy(coords).to(dev)
        tt = torch.from_numpy(table).to(dev).to(dt)
        tg = torch.from_numpy(go).to(dev).to(dt)
        tf = torch.from_numpy(first).to(dev)
        three = dim == 3
        rf = (ref.hashgrid_interpolate_cuda if three else ref.hashgrid_interpolate2d_cuda)(tc, tt, tf, res, bw)
        hf = (hip_ops.hashgrid_interpolate_cuda if three else hip_ops.hashgrid_interpolate2d_cuda)(tc, tt, tf, res, bw)
        torch.cuda.synchronize()
        same = torch.equal(rf, hf) or bool(((rf == hf) | (rf.isnan() & hf.isnan())).all())
        nd = int((rf != hf).sum())
        mx = float((rf.double() - hf.double()).abs().max())
        line = f"{name:8s} {str(dt)[6:]:8s} fwd bit-identical to reference: {same} (differing {nd}, max abs {mx:.3g})"
        if dt == torch.float32:
            of = oc.forward(coords, table, first, res, bw)
            line += f"; oracle==reference: {np.array_equal(of, rf.cpu().numpy(), equal_nan=True)}"
            rb = [(ref.hashgrid_interpolate_backward_cuda if three else ref.hashgrid_interpolate2d_backward_cuda)(
                tc, tg, tt, tf, res, bw, F, False) for _ in range(2)]
            hb = (hip_ops.hashgrid_interpolate_backward_cuda if three else hip_ops.hashgrid_interpolate2d_backward_cuda)(
                tc, tg, tt, tf, res, bw, F, False)
            torch.cuda.synchronize()
            r0 = rb[0].double().cpu().numpy(); r1 = rb[1].double().cpu().numpy(); h = hb.double().cpu().numpy()
            ob = oc.backward(coords, go, (T, F), first, res, bw)
            line += (f"; bwd vs reference {level_rel(h, r0, first, sizes):.2e} of level max (reference run-to-run "
                     f"{level_rel(r1, r0, first, sizes):.2e}; oracle(f64) vs reference {level_rel(ob, r0, first, sizes):.2e})")
        print(line, flush=True)
