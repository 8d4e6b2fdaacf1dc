#!/usr/bin/env python3
"""Randomised sweep of this library's operators against the REFERENCE'S OWN operators (oracle/_ref/shacira_ref_ops.so, built by
oracle/ref_build.py) over the shape space both accept: dimension, level count, resolution range, table bitwidth, feature width,
batch size; fp32 forward + backward, fp16 and double forward. Edge coordinates (+-1, out of range, 1 - 2^-24) are included
whenever no dense level reaches res >= 258 (there the reference reads one row past the level, .cu:34-36 with pos + 1 == res).
    usage: fuzz_vs_reference.py [shapes] [first_seed]    -> a summary line (and one line per failure)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import geo, table_layout  # noqa: E402
from oracle import hashgrid_c as oc  # noqa: E402
from oracle import ref_build  # noqa: E402
from shacira_amd import hip_ops  # noqa: E402

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ref = ref_build.load()
dev = torch.device("cuda:0")
fails = 0
worst_fwd = worst_bwd = worst_h = worst_ref_self = 0.0
ref_over_bar = 0
for seed in range(first_seed, first_seed + n_shapes):
    rng = np.random.default_rng(90_000 + seed)
    dim = int(rng.choice([2, 3]))
    L = int(rng.integers(1, 25))
    lo = int(rng.integers(2, 33))
    hi = int(lo * rng.uniform(1.0, 60.0)) + 1
    res = geo(lo, hi, L) if L > 1 else [lo]
    bw = int(rng.integers(4, 20))
    F = int(rng.choice([2, 4]))
    N = int(rng.choice([1, 63, 257, 4_097, 30_011, 66_000, 140_001, (1 << 18) + 5]))
    sizes, first, T = table_layout(res, bw, dim)
    coords = rng.uniform(-1, 1, (N, dim)).astype(np.float32)
    dense_big = any(r ** dim < 2 ** bw and r >= 258 for r in res) or any(r ** dim == 2 ** bw for r in res)
    if dense_big:
        # a dense level with res >= 258 clamps onto res - 1 exactly (hi rounds to res - 1), and the reference then reads row
        # pos + 1 == res: beyond the level and, for the last level, beyond the TABLE (weight 0, but 0 * garbage can be NaN):
        # keep the samples inside the range where the reference is defined
        coords *= np.float32(0.99)
    if not dense_big and N >= 8:
        coords[0] = 1.0; coords[1] = -1.0; coords[3 % N] = 2.5; coords[4 % N] = -9.0
        coords[5 % N] = np.float32(1.0) - np.float32(2.0 ** -24); coords[6 % N] = np.float32(-1.0) + np.float32(2.0 ** -24)
    table = (rng.standard_normal((T, F)) * 0.01).astype(np.float32)
    go = rng.standard_normal((N, L * F)).astype(np.float32)
    tc, tf = torch.from_numpy(coords).to(dev), torch.from_numpy(first).to(dev)
    three = dim == 3
    try:
        for dt in (torch.float32, torch.float16, torch.float64):
            tt = torch.from_numpy(table).to(dev).to(dt)
            rf = (ref.hashgrid_interpolate_cuda if three else ref.hashgrid_interpolate2d_cuda)(tc, tt, tf, res, bw)
            hf = (hip_ops.hashgrid_interpolate_cuda if three else hip_ops.hashgrid_interpolate2d_cuda)(tc, tt, tf, res, bw)
            d = float((hf.double() - rf.double()).abs().max())
            if dt == torch.float16:
                ulp = 2.0 ** (np.floor(np.log2(max(float(rf.float().abs().max()), 6.2e-5))) - 10)   # half ulp at the largest value
                worst_h = max(worst_h, d / ulp)
                assert d <= ulp, ("fp16 forward", d, ulp)
            else:
                one = float(np.spacing(np.float32(np.abs(table).max() * (1 << dim) / 2)))
                worst_fwd = max(worst_fwd, d / one)
                assert d <= one, ("forward", str(dt), d, one)
        tt, tg = torch.from_numpy(table).to(dev), torch.from_numpy(go).to(dev)
        ob = None
        rb = (ref.hashgrid_interpolate_backward_cuda if three else ref.hashgrid_interpolate2d_backward_cuda)(
            tc, tg, tt, tf, res, bw, F, False).double().cpu().numpy()
        hb = hip_ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, F).double().cpu().numpy()
        for l in range(L):
            a, b = int(first[l]), int(first[l]) + int(sizes[l])
            s = np.abs(rb[a:b]).max()
            if s > 0:
                e = float(np.abs(hb[a:b] - rb[a:b]).max() / s)
                if e > 1e-5:
                    # rows that collect ~10^5 contributions each (a 3 x 3 level under 2^18 samples): the REFERENCE's fp32 atomics
                    # carry ~sqrt(n) * 6e-8 of rounding there. Judge both against the fp64 oracle instead.
                    if ob is None:
                        ob = oc.backward(coords, go, (T, F), first, res, bw)
                    e_hip = float(np.abs(hb[a:b] - ob[a:b]).max() / s)
                    e_ref = float(np.abs(rb[a:b] - ob[a:b]).max() / s)
                    ref_over_bar += 1
                    assert e_hip <= 1e-5, ("backward level vs fp64 oracle", l, e_hip, "reference itself", e_ref)
                    worst_ref_self = max(worst_ref_self, e_ref)
                    e = e_hip
                worst_bwd = max(worst_bwd, e)
    except Exception as exc:   # noqa: BLE001
        fails += 1
        print(f"FAIL seed {seed} dim {dim} L {L} res {res[0]}..{res[-1]} bw {bw} F {F} N {N}: {exc}", flush=True)
print(f"fuzz vs reference kernels: {n_shapes} shapes from seed {first_seed}: {fails} failures; worst forward deviation "
      f"{worst_fwd:.2f} of one rounding (fp32 / double), {worst_h:.2f} half ulp-of-max (fp16); worst gradient deviation "
      f"{worst_bwd:.1e} of a level's largest entry; {ref_over_bar} levels where the reference's own fp32 atomics are further than "
      f"1e-5 from the fp64 sum (up to {worst_ref_self:.1e}): there this library is held to the fp64 oracle", flush=True)
sys.exit(1 if fails else 0)
