#!/usr/bin/env python3
"""PSNR at a fixed step against the REFERENCE'S OWN hash-grid operators (oracle/_ref/shacira_ref_ops.so: the reference's .cu /
.cpp built for gfx950, oracle/ref_build.py) at the size SURVEY 8(d) names: the config-B LatentGrid image fit (2-D L16 F2 ld1
bw11, quantisation + entropy model on), 512x768 image, 1000 steps, same init / batches / entropy noise. Per seed: the fit with
this library's operators, the same fit with the reference's operators substituted underneath the same host code (twice: the
reference's atomics make two of ITS runs differ), tail = mean PSNR of the last 20 steps.
A development tool (run on the GPU box; the product never loads oracle/_ref).   usage: psnr_vs_reference.py [H W steps [seeds]]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_build  # noqa: E402
from shacira_amd import harness, hip_ops  # noqa: E402

H, W, STEPS = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (512, 768, 1000)))
NSEEDS = int(sys.argv[4]) if len(sys.argv) >= 5 else 4
ref = ref_build.load()
dev = torch.device("cuda:0")
saved = (hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_backward)


def r_fwd3(coords, codebook, first_idx, resolution, bw):
    return ref.hashgrid_interpolate_cuda(coords.contiguous(), codebook.contiguous(), first_idx, [int(r) for r in resolution], int(bw))


def r_fwd2(coords, codebook, first_idx, resolution, bw):
    return ref.hashgrid_interpolate2d_cuda(coords.contiguous(), codebook.contiguous(), first_idx, [int(r) for r in resolution], int(bw))


def r_bwd(dim, coords, grad_output, table_rows, table_dtype, first_idx, resolution, bw, feature_dim, **kw):
    # the reference's backward wants the codebook only for its shape / dtype (zeros_like, hashgrid_interpolate.cpp:81)
    codebook = torch.empty(int(table_rows), int(feature_dim), dtype=table_dtype, device=coords.device)
    f = ref.hashgrid_interpolate_backward_cuda if dim == 3 else ref.hashgrid_interpolate2d_backward_cuda
    return f(coords.contiguous(), grad_output.contiguous(), codebook, first_idx, [int(r) for r in resolution], int(bw),
             int(feature_dim), False)


tail = lambda r: float(np.mean([h[2] for h in r["history"][-20:]]))
rows = []
for seed in range(2, 2 + NSEEDS):
    ours = harness.fit_image(dev, steps=STEPS, height=H, width=W, seed=seed, log_every=1)
    refs = []
    try:
        hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_backward = r_fwd3, r_fwd2, r_bwd
        for _ in range(2):
            refs.append(harness.fit_image(dev, steps=STEPS, height=H, width=W, seed=seed, log_every=1))
    finally:
        hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_backward = saved
    row = {"seed": seed, "this_library_db": tail(ours), "reference_kernels_db": tail(refs[0]),
           "reference_kernels_second_run_db": tail(refs[1]), "delta_db": tail(ours) - tail(refs[0]),
           "reference_run_to_run_db": tail(refs[1]) - tail(refs[0]), "bpp": ours["bpp"], "bpp_reference_kernels": refs[0]["bpp"],
           "ms_per_step": ours.get("ms_per_step"), "ms_per_step_reference_kernels": refs[0].get("ms_per_step")}
    rows.append(row)
    print(json.dumps(row), flush=True)
d = [r["delta_db"] for r in rows]
s = [r["reference_run_to_run_db"] for r in rows]
print(json.dumps({"config": f"config-B LatentGrid image fit, {H}x{W}, {STEPS} steps, seeds 2..{1 + NSEEDS}", "bar_db": 0.05,
                  "mean_delta_db": float(np.mean(d)), "max_abs_delta_db": float(np.max(np.abs(d))),
                  "reference_run_to_run_max_abs_db": float(np.max(np.abs(s))),
                  "all_within_bar": bool(np.max(np.abs(d)) <= 0.05)}))
