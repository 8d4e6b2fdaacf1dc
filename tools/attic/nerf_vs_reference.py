#!/usr/bin/env python3
"""The NeRF-style fit of harness.fit_nerf (config D: 3-D 16-level bw-19 hash grid, 4096 rays x 128 candidate samples, resident ray
pool, eager steps) with this library's hash-grid operators and with the REFERENCE'S OWN operators (oracle/_ref) substituted
underneath the same host code: PSNR on held-out rays at a fixed step and time per step. Development tool (GPU box).
    usage: nerf_vs_reference.py [steps [seeds]]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_build  # noqa: E402
from shacira_amd import harness, hip_ops  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 500
NSEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ref = ref_build.load()
dev = torch.device("cuda:0")
saved = (hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_backward)


def r_fwd3(coords, codebook, first_idx, resolution, bw):
    return ref.hashgrid_interpolate_cuda(coords.contiguous(), codebook.contiguous(), first_idx, [int(r) for r in resolution], int(bw))


def r_bwd(dim, coords, grad_output, table_rows, table_dtype, first_idx, resolution, bw, feature_dim, **kw):
    codebook = torch.empty(int(table_rows), int(feature_dim), dtype=table_dtype, device=coords.device)
    return ref.hashgrid_interpolate_backward_cuda(coords.contiguous(), grad_output.contiguous(), codebook, first_idx,
                                                  [int(r) for r in resolution], int(bw), int(feature_dim), False)


rows = []
for seed in range(NSEEDS):
    ours = harness.fit_nerf(dev, steps=STEPS, seed=seed, ray_pool=128)
    try:
        hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_backward = r_fwd3, r_bwd
        theirs = harness.fit_nerf(dev, steps=STEPS, seed=seed, ray_pool=128)
    finally:
        hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_backward = saved
    row = {"seed": seed, "psnr_this_library_db": ours["psnr"], "psnr_reference_kernels_db": theirs["psnr"],
           "ms_per_step": ours["ms_per_step"], "ms_per_step_reference_kernels": theirs["ms_per_step"],
           "samples_per_step": ours.get("samples_per_step")}
    rows.append(row)
    print(json.dumps(row), flush=True)
print(json.dumps({"config": f"NeRF-style fit, config-D grid, 4096 rays x 128 candidates, ray pool of 128 batches, {STEPS} eager steps",
                  "mean_psnr_delta_db": float(np.mean([r["psnr_this_library_db"] - r["psnr_reference_kernels_db"] for r in rows])),
                  "ms_per_step": float(np.mean([r["ms_per_step"] for r in rows])),
                  "ms_per_step_reference_kernels": float(np.mean([r["ms_per_step_reference_kernels"] for r in rows]))}))
