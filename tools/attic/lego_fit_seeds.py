#!/usr/bin/env python3
"""PSNR of the nerf_lego-shaped fit over seeds, fused width-128 decoders vs torch layers (is a gap systematic or run noise?)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import harness, hip_ops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
kw = dict(steps=steps, latent=True, feature_dim=4, num_lods=24, max_grid_res=512, hidden_dim=128, prune_every=100)
real = hip_ops.mlp_supported
for seed in (0, 1, 2):
    row = {}
    for fused in (True, False):
        hip_ops.mlp_supported = real if fused else (lambda *a: False)
        r = harness.fit_nerf(torch.device("cuda:0"), seed=seed, **kw)
        row["fused" if fused else "torch"] = (round(r["psnr"], 2), round(r["ms_per_step"], 2), r["occupied_cells"])
    print(seed, row, flush=True)
hip_ops.mlp_supported = real
