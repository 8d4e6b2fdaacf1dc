#!/usr/bin/env python3
"""nerf_lego.yaml-shaped NeRF fit (LatentGrid F=4 x 24 levels, res 16..512, bw 19, decoders of width 128, SGA + entropy
model): ms/step with the fused width-128 decoders and with the same decoders as torch Linear layers."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import harness, hip_ops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kw = dict(steps=steps, latent=True, feature_dim=4, num_lods=24, max_grid_res=512, hidden_dim=128, prune_every=100)
real = hip_ops.mlp_supported
for fused in (True, False):
    hip_ops.mlp_supported = real if fused else (lambda *a: False)
    r = harness.fit_nerf(torch.device("cuda:0"), **kw)
    print(json.dumps({"decoders": "fused MFMA (width 128)" if fused else "torch Linear layers", **r}), flush=True)
hip_ops.mlp_supported = real
