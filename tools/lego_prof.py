#!/usr/bin/env python3
"""Kernel-level profile target: nerf_lego.yaml-shaped fit, fused decoders (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd import harness
r = harness.fit_nerf(torch.device("cuda:0"), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 80, latent=True, feature_dim=4,
                     num_lods=24, max_grid_res=512, hidden_dim=128, prune_every=100)
print(r)
