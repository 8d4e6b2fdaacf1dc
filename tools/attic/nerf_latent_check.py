#!/usr/bin/env python3
"""NeRF-style fit with the compressed (LatentGrid + SGA + entropy model) table: PSNR, ms/step, coded size."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import harness
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for latent in (False, True):
    r = harness.fit_nerf(torch.device("cuda:0"), steps=steps, latent=latent)
    print(json.dumps({"latent": latent, **r}))
