#!/usr/bin/env python3
"""End-to-end NeRF-style fit of the analytic scene (harness.fit_nerf) on the GPU: PSNR and ms/step."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import harness
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
num_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
r = harness.fit_nerf(torch.device("cuda:0"), steps=steps, num_steps=num_steps)
print(json.dumps(r))
