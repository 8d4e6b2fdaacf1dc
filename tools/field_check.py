#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd import harness
r = harness.fit_field_3d(torch.device("cuda:0"), steps=500)
print(f"config D: 4096 rays x 16 samples/step, nerf_hash grid: PSNR@500 = {r['psnr']:.2f} dB, {r['ms_per_step']:.3f} ms/step")
