#!/usr/bin/env python3
"""Config-B image fit with the SGA warm-up of the shipped configs vs straight-through rounding from the start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd import harness
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for sga in (False, True):
    r = harness.fit_image(torch.device("cuda:0"), steps=steps, use_sga=sga)
    print(f"use_sga={sga}: PSNR {r['psnr']:.2f} dB, bpp {r['bpp']:.3f}, {r['ms_per_step']:.2f} ms/step ({steps} steps)")
