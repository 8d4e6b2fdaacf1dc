#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd import harness
dev = torch.device("cuda:0")
for graphed in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = harness.fit_image(dev, steps=1000, graphed=graphed)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"graphed={graphed}: 1000 steps 512x768 config B: {r['ms_per_step']:.3f} ms/step (wall {dt:.2f} s incl. setup)  PSNR {r['psnr']:.2f} dB  bpp {r['bpp']:.3f}", flush=True)
