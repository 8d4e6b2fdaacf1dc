#!/usr/bin/env python3
"""One workload's training loop for `rocprofv3 --kernel-trace --stats`: prints wall ms/step so that the kernel-time sum of
the trace can be set against it (host gaps). usage: step_breakdown.py nerf|image|image_graphed [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd import harness
what = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
if what == "nerf":
    r = harness.fit_nerf(dev, steps=steps)
elif what == "image":
    r = harness.fit_image(dev, steps=steps)
else:
    r = harness.fit_image(dev, steps=steps, graphed=True)
print(json.dumps({"what": what, "steps": steps, "ms_per_step": r["ms_per_step"], "psnr": r["psnr"]}))
