#!/usr/bin/env python3
"""One workload's training loop for `rocprofv3 --kernel-trace --stats`: prints wall ms/step so that the kernel-time sum of
the trace can be set against it (host gaps). usage: step_breakdown.py nerf|nerf_pool|nerf_graphed|image|image_graphed [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import harness
what = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
WARM = 30      # a short fit first: module loading, allocator growth and autotuning stay out of the timed loop


def fit(n):
    if what == "nerf":
        return harness.fit_nerf(dev, steps=n)
    if what == "nerf_pool":
        return harness.fit_nerf(dev, steps=n, ray_pool=128)
    if what == "nerf_graphed":
        return harness.fit_nerf(dev, steps=n, ray_pool=128, graphed=True)
    return harness.fit_image(dev, steps=n, graphed=(what == "image_graphed"))


fit(WARM)
torch.cuda.synchronize()
r = fit(steps)
# "steps_traced": what a kernel trace of this process has to be divided by (warm-up fit included)
print(json.dumps({"what": what, "steps": steps, "steps_traced": steps + WARM, "ms_per_step": r["ms_per_step"], "psnr": r["psnr"]}))
