#!/usr/bin/env python3
"""Where the HOST time of an eager training step goes (cProfile of the config-B image fit / the NeRF fit): the steps are
launch-bound Python loops, the GPU is busy a third to a half of the wall time.   usage: host_profile.py image|nerf [steps]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import harness
what = sys.argv[1] if len(sys.argv) > 1 else "image"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
fit = (lambda n: harness.fit_image(dev, steps=n)) if what == "image" else (lambda n: harness.fit_nerf(dev, steps=n, ray_pool=64))
fit(30)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
r = fit(steps)
pr.disable()
print(f"{what}: {r['ms_per_step']:.3f} ms/step under the profiler ({steps} steps)")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
