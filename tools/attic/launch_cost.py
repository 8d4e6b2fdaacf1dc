#!/usr/bin/env python3
"""Host time inside individual C-ABI calls of an eager image-fit step (perf_counter around the ctypes call).
usage: launch_cost.py [steps]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import _lib, harness
L = _lib.lib()
acc = collections.defaultdict(lambda: [0.0, 0])
names = ["shacira_adam_step_multi", "shacira_hashgrid_forward", "shacira_hashgrid_backward_levels", "shacira_latent_decode_forward",
         "shacira_latent_decode_backward", "shacira_entropy_bits_forward", "shacira_entropy_bits_backward", "shacira_mlp_forward",
         "shacira_mlp_backward"]
class Wrap:
    def __init__(self, fn, name):
        self.fn, self.name = fn, name
        self.argtypes, self.restype = getattr(fn, "argtypes", None), getattr(fn, "restype", None)
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); dt = time.perf_counter() - t
        acc[self.name][0] += dt; acc[self.name][1] += 1
        return r
for n in names:
    if hasattr(L, n):
        setattr(L, n, Wrap(getattr(L, n), n))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
harness.fit_image(dev, steps=30)
for v in acc.values(): v[0] = 0.0; v[1] = 0
r = harness.fit_image(dev, steps=steps)
print(f"{r['ms_per_step']:.3f} ms/step")
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:40s} {t / max(c, 1) * 1e6:8.1f} us per call  x{c / steps:.1f} per step")
