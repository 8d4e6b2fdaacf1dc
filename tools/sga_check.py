#!/usr/bin/env python3
"""SGA latent decode, fwd+bwd: fused HIP kernels vs the torch-op chain of the reference, nerf_lego-sized table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd.wisp.models.latent_decoders import LatentDecoder
from shacira_amd.wisp.models.latent_decoders.quantizers import sga_sample
dev = torch.device("cuda:0")
def timed(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for rows, ld, F in ((7_879_908, 1, 4), (26_704, 1, 2), (6_098_925, 2, 2)):
    dec = LatentDecoder(ld, F, "none", "sq", True, use_sga=True, diff_sampling=True).to(dev)
    dec.temperature = 0.3
    lat = (torch.randn(rows, ld, device=dev) * 3).requires_grad_(True)
    gy = torch.randn(rows, F, device=dev)
    def fused():
        lat.grad = None
        dec(lat).backward(gy)
    def chain():
        lat.grad = None
        w = sga_sample(lat, dec.temperature, True)
        (dec.layers(w / dec.div)).backward(gy)
    print(f"rows={rows} ld={ld} F={F}: fused {timed(fused):.3f} ms, torch ops {timed(chain):.3f} ms")
