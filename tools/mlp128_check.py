#!/usr/bin/env python3
"""Width-128 decoders (nerf_lego.yaml): fused MFMA kernels vs the same layers through torch (rocBLAS), fwd + bwd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from shacira_amd import hip_ops
from shacira_amd.wisp.models.decoders import BasicDecoder
dev = torch.device("cuda:0")

def timed(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it

for dims in ((96, 128, 1, 16), (43, 128, 2, 3), (32, 64, 1, 16), (43, 64, 2, 3)):
    IN, H, NH, OUT = dims
    for n in (1 << 16, 409600, 1 << 20):
        torch.manual_seed(0)
        dec = BasicDecoder(IN, OUT, torch.relu, True, nn.Linear, NH, H, []).to(dev)
        x = torch.randn(n, IN, device=dev, requires_grad=True); gy = torch.randn(n, OUT, device=dev)
        def fused():
            dec.zero_grad(); x.grad = None
            dec(x).backward(gy)
        def plain():
            dec.zero_grad(); x.grad = None
            h = x
            for lin in dec.layers: h = torch.relu(lin(h))
            dec.lout(h).backward(gy)
        fused(); gf = [p.grad.clone() for p in dec.parameters()] + [x.grad.clone()]
        plain(); gp = [p.grad.clone() for p in dec.parameters()] + [x.grad.clone()]
        err = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(gf, gp))
        print(f"{dims} n={n}: fused fwd+bwd {timed(fused):.3f} ms  torch layers {timed(plain):.3f} ms  max rel grad diff {err:.2e}", flush=True)
