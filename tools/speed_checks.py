#!/usr/bin/env python3
"""Speed comparisons that used to be timing asserts inside `-m gpu` tests (they flake on a busy box): fused decoder MLP vs
torch Linear layers, fused Adam vs torch.optim.Adam. Prints the figures; run on the GPU box: python tools/speed_checks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from shacira_amd.optim import FusedAdam
from shacira_amd.wisp.models.decoders import BasicDecoder


def _torch_mlp(dec, x):
    h = x
    for lin in dec.layers:
        h = torch.relu(lin(h))
    return dec.lout(h)


def speed_fused_mlp_speed_on_image_batch():
    dev = torch.device("cuda:0")
    dec = BasicDecoder(32, 3, torch.relu, True, nn.Linear, 2, 16, []).to(dev)
    x = torch.randn(393216, 32, device=dev, requires_grad=True)
    gy = torch.randn(393216, 3, device=dev)

    def run(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    def fused():
        y = dec(x); y.backward(gy)

    def layers():
        y = _torch_mlp(dec, x); y.backward(gy)
    from shacira_amd import _lib
    tf, tl = run(fused), run(layers)
    _lib.set_option("mlp_variant", 0)
    try:
        tv = run(fused)
    finally:
        _lib.set_option("mlp_variant", -1)
    print(f"decoder MLP fwd+bwd on 393216 px: MFMA 16x16x4 {tf:.3f} ms, VALU kernels {tv:.3f} ms, torch Linear layers "
          f"{tl:.3f} ms")


def speed_mfma_mlp_speed_on_nerf_batch(dims):
    dev = torch.device("cuda:0")
    IN, H, NH, OUT = dims
    dec = BasicDecoder(IN, OUT, torch.relu, True, nn.Linear, NH, H, []).to(dev)
    n = 1 << 19
    x = torch.randn(n, IN, device=dev, requires_grad=True)
    gy = torch.randn(n, OUT, device=dev)

    def run(fn, iters=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    def fused():
        y = dec(x); y.backward(gy)

    def layers():
        y = _torch_mlp(dec, x); y.backward(gy)
    tf, tl = run(fused), run(layers)
    print(f"NeRF decoder {dims} fwd+bwd on {n} samples: MFMA kernels {tf:.3f} ms vs torch Linear layers {tl:.3f} ms")


def speed_fused_adam_throughput_on_table_sized_buffer():
    n = 6_098_925 * 2
    p = torch.nn.Parameter(torch.randn(n, device="cuda:0"))
    p.grad = torch.randn(n, device="cuda:0")
    def time(opt, iters=10):
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            opt.step()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters
    t_fused = time(FusedAdam([p], lr=1e-3))
    t_torch = time(torch.optim.Adam([p], lr=1e-3))
    print(f"adam over {n} params: fused {t_fused:.3f} ms ({n * 28 / t_fused / 1e6:.0f} GB/s) vs torch {t_torch:.3f} ms")


if __name__ == "__main__":
    speed_fused_mlp_speed_on_image_batch()
    for dims in ((32, 64, 1, 16), (43, 64, 2, 3)):
        speed_mfma_mlp_speed_on_nerf_batch(dims)
    speed_fused_adam_throughput_on_table_sized_buffer()
