#!/usr/bin/env python3
"""Kernel-level timing target for the decoder MLPs (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import hip_ops, _lib
dev = torch.device("cuda:0")
for dims, n in (((32, 16, 2, 3), 393216), ((32, 64, 1, 16), 1 << 19), ((43, 64, 2, 3), 1 << 19),
                ((96, 128, 1, 16), 409600), ((43, 128, 2, 3), 409600)):
    IN, H, NH, OUT = dims
    npar = sum((IN if l == 0 else H) * H + H for l in range(NH)) + OUT * H + OUT
    x = torch.randn(n, IN, device=dev); p = torch.randn(npar, device=dev) * 0.2; gy = torch.randn(n, OUT, device=dev)
    for variant in ((-1, 0) if H == 16 else (-1,)):
        _lib.set_option("mlp_variant", variant)
        for _ in range(10):
            hip_ops.mlp_forward(x, p, *dims)
            hip_ops.mlp_backward(x, p, gy, *dims)
    _lib.set_option("mlp_variant", -1)
torch.cuda.synchronize()
