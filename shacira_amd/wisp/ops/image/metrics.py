"""PSNR metrics with the reference's definitions (wisp/ops/image/metrics.py:19-58)."""
import math

import torch


def psnr(rgb, gts):
    """10*log10(1/MSE) for images in [0, 1], shape [..., 3]."""
    assert rgb.max() <= 1.05 and rgb.min() >= -0.05
    assert gts.max() <= 1.05 and gts.min() >= -0.05
    assert rgb.shape[-1] == 3 and gts.shape[-1] == 3
    mse = torch.mean((rgb[..., :3] - gts[..., :3]) ** 2).item()
    return 10 * math.log10(1.0 / mse)


def clamped_psnr(rgb, gts):
    """PSNR after clamping to [0,1] and truncating to uint8 levels, as the image trainer reports it."""
    assert gts.max() <= 1.05 and gts.min() >= -0.05
    assert rgb.shape[-1] == 3 and gts.shape[-1] == 3
    q = lambda t: (torch.clamp(t, 0, 1) * 255).to(torch.uint8)[..., :3].float()
    mse = torch.mean((q(rgb) - q(gts)) ** 2).item()
    return 20 * math.log10(255.0) - 10 * math.log10(mse)
