from .metrics import psnr, clamped_psnr
