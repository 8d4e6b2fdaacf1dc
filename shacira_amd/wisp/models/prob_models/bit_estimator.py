"""Factorised entropy bottleneck (Balle-style CDF) with the reference's module/parameter names
(``f1..f4`` each holding ``h``, ``b`` and, except the final ``f4``, ``a`` of shape [1, channel]) --
reference wisp/models/prob_models/bit_estimator.py:9-65.

``BitEstimator.forward`` evaluates the CDF with torch ops (it is called directly only on a handful of unique
values by ``LatentGrid.size(use_prob_model=True)``). The per-step hot use -- the entropy of the whole latent
table in ``LatentGrid.ent_loss`` -- goes through ``BitEstimator.total_bits``, one fused HIP kernel forward and
one backward (``shacira_entropy_bits_{forward,backward}``) when the table lives on the GPU.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import hip_ops


class Bitparm(nn.Module):
    """One layer of the cumulative: x*softplus(h)+b, then x+tanh(x)*tanh(a) (sigmoid instead for the final layer)."""

    def __init__(self, channel, is_symmetric=False, is_unimodal=False, final=False):
        super().__init__()
        self.final = final
        self.is_unimodal = is_unimodal
        self.h = nn.Parameter(torch.nn.init.normal_(torch.empty(channel).view(1, -1), 0, 0.01))
        if is_symmetric:
            self.b = nn.Parameter(torch.nn.init.zeros_(torch.empty(channel).view(1, -1)), requires_grad=False)
        else:
            self.b = nn.Parameter(torch.nn.init.normal_(torch.empty(channel).view(1, -1), 0, 0.01))
        self.a = None if final else nn.Parameter(torch.nn.init.normal_(torch.empty(channel).view(1, -1), 0, 0.01))

    def forward(self, x, single_channel=None):
        pick = (lambda p: p) if single_channel is None else (lambda p: p[:, single_channel])
        x = x * F.softplus(pick(self.h)) + pick(self.b)
        if self.final:
            return torch.sigmoid(x)
        a = pick(self.a)
        if self.is_unimodal:
            a = torch.abs(a)
        return x + torch.tanh(x) * torch.tanh(a)


class _FusedEntropyBits(torch.autograd.Function):
    """sum over the table of clamp(-log2(CDF(w+.5) - CDF(w-.5) + 1e-10), 0, 50); w = latent+noise or round(latent)."""

    @staticmethod
    def forward(ctx, latent, noise, params, num_layers):
        latent = latent.contiguous()
        noise = noise.contiguous() if noise is not None else None
        params = params.contiguous()
        ctx.save_for_backward(latent, noise, params)
        ctx.num_layers = num_layers
        return hip_ops.entropy_bits_forward(latent, noise, params, num_layers)

    @staticmethod
    def backward(ctx, grad_total):
        latent, noise, params = ctx.saved_tensors
        need_lat = ctx.needs_input_grad[0]
        g_lat, g_par = hip_ops.entropy_bits_backward(latent, noise, params, ctx.num_layers,
                                                     grad_total.contiguous().float(), need_latent=need_lat)
        return (g_lat if need_lat else None, None, g_par if ctx.needs_input_grad[2] else None, None)


class BitEstimator(nn.Module):
    def __init__(self, channel, is_symmetric=False, is_unimodal=False, num_layers=4):
        super().__init__()
        self.num_layers = num_layers
        self._plain = not is_unimodal
        self.f1 = Bitparm(channel, is_symmetric=is_symmetric, is_unimodal=is_unimodal)
        self.f2 = Bitparm(channel, is_symmetric=is_symmetric, is_unimodal=is_unimodal)
        self.f3 = Bitparm(channel, is_symmetric=is_symmetric, is_unimodal=is_unimodal)
        self.f4 = Bitparm(channel, is_symmetric=is_symmetric, is_unimodal=is_unimodal, final=True)

    def forward(self, x, single_channel=None):
        for k, f in enumerate((self.f1, self.f2, self.f3)):
            if self.num_layers > k + 1:
                x = f(x, single_channel)
        return self.f4(x, single_channel)

    # ------------------------------------------------------------------------------------------------------
    def packed_params(self):
        """[4, 3, channel] = (f1.h, f1.b, f1.a, ..., f4.h, f4.b, 0) -- the C-ABI's parameter block (differentiable)."""
        pieces = []
        for f in (self.f1, self.f2, self.f3, self.f4):
            pieces += [f.h, f.b, f.a if f.a is not None else torch.zeros_like(f.h)]
        return torch.cat(pieces, dim=0).view(4, 3, -1)     # one cat kernel; its backward hands out row views

    def total_bits(self, latent, noise=None):
        """Entropy of the whole table in bits (scalar tensor). ``noise`` None means the validation rule round()."""
        if (latent.is_cuda and latent.dtype == torch.float32 and latent.dim() == 2 and self._plain
                and hip_ops.entropy_supported(latent.shape[1])):
            return _FusedEntropyBits.apply(latent, noise, self.packed_params(), self.num_layers)
        if latent.is_cuda:
            hip_ops.warn_unfused("BitEstimator.total_bits", "unsupported latent_dim / dtype, or is_symmetric / is_unimodal")
        weight = (latent + noise) if noise is not None else torch.round(latent)
        prob = self(weight + 0.5) - self(weight - 0.5)
        return torch.sum(torch.clamp(-1.0 * torch.log(prob + 1e-10) / math.log(2.0), 0, 50))
