"""One ``LatentDecoder`` per level of the grid (reference wisp/models/latent_decoders/hierarchical_latent_decoder.py:3-36).

``offsets`` are the row boundaries of the levels inside the concatenated table. The reference builds them as
``cat(codebook_lod_first_idx, codebook_lod_sizes[-1:])`` (latent_grid.py:182), i.e. the final boundary is the LAST
LEVEL'S SIZE rather than the table's end -- a reference quirk kept by ``LatentGrid.setup_decoders`` here so that
results match; rows past that boundary are left as ``torch.empty`` leaves them in the reference and are zero here.

Execution: affine decoders without hidden layers on the GPU (every shipped configuration) run as ONE fused HIP kernel
each way for all levels (``shacira_latent_decode_levels_*``: the level's parameters are rows of stacked arrays), rounding
or SGA; anything else evaluates the per-level decoders one after the other, as the reference does.
"""
import torch
import torch.nn as nn

from .... import hip_ops
from .basic_latent_decoder import LatentDecoder


class _FusedLevelsDecode(torch.autograd.Function):
    """All levels' round / SGA -> /div -> affine -> clamp in one kernel each way; the per-level parameters arrive stacked
    ([L, ...]) so autograd hands every level's decoder its own slice of the stacked gradients."""

    @staticmethod
    def forward(ctx, latent, uniforms, div, matrix, colscale, shift, offsets, temperature, diff_sampling, clamp_weights):
        latent = latent.contiguous()
        ctx.save_for_backward(latent, uniforms, div, matrix, colscale, shift)
        ctx.opts = (offsets, float(temperature), bool(diff_sampling), float(clamp_weights))
        return hip_ops.latent_decode_levels_forward(latent, offsets, uniforms, temperature, diff_sampling, div, matrix,
                                                    colscale, shift, clamp_weights)

    @staticmethod
    def backward(ctx, grad_decoded):
        latent, uniforms, div, matrix, colscale, shift = ctx.saved_tensors
        offsets, temperature, diff_sampling, clamp_weights = ctx.opts
        g_lat, g_mat, g_cs, g_sh = hip_ops.latent_decode_levels_backward(
            latent, offsets, uniforms, temperature, diff_sampling, div, matrix, colscale, shift, clamp_weights,
            grad_decoded.contiguous())
        return (g_lat if ctx.needs_input_grad[0] else None, None, None,
                g_mat if ctx.needs_input_grad[3] else None,
                g_cs if (colscale is not None and ctx.needs_input_grad[4]) else None,
                g_sh if (shift is not None and ctx.needs_input_grad[5]) else None, None, None, None, None)


class HierarchicalLatentDecoder(nn.Module):
    def __init__(self, num_decoders, offsets, conf_decoder):
        super().__init__()
        self.num_decoders = num_decoders
        self.decoders = nn.ModuleList([LatentDecoder(**conf_decoder) for _ in range(num_decoders)])
        self.offsets = offsets

    def _fusable(self, input):
        d0 = self.decoders[0]
        return (d0._fusable(input) and 1 <= self.num_decoders <= 32
                and all(d.num_layers_dec == 0 and d._identity_acts and d.ldecode_matrix == d0.ldecode_matrix
                        and d.use_shift == d0.use_shift and d.clamp_weights == d0.clamp_weights
                        and d.use_sga == d0.use_sga and d.diff_sampling == d0.diff_sampling
                        and d.temperature == d0.temperature for d in self.decoders))

    def forward(self, input):
        bounds = [int(o) for o in self.offsets]
        if self._fusable(input) and all(bounds[l] <= bounds[l + 1] for l in range(self.num_decoders - 1)):
            d0 = self.decoders[0]
            ops = [d._decoder_layers()[0].fused_operands() for d in self.decoders]
            div = torch.stack([d.div for d in self.decoders])
            matrix = torch.stack([m for m, _, _ in ops])                                   # [L, ld, F]
            colscale = torch.stack([c.reshape(-1) for _, c, _ in ops]) if ops[0][1] is not None else None
            shift = torch.stack([s.reshape(-1) for _, _, s in ops]) if ops[0][2] is not None else None
            uniforms = None
            if d0.use_sga:
                # the sampler's uniforms drawn PER LEVEL, in level order, exactly as the per-level decoders of the unfused
                # path (and the reference) consume the generator: a seeded run sees the same noise on either path. Rows no
                # level owns decode to 0 and draw nothing.
                uniforms = torch.zeros(input.shape + (2,), dtype=input.dtype, device=input.device)
                for l in range(self.num_decoders):
                    lo, hi = bounds[l], bounds[l + 1]
                    if hi > lo:
                        uniforms[lo:hi] = torch.rand((hi - lo,) + tuple(input.shape[1:]) + (2,), dtype=input.dtype,
                                                     device=input.device)
            return _FusedLevelsDecode.apply(input, uniforms, div, matrix, colscale, shift, tuple(bounds),
                                            float(d0.temperature), bool(d0.diff_sampling), float(d0.clamp_weights))
        pieces, row = [], 0
        for l in range(self.num_decoders):
            lo, hi = bounds[l], bounds[l + 1]
            if hi <= lo:
                continue
            if lo > row:  # rows no decoder owns
                pieces.append(input.new_zeros((lo - row, self.decoders[0].channels)))
            pieces.append(self.decoders[l](input[lo:hi]))
            row = hi
        if row < input.size(0):
            pieces.append(input.new_zeros((input.size(0) - row, self.decoders[0].channels)))
        return torch.cat(pieces, dim=0)

    @property
    def temperature(self):
        return self.decoders[0].temperature

    @temperature.setter
    def temperature(self, value):
        for d in self.decoders:
            d.temperature = value

    @property
    def use_sga(self):
        return self.decoders[0].use_sga

    @use_sga.setter
    def use_sga(self, value):
        for d in self.decoders:
            d.use_sga = value

    def size(self, use_torchac=False):
        return sum(p.numel() * torch.finfo(p.dtype).bits for p in self.parameters())
