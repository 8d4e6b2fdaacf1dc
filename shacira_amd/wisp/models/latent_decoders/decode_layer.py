"""The affine decode layer of the latent path and its fused HIP form.

``DecoderLayer`` keeps the reference's parameter names and shapes (``scale`` [in,out] for 'sq', [1,out] + frozen
``dft`` [in,out] for 'dft*', optional ``shift`` [1,out]) -- reference basic_latent_decoder.py:12-19, :48-95.
``_FusedLatentDecode`` is round -> /div -> affine -> clamp as one kernel each way (shacira_latent_decode_*)."""
import math

import torch
from torch import Tensor
from torch.nn import Module, Parameter, init

from .... import hip_ops


def get_dft_matrix(conv_dim, channels):
    """DCT-II style basis, one row per latent channel: cos(pi/C (i+1/2) j) / sqrt(C), times sqrt(2) for j > 0."""
    # evaluated per element in Python doubles and narrowed to fp32, then scaled in fp32 (as the reference's
    # element-wise assignment does), so the basis is bit-identical
    dft = torch.zeros(conv_dim, channels)
    root2 = torch.tensor(math.sqrt(2), dtype=torch.float32)
    for i in range(conv_dim):
        for j in range(channels):
            dft[i, j] = math.cos(math.pi / channels * (i + 0.5) * j) / math.sqrt(channels)
    dft[:, 1:] = dft[:, 1:] * root2
    return dft


class _FusedLatentDecode(torch.autograd.Function):
    """round -> /div -> (z @ matrix) * colscale + shift -> clamp, as one HIP kernel each way."""

    @staticmethod
    def forward(ctx, latent, div, matrix, colscale, shift, clamp_weights):
        latent = latent.contiguous()
        ctx.save_for_backward(latent, div, matrix, colscale, shift)
        ctx.clamp_weights = clamp_weights
        return hip_ops.latent_decode_forward(latent, div, matrix.contiguous(), colscale, shift, clamp_weights)

    @staticmethod
    def backward(ctx, grad_decoded):
        latent, div, matrix, colscale, shift = ctx.saved_tensors
        g_lat, g_mat, g_cs, g_sh = hip_ops.latent_decode_backward(
            latent, div, matrix.contiguous(), colscale, shift, ctx.clamp_weights, grad_decoded.contiguous(),
            need_colscale=colscale is not None)
        return (g_lat if ctx.needs_input_grad[0] else None, None,
                g_mat if ctx.needs_input_grad[2] else None,
                g_cs.reshape(colscale.shape) if (colscale is not None and ctx.needs_input_grad[3]) else None,
                g_sh.reshape(shift.shape) if (shift is not None and ctx.needs_input_grad[4]) else None, None)


class _FusedLatentDecodeSGA(torch.autograd.Function):
    """SGA sample between floor and ceil (uniforms drawn by the caller) -> /div -> decode, one HIP kernel each way."""

    @staticmethod
    def forward(ctx, latent, uniforms, temperature, diff_sampling, div, matrix, colscale, shift, clamp_weights):
        latent = latent.contiguous()
        ctx.save_for_backward(latent, uniforms, div, matrix, colscale, shift)
        # (a tensor temperature = one fp32 value on the device, read by the kernels: graph-captured steps anneal it in place)
        ctx.opts = (temperature if torch.is_tensor(temperature) else float(temperature), bool(diff_sampling), clamp_weights)
        return hip_ops.latent_decode_sga_forward(latent, uniforms, temperature, diff_sampling, div, matrix.contiguous(),
                                                 colscale, shift, clamp_weights)

    @staticmethod
    def backward(ctx, grad_decoded):
        latent, uniforms, div, matrix, colscale, shift = ctx.saved_tensors
        temperature, diff_sampling, clamp_weights = ctx.opts
        g_lat, g_mat, g_cs, g_sh = hip_ops.latent_decode_sga_backward(
            latent, uniforms, temperature, diff_sampling, div, matrix.contiguous(), colscale, shift, clamp_weights,
            grad_decoded.contiguous(), need_colscale=colscale is not None)
        return (g_lat if ctx.needs_input_grad[0] else None, None, None, None, None,
                g_mat if ctx.needs_input_grad[5] else None,
                g_cs.reshape(colscale.shape) if (colscale is not None and ctx.needs_input_grad[6]) else None,
                g_sh.reshape(shift.shape) if (shift is not None and ctx.needs_input_grad[7]) else None, None)


class _FusedLatentMLP(torch.autograd.Function):
    """quantise (round or SGA) -> /div -> decoder layers with activations -> final activation -> clamp: the hidden-layer
    decoder of basic_latent_decoder.py:139-147,182-198 as one HIP kernel each way (shacira_latent_mlp_*). ``params`` packs,
    per layer, the effective matrix [in, out] and the shift [out]; its gradient flows on to ``scale`` / ``shift`` through the
    packing ops of ``LatentDecoder._packed_layers``."""

    @staticmethod
    def forward(ctx, latent, uniforms, temperature, diff_sampling, div, params, widths, activation, final_activation,
                clamp_weights):
        latent, params = latent.contiguous(), params.contiguous()
        ctx.save_for_backward(latent, uniforms, div, params)
        ctx.opts = (float(temperature), bool(diff_sampling), tuple(widths), activation, final_activation,
                    float(clamp_weights))
        return hip_ops.latent_mlp_forward(latent, uniforms, temperature, diff_sampling, div, params, tuple(widths), activation,
                                          final_activation, clamp_weights)

    @staticmethod
    def backward(ctx, grad_decoded):
        latent, uniforms, div, params = ctx.saved_tensors
        temperature, diff_sampling, widths, activation, final_activation, clamp_weights = ctx.opts
        g_lat, g_par = hip_ops.latent_mlp_backward(latent, uniforms, temperature, diff_sampling, div, params, widths,
                                                   activation, final_activation, clamp_weights, grad_decoded.contiguous())
        return (g_lat if ctx.needs_input_grad[0] else None, None, None, None, None,
                g_par if ctx.needs_input_grad[5] else None, None, None, None, None)


class DecoderLayer(Module):
    """One affine decode layer: 'sq' learns the full [in, out] matrix; 'dft*' fixes a DCT basis and learns a
    per-output scale. ``shift`` exists only with ``bias=True``."""

    def __init__(self, in_features: int, out_features: int, ldecode_matrix: str, bias: bool = False) -> None:
        super().__init__()
        self.in_features, self.out_features, self.ldecode_matrix = in_features, out_features, ldecode_matrix
        is_dft = "dft" in ldecode_matrix
        if is_dft:
            self.dft = Parameter(get_dft_matrix(in_features, out_features), requires_grad=False)
        self.scale = Parameter(torch.empty((1, out_features) if is_dft else (in_features, out_features)))
        if bias:
            self.shift = Parameter(torch.empty(1, out_features))
        else:
            self.register_parameter("shift", None)
        if ldecode_matrix == "dft_fixed":
            self.scale.requires_grad_(False)
            if not bias:
                self.shift.requires_grad_(False)  # (sic) reference quirk: raises when there is no shift

    def reset_parameters(self, param=1.0, init_type="normal") -> None:
        if init_type == "normal":
            init.normal_(self.scale, std=param)
        elif init_type == "uniform":
            init.uniform_(self.scale, -param, param)
        elif init_type == "constant":
            init.constant_(self.scale, val=param)
        if self.shift is not None:
            init.zeros_(self.shift)

    def clamp(self, val: float = 0.5) -> None:
        with torch.no_grad():
            self.scale.clamp_(-val, val)

    def effective_matrix(self):
        """[in, out] matrix the layer multiplies by: ``scale``, or the fixed basis times the per-output scale."""
        return self.dft * self.scale if "dft" in self.ldecode_matrix else self.scale

    def fused_operands(self):
        """(matrix, colscale, shift) in the C-ABI's convention."""
        if "dft" in self.ldecode_matrix:
            return self.dft, self.scale, self.shift
        return self.scale, None, self.shift

    def forward(self, input: Tensor) -> Tensor:
        shift = self.shift if self.shift is not None else 0
        if "dft" in self.ldecode_matrix:
            return torch.matmul(input, self.dft) * self.scale + shift
        return torch.matmul(input, self.scale) + shift

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}".format(self.in_features, self.out_features,
                                                                  self.shift is not None)


