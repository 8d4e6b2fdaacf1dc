"""``MultiLatentDecoder``: K affine decoders mixed per table entry by a learned selector ``alpha`` [K, T]
(softmax with temperature, or its straight-through arg-max one-hot) -- "next" row f4 of SURVEY.md section 8.

Same constructor arguments, parameter names (``div``, ``alpha``, ``layers.N.scale``, ``layers.N.use_shift``,
``layers.N.dft``), runtime switches (``temperature``, ``straight_through``, ``use_sga``, ``diff_sampling``) and
``size()`` accounting as reference wisp/models/latent_decoders/multi_latent_decoder.py:27-210. Without hidden layers, on the
GPU, the whole forward (softmax / arg-max selector, rounding or SGA, the K decoders, the mixing) is ONE fused HIP kernel
each way (``shacira_latent_multi_decode_*``); otherwise torch ops, as in the reference.

Reference quirks kept on purpose (they change the numbers):
  * the 'sq' branch mixes twice: ``sum_k alpha_k (x @ scale_k)`` is formed, the shift is added, and the result is
    multiplied by alpha and summed over k AGAIN (multi_latent_decoder.py:74-81);
  * the bias parameter is called ``use_shift``.
"""
import math

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import Module, Parameter, init
from torch.nn.modules.utils import _ntuple

from .... import hip_ops
from .basic_latent_decoder import _ACTIVATIONS
from .decode_layer import get_dft_matrix
from .quantizers import StraightThrough, sga_sample


class StraightThroughOneHot(torch.autograd.Function):
    """arg-max over the decoder axis as a one-hot [K, T]; gradient passes through unchanged."""

    @staticmethod
    def forward(ctx, x):
        winner = torch.argmax(x, dim=0)
        return torch.nn.functional.one_hot(winner, num_classes=x.size(0)).permute(1, 0).to(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class _FusedMultiDecode(torch.autograd.Function):
    """selector softmax / arg-max + round or SGA + K affine decoders + mixing (+ clamp), one HIP kernel each way."""

    @staticmethod
    def forward(ctx, latent, alpha, uniforms, div, scale, dft, shift, temperature, straight_through, diff_sampling,
                clamp_weights):
        latent, alpha, scale = latent.contiguous(), alpha.contiguous(), scale.contiguous()
        shift2 = shift.reshape(shift.shape[0], -1).contiguous() if shift is not None else None
        ctx.save_for_backward(latent, alpha, uniforms, div, scale, dft, shift2)
        ctx.opts = (float(temperature), bool(straight_through), bool(diff_sampling), float(clamp_weights))
        ctx.shift_shape = shift.shape if shift is not None else None
        return hip_ops.latent_multi_decode_forward(latent, alpha, uniforms, temperature, straight_through, diff_sampling,
                                                   div, scale, dft, shift2, clamp_weights)

    @staticmethod
    def backward(ctx, grad_decoded):
        latent, alpha, uniforms, div, scale, dft, shift2 = ctx.saved_tensors
        temperature, straight_through, diff_sampling, clamp_weights = ctx.opts
        g_lat, g_alpha, g_scale, g_shift = hip_ops.latent_multi_decode_backward(
            latent, alpha, uniforms, temperature, straight_through, diff_sampling, div, scale, dft, shift2,
            clamp_weights, grad_decoded.contiguous())
        return (g_lat if ctx.needs_input_grad[0] else None, g_alpha if ctx.needs_input_grad[1] else None, None, None,
                g_scale if ctx.needs_input_grad[4] else None, None,
                g_shift.reshape(ctx.shift_shape) if (g_shift is not None and ctx.needs_input_grad[6]) else None,
                None, None, None, None)


class MultiLatentDecoderLayer(Module):
    def __init__(self, in_features: int, out_features: int, ldecode_matrix: str, num_decoders: int = 1,
                 bias: bool = False) -> None:
        super().__init__()
        self.in_features, self.out_features, self.ldecode_matrix = in_features, out_features, ldecode_matrix
        is_dft = "dft" in ldecode_matrix
        if is_dft:
            self.dft = Parameter(get_dft_matrix(in_features, out_features), requires_grad=False)
        self.scale = Parameter(torch.empty((num_decoders, 1 if is_dft else in_features, out_features)))
        if bias:
            self.use_shift = Parameter(torch.empty(num_decoders, 1, out_features))
        else:
            self.register_parameter("use_shift", None)
        if ldecode_matrix == "dft_fixed":
            self.scale.requires_grad_(False)
            if not bias:
                self.use_shift.requires_grad_(False)  # (sic) as in the reference: raises without a bias

    def reset_parameters(self, param=1.0, init_type="normal") -> None:
        if init_type == "normal":
            init.normal_(self.scale, std=param)
        elif init_type == "uniform":
            init.uniform_(self.scale, -param, param)
        elif init_type == "constant":
            init.constant_(self.scale, val=param)
        if self.use_shift is not None:
            init.zeros_(self.use_shift)

    def clamp(self, val: float = 0.5) -> None:
        with torch.no_grad():
            self.scale.clamp_(-val, val)

    def forward(self, input: Tensor, alpha: Tensor) -> Tensor:
        shift = self.use_shift if self.use_shift is not None else 0
        a = alpha.unsqueeze(-1)                                           # [K, T, 1]
        if "dft" in self.ldecode_matrix:
            per_decoder = torch.matmul(input, self.dft).unsqueeze(0) * self.scale + shift   # [K, T, out]
        else:
            mixed = torch.sum(torch.matmul(input.unsqueeze(0), self.scale) * a, dim=0)    # [T, out]
            per_decoder = mixed + shift                                                   # broadcast to [K, T, out]
        return torch.sum(per_decoder * a, dim=0)

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}".format(self.in_features, self.out_features,
                                                                  self.use_shift is not None)


class MultiSequential(nn.Sequential):
    def forward(self, input, alpha):
        for module in self._modules.values():
            input = module(input, alpha) if isinstance(module, MultiLatentDecoderLayer) else module(input)
        return input


class MultiLatentDecoder(Module):
    def __init__(self, latent_dim: int, feature_dim: int, norm: str, ldecode_matrix: str, use_shift: bool,
                 num_entries: int, num_layers_dec: int = 0, hidden_dim_dec: int = 0, activation: str = "none",
                 final_activation: str = "none", clamp_weights: float = 0.0, ldec_std: float = 1.0,
                 num_decoders: int = 1, alpha_std: float = 1.0, use_sga: bool = False, **kwargs) -> None:
        super().__init__()
        latent_dim = latent_dim or feature_dim
        self.ldecode_matrix, self.channels, self.latent_dim, self.norm = ldecode_matrix, feature_dim, latent_dim, norm
        self.num_layers_dec, self.use_shift, self.clamp_weights = num_layers_dec, use_shift, clamp_weights
        self.num_decoders = num_decoders
        self.div = nn.Parameter(torch.ones(latent_dim), requires_grad=False)
        self.act = _ACTIVATIONS[activation]()
        self.final_activation = _ACTIVATIONS[final_activation]()
        widths = [latent_dim]
        if num_layers_dec > 0:
            self.hidden_dim_dec = _ntuple(num_layers_dec)(hidden_dim_dec or feature_dim)
            for h in self.hidden_dim_dec:
                widths.append(h or widths[-1])
        widths.append(feature_dim)
        stack = []
        for k, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            stack.append(MultiLatentDecoderLayer(fan_in, fan_out, ldecode_matrix, num_decoders=num_decoders,
                                                 bias=use_shift))
            if k < num_layers_dec:
                stack.append(self.act)
        # RNG order of the reference: alpha is drawn BEFORE the layer scales are (re)initialised
        self.alpha = nn.Parameter(torch.randn(num_decoders, num_entries) * alpha_std, requires_grad=True)
        self.temperature = 1.0
        self.layers = MultiSequential(*stack)
        self.reset_parameters("normal", ldec_std)
        self.straight_through = True
        self.use_sga = use_sga
        self.diff_sampling = False

    def _decoder_layers(self):
        return [m for m in self.layers.children() if isinstance(m, MultiLatentDecoderLayer)]

    def reset_parameters(self, init_type, param=0.5) -> None:
        for layer in self._decoder_layers():
            layer.reset_parameters(param, init_type)

    def get_scale(self):
        assert self.num_layers_dec == 0, "Can only get scale for 0 hidden layers decoder!"
        return self._decoder_layers()[0].scale

    def clamp(self, val: float = 0.2) -> None:
        for layer in self._decoder_layers():
            layer.clamp(val)

    def size(self, use_torchac=False):
        """fp32 bits of everything but ``alpha`` + the empirical entropy of the arg-max selector per entry."""
        fp_size = sum(p.numel() * torch.finfo(p.dtype).bits for n, p in self.named_parameters() if "alpha" not in n)
        if use_torchac:
            raise NotImplementedError("torchac arithmetic coding is not available; use use_torchac=False")
        _, counts = torch.unique(torch.argmax(self.alpha, dim=0), return_counts=True)
        probs = counts / torch.sum(counts)
        bits = torch.clamp(-1.0 * torch.log(probs + 1e-10) / math.log(2.0), 0, 1000)
        return torch.sum(bits * counts).item() + fp_size

    def _fusable(self, weight: Tensor) -> bool:
        return (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and self.num_layers_dec == 0
                and isinstance(self.act, nn.Identity) and isinstance(self.final_activation, nn.Identity)
                and self.alpha.shape[1] == weight.shape[0]
                and hip_ops.latent_multi_supported(self.latent_dim, self.channels, self.num_decoders))

    def forward(self, weight: Tensor) -> Tensor:
        if self._fusable(weight):
            layer = self._decoder_layers()[0]
            dft = layer.dft if "dft" in self.ldecode_matrix else None
            uniforms = None
            if self.use_sga:
                uniforms = torch.rand(weight.shape + (2,), dtype=weight.dtype, device=weight.device)
            return _FusedMultiDecode.apply(weight, self.alpha, uniforms, self.div, layer.scale, dft, layer.use_shift,
                                           float(self.temperature), bool(self.straight_through),
                                           bool(self.diff_sampling), float(self.clamp_weights))
        alpha = nn.functional.softmax(self.alpha / self.temperature, dim=0)
        if self.straight_through:
            alpha = StraightThroughOneHot.apply(alpha)
        weight = sga_sample(weight, self.temperature, self.diff_sampling) if self.use_sga \
            else StraightThrough.apply(weight)
        w_out = self.final_activation(self.layers(weight / self.div, alpha))
        if self.clamp_weights > 0.0:
            w_out = torch.clamp(w_out, min=-self.clamp_weights, max=self.clamp_weights)
        return w_out
