"""Latent quantise-and-decode modules with the reference's names, constructor arguments, parameter names
(``div``, ``layers.0.scale``, ``layers.0.shift``, ``layers.0.dft``) and runtime switches (``use_sga``,
``temperature``, ``diff_sampling``)  -- reference wisp/models/latent_decoders/basic_latent_decoder.py:12-228.

Execution:
  * deterministic path (``use_sga`` False, no hidden layers, identity activations -- what every shipped config
    runs after the SGA warm-up, at validation, and whenever SGA is off): ONE fused HIP kernel forward
    (round-half-even -> /div -> affine decode -> optional clamp) and one fused kernel + a tiny finishing kernel
    backward (``shacira_latent_decode_{forward,backward}``), when the table lives on the GPU;
  * SGA sampling (draws Gumbel noise from torch's generator), hidden layers and non-identity activations stay
    as torch ops (SURVEY.md section 7 "hard parts"), as do tensors that live on the host.
"""
import math

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import Module, Parameter, init
from torch.nn.modules.utils import _ntuple

from .... import hip_ops

epsilon = 1e-6


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


class SineScaled(nn.Module):
    def __init__(self, w0=1.0):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        return torch.sin(self.w0 * x)


class StraightThrough(torch.autograd.Function):
    """round() forward (half to even, ``torch.round``), identity backward."""

    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class StraightThroughFloor(torch.autograd.Function):
    """floor() forward, identity backward."""

    @staticmethod
    def forward(ctx, x):
        return torch.floor(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


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


_ACTIVATIONS = {"none": nn.Identity, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "relu": nn.ReLU,
                "sine": lambda: SineScaled(30.0)}


class LatentDecoder(Module):
    def __init__(self, latent_dim: int, feature_dim: int, norm: str, ldecode_matrix: str, use_shift: bool,
                 num_layers_dec: int = 0, hidden_dim_dec: int = 0, activation: str = "none",
                 final_activation: str = "none", clamp_weights: float = 0.0, ldec_std: float = 1.0,
                 use_sga: bool = False, diff_sampling: bool = False, **kwargs) -> None:
        super().__init__()
        latent_dim = feature_dim if latent_dim == 0 else latent_dim
        self.ldecode_matrix = ldecode_matrix
        self.channels = feature_dim
        self.latent_dim = latent_dim
        self.norm = norm
        self.div = nn.Parameter(torch.ones(latent_dim), requires_grad=False)
        self.num_layers_dec = num_layers_dec
        if num_layers_dec > 0:
            self.hidden_dim_dec = _ntuple(num_layers_dec)(feature_dim if hidden_dim_dec == 0 else hidden_dim_dec)
        self.use_shift = use_shift
        self.act = _ACTIVATIONS[activation]()
        self.final_activation = _ACTIVATIONS[final_activation]()
        self._identity_acts = activation == "none" and final_activation == "none"
        self.clamp_weights = clamp_weights

        layers, width = [], latent_dim
        for l in range(num_layers_dec):
            hidden = self.hidden_dim_dec[l] or width
            layers += [DecoderLayer(width, hidden, ldecode_matrix, bias=use_shift), self.act]
            width = hidden
        layers.append(DecoderLayer(width, self.channels, ldecode_matrix, bias=use_shift))

        self.use_sga = use_sga
        self.temperature = 1.0
        self.layers = nn.Sequential(*layers)
        self.reset_parameters("normal", ldec_std)
        self.diff_sampling = diff_sampling

    # -- reference helper surface ---------------------------------------------------------------------------
    def _decoder_layers(self):
        return [m for m in self.layers.children() if isinstance(m, DecoderLayer)]

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
        return sum(p.numel() * torch.finfo(p.dtype).bits for p in self.parameters())

    def scale_norm(self):
        if self.num_layers_dec > 0:
            print("Warning: norm is not implemented for multiple layer decoder>0, returning default value 1")
            return 1
        return self._decoder_layers()[0].scale.norm()

    def scale_grad_norm(self):
        if self.num_layers_dec > 0:
            print("Warning: norm is not implemented for multiple layer decoder>0, returning default value 1")
            return 1
        return self._decoder_layers()[0].scale.grad.norm()

    # -- forward ----------------------------------------------------------------------------------------------
    def _fusable(self, weight: Tensor) -> bool:
        return (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2
                and not self.use_sga and self.num_layers_dec == 0 and self._identity_acts
                and hip_ops.latent_decode_supported(self.latent_dim, self.channels))

    def _sga_sample(self, weight: Tensor) -> Tensor:
        """Stochastic Gumbel annealing between floor and ceil (reference :183-191); torch ops, torch's RNG."""
        lo = torch.floor(weight) if self.diff_sampling else StraightThroughFloor.apply(weight)
        hi = lo + 1
        lim = 1 - epsilon
        logit_lo = -torch.tanh(torch.clamp(weight - lo, min=-lim, max=lim)).unsqueeze(-1) / self.temperature
        logit_hi = -torch.tanh(torch.clamp(hi - weight, min=-lim, max=lim)).unsqueeze(-1) / self.temperature
        dist = torch.distributions.relaxed_categorical.RelaxedOneHotCategorical(
            self.temperature, logits=torch.cat((logit_lo, logit_hi), dim=-1))
        sample = dist.rsample() if self.diff_sampling else dist.sample()
        return lo * sample[..., 0] + hi * sample[..., 1]

    def forward(self, weight: Tensor) -> Tensor:
        if self._fusable(weight):
            matrix, colscale, shift = self._decoder_layers()[0].fused_operands()
            return _FusedLatentDecode.apply(weight, self.div, matrix, colscale, shift, float(self.clamp_weights))
        weight = self._sga_sample(weight) if self.use_sga else StraightThrough.apply(weight)
        w_out = self.final_activation(self.layers(weight / self.div))
        if self.clamp_weights > 0.0:
            w_out = torch.clamp(w_out, min=-self.clamp_weights, max=self.clamp_weights)
        return w_out


class DecoderIdentity(Module):
    """Placeholder used when ``ldecode_enabled`` is False: the table is used as stored."""

    def __init__(self) -> None:
        super().__init__()
        self.latent_dim = 1
        self.num_layers_dec = 0
        self.shift = False
        self.norm = "none"

    def reset_parameters(self, init_type, param=1.0) -> None:
        return

    def forward(self, input: Tensor) -> Tensor:
        return input

    def scale_norm(self):
        return 1

    def scale_grad_norm(self):
        return 1

    def size(self, use_torchac=False) -> int:
        return 0
