"""``LatentDecoder`` / ``DecoderIdentity``: the latent quantise-and-decode module with the reference's constructor
arguments, parameter names (``div``, ``layers.0.scale``, ``layers.0.shift``, ``layers.0.dft``) and runtime switches
(``use_sga``, ``temperature``, ``diff_sampling``) -- reference wisp/models/latent_decoders/basic_latent_decoder.py:97-228.

Execution:
  * no hidden layers, identity activations (every shipped config), table on the GPU: ONE fused HIP kernel forward and
    one fused kernel + a tiny finishing kernel backward -- ``decode_layer._FusedLatentDecode`` for the rounding path
    (after the SGA warm-up, at validation, whenever SGA is off) and ``_FusedLatentDecodeSGA`` for the SGA path (the
    two uniforms per latent come from ``torch.rand`` on the device: the single draw the reference's
    RelaxedOneHotCategorical makes, so a seeded run consumes the generator exactly like the reference);
  * hidden layers and / or activations (``num_layers_dec > 0``, ``activation`` / ``final_activation`` != 'none'), table on
    the GPU, widths up to 16 and up to 4 layers: the per-row MLP kernel ``decode_layer._FusedLatentMLP`` (one kernel
    forward, one + a finishing kernel backward; the layers' effective matrices are packed by three tiny torch ops);
  * wider / deeper decoders and tensors that live on the host stay torch ops.
"""
import torch
import torch.nn as nn
from torch import Tensor
from torch.nn.modules.utils import _ntuple

from .... import hip_ops
from .decode_layer import (DecoderLayer, _FusedLatentDecode, _FusedLatentDecodeSGA, _FusedLatentMLP,  # noqa: F401
                           get_dft_matrix)
from .quantizers import StraightThrough, StraightThroughFloor, epsilon, sga_sample  # noqa: F401


class SineScaled(nn.Module):
    def __init__(self, w0=1.0):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        return torch.sin(self.w0 * x)


_ACTIVATIONS = {"none": nn.Identity, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "relu": nn.ReLU,
                "sine": lambda: SineScaled(30.0)}


class LatentDecoder(nn.Module):
    def __init__(self, latent_dim: int, feature_dim: int, norm: str, ldecode_matrix: str, use_shift: bool,
                 num_layers_dec: int = 0, hidden_dim_dec: int = 0, activation: str = "none",
                 final_activation: str = "none", clamp_weights: float = 0.0, ldec_std: float = 1.0,
                 use_sga: bool = False, diff_sampling: bool = False, **kwargs) -> None:
        super().__init__()
        latent_dim = latent_dim or feature_dim
        # plain attributes the trainers and LatentGrid read
        self.ldecode_matrix, self.channels, self.latent_dim, self.norm = ldecode_matrix, feature_dim, latent_dim, norm
        self.num_layers_dec, self.use_shift, self.clamp_weights = num_layers_dec, use_shift, clamp_weights
        self.use_sga, self.diff_sampling, self.temperature = use_sga, diff_sampling, 1.0
        self._identity_acts = activation == "none" and final_activation == "none"
        self._act_names = (activation, final_activation)
        # per-channel normaliser, maintained by the trainer (max-abs or std of the latents); never trained
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
            stack.append(DecoderLayer(fan_in, fan_out, ldecode_matrix, bias=use_shift))
            if k < num_layers_dec:
                stack.append(self.act)
        self.layers = nn.Sequential(*stack)
        self._widths = tuple(widths)
        self.reset_parameters("normal", ldec_std)

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

    def _first_scale(self, what):
        if self.num_layers_dec > 0:
            print("Warning: norm is not implemented for multiple layer decoder>0, returning default value 1")
            return None
        return what(self._decoder_layers()[0].scale)

    def scale_norm(self):
        r = self._first_scale(lambda s: s.norm())
        return 1 if r is None else r

    def scale_grad_norm(self):
        r = self._first_scale(lambda s: s.grad.norm())
        return 1 if r is None else r

    # -- forward ----------------------------------------------------------------------------------------------
    def _fusable(self, weight: Tensor) -> bool:
        return (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2
                and self.num_layers_dec == 0 and self._identity_acts
                and hip_ops.latent_decode_supported(self.latent_dim, self.channels))

    def _mlp_fusable(self, weight: Tensor) -> bool:
        """Hidden layers and / or activations: the per-row MLP kernel (widths up to 16, up to 4 layers)."""
        return (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_contiguous()
                and self.div.is_cuda and self.div.dtype == torch.float32     # (module.half() would make div fp16)
                and hip_ops.latent_mlp_supported(self._widths))

    def _packed_layers(self) -> Tensor:
        """Per layer: effective matrix [in, out] row-major, then the shift (zeros without one), as one fp32 vector."""
        parts = []
        for layer in self._decoder_layers():
            parts.append(layer.effective_matrix().reshape(-1))
            parts.append(layer.shift.reshape(-1) if layer.shift is not None
                         else torch.zeros(layer.out_features, dtype=layer.scale.dtype, device=layer.scale.device))
        return torch.cat(parts)

    def forward(self, weight: Tensor) -> Tensor:
        if self._fusable(weight):
            matrix, colscale, shift = self._decoder_layers()[0].fused_operands()
            if self.use_sga:
                uniforms = torch.rand(weight.shape + (2,), dtype=weight.dtype, device=weight.device)
                temperature = self.temperature if torch.is_tensor(self.temperature) else float(self.temperature)
                return _FusedLatentDecodeSGA.apply(weight, uniforms, temperature, bool(self.diff_sampling),
                                                   self.div, matrix, colscale, shift, float(self.clamp_weights))
            return _FusedLatentDecode.apply(weight, self.div, matrix, colscale, shift, float(self.clamp_weights))
        if self._mlp_fusable(weight):
            uniforms = None
            if self.use_sga:
                uniforms = torch.rand(weight.shape + (2,), dtype=weight.dtype, device=weight.device)
            return _FusedLatentMLP.apply(weight, uniforms, float(self.temperature), bool(self.diff_sampling), self.div,
                                         self._packed_layers(), self._widths, self._act_names[0], self._act_names[1],
                                         float(self.clamp_weights))
        if weight.is_cuda:
            hip_ops.warn_unfused("LatentDecoder.forward", f"decoder widths {self._widths} outside the fused kernels' limits")
        if self.use_sga:
            weight = sga_sample(weight, self.temperature, self.diff_sampling)
        else:
            weight = StraightThrough.apply(weight)
        w_out = self.final_activation(self.layers(weight / self.div))
        if self.clamp_weights > 0.0:
            w_out = torch.clamp(w_out, min=-self.clamp_weights, max=self.clamp_weights)
        return w_out


class DecoderIdentity(nn.Module):
    """Stand-in used when ``ldecode_enabled`` is False: the table is used as stored; same probe methods as a decoder."""

    latent_dim, num_layers_dec, shift, norm = 1, 0, False, "none"

    def reset_parameters(self, init_type, param=1.0) -> None:
        return None

    def forward(self, input: Tensor) -> Tensor:
        return input

    def scale_norm(self):
        return 1

    def scale_grad_norm(self):
        return 1

    def size(self, use_torchac=False) -> int:
        return 0
