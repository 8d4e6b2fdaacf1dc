"""``BasicDecoder``: the small MLP that turns interpolated grid features into colours / densities.

Same constructor arguments and attribute names (``layers`` ModuleList, ``lout``) as reference
wisp/models/decoders/basic_decoders.py:17-101, so state_dicts and the optimiser's name-based parameter groups
('decoder' in the parameter name, base_trainer.py:219-223) carry over.

Execution (row a14 of SURVEY.md section 8): when the module is the plain configuration the image/NeRF nefs build
(``nn.Linear`` layers with bias, ReLU activation, no skip connections) and its shape is one the HIP library compiles,
forward and backward are ONE fused kernel each (``shacira_mlp_{forward,backward}``): activations never leave registers,
the weight gradients are reduced on chip. Anything else runs as the usual chain of torch Linear layers.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import hip_ops
from ...core import WispModule

_RELUS = (torch.relu, F.relu)


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, dims):
        x = x.contiguous()
        params = params.contiguous()
        ctx.save_for_backward(x, params)
        ctx.dims = dims
        return hip_ops.mlp_forward(x, params, *dims)

    @staticmethod
    def backward(ctx, grad_y):
        x, params = ctx.saved_tensors
        gx, gp = hip_ops.mlp_backward(x, params, grad_y.contiguous(), *ctx.dims, need_grad_x=ctx.needs_input_grad[0])
        return gx, (gp if ctx.needs_input_grad[1] else None), None


class BasicDecoder(WispModule):
    def __init__(self, input_dim, output_dim, activation, bias, layer=nn.Linear, num_layers=1, hidden_dim=128,
                 skip=[]):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.activation, self.bias, self.layer = activation, bias, layer
        self.num_layers, self.hidden_dim = num_layers, hidden_dim
        self.skip = [] if skip is None else skip
        self.make()

    def make(self):
        widths = []
        for i in range(self.num_layers):
            fan_in = self.input_dim if i == 0 else self.hidden_dim + (self.input_dim if i in self.skip else 0)
            widths.append(fan_in)
        self.layers = nn.ModuleList([self.layer(w, self.hidden_dim, bias=self.bias) for w in widths])
        self.lout = self.layer(self.hidden_dim, self.output_dim, bias=self.bias)

    def _fused_dims(self, x):
        is_relu = self.activation in _RELUS or isinstance(self.activation, nn.ReLU)
        if not (x.is_cuda and x.dtype == torch.float32 and is_relu and self.bias and self.layer is nn.Linear
                and not self.skip and self.num_layers >= 1):
            return None
        dims = (self.input_dim, self.hidden_dim, self.num_layers, self.output_dim)
        return dims if hip_ops.mlp_supported(*dims) else None

    def packed_params(self):
        """W1, b1, ..., Wout, bout flattened into the C-ABI's parameter block (differentiable: one cat)."""
        pieces = []
        for lin in list(self.layers) + [self.lout]:
            pieces += [lin.weight.reshape(-1), lin.bias]
        return torch.cat(pieces)

    def forward(self, x, return_h=False):
        """x [batch, ..., input_dim] -> [batch, ..., output_dim] (and the last hidden layer if ``return_h``)."""
        dims = None if return_h else self._fused_dims(x)
        if dims is not None:
            lead = x.shape[:-1]
            y = _FusedMLP.apply(x.reshape(-1, self.input_dim), self.packed_params(), dims)
            return y.reshape(*lead, self.output_dim)
        h = None
        for i, lin in enumerate(self.layers):
            h = self.activation(lin(x if i == 0 else h))
            if i > 0 and i in self.skip:
                h = torch.cat([x, h], dim=-1)
        out = self.lout(h)
        return (out, h) if return_h else out

    def initialize(self, get_weight):
        """Re-initialise every layer's weight with ``get_weight(weight)`` (reference basic_decoders.py:103-116)."""
        ms = []
        for lin in list(self.layers) + [self.lout]:
            m = get_weight(lin.weight)
            lin.weight = nn.Parameter(m)
            ms.append(m)
        return ms

    def name(self) -> str:
        return "BasicDecoder"

    def public_properties(self):
        return {"Input Dim": self.input_dim, "Hidden Dim": self.hidden_dim, "Output Dim": self.output_dim,
                "Num. Layers": self.num_layers, "Layer Type": self.layer.__name__, "Bias": self.bias}
