"""Rounding operators of the latent path: straight-through round / floor (identity gradient) and the stochastic
Gumbel annealing (SGA) sampler between floor and ceil. Behaviour follows reference
wisp/models/latent_decoders/basic_latent_decoder.py:28-46 (STE) and :183-191 (SGA)."""
import torch

epsilon = 1e-6


def _straight_through(op, doc):
    class _STE(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return op(x)

        @staticmethod
        def backward(ctx, grad_output):
            return grad_output

    _STE.__doc__ = doc
    return _STE


StraightThrough = _straight_through(torch.round, "round() forward (half to even, ``torch.round``), identity backward.")
StraightThrough.__name__ = StraightThrough.__qualname__ = "StraightThrough"
StraightThroughFloor = _straight_through(torch.floor, "floor() forward, identity backward.")
StraightThroughFloor.__name__ = StraightThroughFloor.__qualname__ = "StraightThroughFloor"


def sga_sample(weight, temperature, diff_sampling):
    """Relaxed one-hot choice between floor(w) and floor(w)+1 with logits -tanh(distance)/T; torch ops, torch's RNG."""
    lo = torch.floor(weight) if diff_sampling else StraightThroughFloor.apply(weight)
    hi = lo + 1
    lim = 1 - epsilon
    logits = torch.cat((-torch.tanh(torch.clamp(weight - lo, min=-lim, max=lim)).unsqueeze(-1) / temperature,
                        -torch.tanh(torch.clamp(hi - weight, min=-lim, max=lim)).unsqueeze(-1) / temperature), dim=-1)
    dist = torch.distributions.relaxed_categorical.RelaxedOneHotCategorical(temperature, logits=logits)
    sample = dist.rsample() if diff_sampling else dist.sample()
    return lo * sample[..., 0] + hi * sample[..., 1]
