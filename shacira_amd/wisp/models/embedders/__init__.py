"""NeRF positional encoding (reference wisp/models/embedders/positional_embedder.py:15-100)."""
import torch
import torch.nn as nn


class PositionalEmbedder(nn.Module):
    """coords [N, d] -> [coords,] sin(coords * 2^k), cos(coords * 2^k) for k log-spaced in [0, max_freq_log2]: all
    sines first, then all cosines, frequency-major inside each (the reference's layout)."""

    def __init__(self, num_freq, max_freq_log2, log_sampling=True, include_input=True, input_dim=3):
        super().__init__()
        self.num_freq, self.max_freq_log2 = num_freq, max_freq_log2
        self.log_sampling, self.include_input = log_sampling, include_input
        bands = 2.0 ** torch.linspace(0.0, max_freq_log2, steps=num_freq) if log_sampling \
            else torch.linspace(1, 2.0 ** max_freq_log2, steps=num_freq)
        self.out_dim = (input_dim if include_input else 0) + bands.shape[0] * input_dim * 2
        self.bands = nn.Parameter(bands).requires_grad_(False)

    def forward(self, coords):
        n = coords.shape[0]
        winded = (coords[:, None] * self.bands[None, :, None]).reshape(n, coords.shape[1] * self.num_freq)
        encoded = torch.cat([torch.sin(winded), torch.cos(winded)], dim=-1)
        return torch.cat([coords, encoded], dim=-1) if self.include_input else encoded


def get_positional_embedder(frequencies, input_dim=3, include_input=True):
    encoder = PositionalEmbedder(frequencies, frequencies - 1, input_dim=input_dim, include_input=include_input)
    return encoder, encoder.out_dim
