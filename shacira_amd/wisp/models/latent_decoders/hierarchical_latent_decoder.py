"""One ``LatentDecoder`` per level of the grid (reference wisp/models/latent_decoders/hierarchical_latent_decoder.py:3-36).

``offsets`` are the row boundaries of the levels inside the concatenated table. The reference builds them as
``cat(codebook_lod_first_idx, codebook_lod_sizes[-1:])`` (latent_grid.py:182), i.e. the final boundary is the LAST
LEVEL'S SIZE rather than the table's end -- a reference quirk kept by ``LatentGrid.setup_decoders`` here so that
results match; rows past that boundary are left as ``torch.empty`` leaves them in the reference and are zero here.
"""
import torch
import torch.nn as nn

from .basic_latent_decoder import LatentDecoder


class HierarchicalLatentDecoder(nn.Module):
    def __init__(self, num_decoders, offsets, conf_decoder):
        super().__init__()
        self.num_decoders = num_decoders
        self.decoders = nn.ModuleList([LatentDecoder(**conf_decoder) for _ in range(num_decoders)])
        self.offsets = offsets

    def forward(self, input):
        bounds = [int(o) for o in self.offsets]
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
