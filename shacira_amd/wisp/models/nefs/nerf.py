"""``NeuralRadianceField``: grid features -> density decoder -> (+ embedded view direction) colour decoder
(reference wisp/models/nefs/nerf.py:31-250): same constructor options, parameter names (``grid.*``,
``decoder_density.*``, ``decoder_color.*`` -- optimiser groups are selected by these substrings), channels and pruning
rule. The grid lookup runs through the HIP operators; the decoders use the fused MLP kernel when their shape is one of
its instantiations and torch Linear layers otherwise.
"""
import numpy as np
import torch
import torch.nn as nn

from ..decoders.basic_decoders import BasicDecoder
from ..embedders import get_positional_embedder
from ..grids import HashGrid, LatentGrid


def sample_unif_sphere(n):
    """n unit vectors, uniform on the sphere (normalised Gaussians, reference wisp/ops/geometric `sample_unif_sphere`)."""
    u = np.random.randn(n, 3)
    return u / np.linalg.norm(u, axis=1, keepdims=True)


class NeuralRadianceField(nn.Module):
    def __init__(self, grid, pos_embedder="none", view_embedder="none", pos_multires=10, view_multires=4,
                 position_input=False, activation_type="relu", layer_type="none", hidden_dim=128, num_layers=1,
                 prune_density_decay=None, prune_min_density=None):
        super().__init__()
        if activation_type != "relu" or layer_type not in ("none", "linear"):
            raise NotImplementedError("relu activations and plain linear layers (the reference's NeRF configs)")
        self.grid = grid
        self.pos_embedder, self.pos_embed_dim = self.init_embedder(pos_embedder, pos_multires, position_input)
        self.view_embedder, self.view_embed_dim = self.init_embedder(view_embedder, view_multires, True)
        self.activation_type, self.layer_type = activation_type, layer_type
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        self.decoder_density = BasicDecoder(self.density_net_input_dim(), 16, torch.relu, True, nn.Linear, num_layers,
                                            hidden_dim, [])
        self.decoder_density.lout.bias.data[0] = 1.0
        self.decoder_color = BasicDecoder(self.color_net_input_dim(), 3, torch.relu, True, nn.Linear, num_layers + 1,
                                          hidden_dim, [])
        self.prune_density_decay, self.prune_min_density = prune_density_decay, prune_min_density

    def init_embedder(self, embedder_type, frequencies=None, include_input=False):
        if embedder_type == "none" and not include_input:
            return None, 0
        if embedder_type == "identity" or (embedder_type == "none" and include_input):
            return nn.Identity(), 3
        if embedder_type == "positional":
            return get_positional_embedder(frequencies=frequencies, include_input=include_input)
        raise NotImplementedError(f"Unsupported embedder type for NeuralRadianceField: {embedder_type}")

    def effective_feature_dim(self):
        if self.grid.multiscale_type == "cat":
            return self.grid.feature_dim * self.grid.num_lods
        return self.grid.feature_dim

    def density_net_input_dim(self):
        return self.effective_feature_dim() + self.pos_embed_dim

    def color_net_input_dim(self):
        return 16 + self.view_embed_dim

    def get_supported_channels(self):
        return {"density", "rgb"}

    def forward(self, channels=None, **kwargs):
        """Dict of the requested channels (a single channel name returns its tensor, like the reference's BaseNeuralField)."""
        out = self.rgba(**kwargs)
        if isinstance(channels, str):
            return out[channels]
        return out if channels is None else {c: out[c] for c in channels}

    def rgba(self, coords, ray_d, lod_idx=None):
        if lod_idx is None:
            lod_idx = len(self.grid.active_lods) - 1
        batch = coords.shape[0]
        feats = self.grid.interpolate(coords, lod_idx).reshape(batch, self.effective_feature_dim())
        if self.pos_embedder is not None:
            feats = torch.cat([feats, self.pos_embedder(coords).view(batch, self.pos_embed_dim)], dim=-1)
        density_feats = self.decoder_density(feats)
        if self.view_embedder is not None:
            fdir = torch.cat([density_feats, self.view_embedder(-ray_d).view(batch, self.view_embed_dim)], dim=-1)
        else:
            fdir = density_feats
        colors = torch.sigmoid(self.decoder_color(fdir))
        density = torch.relu(density_feats[..., 0:1])            # particles / unit length; times delta in the tracer
        return dict(rgb=colors, density=density)

    def prune(self):
        """Decay the running per-cell density, refresh it with one jittered sample per cell, keep the cells above
        ``prune_min_density`` as the new occupancy (Mueller et al. 2022; reference nerf.py:150-185)."""
        if not isinstance(self.grid, (HashGrid, LatentGrid)):
            raise NotImplementedError(f"Pruning not implemented for grid type {self.grid}")
        device = self.grid.codebook.device
        self.grid.occupancy = self.grid.occupancy.to(device) * self.prune_density_decay
        points = self.grid.dense_points.to(device)
        res = 2.0 ** self.grid.blas_level
        samples = (points.float() + torch.rand(points.shape[0], 3, device=device)) / res * 2.0 - 1.0
        views = torch.as_tensor(sample_unif_sphere(samples.shape[0]), dtype=torch.float32, device=device)
        with torch.no_grad():
            density = self.forward(coords=samples, ray_d=views, channels="density")
        self.grid.occupancy = torch.stack([density[:, 0], self.grid.occupancy], -1).max(dim=-1)[0]
        kept = points[self.grid.occupancy > self.prune_min_density]
        if kept.shape[0] == 0:
            return
        self.grid.blas = self.grid.blas.__class__.from_quantized_points(kept, self.grid.blas_level)
