"""``HashGrid``: multi-resolution hash table of features (Instant-NGP style) behind the ``BLASGrid`` interface.

Mirror of reference wisp/models/grids/hash_grid.py:21-287 -- same constructor/classmethod signatures, the same
``codebook`` Parameter ([T, F], levels concatenated) and int32 buffers ``codebook_lod_sizes`` /
``codebook_lod_first_idx`` (state_dict keys), the same RNG draw order at init (one ``randn`` per level), and the
same ``interpolate`` post-processing ('cat' / 'sum', [B,S,d] flattening, RENDERING_FINAL mask).
The lookup itself runs in the HIP kernels through ``wisp.ops.grid.hashgrid[2d]``.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Set, Type

import numpy as np
import torch
import torch.nn as nn

from ...accelstructs import ASRaymarchResults, BaseAS, OctreeAS
from ...ops import grid as grid_ops
from .blas_grid import BLASGrid


def geometric_resolutions(min_grid_res: int, max_grid_res: int, num_lods: int) -> List[int]:
    """Instant-NGP eq. 2-3 as the reference evaluates it (numpy float64; hash_grid.py:176-177)."""
    b = np.exp((np.log(max_grid_res) - np.log(min_grid_res)) / (num_lods - 1))
    return [int(1 + np.floor(min_grid_res * (b ** l))) for l in range(num_lods)]


class _MultiLevelTable(BLASGrid):
    """Everything HashGrid and LatentGrid share: occupancy stub, level bookkeeping, table allocation, lookup."""

    def _setup_occupancy(self, blas_level: int):
        self.blas_level = blas_level
        blas = OctreeAS.make_dense(level=blas_level)
        BLASGrid.__init__(self, blas)
        self.dense_points = self.blas.level_points(blas_level).clone()
        self.num_cells = self.dense_points.shape[0]
        self.occupancy = torch.zeros(self.num_cells)

    def _setup_levels(self, resolutions: List[int], codebook_bitwidth: int):
        self.codebook_bitwidth = codebook_bitwidth
        self.resolutions = resolutions
        self.num_lods = len(resolutions)
        self.active_lods = [x for x in range(self.num_lods)]
        self.max_lod = self.num_lods - 1
        self.codebook_size = 2 ** self.codebook_bitwidth
        self.register_buffer("codebook_lod_sizes", torch.zeros(self.num_lods, dtype=torch.int32))
        self.register_buffer("codebook_lod_first_idx", torch.zeros(self.num_lods, dtype=torch.int32))

    def _allocate_table(self, width: int, resolution_dim: int, fill):
        """rows_l = min(2^bw, res_l^resolution_dim); ``fill(rows, width)`` draws one level (keeps RNG order)."""
        levels, offset = [], 0
        for lod, res in enumerate(self.resolutions):
            rows = min(self.codebook_size, res ** resolution_dim)
            levels.append(fill(rows, width))
            self.codebook_lod_sizes[lod] = rows
            self.codebook_lod_first_idx[lod] = offset
            offset += rows
        self.codebook = nn.Parameter(torch.cat(levels, dim=0))

    def _lookup(self, coords, lod_idx, table):
        """coords [..., d] -> per-level features [N, L*F] from ``table`` plus the shape to restore."""
        output_shape = coords.shape[:-1]
        if coords.ndim == 3:
            coords = coords.reshape(-1, coords.shape[-1])
        fn = grid_ops.hashgrid2d if coords.shape[-1] == 2 else grid_ops.hashgrid
        feats = fn(coords, self.resolutions, self.codebook_bitwidth, lod_idx, table, self.codebook_lod_sizes,
                   self.codebook_lod_first_idx)
        return feats, output_shape

    def _aggregate(self, feats, lod_idx, output_shape):
        if "RENDERING_FINAL" in os.environ:
            mask = torch.zeros_like(feats)
            mask[:, :lod_idx * self.feature_dim] = 1
            feats = feats * mask
        if self.multiscale_type == "cat":
            return feats.reshape(*output_shape, feats.shape[-1])
        if self.multiscale_type == "sum":
            L = len(self.resolutions)
            return feats.reshape(*output_shape, L, feats.shape[-1] // L).sum(-2)
        raise NotImplementedError

    def raymarch(self, rays, raymarch_type, num_samples, level=None) -> ASRaymarchResults:
        return self.blas.raymarch(rays, raymarch_type=raymarch_type, num_samples=num_samples, level=self.blas_level)

    def supported_blas(self) -> Set[Type[BaseAS]]:
        return {OctreeAS}

    def _lod_range(self):
        return None if not self.active_lods else f"{min(self.active_lods)} - {max(self.active_lods)}"


class HashGrid(_MultiLevelTable):
    def __init__(self, feature_dim: int, resolutions: List[int], multiscale_type: str = "sum",
                 resolution_dim: int = 3, feature_std: float = 0.0, feature_bias: float = 0.0,
                 codebook_bitwidth: int = 8, blas_level: int = 7):
        self._setup_occupancy(blas_level)
        self.feature_dim = feature_dim
        self.multiscale_type = multiscale_type
        self.feature_std = feature_std
        self.feature_bias = feature_bias  # accepted and unused, as in the reference
        self._setup_levels(resolutions, codebook_bitwidth)
        self._allocate_table(feature_dim, resolution_dim,
                             lambda rows, width: torch.zeros(rows, width) + torch.randn(rows, width) * feature_std)

    def size(self, use_torchac=False, use_prob_model=False):
        return 0.0, self.codebook.numel() * torch.finfo(self.codebook.dtype).bits

    @classmethod
    def from_octree(cls, feature_dim: int, base_lod: int = 2, num_lods: int = 1, multiscale_type: str = "sum",
                    resolution_dim: int = 3, feature_std: float = 0.0, feature_bias: float = 0.0,
                    codebook_bitwidth: int = 8, blas_level: int = 7) -> HashGrid:
        """Octree sampling pattern: res_l = 2^(base_lod + l)."""
        resolutions = [2 ** (base_lod + x) for x in range(num_lods)]
        return cls(feature_dim=feature_dim, resolutions=resolutions, multiscale_type=multiscale_type,
                   feature_std=feature_std, feature_bias=feature_bias, codebook_bitwidth=codebook_bitwidth,
                   blas_level=blas_level, resolution_dim=resolution_dim)

    @classmethod
    def from_geometric(cls, feature_dim: int, num_lods: int, multiscale_type: str = "sum", resolution_dim: int = 3,
                       feature_std: float = 0.0, feature_bias: float = 0.0, codebook_bitwidth: int = 8,
                       min_grid_res: int = 16, max_grid_res: int = None, blas_level: int = 7) -> HashGrid:
        """Geometric progression of resolutions between min_grid_res and max_grid_res (Instant-NGP)."""
        return cls(feature_dim=feature_dim, resolutions=geometric_resolutions(min_grid_res, max_grid_res, num_lods),
                   multiscale_type=multiscale_type, feature_std=feature_std, feature_bias=feature_bias,
                   codebook_bitwidth=codebook_bitwidth, blas_level=blas_level, resolution_dim=resolution_dim)

    @classmethod
    def from_resolutions(cls, feature_dim: int, resolutions: List[int], multiscale_type: str = "sum",
                         resolution_dim: int = 3, feature_std: float = 0.0, feature_bias: float = 0.0,
                         codebook_bitwidth: int = 8, blas_level: int = 7) -> HashGrid:
        return cls(feature_dim=feature_dim, resolutions=resolutions, multiscale_type=multiscale_type,
                   feature_std=feature_std, feature_bias=feature_bias, codebook_bitwidth=codebook_bitwidth,
                   blas_level=blas_level, resolution_dim=resolution_dim)

    def freeze(self):
        self.codebook.requires_grad_(False)

    def interpolate(self, coords, lod_idx):
        """coords [batch, (num_samples,) 2|3] -> features [batch, (num_samples,) F*L ('cat') or F ('sum')]."""
        feats, output_shape = self._lookup(coords, lod_idx, self.codebook)
        return self._aggregate(feats, lod_idx, output_shape)

    def name(self) -> str:
        return "Hash Grid"

    def public_properties(self) -> Dict[str, Any]:
        properties = {
            "Feature Dims": self.feature_dim,
            "Total LODs": self.max_lod,
            "Active feature LODs": self._lod_range(),
            "Interpolation": "linear",
            "Multiscale aggregation": self.multiscale_type,
            "HashTable Size": f"2^{self.codebook_bitwidth}",
        }
        return {**super().public_properties(), **properties}
