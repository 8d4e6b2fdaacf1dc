"""Kaolin-free stand-in for the occupancy structure the grids carry (reference: wisp/accelstructs/octree_as.py,
used by hash_grid.py:60-66 / latent_grid.py:70-76 through ``OctreeAS.make_dense``).

Only what the hash-grid path touches is provided: a dense occupancy level with Morton-ordered cell coordinates
(``points``) so that ``dense_points`` / ``num_cells`` / ``occupancy`` have the reference's shapes. Ray marching,
ray tracing and point queries live in un-vendored kaolin CUDA in the reference and are out of scope here
(SURVEY.md section 8 row f2): they raise NotImplementedError.
"""
from collections import namedtuple

import torch

ASQueryResults = namedtuple("ASQueryResults", ["pidx"])
ASRaytraceResults = namedtuple("ASRaytraceResults", ["ridx", "pidx", "depth"])
ASRaymarchResults = namedtuple("ASRaymarchResults", ["ridx", "samples", "depth_samples", "deltas", "boundary"])


def _morton_points(level: int) -> torch.Tensor:
    """All (x, y, z) cells of a 2^level cube in Morton order (x is the most significant of each bit triple)."""
    n = 1 << (3 * level)
    code = torch.arange(n, dtype=torch.int64)
    xyz = torch.zeros(n, 3, dtype=torch.int64)
    for b in range(level):
        xyz[:, 0] |= ((code >> (3 * b + 2)) & 1) << b
        xyz[:, 1] |= ((code >> (3 * b + 1)) & 1) << b
        xyz[:, 2] |= ((code >> (3 * b + 0)) & 1) << b
    return xyz.to(torch.int16)


class BaseAS:
    def raymarch(self, *args, **kwargs):
        raise NotImplementedError("ray marching is outside the hash-grid path (SURVEY.md section 8 f2)")

    def raytrace(self, *args, **kwargs):
        raise NotImplementedError("ray tracing is outside the hash-grid path (SURVEY.md section 8 f2)")

    def query(self, *args, **kwargs):
        raise NotImplementedError("point queries are outside the hash-grid path (SURVEY.md section 8 f2)")


class OctreeAS(BaseAS):
    """Dense occupancy at one level. ``points`` holds the cells of ``level`` only (the reference's SPC holds the
    whole pyramid; the grids only ever ask for the cells of ``blas_level``)."""

    def __init__(self, level: int):
        self.max_level = level
        self.points = _morton_points(level)
        self.pyramid = torch.tensor([[self.points.shape[0]], [0]], dtype=torch.int32)

    @classmethod
    def make_dense(cls, level: int):
        return cls(level)

    def level_points(self, level: int) -> torch.Tensor:
        if level != self.max_level:
            raise NotImplementedError("only the dense BLAS level is materialised")
        return self.points
