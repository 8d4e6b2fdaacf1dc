"""Kaolin-free stand-in for the occupancy structure the grids carry (reference: wisp/accelstructs/octree_as.py,
used by hash_grid.py:60-66 / latent_grid.py:70-76 through ``OctreeAS.make_dense``).

One dense occupancy level with Morton-ordered cell coordinates (``points``) so that ``dense_points`` / ``num_cells``
/ ``occupancy`` have the reference's shapes, plus -- SURVEY.md section 8 "next" row f2 -- the queries the NeRF pipeline
makes of it: ``query``, ``raytrace``, ``raymarch`` ('ray' and 'voxel', octree_as.py:129-307) and
``from_quantized_points`` (pruning, nerf.py:150-185). The reference answers them with kaolin's sparse-octree CUDA
(un-vendored); here the occupied set is a dense [G, G, G] bit grid walked by the HIP kernels of render.hip.
``pidx`` values are Morton indices of the level's cells (the reference's index into its SPC point hierarchy).
"""
from collections import namedtuple

import torch

ASQueryResults = namedtuple("ASQueryResults", ["pidx"])
ASRaytraceResults = namedtuple("ASRaytraceResults", ["ridx", "pidx", "depth"])
ASRaymarchResults = namedtuple("ASRaymarchResults", ["ridx", "samples", "depth_samples", "deltas", "boundary",
                                                     "ray_offsets"], defaults=[None])


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


def _morton_index(q: torch.Tensor, level: int) -> torch.Tensor:
    """int cells [N, 3] -> Morton index (x most significant of each bit triple), the row of ``_morton_points``."""
    q = q.long()
    m = torch.zeros(q.shape[0], dtype=torch.int64, device=q.device)
    for b in range(level):
        m |= (((q[:, 0] >> b) & 1) << (3 * b + 2)) | (((q[:, 1] >> b) & 1) << (3 * b + 1)) | (((q[:, 2] >> b) & 1) << (3 * b))
    return m


class BaseAS:
    def raymarch(self, *args, **kwargs):
        raise NotImplementedError

    def raytrace(self, *args, **kwargs):
        raise NotImplementedError

    def query(self, *args, **kwargs):
        raise NotImplementedError


class OctreeAS(BaseAS):
    """Occupancy at one level. ``points`` holds the occupied cells of ``level`` in Morton order (the reference's SPC
    holds the whole pyramid; the grids only ever ask for the cells of ``blas_level``); ``occupancy_grid`` is the same
    set as a dense bool [G, G, G] indexed [x][y][z]."""

    def __init__(self, level: int, occupancy_grid: torch.Tensor = None):
        self.max_level = level
        G = 1 << level
        if occupancy_grid is None:
            self.points = _morton_points(level)
            self.occupancy_grid = torch.ones((G, G, G), dtype=torch.bool)
        else:
            self.occupancy_grid = occupancy_grid.bool()
            cells = torch.nonzero(self.occupancy_grid)
            order = torch.argsort(_morton_index(cells, level))
            self.points = cells[order].to(torch.int16)
        self.pyramid = torch.tensor([[self.points.shape[0]], [0]], dtype=torch.int32)
        self.extent = dict()

    @classmethod
    def make_dense(cls, level: int):
        return cls(level)

    @classmethod
    def from_quantized_points(cls, quantized_points: torch.Tensor, level: int):
        """Occupied set = the given integer cells [N, 3] (reference octree_as.py `from_quantized_points`)."""
        G = 1 << level
        grid = torch.zeros((G, G, G), dtype=torch.bool, device=quantized_points.device)
        q = quantized_points.long()
        grid[q[:, 0], q[:, 1], q[:, 2]] = True
        return cls(level, grid)

    def _grid_on(self, device):
        if self.occupancy_grid.device != device:
            self.occupancy_grid = self.occupancy_grid.to(device)
        return self.occupancy_grid

    def _level(self, level):
        if level is not None and level != self.max_level:
            raise NotImplementedError("only the BLAS level is materialised")
        return self.max_level

    def query(self, coords, level=None, with_parents=False) -> ASQueryResults:
        """pidx [N]: Morton index of the cell holding each point, -1 if that cell is unoccupied or the point lies outside
        the cube (kaolin's float query: cell = floor(G * (x + 1) / 2), out of bounds -> -1)."""
        if with_parents:
            raise NotImplementedError("with_parents needs the point hierarchy")
        level = self._level(level)
        G = 1 << level
        cell = torch.floor(G * (coords + 1.0) / 2.0)
        inside = ((cell >= 0) & (cell < G)).all(dim=-1)
        q = torch.nan_to_num(cell, nan=0.0).clamp(0, G - 1).long()
        occ = inside & self._grid_on(coords.device)[q[:, 0], q[:, 1], q[:, 2]]
        pidx = torch.where(occ, _morton_index(q, level), torch.full_like(q[:, 0], -1))
        return ASQueryResults(pidx=pidx)

    def raytrace(self, rays, level=None, with_exit=False) -> ASRaytraceResults:
        from ... import render
        level = self._level(level)
        ridx, pidx, depth = render.raytrace_dense(rays.origins, rays.dirs, self._grid_on(rays.origins.device), level)
        return ASRaytraceResults(ridx=ridx, pidx=pidx, depth=depth if with_exit else depth[:, 0:1])

    def _raymarch_voxel(self, rays, num_samples, level=None) -> ASRaymarchResults:
        """num_samples jittered samples inside every intersected cell (reference octree_as.py:171-233)."""
        from ... import render
        res = self.raytrace(rays, level, with_exit=True)
        ridx, depth = res.ridx.long(), res.depth
        K = ridx.shape[0]
        steps = torch.arange(num_samples, device=depth.device)[None].float().repeat([K, 1])
        steps += torch.rand_like(steps)
        steps *= (1.0 / num_samples)
        depth_samples = (depth[..., 0:1] + (depth[..., 1:2] - depth[..., 0:1]) * steps)[..., None]
        deltas = depth_samples[..., 0].diff(dim=-1, prepend=depth[..., 0:1]).reshape(K * num_samples, 1)
        samples = torch.addcmul(rays.origins.index_select(0, ridx)[:, None], rays.dirs.index_select(0, ridx)[:, None],
                                depth_samples)
        boundary = torch.zeros(K * num_samples, dtype=torch.bool, device=depth.device)
        boundary[torch.nonzero(render.mark_pack_boundaries(ridx)).flatten() * num_samples] = True
        ridx = ridx[:, None].expand(K, num_samples).reshape(K * num_samples)
        return ASRaymarchResults(ridx=ridx, samples=samples.reshape(K * num_samples, 3),
                                 depth_samples=depth_samples.reshape(K * num_samples, 1), deltas=deltas,
                                 boundary=boundary)

    def _raymarch_ray(self, rays, num_samples, level=None) -> ASRaymarchResults:
        """num_samples stratified samples per ray between dist_min and dist_max, kept where occupied (reference
        octree_as.py:235-290): generation, occupancy filter and compaction in two HIP launches."""
        from ... import render
        level = self._level(level)
        capacity = getattr(self, "sample_capacity", None)
        if capacity is not None:
            # fixed-size outputs for a step captured into a HIP graph (harness.GraphedNerfFitter): no count read-back;
            # `last_sample_count` keeps the true count on the device so that the owner can watch for dropped samples
            ridx, samples, depth, deltas, boundary, offsets, self.last_sample_count = render.raymarch_ray(
                rays.origins, rays.dirs, rays.dist_min, rays.dist_max, self._grid_on(rays.origins.device), level,
                num_samples, capacity=capacity)
        else:
            ridx, samples, depth, deltas, boundary, offsets = render.raymarch_ray(
                rays.origins, rays.dirs, rays.dist_min, rays.dist_max, self._grid_on(rays.origins.device), level,
                num_samples)
        return ASRaymarchResults(ridx=ridx, samples=samples, depth_samples=depth, deltas=deltas, boundary=boundary,
                                 ray_offsets=offsets)

    def raymarch(self, rays, raymarch_type, num_samples, level=None) -> ASRaymarchResults:
        if raymarch_type == "voxel":
            return self._raymarch_voxel(rays, num_samples, level)
        if raymarch_type == "ray":
            return self._raymarch_ray(rays, num_samples, level)
        raise TypeError(f"Raymarch sampler type: {raymarch_type} is not supported by OctreeAS.")

    def level_points(self, level: int) -> torch.Tensor:
        if level != self.max_level:
            raise NotImplementedError("only the dense BLAS level is materialised")
        return self.points
