"""CPU restatement of the sample generation and volume integration around the hash-grid path (SURVEY.md section 8,
"next" row f2). TEST INFRASTRUCTURE ONLY -- imported by tests/ (and nothing in the product path).

Where the algorithm comes from:
  * `raymarch_ray`        reference wisp/accelstructs/octree_as.py:235-290 (`OctreeAS._raymarch_ray`, the reference's own
                          Python) with `fast_filter_method` :20-32; the occupancy query it calls is kaolin's
                          `spc_ops.unbatched_query` on FLOAT points: cell = floor(res*(x*0.5+0.5)) per axis, and kaolin's
                          `identify` (spc_math.h, "Check if in bounds") answers -1 for a cell outside [0, res) -- a point
                          outside the cube (or exactly on its +1 face) belongs to no cell.
  * `raymarch_voxel`      reference octree_as.py:171-233 (`_raymarch_voxel`) + wisp/ops/spc/sampling.py:35-71
                          (`sample_from_depth_intervals`, `expand_pack_boundary`); the ray / cell intersections it starts
                          from come from kaolin's `spc_render.unbatched_raytrace(..., with_exit=True)`.
  * `exponential_integration`, `sum_reduce`, `cumsum`, `mark_pack_boundaries`
                          kaolin.render.spc (kaolin 0.13.0, pinned in the reference's README.md:35; NOT vendored in
                          /root/reference). Restated from its published Python (`kaolin/render/spc/raytrace.py`):
                              alpha = 1 - exp(-tau); T = exp(-cumsum(tau, boundary, exclusive));
                              w = T * alpha; feats_out = sum_reduce(w * feats, boundary); return feats_out, w
                          Call sites that fix the semantics used here: wisp/tracers/packed_rf_tracer.py:131-151.

PARITY UNPINNED for the kaolin parts: the dependency is absent, the reference has no tests or golden vectors for
them, so this restatement is anchored on the published formula and the reference's call sites only. The
reference-owned Python parts (`_raymarch_ray`, `sample_from_depth_intervals`, `expand_pack_boundary`) are restated
line by line.
"""
import numpy as np
import torch


def mark_pack_boundaries(ridx):
    """True at the first element of every run of equal ids."""
    ridx = torch.as_tensor(ridx)
    if ridx.numel() == 0:
        return torch.zeros(0, dtype=torch.bool)
    return torch.cat([torch.ones(1, dtype=torch.bool), ridx[1:] != ridx[:-1]])


def _pack_ids(boundary):
    return torch.cumsum(boundary.long(), 0) - 1


def cumsum(x, boundary, exclusive=False):
    """Per-pack cumulative sum of x [S, C] (fp64 inside)."""
    x64 = x.double()
    total = torch.cumsum(x64, 0)
    pid = _pack_ids(boundary)
    starts = torch.nonzero(boundary).flatten()
    before = torch.cat([torch.zeros(1, x.shape[1], dtype=torch.float64), total])[starts]   # sum before each pack
    out = total - before[pid]
    if exclusive:
        out = out - x64
    return out.to(x.dtype)


def sum_reduce(x, boundary):
    pid = _pack_ids(boundary)
    n = int(boundary.sum())
    out = torch.zeros(n, x.shape[1], dtype=torch.float64)
    out.index_add_(0, pid, x.double())
    return out.to(x.dtype)


def exponential_integration(feats, tau, boundary, exclusive=True):
    alpha = 1.0 - torch.exp(-tau)
    transmittance = torch.exp(-1.0 * cumsum(tau, boundary, exclusive=exclusive))
    transmittance = transmittance * alpha
    return sum_reduce(transmittance * feats, boundary), transmittance


def quantize_points(x, level):
    res = 2 ** level
    return torch.floor(torch.clamp(res * (x + 1.0) / 2.0, 0, res - 1.0)).long()


def query_dense(occupancy, coords, level):
    """occupancy: bool [G, G, G] indexed [x, y, z]. -> bool [N] (the reference's `pidx > -1`): the point's cell is
    occupied; points outside the cube have no cell (kaolin's identify returns -1 out of bounds)."""
    res = 2 ** level
    cell = torch.floor(res * (coords + 1.0) / 2.0)
    inside = ((cell >= 0) & (cell < res)).all(dim=-1)
    q = torch.nan_to_num(cell, nan=0.0).clamp(0, res - 1).long()
    return inside & occupancy[q[:, 0], q[:, 1], q[:, 2]]


def raymarch_ray(origins, dirs, dist_min, dist_max, occupancy, level, num_samples, jitter):
    """octree_as.py:257-290 with the random numbers injected (`jitter` = torch.rand(num_rays, num_samples))."""
    num_rays = origins.shape[0]
    depth = torch.linspace(0, 1.0, num_samples)[None] + (jitter / num_samples)
    depth = depth * (dist_max - dist_min)
    depth = depth + dist_min
    samples = torch.addcmul(origins[:, None], dirs[:, None], depth[..., None])
    mask = query_dense(occupancy, samples.reshape(num_rays * num_samples, 3), level).reshape(num_rays, num_samples)
    idx = torch.nonzero(mask)
    deltas = depth.diff(dim=-1, prepend=(torch.zeros(num_rays, 1) + dist_min))
    depth_samples = depth[idx[:, 0], idx[:, 1]][:, None]
    deltas = deltas[idx[:, 0], idx[:, 1]].reshape(-1, 1)
    samples = samples[idx[:, 0], idx[:, 1], :]
    ridx = idx[:, 0]
    return ridx, samples, depth_samples, deltas, mark_pack_boundaries(ridx)


def filter_samples(all_samples, all_depth, all_deltas, occupancy, level, num_rays, num_samples):
    """The filtering half of `raymarch_ray` applied to GIVEN per-(ray, sample) positions [num_rays*num_samples, 3]."""
    mask = query_dense(occupancy, all_samples, level).reshape(num_rays, num_samples)
    idx = torch.nonzero(mask)
    flat = idx[:, 0] * num_samples + idx[:, 1]
    ridx = idx[:, 0]
    return ridx, all_samples[flat], all_depth[flat], all_deltas[flat], mark_pack_boundaries(ridx)


def raytrace_dense(origins, dirs, occupancy, level):
    """All (ray, occupied cell) intersections, ordered by ray then entry depth: ridx [K], cell [K, 3], depth [K, 2]
    (entry clipped to 0 for rays that start inside a cell). Brute force in fp64: slab test against every occupied
    cell -- the published behaviour of kaolin's unbatched_raytrace(with_exit=True) on a one-level dense octree."""
    G = 2 ** level
    cells = torch.nonzero(occupancy).double()           # [M, 3] integer coords
    lo = cells / G * 2.0 - 1.0
    hi = (cells + 1.0) / G * 2.0 - 1.0
    out_r, out_c, out_d = [], [], []
    for r in range(origins.shape[0]):
        o, d = origins[r].double(), dirs[r].double()
        d = torch.where(d.abs() < 1e-30, torch.full_like(d, 1e-30), d)
        t0, t1 = (lo - o) / d, (hi - o) / d
        tn = torch.minimum(t0, t1).max(dim=1)[0]
        tf = torch.maximum(t0, t1).min(dim=1)[0]
        hit = (tf > tn) & (tf > 0)
        k = torch.nonzero(hit).flatten()
        order = torch.argsort(tn[k])
        k = k[order]
        out_r.append(torch.full((k.numel(),), r, dtype=torch.long))
        out_c.append(cells[k].long())
        out_d.append(torch.stack([tn[k].clamp(min=0.0), tf[k]], -1))
    return torch.cat(out_r), torch.cat(out_c), torch.cat(out_d).float()


def sample_from_depth_intervals(depth_intervals, num_samples, jitter):
    """sampling.py:49-55 with `torch.rand_like(steps)` injected."""
    steps = torch.arange(num_samples)[None].float().repeat([depth_intervals.shape[0], 1])
    steps = steps + jitter
    steps = steps * (1.0 / num_samples)
    return depth_intervals[..., 0:1] + (depth_intervals[..., 1:2] - depth_intervals[..., 0:1]) * steps


def raymarch_voxel(origins, dirs, ridx, depth, num_samples, jitter):
    """octree_as.py:197-233 given the ray / cell intersections (ridx [K], depth [K, 2])."""
    K = ridx.shape[0]
    depth_samples = sample_from_depth_intervals(depth, num_samples, jitter)[..., None]
    deltas = depth_samples[..., 0].diff(dim=-1, prepend=depth[..., 0:1]).reshape(K * num_samples, 1)
    samples = torch.addcmul(origins.index_select(0, ridx)[:, None], dirs.index_select(0, ridx)[:, None], depth_samples)
    first = mark_pack_boundaries(ridx)
    boundary = torch.zeros(K * num_samples, dtype=torch.bool)
    boundary[torch.nonzero(first).flatten() * num_samples] = True
    ridx_s = ridx[:, None].expand(K, num_samples).reshape(K * num_samples)
    return ridx_s, samples.reshape(K * num_samples, 3), depth_samples.reshape(K * num_samples, 1), deltas, boundary
