"""Sample generation on a dense occupancy grid and volume integration over ray packs (SURVEY.md section 8 "next" f2):
tensor wrappers + autograd Functions over libshacira_hip.so's `shacira_pack_*`, `shacira_raymarch_ray_*`,
`shacira_raytrace_dense_*`.

Function names and argument order follow what the reference calls in kaolin 0.13 (`kaolin.render.spc` imported as
``spc_render`` in wisp/tracers/packed_rf_tracer.py:131-151 and wisp/accelstructs/octree_as.py:163-289), so the tracer
and the acceleration structure mirrors read like the reference's. HIP only: CPU tensors raise.
"""
import ctypes

import torch

from . import _lib
from .hip_ops import _need_gpu, _on_device, _ptr, _stream


def mark_pack_boundaries(ridx):
    """True at the first element of every run of equal ray ids (kaolin spc_render.mark_pack_boundaries)."""
    if ridx.numel() == 0:
        return torch.zeros(0, dtype=torch.bool, device=ridx.device)
    return torch.cat([torch.ones(1, dtype=torch.bool, device=ridx.device), ridx[1:] != ridx[:-1]])


def pack_offsets(boundary):
    """bool [S] -> int64 [R + 1] start row of every pack plus the end sentinel (one device->host size read-back)."""
    starts = torch.nonzero(boundary).flatten()
    end = torch.full((1,), boundary.shape[0], dtype=torch.int64, device=boundary.device)
    return torch.cat([starts, end])


class _ExponentialIntegration(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, tau, pack_start):
        _need_gpu(feats, tau, pack_start)
        feats = feats.float().contiguous()
        tau = tau.float().contiguous().reshape(-1)
        S, C = feats.shape
        R = pack_start.shape[0] - 1
        ray_feats = torch.empty((R, C), dtype=torch.float32, device=feats.device)
        weights = torch.empty((S,), dtype=torch.float32, device=feats.device)
        with _on_device(feats.device):
            _lib.check(_lib.lib().shacira_pack_integrate_forward(S, R, C, _ptr(feats), _ptr(tau), _ptr(pack_start),
                                                                 _ptr(ray_feats), _ptr(weights), _stream(feats)),
                       "shacira_pack_integrate_forward")
        ctx.save_for_backward(feats, tau, pack_start)
        return ray_feats, weights.reshape(S, 1)

    @staticmethod
    def backward(ctx, g_ray, g_w):
        feats, tau, pack_start = ctx.saved_tensors
        S, C = feats.shape
        R = pack_start.shape[0] - 1
        g_ray = g_ray.float().contiguous()
        g_w = g_w.float().contiguous().reshape(-1) if g_w is not None else None
        g_feats = torch.zeros_like(feats)
        g_tau = torch.zeros_like(tau)
        with _on_device(feats.device):
            _lib.check(_lib.lib().shacira_pack_integrate_backward(S, R, C, _ptr(feats), _ptr(tau), _ptr(pack_start),
                                                                  _ptr(g_ray), _ptr(g_w), _ptr(g_feats), _ptr(g_tau),
                                                                  _stream(feats)), "shacira_pack_integrate_backward")
        return g_feats, g_tau.reshape(S, 1), None


class _SumReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pack_start):
        _need_gpu(x, pack_start)
        x = x.float().contiguous()
        S, C = x.shape
        R = pack_start.shape[0] - 1
        out = torch.empty((R, C), dtype=torch.float32, device=x.device)
        with _on_device(x.device):
            _lib.check(_lib.lib().shacira_pack_sum(S, R, C, _ptr(x), _ptr(pack_start), _ptr(out), _stream(x)),
                       "shacira_pack_sum")
        ctx.save_for_backward(pack_start)
        ctx.shape = (S, C)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (pack_start,) = ctx.saved_tensors
        S, C = ctx.shape
        R = pack_start.shape[0] - 1
        g_out = g_out.float().contiguous()
        g_x = torch.zeros((S, C), dtype=torch.float32, device=g_out.device)
        with _on_device(g_out.device):
            _lib.check(_lib.lib().shacira_pack_broadcast(S, R, C, _ptr(g_out), _ptr(pack_start), _ptr(g_x),
                                                         _stream(g_out)), "shacira_pack_broadcast")
        return g_x, None


def exponential_integration(feats, tau, boundary, exclusive=True, pack_start=None):
    """kaolin spc_render.exponential_integration: (ray_feats [R, C], transmittance*alpha [S, 1])."""
    if not exclusive:
        raise NotImplementedError("only exclusive=True (the reference's call sites) is implemented")
    if pack_start is None:
        pack_start = pack_offsets(boundary)
    return _ExponentialIntegration.apply(feats, tau.reshape(-1, 1), pack_start)


def sum_reduce(x, boundary, pack_start=None):
    """kaolin spc_render.sum_reduce: per-pack sums [R, C]."""
    if pack_start is None:
        pack_start = pack_offsets(boundary)
    return _SumReduce.apply(x, pack_start)


def _occupancy_u8(occupancy, level):
    G = 1 << level
    if tuple(occupancy.shape) != (G, G, G):
        raise RuntimeError(f"occupancy must be [{G}, {G}, {G}] for level {level}")
    return occupancy.to(torch.uint8).contiguous()


_pad_cache = {}


def _padding_positions(n, device):
    """[n, 3] fixed positions spread over the cube for the padding rows of a capped emit: the field is still evaluated on
    them (their gradient is zero), and identical positions would pile its scatter-add onto a handful of table rows."""
    key = (n, device.index)
    if key not in _pad_cache:
        # (built on the host: never inside a graph capture -- GraphedNerfFitter runs one eager step at every new capacity)
        if len(_pad_cache) >= 16:
            _pad_cache.clear()
        g = torch.Generator().manual_seed(12345)
        _pad_cache[key] = (torch.rand(n, 3, generator=g) * 2 - 1).to(device)
    return _pad_cache[key]


def raymarch_ray(origins, dirs, dist_min, dist_max, occupancy, level, num_samples, jitter=None, capacity=None):
    """`OctreeAS._raymarch_ray` (reference octree_as.py:235-290) on a dense occupancy grid [G, G, G] (bool, [x][y][z]).
    -> ridx int64 [S], samples [S, 3], depth_samples [S, 1], deltas [S, 1], boundary bool [S], and the per-ray pack
    offsets int64 [num_rays + 1] (rays without samples have empty packs) for callers that want to integrate without
    compacting the hit rays first.
    ``capacity`` (for steps captured into a HIP graph): no count is read back -- the outputs have exactly ``capacity``
    rows, survivors beyond it are dropped, the pack offsets are clamped to it, and the rows behind the last survivor are
    padding that belongs to no pack (ray 0, position 0, delta 0: whatever is computed for them integrates into nothing
    and receives a zero gradient). A seventh value is returned then: the true survivor count as a device int64 scalar."""
    _need_gpu(origins, dirs, occupancy)
    origins, dirs = origins.float().contiguous(), dirs.float().contiguous()
    N, dev = origins.shape[0], origins.device
    if jitter is None:
        jitter = torch.rand(N, num_samples, device=dev)
    jitter = jitter.float().contiguous()
    lin = torch.linspace(0, 1.0, num_samples, device=dev)
    occ = _occupancy_u8(occupancy, level)
    L = _lib.lib()
    with _on_device(dev):
        counts = torch.empty((N,), dtype=torch.int32, device=dev)
        args = (N, int(num_samples), _ptr(origins), _ptr(dirs), float(dist_min), float(dist_max), _ptr(lin),
                _ptr(jitter), _ptr(occ), int(level))
        _lib.check(L.shacira_raymarch_ray_count(*args, _ptr(counts), _stream(origins)), "shacira_raymarch_ray_count")
        offsets = torch.zeros((N + 1,), dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=offsets[1:])
        if capacity is not None:
            S = int(capacity)
            ridx = torch.zeros((S,), dtype=torch.int64, device=dev)
            samples = _padding_positions(S, dev).clone()     # rows behind the last survivor keep these
            depth = torch.zeros((S, 1), dtype=torch.float32, device=dev)
            deltas = torch.zeros((S, 1), dtype=torch.float32, device=dev)
            boundary = torch.zeros((S,), dtype=torch.uint8, device=dev)
            _lib.check(L.shacira_raymarch_ray_emit_capped(*args, _ptr(offsets), S, _ptr(ridx), _ptr(samples), _ptr(depth),
                                                          _ptr(deltas), _ptr(boundary), _stream(origins)),
                       "shacira_raymarch_ray_emit_capped")
            total = offsets[-1].clone()
            return ridx, samples, depth, deltas, boundary.bool(), offsets.clamp(max=S), total
        S = int(offsets[-1].item())
        ridx = torch.empty((S,), dtype=torch.int64, device=dev)
        samples = torch.empty((S, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((S, 1), dtype=torch.float32, device=dev)
        deltas = torch.empty((S, 1), dtype=torch.float32, device=dev)
        boundary = torch.empty((S,), dtype=torch.uint8, device=dev)
        _lib.check(L.shacira_raymarch_ray_emit(*args, _ptr(offsets), _ptr(ridx), _ptr(samples), _ptr(depth),
                                               _ptr(deltas), _ptr(boundary), _stream(origins)),
                   "shacira_raymarch_ray_emit")
    return ridx, samples, depth, deltas, boundary.bool(), offsets


def raytrace_dense(origins, dirs, occupancy, level):
    """Ray / occupied-cell intersections in depth order (stands in for kaolin's unbatched_raytrace(with_exit=True) on a
    one-level dense octree): ridx int32 [K], pidx int32 [K] (Morton index of the cell), depth [K, 2]."""
    _need_gpu(origins, dirs, occupancy)
    origins, dirs = origins.float().contiguous(), dirs.float().contiguous()
    N, dev = origins.shape[0], origins.device
    occ = _occupancy_u8(occupancy, level)
    L = _lib.lib()
    with _on_device(dev):
        counts = torch.empty((N,), dtype=torch.int32, device=dev)
        _lib.check(L.shacira_raytrace_dense_count(N, _ptr(origins), _ptr(dirs), _ptr(occ), int(level), _ptr(counts),
                                                  _stream(origins)), "shacira_raytrace_dense_count")
        offsets = torch.zeros((N + 1,), dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=offsets[1:])
        K = int(offsets[-1].item())
        ridx = torch.empty((K,), dtype=torch.int32, device=dev)
        pidx = torch.empty((K,), dtype=torch.int32, device=dev)
        depth = torch.empty((K, 2), dtype=torch.float32, device=dev)
        _lib.check(L.shacira_raytrace_dense_emit(N, _ptr(origins), _ptr(dirs), _ptr(occ), int(level), _ptr(offsets),
                                                 _ptr(ridx), _ptr(pidx), _ptr(depth), _stream(origins)),
                   "shacira_raytrace_dense_emit")
    return ridx, pidx, depth
