"""``PackedRFTracer``: march rays through the grid's occupancy, query the radiance field at the samples, composite
(reference wisp/tracers/packed_rf_tracer.py:68-170). Sample generation, the field's hash-grid lookup and the volume
integration each run as HIP kernels (shacira_amd/render.py, hip_ops.py); this file is the glue in the reference's order.
"""
import torch

from ... import render as spc_render
from ..core import RenderBuffer


class PackedRFTracer:
    def __init__(self, raymarch_type="voxel", num_steps=64, step_size=1.0, bg_color="white"):
        self.raymarch_type, self.num_steps, self.step_size, self.bg_color = raymarch_type, num_steps, step_size, bg_color

    def get_supported_channels(self):
        return {"depth", "hit", "rgb", "alpha"}

    def get_required_nef_channels(self):
        return {"rgb", "density"}

    def __call__(self, nef, rays, channels=("rgb",), extra_channels=(), lod_idx=None, **overrides):
        opts = dict(raymarch_type=self.raymarch_type, num_steps=self.num_steps, step_size=self.step_size,
                    bg_color=self.bg_color)
        opts.update(overrides)
        return self.trace(nef, rays, set(channels), set(extra_channels), lod_idx=lod_idx, **opts)

    def trace(self, nef, rays, channels, extra_channels, lod_idx=None, raymarch_type="voxel", num_steps=64,
              step_size=1.0, bg_color="white"):
        assert nef.grid is not None, "this tracer requires a grid"
        N, dev = rays.origins.shape[0], rays.origins.device
        depth = torch.zeros(N, 1, device=dev) if "depth" in channels else None
        rgb = torch.ones(N, 3, device=dev) if bg_color == "white" else torch.zeros(N, 3, device=dev)
        hit = torch.zeros(N, device=dev, dtype=torch.bool)
        out_alpha = torch.zeros(N, 1, device=dev)
        if lod_idx is None:
            lod_idx = nef.grid.num_lods - 1

        marched = nef.grid.raymarch(rays, level=nef.grid.active_lods[lod_idx], num_samples=num_steps,
                                    raymarch_type=raymarch_type)
        ridx, samples, deltas, boundary = marched.ridx, marched.samples, marched.deltas, marched.boundary
        num_samples = samples.shape[0]
        if num_samples == 0:
            return RenderBuffer(depth=depth, hit=hit, rgb=rgb, alpha=out_alpha)
        hit_ray_d = rays.dirs.index_select(0, ridx)
        field = nef(coords=samples, ray_d=hit_ray_d, lod_idx=lod_idx, channels=["rgb", "density"])
        color, density = field["rgb"], field["density"].reshape(num_samples, 1)
        tau = density * deltas                                   # optical thickness

        offsets = getattr(marched, "ray_offsets", None)
        if offsets is not None and not extra_channels:
            # the marcher's per-ray pack offsets cover EVERY ray (empty packs integrate to zero), so the reductions
            # come out per ray directly: no nonzero() read-back, no scatter of the hit rays (same values as below)
            ray_colors, transmittance = spc_render.exponential_integration(color, tau, boundary, exclusive=True,
                                                                           pack_start=offsets)
            alpha = spc_render.sum_reduce(transmittance, boundary, pack_start=offsets)
            if depth is not None:
                depth = spc_render.sum_reduce(marched.depth_samples.reshape(num_samples, 1) * transmittance, boundary,
                                              pack_start=offsets)
            rgb = (1.0 - alpha) + ray_colors if bg_color == "white" else alpha * ray_colors
            return RenderBuffer(depth=depth, hit=alpha[..., 0] > 0.0, rgb=rgb, alpha=alpha)

        ridx_hit = ridx[boundary]
        pack_start = spc_render.pack_offsets(boundary)           # one scan of the boundary flags for all reductions
        ray_colors, transmittance = spc_render.exponential_integration(color, tau, boundary, exclusive=True,
                                                                       pack_start=pack_start)
        if depth is not None:
            depth[ridx_hit, :] = spc_render.sum_reduce(marched.depth_samples.reshape(num_samples, 1) * transmittance,
                                                       boundary, pack_start=pack_start)
        alpha = spc_render.sum_reduce(transmittance, boundary, pack_start=pack_start)
        out_alpha[ridx_hit] = alpha
        hit[ridx_hit] = alpha[..., 0] > 0.0
        rgb[ridx_hit] = (1.0 - alpha) + ray_colors if bg_color == "white" else alpha * ray_colors

        extra_outputs = {}
        for channel in extra_channels:
            feats = nef(coords=samples, ray_d=hit_ray_d, lod_idx=lod_idx, channels=[channel])[channel]
            ray_feats, _ = spc_render.exponential_integration(feats.view(num_samples, -1), tau, boundary,
                                                              exclusive=True, pack_start=pack_start)
            out_feats = torch.zeros(N, feats.shape[-1], device=dev)
            out_feats[ridx_hit] = alpha * ray_feats
            extra_outputs[channel] = out_feats
        return RenderBuffer(depth=depth, hit=hit, rgb=rgb, alpha=out_alpha, **extra_outputs)
