"""Mirror of the reference's ``wisp/ops/grid.py`` (:69-196): the autograd wrappers and functional API of the
hash-grid operator, bound to the MI355X C-ABI instead of ``wisp._C``.

Kept from the reference, because callers depend on it:
  * ``HashGridInterpolate`` / ``HashGridInterpolate2D``: ``forward(ctx, coords, resolutions, codebook_bitwidth,
    lod_idx, codebook, codebook_sizes, codebook_first_idx)`` and a 7-tuple ``backward`` whose only non-None entry is
    the codebook gradient (grid.py:111, :176) -- coordinates never receive a gradient;
  * odd feature dims raise ``Exception("The codebook feature dimension needs to be a multiple of 2.")`` (grid.py:75);
  * ``lod_idx`` and ``codebook_sizes`` are accepted and ignored by the operator (grid.py:79 TODO);
  * autocast: float inputs are cast to fp16 when autocast is active (``custom_fwd(cast_inputs=torch.half)``, grid.py:73).

Not kept: ``hashgrid_naive`` (deprecated in the reference, needs kaolin, not numerically equivalent).
The reference saves the (decoded) codebook for backward although only its shape and dtype are used; we save those.
"""
import torch

from ... import hip_ops


def _check_feature_dim(codebook):
    if codebook[0].shape[-1] % 2 == 1:
        raise Exception("The codebook feature dimension needs to be a multiple of 2.")


def _forward(ctx, dim, coords, resolutions, codebook_bitwidth, codebook, codebook_first_idx):
    _check_feature_dim(codebook)
    op = hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda
    # the batch's plan (its samples sorted by spatial block: what the forward of a large batch computes first) is kept with
    # the coordinates the reference saves (grid.py:86), so that the backward does not have to rediscover the batch's layout
    plan = None
    if coords.is_cuda and ctx.needs_input_grad[4]:   # (codebook is forward()'s fifth argument)
        plan = hip_ops.hashgrid_plan_buffer(dim, coords, codebook, resolutions, codebook_bitwidth)
    extra = {} if plan is None else {"plan": plan}   # (the operator's own signature when there is none: hashgrid_interpolate.h)
    feats_out = op(coords.float().contiguous(), codebook.contiguous(), codebook_first_idx, resolutions,
                   codebook_bitwidth, **extra).contiguous()
    ctx.save_for_backward(coords, codebook_first_idx)
    ctx.plan = plan
    ctx.resolutions = resolutions
    ctx.num_lods = len(resolutions)
    ctx.codebook_size = 2 ** codebook_bitwidth
    ctx.codebook_bitwidth = codebook_bitwidth
    ctx.feature_dim = codebook.shape[-1]
    ctx.table_rows = codebook.shape[0]
    ctx.table_dtype = codebook.dtype
    return feats_out


def _backward(ctx, dim, grad_output):
    coords, codebook_first_idx = ctx.saved_tensors
    grad_codebook = hip_ops.hashgrid_backward(dim, coords.float().contiguous(), grad_output.contiguous(),
                                              ctx.table_rows, ctx.table_dtype, codebook_first_idx, ctx.resolutions,
                                              ctx.codebook_bitwidth, ctx.feature_dim,
                                              **({} if ctx.plan is None else {"plan": ctx.plan}))
    return (None, None, None, None, grad_codebook, None, None)


class HashGridInterpolate(torch.autograd.Function):
    """3-D operator (reference grid.py:69-111)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, coords, resolutions, codebook_bitwidth, lod_idx, codebook, codebook_sizes, codebook_first_idx):
        return _forward(ctx, 3, coords, resolutions, codebook_bitwidth, codebook, codebook_first_idx)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        return _backward(ctx, 3, grad_output)


class HashGridInterpolate2D(torch.autograd.Function):
    """2-D operator (reference grid.py:135-176)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, coords, resolutions, codebook_bitwidth, lod_idx, codebook, codebook_sizes, codebook_first_idx):
        return _forward(ctx, 2, coords, resolutions, codebook_bitwidth, codebook, codebook_first_idx)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        return _backward(ctx, 2, grad_output)


def hashgrid(coords, resolutions, codebook_bitwidth, lod_idx, codebook, codebook_sizes, codebook_first_idx):
    """3-D hash-grid query + trilinear interpolation: coords [N,3] -> features [N, F*L] (reference grid.py:113-131)."""
    batch, _ = coords.shape
    feats = HashGridInterpolate.apply(coords.contiguous(), resolutions, codebook_bitwidth, lod_idx, codebook,
                                      codebook_sizes, codebook_first_idx)
    return feats.reshape(batch, codebook.shape[1] * len(resolutions))


def hashgrid2d(coords, resolutions, codebook_bitwidth, lod_idx, codebook, codebook_sizes, codebook_first_idx):
    """2-D hash-grid query + bilinear interpolation: coords [N,2] -> features [N, F*L] (reference grid.py:178-196)."""
    batch, _ = coords.shape
    feats = HashGridInterpolate2D.apply(coords.contiguous(), resolutions, codebook_bitwidth, lod_idx, codebook,
                                        codebook_sizes, codebook_first_idx)
    return feats.reshape(batch, codebook.shape[1] * len(resolutions))
