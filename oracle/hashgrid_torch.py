"""Pure-PyTorch restatement of the reference hash-grid kernels. TEST INFRASTRUCTURE ONLY.

Plays two roles: (1) an independent second formulation that cross-checks the scalar C oracle, and
(2) the "reference pure-PyTorch CPU path" that ``bench.py`` times as ``cpu_baseline`` (kind "port").

Follows wisp/csrc/ops/hashgrid_interpolate_cuda.cu:17-109, :143-221 and
wisp/csrc/ops/hashgrid_interpolate2d_cuda.cu:17-99, :133-208 (see hashgrid_oracle.c for the arithmetic notes).
Index math is done in int64 and masked to 32 bits; coordinates are scaled in fp64 and narrowed to fp32.
Backward is whatever autograd derives for the gather (``index_add_`` in fp32, sample order on CPU).
"""
import torch

_P1 = 2654435761
_P2 = 805459861
_M32 = 0xFFFFFFFF


def _wrap_i32(v):
    """int64 tensor/py-int -> value reinterpreted as int32 after uint32 wraparound."""
    v = v & _M32
    return v - ((v >> 31) << 32)


def _is_dense(res, cs, dim):
    r2 = _wrap_i32(res * res)
    if dim == 2:
        return res < cs and r2 < cs
    return res < cs and r2 < cs and _wrap_i32(r2 * res) < cs


def corner_rows_and_weights(coords, res, cs):
    """coords [N,d] fp32 -> (rows int64 [N, 2^d] level-local, weights fp32 [N, 2^d])."""
    N, dim = coords.shape
    hi = torch.tensor(float(res) - 1.0 - 1e-5, dtype=torch.float64).to(torch.float32)
    x = (float(res) * (coords.double() * 0.5 + 0.5)).to(torch.float32)
    # CUDA min/max: NaN operand -> the other operand
    x = torch.where(torch.isnan(x), hi.expand_as(x), torch.minimum(x, hi))
    x = torch.maximum(x, torch.zeros((), dtype=torch.float32))
    pos = torch.floor(x)
    frac = x - pos
    ifrac = (1.0 - frac.double()).to(torch.float32)
    pos = pos.to(torch.int64)
    dense = _is_dense(res, cs, dim)
    rows, weights = [], []
    for j in range(1 << dim):
        bits = [(j >> (dim - 1 - a)) & 1 for a in range(dim)]  # MSB -> x
        w = None
        for a in range(dim):
            wa = frac[:, a] if bits[a] else ifrac[:, a]
            w = wa if w is None else w * wa
        c = [pos[:, a] + bits[a] for a in range(dim)]
        if dense:
            r = c[0] + c[1] * res
            if dim == 3:
                r = r + c[2] * res * res
            r = _wrap_i32(r)
        else:
            h = (c[0] & _M32) ^ ((c[1] * _P1) & _M32)
            if dim == 3:
                h = h ^ ((c[2] * _P2) & _M32)
            r = h % cs
        rows.append(r)
        weights.append(w)
    return torch.stack(rows, 1), torch.stack(weights, 1)


def hashgrid_forward(coords, table, first_idx, resolutions, bitwidth):
    """coords [N,d] fp32, table [T,F] fp32 (requires_grad ok) -> feats [N, L*F] fp32."""
    cs = 2 ** int(bitwidth)
    T = table.shape[0]
    first = [int(v) for v in first_idx]
    out = []
    for l, res in enumerate(resolutions):
        rows, w = corner_rows_and_weights(coords, int(res), cs)
        g = rows + first[l]
        ok = (g >= 0) & (g < T)
        vals = table[g.clamp(0, T - 1)]                      # [N, 2^d, F]
        vals = vals * (w * ok.to(w.dtype)).unsqueeze(-1)
        out.append(vals.sum(1))
    return torch.cat(out, 1)


def hashgrid_fwd_bwd(coords, table, first_idx, resolutions, bitwidth, grad_out):
    """One fwd+bwd pass (the timed unit of the CPU baseline). Returns (feats, grad_table)."""
    table = table.detach().requires_grad_(True)
    feats = hashgrid_forward(coords, table, first_idx, resolutions, bitwidth)
    feats.backward(grad_out)
    return feats.detach(), table.grad
