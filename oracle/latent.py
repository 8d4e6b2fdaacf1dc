"""numpy restatement of the reference's latent decode and entropy-bottleneck arithmetic. TEST INFRASTRUCTURE ONLY.

Follows (reference file:line):
  wisp/models/latent_decoders/basic_latent_decoder.py:12-19   get_dft_matrix
  wisp/models/latent_decoders/basic_latent_decoder.py:28-36   StraightThrough (round forward, identity backward)
  wisp/models/latent_decoders/basic_latent_decoder.py:86-91   DecoderLayer.forward ('sq' and 'dft')
  wisp/models/latent_decoders/basic_latent_decoder.py:192-198 LatentDecoder.forward, non-SGA path
  wisp/models/latent_decoders/basic_latent_decoder.py:183-191 LatentDecoder.forward, SGA path (stochastic Gumbel annealing;
      the sampler is torch.distributions RelaxedOneHotCategorical = exp(ExpRelaxedCategorical.rsample()):
      u = clamp(rand, eps, 1-eps); g = -log(-log u); score = (logits + g)/T; sample = exp(score - logsumexp(score)))
  wisp/models/prob_models/bit_estimator.py:27-44, :58-65      Bitparm / BitEstimator forward
  wisp/models/grids/latent_grid.py:122-136                    LatentGrid.ent_loss

Backward passes are written out by hand (not autograd). PARITY: pinned -- tests/test_oracle_golden.py checks every
function here against vectors produced by running the reference's own modules (tests/golden/make_golden.py).
Forward math is float32 like the reference; reductions over the table are accumulated in float64.
"""
import numpy as np

f32 = np.float32


def dft_matrix(latent_dim, feature_dim):
    m = np.zeros((latent_dim, feature_dim), np.float64)
    for i in range(latent_dim):
        for j in range(feature_dim):
            m[i, j] = np.cos(np.pi / feature_dim * (i + 0.5) * j) / np.sqrt(feature_dim)
    m = m.astype(f32)
    m[:, 1:] = m[:, 1:] * f32(np.sqrt(2))
    return m


def decode_forward(latent, div, matrix, colscale=None, shift=None, clamp_weights=0.0):
    """q = rint(latent) (half to even); z = q/div; y = (z @ matrix) * colscale + shift; optional clamp."""
    z = (np.rint(latent.astype(f32)) / div.astype(f32)).astype(f32)
    zm = (z @ matrix.astype(f32)).astype(f32)
    y = zm
    if colscale is not None:
        y = (y * colscale.reshape(1, -1).astype(f32)).astype(f32)
    if shift is not None:
        y = (y + shift.reshape(1, -1).astype(f32)).astype(f32)
    out = np.clip(y, -clamp_weights, clamp_weights).astype(f32) if clamp_weights > 0 else y
    return out, (z, zm, y)


def decode_backward(latent, div, matrix, colscale, shift, clamp_weights, grad_out):
    """-> dict(latent, matrix, colscale, shift); straight-through rounding, clamp passes gradient on [-c, c]."""
    _, (z, zm, y) = decode_forward(latent, div, matrix, colscale, shift, clamp_weights)
    gy = grad_out.astype(np.float64).copy()
    if clamp_weights > 0:
        gy[(y < -clamp_weights) | (y > clamp_weights)] = 0.0
    cs = np.ones(matrix.shape[1]) if colscale is None else colscale.reshape(-1).astype(np.float64)
    gs = gy * cs.reshape(1, -1)
    return dict(
        latent=((gs @ matrix.astype(np.float64).T) / div.astype(np.float64).reshape(1, -1)).astype(f32),
        matrix=(z.astype(np.float64).T @ gs).astype(f32),
        colscale=(gy * zm.astype(np.float64)).sum(0).astype(f32),
        shift=gy.sum(0).astype(f32),
    )


def sga_quantise(latent, uniforms, temperature, diff_sampling=True):
    """SGA sample between floor(w) and floor(w)+1 and its derivative w.r.t. w. uniforms: [T, ld, 2] in [0, 1).
    float64 inside; -> (q [T, ld] float32, dq_dw [T, ld] float64).
    diff_sampling: rsample() through the relaxed categorical, floor without gradient; otherwise sample() (no gradient
    through the sampler) and a straight-through floor (dq/dw = s0 + s1)."""
    w = latent.astype(np.float64)
    T = float(temperature)
    lim = 1.0 - 1e-6
    wf = np.floor(w)
    wc = wf + 1.0
    df = np.clip(w - wf, -lim, lim)
    dc = np.clip(wc - w, -lim, lim)
    lf, lc = -np.tanh(df) / T, -np.tanh(dc) / T
    eps = float(np.finfo(np.float32).eps)
    u = np.clip(uniforms.astype(np.float64), eps, 1.0 - eps)
    g = -np.log(-np.log(u))
    z0, z1 = (lf + g[..., 0]) / T, (lc + g[..., 1]) / T
    m = np.maximum(z0, z1)
    lse = m + np.log(np.exp(z0 - m) + np.exp(z1 - m))
    s0, s1 = np.exp(z0 - lse), np.exp(z1 - lse)
    q = wf * s0 + wc * s1
    if diff_sampling:
        in_f = (w - wf > -lim) & (w - wf < lim)
        in_c = (wc - w > -lim) & (wc - w < lim)
        dz = ((1.0 - np.tanh(dc) ** 2) * in_c + (1.0 - np.tanh(df) ** 2) * in_f) / (T * T)   # d(z1 - z0)/dw
        dq = (wc - wf) * s0 * s1 * dz
    else:
        dq = s0 + s1
    return q.astype(f32), dq


def decode_sga_forward(latent, uniforms, temperature, diff_sampling, div, matrix, colscale=None, shift=None,
                       clamp_weights=0.0):
    q, _ = sga_quantise(latent, uniforms, temperature, diff_sampling)
    z = (q / div.astype(f32)).astype(f32)
    zm = (z @ matrix.astype(f32)).astype(f32)
    y = zm
    if colscale is not None:
        y = (y * colscale.reshape(1, -1).astype(f32)).astype(f32)
    if shift is not None:
        y = (y + shift.reshape(1, -1).astype(f32)).astype(f32)
    return (np.clip(y, -clamp_weights, clamp_weights).astype(f32) if clamp_weights > 0 else y), (z, zm, y)


def decode_sga_backward(latent, uniforms, temperature, diff_sampling, div, matrix, colscale, shift, clamp_weights,
                        grad_out):
    _, dq = sga_quantise(latent, uniforms, temperature, diff_sampling)
    _, (z, zm, y) = decode_sga_forward(latent, uniforms, temperature, diff_sampling, div, matrix, colscale, shift,
                                       clamp_weights)
    gy = grad_out.astype(np.float64).copy()
    if clamp_weights > 0:
        gy[(y < -clamp_weights) | (y > clamp_weights)] = 0.0
    cs = np.ones(matrix.shape[1]) if colscale is None else colscale.reshape(-1).astype(np.float64)
    gs = gy * cs.reshape(1, -1)
    gq = (gs @ matrix.astype(np.float64).T) / div.astype(np.float64).reshape(1, -1)
    return dict(latent=(gq * dq).astype(f32), matrix=(z.astype(np.float64).T @ gs).astype(f32),
                colscale=(gy * zm.astype(np.float64)).sum(0).astype(f32), shift=gy.sum(0).astype(f32))


def _softplus(h):
    h = h.astype(np.float64)
    return np.where(h > 20.0, h, np.log1p(np.exp(np.minimum(h, 20.0))))


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def cdf(x, params, num_layers):
    """params: float [4, 3, C] = (f1.h, f1.b, f1.a, ..., f4.h, f4.b, unused). x: [T, C]. float64 math."""
    x = x.astype(np.float64)
    trace = []
    for k in range(3):
        if num_layers > k + 1:
            sp, b, ta = _softplus(params[k, 0]), params[k, 1].astype(np.float64), np.tanh(params[k, 2].astype(np.float64))
            u = x * sp + b
            th = np.tanh(u)
            trace.append((k, x, th))
            x = u + th * ta
    s = _sigmoid(x * _softplus(params[3, 0]) + params[3, 1].astype(np.float64))
    return s, (trace, x)


def _cdf_backward(params, trace_and_x, s, g):
    """g = dL/ds [T,C] -> (dL/dx_in [T,C], dparams [4,3,C])."""
    trace, xin4 = trace_and_x
    dp = np.zeros(params.shape, np.float64)
    sgh = lambda h: np.where(h > 20.0, 1.0, _sigmoid(h.astype(np.float64)))
    du = g * s * (1.0 - s)
    dp[3, 0] = (du * xin4).sum(0) * sgh(params[3, 0])
    dp[3, 1] = du.sum(0)
    dx = du * _softplus(params[3, 0])
    for (k, xin, th) in reversed(trace):
        ta = np.tanh(params[k, 2].astype(np.float64))
        dp[k, 2] = (dx * th).sum(0) * (1.0 - ta * ta)
        du = dx * (1.0 + (1.0 - th * th) * ta)
        dp[k, 0] = (du * xin).sum(0) * sgh(params[k, 0])
        dp[k, 1] = du.sum(0)
        dx = du * _softplus(params[k, 0])
    return dx, dp


def entropy_bits(latent, noise, params, num_layers):
    """total bits (float64 scalar); noise None -> validation rule round(latent)."""
    w = (latent.astype(f32) + noise.astype(f32)).astype(f32) if noise is not None else np.rint(latent.astype(f32))
    sp, _ = cdf((w + f32(0.5)).astype(f32), params, num_layers)
    sn, _ = cdf((w - f32(0.5)).astype(f32), params, num_layers)
    prob = sp - sn
    bits = np.clip(-np.log(prob + 1e-10) / np.log(2.0), 0, 50)
    return float(bits.sum())


def entropy_bits_backward(latent, noise, params, num_layers, grad_total=1.0):
    """-> (grad_latent [T,C] (zeros in validation mode), grad_params [4,3,C])."""
    w = (latent.astype(f32) + noise.astype(f32)).astype(f32) if noise is not None else np.rint(latent.astype(f32))
    sp, tp = cdf((w + f32(0.5)).astype(f32), params, num_layers)
    sn, tn = cdf((w - f32(0.5)).astype(f32), params, num_layers)
    q = sp - sn + 1e-10
    bits = -np.log(q) / np.log(2.0)
    gp = np.where((bits >= 0) & (bits <= 50), -1.0 / (q * np.log(2.0)), 0.0) * grad_total
    dxp, dpp = _cdf_backward(params, tp, sp, gp)
    dxn, dpn = _cdf_backward(params, tn, sn, -gp)
    glat = (dxp + dxn) if noise is not None else np.zeros_like(dxp)
    return glat.astype(f32), (dpp + dpn).astype(f32)


def pack_params(state, prefix, channels):
    """Collect f1..f4 {h,b,a} from a dict of arrays (golden npz / state_dict) into [4,3,C]."""
    out = np.zeros((4, 3, channels), f32)
    for k, f in enumerate(("f1", "f2", "f3", "f4")):
        out[k, 0] = np.asarray(state[f"{prefix}{f}.h"]).reshape(-1)
        out[k, 1] = np.asarray(state[f"{prefix}{f}.b"]).reshape(-1)
        if f != "f4":
            out[k, 2] = np.asarray(state[f"{prefix}{f}.a"]).reshape(-1)
    return out


def size_bits(latent):
    """LatentGrid.size(use_torchac=False, use_prob_model=False) latent part (latent_grid.py:141-153)."""
    total = 0.0
    T = latent.shape[0]
    for c in range(latent.shape[1]):
        q = np.rint(latent[:, c].astype(f32)).astype(np.int64)
        _, counts = np.unique(q, return_counts=True)
        p = (counts / counts.sum()).astype(f32)
        info = np.clip(-np.log(p + f32(1e-10)) / f32(np.log(2.0)), 0, 1000).astype(f32)
        total += float((info * counts.astype(f32)).sum())
    return total
