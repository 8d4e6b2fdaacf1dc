"""ctypes binding of the scalar C oracle (``oracle/hashgrid_oracle.c``). TEST INFRASTRUCTURE ONLY.

``build()`` compiles it with gcc through ``oracle/Makefile`` into ``oracle/_build/``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libshacira_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "hashgrid_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i32, i64, f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        p = ctypes.c_void_p
        L.shacira_oracle_hash_index3.restype = i32
        L.shacira_oracle_hash_index3.argtypes = [i32] * 5
        L.shacira_oracle_hash_index2.restype = i32
        L.shacira_oracle_hash_index2.argtypes = [i32] * 4
        L.shacira_oracle_hashgrid_fwd.restype = None
        L.shacira_oracle_hashgrid_fwd.argtypes = [ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  p, p, i64, p, p, p, p, p]
        L.shacira_oracle_hashgrid_fwd_last64.restype = None
        L.shacira_oracle_hashgrid_fwd_last64.argtypes = [ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         p, p, i64, p, p, p]
        for name in ("shacira_oracle_hashgrid_bwd", "shacira_oracle_hashgrid_bwd_f32", "shacira_oracle_hashgrid_bwd_f64"):
            fn = getattr(L, name)
            fn.restype = None
            fn.argtypes = [ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, p, p, i64, p, p, p]
        L.shacira_oracle_axis.restype = None
        L.shacira_oracle_axis.argtypes = [f32, i32, p, p, p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def hash_index3(x, y, z, res, cs):
    return int(lib().shacira_oracle_hash_index3(x, y, z, res, cs))


def hash_index2(x, y, res, cs):
    return int(lib().shacira_oracle_hash_index2(x, y, res, cs))


def axis(c, res):
    pos = np.zeros(1, np.int32)
    fr = np.zeros(1, np.float32)
    ifr = np.zeros(1, np.float32)
    lib().shacira_oracle_axis(float(np.float32(c)), int(res), _ptr(pos), _ptr(fr), _ptr(ifr))
    return int(pos[0]), np.float32(fr[0]), np.float32(ifr[0])


def _prep(coords, table, resolutions, first_idx):
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    table = np.ascontiguousarray(table, dtype=np.float32)
    res = np.ascontiguousarray(resolutions, dtype=np.int32)
    fi = np.ascontiguousarray(first_idx, dtype=np.int32)
    assert coords.ndim == 2 and coords.shape[1] in (2, 3)
    assert table.ndim == 2 and res.shape == fi.shape
    return coords, table, res, fi


def forward(coords, table, first_idx, resolutions, bitwidth, want_corners=False):
    """feats [N, L*F] float32 (and optionally level-local corner rows / weights [N, L, 2^d])."""
    coords, table, res, fi = _prep(coords, table, resolutions, first_idx)
    N, dim = coords.shape
    T, F = table.shape
    L = len(res)
    feats = np.empty((N, L * F), np.float32)
    idx = np.empty((N, L, 1 << dim), np.int32) if want_corners else None
    w = np.empty((N, L, 1 << dim), np.float32) if want_corners else None
    lib().shacira_oracle_hashgrid_fwd(dim, N, L, F, int(bitwidth), _ptr(res), _ptr(fi), T, _ptr(coords),
                                      _ptr(table), _ptr(feats), _ptr(idx), _ptr(w))
    return (feats, idx, w) if want_corners else feats


def forward_half_llvm(coords, table16, first_idx, resolutions, bitwidth):
    """fp16 table as the HIPCC build of the reference evaluates it (oracle/_ref; read from the ISA of the built code object
    and confirmed on the vectors): the feature loop of .cu:96-107 is unrolled by four -- features j < 4*(F//4): packed
    fp32 fmas in LLVM's contraction order (mode 1), rounded to fp32, then converted to half; the remainder loop
    (j >= 4*(F//4)): t0*c0 and t2*c2 from one packed multiply, fma(t1,c1, t0*c0), a plain ADD of t2*c2, then a `v_fma_mix_f32`
    chain whose last step is fused with the conversion into `v_fma_mixlo_f16` = ONE rounding of the exact sum (mode 2). Explains tests/golden/ref_kernels.npz's fp16 cases bit for bit; NOT
    what the product is held to (nvcc rounds to fp32 first: `forward(...).astype(float16)`)."""
    coords, table, res, fi = _prep(coords, np.asarray(table16, dtype=np.float16).astype(np.float32), resolutions, first_idx)
    N, dim = coords.shape
    T, F = table.shape
    L = len(res)
    pair = np.empty((N, L * F, 2), np.float64)
    old = set_contraction(1)
    try:
        plain = forward(coords, table, fi, res, bitwidth).astype(np.float16)
        set_contraction(2)
        lib().shacira_oracle_hashgrid_fwd_last64(dim, N, L, F, int(bitwidth), _ptr(res), _ptr(fi), T, _ptr(coords),
                                                 _ptr(table), _ptr(pair))
    finally:
        set_contraction(old)
    # one rounding of s + e (exact) to half: RNE of the double s is right unless s is a half-way point and e != 0
    s, e = pair[..., 0], pair[..., 1]
    h = s.astype(np.float16)
    hd = h.astype(np.float64)
    other = np.nextafter(h, np.where(s > hd, np.float16(np.inf), np.float16(-np.inf)).astype(np.float16))
    od = other.astype(np.float64)
    tie = (s != hd) & (s - hd == od - s)
    lo16 = np.where(hd <= od, h, other)
    hi16 = np.where(hd <= od, other, h)
    fused = np.where(tie & (e < 0), lo16, np.where(tie & (e > 0), hi16, h))
    jj = np.arange(L * F) % F
    return np.where((jj >= 4 * (F // 4))[None, :], fused, plain).astype(np.float16)


def set_contraction(mode):
    """0 (default): nvcc's contraction order of the feature sum; 1: LLVM's, i.e. what the reference's kernels compute when
    built by hipcc (oracle/_ref). See the header of hashgrid_oracle.c. Returns the previous mode."""
    old = lib().shacira_oracle_get_contraction()
    lib().shacira_oracle_set_contraction(int(mode))
    return old


def backward(coords, grad_out, table_shape, first_idx, resolutions, bitwidth, accumulate="f64"):
    """grad_table [T, F]; float64 (sample-order double accumulation) or float32 (sample-order fp32)."""
    T, F = table_shape
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    res = np.ascontiguousarray(resolutions, dtype=np.int32)
    fi = np.ascontiguousarray(first_idx, dtype=np.int32)
    N, dim = coords.shape
    L = len(res)
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32).reshape(N, L * F)
    if accumulate == "f64":
        out = np.zeros((T, F), np.float64)
        fn = lib().shacira_oracle_hashgrid_bwd
    else:
        out = np.zeros((T, F), np.float32)
        fn = lib().shacira_oracle_hashgrid_bwd_f32
    fn(dim, N, L, F, int(bitwidth), _ptr(res), _ptr(fi), T, _ptr(coords), _ptr(grad_out), _ptr(out))
    return out


def half_round(x):
    """The C model's float -> binary16 -> float rounding (RNE), element-wise (checked against numpy's float16 in the tests)."""
    L = lib()
    L.shacira_oracle_half_round.restype = ctypes.c_float
    L.shacira_oracle_half_round.argtypes = [ctypes.c_float]
    return np.array([L.shacira_oracle_half_round(float(v)) for v in np.asarray(x, dtype=np.float32).ravel()],
                    dtype=np.float32).reshape(np.shape(x))


def backward_half_model(coords, grad_out16, table_shape, first_idx, resolutions, bitwidth):
    """Software model of the reference's `__half2` backward (hashgrid_interpolate_cuda.cu:198-211): every fp32 product
    rounded to half, the table entry takes `half(entry + product)` per atomicAdd, sample order. Returns
    (table float32 holding half values, bound, sumabs, sumg): `bound` = this schedule's rigorous |model - exact sum| bound,
    `sumabs` = sum of |fp32 products| per entry (what any schedule's rounding walk is bounded with), `sumg` = sum of the
    |gradients| that reach the entry (bounds the effect of a quantised weight)."""
    T, F = table_shape
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    res = np.ascontiguousarray(resolutions, dtype=np.int32)
    fi = np.ascontiguousarray(first_idx, dtype=np.int32)
    N, dim = coords.shape
    L = len(res)
    go = np.ascontiguousarray(np.asarray(grad_out16, dtype=np.float16).astype(np.float32)).reshape(N, L * F)
    out = np.zeros((T, F), np.float32)
    bound = np.zeros((T, F), np.float32)
    sumabs = np.zeros((T, F), np.float32)
    sumg = np.zeros((T, F), np.float32)
    fn = lib().shacira_oracle_hashgrid_bwd_half_model
    p = ctypes.c_void_p
    fn.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, p, p, ctypes.c_int64, p, p, p, p, p, p]
    fn.restype = None
    fn(dim, N, L, F, int(bitwidth), _ptr(res), _ptr(fi), T, _ptr(coords), _ptr(go), _ptr(out), _ptr(bound), _ptr(sumabs),
       _ptr(sumg))
    return out, bound, sumabs, sumg


def forward_f64(coords, table64, first_idx, resolutions, bitwidth):
    """scalar_t = double (the third type of the reference's dispatch, hashgrid_interpolate_cuda.cu:125): every table value is
    narrowed with static_cast<float>, the interpolation runs in fp32 and the result is widened (.cu:96-107)."""
    return forward(coords, np.asarray(table64, dtype=np.float64).astype(np.float32), first_idx, resolutions,
                   bitwidth).astype(np.float64)


def backward_f64(coords, grad_out64, table_shape, first_idx, resolutions, bitwidth):
    """grad_table [T, F] float64 for a double table: float products of double gradients (.cu:215-217), summed in double --
    the INTENDED result; the reference's own kernel adds them into the low words of the doubles (see hashgrid_oracle.c)."""
    T, F = table_shape
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    res = np.ascontiguousarray(resolutions, dtype=np.int32)
    fi = np.ascontiguousarray(first_idx, dtype=np.int32)
    N, dim = coords.shape
    L = len(res)
    grad_out64 = np.ascontiguousarray(grad_out64, dtype=np.float64).reshape(N, L * F)
    out = np.zeros((T, F), np.float64)
    lib().shacira_oracle_hashgrid_bwd_f64(dim, N, L, F, int(bitwidth), _ptr(res), _ptr(fi), T, _ptr(coords),
                                          _ptr(grad_out64), _ptr(out))
    return out
