"""Known-answer tests that pin the CPU oracle to the reference SOURCE (there are no reference tests/goldens for the
CUDA kernels: "parity unpinned"). Every expected value below is derived by hand from
hashgrid_interpolate_cuda.cu / hashgrid_interpolate2d_cuda.cu, not from the oracle itself."""
import numpy as np
import pytest
import torch

from conftest import CONFIGS, geo, table_layout
from oracle import hashgrid_c as oc
from oracle import hashgrid_torch as ot

P1, P2 = 2654435761, 805459861


def test_hash_index_known_answers():
    cs = 1 << 19
    # .cu:34-36 evaluated with Python big ints then truncated to 32 bits
    assert oc.hash_index3(100, 200, 300, 2049, cs) == 110768
    assert oc.hash_index3(2048, 2048, 2048, 2049, cs) == 75776
    for (x, y, z) in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (2049, 2049, 2049), (7, 11, 13)]:
        want = ((x & 0xFFFFFFFF) ^ ((y * P1) & 0xFFFFFFFF) ^ ((z * P2) & 0xFFFFFFFF)) % cs
        assert oc.hash_index3(x, y, z, 2049, cs) == want
        assert oc.hash_index2(x, y, 2049, 1 << 11) == ((x ^ ((y * P1) & 0xFFFFFFFF)) % (1 << 11))


def test_dense_index_and_switch():
    cs = 1 << 19
    assert oc.hash_index3(3, 4, 5, 17, cs) == 3 + 4 * 17 + 5 * 17 * 17     # 17^3 < 2^19 -> dense
    assert oc.hash_index3(3, 4, 5, 59, cs) == 3 + 4 * 59 + 5 * 59 * 59     # 59^3 = 205379 < 2^19
    assert oc.hash_index3(3, 4, 5, 81, cs) != 3 + 4 * 81 + 5 * 81 * 81     # 81^3 = 531441 > 2^19 -> hashed
    assert oc.hash_index2(3, 4, 562, cs) == 3 + 4 * 562                    # 562^2 < 2^19
    assert oc.hash_index2(3, 4, 777, cs) == ((3 ^ (4 * P1 & 0xFFFFFFFF)) % cs)
    # strict '<': res^3 == cs is hashed (8^3 == 2^9)
    assert oc.hash_index3(1, 1, 1, 8, 512) == ((1 ^ (P1 & 0xFFFFFFFF) ^ P2) % 512)
    # dense levels of the SURVEY table
    for name, dense_upto in (("A", 2), ("B", 4), ("Bp", 11), ("D", 4)):
        dim, res, bw = CONFIGS[name]
        flags = [ot._is_dense(r, 2 ** bw, dim) for r in res]
        assert flags == [l <= dense_upto for l in range(len(res))], name


def test_int32_overflow_of_dense_test_matches_wraparound():
    # .cu:29: res*res*res in int32. res=2049, cs=2^30: 2049^2 = 4198401 < cs, 2049^3 = 8602523649 wraps to
    # 8602523649 - 2*2^32 = 12589057 < cs -> the kernel takes the DENSE branch although the grid has > cs cells.
    assert ot._is_dense(2049, 1 << 30, 3) is True
    assert oc.hash_index3(1, 2, 3, 2049, 1 << 30) == 1 + 2 * 2049 + 3 * 2049 * 2049


@pytest.mark.parametrize("res", [17, 257, 258, 513, 2049])
def test_clamp_and_floor_edges(res):
    # upper clamp (float)(res-1-1e-5): rounds to exactly res-1 for res >= 258 (fp32 spacing), below for small res
    hi = np.float32(np.float64(res) - 1.0 - 1e-5)
    assert (hi == np.float32(res - 1)) == (res >= 258)
    pos, fr, ifr = oc.axis(1.0, res)        # coord +1 -> res*(1.0) = res -> clamped to hi
    assert pos == int(np.floor(hi)) and fr == np.float32(hi - np.floor(hi))
    pos, fr, ifr = oc.axis(-1.0, res)       # coord -1 -> 0
    assert (pos, float(fr), float(ifr)) == (0, 0.0, 1.0)
    pos, _, _ = oc.axis(5.0, res)
    assert pos == int(np.floor(hi))
    pos, fr, _ = oc.axis(-7.0, res)
    assert pos == 0 and fr == 0
    pos, _, _ = oc.axis(float("nan"), res)  # CUDA fminf(b, NaN) = b
    assert pos == int(np.floor(hi))
    c = np.float32(1.0) - np.float32(2.0 ** -24)
    want = np.float32(np.float64(res) * (np.float64(c) * 0.5 + 0.5))
    pos, fr, _ = oc.axis(c, res)
    assert pos == int(np.floor(min(want, hi)))


def test_fp64_scale_differs_from_fp32_scale():
    # the implicit double arithmetic matters: at res 2049 some coords floor differently when scaled in fp32
    rng = np.random.default_rng(0)
    c = rng.uniform(-1, 1, 2_000_000).astype(np.float32)
    x64 = (np.float64(2049) * (c.astype(np.float64) * 0.5 + 0.5)).astype(np.float32)
    x32 = np.float32(2049) * (c * np.float32(0.5) + np.float32(0.5))
    flips = int((np.floor(x64) != np.floor(x32)).sum())
    assert flips > 0
    pos = np.array([oc.axis(v, 2049)[0] for v in c[np.floor(x64) != np.floor(x32)][:50]])
    hi = np.float32(2048.0)
    assert (pos == np.floor(np.minimum(x64[np.floor(x64) != np.floor(x32)][:50], hi)).astype(int)).all()


@pytest.mark.parametrize("name", ["A", "B", "Bp", "D"])
def test_c_and_torch_oracles_agree(name):
    dim, res, bw = CONFIGS[name]
    sizes, first, T = table_layout(res, bw, dim)
    rng = np.random.default_rng(1)
    N = 3000
    coords = rng.uniform(-1, 1, (N, dim)).astype(np.float32)
    coords[:3] = 1.0
    coords[3:6] = -1.0
    coords[6] = np.nan
    coords[7] = 3.0
    table = (rng.standard_normal((T, 2)) * 0.01).astype(np.float32)
    feats, idx, w = oc.forward(coords, table, first, res, bw, want_corners=True)
    for l, r in enumerate(res):
        rows, ww = ot.corner_rows_and_weights(torch.from_numpy(coords), r, 2 ** bw)
        assert (rows.numpy() == idx[:, l]).all()
        assert (ww.numpy() == w[:, l]).all()
    tf = ot.hashgrid_forward(torch.from_numpy(coords), torch.from_numpy(table), first, res, bw).numpy()
    np.testing.assert_allclose(feats, tf, rtol=1e-5, atol=1e-8)
    go = rng.standard_normal((N, len(res) * 2)).astype(np.float32)
    g64 = oc.backward(coords, go, table.shape, first, res, bw)
    _, gt = ot.hashgrid_fwd_bwd(torch.from_numpy(coords), torch.from_numpy(table), first, res, bw, torch.from_numpy(go))
    np.testing.assert_allclose(gt.numpy(), g64, rtol=1e-4, atol=1e-5 * np.abs(g64).max())


@pytest.mark.parametrize("name", ["A", "D"])
def test_partition_of_unity_and_conservation(name):
    dim, res, bw = CONFIGS[name]
    sizes, first, T = table_layout(res, bw, dim)
    rng = np.random.default_rng(2)
    N = 2000
    coords = rng.uniform(-1, 1, (N, dim)).astype(np.float32)
    _, idx, w = oc.forward(coords, np.zeros((T, 2), np.float32), first, res, bw, want_corners=True)
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-6)                       # sum_k c_k = 1
    assert (idx >= 0).all() and (idx < np.array(sizes)[None, :, None]).all()    # rows inside their level
    const = oc.forward(coords, np.full((T, 2), 0.25, np.float32), first, res, bw)
    np.testing.assert_allclose(const, 0.25, rtol=1e-6)                          # constant table -> constant output
    go = rng.standard_normal((N, len(res) * 2)).astype(np.float32)
    g = oc.backward(coords, go, (T, 2), first, res, bw)
    for l in range(len(res)):                                                   # sum of grads per level/feature
        lo, hi = first[l], first[l] + sizes[l]
        np.testing.assert_allclose(g[lo:hi].sum(0), go[:, 2 * l:2 * l + 2].astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)


def test_out_of_table_corner_is_skipped():
    # dense 2-D level with res >= 258: coord == +1 puts pos at res-1 so corner pos+1 == res; at the table's last
    # row that corner lies past the end of the table (reference UB, weight 0). The oracle must not touch it.
    res, bw = [300], 19
    sizes, first, T = table_layout(res, bw, 2)
    coords = np.array([[1.0, 1.0], [1.0, -1.0]], np.float32)
    table = np.ones((T, 2), np.float32)
    feats, idx, w = oc.forward(coords, table, first, res, bw, want_corners=True)
    assert idx[0, 0, 3] == 300 + 300 * 300 and idx[0, 0, 3] >= T
    assert w[0, 0, 3] == 0 and w[0, 0, 0] == 1
    np.testing.assert_allclose(feats, 1.0)
    g = oc.backward(coords, np.ones((2, 2), np.float32), (T, 2), first, res, bw)
    assert np.isfinite(g).all() and g.sum() == pytest.approx(4.0)


def test_empty_and_single():
    dim, res, bw = CONFIGS["A"]
    _, first, T = table_layout(res, bw, dim)
    table = np.zeros((T, 2), np.float32)
    assert oc.forward(np.zeros((0, 2), np.float32), table, first, res, bw).shape == (0, 16)
    assert oc.backward(np.zeros((0, 2), np.float32), np.zeros((0, 16), np.float32), (T, 2), first, res, bw).sum() == 0
    assert oc.forward(np.zeros((1, 2), np.float32), table, first, res, bw).shape == (1, 16)


def test_double_table_restatement():
    """scalar_t = double (hashgrid_interpolate_cuda.cu:125,290): the forward narrows every table value to float and widens
    the fp32 result (.cu:96-107); the backward's contribution is the float-narrowed product of a DOUBLE gradient and the
    weight (.cu:215-217), so a gradient of 1 + 2^-30 (not a float) gives the same contributions as 1.0, and sums conserve."""
    from oracle import hashgrid_c as oc
    res, bw, dim = [5, 9, 17], 6, 2
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    T = int(sum(sizes))
    rng = np.random.default_rng(3)
    coords = rng.uniform(-1, 1, (257, dim)).astype(np.float32)
    table64 = rng.standard_normal((T, 2)) * (1.0 + 2.0 ** -40)
    f64 = oc.forward_f64(coords, table64, first, res, bw)
    assert f64.dtype == np.float64
    assert np.array_equal(f64, oc.forward(coords, table64.astype(np.float32), first, res, bw).astype(np.float64))
    ones = np.ones((257, len(res) * 2))
    g_a = oc.backward_f64(coords, ones, (T, 2), first, res, bw)
    g_b = oc.backward_f64(coords, ones * (1.0 + 2.0 ** -30), (T, 2), first, res, bw)
    assert np.array_equal(g_a, g_b)                                   # products are narrowed to float
    assert np.array_equal(g_a, oc.backward(coords, ones.astype(np.float32), (T, 2), first, res, bw))
    for l in range(len(res)):                                         # partition of unity: every sample adds 1 per level, feature
        assert abs(g_a[first[l]:first[l] + sizes[l]].sum() - 257 * 2) < 1e-3


# ----------------------------------------------------------------------------------------------- fp16 tables: `__half2` atomics
def test_half_rounding_of_the_model_equals_ieee_binary16():
    """The C model's float -> half -> float rounding against numpy's float16 (RNE, subnormals, overflow to inf)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-7, 1e-5, 1e-3, 1.0, 300.0, 7e4)]
                       + [np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, -65520.0, 6.1035e-5, 6.0976e-5, 5.96e-8, 2.98e-8,
                                    2.9e-8, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, np.inf, -np.inf], dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(oc.half_round(x), want)


def test_half2_atomics_model_known_answers():
    """hashgrid_interpolate_cuda.cu:198-211 by hand. One level of resolution 2 with bw 4 is dense (8 rows = the 8 corners of
    the single cell); a sample at the cell centre has all weights 0.125 exactly. Gradient 1.0 (a half): every product is
    0.125, every entry takes half(0 + 0.125) = 0.125. 2 049 such samples: the running sum climbs in steps of 0.125 until
    the half spacing exceeds twice the step (at 256 the spacing is 0.25: 256 + 0.125 ties back to 256, round-to-even), so
    the reference's table entry STOPS at 256 while the exact sum is 256.125 -- the model must show exactly that."""
    res, bw, F = [2], 4, 2
    first = np.array([0], dtype=np.int32)
    coords = np.full((2049, 3), -0.5, dtype=np.float32)     # 2 * (-0.5 * 0.5 + 0.5) = position 0.5 in each axis: weights 1/8
    go = np.ones((2049, 2), dtype=np.float16)
    go[:, 1] = -0.5
    tab, bound, sumabs, sumg = oc.backward_half_model(coords, go, (8, F), first, res, bw)
    assert np.all(tab[:, 0] == 256.0) and np.all(tab[:, 1] == -128.0), tab     # stuck where spacing / 2 == the step
    exact = np.array([2049 * 0.125, -2049 * 0.0625])
    assert np.allclose(sumabs, np.abs(exact)[None, :]) and np.allclose(sumg, np.array([2049.0, 1024.5])[None, :])
    assert np.all(np.abs(tab - exact[None, :]) <= bound), (tab[0], bound[0])   # the schedule's own bound is rigorous
    # products that need rounding: g = half(0.3), position 0.25 in x (weights 0.75 / 0.25), 0.5 in y and z
    coords = np.array([[-0.75, -0.5, -0.5]], dtype=np.float32)
    go = np.array([[0.3, 0.0]], dtype=np.float16)
    tab, bound, _, _ = oc.backward_half_model(coords, go, (8, F), first, res, bw)
    g = np.float32(go[0, 0])
    # every entry equals half(g * w_k) for the kernel's left-to-right weight products (one add from zero: no sum rounding)
    idx_w = oc.forward(coords, np.zeros((8, F), np.float32), first, res, bw, want_corners=True)
    rows, ws = idx_w[1].reshape(-1), idx_w[2].reshape(-1)
    for r, w in zip(rows, ws):
        assert tab[r, 0] == np.float32(np.float16(g * np.float32(w))), (r, w)


def test_half2_atomics_model_stays_within_its_bound_of_the_exact_sum():
    """On a random problem the model differs from the fp64 sum of the same fp32 products by no more than its own per-entry
    bound, and by far less than the crude n * 2^-11 * sum|product| any schedule obeys."""
    dim, res, bw = CONFIGS["A"]
    sizes, first, T = table_layout(res, bw, dim)
    rng = np.random.default_rng(9)
    coords = rng.uniform(-1, 1, (3000, dim)).astype(np.float32)
    go = rng.standard_normal((3000, len(res) * 2)).astype(np.float16)
    tab, bound, sumabs, _ = oc.backward_half_model(coords, go, (T, 2), first, res, bw)
    exact = oc.backward(coords, go.astype(np.float32), (T, 2), first, res, bw)
    err = np.abs(tab.astype(np.float64) - exact)
    assert np.all(err <= bound.astype(np.float64) * (1 + 1e-6) + 1e-30)
    assert err.max() > 0                                         # the roundings are real
