"""Pins oracle/latent.py (the numpy restatement) to vectors produced by RUNNING the reference's own modules
(tests/golden/make_golden.py): LatentDecoder fwd/bwd, BitEstimator CDF + grads, LatentGrid.ent_loss, size()."""
import numpy as np
import pytest

from conftest import npz_json
from oracle import latent as ol


def _cases(g):
    return npz_json(g["cases_json"])


def test_dft_matrix_bit_exact(golden):
    g = golden("latent_decoder.npz")
    for (a, b) in [(1, 2), (2, 2), (2, 4), (4, 4)]:
        assert (ol.dft_matrix(a, b) == g[f"dft_{a}_{b}"]).all()


def test_latent_decoder_forward_backward(golden):
    g = golden("latent_decoder.npz")
    for ci, case in enumerate(_cases(g)):
        p = f"c{ci}_"
        dft = "dft" in case["ldecode_matrix"]
        matrix = g[p + "dft"] if dft else g[p + "scale"]
        colscale = g[p + "scale"] if dft else None
        shift = g[p + "shift"] if case["use_shift"] else None
        out, _ = ol.decode_forward(g[p + "latent"], g[p + "div"], matrix, colscale, shift, case["clamp_weights"])
        np.testing.assert_allclose(out, g[p + "out"], rtol=2e-6, atol=1e-7)
        grads = ol.decode_backward(g[p + "latent"], g[p + "div"], matrix, colscale, shift, case["clamp_weights"],
                                   g[p + "grad_out"])
        np.testing.assert_allclose(grads["latent"], g[p + "grad_latent"], rtol=1e-5, atol=1e-7)
        gscale = grads["colscale"].reshape(1, -1) if dft else grads["matrix"]
        np.testing.assert_allclose(gscale, g[p + "grad_scale"], rtol=1e-5, atol=1e-5)
        if case["use_shift"]:
            np.testing.assert_allclose(grads["shift"].reshape(1, -1), g[p + "grad_shift"], rtol=1e-5, atol=1e-5)


def test_rounding_is_half_to_even(golden):
    g = golden("latent_decoder.npz")
    lat = g["c0_latent"]
    assert list(lat[:5, 0]) == [0.5, -0.5, 1.5, 2.5, -2.5]
    out = g["c0_out"]
    scale, shift = g["c0_scale"], g["c0_shift"]
    want = np.array([0.0, -0.0, 2.0, 2.0, -2.0], np.float32)[:, None] @ scale + shift
    np.testing.assert_allclose(out[:5], want, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("nl", [1, 2, 3, 4])
@pytest.mark.parametrize("ch", [1, 2])
def test_bit_estimator_cdf_and_grads(golden, nl, ch):
    g = golden("bit_estimator.npz")
    p = f"l{nl}_c{ch}_"
    params = ol.pack_params({k[len(p) + 2:]: v for k, v in g.items() if k.startswith(p + "p_")}, "", ch)
    s, tr = ol.cdf(g[p + "x"], params, nl)
    np.testing.assert_allclose(s, g[p + "cdf"], rtol=2e-6, atol=1e-7)
    dx, dp = ol._cdf_backward(params, tr, s, g[p + "grad_cdf"].astype(np.float64))
    np.testing.assert_allclose(dx, g[p + "grad_x"], rtol=2e-5, atol=1e-6)
    for k, f in enumerate(("f1", "f2", "f3", "f4")):
        for s_i, slot in enumerate(("h", "b", "a")):
            key = f"{p}g_{f}.{slot}"
            if key in g:
                np.testing.assert_allclose(dp[k, s_i], g[key].reshape(-1), rtol=2e-4, atol=2e-5, err_msg=key)


@pytest.mark.parametrize("name", ["g2cat", "g2sum", "g2rep", "g3cat", "g3sum"])
def test_ent_loss_and_size(golden, name):
    g = golden("latent_grid.npz")
    meta = npz_json(g["meta_json"])[name]
    p = name + "_"
    ld = meta["latent_dim"]
    params = ol.pack_params({k[len(p) + 2:]: v for k, v in g.items() if k.startswith(p + "p_prob_model.")},
                            "prob_model.", ld)
    lat, noise = g[p + "codebook"], g[p + "noise"]
    tot = ol.entropy_bits(lat, noise, params, 2)
    assert tot == pytest.approx(float(g[p + "ent_total"]), rel=2e-6)
    assert tot / lat.shape[0] == pytest.approx(float(g[p + "ent_avg"]), rel=2e-6)
    assert ol.entropy_bits(lat, None, params, 2) == pytest.approx(float(g[p + "ent_total_val"]), rel=2e-6)
    glat, gpar = ol.entropy_bits_backward(lat, noise, params, 2)
    np.testing.assert_allclose(glat, g[p + "ent_grad_codebook"], rtol=2e-4, atol=2e-6)
    for k, f in enumerate(("f1", "f2", "f3", "f4")):
        for s_i, slot in enumerate(("h", "b", "a")):
            key = f"{p}ent_g_{f}.{slot}"
            if key in g:
                np.testing.assert_allclose(gpar[k, s_i], g[key].reshape(-1), rtol=1e-3, atol=1e-3, err_msg=key)
    assert ol.size_bits(lat) == pytest.approx(float(g[p + "size"][1]), rel=1e-5)


def test_sga_decode_restatement_against_reference_run(golden):
    """SGA path of LatentDecoder (use_sga, diff_sampling on/off): the numpy restatement, fed the uniforms the reference's
    sampler drew, reproduces the reference's outputs and gradients (vectors from `make_golden.py sga`)."""
    g = golden("latent_decoder_sga.npz")
    for ci, case in enumerate(npz_json(g["cases_json"])):
        p = f"c{ci}_"
        dft = "dft" in case["ldecode_matrix"]
        matrix = g[p + "dft"] if dft else g[p + "scale"]
        cs = g[p + "scale"] if dft else None
        args = (g[p + "latent"], g[p + "uniforms"], case["temperature"], case["diff_sampling"], g[p + "div"], matrix, cs,
                g[p + "shift"])
        out, _ = ol.decode_sga_forward(*args)
        np.testing.assert_allclose(out, g[p + "out"], rtol=1e-5, atol=1e-6)
        bw = ol.decode_sga_backward(*args, 0.0, g[p + "grad_out"])
        np.testing.assert_allclose(bw["latent"], g[p + "grad_latent"], rtol=2e-4, atol=5e-5 * np.abs(g[p + "grad_latent"]).max())   # fp32 reference, 1/T^2 up to 400
        gscale = bw["colscale"].reshape(1, -1) if dft else bw["matrix"]
        np.testing.assert_allclose(gscale, g[p + "grad_scale"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(bw["shift"].reshape(1, -1), g[p + "grad_shift"], rtol=1e-4, atol=2e-5)
