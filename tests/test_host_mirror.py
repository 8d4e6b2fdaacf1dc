"""Host-side mirror (shacira_amd.wisp.*) against vectors produced by running the reference's own Python modules.
CPU only: the hash-grid operator itself has no CPU path, so where the glue around it is tested the operator is
replaced (in the test only) by the C oracle -- exactly how the goldens were generated."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import npz_json
from oracle import hashgrid_c as oc
from shacira_amd import hip_ops
from shacira_amd.wisp.models.grids import HashGrid, LatentGrid
from shacira_amd.wisp.models.latent_decoders import (DecoderIdentity, HierarchicalLatentDecoder, LatentDecoder,
                                                      MultiLatentDecoder, get_dft_matrix)
from shacira_amd.wisp.models.prob_models import BitEstimator
from shacira_amd.wisp.ops import grid as grid_ops
from shacira_amd.wisp.ops.image.metrics import clamped_psnr, psnr
from shacira_amd.wisp.utils.schedulers import DecayScheduler


def conf(ld, mat="sq", shift=True, enabled=True, ltype="single"):
    cdec = dict(ldecode_enabled=enabled, ldecode_type=ltype, use_sga=False, diff_sampling=False, ldecode_matrix=mat,
                latent_dim=ld, norm="none", norm_every=10, use_shift=shift, num_layers_dec=0, hidden_dim_dec=0,
                activation="none", final_activation="none", clamp_weights=0.0, ldec_std=0.1, num_decoders=1,
                temperature=0.1, decay_period=0.9, alpha_std=1.0)
    cent = dict(num_prob_layers=2, entropy_reg=1e-4, entropy_reg_end=1e-4, entropy_reg_sched="cosine", noise_freq=1)
    return cdec, cent


@pytest.fixture
def oracle_op(monkeypatch):
    """Swap the GPU operator for the C oracle (test-only) so the Python glue can be exercised on CPU."""
    def fwd(dim):
        def f(coords, codebook, first_idx, resolution, bw):
            assert coords.shape[1] == dim
            return torch.from_numpy(oc.forward(coords.detach().numpy(), codebook.detach().numpy(), first_idx.numpy(),
                                               list(resolution), bw))
        return f

    def bwd(dim, coords, grad_output, table_rows, table_dtype, first_idx, resolution, bw, feature_dim):
        g = oc.backward(coords.detach().numpy(), grad_output.detach().numpy(), (table_rows, feature_dim),
                        first_idx.numpy(), list(resolution), bw)
        return torch.from_numpy(g.astype(np.float32))

    monkeypatch.setattr(hip_ops, "hashgrid_interpolate_cuda", fwd(3))
    monkeypatch.setattr(hip_ops, "hashgrid_interpolate2d_cuda", fwd(2))
    monkeypatch.setattr(hip_ops, "hashgrid_backward", bwd)


# ------------------------------------------------------------------------------------------------ construction
@pytest.mark.parametrize("name", ["A", "B", "Bp", "D", "kodak", "lego"])
def test_latent_grid_layout_matches_reference(golden, name):
    meta = npz_json(golden("latent_grid.npz")["meta_json"])[name]
    cdec, cent = conf(meta["kwargs"]["latent_dim"])
    torch.manual_seed(11)
    grid = LatentGrid.from_geometric(multiscale_type="cat", feature_std=0.1, feature_bias=0.0, blas_level=3,
                                     init_grid="uniform", conf_latent_decoder=cdec, conf_entropy_reg=cent,
                                     **meta["kwargs"])
    assert [int(r) for r in grid.resolutions] == meta["resolutions"]
    assert grid.codebook_lod_sizes.tolist() == meta["lod_sizes"]
    assert grid.codebook_lod_first_idx.tolist() == meta["first_idx"]
    assert grid.codebook_lod_sizes.dtype == torch.int32 and grid.codebook_lod_first_idx.dtype == torch.int32
    assert list(grid.codebook.shape) == meta["codebook_shape"]
    assert [n for n, _ in grid.named_parameters()] == meta["param_names"]
    assert list(grid.state_dict().keys()) == meta["state_keys"]
    assert (grid.num_lods, grid.max_lod, grid.active_lods, grid.codebook_size, grid.latent_dim, grid.name()) == \
        (meta["num_lods"], meta["max_lod"], meta["active_lods"], meta["codebook_size"], meta["latent_dim"], meta["name"])


def test_hash_grid_layout_and_init_stream(golden):
    h = golden("hash_grid.json")
    for name in ("A", "D", "img3"):
        torch.manual_seed(3)
        grid = HashGrid.from_geometric(multiscale_type="cat", feature_std=0.01, blas_level=3, **h[name]["kwargs"])
        assert [int(r) for r in grid.resolutions] == h[name]["resolutions"]
        assert grid.codebook_lod_sizes.tolist() == h[name]["lod_sizes"]
        assert grid.codebook_lod_first_idx.tolist() == h[name]["first_idx"]
        assert list(grid.state_dict().keys()) == h[name]["state_keys"]
        assert list(grid.size()) == h[name]["size"] and grid.name() == h[name]["name"]
        assert float(grid.codebook.detach().std()) == h[name]["codebook_std"]   # same RNG draw order as the reference
    torch.manual_seed(3)
    grid = HashGrid.from_octree(feature_dim=2, base_lod=3, num_lods=4, codebook_bitwidth=8, blas_level=3)
    assert [int(r) for r in grid.resolutions] == h["octree"]["resolutions"]
    assert grid.codebook_lod_sizes.tolist() == h["octree"]["lod_sizes"]


def _small_grid(meta, g, name, blas_level=3):
    torch.manual_seed(23)
    cdec, cent = conf(meta["latent_dim"])
    grid = LatentGrid.from_geometric(feature_dim=meta["feature_dim"], num_lods=6, latent_dim=meta["latent_dim"],
                                     multiscale_type=meta["multiscale_type"], resolution_dim=meta["dim"],
                                     feature_std=2.0, codebook_bitwidth=9, min_grid_res=4, max_grid_res=64,
                                     init_grid="uniform", blas_level=blas_level, conf_latent_decoder=cdec,
                                     conf_entropy_reg=cent)
    return grid


@pytest.mark.parametrize("name", ["g2cat", "g2sum", "g2rep", "g3cat", "g3sum"])
def test_small_grid_numerics(golden, oracle_op, name):
    g = golden("latent_grid.npz")
    meta = npz_json(g["meta_json"])[name]
    p = name + "_"
    grid = _small_grid(meta, g, name)
    # identical seed -> identical init stream (table first, then decoder scale), bit for bit
    assert (grid.codebook.detach().numpy() == g[p + "codebook"]).all()
    assert (grid.latent_dec.layers[0].scale.detach().numpy() == g[p + "p_latent_dec.layers.0.scale"]).all()
    sd = {k[len(p) + 2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(p + "p_")}
    sd["codebook"] = torch.from_numpy(g[p + "codebook"])
    missing, unexpected = grid.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"codebook_lod_sizes", "codebook_lod_first_idx"}

    # ent_loss, train mode with injected noise, and validation mode
    grid.noise_freq = 1000
    grid.noise = torch.from_numpy(g[p + "noise"])
    avg, tot = grid.ent_loss(1, is_val=False)
    tot.backward()
    assert tot.item() == pytest.approx(float(g[p + "ent_total"]), rel=1e-5)
    assert avg.item() == pytest.approx(float(g[p + "ent_avg"]), rel=1e-5)
    np.testing.assert_allclose(grid.codebook.grad.numpy(), g[p + "ent_grad_codebook"], rtol=1e-4, atol=1e-6)
    for n, prm in grid.prob_model.named_parameters():
        if prm.grad is not None:
            np.testing.assert_allclose(prm.grad.numpy(), g[p + "ent_g_" + n], rtol=1e-3, atol=1e-3)
    avgv, totv = grid.ent_loss(1, is_val=True)
    assert totv.item() == pytest.approx(float(g[p + "ent_total_val"]), rel=1e-5)

    # size()
    ldec_bits, cb_bits = grid.size(use_torchac=False, use_prob_model=False)
    _, cb_bits_pm = grid.size(use_torchac=False, use_prob_model=True)
    np.testing.assert_allclose([ldec_bits, cb_bits, cb_bits_pm], g[p + "size"], rtol=1e-5)
    # the arithmetic-coded size (this package's range coder, not torchac): a whole number of bytes per channel,
    # never below the entropy and within 5 bytes + 0.2 % per channel of it; the container restores round(latent)
    _, coded_bits = grid.size(use_torchac=True)
    ld = grid.codebook.shape[1]
    assert coded_bits % 8 == 0 and cb_bits - 1e-3 <= coded_bits <= cb_bits * 1.002 + 40 * ld
    blob = grid.compress()
    rounded = torch.round(grid.codebook.detach())
    keep = grid.codebook.detach().clone()
    grid.load_compressed(blob)
    assert torch.equal(grid.codebook.detach(), rounded)
    with torch.no_grad():
        grid.codebook.copy_(keep)

    # interpolate glue (decode -> lookup -> aggregate) forward + backward
    grid.zero_grad()
    coords = torch.from_numpy(g[p + "coords"])
    f = grid.interpolate(coords, 0)
    np.testing.assert_allclose(f.detach().numpy(), g[p + "interp"], rtol=1e-5, atol=1e-6)
    f.backward(torch.from_numpy(g[p + "interp_grad_out"]))
    np.testing.assert_allclose(grid.codebook.grad.numpy(), g[p + "interp_grad_codebook"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(grid.latent_dec.layers[0].scale.grad.numpy(), g[p + "interp_grad_scale"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(grid.latent_dec.layers[0].shift.grad.numpy(), g[p + "interp_grad_shift"], rtol=1e-4, atol=1e-4)
    f3 = grid.interpolate(coords.reshape(8, 8, meta["dim"]), 0)
    assert list(f3.shape) == g[p + "interp_bs_shape"].tolist()
    os.environ["RENDERING_FINAL"] = "1"
    try:
        ff = grid.interpolate(coords, 2)
    finally:
        del os.environ["RENDERING_FINAL"]
    np.testing.assert_allclose(ff.detach().numpy(), g[p + "interp_final_lod2"], rtol=1e-5, atol=1e-6)


def test_identity_decoder_grid(golden):
    meta = npz_json(golden("latent_grid.npz")["meta_json"])["identity"]
    torch.manual_seed(5)
    cdec, cent = conf(2, enabled=False)
    grid = LatentGrid.from_geometric(feature_dim=2, num_lods=4, latent_dim=0, multiscale_type="cat", resolution_dim=2,
                                     feature_std=0.5, codebook_bitwidth=8, min_grid_res=4, max_grid_res=32,
                                     blas_level=3, conf_latent_decoder=cdec, conf_entropy_reg=cent)
    assert isinstance(grid.latent_dec, DecoderIdentity) and grid.prob_model is None
    assert list(grid.ent_loss(0)) == meta["ent_loss"]
    assert [n for n, _ in grid.named_parameters()] == meta["param_names"]
    assert grid.latent_dim == meta["latent_dim"]
    assert [float(v) for v in grid.size()] == meta["size"]


def test_freeze_and_hierarchical():
    cdec, cent = conf(1, ltype="hierarchical")
    grid = LatentGrid.from_geometric(feature_dim=2, num_lods=3, latent_dim=1, multiscale_type="cat", resolution_dim=2,
                                     feature_std=1.0, codebook_bitwidth=6, min_grid_res=4, max_grid_res=16,
                                     blas_level=2, conf_latent_decoder=cdec, conf_entropy_reg=cent)
    assert isinstance(grid.latent_dec, HierarchicalLatentDecoder) and len(grid.latent_dec.decoders) == 3
    # reference quirk kept: last boundary is the last level's size, not the table end (latent_grid.py:182)
    assert grid.latent_dec.offsets.tolist() == grid.codebook_lod_first_idx.tolist() + [int(grid.codebook_lod_sizes[-1])]
    out = grid.latent_dec(grid.codebook)
    assert out.shape == (grid.codebook.shape[0], 2)
    grid.latent_dec.temperature = 0.3
    grid.latent_dec.use_sga = True
    assert all(d.temperature == 0.3 and d.use_sga for d in grid.latent_dec.decoders)
    grid.freeze()
    assert not any(p.requires_grad for p in grid.parameters())
    cm = conf(2, ltype="multi")[0]
    cm["num_decoders"] = 3
    gm = LatentGrid.from_geometric(feature_dim=2, num_lods=3, latent_dim=2, resolution_dim=2, codebook_bitwidth=6,
                                   min_grid_res=4, max_grid_res=16, blas_level=2, multiscale_type="cat",
                                   conf_latent_decoder=cm, conf_entropy_reg=cent)
    assert isinstance(gm.latent_dec, MultiLatentDecoder) and gm.latent_dec.alpha.shape == (3, gm.codebook.shape[0])
    assert "num_entries" not in cm and gm.latent_dec(gm.codebook).shape == (gm.codebook.shape[0], 2)


def test_multi_latent_decoder_module(golden):
    """Row f4: MultiLatentDecoder against vectors produced by the reference's own module (incl. its double-mixing
    quirk in the 'sq' branch), same seed -> same alpha / scale init."""
    g = golden("multi_decoder.npz")
    for ci, case in enumerate(npz_json(g["cases_json"])):
        p = f"m{ci}_"
        torch.manual_seed(500 + ci)
        dec = MultiLatentDecoder(latent_dim=case["latent_dim"], feature_dim=case["feature_dim"], norm="none",
                                 ldecode_matrix=case["ldecode_matrix"], use_shift=case["use_shift"], num_entries=97,
                                 ldec_std=0.1, num_decoders=case["num_decoders"], alpha_std=1.0)
        assert sorted(dec.state_dict().keys()) == case["state_keys"]
        assert (dec.alpha.detach().numpy() == g[p + "p_alpha"]).all()                 # same RNG draw order
        assert (dec.layers[0].scale.detach().numpy() == g[p + "p_layers.0.scale"]).all()
        dec.load_state_dict({k[len(p) + 2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(p + "p_")})
        dec.straight_through = case["straight_through"]
        dec.temperature = 0.7
        lat = torch.from_numpy(g[p + "latent"]).requires_grad_(True)
        y = dec(lat)
        np.testing.assert_allclose(y.detach().numpy(), g[p + "out"], rtol=1e-5, atol=1e-7)
        y.backward(torch.from_numpy(g[p + "grad_out"]))
        np.testing.assert_allclose(lat.grad.numpy(), g[p + "grad_latent"], rtol=1e-5, atol=1e-7)
        for n, prm in dec.named_parameters():
            want = g[p + "g_" + n]
            got = prm.grad.numpy() if prm.grad is not None else np.zeros_like(want)
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6, err_msg=n)
        assert dec.size() == pytest.approx(float(g[p + "size"]), rel=1e-6)


def _multi_mlp_case(g, ci, case, device):
    p = f"h{ci}_"
    torch.manual_seed(700 + ci)
    dec = MultiLatentDecoder(latent_dim=case["latent_dim"], feature_dim=case["feature_dim"], norm="none",
                             ldecode_matrix=case["ldecode_matrix"], use_shift=case["use_shift"], num_entries=131,
                             num_layers_dec=case["num_layers_dec"], hidden_dim_dec=case["hidden_dim_dec"],
                             activation=case["activation"], final_activation=case["final_activation"],
                             clamp_weights=case["clamp_weights"], ldec_std=0.4, num_decoders=case["num_decoders"],
                             alpha_std=1.0, use_sga=case["use_sga"])
    assert sorted(dec.state_dict().keys()) == case["state_keys"]
    assert (dec.alpha.detach().numpy() == g[p + "p_alpha"]).all()                 # same RNG draw order as the reference
    dec.load_state_dict({k[len(p) + 2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(p + "p_")})
    dec.straight_through = case["straight_through"]
    dec.temperature = case["temperature"]
    dec = dec.to(device)
    lat = torch.from_numpy(g[p + "latent"]).to(device).requires_grad_(True)
    if case["use_sga"] and device.type != "cpu":
        # the sampler's noise comes from the device generator on the GPU: draw it on the CPU with the reference's seed instead
        pytest.skip("SGA case: noise parity is a CPU-generator property")
    torch.manual_seed(case["seed"])
    y = dec(lat)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[p + "out"], rtol=2e-5, atol=1e-6)
    y.backward(torch.from_numpy(g[p + "grad_out"]).to(device))
    np.testing.assert_allclose(lat.grad.cpu().numpy(), g[p + "grad_latent"], rtol=1e-4, atol=1e-6)
    for n, prm in dec.named_parameters():
        want = g[p + "g_" + n]
        got = prm.grad.cpu().numpy() if prm.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6, err_msg=n)


def test_multi_latent_decoder_with_hidden_layers(golden):
    """Row f4, the one form that stays a torch-op chain (no shipped configuration selects it): MultiLatentDecoder with hidden
    layers / activations -- every layer mixes the K decoders by the selector (reference multi_latent_decoder.py:27-68, 112-135) --
    against vectors of the executed reference module: straight-through and soft selector, 'sq' and 'dft', clamp, SGA."""
    g = golden("multi_decoder_mlp.npz")
    for ci, case in enumerate(npz_json(g["cases_json"])):
        _multi_mlp_case(g, ci, case, torch.device("cpu"))


def _hier_case(g, ci, case, device):
    p = f"c{ci}_"
    L = len(case["offsets"]) - 1
    conf_d = dict(latent_dim=case["latent_dim"], feature_dim=case["feature_dim"], norm="none",
                  ldecode_matrix=case["ldecode_matrix"], use_shift=case["use_shift"], ldec_std=0.1,
                  clamp_weights=case["clamp_weights"])
    dec = HierarchicalLatentDecoder(L, torch.tensor(case["offsets"], dtype=torch.int32), conf_d)
    with torch.no_grad():
        for l, d in enumerate(dec.decoders):
            d.div.copy_(torch.from_numpy(g[p + f"div{l}"]))
            d.layers[0].scale.copy_(torch.from_numpy(g[p + f"scale{l}"]))
            if case["use_shift"]:
                d.layers[0].shift.copy_(torch.from_numpy(g[p + f"shift{l}"]))
    dec = dec.to(device)
    lat = torch.from_numpy(g[p + "latent"]).to(device).requires_grad_(True)
    owned = torch.from_numpy(g[p + "owned"]).to(device)
    y = dec(lat)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[p + "out"], rtol=1e-5, atol=1e-7)   # unowned rows: zeros
    gy = torch.from_numpy(g[p + "grad_out"]).to(device)
    (y[owned] * gy[owned]).sum().backward()
    np.testing.assert_allclose(lat.grad.cpu().numpy(), g[p + "grad_latent"], rtol=1e-5, atol=1e-7)
    for l, d in enumerate(dec.decoders):
        gs = d.layers[0].scale.grad
        got = gs.cpu().numpy() if gs is not None else np.zeros_like(g[p + f"grad_scale{l}"])
        np.testing.assert_allclose(got, g[p + f"grad_scale{l}"], rtol=1e-4, atol=1e-6, err_msg=f"scale {l}")
        if case["use_shift"]:
            gh = d.layers[0].shift.grad
            got = gh.cpu().numpy() if gh is not None else np.zeros_like(g[p + f"grad_shift{l}"])
            np.testing.assert_allclose(got, g[p + f"grad_shift{l}"], rtol=1e-4, atol=1e-6, err_msg=f"shift {l}")


def test_hierarchical_latent_decoder_module(golden):
    """Row f4: HierarchicalLatentDecoder (host path: per-level decoders) against vectors produced by the reference's own
    module -- row ranges, an empty level, the (sic) last offset of latent_grid.py:182."""
    g = golden("hierarchical_decoder.npz")
    for ci, case in enumerate(npz_json(g["cases_json"])):
        _hier_case(g, ci, case, torch.device("cpu"))


# ------------------------------------------------------------------------------------------------ decoder / CDF
def test_latent_decoder_module(golden):
    g = golden("latent_decoder.npz")
    for (a, b) in [(1, 2), (2, 2), (2, 4), (4, 4)]:
        assert (get_dft_matrix(a, b).numpy() == g[f"dft_{a}_{b}"]).all()
    for ci, case in enumerate(npz_json(g["cases_json"])):
        p = f"c{ci}_"
        dec = LatentDecoder(latent_dim=case["latent_dim"], feature_dim=case["feature_dim"], norm="none",
                            ldecode_matrix=case["ldecode_matrix"], use_shift=case["use_shift"],
                            clamp_weights=case["clamp_weights"], ldec_std=0.1, extra_unused_key=1)
        assert sorted(dec.state_dict().keys()) == case["state_keys"]
        with torch.no_grad():
            dec.div.copy_(torch.from_numpy(g[p + "div"]))
            dec.layers[0].scale.copy_(torch.from_numpy(g[p + "scale"]))
            if case["use_shift"]:
                dec.layers[0].shift.copy_(torch.from_numpy(g[p + "shift"]))
        lat = torch.from_numpy(g[p + "latent"]).requires_grad_(True)
        y = dec(lat)
        np.testing.assert_allclose(y.detach().numpy(), g[p + "out"], rtol=1e-6, atol=1e-7)
        y.backward(torch.from_numpy(g[p + "grad_out"]))
        np.testing.assert_allclose(lat.grad.numpy(), g[p + "grad_latent"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(dec.layers[0].scale.grad.numpy(), g[p + "grad_scale"], rtol=1e-5, atol=1e-5)
        assert dec.size() == sum(p_.numel() * 32 for p_ in dec.parameters())
        assert not dec.div.requires_grad
    dec = LatentDecoder(2, 2, "none", "sq", True, use_sga=True)
    dec.temperature = 0.5
    out = dec(torch.randn(10, 2) * 3)          # SGA path runs (torch ops) and keeps the shape
    assert out.shape == (10, 2) and torch.isfinite(out).all()
    assert dec.get_scale() is dec.layers[0].scale and float(dec.scale_norm()) > 0
    dec.clamp(0.01)
    assert float(dec.layers[0].scale.abs().max()) <= 0.01


def build_mlp_decoder_case(g, ci, case, device="cpu"):
    """LatentDecoder with hidden layers / activations loaded with case ``ci`` of latent_decoder_mlp.npz (vectors of the
    executed reference module). Returns (decoder, its DecoderLayers)."""
    from shacira_amd.wisp.models.latent_decoders.decode_layer import DecoderLayer
    p = f"c{ci}_"
    hid = case["hidden_dim_dec"]
    dec = LatentDecoder(latent_dim=case["latent_dim"], feature_dim=case["feature_dim"], norm="none",
                        ldecode_matrix=case["ldecode_matrix"], use_shift=case["use_shift"],
                        num_layers_dec=case["num_layers_dec"], hidden_dim_dec=tuple(hid) if isinstance(hid, list) else hid,
                        activation=case["activation"], final_activation=case["final_activation"],
                        clamp_weights=case["clamp_weights"], ldec_std=0.4, use_sga=case["use_sga"],
                        diff_sampling=case["diff_sampling"])
    dec.temperature = case["temperature"]
    assert sorted(dec.state_dict().keys()) == case["state_keys"]
    layers = [m for m in dec.layers.children() if isinstance(m, DecoderLayer)]
    assert len(layers) == case["num_layers"]
    with torch.no_grad():
        dec.div.copy_(torch.from_numpy(g[p + "div"]))
        for k, m in enumerate(layers):
            m.scale.copy_(torch.from_numpy(g[p + f"scale{k}"]))
            if case["use_shift"]:
                m.shift.copy_(torch.from_numpy(g[p + f"shift{k}"]))
    return dec.to(device), layers


def test_latent_decoder_with_hidden_layers_module(golden, monkeypatch):
    """The mirror's own evaluation (torch ops: CPU tensors never reach the fused kernel) of decoders with hidden layers and
    activations against the executed reference -- pins the module semantics the fused GPU kernel is then held to."""
    import shacira_amd.wisp.models.latent_decoders.quantizers as quant
    g = golden("latent_decoder_mlp.npz")
    for ci, case in enumerate(npz_json(g["cases_json"])):
        p = f"c{ci}_"
        dec, layers = build_mlp_decoder_case(g, ci, case)
        lat = torch.from_numpy(g[p + "latent"]).requires_grad_(True)
        if case["use_sga"]:      # the sampler draws ONE torch.rand([rows, ld, 2]): feed it the recorded uniforms
            uni = torch.from_numpy(g[p + "uniforms"])
            monkeypatch.setattr(torch, "rand", lambda *a, **k: uni.clone())
        y = dec(lat)
        monkeypatch.undo()
        np.testing.assert_allclose(y.detach().numpy(), g[p + "out"], rtol=2e-5, atol=2e-6, err_msg=f"case {ci}")
        y.backward(torch.from_numpy(g[p + "grad_out"]))
        np.testing.assert_allclose(lat.grad.numpy(), g[p + "grad_latent"], rtol=1e-4, atol=2e-5, err_msg=f"case {ci}")
        for k, m in enumerate(layers):
            ref = g[p + f"grad_scale{k}"]
            np.testing.assert_allclose(m.scale.grad.numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max() + 1e-7,
                                       err_msg=f"case {ci} layer {k}")


@pytest.mark.parametrize("nl", [1, 2, 3, 4])
def test_bit_estimator_module(golden, nl):
    g = golden("bit_estimator.npz")
    assert sorted(BitEstimator(2).state_dict().keys()) == npz_json(g["state_keys_json"])
    for ch in (1, 2):
        p = f"l{nl}_c{ch}_"
        be = BitEstimator(ch, num_layers=nl)
        be.load_state_dict({k[len(p) + 2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(p + "p_")})
        x = torch.from_numpy(g[p + "x"]).requires_grad_(True)
        y = be(x)
        np.testing.assert_allclose(y.detach().numpy(), g[p + "cdf"], rtol=1e-6, atol=1e-7)
        y.backward(torch.from_numpy(g[p + "grad_cdf"]))
        np.testing.assert_allclose(x.grad.numpy(), g[p + "grad_x"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(be(x.detach()[:, ch - 1], single_channel=ch - 1).detach().numpy(), g[p + "single1"],
                                   rtol=1e-6, atol=1e-7)
        assert be.packed_params().shape == (4, 3, ch)


# ------------------------------------------------------------------------------------------------ misc
def test_schedulers_and_metrics(golden):
    for s in golden("schedulers.json"):
        sch = DecayScheduler(s["total"], s["name"], s["start"], s["end"], s["params"])
        np.testing.assert_allclose([float(sch(t)) for t in s["steps"]], s["values"], rtol=1e-12)
    with pytest.raises(ValueError):
        DecayScheduler(10, "nope")(1)
    m = golden("metrics.npz")
    a, b = torch.from_numpy(m["a"]), torch.from_numpy(m["b"])
    assert psnr(b, a) == pytest.approx(float(m["psnr"]), rel=1e-9)
    assert clamped_psnr(b, a) == pytest.approx(float(m["clamped_psnr"]), rel=1e-9)


def test_operator_has_no_cpu_path_and_rejects_odd_feature_dim():
    coords = torch.zeros(4, 2)
    table = torch.zeros(100, 2)
    first = torch.zeros(1, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        grid_ops.hashgrid2d(coords, [8], 8, 0, table, None, first)
    with pytest.raises(Exception, match="multiple of 2"):
        grid_ops.hashgrid2d(coords, [8], 8, 0, torch.zeros(100, 3), None, first)
    with pytest.raises(Exception, match="multiple of 2"):
        grid_ops.hashgrid(torch.zeros(4, 3), [8], 8, 0, torch.zeros(100, 1), None, first)


def test_install_as_wisp_aliases():
    import sys
    import shacira_amd.wisp as w
    saved = {k: v for k, v in sys.modules.items() if k == "wisp" or k.startswith("wisp.")}
    try:
        for k in saved:
            del sys.modules[k]
        w.install_as_wisp()
        import wisp.ops.grid as go
        from wisp.models.grids import LatentGrid as LG
        assert go is grid_ops and LG is LatentGrid
        # the NeRF-side mirrors resolve under the reference's import paths too
        from wisp.accelstructs import OctreeAS
        from wisp.core import Rays
        from wisp.models.embedders import get_positional_embedder
        from wisp.models.nefs.nerf import NeuralRadianceField
        from wisp.tracers.packed_rf_tracer import PackedRFTracer
        assert OctreeAS.make_dense(2).points.shape == (64, 3) and get_positional_embedder(4)[1] == 27
        assert PackedRFTracer().get_required_nef_channels() == {"rgb", "density"} and Rays and NeuralRadianceField
    finally:
        for k in [k for k in sys.modules if k == "wisp" or k.startswith("wisp.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_nerf_field_and_embedder_host_logic():
    """NeuralRadianceField mirror: parameter names (optimiser groups key on them), decoder shapes of nerf.py:121-147,
    embedder layout of positional_embedder.py:60-66 -- host-side checks, no kernels involved."""
    from shacira_amd.wisp.models.embedders import PositionalEmbedder, get_positional_embedder
    from shacira_amd.wisp.models.grids import HashGrid
    from shacira_amd.wisp.models.nefs import NeuralRadianceField
    emb, dim = get_positional_embedder(4)
    assert dim == 3 + 4 * 3 * 2 and isinstance(emb, PositionalEmbedder)
    x = torch.tensor([[0.1, -0.2, 0.3]])
    y = emb(x)
    assert y.shape == (1, 27) and torch.equal(y[:, :3], x)
    bands = torch.tensor([1.0, 2.0, 4.0, 8.0])
    np.testing.assert_allclose(y[0, 3:15].numpy(), torch.sin(x[0][None, :] * bands[:, None]).reshape(-1).numpy(), rtol=1e-6)
    np.testing.assert_allclose(y[0, 15:27].numpy(), torch.cos(x[0][None, :] * bands[:, None]).reshape(-1).numpy(), rtol=1e-6)
    grid = HashGrid.from_geometric(feature_dim=2, num_lods=16, multiscale_type="cat", resolution_dim=3, feature_std=0.01,
                                   codebook_bitwidth=8, min_grid_res=4, max_grid_res=64, blas_level=2)
    nef = NeuralRadianceField(grid, view_embedder="positional", view_multires=4, hidden_dim=64, num_layers=1,
                              prune_density_decay=0.95, prune_min_density=1.0)
    names = [n for n, _ in nef.named_parameters()]
    assert names[0] == "grid.codebook" and any(n.startswith("decoder_density.") for n in names) \
        and any(n.startswith("decoder_color.") for n in names)
    assert nef.density_net_input_dim() == 32 and nef.color_net_input_dim() == 16 + 27
    assert [l.in_features for l in nef.decoder_density.layers] == [32] and nef.decoder_density.lout.out_features == 16
    assert [l.in_features for l in nef.decoder_color.layers] == [43, 64] and nef.decoder_color.lout.out_features == 3
    assert float(nef.decoder_density.lout.bias[0]) == 1.0                       # nerf.py:138
    assert nef.get_supported_channels() == {"density", "rgb"}
    with pytest.raises(NotImplementedError):
        NeuralRadianceField(grid, activation_type="sin")
