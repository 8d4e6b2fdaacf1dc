#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's pure-Python modules (dev container only).

Run:  python tests/golden/make_golden.py            (needs /root/reference; writes tests/golden/*.npz|json)

What is pinned by the real reference code (executed, not restated):
  multi_decoder.npz   MultiLatentDecoder.forward/backward/size ('sq'/'dft', soft vs straight-through alpha)   multi_latent_decoder.py:27-210
  latent_decoder.npz  LatentDecoder.forward/backward ('sq'/'dft', shift on/off, div != 1, clamp)   basic_latent_decoder.py:97-198
  latent_decoder_sga.npz  LatentDecoder.forward/backward on the SGA path (use_sga, diff_sampling on/off), with the
                      uniforms the reference's RelaxedOneHotCategorical drew                            basic_latent_decoder.py:183-191
                      (written by `python tests/golden/make_golden.py sga`, leaves the other files untouched)
  latent_decoder_mlp.npz  LatentDecoder with hidden layers / activations (num_layers_dec, activation, final_activation),
                      rounding and SGA paths   basic_latent_decoder.py:139-147,182-198       (python make_golden.py mlp)
  multi_decoder_mlp.npz  MultiLatentDecoder with hidden layers / activations (every layer mixes the K decoders), straight-through
                      and soft selector, clamp, SGA   multi_latent_decoder.py:27-68,112-135   (python make_golden.py multi_mlp)
  hierarchical_decoder.npz  HierarchicalLatentDecoder.forward/backward over row ranges, incl. an empty level and the
                      (sic) last offset of latent_grid.py:182 (`python tests/golden/make_golden.py hier`)   hierarchical_latent_decoder.py:3-36
  bit_estimator.npz   BitEstimator CDF + gradients for num_layers 1..4                               bit_estimator.py:9-65
  latent_grid.npz     LatentGrid.from_geometric tables/buffers/param names, ent_loss (train + val),
                      size(), interpolate() glue ('cat'/'sum', rep trick, [B,S,d] flattening)         latent_grid.py:32-382
  hash_grid.json      HashGrid.from_geometric tables/buffers                                           hash_grid.py:29-180
  schedulers.json     DecayScheduler values                                                             schedulers.py:4-31
  metrics.npz         psnr / clamped_psnr                                                               ops/image/metrics.py:19-58

The reference's CUDA op (wisp._C.ops.hashgrid_interpolate*_cuda) cannot run here; for the interpolate()
glue goldens it is monkey-patched to the C oracle, so those vectors pin the *Python glue around the op*,
not the kernel. Nothing from /root/reference is copied: outputs are plain arrays.
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import hashgrid_c as oc


def _install_shims():
    """Namespace shims so package __init__ files that pull kaolin/CUDA are skipped."""
    for name in ["kaolin", "kaolin.ops", "kaolin.ops.spc", "kaolin.render", "kaolin.render.spc",
                 "kaolin.render.camera", "kaolin.rep", "kaolin.io", "torchac", "torchvision",
                 "torchvision.transforms", "skimage", "skimage.metrics", "cv2", "wandb", "tinyobjloader",
                 "polyscope", "lpips"]:
        sys.modules[name] = MagicMock(name=name)
    import kaolin.ops.spc as spc
    spc.unbatched_get_level_points.return_value = torch.zeros(8, 3)
    sys.modules["kaolin"]._C = MagicMock()

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    ns("wisp", f"{REF}/wisp")
    ns("wisp.models", f"{REF}/wisp/models")
    ns("wisp.models.grids", f"{REF}/wisp/models/grids")
    ns("wisp.utils", f"{REF}/wisp/utils")
    ns("wisp.ops", f"{REF}/wisp/ops")
    ns("wisp.ops.image", f"{REF}/wisp/ops/image")
    ns("wisp.core", f"{REF}/wisp/core")
    wc = MagicMock(name="wisp._C")
    sys.modules["wisp._C"] = wc
    sys.modules["wisp"]._C = wc
    acc = types.ModuleType("wisp.accelstructs")

    class _AS:
        def __init__(self):
            self.points = torch.zeros(1, 3)
            self.pyramid = torch.zeros(2, 2)

        @classmethod
        def make_dense(cls, level):
            return cls()

    acc.OctreeAS = _AS
    acc.BaseAS = _AS
    acc.ASRaymarchResults = object
    acc.ASQueryResults = object
    acc.ASRaytraceResults = object
    sys.modules["wisp.accelstructs"] = acc
    return wc


def _oracle_op(dim):
    def op(coords, codebook, first_idx, resolutions, bw):
        f = oc.forward(coords.detach().numpy(), codebook.detach().numpy(), first_idx.numpy(), list(resolutions), bw)
        return torch.from_numpy(f)
    return op


def _oracle_bwd(coords, grad_output, codebook, first_idx, resolutions, bw, fdim, req):
    g = oc.backward(coords.detach().numpy(), grad_output.detach().numpy(), tuple(codebook.shape),
                    first_idx.numpy(), list(resolutions), bw)
    return torch.from_numpy(g.astype(np.float32))


def main():
    wc = _install_shims()
    wc.ops.hashgrid_interpolate_cuda.side_effect = _oracle_op(3)
    wc.ops.hashgrid_interpolate2d_cuda.side_effect = _oracle_op(2)
    wc.ops.hashgrid_interpolate_backward_cuda.side_effect = _oracle_bwd
    wc.ops.hashgrid_interpolate2d_backward_cuda.side_effect = _oracle_bwd

    import importlib
    core_mod = importlib.import_module("wisp.core.wisp_module")
    sys.modules["wisp.core"].WispModule = core_mod.WispModule
    blas = importlib.import_module("wisp.models.grids.blas_grid")
    sys.modules["wisp.models.grids"].BLASGrid = blas.BLASGrid
    ldec = importlib.import_module("wisp.models.latent_decoders")
    prob = importlib.import_module("wisp.models.prob_models")
    lg = importlib.import_module("wisp.models.grids.latent_grid")
    hg = importlib.import_module("wisp.models.grids.hash_grid")
    sched = importlib.import_module("wisp.utils.schedulers")
    metrics = importlib.import_module("wisp.ops.image.metrics")

    # ------------------------------------------------------------------ (ii) LatentDecoder
    out = {}
    cases = []
    g = torch.Generator().manual_seed(1234)
    for ci, (ld, fd, mat, shift, clampw, divv) in enumerate([
            (1, 2, "sq", True, 0.0, 1.0), (2, 2, "sq", True, 0.0, 3.5), (2, 2, "dft", True, 0.0, 0.7),
            (1, 2, "sq", False, 0.0, 2.0), (2, 4, "dft", False, 0.0, 1.0), (3, 2, "sq", True, 0.05, 1.5),
            (1, 1, "sq", True, 0.0, 1.0)]):
        torch.manual_seed(100 + ci)
        dec = ldec.LatentDecoder(latent_dim=ld, feature_dim=fd, norm="none", ldecode_matrix=mat, use_shift=shift,
                                 clamp_weights=clampw, ldec_std=0.1, extra_unused_key=1)
        with torch.no_grad():
            dec.div.fill_(divv)
            if ld > 1:
                dec.div[1] = divv * 1.25
            if shift:
                dec.layers[0].shift.copy_(torch.randn(1, fd, generator=g) * 0.01)
        lat = ((torch.rand(257, ld, generator=g) - 0.5) * 9.0)
        lat[0] = 0.5; lat[1] = -0.5; lat[2] = 1.5; lat[3] = 2.5; lat[4] = -2.5  # half-to-even ties
        lat.requires_grad_(True)
        y = dec(lat)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        pre = f"c{ci}_"
        out[pre + "latent"] = lat.detach().numpy()
        out[pre + "div"] = dec.div.detach().numpy()
        out[pre + "scale"] = dec.layers[0].scale.detach().numpy()
        out[pre + "shift"] = dec.layers[0].shift.detach().numpy() if shift else np.zeros((1, fd), np.float32)
        if "dft" in mat:
            out[pre + "dft"] = dec.layers[0].dft.detach().numpy()
        out[pre + "out"] = y.detach().numpy()
        out[pre + "grad_out"] = gy.numpy()
        out[pre + "grad_latent"] = lat.grad.numpy()
        out[pre + "grad_scale"] = dec.layers[0].scale.grad.numpy()
        out[pre + "grad_shift"] = dec.layers[0].shift.grad.numpy() if shift else np.zeros((1, fd), np.float32)
        cases.append(dict(latent_dim=ld, feature_dim=fd, ldecode_matrix=mat, use_shift=shift, clamp_weights=clampw,
                          state_keys=sorted(dec.state_dict().keys())))
    # dft matrix values on their own (basic_latent_decoder.py:12-19)
    for (a, b) in [(1, 2), (2, 2), (2, 4), (4, 4)]:
        out[f"dft_{a}_{b}"] = ldec.get_dft_matrix(a, b).numpy()
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "latent_decoder.npz"), **out)

    # ------------------------------------------------------------------ MultiLatentDecoder (row f4)
    mout = {}
    mcases = []
    for ci, (ld, fd, mat, shift, K, st) in enumerate([(2, 2, "sq", True, 3, True), (2, 2, "sq", True, 3, False),
                                                      (2, 4, "dft", True, 2, True), (1, 2, "dft", False, 4, False)]):
        torch.manual_seed(500 + ci)
        dec = ldec.MultiLatentDecoder(latent_dim=ld, feature_dim=fd, norm="none", ldecode_matrix=mat, use_shift=shift,
                                      num_entries=97, ldec_std=0.1, num_decoders=K, alpha_std=1.0)
        dec.straight_through = st
        dec.temperature = 0.7
        with torch.no_grad():
            dec.div.fill_(1.3)
            if shift:
                dec.layers[0].use_shift.copy_(torch.randn(dec.layers[0].use_shift.shape, generator=g) * 0.01)
        lat = ((torch.rand(97, ld, generator=g) - 0.5) * 7.0).requires_grad_(True)
        y = dec(lat)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        pre = f"m{ci}_"
        mout[pre + "latent"] = lat.detach().numpy()
        mout[pre + "out"] = y.detach().numpy()
        mout[pre + "grad_out"] = gy.numpy()
        mout[pre + "grad_latent"] = lat.grad.numpy()
        for n, p_ in dec.named_parameters():
            mout[pre + "p_" + n] = p_.detach().numpy()
            mout[pre + "g_" + n] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy()
        mout[pre + "size"] = np.float64(dec.size())
        mcases.append(dict(latent_dim=ld, feature_dim=fd, ldecode_matrix=mat, use_shift=shift, num_decoders=K,
                           straight_through=st, state_keys=sorted(dec.state_dict().keys())))
    mout["cases_json"] = np.frombuffer(json.dumps(mcases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "multi_decoder.npz"), **mout)

    # ------------------------------------------------------------------ (iii) BitEstimator
    out = {}
    for nl in (1, 2, 3, 4):
        for ch in (1, 2):
            torch.manual_seed(7 * nl + ch)
            be = prob.BitEstimator(ch, num_layers=nl)
            with torch.no_grad():  # move away from the N(0,0.01) init so every term matters
                for p in be.parameters():
                    p.add_(torch.randn(p.shape, generator=g) * 0.5)
            x = ((torch.rand(301, ch, generator=g) - 0.5) * 12.0).requires_grad_(True)
            y = be(x)
            gy = torch.randn(y.shape, generator=g)
            y.backward(gy)
            pre = f"l{nl}_c{ch}_"
            out[pre + "x"] = x.detach().numpy()
            out[pre + "cdf"] = y.detach().numpy()
            out[pre + "grad_cdf"] = gy.numpy()
            out[pre + "grad_x"] = x.grad.numpy()
            for n, p in be.named_parameters():
                out[pre + "p_" + n] = p.detach().numpy()
                out[pre + "g_" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
            out[pre + "single1"] = be(x.detach()[:, ch - 1], single_channel=ch - 1).detach().numpy()
    out["state_keys_json"] = np.frombuffer(json.dumps(sorted(prob.BitEstimator(2).state_dict().keys())).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "bit_estimator.npz"), **out)

    # ------------------------------------------------------------------ (i),(iii),(iv),(v) LatentGrid
    def conf(ld, mat="sq", shift=True, enabled=True, ltype="single"):
        cdec = dict(ldecode_enabled=enabled, ldecode_type=ltype, use_sga=False, diff_sampling=False,
                    ldecode_matrix=mat, latent_dim=ld, norm="none", norm_every=10, use_shift=shift,
                    num_layers_dec=0, hidden_dim_dec=0, activation="none", final_activation="none",
                    clamp_weights=0.0, ldec_std=0.1, num_decoders=1, temperature=0.1, decay_period=0.9,
                    alpha_std=1.0)
        cent = dict(num_prob_layers=2, entropy_reg=1e-4, entropy_reg_end=1e-4, entropy_reg_sched="cosine",
                    noise_freq=1)
        return cdec, cent

    out = {}
    meta = {}
    grid_cfgs = {
        "A": dict(feature_dim=2, num_lods=8, latent_dim=1, codebook_bitwidth=11, min_grid_res=16, max_grid_res=512, resolution_dim=2),
        "B": dict(feature_dim=2, num_lods=16, latent_dim=1, codebook_bitwidth=11, min_grid_res=16, max_grid_res=512, resolution_dim=2),
        "Bp": dict(feature_dim=2, num_lods=16, latent_dim=2, codebook_bitwidth=19, min_grid_res=16, max_grid_res=2048, resolution_dim=2),
        "D": dict(feature_dim=2, num_lods=16, latent_dim=2, codebook_bitwidth=19, min_grid_res=16, max_grid_res=2048, resolution_dim=3),
        "kodak": dict(feature_dim=1, num_lods=24, latent_dim=1, codebook_bitwidth=11, min_grid_res=16, max_grid_res=512, resolution_dim=2),
        "lego": dict(feature_dim=4, num_lods=24, latent_dim=1, codebook_bitwidth=19, min_grid_res=16, max_grid_res=512, resolution_dim=3),
    }
    for name, kw in grid_cfgs.items():
        torch.manual_seed(11)
        cdec, cent = conf(kw["latent_dim"])
        grid = lg.LatentGrid.from_geometric(multiscale_type="cat", feature_std=0.1, feature_bias=0.0, blas_level=7,
                                            init_grid="uniform", conf_latent_decoder=cdec, conf_entropy_reg=cent, **kw)
        meta[name] = dict(kwargs=kw, resolutions=[int(r) for r in grid.resolutions],
                          lod_sizes=grid.codebook_lod_sizes.tolist(), first_idx=grid.codebook_lod_first_idx.tolist(),
                          codebook_shape=list(grid.codebook.shape),
                          param_names=[n for n, _ in grid.named_parameters()],
                          state_keys=list(grid.state_dict().keys()),
                          num_lods=grid.num_lods, max_lod=grid.max_lod, active_lods=grid.active_lods,
                          codebook_size=grid.codebook_size, latent_dim=grid.latent_dim, name=grid.name())

    # small grids with full numerics: ent_loss / size / interpolate
    for name, dim, ld, fd, ms in [("g2cat", 2, 1, 2, "cat"), ("g2sum", 2, 2, 2, "sum"), ("g2rep", 2, 1, 1, "cat"),
                                  ("g3cat", 3, 2, 2, "cat"), ("g3sum", 3, 1, 4, "sum")]:
        torch.manual_seed(23)
        cdec, cent = conf(ld)
        grid = lg.LatentGrid.from_geometric(feature_dim=fd, num_lods=6, latent_dim=ld, multiscale_type=ms,
                                            resolution_dim=dim, feature_std=2.0, codebook_bitwidth=9,
                                            min_grid_res=4, max_grid_res=64, init_grid="uniform",
                                            conf_latent_decoder=cdec, conf_entropy_reg=cent)
        with torch.no_grad():
            grid.latent_dec.div.fill_(1.7)
            for p in grid.prob_model.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.3)
        pre = name + "_"
        out[pre + "codebook"] = grid.codebook.detach().numpy().copy()
        for n, p in grid.named_parameters():
            if n != "codebook":
                out[pre + "p_" + n] = p.detach().numpy().copy()
        meta[name] = dict(dim=dim, latent_dim=ld, feature_dim=fd, multiscale_type=ms, bitwidth=9,
                          resolutions=[int(r) for r in grid.resolutions],
                          first_idx=grid.codebook_lod_first_idx.tolist(), lod_sizes=grid.codebook_lod_sizes.tolist())
        # ent_loss, training mode with injected noise (noise_freq != 1 keeps grid.noise)
        grid.noise_freq = 1000
        noise = torch.rand(grid.codebook.shape, generator=g) - 0.5
        grid.noise = noise
        grid.zero_grad()
        avg, tot = grid.ent_loss(1, is_val=False)
        tot.backward()
        out[pre + "noise"] = noise.numpy()
        out[pre + "ent_total"] = np.float64(tot.item())
        out[pre + "ent_avg"] = np.float64(avg.item())
        out[pre + "ent_grad_codebook"] = grid.codebook.grad.numpy().copy()
        for n, p in grid.prob_model.named_parameters():
            out[pre + "ent_g_" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        avgv, totv = grid.ent_loss(1, is_val=True)
        out[pre + "ent_total_val"] = np.float64(totv.item())
        out[pre + "ent_avg_val"] = np.float64(avgv.item())
        # size()
        ldec_bits, cb_bits = grid.size(use_torchac=False, use_prob_model=False)
        _, cb_bits_pm = grid.size(use_torchac=False, use_prob_model=True)
        out[pre + "size"] = np.array([ldec_bits, cb_bits, cb_bits_pm], np.float64)
        # interpolate glue, [N,d] and [B,S,d]
        grid.zero_grad()
        coords = (torch.rand(64, dim, generator=g) * 2 - 1)
        coords[0] = 1.0
        coords[1] = -1.0
        f = grid.interpolate(coords, 0)
        gy = torch.randn(f.shape, generator=g)
        f.backward(gy)
        out[pre + "coords"] = coords.numpy()
        out[pre + "interp"] = f.detach().numpy()
        out[pre + "interp_grad_out"] = gy.numpy()
        out[pre + "interp_grad_codebook"] = grid.codebook.grad.numpy().copy()
        out[pre + "interp_grad_scale"] = grid.latent_dec.layers[0].scale.grad.numpy().copy()
        out[pre + "interp_grad_shift"] = grid.latent_dec.layers[0].shift.grad.numpy().copy()
        f3 = grid.interpolate(coords.reshape(8, 8, dim), 0)
        out[pre + "interp_bs_shape"] = np.array(f3.shape)
        os.environ["RENDERING_FINAL"] = "1"
        out[pre + "interp_final_lod2"] = grid.interpolate(coords, 2).detach().numpy()
        del os.environ["RENDERING_FINAL"]

    # DecoderIdentity path (ldecode_enabled False): prob_model None, ent_loss -> (0.0, 0.0)
    torch.manual_seed(5)
    cdec, cent = conf(2, enabled=False)
    grid = lg.LatentGrid.from_geometric(feature_dim=2, num_lods=4, latent_dim=0, multiscale_type="cat",
                                        resolution_dim=2, feature_std=0.5, codebook_bitwidth=8, min_grid_res=4,
                                        max_grid_res=32, conf_latent_decoder=cdec, conf_entropy_reg=cent)
    meta["identity"] = dict(ent_loss=list(grid.ent_loss(0)), prob_model_is_none=grid.prob_model is None,
                            param_names=[n for n, _ in grid.named_parameters()], latent_dim=grid.latent_dim,
                            size=[float(v) for v in grid.size()])
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "latent_grid.npz"), **out)

    # ------------------------------------------------------------------ HashGrid
    hmeta = {}
    for name, kw in {"A": dict(feature_dim=2, num_lods=8, codebook_bitwidth=11, min_grid_res=16, max_grid_res=512, resolution_dim=2),
                     "D": dict(feature_dim=2, num_lods=16, codebook_bitwidth=19, min_grid_res=16, max_grid_res=2048, resolution_dim=3),
                     "img3": dict(feature_dim=2, num_lods=16, codebook_bitwidth=11, min_grid_res=16, max_grid_res=512, resolution_dim=3)}.items():
        torch.manual_seed(3)
        grid = hg.HashGrid.from_geometric(multiscale_type="cat", feature_std=0.01, **kw)
        hmeta[name] = dict(kwargs=kw, resolutions=[int(r) for r in grid.resolutions],
                           lod_sizes=grid.codebook_lod_sizes.tolist(), first_idx=grid.codebook_lod_first_idx.tolist(),
                           codebook_shape=list(grid.codebook.shape), state_keys=list(grid.state_dict().keys()),
                           size=list(grid.size()), name=grid.name(), codebook_std=float(grid.codebook.std()))
    torch.manual_seed(3)
    grid = hg.HashGrid.from_octree(feature_dim=2, base_lod=3, num_lods=4, codebook_bitwidth=8)
    hmeta["octree"] = dict(resolutions=[int(r) for r in grid.resolutions], lod_sizes=grid.codebook_lod_sizes.tolist())
    with open(os.path.join(HERE, "hash_grid.json"), "w") as fh:
        json.dump(hmeta, fh, indent=1)

    # ------------------------------------------------------------------ (vi) DecayScheduler
    smeta = []
    for nm, start, end, params in [("fix", 0.3, 0.0, None), ("linear", 1.0, 0.1, None), ("cosine", 1e-4, 1e-2, None),
                                   ("exp", 1.0, 0.05, dict(temperature=0.1, decay_period=0.9)), ("inv_sqrt", 1.0, 0.0, None)]:
        s = sched.DecayScheduler(1000, nm, start, end, params)
        smeta.append(dict(name=nm, start=start, end=end, params=params, total=1000,
                          steps=[0, 1, 10, 500, 999, 1000, 1500], values=[float(s(t)) for t in [0, 1, 10, 500, 999, 1000, 1500]]))
    with open(os.path.join(HERE, "schedulers.json"), "w") as fh:
        json.dump(smeta, fh, indent=1)

    # ------------------------------------------------------------------ (vii) metrics
    a = torch.rand(16, 24, 3, generator=g)
    b = (a + torch.randn(a.shape, generator=g) * 0.05).clamp(-0.04, 1.04)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), a=a.numpy(), b=b.numpy(),
                        psnr=np.float64(metrics.psnr(b, a)), clamped_psnr=np.float64(metrics.clamped_psnr(b, a)))
    print("golden vectors written to", HERE)


def make_sga():
    """SGA path of the reference's LatentDecoder, executed; the uniform numbers its sampler consumed are recovered by
    re-seeding (RelaxedOneHotCategorical.rsample draws ONE torch.rand of shape [rows, ld, 2])."""
    _install_shims()
    import importlib
    core_mod = importlib.import_module("wisp.core.wisp_module")
    sys.modules["wisp.core"].WispModule = core_mod.WispModule
    ldec = importlib.import_module("wisp.models.latent_decoders")
    out, cases = {}, []
    g = torch.Generator().manual_seed(4321)
    for ci, (ld, fd, mat, temp, diff, divv) in enumerate([
            (1, 2, "sq", 0.5, True, 1.0), (2, 2, "sq", 0.2, True, 2.5), (2, 4, "dft", 1.0, True, 0.8),
            (1, 2, "sq", 0.3, False, 1.0), (3, 2, "sq", 0.05, True, 1.5)]):
        torch.manual_seed(200 + ci)
        dec = ldec.LatentDecoder(latent_dim=ld, feature_dim=fd, norm="none", ldecode_matrix=mat, use_shift=True,
                                 ldec_std=0.1, use_sga=True, diff_sampling=diff)
        dec.temperature = temp
        with torch.no_grad():
            dec.div.fill_(divv)
            dec.layers[0].shift.copy_(torch.randn(1, fd, generator=g) * 0.01)
        lat = ((torch.rand(301, ld, generator=g) - 0.5) * 9.0)
        lat[0] = 0.0; lat[1] = 2.0; lat[2] = -3.0; lat[3] = 0.5; lat[4] = -1.999999     # integers, ties, near-integers
        lat.requires_grad_(True)
        seed = 900 + ci
        torch.manual_seed(seed)
        y = dec(lat)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        torch.manual_seed(seed)
        uniforms = torch.rand(301, ld, 2)
        pre = f"c{ci}_"
        out[pre + "latent"] = lat.detach().numpy()
        out[pre + "uniforms"] = uniforms.numpy()
        out[pre + "div"] = dec.div.detach().numpy()
        out[pre + "scale"] = dec.layers[0].scale.detach().numpy()
        out[pre + "shift"] = dec.layers[0].shift.detach().numpy()
        if "dft" in mat:
            out[pre + "dft"] = dec.layers[0].dft.detach().numpy()
        out[pre + "out"] = y.detach().numpy()
        out[pre + "grad_out"] = gy.numpy()
        out[pre + "grad_latent"] = lat.grad.numpy()
        out[pre + "grad_scale"] = dec.layers[0].scale.grad.numpy()
        out[pre + "grad_shift"] = dec.layers[0].shift.grad.numpy()
        cases.append(dict(latent_dim=ld, feature_dim=fd, ldecode_matrix=mat, temperature=temp, diff_sampling=diff))
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "latent_decoder_sga.npz"), **out)
    print("SGA golden vectors written")


def make_hier():
    """HierarchicalLatentDecoder of the reference, executed (rounding path): per-level decoders over row ranges, with
    well-formed offsets and with the (sic) offsets LatentGrid.setup_decoders builds (last boundary = last level's size)."""
    _install_shims()
    import importlib
    core_mod = importlib.import_module("wisp.core.wisp_module")
    sys.modules["wisp.core"].WispModule = core_mod.WispModule
    importlib.import_module("wisp.models.latent_decoders")
    hmod = importlib.import_module("wisp.models.latent_decoders.hierarchical_latent_decoder")
    out, cases = {}, []
    g = torch.Generator().manual_seed(777)
    for ci, (ld, fd, mat, shift, offsets, clampw) in enumerate([
            (1, 2, "sq", True, [0, 40, 100, 257], 0.0),
            (2, 2, "sq", False, [0, 17, 17, 90, 301], 0.0),          # an empty level
            (2, 4, "dft", True, [0, 64, 200, 300], 0.05),
            (1, 2, "sq", True, [0, 50, 150, 120], 0.0)]):            # (sic): last boundary before the last start
        torch.manual_seed(300 + ci)
        L = len(offsets) - 1
        conf = dict(latent_dim=ld, feature_dim=fd, norm="none", ldecode_matrix=mat, use_shift=shift, ldec_std=0.1,
                    clamp_weights=clampw)
        dec = hmod.HierarchicalLatentDecoder(L, torch.tensor(offsets, dtype=torch.int32), conf)
        with torch.no_grad():
            for l, d in enumerate(dec.decoders):
                d.div.copy_(torch.rand(ld, generator=g) * 2.0 + 0.5)
                if shift:
                    d.layers[0].shift.copy_(torch.randn(1, fd, generator=g) * 0.01)
        T = max(offsets) + 9
        lat = ((torch.rand(T, ld, generator=g) - 0.5) * 9.0)
        lat[0] = 0.5; lat[1] = 1.5; lat[2] = -2.5                      # ties round to even
        lat.requires_grad_(True)
        y = dec(lat)
        gy = torch.randn(y.shape, generator=g)
        owned = torch.zeros(T, dtype=torch.bool)
        for l in range(L):
            owned[offsets[l]:offsets[l + 1]] = True
        (y[owned] * gy[owned]).sum().backward()                         # rows no level owns are torch.empty garbage
        pre = f"c{ci}_"
        out[pre + "latent"] = lat.detach().numpy()
        out[pre + "owned"] = owned.numpy()
        out[pre + "out"] = torch.where(owned[:, None], y.detach(), torch.zeros_like(y)).numpy()
        out[pre + "grad_out"] = gy.numpy()
        out[pre + "grad_latent"] = lat.grad.numpy()
        for l, d in enumerate(dec.decoders):
            out[pre + f"div{l}"] = d.div.detach().numpy()
            out[pre + f"scale{l}"] = d.layers[0].scale.detach().numpy()
            gs = d.layers[0].scale.grad
            out[pre + f"grad_scale{l}"] = (gs if gs is not None else torch.zeros_like(d.layers[0].scale)).numpy()
            if shift:
                out[pre + f"shift{l}"] = d.layers[0].shift.detach().numpy()
                gh = d.layers[0].shift.grad
                out[pre + f"grad_shift{l}"] = (gh if gh is not None else torch.zeros_like(d.layers[0].shift)).numpy()
        cases.append(dict(latent_dim=ld, feature_dim=fd, ldecode_matrix=mat, use_shift=shift, offsets=offsets,
                          clamp_weights=clampw, rows=T))
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "hierarchical_decoder.npz"), **out)
    print("hierarchical decoder golden vectors written")


def make_mlp():
    """LatentDecoder of the reference with HIDDEN layers / activations (num_layers_dec > 0, activation, final_activation;
    basic_latent_decoder.py:139-147,182-198), executed: rounding path and SGA path (uniforms recovered by re-seeding)."""
    _install_shims()
    import importlib
    core_mod = importlib.import_module("wisp.core.wisp_module")
    sys.modules["wisp.core"].WispModule = core_mod.WispModule
    ldec = importlib.import_module("wisp.models.latent_decoders")
    out, cases = {}, []
    g = torch.Generator().manual_seed(2468)
    rows = 700                                   # three 256-row tiles, the last one ragged
    for ci, (ld, fd, mat, shift, nl, hid, act, fact, clampw, sga, diff, temp) in enumerate([
            (1, 2, "sq", True, 1, 8, "relu", "none", 0.0, False, False, 1.0),
            (2, 2, "sq", True, 2, (8, 4), "tanh", "none", 0.0, False, False, 1.0),
            (2, 4, "dft", True, 1, 4, "sigmoid", "tanh", 0.0, False, False, 1.0),
            (3, 2, "sq", False, 3, 16, "sine", "none", 0.0, False, False, 1.0),
            (4, 4, "sq", True, 1, 0, "relu", "sigmoid", 0.6, False, False, 1.0),       # hidden_dim_dec 0 -> feature_dim
            (2, 2, "sq", True, 0, 0, "none", "tanh", 0.0, False, False, 1.0),          # no hidden layer, final activation only
            (2, 2, "sq", True, 2, 8, "relu", "none", 0.0, True, True, 0.4),
            (1, 2, "dft", True, 1, 8, "tanh", "none", 0.0, True, False, 0.7)]):
        torch.manual_seed(300 + ci)
        dec = ldec.LatentDecoder(latent_dim=ld, feature_dim=fd, norm="none", ldecode_matrix=mat, use_shift=shift,
                                 num_layers_dec=nl, hidden_dim_dec=hid, activation=act, final_activation=fact,
                                 clamp_weights=clampw, ldec_std=0.4, use_sga=sga, diff_sampling=diff)
        dec.temperature = temp
        layers = [m for m in dec.layers.children() if isinstance(m, ldec.DecoderLayer)]
        with torch.no_grad():
            dec.div.fill_(1.7)
            if ld > 1:
                dec.div[1] = 0.9
            if act == "sine":                    # keep 30 * z in a range where fp32 sin() is well conditioned
                for m in layers:
                    m.scale.mul_(0.1)
            if shift:
                for m in layers:
                    m.shift.copy_(torch.randn(m.shift.shape, generator=g) * 0.05)
        lat = ((torch.rand(rows, ld, generator=g) - 0.5) * 7.0)
        lat[0] = 0.5; lat[1] = -0.5; lat[2] = 1.5; lat[3] = 2.0
        lat.requires_grad_(True)
        seed = 950 + ci
        torch.manual_seed(seed)
        y = dec(lat)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        pre = f"c{ci}_"
        if sga:
            torch.manual_seed(seed)
            out[pre + "uniforms"] = torch.rand(rows, ld, 2).numpy()
        out[pre + "latent"] = lat.detach().numpy()
        out[pre + "div"] = dec.div.detach().numpy()
        for k, m in enumerate(layers):
            out[pre + f"scale{k}"] = m.scale.detach().numpy()
            out[pre + f"grad_scale{k}"] = m.scale.grad.numpy()
            if shift:
                out[pre + f"shift{k}"] = m.shift.detach().numpy()
                out[pre + f"grad_shift{k}"] = m.shift.grad.numpy()
        out[pre + "out"] = y.detach().numpy()
        out[pre + "grad_out"] = gy.numpy()
        out[pre + "grad_latent"] = lat.grad.numpy()
        cases.append(dict(latent_dim=ld, feature_dim=fd, ldecode_matrix=mat, use_shift=shift, num_layers_dec=nl,
                          hidden_dim_dec=hid, activation=act, final_activation=fact, clamp_weights=clampw, use_sga=sga,
                          diff_sampling=diff, temperature=temp, num_layers=len(layers),
                          state_keys=sorted(dec.state_dict().keys())))
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "latent_decoder_mlp.npz"), **out)
    print("hidden-layer decoder golden vectors written")


def make_multi_mlp():
    """MultiLatentDecoder of the reference WITH hidden layers / activations (num_layers_dec > 0: every layer mixes the K decoders
    by the selector, multi_latent_decoder.py:112-135, :27-68 for the layer), executed: straight-through and soft selector, 'sq' and
    'dft', clamp, SGA. No shipped configuration selects this form; this library evaluates it with torch ops like the reference, and
    these vectors pin that path (`python tests/golden/make_golden.py multi_mlp`)."""
    _install_shims()
    import importlib
    core_mod = importlib.import_module("wisp.core.wisp_module")
    sys.modules["wisp.core"].WispModule = core_mod.WispModule
    ldec = importlib.import_module("wisp.models.latent_decoders")
    out, cases = {}, []
    g = torch.Generator().manual_seed(1357)
    rows = 131
    for ci, (ld, fd, mat, shift, K, st, nl, hid, act, fact, clampw, sga, temp) in enumerate([
            # ('sq' layers of the reference only run square: `out[i] = matmul(input, scale[i])` is assigned into a buffer shaped
            # like the INPUT, multi_latent_decoder.py:71-73 -- so latent_dim == hidden == feature_dim there; 'dft' takes any widths)
            (2, 2, "sq", True, 3, True, 1, 2, "relu", "none", 0.0, False, 0.7),
            (2, 4, "dft", True, 2, False, 2, (4, 4), "tanh", "none", 0.0, False, 0.9),
            (2, 2, "sq", False, 4, True, 1, 0, "sigmoid", "tanh", 0.0, False, 1.0),
            (4, 4, "sq", True, 2, False, 2, 4, "relu", "none", 0.3, False, 0.5),
            (2, 2, "sq", True, 3, True, 1, 2, "tanh", "none", 0.0, True, 0.6)]):
        torch.manual_seed(700 + ci)
        dec = ldec.MultiLatentDecoder(latent_dim=ld, feature_dim=fd, norm="none", ldecode_matrix=mat, use_shift=shift,
                                      num_entries=rows, num_layers_dec=nl, hidden_dim_dec=hid, activation=act,
                                      final_activation=fact, clamp_weights=clampw, ldec_std=0.4, num_decoders=K,
                                      alpha_std=1.0, use_sga=sga)
        dec.straight_through = st
        dec.temperature = temp
        with torch.no_grad():
            dec.div.fill_(1.3)
            for n_, p_ in dec.named_parameters():
                if n_.endswith("use_shift"):
                    p_.copy_(torch.randn(p_.shape, generator=g) * 0.05)
        lat = ((torch.rand(rows, ld, generator=g) - 0.5) * 7.0)
        lat[0] = 0.5; lat[1] = -0.5; lat[2] = 1.5
        lat.requires_grad_(True)
        seed = 1900 + ci
        torch.manual_seed(seed)
        y = dec(lat)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        pre = f"h{ci}_"
        out[pre + "latent"] = lat.detach().numpy()
        out[pre + "out"] = y.detach().numpy()
        out[pre + "grad_out"] = gy.numpy()
        out[pre + "grad_latent"] = lat.grad.numpy()
        for n_, p_ in dec.named_parameters():
            out[pre + "p_" + n_] = p_.detach().numpy()
            out[pre + "g_" + n_] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy()
        cases.append(dict(latent_dim=ld, feature_dim=fd, ldecode_matrix=mat, use_shift=shift, num_decoders=K,
                          straight_through=st, num_layers_dec=nl, hidden_dim_dec=hid, activation=act, final_activation=fact,
                          clamp_weights=clampw, use_sga=sga, temperature=temp, seed=seed,
                          state_keys=sorted(dec.state_dict().keys())))
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "multi_decoder_mlp.npz"), **out)
    print("multi decoder (hidden layers) golden vectors written")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; goldens can only be regenerated in the dev container")
    if len(sys.argv) > 1 and sys.argv[1] == "sga":
        make_sga()
    elif len(sys.argv) > 1 and sys.argv[1] == "hier":
        make_hier()
    elif len(sys.argv) > 1 and sys.argv[1] == "mlp":
        make_mlp()
    elif len(sys.argv) > 1 and sys.argv[1] == "multi_mlp":
        make_multi_mlp()
    else:
        main()
