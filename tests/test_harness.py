"""The minimal image-fit caller (shacira_amd/harness.py): host logic on CPU with the operator swapped for the oracle
(test-only), and -- on the GPU -- PSNR at a fixed step of the HIP path against that CPU restatement."""
import numpy as np
import pytest
import torch

from oracle import hashgrid_c as oc
from shacira_amd import harness, hip_ops


@pytest.fixture
def oracle_op(monkeypatch):
    def fwd(dim):
        def f(coords, codebook, first_idx, resolution, bw):
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


def test_image_and_coords_are_deterministic():
    a, b = harness.make_test_image(32, 48, seed=3), harness.make_test_image(32, 48, seed=3)
    assert a.shape == (32, 48, 3) and a.dtype == np.float32 and np.array_equal(a, b)
    assert 0.0 <= a.min() and a.max() <= 1.0 and a.std() > 0.05
    c = harness.image_coords(4, 6)
    assert c.shape == (24, 2) and float(c.min()) == -1.0 and float(c.max()) < 1.0
    assert torch.allclose(c[7], torch.tensor([(1 / 4 - 0.5) * 2, (1 / 6 - 0.5) * 2]))   # axis 0 = image row


def test_param_groups_follow_reference_names():
    grid, cdec, cent = harness.kodak_like_grid(num_lods=4)
    nef = harness.NeuralImage(grid, hidden_dim=8)
    groups = {g["name"]: g for g in harness.param_groups(nef)}
    names = {id(p): n for n, p in nef.named_parameters()}
    assert sorted(names[id(p)] for p in groups["grid"]["params"]) == ["grid.codebook"]
    assert sorted(names[id(p)] for p in groups["latent_dec"]["params"]) == \
        ["grid.latent_dec.div", "grid.latent_dec.layers.0.scale", "grid.latent_dec.layers.0.shift"]
    assert all(names[id(p)].startswith("grid.prob_model.") for p in groups["prob_models"]["params"])
    assert all(names[id(p)].startswith("decoder_color.") for p in groups["decoder"]["params"])
    assert (groups["grid"]["lr"], groups["latent_dec"]["lr"], groups["prob_models"]["lr"]) == (0.02, 0.01, 1e-4)
    assert groups["latent_dec"]["weight_decay"] == 0.01 and groups["decoder"]["weight_decay"] == 0.0


def test_fit_improves_psnr_on_cpu_restatement(oracle_op):
    r = harness.fit_image(torch.device("cpu"), steps=60, height=48, width=64, seed=1, log_every=20)
    first, last = r["history"][0], r["history"][-1]
    assert last[2] > first[2] + 1.0, r["history"]          # PSNR rises
    assert np.isfinite(r["bpp"]) and r["bpp"] > 0 and r["avg_bits"] > 0


@pytest.mark.gpu
def test_psnr_at_fixed_step_gpu_vs_cpu_restatement(oracle_op, monkeypatch):
    """Same init, same batches, same (CPU-drawn) entropy noise: the HIP path and the CPU restatement of the
    reference kernels must reach the same PSNR at a fixed step (BASELINE.md section 2 / SURVEY 8d: |delta| <= 0.05 dB).
    fp32 summation order differs and rounding (STE) makes a single trajectory chaotic in its last bits (the torch MLP
    GEMMs differ between the devices too), so the LEVEL is compared -- the mean PSNR over the last 20 steps -- and the
    0.05 dB bar is held on its mean over four seeds (a single seed scatters by up to ~0.1 dB either way: 0.2 dB bar)."""
    seeds, deltas = (2, 3, 4, 5), []
    tail = lambda r: float(np.mean([h[2] for h in r["history"][-20:]]))
    first10 = lambda r: np.array([h[2] for h in r["history"][:10]])
    cpu = {sd: harness.fit_image(torch.device("cpu"), steps=200, height=64, width=96, seed=sd, log_every=1) for sd in seeds}
    monkeypatch.undo()                                      # the real HIP operators
    for sd in seeds:
        gpu = harness.fit_image(torch.device("cuda:0"), steps=200, height=64, width=96, seed=sd, log_every=1)
        deltas.append(tail(gpu) - tail(cpu[sd]))
        np.testing.assert_allclose(first10(gpu), first10(cpu[sd]), atol=0.02)   # early steps: still the same trajectory
        assert gpu["psnr"] > 20.0
        assert abs(gpu["bpp"] - cpu[sd]["bpp"]) / cpu[sd]["bpp"] < 0.05
    assert max(abs(d) for d in deltas) <= 0.2, deltas
    assert abs(float(np.mean(deltas))) <= 0.05, deltas


@pytest.mark.gpu
def test_graph_captured_step_reaches_the_same_psnr():
    """The HIP-graph replayed step (device noise, device-side Adam step count) trains to the same level as the eager
    step; the noise stream differs (device vs CPU generator), so compare the level, not the trajectory."""
    import time
    dev = torch.device("cuda:0")
    eager = harness.fit_image(dev, steps=300, height=96, width=128, seed=2, log_every=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    graphed = harness.fit_image(dev, steps=300, height=96, width=128, seed=2, log_every=10, graphed=True)
    torch.cuda.synchronize()
    tail_e = float(np.mean([h[2] for h in eager["history"][-20:]]))
    tail_g = float(np.mean([h[2] for h in graphed["history"][-2:]]))
    assert abs(tail_e - tail_g) <= 0.3, (tail_e, tail_g)
    assert abs(graphed["bpp"] - eager["bpp"]) / eager["bpp"] < 0.05


@pytest.mark.gpu
def test_sga_warm_up_schedule_trains_and_switches_off():
    """kodak.yaml's mode: stochastic Gumbel annealing (fused SGA decode kernels) with the exponential temperature
    schedule until decay_period, rounding afterwards; lands at the same quality level as rounding throughout."""
    dev = torch.device("cuda:0")
    ste = harness.fit_image(dev, steps=400, height=96, width=128, seed=2, log_every=1)
    sga = harness.fit_image(dev, steps=400, height=96, width=128, seed=2, log_every=1, use_sga=True)
    tail = lambda r: float(np.mean([h[2] for h in r["history"][-20:]]))
    assert abs(tail(ste) - tail(sga)) <= 1.5, (tail(ste), tail(sga))
    assert tail(sga) > sga["history"][0][2] + 5.0            # it did train
    assert 0.3 < sga["bpp"] / ste["bpp"] < 3.0
    # the same schedule replayed from a HIP graph (round 4): the temperature is a device float refreshed before each replay,
    # the step is captured again when SGA is switched off
    gsga = harness.fit_image(dev, steps=400, height=96, width=128, seed=2, log_every=20, use_sga=True, graphed=True)
    tail_g = float(np.mean([h[2] for h in gsga["history"][-2:]]))
    assert abs(tail_g - tail(sga)) <= 1.5, (tail_g, tail(sga))
    assert 0.3 < gsga["bpp"] / sga["bpp"] < 3.0


def test_ray_points_and_field():
    g = torch.Generator().manual_seed(0)
    pts = harness.ray_points(64, 16, g)
    assert pts.shape == (1024, 3) and float(pts.abs().max()) <= 1.0
    rgb = harness.analytic_field(pts)
    assert rgb.shape == (1024, 3) and 0.0 <= float(rgb.min()) and float(rgb.max()) <= 1.0 and float(rgb.std()) > 0.05


def test_field_fit_3d_on_cpu_restatement(oracle_op):
    r = harness.fit_field_3d(torch.device("cpu"), steps=40, rays=128, samples_per_ray=8, codebook_bitwidth=10,
                             max_grid_res=64, num_lods=6, val_points=2048)
    assert r["psnr"] > 11.0 and r["samples_per_step"] == 1024


@pytest.mark.gpu
def test_field_fit_3d_gpu_matches_cpu_restatement(oracle_op, monkeypatch):
    kw = dict(steps=120, rays=256, samples_per_ray=16, codebook_bitwidth=12, max_grid_res=128, num_lods=8,
              val_points=8192)
    cpu = harness.fit_field_3d(torch.device("cpu"), **kw)
    monkeypatch.undo()
    gpu = harness.fit_field_3d(torch.device("cuda:0"), **kw)
    assert abs(gpu["psnr"] - cpu["psnr"]) <= 0.1, (gpu["psnr"], cpu["psnr"])
