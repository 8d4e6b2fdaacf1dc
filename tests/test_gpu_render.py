"""Volume integration and sample generation kernels (render.hip) against the CPU restatement in oracle/render.py,
through the C-ABI. fp32: tolerance 1e-5 relative (north_star); ids, counts and boundaries exact."""
import numpy as np
import pytest
import torch

from oracle import render as orr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from shacira_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _packs(rng, R, max_len, allow_long=True):
    lens = rng.integers(1, max_len, R)
    if allow_long and R > 3:
        lens[1] = 1
        lens[2] = 64
        lens[3] = 257
    boundary = np.zeros(lens.sum(), dtype=bool)
    boundary[np.concatenate([[0], np.cumsum(lens)[:-1]])] = True
    return torch.from_numpy(boundary)


@pytest.mark.parametrize("C", [1, 3, 4, 7])
@pytest.mark.parametrize("R,max_len", [(1, 5), (37, 40), (300, 200)])
def test_exponential_integration_forward_backward(dev, C, R, max_len):
    from shacira_amd import render
    rng = np.random.default_rng(C * 100 + R)
    boundary = _packs(rng, R, max_len)
    S = boundary.shape[0]
    feats = torch.from_numpy(rng.random((S, C)).astype(np.float32))
    tau = torch.from_numpy((rng.random((S, 1)) ** 3 * 2.0).astype(np.float32))
    g_ray = torch.from_numpy(rng.standard_normal((R, C)).astype(np.float32))
    g_w = torch.from_numpy(rng.standard_normal((S, 1)).astype(np.float32))
    # oracle (fp64 autograd of the published formula)
    f64, t64 = feats.double().requires_grad_(), tau.double().requires_grad_()
    ray_o, w_o = orr.exponential_integration(f64, t64, boundary)
    (ray_o * g_ray.double()).sum().add((w_o * g_w.double()).sum()).backward()
    # HIP
    fd, td = feats.to(dev).requires_grad_(), tau.to(dev).requires_grad_()
    ray, w = render.exponential_integration(fd, td, boundary.to(dev))
    assert tuple(ray.shape) == (R, C) and tuple(w.shape) == (S, 1)
    np.testing.assert_allclose(ray.detach().cpu().numpy(), ray_o.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(w.detach().cpu().numpy(), w_o.detach().numpy(), rtol=1e-5, atol=1e-7)
    ((ray * g_ray.to(dev)).sum() + (w * g_w.to(dev)).sum()).backward()
    np.testing.assert_allclose(fd.grad.cpu().numpy(), f64.grad.numpy(), rtol=1e-5, atol=1e-6)
    scale = float(t64.grad.abs().max())
    np.testing.assert_allclose(td.grad.cpu().numpy(), t64.grad.numpy(), rtol=1e-4, atol=2e-6 * scale)
    # sum_reduce + its gradient, and the alpha / depth uses of the tracer
    xs = torch.from_numpy(rng.standard_normal((S, C)).astype(np.float32))
    xd = xs.to(dev).requires_grad_()
    out = render.sum_reduce(xd, boundary.to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), orr.sum_reduce(xs, boundary).numpy(), rtol=1e-5, atol=1e-5)
    (out * g_ray.to(dev)).sum().backward()
    pid = torch.cumsum(boundary.long(), 0) - 1
    assert torch.equal(xd.grad.cpu(), g_ray[pid])
    with pytest.raises(RuntimeError):
        render.exponential_integration(feats, tau, boundary)   # host tensors: no CPU fallback


def test_parity_unpinned_dependency_absent__hip_integration_of_a_homogeneous_medium(dev):
    """The HIP kernel against the CLOSED FORM, not against the restatement: constant density sigma and colour c over n equal
    steps -> weights exp(-k sigma d) (1 - exp(-sigma d)), ray colour c (1 - exp(-sigma n d)), and d(ray colour)/d(sigma) = c n d
    exp(-sigma n d) through the kernel's backward (every sample's tau = sigma d, so the gradients of the n samples add up)."""
    from shacira_amd import render
    for n, sigma, d in ((16, 2.0, 0.05), (200, 0.3, 0.01), (1, 0.7, 0.3)):
        colour = torch.tensor([0.2, 0.5, 0.9], device=dev)
        sig = torch.tensor(sigma, device=dev, requires_grad=True)
        tau = (sig * d).expand(2 * n, 1)
        feats = colour[None].repeat(2 * n, 1)
        boundary = torch.zeros(2 * n, dtype=torch.bool, device=dev)
        boundary[0] = boundary[n] = True                       # two identical rays
        ray, w = render.exponential_integration(feats, tau, boundary)
        k = np.arange(n)
        np.testing.assert_allclose(w[:n, 0].detach().cpu().numpy(), np.exp(-k * sigma * d) * (1 - np.exp(-sigma * d)), rtol=2e-5)
        want = colour.cpu().numpy() * (1 - np.exp(-sigma * n * d))
        np.testing.assert_allclose(ray[0].detach().cpu().numpy(), want, rtol=2e-5)
        np.testing.assert_allclose(ray[1].detach().cpu().numpy(), want, rtol=2e-5)
        ray[0].sum().backward()
        np.testing.assert_allclose(sig.grad.item(), float(colour.sum()) * n * d * np.exp(-sigma * n * d), rtol=1e-4)


def _rays(rng, n):
    o = rng.standard_normal((n, 3))
    o = 3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = (rng.random((n, 3)) - 0.5) - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    return torch.from_numpy(o.astype(np.float32)), torch.from_numpy(d.astype(np.float32))


@pytest.mark.parametrize("level,ns", [(3, 16), (5, 64), (7, 200)])
def test_raymarch_ray_matches_reference_python(dev, level, ns):
    from shacira_amd import render
    rng = np.random.default_rng(level)
    N, G = 500, 1 << level
    o, d = _rays(rng, N)
    occ = torch.from_numpy(rng.random((G, G, G)) < 0.3)
    jit = torch.from_numpy(rng.random((N, ns)).astype(np.float32))
    near, far = 1.5, 4.5
    full = torch.ones((G, G, G), dtype=torch.bool)
    # all cells occupied: every sample INSIDE the cube is emitted (a point outside has no cell, as in kaolin's query)
    r_all, s_all, dep_all, del_all, b_all, off_all = [t.cpu() for t in render.raymarch_ray(
        o.to(dev), d.to(dev), near, far, full.to(dev), level, ns, jit.to(dev))]
    ro, so, do, dlo, bo = orr.raymarch_ray(o, d, near, far, full, level, ns, jit)
    assert 0 < r_all.shape[0] < N * ns and torch.equal(r_all, ro)
    assert torch.equal(off_all[1:] - off_all[:-1], torch.bincount(r_all, minlength=N)) and off_all[0] == 0
    assert bool((s_all.abs() <= 1.0).all())
    np.testing.assert_allclose(dep_all.numpy(), do.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(del_all.numpy(), dlo.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(s_all.numpy(), so.numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(b_all, bo)
    # real occupancy: exactly the reference's filter applied to the very same positions (deltas are differences of the
    # sample LATTICE, so they do not change when neighbours are dropped)
    r, s, dep, dl, b, off = [t.cpu() for t in render.raymarch_ray(o.to(dev), d.to(dev), near, far, occ.to(dev), level,
                                                                  ns, jit.to(dev))]
    assert torch.equal(off[1:] - off[:-1], torch.bincount(r, minlength=N))        # per-ray pack sizes, empty ones too
    keep = orr.query_dense(occ, s_all, level)
    assert torch.equal(r, r_all[keep]) and torch.equal(b, orr.mark_pack_boundaries(r))
    assert torch.equal(s, s_all[keep]) and torch.equal(dep, dep_all[keep]) and torch.equal(dl, del_all[keep])
    # nothing occupied -> nothing emitted
    none = render.raymarch_ray(o.to(dev), d.to(dev), near, far, torch.zeros_like(full).to(dev), level, ns, jit.to(dev))
    assert none[0].numel() == 0 and none[4].numel() == 0


@pytest.mark.parametrize("level", [2, 4, 6])
def test_raytrace_dense_matches_brute_force(dev, level):
    from shacira_amd import render
    rng = np.random.default_rng(10 + level)
    N, G = 200, 1 << level
    o, d = _rays(rng, N)
    o[0] = torch.tensor([0.05, 0.02, -0.03])           # a ray that starts inside the volume
    o[1], d[1] = torch.tensor([0.3, 0.3, -3.0]), torch.tensor([0.0, 0.0, 1.0])   # axis aligned
    o[2], d[2] = torch.tensor([5.0, 5.0, 5.0]), torch.tensor([0.0, 1.0, 0.0])    # misses
    occ = torch.from_numpy(rng.random((G, G, G)) < 0.4)
    q0 = orr.quantize_points(o[0:1], level)[0]
    occ[q0[0], q0[1], q0[2]] = True                   # ... inside an occupied cell: entry depth clipped to 0
    ridx, pidx, depth = [t.cpu() for t in render.raytrace_dense(o.to(dev), d.to(dev), occ.to(dev), level)]
    ro, co, do = orr.raytrace_dense(o, d, occ, level)
    # drop grazing contacts (shorter than 1e-4) on both sides, then compare run by run
    keep = (depth[:, 1] - depth[:, 0]) > 1e-4
    keep_o = (do[:, 1] - do[:, 0]) > 1e-4
    ridx, pidx, depth = ridx[keep], pidx[keep], depth[keep]
    ro, co, do = ro[keep_o], co[keep_o], do[keep_o]
    assert torch.equal(ridx.long(), ro)
    morton = torch.zeros(co.shape[0], dtype=torch.long)
    for b in range(level):
        morton |= (((co[:, 0] >> b) & 1) << (3 * b + 2)) | (((co[:, 1] >> b) & 1) << (3 * b + 1)) \
            | (((co[:, 2] >> b) & 1) << (3 * b))
    assert torch.equal(pidx.long(), morton)
    np.testing.assert_allclose(depth.numpy(), do.numpy(), rtol=1e-4, atol=2e-5)
    assert not (ridx == 2).any() and (ridx == 0).any() and float(depth[ridx == 0][0, 0]) == 0.0


def test_tracer_composites_like_the_reference_formula(dev):
    """PackedRFTracer ('ray' and 'voxel' sampling) on a closed-form field: rgb / alpha / depth / hit equal the
    reference's composition (packed_rf_tracer.py:131-151) evaluated by the CPU restatement on the tracer's own samples."""
    from shacira_amd import harness
    from shacira_amd.wisp.core import Rays
    from shacira_amd.wisp.models.grids import HashGrid
    from shacira_amd.wisp.tracers import PackedRFTracer
    grid = HashGrid.from_geometric(feature_dim=2, num_lods=2, multiscale_type="cat", resolution_dim=3, feature_std=0.0,
                                   codebook_bitwidth=4, min_grid_res=2, max_grid_res=4, blas_level=4)
    rng = np.random.default_rng(3)
    occ = torch.from_numpy(rng.random((16, 16, 16)) < 0.6)
    grid.blas = grid.blas.__class__.from_quantized_points(torch.nonzero(occ), 4)
    nef = harness._AnalyticNef(grid)
    o, d = harness.camera_rays(300, torch.Generator().manual_seed(1), dev)
    rays = Rays(o, d, dist_min=1.0, dist_max=5.0)
    for kind, ns in (("ray", 96), ("voxel", 3)):
        torch.manual_seed(7)
        marched = grid.raymarch(rays, raymarch_type=kind, num_samples=ns, level=grid.active_lods[-1])
        torch.manual_seed(7)                                      # same jitter inside the tracer
        rb = PackedRFTracer(raymarch_type=kind, num_steps=ns, bg_color="white")(nef, rays, channels=("rgb", "depth"))
        ridx, boundary = marched.ridx.cpu(), marched.boundary.cpu()
        assert orr.query_dense(occ, marched.samples.cpu(), 4).all()          # every sample sits in an occupied cell
        density, color = harness.analytic_scene(marched.samples.cpu())
        tau = density * marched.deltas.cpu()
        ray_colors, w = orr.exponential_integration(color.double(), tau.double(), boundary)
        alpha = orr.sum_reduce(w, boundary)
        depth = orr.sum_reduce(marched.depth_samples.cpu().double() * w, boundary)
        hit_rays = ridx[boundary]
        want_rgb = torch.ones(300, 3, dtype=torch.float64)
        want_rgb[hit_rays] = (1.0 - alpha) + ray_colors
        np.testing.assert_allclose(rb.rgb.cpu().numpy(), want_rgb.numpy(), rtol=1e-5, atol=2e-6)
        want_alpha = torch.zeros(300, 1, dtype=torch.float64)
        want_alpha[hit_rays] = alpha
        np.testing.assert_allclose(rb.alpha.cpu().numpy(), want_alpha.numpy(), rtol=1e-5, atol=2e-6)
        want_depth = torch.zeros(300, 1, dtype=torch.float64)
        want_depth[hit_rays] = depth
        np.testing.assert_allclose(rb.depth.cpu().numpy(), want_depth.numpy(), rtol=1e-5, atol=1e-5)
        sure = (want_alpha[:, 0] > 1e-6) | (want_alpha[:, 0] == 0)    # fp32 rounds 1 - exp(-tau) to 0 below ~6e-8
        assert torch.equal(rb.hit.cpu()[sure], (want_alpha[:, 0] > 0)[sure])


def test_nerf_fit_learns_the_scene(dev):
    """End to end: marcher -> hash grid -> decoders -> volume integration -> L1 -> fused Adam, occupancy pruning on."""
    from shacira_amd import harness
    r = harness.fit_nerf(dev, steps=350, rays=2048, num_steps=96, codebook_bitwidth=16, max_grid_res=512,
                         prune_every=100, val_rays=4096)
    assert r["psnr"] > 17.5, r                                 # an all-background render scores ~11 dB
    assert 0 < r["occupied_cells"] < r["total_cells"], r      # pruning removed the empty space, kept the blobs


def test_compressed_nerf_fit_trains_and_codes(dev):
    """nerf_lego.yaml's mode end to end: 3-D LatentGrid (fused SGA decode, entropy model, device noise), marcher,
    MFMA decoders, volume integration; the model file is a small fraction of the fp32 table and the fit learns."""
    from shacira_amd import harness
    r = harness.fit_nerf(dev, steps=300, rays=2048, num_steps=96, codebook_bitwidth=16, max_grid_res=512,
                         prune_every=100, val_rays=4096, latent=True)
    assert r["psnr"] > 15.0, r
    assert r["file_bytes"] < 0.25 * r["table_bytes_fp32"], r
    assert 0.5 < r["file_bytes"] / (r["latent_bytes_estimate"] + 60_000) < 2.0, r   # + decoders stored raw


def test_capped_raymarch_emit(dev):
    """shacira_raymarch_ray_emit_capped (ABI 9, for steps captured into a HIP graph): fixed-size outputs without a count
    read-back. Capacity above the count: the first S rows equal the uncapped emit, the rows behind are padding (ray 0, delta
    0, positions inside the cube) and no pack covers them. Capacity below the count: the first `capacity` rows equal the
    uncapped emit's, offsets are clamped, the true count comes back on the device."""
    from shacira_amd import render
    rng = np.random.default_rng(11)
    N, level, ns = 700, 5, 64
    G = 1 << level
    o, d = _rays(rng, N)
    occ = torch.from_numpy(rng.random((G, G, G)) < 0.3).to(dev)
    jit = torch.from_numpy(rng.random((N, ns)).astype(np.float32)).to(dev)
    args = (o.to(dev), d.to(dev), 1.5, 4.5, occ, level, ns, jit)
    r, s, dep, dl, b, off = render.raymarch_ray(*args)
    S = r.shape[0]
    for cap in (S + 1000, S, S - 777, 64):
        rc, sc, depc, dlc, bc, offc, total = render.raymarch_ray(*args, capacity=cap)
        assert int(total) == S and rc.shape[0] == cap and sc.shape == (cap, 3)
        k = min(S, cap)
        assert torch.equal(rc[:k], r[:k]) and torch.equal(sc[:k], s[:k]) and torch.equal(depc[:k], dep[:k])
        assert torch.equal(dlc[:k], dl[:k]) and torch.equal(bc[:k], b[:k])
        assert torch.equal(offc, off.clamp(max=cap)) and int(offc[-1]) == k
        if cap > S:
            assert not rc[S:].any() and not dlc[S:].any() and not bc[S:].any() and bool((sc[S:].abs() <= 1).all())
            assert sc[S:].unique(dim=0).shape[0] > (cap - S) // 2          # spread, not one point


def test_graphed_nerf_fit_matches_the_eager_fit(dev):
    """The NeRF step replayed from HIP graphs (GraphedNerfFitter: capacity-sized sample buffers, occupancy updated in
    place, device-side rays and Adam step count) learns the scene like the eager loop: same PSNR level at the same step
    (same ray pool; the paths differ by the padding rows' zero gradients and the add order), no step dropped samples, a re-capture per prune that moved the sample count."""
    from shacira_amd import harness
    kw = dict(steps=350, rays=2048, num_steps=96, codebook_bitwidth=16, max_grid_res=512, prune_every=100, val_rays=4096)
    eager = harness.fit_nerf(dev, ray_pool=32, **kw)             # the same 32 batches of rays + targets, in the same order
    graphed = harness.fit_nerf(dev, graphed=True, ray_pool=32, **kw)
    assert graphed["overflow_steps"] == 0, graphed
    assert 1 <= graphed["graph_captures"] <= 4, graphed
    assert graphed["psnr"] > 17.5 and abs(graphed["psnr"] - eager["psnr"]) < 1.5, (eager, graphed)
    assert 0 < graphed["occupied_cells"] < graphed["total_cells"], graphed
    assert graphed["sample_capacity"] < kw["rays"] * kw["num_steps"] // 2, graphed     # sized to the pruned scene, not the maximum


def test_graphed_compressed_nerf_fit(dev):
    """The reference's nerf_lego.yaml mode (3-D LatentGrid, SGA warm-up, entropy model) replayed from HIP graphs: the SGA
    temperature is annealed through ONE device float the decode kernels read (shacira_latent_decode_sga_*_tdev), the step is
    re-captured once when SGA is switched off; it learns like the eager loop and codes to a small fraction of the fp32 table."""
    from shacira_amd import harness
    kw = dict(steps=300, rays=2048, num_steps=96, codebook_bitwidth=16, max_grid_res=512, prune_every=100, val_rays=4096,
              latent=True, ray_pool=32)
    eager = harness.fit_nerf(dev, **kw)
    graphed = harness.fit_nerf(dev, graphed=True, **kw)
    assert graphed["overflow_steps"] == 0 and graphed["graph_captures"] >= 2, graphed       # >= one capture per SGA mode
    assert graphed["psnr"] > 15.0 and abs(graphed["psnr"] - eager["psnr"]) < 2.0, (eager, graphed)
    assert graphed["file_bytes"] < 0.25 * graphed["table_bytes_fp32"], graphed
