"""The batch PLAN (ABI 10 / 11): the forward's sample sort handed to the backward of the same coordinates
(shacira_hashgrid_plan_bytes / _forward_planned / _backward_planned). The planned calls must give what the plain calls give --
forward bit-identical to the oracle, gradients within the 1e-5 bar of the fp64-accumulating oracle -- for every placement of
the brick pass, on clustered / degenerate batches, on half tables and F = 4, and through the autograd Functions.
Reference semantics: wisp/ops/grid.py:69-111 (coords saved by forward, reused by backward), kernels
wisp/csrc/ops/hashgrid_interpolate_cuda.cu:47-109, :143-221."""
import numpy as np
import pytest
import torch

from conftest import CONFIGS, geo, table_layout
from oracle import hashgrid_c as oc
from test_gpu_parity import RTOL, _assert_grad_close, _level_margin, _ops, _problem, dev  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def brick_wherever_the_shape_allows():
    """The automatic rule takes the brick pass only where it was measured to win (fp32, F = 2); these
    tests run it on every shape that allows it (option bwd_brick = 1) at sizes the oracle finishes in seconds."""
    from shacira_amd import _lib
    _lib.set_option("bwd_brick", 1)
    yield
    _lib.set_option("bwd_brick", -1)


def _to(dev, *arrays):
    return [torch.from_numpy(a).to(dev) for a in arrays]


def _planned_pair(dev, dim, res, bw, coords, table, go, first, dtype=torch.float32, F=2):
    ops = _ops()
    tc, tf = _to(dev, coords, first)
    tt = torch.from_numpy(table).to(dev).to(dtype)
    tg = torch.from_numpy(go).to(dev).to(dtype)
    plan = ops.hashgrid_plan_buffer(dim, tc, tt, res, bw)
    fwd = ops.hashgrid_interpolate_cuda if dim == 3 else ops.hashgrid_interpolate2d_cuda
    feats = fwd(tc, tt, tf, res, bw, plan=plan)
    grad = ops.hashgrid_backward(dim, tc, tg, table.shape[0], dtype, tf, res, bw, F, plan=plan)
    torch.cuda.synchronize()
    return plan, feats, grad


def test_plan_buffer_only_for_shapes_that_sort(dev):
    """plan_bytes is 0 for shapes whose forward sorts nothing (small batches, cache-resident tables, 2-D below its
    threshold); the planned entry points then behave like the plain ones with plan = NULL."""
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    sizes, first, T = table_layout(res, bw, dim)
    assert ops.hashgrid_plan_bytes(dim, 4096, T, torch.float32, res, bw, 2) == 0
    assert ops.hashgrid_plan_bytes(dim, 1 << 18, T, torch.float32, res, bw, 2) >= (1 << 18) * 16
    assert ops.hashgrid_plan_bytes(dim, 1 << 18, T, torch.float64, res, bw, 2) == 0
    dimb, resb, bwb = CONFIGS["B"]                      # Kodak table: LDS resident, never sorted
    _, _, Tb = table_layout(resb, bwb, dimb)
    assert ops.hashgrid_plan_bytes(dimb, 1 << 20, Tb, torch.float32, resb, bwb, 2) == 0


@pytest.mark.parametrize("n", [(1 << 18) + 13, (1 << 17) * 3 + 1])
def test_planned_forward_and_backward_against_the_oracle(dev, n):
    """Config D's table at sizes that sort: edge coordinates (+-1, NaN, out of range), a cluster that overfills one block and
    one coarse bin of the sort. Forward bit-identical to the oracle, planned gradient within the bar AND with the margin the
    plain path is held to (a tenth of the bar)."""
    dim, res, bw = CONFIGS["D"]
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=5)
    coords[1000:41_000] = (np.float32(-0.43) + np.random.default_rng(6).normal(0, 0.002, (40_000, dim))).astype(np.float32)
    plan, feats, grad = _planned_pair(dev, dim, res, bw, coords, table, go, first)
    assert plan is not None
    sl = np.r_[0:64, 900:1200, n - 200:n]
    assert np.array_equal(feats.cpu().numpy()[sl], oc.forward(coords[sl], table, first, res, bw))
    ref = oc.backward(coords, go, (T, 2), first, res, bw)
    got = grad.cpu().numpy()
    _assert_grad_close(got, ref, first, sizes)
    assert _level_margin(got, ref, first, sizes) < 1e-6
    # conservation per level
    for l in range(len(res)):
        lo, hi = int(first[l]), int(first[l]) + sizes[l]
        np.testing.assert_allclose(got[lo:hi].astype(np.float64).sum(0), go[:, 2 * l:2 * l + 2].astype(np.float64).sum(0),
                                   rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("opts", [{"bwd_brick_fork": 0}, {"bwd_brick_fork": 1}, {"bwd_brick_fork": 2, "bwd_brick_span": 1},
                                  {"bwd_brick_span": 3}, {"bwd_brick_lo": 2, "bwd_brick_hi": 7, "bwd_brick_fork": 0},
                                  {"bwd_brick_lo": 5, "bwd_brick_hi": 6}, {"bwd_brick": 0}, {"bwd_item12": 0},
                                  {"bwd_item12": 1, "bwd_brick_fork": 0}])
def test_brick_pass_placements_and_level_ranges(dev, opts):
    """Every placement of the brick pass (last on the stream, beside the scatter / consume pass), unit widths, explicit level
    ranges incl. hashed levels, the plan given but the brick pass switched off, and the item stream in 16-byte units (the
    planned path's default is the 12-byte stream): same gradient, within the bar."""
    from shacira_amd import _lib
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    n = (1 << 18) + 77
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=9)
    tc, tt, tg, tf = _to(dev, coords, table, go, first)
    plan = ops.hashgrid_plan_buffer(dim, tc, tt, res, bw)
    ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw, plan=plan)
    ref = oc.backward(coords, go, (T, 2), first, res, bw)
    saved = {k: _lib.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            _lib.set_option(k, v)
        got = ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, plan=plan).cpu().numpy()
    finally:
        for k, v in saved.items():
            _lib.set_option(k, v)
    _assert_grad_close(got, ref, first, sizes)
    assert _level_margin(got, ref, first, sizes) < 1e-6


def test_plan_reuse_across_steps(dev):
    """A caller that trains on a fixed batch keeps the plan: the forward with SHACIRA_PLAN_READY skips its sort and gives the
    same features; the backward reads the same plan again."""
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    n = 1 << 18
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=12)
    tc, tt, tg, tf = _to(dev, coords, table, go, first)
    plan = ops.hashgrid_plan_buffer(dim, tc, tt, res, bw)
    f0 = ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw, plan=plan)
    g0 = ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, plan=plan)
    tt2 = tt * 1.5
    f1 = ops.hashgrid_interpolate_cuda(tc, tt2, tf, res, bw, plan=plan, plan_ready=True)
    assert torch.equal(f1, ops.hashgrid_interpolate_cuda(tc, tt2, tf, res, bw))
    g1 = ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, plan=plan)
    ref = oc.backward(coords, go, (T, 2), first, res, bw)
    _assert_grad_close(g0.cpu().numpy(), ref, first, sizes)
    _assert_grad_close(g1.cpu().numpy(), ref, first, sizes)
    assert torch.equal(f0, ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw))


@pytest.mark.parametrize("kind", ["one_block", "blob", "one_plane", "two_points"])
def test_degenerate_batches_through_the_plan(dev, kind):
    """Batches the sort's coarse bins and the brick units are not sized for: everything in ONE block (one over-full bin,
    its 8 192-record chunks on a workgroup each; the brick pass's windows beyond a group's first unit), a Gaussian blob over a
    uniform background, everything on one z-plane, two points."""
    dim, res, bw = CONFIGS["D"]
    n = (1 << 18) + 5
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=31, edge=False)
    rng = np.random.default_rng(32)
    if kind == "one_block":
        coords[:] = (0.123 + rng.uniform(-0.01, 0.01, coords.shape)).astype(np.float32)
    elif kind == "blob":   # several over-full bins and groups of different sizes beside ordinary ones
        coords[:] = np.clip(rng.normal(0.2, 0.07, coords.shape), -1.0, 1.0).astype(np.float32)
        coords[::7] = rng.uniform(-1.0, 1.0, coords[::7].shape).astype(np.float32)
    elif kind == "one_plane":
        coords[:, 2] = np.float32(0.5)
    else:
        coords[: n // 2] = np.float32(-0.77)
        coords[n // 2:] = np.float32(0.31)
    plan, feats, grad = _planned_pair(dev, dim, res, bw, coords, table, go, first)
    sl = np.r_[0:256, n - 256:n]
    assert np.array_equal(feats.cpu().numpy()[sl], oc.forward(coords[sl], table, first, res, bw))
    ref = oc.backward(coords, go, (T, 2), first, res, bw)
    _assert_grad_close(grad.cpu().numpy(), ref, first, sizes)


def test_non_finite_gradients_through_the_brick_pass(dev):
    """inf / NaN in a level the brick pass accumulates (and in one it does not): that level's touched rows become non-finite,
    every other level keeps its accuracy (the level's max |gradient| turns its images to fp64)."""
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    n = (1 << 18) + 1
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=41, edge=False)
    tc, tt, tf = _to(dev, coords, table, first)
    plan = ops.hashgrid_plan_buffer(dim, tc, tt, res, bw)
    ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw, plan=plan)
    bad = go.copy()
    bad[100, 2 * 1] = np.inf          # level 1: brick level
    bad[200, 2 * 9 + 1] = np.nan      # level 9: item stream
    got = ops.hashgrid_backward(dim, tc, torch.from_numpy(bad).to(dev), T, torch.float32, tf, res, bw, 2, plan=plan).cpu().numpy()
    ref = oc.backward(coords, go, (T, 2), first, res, bw)
    for l in range(len(res)):
        lo, hi = int(first[l]), int(first[l]) + sizes[l]
        if l in (1, 9):
            assert not np.isfinite(got[lo:hi]).all()
            assert np.isfinite(got[lo:hi]).mean() > 0.5
        else:
            np.testing.assert_allclose(got[lo:hi], ref[lo:hi], rtol=RTOL, atol=RTOL * np.abs(ref[lo:hi]).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_planned_backward_nerf_lego_table(dev, dtype):
    """nerf_lego.yaml's table (24 levels, F = 4, bw 19, res 16..512: eleven dense levels) in fp32 and as the half table AMP
    trains: planned backward against the fp64 oracle (half: the half-precision bar of the plain path)."""
    dim, bw, F = 3, 19, 4
    res = geo(16, 512, 24)
    n = 150_001
    sizes, first, T = table_layout(res, bw, dim)
    rng = np.random.default_rng(51)
    coords = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    coords[0], coords[1], coords[2] = 1.0, -1.0, np.nan
    table = (rng.standard_normal((T, F)) * 0.01).astype(np.float32)
    go = rng.standard_normal((n, len(res) * F)).astype(np.float32)
    if dtype == torch.float16:
        go = go.astype(np.float16).astype(np.float32)
    plan, feats, grad = _planned_pair(dev, dim, res, bw, coords, table, go, first, dtype=dtype, F=F)
    assert plan is not None
    ref = oc.backward(coords, go, (T, F), first, res, bw)
    _assert_grad_close(grad.float().cpu().numpy(), ref, first, sizes, rtol=RTOL if dtype == torch.float32 else 2e-3)


def test_autograd_function_keeps_the_plan(dev):
    """wisp.ops.grid.hashgrid: the Function's forward allocates the plan when the codebook needs a gradient, saves it with
    the coordinates, and its backward passes it on -- same gradient as the operator called plainly."""
    from shacira_amd.wisp.ops import grid as G
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    n = 1 << 18
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=61)
    tc, tg, tf = _to(dev, coords, go, first)
    cb = torch.from_numpy(table).to(dev).requires_grad_(True)
    szs = torch.tensor(sizes, dtype=torch.int32, device=dev)
    feats = G.hashgrid(tc, res, bw, 0, cb, szs, tf)
    assert feats.grad_fn is not None
    feats.backward(tg)
    plain = ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2)
    ref = oc.backward(coords, go, (T, 2), first, res, bw)
    _assert_grad_close(cb.grad.cpu().numpy(), ref, first, sizes)
    assert float((cb.grad - plain).abs().max()) <= 2e-6 * float(plain.abs().max())
    with torch.no_grad():                     # no gradient needed: no plan buffer is allocated, same features
        assert torch.equal(G.hashgrid(tc, res, bw, 0, cb, szs, tf), feats)


def test_planned_pair_replays_from_one_graph(dev):
    """Planned forward + backward (side stream of the brick pass included) captured once and replayed on new data."""
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    n = (1 << 18) + 100
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=71)
    tc, tt, tg, tf = _to(dev, coords, table, go, first)
    out = torch.empty((T, 2), device=dev)
    plan = ops.hashgrid_plan_buffer(dim, tc, tt, res, bw)
    ws = ops.backward_workspace(dim, n, T, torch.float32, res, bw, 2, dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # one eager call creates the library's side-stream objects / kernel attributes
        ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw, plan=plan)
        ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws, plan=plan)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        feats = ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw, plan=plan)
        ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, out=out, workspace=ws, plan=plan)
    rng = np.random.default_rng(72)
    coords2 = (rng.uniform(-1, 1, (n, dim)) ** 3).astype(np.float32)
    go2 = rng.standard_normal(go.shape).astype(np.float32) * 3.0
    tc.copy_(torch.from_numpy(coords2))
    tg.copy_(torch.from_numpy(go2))
    out.fill_(-3.0)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(feats[:4096].cpu().numpy(), oc.forward(coords2[:4096], table, first, res, bw))
    _assert_grad_close(out.cpu().numpy(), oc.backward(coords2, go2, (T, 2), first, res, bw), first, sizes)


def test_planned_call_in_its_own_smaller_workspace(dev):
    """shacira_hashgrid_backward_planned_workspace_bytes: the planned call runs in exactly that many bytes (a guard pattern
    behind them stays untouched); a planned call on a gradient that is not 16-byte aligned runs the plain passes and refuses
    the smaller buffer loudly."""
    from shacira_amd import _lib
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    n = (1 << 18) + 36
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=91)
    tc, tt, tg, tf = _to(dev, coords, table, go, first)
    plan = ops.hashgrid_plan_buffer(dim, tc, tt, res, bw)
    ops.hashgrid_interpolate_cuda(tc, tt, tf, res, bw, plan=plan)
    small = ops.backward_workspace(dim, n, T, torch.float32, res, bw, 2, dev, planned=True).numel()
    plain = ops.backward_workspace(dim, n, T, torch.float32, res, bw, 2, dev).numel()
    assert small < 0.85 * plain
    buf = torch.full((small + 4096,), 0x5A, dtype=torch.uint8, device=dev)
    grad = ops.hashgrid_backward(dim, tc, tg, T, torch.float32, tf, res, bw, 2, workspace=buf[:small], plan=plan)
    torch.cuda.synchronize()
    assert bool((buf[small:] == 0x5A).all())
    _assert_grad_close(grad.cpu().numpy(), oc.backward(coords, go, (T, 2), first, res, bw), first, sizes)
    # unaligned gradient rows: the same values at an odd 4-byte offset
    flat = torch.empty((tg.numel() + 1,), device=dev)
    flat[1:].copy_(tg.reshape(-1))
    tg_odd = flat[1:].view(tg.shape)
    assert tg_odd.data_ptr() % 16 != 0
    with pytest.raises(RuntimeError, match="workspace"):
        ops.hashgrid_backward(dim, tc, tg_odd, T, torch.float32, tf, res, bw, 2, workspace=buf[:small], plan=plan)
    grad2 = ops.hashgrid_backward(dim, tc, tg_odd, T, torch.float32, tf, res, bw, 2, plan=plan)
    torch.cuda.synchronize()
    _assert_grad_close(grad2.cpu().numpy(), oc.backward(coords, go, (T, 2), first, res, bw), first, sizes)


def test_full_size_planned_step_with_the_automatic_rule(dev):
    """BASELINE's headline batch (2^20 samples, config D's table) with every option at its default: the planned pair takes
    the sorted path with the brick pass by itself; forward slices bit-identical, gradient within the bar of the oracle."""
    from shacira_amd import _lib
    _lib.set_option("bwd_brick", -1)
    dim, res, bw = CONFIGS["D"]
    n = 1 << 20
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=81)
    plan, feats, grad = _planned_pair(dev, dim, res, bw, coords, table, go, first)
    sl = np.r_[0:512, n - 512:n]
    assert np.array_equal(feats.cpu().numpy()[sl], oc.forward(coords[sl], table, first, res, bw))
    ref = oc.backward(coords, go, (T, 2), first, res, bw)
    got = grad.cpu().numpy()
    _assert_grad_close(got, ref, first, sizes)
    assert _level_margin(got, ref, first, sizes) < 1e-6
    lhs = float((feats.double() * torch.from_numpy(go).to(dev).double()).sum())
    rhs = float((torch.from_numpy(table).to(dev).double() * grad.double()).sum())
    assert lhs == pytest.approx(rhs, rel=1e-6)                      # adjointness <F t, go> == <t, F^T go>


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_through_the_plan(dev, seed):
    """Randomised 3-D shapes and sample distributions through the planned pair with the brick pass wherever the shape allows it
    (tools/fuzz_shapes.py runs many more seeds): level counts, resolutions, table sizes, F = 2 / 4, fp32 / fp16, batches from
    2^18; uniform, cubed (dense centre), a blob over a background, ray points."""
    from shacira_amd import harness
    rng = np.random.default_rng(9000 + seed)
    dim = 3
    L = int(rng.choice([4, 8, 12, 16, 20]))
    res = geo(int(rng.integers(4, 20)), int(rng.integers(256, 2049)), L)
    bw = int(rng.integers(15, 20))
    F = int(rng.choice([2, 4]))
    N = int(rng.choice([(1 << 18) + 5, 300_001, 400_003]))
    dtype = torch.float16 if seed % 3 == 2 else torch.float32
    sizes, first, T, coords, table, go = _problem(dim, res, bw, N, F=F, seed=seed, edge=bool(seed & 1))
    kind = int(rng.integers(0, 4))
    if kind == 1:
        coords[:] = (rng.uniform(-1, 1, coords.shape) ** 3).astype(np.float32)
    elif kind == 2:
        coords[:] = np.clip(rng.normal(rng.uniform(-0.5, 0.5), 0.05, coords.shape), -1, 1).astype(np.float32)
        coords[::5] = rng.uniform(-1, 1, coords[::5].shape).astype(np.float32)
    elif kind == 3:
        rays = harness.ray_points((N + 15) // 16, 16, torch.Generator().manual_seed(seed)).numpy()
        coords[:] = rays[:N]
    stored = table.astype(np.float16).astype(np.float32) if dtype == torch.float16 else table
    go_s = go.astype(np.float16).astype(np.float32) if dtype == torch.float16 else go
    plan, feats, grad = _planned_pair(dev, dim, res, bw, coords, table, go, first, dtype=dtype, F=F)
    sl = np.r_[0:2048, N - 2048:N]
    ref = oc.forward(coords[sl], stored, first, res, bw)
    want = ref.astype(np.float16) if dtype == torch.float16 else ref
    assert np.array_equal(feats.cpu().numpy()[sl], want), (res, bw, F, N, dtype, kind)
    ref_g = oc.backward(coords, go_s, (T, F), first, res, bw)
    _assert_grad_close(grad.float().cpu().numpy(), ref_g, first, sizes, rtol=RTOL if dtype == torch.float32 else 2e-3)


def test_planned_call_split_into_sub_batches_falls_back_to_the_plain_passes(dev):
    """A call whose item array exceeds the cap (option bin_batch_mib) runs in sub-batches; the brick pass takes whole calls
    only, so such a planned call ignores its plan in the backward -- same results, and the workspace query says the plain size."""
    from shacira_amd import _lib
    ops = _ops()
    dim, res, bw = CONFIGS["D"]
    n = 300_001
    sizes, first, T, coords, table, go = _problem(dim, res, bw, n, seed=97)
    _lib.set_option("bin_batch_mib", 64)
    try:
        assert (ops.hashgrid_backward_workspace_bytes(dim, n, T, torch.float32, res, bw, 2, planned=True)
                == ops.hashgrid_backward_workspace_bytes(dim, n, T, torch.float32, res, bw, 2))
        plan, feats, grad = _planned_pair(dev, dim, res, bw, coords, table, go, first)
    finally:
        _lib.set_option("bin_batch_mib", 1536)
    _assert_grad_close(grad.cpu().numpy(), oc.backward(coords, go, (T, 2), first, res, bw), first, sizes)
