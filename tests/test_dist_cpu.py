"""world_size-2 gloo test of the data-parallel path: sharded batch + one flat all-reduce == single-process gradient.
The hash-grid op on CPU is the torch oracle (test-only stand-in: the product op has no CPU path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CONFIGS, table_layout
from oracle import hashgrid_torch as ot
from shacira_amd import dist as sdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    dim, res, bw = CONFIGS["A"]
    _, first, T = table_layout(res, bw, dim)
    g = torch.Generator().manual_seed(0)
    coords = torch.rand(1001, dim, generator=g) * 2 - 1          # odd size: ragged shards
    target = torch.randn(1001, len(res) * 2, generator=g)
    table = torch.randn(T, 2, generator=g) * 0.1
    extra = torch.randn(7, generator=g)
    return res, bw, first, coords, target, table, extra


def _loss_sum(table, extra, coords, target, res, bw, first):
    feats = ot.hashgrid_forward(coords, table, first, res, bw)
    return ((feats * extra.sum() - target) ** 2).sum()


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, dev = sdist.init_from_env("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    res, bw, first, coords, target, table, extra = _problem()
    table = torch.nn.Parameter(table)
    extra = torch.nn.Parameter(extra)
    bucket = sdist.FlatGradients([table, extra])
    c, t = sdist.shard_batch(coords, rank, world), sdist.shard_batch(target, rank, world)
    for step in range(2):                                   # second step checks zero_() + re-accumulation into views
        bucket.zero_()
        loss = _loss_sum(table, extra, c, t, res, bw, first) / (coords.shape[0] * target.shape[1])
        loss.backward()
        assert table.grad.data_ptr() == bucket.view_of(table).data_ptr()   # autograd wrote into the flat buffer
        bucket.allreduce()
    if rank == 0:
        torch.save({"table": table.grad.clone(), "extra": extra.grad.clone(), "nbytes": bucket.nbytes}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    res, bw, first, coords, target, table, extra = _problem()
    table.requires_grad_(True)
    extra.requires_grad_(True)
    (_loss_sum(table, extra, coords, target, res, bw, first) / (coords.shape[0] * target.shape[1])).backward()
    np.testing.assert_allclose(got["table"].numpy(), table.grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(got["extra"].numpy(), extra.grad.numpy(), rtol=1e-4, atol=1e-6)
    assert got["nbytes"] >= (table.numel() + extra.numel()) * 4


# ------------------------------------------------------------------------------------------------ bench.py's step loop
def _patch_oracle_ops_levels():
    """Test-only stand-ins with the level-range / out / workspace / flags arguments of hip_ops.hashgrid_backward (the
    product has no CPU path): the torch oracle computes the whole gradient, the rows of the requested levels are written."""
    from shacira_amd import hip_ops

    def fwd(coords, codebook, first_idx, resolution, bw):
        return ot.hashgrid_forward(coords, codebook, first_idx.numpy(), list(resolution), bw)

    def bwd(dim, coords, grad_output, table_rows, table_dtype, first_idx, resolution, bw, feature_dim, levels=None,
            out=None, workspace=None, flags=0):
        table = torch.zeros(table_rows, feature_dim, requires_grad=True)
        feats = ot.hashgrid_forward(coords, table, first_idx.numpy(), list(resolution), bw)
        (g,) = torch.autograd.grad(feats, table, grad_output)
        if out is None:
            return g
        first = first_idx.tolist() + [table_rows]
        lb, le = (0, len(resolution)) if levels is None else levels
        out[first[lb]:first[le]] = g[first[lb]:first[le]]
        return out

    hip_ops.hashgrid_interpolate2d_cuda = fwd
    hip_ops.hashgrid_interpolate_cuda = fwd
    hip_ops.hashgrid_backward = bwd
    hip_ops.backward_workspace = lambda *a, **k: None


def _bench_step_worker(rank, world, port, out, chunks, collective):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    _patch_oracle_ops_levels()
    sdist.init_from_env("gloo")
    dim, res, bw = CONFIGS["A"]
    st = bench.build_step(torch.device("cpu"), rank, world, dim, res, bw, 2, 257, ar_chunks=chunks, collective=collective)
    for _ in range(2):                      # twice: the communication buffer is reused across steps
        feats, grad = st["step"]()
    torch.save({"grad": grad.clone(), "coords": st["coords"], "go": st["grad_out"], "groups": st["groups"]},
               f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunks,collective", [(3, "allreduce"), (1, "rs_ag"), (1, "allreduce")])
def test_bench_step_loop_four_ranks(tmp_path, chunks, collective):
    """bench.py's real step (build_step: level groups with overlapped all-reduces of row ranges, or reduce-scatter +
    all-gather on the padded flat buffer) on 4 gloo ranks: the reduced gradient equals the single-process gradient of the
    concatenated batch, on every rank."""
    out = str(tmp_path / "step")
    mp.spawn(_bench_step_worker, args=(4, _free_port(), out, chunks, collective), nprocs=4, join=True)
    recs = [torch.load(f"{out}.{r}") for r in range(4)]
    dim, res, bw = CONFIGS["A"]
    _, first, T = table_layout(res, bw, dim)
    coords = torch.cat([r["coords"] for r in recs])
    go = torch.cat([r["go"] for r in recs])
    table = torch.zeros(T, 2, requires_grad=True)
    (ref,) = torch.autograd.grad(ot.hashgrid_forward(coords, table, first, res, bw), table, go)
    assert len(recs[0]["groups"]) == (3 if chunks == 3 else 1)
    for r in recs:
        np.testing.assert_allclose(r["grad"].numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_shard_bounds_cover_batch_exactly():
    for n in (0, 1, 7, 1001, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [sdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert sdist.global_mean_loss(torch.tensor(8.0), 8).item() == 1.0      # local sum over the GLOBAL count


# ------------------------------------------------------------------------------------------------ DP image fit
def _patch_oracle_ops():
    """Test-only: the hash-grid operator on CPU is the C oracle (the product has no CPU path)."""
    from oracle import hashgrid_c as oc
    from shacira_amd import hip_ops

    def fwd(coords, codebook, first_idx, resolution, bw):
        return torch.from_numpy(oc.forward(coords.detach().numpy(), codebook.detach().numpy(), first_idx.numpy(),
                                           list(resolution), bw))

    def bwd(dim, coords, grad_output, table_rows, table_dtype, first_idx, resolution, bw, feature_dim):
        g = oc.backward(coords.detach().numpy(), grad_output.detach().numpy(), (table_rows, feature_dim),
                        first_idx.numpy(), list(resolution), bw)
        return torch.from_numpy(g.astype(np.float32))

    hip_ops.hashgrid_interpolate2d_cuda = fwd
    hip_ops.hashgrid_interpolate_cuda = fwd
    hip_ops.hashgrid_backward = bwd


def _fit_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from shacira_amd import harness
    _patch_oracle_ops()
    sdist.init_from_env("gloo")
    r = harness.fit_image(torch.device("cpu"), steps=25, height=32, width=48, seed=4, rank=rank, world=world)
    if rank == 0:
        torch.save(r, out)
    dist.barrier()
    dist.destroy_process_group()


def test_image_fit_two_ranks_matches_one_rank(tmp_path, monkeypatch):
    """Sharded pixels + one gradient all-reduce per step reproduce the single-process fit (PSNR at a fixed step)."""
    out = str(tmp_path / "fit.pt")
    mp.spawn(_fit_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    two = torch.load(out)
    from shacira_amd import harness, hip_ops
    saved = (hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_backward)
    try:
        _patch_oracle_ops()
        one = harness.fit_image(torch.device("cpu"), steps=25, height=32, width=48, seed=4)
    finally:
        hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_backward = saved
    assert abs(two["psnr"] - one["psnr"]) < 0.05, (two["psnr"], one["psnr"])
    assert two["rgb_loss"] == pytest.approx(one["rgb_loss"], rel=1e-3)
    assert two["bpp"] == pytest.approx(one["bpp"], rel=1e-2)


def _field_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from shacira_amd import harness
    _patch_oracle_ops()
    sdist.init_from_env("gloo")
    r = harness.fit_field_3d(torch.device("cpu"), steps=15, rays=64, samples_per_ray=8, codebook_bitwidth=9,
                             max_grid_res=32, num_lods=5, val_points=1024, rank=rank, world=world)
    if rank == 0:
        torch.save(r, out)
    dist.barrier()
    dist.destroy_process_group()


def test_field_fit_3d_two_ranks_matches_one_rank(tmp_path):
    """Config E in miniature: ray points sharded over 2 ranks + one all-reduce of the flat gradient buffer per step
    gives the same PSNR at a fixed step as one process on the same global batch."""
    out = str(tmp_path / "field.pt")
    mp.spawn(_field_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    two = torch.load(out)
    from shacira_amd import harness, hip_ops
    saved = (hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_backward)
    try:
        _patch_oracle_ops()
        one = harness.fit_field_3d(torch.device("cpu"), steps=15, rays=64, samples_per_ray=8, codebook_bitwidth=9,
                                   max_grid_res=32, num_lods=5, val_points=1024)
    finally:
        hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_backward = saved
    assert abs(two["psnr"] - one["psnr"]) < 0.05, (two["psnr"], one["psnr"])


def test_flat_gradients_hide_parameters_outside_the_loss():
    """A parameter that takes no part in a step's loss keeps grad=None for the optimiser (as with one GPU, where
    zero_grad(set_to_none=True) leaves it None): no weight decay / moment decay is applied to it."""
    a = torch.nn.Parameter(torch.ones(4))
    b = torch.nn.Parameter(torch.ones(3))
    bucket = sdist.FlatGradients([a, b])
    opt = torch.optim.Adam([a, b], lr=0.1, weight_decay=0.5)
    for _ in range(2):
        bucket.zero_()
        (a * 2.0).sum().backward()                 # b is not in the graph
        hidden = bucket.hide_untouched()
        assert [id(p) for p in hidden] == [id(b)] and b.grad is None and a.grad is not None
        opt.step()
        bucket.restore(hidden)
        assert b.grad is not None and b.grad.data_ptr() == bucket.view_of(b).data_ptr()
    assert torch.equal(b.detach(), torch.ones(3)) and not torch.equal(a.detach(), torch.ones(4))
    assert b not in opt.state or not opt.state[b]
