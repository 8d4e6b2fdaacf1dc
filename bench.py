#!/usr/bin/env python3
"""Headline benchmark: hash-grid fwd+bwd samples/s (16 levels, F=2) on synthetic point batches.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one pass of the hot path over one batch: the forward operator (all 16 levels; it builds the batch's plan -- the
sample sort -- INSIDE the timed step, every step, into a caller-owned buffer), the backward operator (gradient scatter-add
into the 48.8 MB codebook, reading that plan: the hand-off the autograd Function makes, shacira_amd/wisp/ops/grid.py) and,
for N > 1, ONE RCCL all-reduce of that gradient over xGMI.
Workload = BASELINE.json's headline point "H"/config D shape: 3-D `nerf_hash` grid (L=16, F=2, bw=19, res 17..2049,
T = 6 098 925 rows, fp32), 2^20 uniformly random samples per GPU (weak scaling), inputs resident in HBM.
Rank 0 prints ONE JSON line (see the driver contract); `roofline` prices the dominant operator against the
8 TB/s HBM roof with ALGORITHMIC bytes, `cpu_baseline` times the pure-PyTorch restatement of the
reference kernels (oracle/hashgrid_torch.py) on this box's host cores in the same run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid argument` otherwise);
# it is exported on the boxes already -- kept here for launches from a stripped environment, before HIP is loaded
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak is ~6.3 TB/s


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


WORKLOADS = {
    # name: (dim, resolutions, bitwidth, feature_dim, samples per GPU)
    "S1_nerf_hash_3d_L16_F2_bw19_N2^20": (3, geo(16, 2048, 16), 19, 2, 1 << 20),
    "S2_kodak_2d_L16_F2_bw19_N2^20": (2, geo(16, 2048, 16), 19, 2, 1 << 20),
    "B_kodak_2d_L16_F2_bw11_N393216": (2, geo(16, 512, 16), 11, 2, 393216),
    "C_kodak24_2d_L16_F2_bw11_N9437184": (2, geo(16, 512, 16), 11, 2, 24 * 393216),
    "D_nerf_lego_3d_L16_F2_bw19_N65536": (3, geo(16, 2048, 16), 19, 2, 65536),
    # config E's per-GPU shard: the same 4096-ray batch split over 8 GPUs = 512 rays x 16 samples per rank
    "E_shard_3d_L16_F2_bw19_N8192": (3, geo(16, 2048, 16), 19, 2, 8192),
    # the reference's SHIPPED configurations (not BASELINE's 16-level F = 2 metric shape): kodak.yaml's table as the operator
    # sees it (24 levels, feature_dim 1 repeated to F = 2, bw 11) and nerf_lego.yaml's (24 levels, F = 4, bw 19; 4096 rays x
    # 100 steps)
    "kodak_yaml_2d_L24_F1rep2_bw11_N393216": (2, geo(16, 512, 24), 11, 2, 393216),
    "nerf_lego_yaml_3d_L24_F4_bw19_N409600": (3, geo(16, 512, 24), 19, 4, 409600),
}
# the same shapes with an fp16 table and fp16 gradients: the reference's NeRF runs under AMP (wisp/ops/grid.py:73 casts the
# operator's inputs to half), so this is the precision its trainer actually calls the kernels with
HALF_VARIANTS = {"nerf_lego_yaml_3d_L24_F4_bw19_N409600_fp16": "nerf_lego_yaml_3d_L24_F4_bw19_N409600",
                 "S1_nerf_hash_3d_L16_F2_bw19_N2^20_fp16": "S1_nerf_hash_3d_L16_F2_bw19_N2^20"}
SECONDARY = ["S2_kodak_2d_L16_F2_bw19_N2^20", "B_kodak_2d_L16_F2_bw11_N393216", "C_kodak24_2d_L16_F2_bw11_N9437184",
             "D_nerf_lego_3d_L16_F2_bw19_N65536", "E_shard_3d_L16_F2_bw19_N8192", "kodak_yaml_2d_L24_F1rep2_bw11_N393216",
             "nerf_lego_yaml_3d_L24_F4_bw19_N409600", "nerf_lego_yaml_3d_L24_F4_bw19_N409600_fp16",
             "S1_nerf_hash_3d_L16_F2_bw19_N2^20_fp16"]


def quick_measure(name, device, iters=20):
    """fwd / bwd operator times of another BASELINE config (same protocol, fewer iterations); not the headline."""
    from shacira_amd import hip_ops
    half = name in HALF_VARIANTS
    dim, res, bw, F, n = WORKLOADS[HALF_VARIANTS.get(name, name)]
    L = len(res)
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)).to(device)
    T = int(sum(sizes))
    g = torch.Generator().manual_seed(7)
    table = (torch.randn(T, F, generator=g) * 0.01).to(device)
    if half:
        table = table.half()
    if name.startswith("D_") or name.startswith("E_shard"):
        # NeRF-like ray points -- 4096 rays from the radius-3 sphere, 16 stratified samples inside the cube
        from shacira_amd import harness
        coords = harness.ray_points(n // 16, 16, g).contiguous().to(device)
    else:
        coords = (torch.rand(n, dim, generator=g) * 2 - 1).to(device)
    go = torch.randn(n, L * F, generator=g).to(device).to(table.dtype)
    fwd = hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda
    # the plan hand-off of the autograd Function (shacira_amd/wisp/ops/grid.py): the forward fills it every call, the backward
    # reads it; None for the shapes whose forward sorts nothing
    plan = hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)
    pkw = {} if plan is None else {"plan": plan}
    for _ in range(3):
        fwd(coords, table, first, res, bw, **pkw)
        hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, **pkw)
    torch.cuda.synchronize()
    # the headline loop's protocol: steps issued back to back (no host synchronisation inside the region, so that small
    # batches are not charged the Python wrapper's ~20 us per call while the GPU sits idle), HIP events around each operator
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    for ev in evs:
        ev[0].record()
        fwd(coords, table, first, res, bw, **pkw)
        ev[1].record()
        hip_ops.hashgrid_backward(dim, coords, go, T, table.dtype, first, res, bw, F, **pkw)
        ev[2].record()
    torch.cuda.synchronize()
    tf = float(np.mean([ev[0].elapsed_time(ev[1]) for ev in evs]))
    tb = float(np.mean([ev[1].elapsed_time(ev[2]) for ev in evs]))
    bf, bb = algorithmic_bytes_per_sample(dim, L, F, 2 if half else 4)
    gbs = (bf + bb) * n / ((tf + tb) * 1e-3) / 1e9
    # SURVEY 8(d)'s accounting charges every corner's 8 bytes as HBM bytes; a table that sits in the caches (the Kodak tables
    # are 0.2-0.3 MB) is read from L1 / LDS, so its fraction says how far the kernels are from "every algorithmic byte at HBM
    # speed", not from the HBM limit (it can exceed 1). The sample streams alone (coordinates + feature rows out, coordinates +
    # gradient rows in) are what such a table really moves through HBM:
    esz = 2 if half else 4
    stream = (2 * dim * 4 + 2 * L * F * esz) * n / ((tf + tb) * 1e-3) / 1e9
    table_mb = T * F * esz / 1e6
    # What binds these operators is not the precision of the bytes: the forward pays one L2 -> L1 line
    # transfer (2 clk per 128-byte line and CU) per x-pair of corners on every level finer than the batch's density, whatever
    # the row size; the backward moves its items once out and once in (8 + 4F bytes per x-pair with fp32 payloads, 8 / 16 bytes
    # with half payloads; levels that fit LDS images move none) -- the binding part for binned levels -- and pays one 64-bit LDS
    # atomic per (corner, feature) (2.3 T/s measured, profiles/r01_microbench2), which binds only where there is no item stream
    # (direct levels: the image configs).
    n_fine = sum(1 for r in res if float(r) ** dim > 4.0 * n) if dim == 3 else 0
    fwd_floor = n_fine * n * (2 ** (dim - 1)) * 2.0 / (256 * 2.4e9) * 1e3
    lds_floor = n * L * (2 ** dim) * F / 2.3e12 * 1e3
    binned = sum(1 for sz in sizes if sz * F * 8 > 128 * 1024)            # levels larger than one 128 KiB image
    item_b = (8 if F == 2 else 16) if half else 8 + 4 * F
    stream_floor = 2.0 * binned * n * (2 ** (dim - 1)) * item_b / 5.4e12 * 1e3
    # A table that lives in the caches / in LDS (a few hundred KB: the image configs) moves only its sample streams through HBM:
    # for those the roofline fraction is quoted on the streams (the algorithmic figure came out above 1)
    cache_resident = table_mb < 4.0
    return {"samples_per_s": n / ((tf + tb) * 1e-3), "ms_forward": tf, "ms_backward": tb, "samples": n,
            "algorithmic_GBps": gbs, "table_MB": table_mb,
            "frac_of_8TBps": (stream if cache_resident else gbs) / HBM_PEAK_GBS,
            "frac_basis": "sample streams (table is cache / LDS resident)" if cache_resident else "algorithmic bytes",
            "sample_streams_GBps": stream, "dtype": "f16" if half else "f32",
            "bound_model": {"forward_line_rate_floor_ms": fwd_floor, "backward_lds_atomic_floor_ms": lds_floor,
                            "backward_item_stream_floor_ms": stream_floor,
                            "note": "floors of the binding resources (L2->L1 lines of the fine levels; 64-bit LDS atomics at "
                                    "2.3 T/s; item stream written + read at 5.4 TB/s), not of HBM bytes"}}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the HIP sources and headers of the library: identifies WHICH kernels a committed
    counter file was measured on (a git hash cannot: committing the file changes it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "shacira_amd", "csrc", "*.hip")) +
                       glob.glob(os.path.join(ROOT, "shacira_amd", "csrc", "*.h")) +
                       glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_sample(dim, L, F, s=4):
    """fwd = 4d + L*2^d*F*s + L*F*s ; bwd = 4d + L*F*s + L*2^d*F*s."""
    one = 4 * dim + L * (2 ** dim) * F * s + L * F * s
    return one, one


def cpu_baseline(dim, res, bw, F, first, T, n_samples, budget_s, seed=0):
    """Pure-PyTorch restatement of the reference kernels on the host cores, same synthetic distribution, bounded to
    about `budget_s` seconds of CPU work (the first pass sizes the number of timed passes)."""
    from oracle import hashgrid_c as oc
    from oracle import hashgrid_torch as ot
    # Threads: BASELINE.md section 3 says os.cpu_count(); measured on the 256-thread GPU box that protocol is 130x SLOWER
    # than 32 threads on this workload (index_add_ contention: 300 s per 2^20-sample pass against 0.57 s per 2^17 samples
    # at 32 threads, profiles/r02_cpu_baseline_threads.md), so the leg is bounded as the bench contract asks: <= 32
    # threads, a 2^17-sample slice of the same workload. Both deviations are stated in the line's `sample` text.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(seed)
    coords = torch.rand(n_samples, dim, generator=g) * 2 - 1
    table = torch.randn(T, F, generator=g) * 0.01
    go = torch.randn(n_samples, len(res) * F, generator=g)
    t0 = time.perf_counter()
    ot.hashgrid_fwd_bwd(coords, table, first, res, bw, go)  # warm-up, also sizes the run
    warm = time.perf_counter() - t0
    iters = max(1, min(10, int(budget_s / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(iters):
        ot.hashgrid_fwd_bwd(coords, table, first, res, bw, go)
    dt = (time.perf_counter() - t0) / iters
    # sanity figure: the scalar C oracle, one thread, on a slice
    n_c = min(n_samples, 1 << 16)
    t0 = time.perf_counter()
    oc.forward(coords[:n_c].numpy(), table.numpy(), first, res, bw)
    oc.backward(coords[:n_c].numpy(), go[:n_c].numpy(), (T, F), first, res, bw)
    dt_c = time.perf_counter() - t0
    return {"value": n_samples / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{iters} timed fwd+bwd passes (+1 warm-up) over {n_samples} samples of the same workload "
                      f"(same table, same coordinate distribution), oracle/hashgrid_torch.py, {cores} torch threads",
            "s_per_pass": dt, "c_oracle_1thread_samples_per_s": n_c / dt_c}


def reference_kernels_on_this_gpu(device, dim, res, bw, F, first, T, n_samples, seed=0, iters=3):
    """Part of the cpu_baseline leg (a reported baseline, never the thing measured): the reference's OWN operators
    (oracle/_ref/shacira_ref_ops.so = reference hashgrid_interpolate{,2d}_cuda.cu + hashgrid_interpolate.cpp built for gfx950,
    oracle/ref_build.py) timed on the same GPU, the same workload at full size. None when that library did not travel."""
    from oracle import ref_build
    if not os.path.exists(ref_build.OUT):
        return None
    ref = ref_build.load()
    g = torch.Generator().manual_seed(seed)
    coords = (torch.rand(n_samples, dim, generator=g) * 2 - 1).to(device)
    table = (torch.randn(T, F, generator=g) * 0.01).to(device)
    go = torch.randn(n_samples, len(res) * F, generator=g).to(device)
    tf = torch.as_tensor(first, dtype=torch.int32, device=device)
    fwd = ref.hashgrid_interpolate_cuda if dim == 3 else ref.hashgrid_interpolate2d_cuda
    bwd = ref.hashgrid_interpolate_backward_cuda if dim == 3 else ref.hashgrid_interpolate2d_backward_cuda
    res = [int(r) for r in res]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_f = t_b = 0.0
    for it in range(iters + 1):
        ev[0].record()
        fwd(coords, table, tf, res, bw)
        ev[1].record()
        bwd(coords, go, table, tf, res, bw, F, False)
        ev[2].record()
        torch.cuda.synchronize()
        if it:  # first pass = warm-up
            t_f += ev[0].elapsed_time(ev[1])
            t_b += ev[1].elapsed_time(ev[2])
    t_f /= iters
    t_b /= iters
    return {"value": n_samples / ((t_f + t_b) * 1e-3), "unit": "samples/s", "ms_forward": t_f, "ms_backward": t_b,
            "samples": n_samples, "kind": "reference",
            "what": "the reference's hash-grid operators (L launches each way, one atomicAdd per corner and feature) built "
                    "for gfx950 by oracle/ref_build.py, torch events on the current stream"}


def psnr_parity(device, steps=200, height=64, width=96, seeds=(2, 3, 4, 5)):
    """PSNR at a fixed step, HIP path vs the CPU restatement of the reference kernels (the C
    oracle behind the same host code), same init / batches / entropy noise. A reduced image so that the scalar CPU leg
    stays within seconds; both values are REPORTED per seed (level = mean over the last 20 steps; rounding makes single
    steps chaotic in their last bits) and the 0.05 dB bar of SURVEY 8d is applied to the mean difference over the seeds
    (`delta_flag` = 1 when it is exceeded). Part of the cpu_baseline leg: the only place the product code runs on the oracle."""
    import numpy as _np
    from oracle import hashgrid_c as oc
    from shacira_amd import harness, hip_ops
    torch.set_num_threads(min(os.cpu_count() or 1, 8))   # tiny tensors: many threads only oversubscribe (measured 30x)
    saved = (hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_backward)

    def fwd(coords, codebook, first_idx, resolution, bw):
        return torch.from_numpy(oc.forward(coords.detach().numpy(), codebook.detach().numpy(), first_idx.numpy(),
                                           list(resolution), bw))

    def bwd(dim, coords, grad_output, table_rows, table_dtype, first_idx, resolution, bw, feature_dim, **kw):
        g = oc.backward(coords.detach().numpy(), grad_output.detach().numpy(), (table_rows, feature_dim),
                        first_idx.numpy(), list(resolution), bw)
        return torch.from_numpy(g.astype(_np.float32))
    tail = lambda r: float(_np.mean([h[2] for h in r["history"][-20:]]))
    t0 = time.perf_counter()
    per_seed = []
    for seed in seeds:
        gpu = harness.fit_image(device, steps=steps, height=height, width=width, seed=seed, log_every=1)
        try:
            hip_ops.hashgrid_interpolate_cuda = hip_ops.hashgrid_interpolate2d_cuda = fwd
            hip_ops.hashgrid_backward = bwd
            cpu = harness.fit_image(torch.device("cpu"), steps=steps, height=height, width=width, seed=seed, log_every=1)
        finally:
            hip_ops.hashgrid_interpolate_cuda, hip_ops.hashgrid_interpolate2d_cuda, hip_ops.hashgrid_backward = saved
        per_seed.append({"seed": seed, "gpu_db": tail(gpu), "cpu_restatement_db": tail(cpu),
                         "delta_db": tail(gpu) - tail(cpu), "bpp_gpu": gpu["bpp"], "bpp_cpu_restatement": cpu["bpp"]})
    mean_delta = float(_np.mean([p["delta_db"] for p in per_seed]))
    return {"config": f"config-B LatentGrid image fit, {height}x{width} procedural image, {steps} steps, seeds {list(seeds)}",
            "gpu_db": float(_np.mean([p["gpu_db"] for p in per_seed])),
            "cpu_restatement_db": float(_np.mean([p["cpu_restatement_db"] for p in per_seed])),
            "delta_db": mean_delta, "bar_db": 0.05, "delta_flag": int(abs(mean_delta) > 0.05), "per_seed": per_seed,
            "cpu_seconds": time.perf_counter() - t0}


def ranks_proof(rank, world, device):
    """What lets a reader of the JSON line see that the collective backend really joined `world` ranks:
    a ones tensor summed over the ranks, every rank's device ordinal gathered, the backend's name and version. Raises
    SystemExit(3) on every rank when the sum is not `world`, so a launch that silently ran N independent replicas -- or lost
    a rank -- cannot produce a scaling figure."""
    info = {"ranks_seen": 1, "backend": None, "nccl_version": None, "rank_devices": None, "hostname_count": 1}
    if device.type == "cuda":
        info["rank_devices"] = [int(torch.cuda.current_device())]
    if world > 1 or dist.is_initialized():
        ones = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(ones)
        info["ranks_seen"] = int(round(float(ones.item())))
        info["backend"] = dist.get_backend()
        dev_id = torch.tensor([torch.cuda.current_device() if device.type == "cuda" else -1], dtype=torch.int64, device=device)
        gathered = [torch.zeros_like(dev_id) for _ in range(world)]
        dist.all_gather(gathered, dev_id)
        info["rank_devices"] = [int(g.item()) for g in gathered]
        if device.type == "cuda" and info["backend"] == "nccl":
            try:
                info["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as exc:   # noqa: BLE001 -- a reported extra
                info["nccl_version"] = f"unavailable ({type(exc).__name__})"
        if info["ranks_seen"] != world:
            if rank == 0:
                print(f"bench.py: the all-reduce of ones saw {info['ranks_seen']} ranks, expected {world}", file=sys.stderr)
            raise SystemExit(3)
    return info


def roofline_record(ms_fwd, ms_bwd, b_fwd, b_bwd, n_local, traffic_fwd=None, traffic_bwd=None, traffic_source=None,
                    kernels=None):
    """The `roofline` object of the bench line. Top level = the BASELINE metric itself, hash-grid fwd+bwd: ALGORITHMIC bytes
    of both operators (2328 B per sample in 3-D, L16, F2) over their summed HIP-event times, against the
    8 TB/s HBM roof. `operators` nests the same figures per operator (what the contract calls the dominant kernel is
    `operators[dominant]`), `hbm_utilisation` = counter-measured HBM bytes of both operators over the same time over the peak
    (how busy the memory system is, on whatever bytes), `traffic_source` says where the counter bytes come from. Every value is
    None in the launch self-test (no kernels run there); the keys are the same."""
    def op(ms, b, traffic):
        if ms is None:
            return {"achieved": None, "frac": None, "ms_per_launch": None, "algorithmic_bytes_per_launch": None,
                    "traffic": None}
        ach = b * n_local / (ms * 1e-3) / 1e9
        return {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "ms_per_launch": ms, "algorithmic_bytes_per_launch": b * n_local,
                "traffic": traffic}
    have = ms_fwd is not None and ms_bwd is not None
    path = (b_fwd + b_bwd) * n_local / ((ms_fwd + ms_bwd) * 1e-3) / 1e9 if have else None
    traffic = (traffic_fwd + traffic_bwd) if (traffic_fwd is not None and traffic_bwd is not None) else None
    return {"bound": "hbm", "what": "hash-grid forward + backward operators (the BASELINE metric), algorithmic bytes / HIP-event time",
            "kernel": kernels, "achieved": path, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (path / HBM_PEAK_GBS) if have else None,
            "bytes_per_sample": b_fwd + b_bwd, "traffic": traffic, "traffic_source": traffic_source,
            "hbm_utilisation": (traffic / ((ms_fwd + ms_bwd) * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic is not None and have) else None,
            "dominant": (("backward" if ms_bwd >= ms_fwd else "forward") if have else None),
            "operators": {"forward": op(ms_fwd, b_fwd, traffic_fwd), "backward": op(ms_bwd, b_bwd, traffic_bwd)}}


def build_step(device, rank, world, dim, res, bw, F, n_local, ar_chunks=1, collective="allreduce", use_plan=True):
    """The benchmark step on this rank: synthetic inputs resident on `device` (parameters replicated: same seed; samples
    per rank), forward operator, backward operator into the communication buffer, gradient reduction. Used by main() and,
    on CPU with the oracle standing in for the operators, by tests/test_dist_cpu.py (world-size-4 gloo run of the same loop)."""
    from shacira_amd import _lib
    from shacira_amd import dist as sdist
    from shacira_amd import hip_ops
    L = len(res)
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    T = int(sum(sizes))
    gp = torch.Generator().manual_seed(0)
    table = (torch.randn(T, F, generator=gp) * 0.01).to(device)            # reference init: randn * feature_std
    gs = torch.Generator().manual_seed(1000 + rank)
    coords = (torch.rand(n_local, dim, generator=gs) * 2 - 1).to(device)   # U(-1,1)^d, full-entropy mantissas
    grad_out = torch.randn(n_local, L * F, generator=gs).to(device)
    first = torch.from_numpy(first_np).to(device)
    fwd = hip_ops.hashgrid_interpolate_cuda if dim == 3 else hip_ops.hashgrid_interpolate2d_cuda

    # N > 1: the codebook gradient is still reduced exactly once per step, but in `chunks` row ranges: the backward
    # is issued per level group and each group's (contiguous) rows start their all-reduce on RCCL's stream while
    # the next group is computed (xGMI transfer hidden behind compute). N = 1: one call, no collective.
    groups = sdist.level_groups(L, ar_chunks)
    row_of = lambda l: int(first_np[l]) if l < L else T
    reducer = sdist.GradientReducer(collective)
    # the gradient buffer: [T, F] view of a flat tensor padded to a multiple of the world size (rs_ag shards it evenly)
    grad_flat = torch.zeros(sdist.GradientReducer.padded_numel(T * F, world), dtype=table.dtype, device=device)
    grad_buf = grad_flat[:T * F].view(T, F)
    ws_shared = (hip_ops.backward_workspace(dim, n_local, T, table.dtype, res, bw, F, device)
                 if (len(groups) > 1 or world > 1) else None)
    # the batch's plan: written by every forward call of the timed loop (never reused across steps here), read by the
    # backward of the same step. None for shapes whose forward sorts nothing (and on CPU, where the oracle stands in).
    plan = (hip_ops.hashgrid_plan_buffer(dim, coords, table, res, bw)
            if device.type == "cuda" and len(groups) == 1 and use_plan else None)
    pkw = {} if plan is None else {"plan": plan}

    def step(ev=None):
        if ev:
            ev[0].record()
        feats = fwd(coords, table, first, res, bw, **pkw)
        if ev:
            ev[1].record()
        if len(groups) == 1 and world == 1:
            grad = hip_ops.hashgrid_backward(dim, coords, grad_out, T, table.dtype, first, res, bw, F, **pkw)
            if ev:
                ev[2].record()
        else:
            # N > 1: the backward writes straight into the (padded) communication buffer; one collective per step, or one
            # per level group overlapped with the next group's backward (shacira_amd.dist.backward_in_groups)
            grad = grad_buf

            def backward_levels(lb, le, first_group):
                if len(groups) == 1:
                    hip_ops.hashgrid_backward(dim, coords, grad_out, T, table.dtype, first, res, bw, F, out=grad,
                                              workspace=ws_shared, **pkw)
                else:
                    hip_ops.hashgrid_backward(dim, coords, grad_out, T, table.dtype, first, res, bw, F, levels=(lb, le),
                                              out=grad, workspace=ws_shared,
                                              flags=_lib.BWD_STAGE_ALL_LEVELS if first_group else _lib.BWD_REUSE_STAGED)
                if ev and le == L:
                    ev[2].record()
            sdist.backward_in_groups(backward_levels, grad_flat if len(groups) == 1 else grad, groups, row_of, reducer)
        if ev:
            ev[3].record()
        return feats, grad

    return {"step": step, "groups": groups, "first_np": first_np, "T": T, "table": table, "coords": coords,
            "grad_out": grad_out, "first": first, "plan_bytes": 0 if plan is None else plan.numel(),
            "backward_workspace_bytes": (hip_ops.hashgrid_backward_workspace_bytes(dim, n_local, T, table.dtype, res, bw, F,
                                                                                   planned=plan is not None)
                                         if device.type == "cuda" else None)}


SWEEP_CASES = [
    # (label, collective, ar_chunks, NCCL_ALGO or None)
    ("allreduce", "allreduce", 1, None),
    ("allreduce_ring", "allreduce", 1, "Ring"),
    ("allreduce_tree", "allreduce", 1, "Tree"),
    ("allreduce_chunks3", "allreduce", 3, None),
    ("rs_ag", "rs_ag", 1, None),
]
SWEEP_LOADS = [("weak_S1", "S1_nerf_hash_3d_L16_F2_bw19_N2^20", "weak"),
               ("strong_D", "D_nerf_lego_3d_L16_F2_bw19_N65536", "strong")]


def run_sweep(args):
    """bench.py --gpus N --sweep: every (collective setting) x (workload, scaling) pair as its own `bench.py --gpus N ...`
    child (which in turn spawns its ranks as children): no process that has touched HIP is ever re-executed, and a hung or
    failed configuration costs its own timeout only. One JSON line per configuration as it finishes + a summary line."""
    import subprocess
    if "RANK" in os.environ:
        raise SystemExit("bench.py --sweep is a launcher mode: start it plainly (python bench.py --gpus N --sweep)")
    assert not torch.cuda.is_initialized(), "the sweep parent must not initialise the GPU"
    results = []
    only = set(filter(None, args.sweep_filter.split(",")))
    known = {f"{load}/{label}" for load, _, _ in SWEEP_LOADS for label, _, _, _ in SWEEP_CASES}
    if only - known:
        raise SystemExit(f"bench.py --sweep-filter: unknown configuration(s) {sorted(only - known)}; known: {sorted(known)}")
    for load, workload, scaling in SWEEP_LOADS:
        for label, collective, chunks, algo in SWEEP_CASES:
            if args.gpus == 1 and label != "allreduce":
                continue                      # one GPU: no collective to vary
            if only and f"{load}/{label}" not in only:
                continue
            env = dict(os.environ)
            env.pop("NCCL_ALGO", None)
            if algo:
                env["NCCL_ALGO"] = algo
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--workload", workload, "--scaling",
                   scaling, "--collective", collective, "--ar-chunks", str(chunks), "--steps", str(args.steps), "--warmup",
                   str(args.warmup), "--no-cpu-baseline", "--psnr-steps", "0", "--nerf-steps", "0", "--no-secondary"]
            if args.selftest_launch:
                cmd.append("--selftest-launch")
            name = f"{load}/{label}"
            try:
                cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                rec = json.loads(lines[-1]) if lines else {"error": f"no JSON line (rc {cp.returncode})",
                                                           "stderr_tail": cp.stderr[-400:]}
            except subprocess.TimeoutExpired:
                rec = {"error": "timeout (600 s)"}
            rec.setdefault("config", {})
            if isinstance(rec["config"], dict):
                rec["config"].update({"sweep": name, "nccl_algo": algo or "unset (RCCL chooses)", "ar_chunks": chunks,
                                      "collective": collective if args.gpus > 1 else None})
            print(json.dumps(rec), flush=True)
            results.append((name, rec))
    ok = [(n, r) for n, r in results if r.get("value") is not None]
    summary = {"sweep_summary": {n: {"value": r.get("value"), "ms_per_step": r.get("ms_per_step"),
                                     "ms_allreduce": (r.get("ms") or {}).get("allreduce"), "error": r.get("error")}
                                 for n, r in results},
               "n_gpus": args.gpus, "unit": "samples/s (whole job)",
               "best": {load: max(((n, r["value"]) for n, r in ok if n.startswith(load)), key=lambda t: t[1], default=None)
                        for load, _, _ in SWEEP_LOADS}}
    print(json.dumps(summary), flush=True)
    return 0 if all("error" not in r for _, r in results) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # SURVEY 8(d) protocol: 20 warm-up + 100 timed iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): the workload's samples PER GPU; strong: the same total "
                         "split over the ranks")
    ap.add_argument("--workload", default="S1_nerf_hash_3d_L16_F2_bw19_N2^20", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ar-chunks", type=int, default=1,
                    help="N > 1: build the gradient in this many level groups and all-reduce each group's rows "
                         "asynchronously while the next group is computed. Default 1: measured on one GPU the split "
                         "costs ~0.11 ms per extra group, about what it can hide at 8 GPUs; kept opt-in until it "
                         "has been measured on a multi-GPU node")
    ap.add_argument("--collective", choices=["allreduce", "rs_ag"], default="allreduce",
                    help="N > 1: how the codebook gradient is summed: one all_reduce (RCCL chooses the algorithm; steer it "
                         "with NCCL_ALGO=Ring|Tree for the comparison SURVEY.md section 5 asks for), or reduce_scatter + "
                         "all_gather on the flat buffer (every xGMI link carries 1/world of each phase)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the quick figures for the other BASELINE configs")
    ap.add_argument("--no-plan", action="store_true",
                    help="A/B switch: call the plain operators (no plan hand-off from the forward to the backward)")
    ap.add_argument("--psnr-steps", type=int, default=1000, help="image-fit steps for the PSNR figure (0 = skip)")
    ap.add_argument("--nerf-steps", type=int, default=500,
                    help="steps of the NeRF-style render-and-fit on the analytic scene for the second PSNR figure (0 = skip)")
    ap.add_argument("--cpu-samples", type=int, default=1 << 17,
                    help="samples of the CPU baseline leg (a bounded slice of the headline batch)")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--sweep", action="store_true",
                    help="the first node-hour in one command: with --gpus N, run {allreduce (NCCL_ALGO unset / Ring / Tree), "
                         "rs_ag} x {--ar-chunks 1, 3} x {weak S1, strong D (config E: 65 536 samples over the ranks)} -- each "
                         "in fresh child processes (this parent never touches HIP) -- and print one JSON line per "
                         "configuration (config.sweep names it, config.nccl_algo records the RCCL algorithm setting), then "
                         "a summary line. Combine with --selftest-launch to check the control flow without GPUs.")
    ap.add_argument("--sweep-filter", default="",
                    help="comma-separated subset of the sweep's configurations (names as in config.sweep, e.g. "
                         "weak_S1/rs_ag,strong_D/allreduce_chunks3); default: all")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="only exercise the multi-rank launch (spawn, rendezvous, one all-reduce, JSON line with n_gpus); "
                         "needs no GPU: ranks use the gloo backend. The metric value is null.")
    args = ap.parse_args()

    # --gpus N launched plainly (no torchrun environment): start the N rank processes ourselves, BEFORE anything in this
    # process touches the GPU (a process that has initialised HIP must never be replaced or forked into ranks); this
    # parent only waits and relays the children's output (rank 0 prints the JSON line) and exit code.
    if args.ar_chunks > 1 and args.collective != "allreduce":
        raise SystemExit("bench.py: --ar-chunks > 1 overlaps all-reduces of row ranges; it needs --collective allreduce")
    if args.sweep:
        raise SystemExit(run_sweep(args))
    if args.gpus > 1 and "RANK" not in os.environ:
        import socket
        import subprocess
        from shacira_amd import dist as sdist_launch
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # The parent never touches the HIP runtime: GPUs are counted from the environment / sysfs (torch's device_count()
        # can fall through to hipGetDeviceCount), and the ranks are fresh CHILD processes -- never an exec of this one.
        visible = 0 if args.selftest_launch else sdist_launch.visible_gpu_count()
        assert not torch.cuda.is_initialized(), "the launcher parent must not initialise the GPU before it spawns the ranks"
        if not args.selftest_launch and visible == 1:
            # one visible GPU: all ranks share it over gloo -- exercises the control flow only, not a scaling figure
            print(f"bench.py: --gpus {args.gpus} with 1 visible GPU: ranks share cuda:0 over gloo "
                  "(SHACIRA_TEST_SINGLE_GPU=1); control-flow check only", file=sys.stderr)
            env["SHACIRA_TEST_SINGLE_GPU"] = "1"
        elif not args.selftest_launch and 1 < visible < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {visible} GPUs are visible")
        # (visible == 0: nothing could be counted without the runtime -- let the ranks fail loudly if there is no GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    if args.selftest_launch:
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        proof = ranks_proof(rank, world, torch.device("cpu"))
        if rank == 0:
            print(json.dumps({"metric": "hash-grid samples/sec fwd+bwd (16 lvl, F=2)", "value": None, "unit": "samples/s",
                              "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True,
                              "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "none",
                              "config": {"workload": "launch self-test (gloo, no kernels)", "scaling": args.scaling,
                                         "allreduce_check": float(t.item()) == world * (world + 1) / 2, **proof},
                              "roofline": roofline_record(None, None, *algorithmic_bytes_per_sample(3, 16, 2), 0),
                              "cpu_baseline": None}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from shacira_amd import _lib
    from shacira_amd import dist as sdist
    from shacira_amd import hip_ops
    rank, world, device = sdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: using the launcher's world size", file=sys.stderr)
        args.gpus = world
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    proof = ranks_proof(rank, world, device)   # exits non-zero unless the collective backend joined exactly `world` ranks

    dim, res, bw, F, n_local = WORKLOADS[args.workload]
    if args.scaling == "strong":
        lo, hi = sdist.shard_bounds(n_local, rank, world)
        n_local = hi - lo
    L = len(res)
    st = build_step(device, rank, world, dim, res, bw, F, n_local, args.ar_chunks, args.collective, use_plan=not args.no_plan)
    step, groups, first_np, T, table, coords, grad_out = (st["step"], st["groups"], st["first_np"], st["T"], st["table"],
                                                          st["coords"], st["grad_out"])
    st_plan_bytes, st_ws_bytes = st["plan_bytes"], st["backward_workspace_bytes"]

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # host-side preparation first (event objects, a full pass of the cyclic collector): between the warm-up and the timed region
    # nothing but the fence, so the GPU enters the region in the state the warm-up left it in (tens of ms of idling there let
    # its clocks fall back: the first timed steps then ran 3-5 % slow, 0.766 vs 0.754 ms/step between 50- and 100-step runs)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    # the timed region is ~0.1 s of asynchronous launches: a generation-2 pass of the cyclic collector over torch's heap
    # in the middle of it stalls the launch thread for tens of ms (seen as 1.06 vs 0.99 ms/step between two runs whose
    # HIP-event times agreed to 2 %), so it is run now and paused until the region ends
    import gc
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # PSNR at a fixed step (second half of BASELINE.json's metric): config-B LatentGrid (16-level 2-D, F=2, latent
    # quantisation + entropy model on) fitted to a 512x768 procedural image, pixels sharded over the ranks with one
    # gradient all-reduce per step; OUTSIDE the timed throughput region.
    psnr = None
    if args.psnr_steps > 0:
        from shacira_amd import harness
        torch.cuda.synchronize()
        tp = time.perf_counter()
        fit = harness.fit_image(device, steps=args.psnr_steps, rank=rank, world=world)
        torch.cuda.synchronize()
        psnr = {"value": fit["psnr"], "unit": "dB (clamped_psnr)", "step": args.psnr_steps,
                "config": "B: 2-D LatentGrid L16 F2 ld1 bw11 res16..512, quant+entropy on, 512x768 procedural image, "
                          "Adam (kodak.yaml learning rates)", "bpp": fit["bpp"], "bpp_file": fit["bpp_file"],
                "file_bytes": fit["file_bytes"], "rgb_loss": fit["rgb_loss"],
                "seconds": time.perf_counter() - tp, "ms_per_step": fit["ms_per_step"], "n_gpus": world,
                "mode": "eager step, cold (Python issues ~60 launches per step; warmed up it takes ~1 ms with the GPU busy "
                        "a third of it: profiles/r04_imagefit_step.md)"}
        if world == 1:
            # the same fit with the step captured once into a HIP graph and replayed (GraphedImageFitter: device-side
            # entropy noise and Adam step count): what a user who cares about wall time runs
            tg = time.perf_counter()
            gfit = harness.fit_image(device, steps=args.psnr_steps, graphed=True)
            torch.cuda.synchronize()
            psnr["graph_replay"] = {"value": gfit["psnr"], "bpp": gfit["bpp"], "ms_per_step": gfit["ms_per_step"],
                                    "seconds": time.perf_counter() - tg}

    # second PSNR figure (rank 0 only, outside the timed region): the reference's per-step NeRF pipeline -- ray marching
    # on the occupancy grid, hash-grid lookup, MFMA decoders, volume integration, L1, fused Adam -- on a closed-form scene
    psnr_nerf = None
    if rank == 0 and args.nerf_steps > 0:
        from shacira_amd import harness
        torch.cuda.synchronize()
        tp = time.perf_counter()
        fit = harness.fit_nerf(device, steps=args.nerf_steps)
        psnr_nerf = {"value": fit["psnr"], "unit": "dB (psnr, held-out rays)", "step": args.nerf_steps,
                     "config": "D: 3-D HashGrid L16 F2 bw19 res16..2048, 4096 rays x 128 candidate samples ('ray' "
                               "marcher, occupancy level 5 pruned every 100 steps), density 32-64-16 + colour 43-64-64-3 "
                               "decoders, analytic scene", "ms_per_step": fit["ms_per_step"],
                     "occupied_cells": fit["occupied_cells"], "seconds": time.perf_counter() - tp}
        # the same fit fed from a resident pool of rays + target colours (the role of the reference's MultiviewDataset: the
        # figure above renders its targets from the closed-form scene inside every step), eagerly and with the step replayed
        # from HIP graphs (harness.GraphedNerfFitter: capacity-sized sample buffers, no count read-back, re-captured when a
        # prune moves the sample count): what a user who cares about wall time runs
        tg = time.perf_counter()
        efit = harness.fit_nerf(device, steps=args.nerf_steps, ray_pool=128)
        gfit = harness.fit_nerf(device, steps=args.nerf_steps, ray_pool=128, graphed=True)
        psnr_nerf["ray_pool_128_batches"] = {
            "eager": {"value": efit["psnr"], "ms_per_step": efit["ms_per_step"]},
            "graph_replay": {"value": gfit["psnr"], "ms_per_step": gfit["ms_per_step"],
                             "graph_captures": gfit["graph_captures"], "sample_capacity_last": gfit["sample_capacity"],
                             "overflow_steps": gfit["overflow_steps"], "capture_seconds": gfit["capture_seconds"]},
            "seconds": time.perf_counter() - tg}
        # the compressed variant the reference's nerf_lego.yaml trains (3-D LatentGrid: SGA warm-up, entropy model, lambda 1e-4)
        # on the same ray pool, eager and graph-replayed (the SGA temperature annealed through a device float between replays)
        tc = time.perf_counter()
        ce = harness.fit_nerf(device, steps=args.nerf_steps, ray_pool=128, latent=True)
        cg = harness.fit_nerf(device, steps=args.nerf_steps, ray_pool=128, latent=True, graphed=True)
        psnr_nerf["compressed_latent_grid_ray_pool_128"] = {
            "eager": {"value": ce["psnr"], "ms_per_step": ce["ms_per_step"], "file_bytes": ce["file_bytes"]},
            "graph_replay": {"value": cg["psnr"], "ms_per_step": cg["ms_per_step"], "file_bytes": cg["file_bytes"],
                             "graph_captures": cg["graph_captures"], "overflow_steps": cg["overflow_steps"]},
            "table_bytes_fp32": ce["table_bytes_fp32"], "seconds": time.perf_counter() - tc}

    # the optimiser pass that follows the backward in training (SURVEY 8d "second figure"): fused Adam over the table
    ms_adam = None
    if rank == 0:
        from shacira_amd.optim import FusedAdam
        ptab = torch.nn.Parameter(table.clone())
        ptab.grad = torch.randn_like(ptab)
        opt = FusedAdam([ptab], lr=1e-2, eps=1e-15)
        for _ in range(3):
            opt.step()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(20):
            opt.step()
        eb.record()
        torch.cuda.synchronize()
        ms_adam = ea.elapsed_time(eb) / 20
        del ptab, opt

    secondary = None
    if rank == 0 and not args.no_secondary:
        st.clear()
        del coords, grad_out, step
        torch.cuda.empty_cache()
        secondary = {name: quick_measure(name, device) for name in SECONDARY if name != args.workload}

    if rank == 0:
        ms_fwd = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
        ms_bwd = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
        ms_ar = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))
        b_fwd, b_bwd = algorithmic_bytes_per_sample(dim, L, F)
        ms_step = elapsed / args.steps * 1e3
        n_total = world * n_local if args.scaling == "weak" else WORKLOADS[args.workload][4]
        value = n_total * args.steps / elapsed
        per_step = np.array([e[0].elapsed_time(e[2]) for e in events])
        dom = ("backward", ms_bwd, b_bwd) if ms_bwd >= ms_fwd else ("forward", ms_fwd, b_fwd)
        # HBM bytes per launch from the PMC counters (collected offline with rocprofv3 --pmc in separate passes on
        # the same operators and workload; see the note inside the file). Only valid for the workload it was taken on.
        # Attached ONLY when the file was measured on exactly these kernel sources (kernel_source_hash); else null.
        traffic_fwd, traffic_bwd, traffic_source = None, None, None
        # (the newest round's file: profiles/rNN_pmc_traffic.json)
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
        tpath = tfiles[-1] if tfiles else os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
        if args.workload.startswith("S1_") and os.path.exists(tpath):
            with open(tpath) as fh:
                rec = json.load(fh)
            if rec.get("kernel_source_hash") == kernel_source_hash():
                traffic_fwd = rec["operators"].get("forward", {}).get("hbm_bytes_per_launch")
                traffic_bwd = rec["operators"].get("backward", {}).get("hbm_bytes_per_launch")
                traffic_source = (f"profiles/{os.path.basename(tpath)}: rocprofv3 --pmc passes of the same operators and workload "
                                  "on the builder's box (same kernel sources by hash), NOT counters of this run; the time it is "
                                  "divided by IS this run's")
            else:
                traffic_source = (f"none: profiles/{os.path.basename(tpath)} was measured on other kernel sources "
                                  f"({rec.get('kernel_source_hash')} != {kernel_source_hash()})")
        traffic = traffic_bwd if dom[0] == "backward" else traffic_fwd
        # what this chip's memory system sustains on plain streams, measured in the same run (SURVEY 8d asks for the
        # nominal AND the measured peak): a 1 GiB device-to-device copy (read + write), a fill (write only) and a
        # reduction (read only), bytes moved / time. The nominal 8 TB/s stays the `peak` of the contract.
        def stream_gbs(fn, nbytes, it=5):
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(it):
                fn()
            b.record()
            torch.cuda.synchronize()
            return nbytes * it / (a.elapsed_time(b) * 1e-3) / 1e9
        src = torch.empty(1 << 28, dtype=torch.float32, device=device).normal_()
        dst = torch.empty_like(src)
        nb = src.numel() * 4
        Lh = _lib.lib()
        strm = torch.cuda.current_stream(device).cuda_stream

        def probe(kind):
            _lib.check(Lh.shacira_stream_probe(kind, src.data_ptr(), dst.data_ptr(), nb, strm), "stream_probe")
        measured = {"copy_read+write": stream_gbs(lambda: probe(2), 2 * nb),
                    "write_only": stream_gbs(lambda: probe(1), nb),
                    "read_only": stream_gbs(lambda: probe(0), nb),
                    "torch_copy_read+write": stream_gbs(lambda: dst.copy_(src), 2 * nb),
                    "unit": "GB/s", "how": "shacira_stream_probe (16 B per lane, 8 in flight, non-temporal: the access shape of "
                                           "the hash-grid kernels) over 1 GiB, HIP events, 5 repetitions; torch copy_ beside it"}
        del src, dst
        if traffic is not None:
            measured["operator_traffic_over_time_GBps"] = traffic / (dom[1] * 1e-3) / 1e9
        # What actually binds the two operators on this access pattern: next to the contract's HBM
        # fraction (algorithmic bytes / 8 TB/s), the floor each operator has on the resource that limits it, and how close
        # the measured time is to that floor.
        #  forward: a hashed level's x-pair of corners is one 128-byte line; with uniformly random samples every pair of a
        #   level finer than the batch's density is its own line, and a CU's L2 -> L1 path moves 64 B/clk: 2 clk per line
        #   (measured: 38 TCP->TCC requests and 81 TA-busy cycles per gather instruction, profiles/r02_fwd_counters.md).
        #  backward: the traffic its kernels move (PMC counters, profiles/rNN_pmc_traffic.json; without them: algorithmic
        #   bytes x the 1.93 measured in round 2) at the copy rate this chip sustains in the same run.
        n_fine = sum(1 for r in res if float(r) ** dim > 4.0 * n_local) if dim == 3 else 0
        lines = n_fine * n_local * (2 ** (dim - 1))
        cus, clk_ghz = 256, 2.4
        fwd_floor_ms = lines * 2.0 / (cus * clk_ghz * 1e9) * 1e3
        bwd_traffic = traffic_bwd
        bwd_bytes = bwd_traffic if bwd_traffic is not None else 1.93 * b_bwd * n_local
        bwd_floor_ms = bwd_bytes / (measured["copy_read+write"] * 1e9) * 1e3
        bound_model = {
            "forward": {"resource": "L2 -> L1 line path of the CUs: 64 B/clk/CU = 2 clk per 128-byte line, 256 CUs at 2.4 GHz",
                        "what": (f"{n_fine} levels with more than 4 cells per sample (no two samples share a table line in any order; "
                                 f"the kernel split is another matter: the rows kernel takes levels up to 32 cells per sample) x "
                                 f"{n_local} samples x {2 ** (dim - 1)} lines (one per x-pair of corners)"),
                        "ms": fwd_floor_ms, "operator_ms": ms_fwd,
                        "frac_of_floor": (fwd_floor_ms / ms_fwd) if ms_fwd > 0 else None},
            "backward": {"resource": "copy rate measured in this run (shacira_stream_probe, read + write)",
                         "what": ("PMC traffic of the backward's kernels" if bwd_traffic is not None else
                                  "algorithmic bytes x 1.93 (round-2 PMC ratio; no counters for these kernel sources)"),
                         "bytes": bwd_bytes, "GBps": measured["copy_read+write"], "ms": bwd_floor_ms, "operator_ms": ms_bwd,
                         "frac_of_floor": (bwd_floor_ms / ms_bwd) if ms_bwd > 0 else None},
        }
        out = {
            "metric": "hash-grid samples/sec fwd+bwd (16 lvl, F=2)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "dim": dim, "levels": L, "feature_dim": F, "bitwidth": bw,
                       "table_rows": T, "samples_per_gpu": n_local, "scaling": args.scaling,
                       "plan_hand_off": ("forward builds the batch's plan every step, backward reads it"
                                         if st_plan_bytes else "none (plain operators)"), "plan_bytes": st_plan_bytes,
                       "backward_workspace_bytes": st_ws_bytes,
                       "collective": args.collective if world > 1 else None,
                       "nccl_algo": (os.environ.get("NCCL_ALGO") or "unset (RCCL chooses)") if world > 1 else None,
                       **proof,
                       "parallelism": f"dp{world}" + ("" if world == 1 else
                                                      (f"+one {args.collective}(grad_codebook) after the backward")
                                                      if len(groups) == 1 else
                                                      f"+allreduce(grad_codebook) in {len(groups)} level groups "
                                                      f"{groups}, overlapped with the backward")},
            "roofline": {**roofline_record(
                             ms_fwd, ms_bwd, b_fwd, b_bwd, n_local, traffic_fwd, traffic_bwd, traffic_source,
                             kernels=("one C-ABI call per operator, HIP events on its stream. forward: psort_count + psort_partition + "
                                      "psort_local (the batch's plan) + hashgrid_fwd_level_pair (fine levels) + hashgrid_fwd_rows; backward: "
                                      "zero_unowned_rows + front16 (gradient rows gathered in plan order + bucket counts) + bin_scan_buckets "
                                      "+ bin_scatter + bin_consume, brick_accumulate (coarse levels) beside the consume pass")),
                         "bound_model": bound_model,
                         "measured_stream_rates": measured},
            "ms": {"forward": ms_fwd, "backward": ms_bwd, "allreduce": ms_ar,
                   "fwd+bwd_p10_p50_p90": [float(np.percentile(per_step, q)) for q in (10, 50, 90)],
                   "adam_table": ms_adam,
                   "samples_per_s_with_adam": (n_local / ((ms_fwd + ms_bwd + ms_ar + ms_adam) * 1e-3)
                                               if ms_adam is not None else None)},
            "psnr": psnr,
            "psnr_nerf": psnr_nerf,
            "other_configs_1gpu": secondary,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dim, res, bw, F, first_np, T, min(args.cpu_samples, n_local),
                                               args.cpu_budget_s)
            try:   # a reported extra: whatever goes wrong with the prebuilt reference library must not cost the bench line
                out["cpu_baseline"]["reference_kernels_on_this_gpu"] = reference_kernels_on_this_gpu(
                    device, dim, res, bw, F, first_np, T, n_local)
            except Exception as exc:   # noqa: BLE001
                out["cpu_baseline"]["reference_kernels_on_this_gpu"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
            if args.psnr_steps > 0:
                out["psnr_parity"] = psnr_parity(device)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
