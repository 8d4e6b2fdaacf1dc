"""Produces tests/golden/ref_kernels.npz: inputs and outputs of the REFERENCE'S OWN hash-grid operators
(oracle/_ref/shacira_ref_ops.so = reference wisp/csrc/ops/hashgrid_interpolate{,2d}_cuda.cu + hashgrid_interpolate.cpp,
built by oracle/ref_build.py) executed on an MI355X. Arrays only; no reference source text.

Run on the GPU box:   python tests/golden/make_ref_kernel_vectors.py gpurun_out/ref_kernels.npz
then copy the file to tests/golden/. `tests/test_ref_kernel_vectors.py` holds the C restatement (oracle/hashgrid_oracle.c)
to these vectors without a GPU; `tests/test_gpu_parity.py::test_against_the_reference_kernels*` compares the HIP path with
the same operators directly at larger sizes.

Inputs are exactly reproducible anywhere (integer hashes -> fp32, no RNG stream involved): `case_inputs` below is imported
by the tests. Coordinates include the edge cases of tests/test_gpu_parity.py::_problem (+-1, out of range, 1 - 2^-24;
NaN only where the reference's behaviour is defined, see `case_inputs`).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


# name: (dim, resolutions, bitwidth, F, N, table dtype)
CASES = {
    "A_2d_L8_bw11": (2, geo(16, 512, 8), 11, 2, 2048, "f32"),
    "B_2d_L16_bw11": (2, geo(16, 512, 16), 11, 2, 2048, "f32"),
    "Bp_2d_L16_bw19": (2, geo(16, 2048, 16), 19, 2, 1024, "f32"),
    "D_3d_L16_bw19": (3, geo(16, 2048, 16), 19, 2, 1024, "f32"),
    "D_3d_L16_bw19_f16": (3, geo(16, 2048, 16), 19, 2, 1024, "f16"),
    "lego_3d_L8_F4_bw14": (3, geo(16, 512, 8), 14, 4, 1024, "f32"),
    "lego_3d_L8_F4_bw14_f16": (3, geo(16, 512, 8), 14, 4, 1024, "f16"),
    "kodak_2d_L12_F4_bw11_f16": (2, geo(16, 512, 12), 11, 4, 2048, "f16"),
    "kodak_2d_L16_F2_bw11_f16": (2, geo(16, 512, 16), 11, 2, 2048, "f16"),
    "small_3d_L4_bw9": (3, [4, 7, 12, 33], 9, 2, 512, "f32"),
    "D_3d_L16_bw19_f64": (3, geo(16, 2048, 16), 19, 2, 512, "f64"),
}


def _mix(a):
    """splitmix-style 32-bit integer hash on uint64 arrays (exact everywhere)."""
    a = (a * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    a ^= a >> np.uint64(29)
    a = (a * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    a ^= a >> np.uint64(32)
    return (a & np.uint64(0xFFFFFF)).astype(np.float64)  # 24 bits -> exact in fp32


def unit(n, salt):
    """n values in [-1, 1) on a 2^-23 lattice (fp32-exact), a pure function of (index, salt)."""
    with np.errstate(over="ignore"):
        h = _mix(np.arange(n, dtype=np.uint64) + np.uint64(salt) * np.uint64(0x1000003))
    return (h / 2.0 ** 23 - 1.0).astype(np.float32)


def table_layout(res, bw, dim):
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    return sizes, first, int(sum(sizes))


def case_inputs(name):
    dim, res, bw, F, N, dt = CASES[name]
    sizes, first, T = table_layout(res, bw, dim)
    salt = sum(ord(c) for c in name)
    coords = unit(N * dim, salt + 1).reshape(N, dim)
    # edge coordinates (same set as tests/test_gpu_parity.py::_problem, minus NaN: `max(a, min(b, NaN))` gives the
    # clamp's lower bound on both compilers, but keep the vectors to inputs whose reference behaviour is unarguable)
    coords[0] = 1.0
    coords[1] = -1.0
    coords[3] = 2.5
    coords[4] = -9.0
    coords[5] = np.float32(1.0) - np.float32(2.0 ** -24)
    coords[6] = np.float32(-1.0) + np.float32(2.0 ** -24)
    coords[7, 0] = 1.0
    table = (unit(T * F, salt + 2) * np.float32(0.0625)).reshape(T, F)
    go = unit(N * len(res) * F, salt + 3).reshape(N, len(res) * F)
    npdt = {"f32": np.float32, "f16": np.float16, "f64": np.float64}[dt]
    return dict(dim=dim, res=res, bw=bw, F=F, N=N, dtype=dt, sizes=sizes, first=first, T=T, coords=coords,
                table=table.astype(npdt), grad_out=go.astype(npdt))


def main(out_path):
    import torch

    from oracle import ref_build
    ref = ref_build.load()
    dev = torch.device("cuda:0")
    tdt = {"f32": torch.float32, "f16": torch.float16, "f64": torch.float64}
    out = {}
    meta = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0),
            "producer": "oracle/_ref/shacira_ref_ops.so (reference kernels via torch hipify + hipcc -O3, gfx950)",
            "cases": {}}
    for name in CASES:
        c = case_inputs(name)
        tc = torch.from_numpy(c["coords"]).to(dev)
        tt = torch.from_numpy(c["table"]).to(dev)
        tg = torch.from_numpy(c["grad_out"]).to(dev)
        tf = torch.from_numpy(c["first"]).to(dev)
        assert tt.dtype == tdt[c["dtype"]]
        three_d = c["dim"] == 3
        fwd = ref.hashgrid_interpolate_cuda if three_d else ref.hashgrid_interpolate2d_cuda
        bwd = ref.hashgrid_interpolate_backward_cuda if three_d else ref.hashgrid_interpolate2d_backward_cuda
        feats = fwd(tc, tt, tf, c["res"], c["bw"]).cpu().numpy()
        out[f"{name}/coords"] = c["coords"]
        out[f"{name}/feats"] = feats
        info = dict(dim=c["dim"], res=c["res"], bw=c["bw"], F=c["F"], N=c["N"], dtype=c["dtype"], T=c["T"])
        if c["dtype"] != "f32":
            # backward vectors exist for fp32 tables only: the reference's `__half2` atomics are compiled under
            # `#if defined(__CUDA_ARCH__) && __CUDA_ARCH__ >= 600` (.cu:198), which no HIP build defines, and what is
            # left for Half / double punning the table to float* (.cu:218) is the reference's pre-sm_60 fallback,
            # not its behaviour on the hardware it was written for.
            meta["cases"][name] = info
            print(name, "feats", feats.shape, feats.dtype, flush=True)
            continue
        grads = [bwd(tc, tg, tt, tf, c["res"], c["bw"], c["F"], False) for _ in range(3)]
        torch.cuda.synchronize()
        g = grads[0].cpu().numpy()
        # run-to-run spread of the reference's own atomics (what "equal to the reference" can mean for the gradient)
        spread = max(float((grads[0].double() - gi.double()).abs().max()) for gi in grads[1:])
        rows = np.flatnonzero(np.any(g != 0, axis=1)).astype(np.int32)
        out[f"{name}/grad_rows"] = rows
        out[f"{name}/grad_vals"] = g[rows]
        meta["cases"][name] = dict(info, grad_nonzero_rows=int(rows.size), grad_run_to_run_max_abs=spread,
                                   grad_max_abs=float(np.abs(g.astype(np.float64)).max()))
        print(name, "feats", feats.shape, feats.dtype, "grad rows", rows.size, "atomics spread", spread, flush=True)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_kernels.npz"))
