"""The C restatement (oracle/hashgrid_oracle.c) held to outputs of the REFERENCE'S OWN kernels: tests/golden/ref_kernels.npz was
produced by tests/golden/make_ref_kernel_vectors.py from oracle/_ref/shacira_ref_ops.so (reference
wisp/csrc/ops/hashgrid_interpolate{,2d}_cuda.cu + hashgrid_interpolate.cpp built for gfx950 by oracle/ref_build.py) on an
MI355X. No GPU and no /root/reference needed here: the inputs are pure functions of the case name.

What is shown, case by case (2-D / 3-D, fp32 / fp16 / double tables, F = 2 / 4, edge coordinates):
 * the restatement reproduces the reference kernels' forward BIT FOR BIT once the one thing a compiler is free to choose --
   which products of `t0*c0 + t1*c1 + ...` are fused into fmas -- is set to what hipcc chose for that build (mode 1; fp16:
   `forward_half_llvm`). Transform, clamp, floor, hash, dense test, corner order, weights and layout are thereby pinned.
 * in its default mode (nvcc's contraction, what the product is held to bit for bit) it differs from those outputs by at
   most 1 ulp of the result's magnitude class (measured 3.7e-9 abs on |values| <= 0.06), far inside the 1e-5 bar.
 * the fp64-accumulated backward agrees with the reference's atomics to within their own run-to-run spread.
"""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_ref_kernel_vectors as mk  # noqa: E402
from oracle import hashgrid_c as oc  # noqa: E402

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kernels.npz")
Z = np.load(PATH)
META = json.loads(bytes(Z["meta"].tolist()).decode())
CASES = list(META["cases"])


def test_vectors_cover_every_case_of_the_generator():
    assert set(CASES) == set(mk.CASES)
    assert "shacira_ref_ops.so" in META["producer"]


@pytest.fixture(params=CASES)
def case(request):
    name = request.param
    c = mk.case_inputs(name)
    assert np.array_equal(c["coords"], Z[name + "/coords"]), "inputs must regenerate exactly"
    return name, c


def test_forward_bit_identical_in_the_build_s_contraction_order(case):
    name, c = case
    ref = Z[name + "/feats"]
    if c["dtype"] == "f16":
        got = oc.forward_half_llvm(c["coords"], c["table"], c["first"], c["res"], c["bw"])
    else:
        old = oc.set_contraction(1)
        try:
            got = oc.forward(c["coords"], c["table"].astype(np.float32), c["first"], c["res"], c["bw"])
        finally:
            oc.set_contraction(old)
        got = got.astype(ref.dtype)  # double tables: fp32 arithmetic on narrowed values, widened (.cu:96-107)
    assert got.dtype == ref.dtype and got.shape == ref.shape
    assert np.array_equal(got, ref), f"{name}: {(got != ref).sum()} values differ from the reference kernels"


def test_forward_default_mode_within_one_rounding(case):
    """Default (nvcc-order) restatement vs the hipcc-built reference: only the pairing of the first two products differs."""
    name, c = case
    ref = Z[name + "/feats"].astype(np.float64)
    got = oc.forward(c["coords"], c["table"].astype(np.float32), c["first"], c["res"], c["bw"])
    if c["dtype"] == "f16":
        got = got.astype(np.float16)
        ulp = np.spacing(np.abs(Z[name + "/feats"]).astype(np.float16)).astype(np.float64)
    else:
        ulp = np.spacing(np.abs(got).astype(np.float32)).astype(np.float64)
    # one rounding of a product whose magnitude is at most the sum of |terms| (<= 8 x the largest |table value| = 0.0625)
    bound = np.maximum(ulp, np.spacing(np.float32(0.0625)) if c["dtype"] != "f16" else ulp)
    assert np.all(np.abs(got.astype(np.float64) - ref) <= bound)
    np.testing.assert_allclose(got.astype(np.float64), ref, rtol=1e-5 if c["dtype"] != "f16" else 2e-3, atol=1e-8)


def test_backward_within_the_reference_s_own_atomics_spread(case):
    name, c = case
    if name + "/grad_rows" not in Z:
        pytest.skip("backward vectors exist for fp32 tables only (see make_ref_kernel_vectors.py)")
    info = META["cases"][name]
    ref = np.zeros((c["T"], c["F"]), np.float64)
    ref[Z[name + "/grad_rows"]] = Z[name + "/grad_vals"]
    got = oc.backward(c["coords"], c["grad_out"], (c["T"], c["F"]), c["first"], c["res"], c["bw"])
    # rows: exactly the rows the reference touched with a non-zero sum (a sum that cancels to 0.0 exactly may be missing)
    touched = np.flatnonzero(np.any(got != 0, axis=1))
    assert np.isin(Z[name + "/grad_rows"], touched).all()
    for lo, sz in zip(c["first"], c["sizes"]):
        scale = np.abs(ref[lo:lo + sz]).max()
        np.testing.assert_allclose(got[lo:lo + sz], ref[lo:lo + sz], rtol=1e-5, atol=1e-6 * max(scale, 1e-30))
    # fp32 sample-order accumulation (the reference's arithmetic, one of its possible orders) is as close
    got32 = oc.backward(c["coords"], c["grad_out"], (c["T"], c["F"]), c["first"], c["res"], c["bw"], accumulate="f32")
    assert np.abs(got32 - ref).max() <= max(4 * info["grad_run_to_run_max_abs"], 4e-6 * info["grad_max_abs"])


def test_the_reference_build_leaves_only_a_library_behind():
    """oracle/_ref/ (git-ignored, travels with gpurun) may hold the built library and nothing else: no translated or copied
    reference source text stays in the tree after oracle/ref_build.py has run."""
    ref_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    if not os.path.isdir(ref_dir):
        pytest.skip("oracle/_ref not built here")
    left = sorted(os.listdir(ref_dir))
    assert left == ["shacira_ref_ops.so"], left
