"""Build check on the compiled code objects (no GPU): no hash-grid kernel may use private (scratch) memory. Round 4 found the
F = 4 consume kernels reading a dynamically indexed stack array through `scratch_load_dword` in front of every LDS atomic -- a
2x slowdown that no test or timing of the F = 2 headline could see (profiles/r04_experiments.md section 10)."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "shacira_amd", "lib", "libshacira_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernels outside the hash-grid path that are known to spill (VALU decoder with three hidden layers, the width-128 split
# backward: 512 registers by design); everything else must be scratch-free
ALLOWED = ("mlp_backward_kernel", "split_mlp_backward_kernel")


def _kernels():
    if not os.path.exists(LIB):
        pytest.skip("libshacira_hip.so not built")
    if not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("llvm-readelf not available")
    tmp = tempfile.mkdtemp(prefix="shacira_co_")
    try:
        shutil.copy(LIB, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for co in glob.glob(os.path.join(tmp, "lib.so.*gfx950*")):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                   text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+)(.*?)(?=\.name:\s+_Z|\Z)", notes, re.S):
                body = m.group(2)
                priv = re.search(r"\.private_segment_fixed_size:\s+(\d+)", body)
                vgpr = re.search(r"\.vgpr_count:\s+(\d+)", body)
                if priv:
                    out[m.group(1)] = (int(priv.group(1)), int(vgpr.group(1)) if vgpr else -1)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_no_hashgrid_kernel_uses_scratch_memory():
    ks = _kernels()
    names = " ".join(ks)
    for must in ("bin_consume_kernel", "bin_scatter_kernel", "front16_kernel", "direct_accumulate_kernel",
                 "hashgrid_fwd_level_pair_kernel", "hashgrid_fwd_rows_kernel", "hashgrid_fwd_lds_kernel"):
        assert must in names, f"{must} not found in the code objects"
    bad = {k: v for k, v in ks.items() if v[0] > 0 and not any(a in k for a in ALLOWED)}
    assert not bad, f"kernels with a private segment (bytes, vgprs): {bad}"
