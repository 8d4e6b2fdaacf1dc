"""The C-ABI library loads and exports every symbol include/shacira_hip.h declares; argument validation returns the
documented codes WITHOUT touching a GPU (validation precedes any HIP call). No compute here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from shacira_amd import _lib

HEADER = os.path.join(ROOT, "include", "shacira_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"SHACIRA_API[^;(]*?\b(shacira_\w+)\s*\(", text)))


def test_header_and_binding_list_the_same_symbols():
    syms = declared_symbols()
    assert len(syms) >= 20
    assert sorted(_lib.SIGNATURES.keys()) == syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C shacira_amd/csrc` (or __graft_entry__.build())"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), name
    assert _lib.lib().shacira_abi_version() == 11


def test_argument_validation_codes():
    L = _lib.lib()
    res = (ctypes.c_int32 * 2)(16, 32)
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first or N == 0
    fwd = L.shacira_hashgrid_forward
    assert fwd(4, 0, 2, 2, 8, res, one, 10, one, one, 0, one, None, 0, None) == _lib.EINVAL     # dim
    assert fwd(2, 0, 0, 2, 8, res, one, 10, one, one, 0, one, None, 0, None) == _lib.EINVAL     # num_lods
    assert fwd(2, 0, 33, 2, 8, res, one, 10, one, one, 0, one, None, 0, None) == _lib.EINVAL    # > SHACIRA_MAX_LODS
    assert fwd(2, 0, 2, 3, 8, res, one, 10, one, one, 0, one, None, 0, None) == _lib.EODD       # odd feature dim
    assert fwd(2, 0, 2, 2, 31, res, one, 10, one, one, 0, one, None, 0, None) == _lib.EINVAL    # bitwidth
    assert fwd(2, 0, 2, 2, 8, res, one, 10, one, one, 7, one, None, 0, None) == _lib.EDTYPE     # dtype
    assert fwd(2, -1, 2, 2, 8, res, one, 10, one, one, 0, one, None, 0, None) == _lib.EINVAL    # negative N
    assert fwd(2, 0, 2, 2, 8, res, one, 10, one, one, 0, one, None, 0, None) == 0               # N == 0: nothing to do
    assert fwd(2, 5, 2, 2, 8, res, None, 10, one, one, 0, one, None, 0, None) == _lib.EINVAL    # null pointer
    assert fwd(2, 5, 2, 2, 8, res, one, 10, one, one, 0, one, None, 0, None) == _lib.EWORKSPACE # workspace missing
    assert L.shacira_hashgrid_forward_workspace_bytes(2, 5, 2, 2, 8, res, 10, 0) >= 5 * 2 * 2 * 4
    assert L.shacira_latent_decode_forward(0, 5, 2, one, one, one, None, None, 0.0, one, None) == _lib.EDTYPE
    assert L.shacira_latent_decode_forward(0, 2, 2, one, one, one, None, None, 0.0, one, None) == 0
    assert L.shacira_entropy_bits_forward(0, 2, 5, one, None, one, one, one, 1 << 20, None) == _lib.EINVAL
    assert L.shacira_entropy_bits_forward(0, 2, 2, one, None, one, one, one, 16, None) == _lib.EWORKSPACE
    assert L.shacira_hashgrid_backward_workspace_bytes(3, 1000, 2, 2, 8, res, 100, 1) >= 100 * 2 * 4   # fp16: fp32 image
    assert L.shacira_hashgrid_backward_workspace_bytes(3, 0, 2, 2, 8, res, 100, 0) == 0
    assert b"multiple of 2" in L.shacira_strerror(_lib.EODD)
    assert L.shacira_set_option(b"nope", 1) == _lib.EINVAL
    # pruned in ABI 7 (round 3): the lost code paths' switches are gone, and forward variants 1, 2, 4, 5, 7 are refused
    for name in (b"bwd_rows", b"bwd_groups", b"bwd_direct_side", b"bwd_fuse"):
        assert L.shacira_set_option(name, 1) == _lib.EINVAL
    for v in (1, 2, 4, 5, 7, 10):
        assert L.shacira_set_option(b"fwd_variant", v) == _lib.EINVAL
    assert L.shacira_set_option(b"bin_acc_kib", 96) == _lib.EINVAL
    for v in (0, 3, 6, 8, 9, -1):
        assert L.shacira_set_option(b"fwd_variant", v) == 0 and L.shacira_get_option(b"fwd_variant") == v
    with pytest.raises(Exception, match="multiple of 2"):
        _lib.check(_lib.EODD)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.EINVAL, "x")


def test_latent_mlp_entry_points_validate_their_arguments():
    """shacira_latent_mlp_*: shape limits and NULL operands are refused before anything is launched (no GPU needed)."""
    import ctypes
    lib = _lib.lib()
    arr = lambda *w: (ctypes.c_int32 * len(w))(*w)
    assert lib.shacira_latent_mlp_supported(2, arr(2, 8, 2)) == 1
    assert lib.shacira_latent_mlp_supported(4, arr(1, 16, 16, 16, 4)) == 1
    assert lib.shacira_latent_mlp_supported(5, arr(1, 2, 2, 2, 2, 2)) == 0          # too deep
    assert lib.shacira_latent_mlp_supported(1, arr(2, 17)) == 0                      # too wide
    assert lib.shacira_latent_mlp_supported(1, arr(0, 2)) == 0
    assert lib.shacira_latent_mlp_backward_workspace_bytes(2, arr(2, 8, 2)) == 512 * (2 * 8 + 8 + 8 * 2 + 2) * 8
    einval = -1
    rc = lib.shacira_latent_mlp_forward(10, 5, arr(1, 2, 2, 2, 2, 2), None, None, 1.0, 0, None, None, 0, 0, 0.0, None, None)
    assert rc != 0
    rc = lib.shacira_latent_mlp_forward(10, 1, arr(2, 2), None, None, 1.0, 0, None, None, 0, 0, 0.0, None, None)
    assert rc != 0                                                                   # NULL div / params
    rc = lib.shacira_latent_mlp_forward(10, 1, arr(2, 2), None, None, 1.0, 0, None, None, 9, 0, 0.0, None, None)
    assert rc != 0                                                                   # unknown activation


def test_backward_workspace_is_sized_by_the_selected_path():
    """A table whose levels all fit LDS images (the Kodak tables of configs B / C, kodak.yaml) stages no transposed gradients
    and writes no items: its backward workspace is control words only, whatever the batch (round 3 asked 1.21 GB for config C).
    The NeRF table's workspace is the staged gradients + the item slots its levels can emit."""
    from conftest import CONFIGS, geo, table_layout
    L = _lib.lib()

    def query(dim, res, bw, n, F=2, dtype=_lib.F32):
        _, _, T = table_layout(res, bw, dim)
        arr = (ctypes.c_int32 * len(res))(*res)
        return L.shacira_hashgrid_backward_workspace_bytes(dim, n, len(res), F, bw, arr, T, dtype)

    dim, res, bw = CONFIGS["B"]
    assert query(dim, res, bw, 393_216) < (1 << 20)                 # config B
    assert query(dim, res, bw, 24 * 393_216) < (1 << 20)            # config C: 9.4 M samples
    assert query(2, geo(16, 512, 24), 11, 393_216) < (1 << 20)      # kodak.yaml's 24 levels
    dim, res, bw = CONFIGS["D"]
    n = 1 << 20
    s1 = query(dim, res, bw, n)
    staged = n * 16 * 2 * 4                                         # gT [L][N][F] fp32
    # round 5: runs are reserved in whole 64-byte pieces = multiples of 4 item units: at most 3 pad units per (1 024-sample
    # tile, bucket): 1 024 tiles x 733 buckets
    slots = n * (5 * 2 + 11 * 4) * 16                               # 5 compact levels (2 slots), 11 hashed (4 pair items)
    pads = 1024 * 733 * 3 * 16
    assert staged + slots <= s1 <= staged + slots + pads + (32 << 20), s1  # S1: 1.10 GB (round 4: 1.07)
    _lib.set_option("bwd_item12", 1)                                # the optional 12-byte stream: 16 units per 192 bytes
    try:
        s1 = query(dim, res, bw, n)
    finally:
        _lib.set_option("bwd_item12", -1)
    slots, pads = n * (5 * 2 + 11 * 4) * 12, 1024 * 733 * 15 * 12
    assert staged + slots <= s1 <= staged + slots + pads + (48 << 20), s1  # 0.98 GB
    # round 6: a PLANNED call (the batch's plan at hand) accumulates the five dense levels block by block on chip: they write no
    # items, the eleven hashed levels 12-byte units. 0.84 GB (plain: 1.10)
    arr = (ctypes.c_int32 * len(res))(*res)
    _, _, T = table_layout(res, bw, dim)
    planned = L.shacira_hashgrid_backward_planned_workspace_bytes(dim, n, len(res), 2, bw, arr, T, _lib.F32)
    slots, pads = n * 11 * 4 * 12, 1024 * 704 * 15 * 12
    assert staged + slots <= planned <= staged + slots + pads + (32 << 20), planned
    assert planned <= 0.85e9 and planned < 0.77 * query(dim, res, bw, n)
    # shapes without a plan / outside the brick pass's rule: the plain size
    assert L.shacira_hashgrid_backward_planned_workspace_bytes(dim, 65536, len(res), 2, bw, arr, T, _lib.F32) == query(dim, res, bw, 65536)
    assert L.shacira_hashgrid_backward_planned_workspace_bytes(dim, n, len(res), 2, bw, arr, T, _lib.F16) == query(dim, res, bw, n, dtype=_lib.F16)
