"""ctypes binding of ``libshacira_hip.so`` (the C-ABI declared in ``include/shacira_hip.h``).

The library is the product: there is no eager/CPU fallback. If it is missing, ``lib()`` raises
``RuntimeError`` telling the user to build it (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C shacira_amd/csrc``).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHACIRA_HIP_LIB") or os.path.join(_HERE, "lib", "libshacira_hip.so")   # env: A/B builds (tools/)

F32, F16, F64 = 0, 1, 2
EINVAL, EDTYPE, EODD, EWORKSPACE = -1, -2, -3, -4
BWD_STAGE_ALL_LEVELS, BWD_REUSE_STAGED = 1, 2
PLAN_READY = 1

_lock = threading.Lock()
_lib = None

_i, _i64, _f, _p, _sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list EVERY symbol include/shacira_hip.h declares (tests check this)
SIGNATURES = {
    "shacira_abi_version": (_i, []),
    "shacira_strerror": (ctypes.c_char_p, [_i]),
    "shacira_set_option": (_i, [ctypes.c_char_p, _i]),
    "shacira_get_option": (_i, [ctypes.c_char_p]),
    "shacira_stream_probe": (_i, [_i, _p, _p, _sz, _p]),
    "shacira_hashgrid_forward_workspace_bytes": (_sz, [_i, _i64, _i, _i, _i, _p, _i64, _i]),
    "shacira_hashgrid_forward": (_i, [_i, _i64, _i, _i, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _sz, _p]),
    "shacira_hashgrid_debug_corners": (_i, [_i, _i64, _i, _i, _p, _p, _p, _p, _p]),
    "shacira_hashgrid_plan_bytes": (_sz, [_i, _i64, _i, _i, _i, _p, _i64, _i]),
    "shacira_hashgrid_forward_planned": (_i, [_i, _i64, _i, _i, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _sz, _i, _p, _sz, _p]),
    "shacira_hashgrid_backward_planned": (_i, [_i, _i64, _i, _i, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _sz, _p, _sz, _p]),
    "shacira_hashgrid_backward_workspace_bytes": (_sz, [_i, _i64, _i, _i, _i, _p, _i64, _i]),
    "shacira_hashgrid_backward_planned_workspace_bytes": (_sz, [_i, _i64, _i, _i, _i, _p, _i64, _i]),
    "shacira_hashgrid_backward": (_i, [_i, _i64, _i, _i, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _sz, _p]),
    "shacira_hashgrid_backward_levels": (_i, [_i, _i64, _i, _i, _i, _p, _p, _i64, _p, _p, _i, _p, _i, _i, _i, _p, _sz,
                                              _p]),
    "shacira_latent_decode_forward": (_i, [_i64, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p]),
    "shacira_latent_decode_backward_workspace_bytes": (_sz, [_i64, _i, _i]),
    "shacira_latent_decode_backward": (_i, [_i64, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "shacira_entropy_bits_workspace_bytes": (_sz, [_i64, _i]),
    "shacira_entropy_bits_forward": (_i, [_i64, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "shacira_entropy_bits_backward": (_i, [_i64, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "shacira_latent_decode_sga_forward": (_i, [_i64, _i, _i, _p, _p, _f, _i, _p, _p, _p, _p, _f, _p, _p]),
    "shacira_latent_decode_sga_forward_tdev": (_i, [_i64, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _f, _p, _p]),
    "shacira_latent_decode_sga_backward": (_i, [_i64, _i, _i, _p, _p, _f, _i, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p,
                                                _p, _sz, _p]),
    "shacira_latent_decode_sga_backward_tdev": (_i, [_i64, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p,
                                                     _p, _sz, _p]),
    "shacira_latent_mlp_supported": (_i, [_i, _p]),
    "shacira_latent_mlp_backward_workspace_bytes": (_sz, [_i, _p]),
    "shacira_latent_mlp_forward": (_i, [_i64, _i, _p, _p, _p, _f, _i, _p, _p, _i, _i, _f, _p, _p]),
    "shacira_latent_mlp_backward": (_i, [_i64, _i, _p, _p, _p, _f, _i, _p, _p, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "shacira_latent_decode_levels_forward": (_i, [_i, _p, _i64, _i, _i, _p, _p, _f, _i, _p, _p, _p, _p, _f, _p, _p]),
    "shacira_latent_decode_levels_backward": (_i, [_i, _p, _i64, _i, _i, _p, _p, _f, _i, _p, _p, _p, _p, _f, _p, _p, _p,
                                                   _p, _p, _p, _sz, _p]),
    "shacira_latent_multi_supported": (_i, [_i, _i, _i]),
    "shacira_latent_multi_decode_forward": (_i, [_i64, _i, _i, _i, _p, _p, _p, _f, _i, _i, _p, _p, _p, _p, _f, _p, _p]),
    "shacira_latent_multi_decode_backward": (_i, [_i64, _i, _i, _i, _p, _p, _p, _f, _i, _i, _p, _p, _p, _p, _f, _p, _p,
                                                  _p, _p, _p, _p, _sz, _p]),
    "shacira_latent_symbol_range": (_i, [_i64, _i, _p, _p, _p]),
    "shacira_latent_symbol_histogram": (_i, [_i64, _i, _p, _p, _i, _p, _p]),
    "shacira_rc_encode_bound": (_sz, [_i64]),
    "shacira_rc_encode": (_i, [_p, _i64, _p, _i, _p, _sz, _p]),
    "shacira_rc_decode": (_i, [_p, _sz, _p, _i, _i64, _p]),
    "shacira_pack_integrate_forward": (_i, [_i64, _i64, _i, _p, _p, _p, _p, _p, _p]),
    "shacira_pack_integrate_backward": (_i, [_i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "shacira_pack_sum": (_i, [_i64, _i64, _i, _p, _p, _p, _p]),
    "shacira_pack_broadcast": (_i, [_i64, _i64, _i, _p, _p, _p, _p]),
    "shacira_raymarch_ray_count": (_i, [_i64, _i, _p, _p, _f, _f, _p, _p, _p, _i, _p, _p]),
    "shacira_raymarch_ray_emit": (_i, [_i64, _i, _p, _p, _f, _f, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "shacira_raymarch_ray_emit_capped": (_i, [_i64, _i, _p, _p, _f, _f, _p, _p, _p, _i, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "shacira_raytrace_dense_count": (_i, [_i64, _p, _p, _p, _i, _p, _p]),
    "shacira_raytrace_dense_emit": (_i, [_i64, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "shacira_mlp_supported": (_i, [_i, _i, _i, _i]),
    "shacira_mlp_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "shacira_mlp_forward": (_i, [_i64, _i, _i, _i, _i, _p, _p, _p, _p]),
    "shacira_mlp_backward": (_i, [_i64, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "shacira_adam_step": (_i, [_i64, _p, _p, _p, _p, _f, _f, _f, _f, _f, _i, _i, _p]),
    "shacira_adam_step_multi": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _i, _p, _i, _p]),
    "shacira_adam_step_capturable": (_i, [_i64, _p, _p, _p, _p, _f, _f, _f, _f, _f, _p, _i, _p]),
}


def lib():
    """Load (once) and return the ctypes handle of libshacira_hip.so."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"shacira_amd: HIP library not built ({LIB_PATH} missing). Build it with "
                        "`make -C shacira_amd/csrc` (hipcc --offload-arch=gfx950); there is no CPU fallback.")
                # torch's own HIP runtime must be the one in the process: loaded first, libshacira_hip.so binds to it. Loaded
                # the other way round (this library pulling in the system libamdhip64 before `import torch`) the two
                # runtimes disagree and every launch fails with "no ROCm-capable device" (seen on the GPU box, round 4).
                import torch  # noqa: F401
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                # A/B without code changes: SHACIRA_OPTIONS="bin_acc_kib=64,fwd_variant=3"
                for item in filter(None, os.environ.get("SHACIRA_OPTIONS", "").split(",")):
                    name, _, value = item.partition("=")
                    if handle.shacira_set_option(name.strip().encode(), int(value)) != 0:
                        raise RuntimeError(f"SHACIRA_OPTIONS: unknown option or bad value: {item!r}")
                _lib = handle
    return _lib


def check(code, what=""):
    """Raise like the reference does: RuntimeError for operator failures (AT_ERROR -> RuntimeError)."""
    if code == 0:
        return
    msg = lib().shacira_strerror(int(code)).decode()
    if code == EODD:  # wisp/ops/grid.py:75-76 raises a bare Exception with this text
        raise Exception(msg)
    raise RuntimeError(f"{what}: {msg} (code {code})" if what else f"{msg} (code {code})")


options_epoch = 0   # bumped by set_option: cached workspace sizes (hip_ops) depend on the tunables


def set_option(name, value):
    global options_epoch
    check(lib().shacira_set_option(name.encode(), int(value)), "shacira_set_option")
    options_epoch += 1


def get_option(name):
    return int(lib().shacira_get_option(name.encode()))
