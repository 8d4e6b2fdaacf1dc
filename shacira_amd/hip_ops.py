"""Tensor-level wrappers over the C-ABI: PyTorch supplies device memory and the stream, nothing else.

These functions have the signatures of the reference's pybind11 operators
(``wisp._C.ops.hashgrid_interpolate[2d]_cuda`` / ``..._backward_cuda``, wisp/csrc/bindings.cpp:24-28 and
wisp/csrc/ops/hashgrid_interpolate.h:18-50) so ``wisp/ops/grid.py``-style callers read the same.
Inputs must live on a HIP device; there is no CPU path (RuntimeError otherwise).
"""
import ctypes
import functools
import threading
import weakref

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.float64: _lib.F64}


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@functools.lru_cache(maxsize=64)
def _res_array(resolutions):
    return (ctypes.c_int32 * len(resolutions))(*resolutions)


_warned = set()


def warn_unfused(what, why):
    """One warning per (what, why): a GPU tensor is about to take the torch-op chain instead of a fused HIP kernel."""
    key = (what, why)
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(f"shacira_amd: {what} runs as torch ops, not as the fused HIP kernel ({why})", RuntimeWarning,
                      stacklevel=3)


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("shacira_amd: operands must be on the MI355X (HIP) device; there is no CPU fallback")


def _check_coords(dim, coords):
    # the reference reads coords as a flat [N, dim] array without looking at its shape (a [N, 2] tensor handed to the
    # 3-D op is read past its end): a wrong shape is refused here instead of faulting on the device
    if coords.dim() != 2 or coords.shape[1] != dim or not coords.is_contiguous():
        raise RuntimeError(f"shacira_amd: coords must be a contiguous [N, {dim}] tensor for the {dim}-D operator, got "
                           f"{tuple(coords.shape)}")


# ---- scratch memory: one grow-only buffer per (device, stream, host thread) -----------------------------------------------
# The operators' workspaces (up to ~1 GB for a 2^20-sample backward) were a fresh torch.empty per call. Work one thread puts
# on one stream is ordered, so its calls can share one buffer that only grows. Per host THREAD, because an operator is several
# launches and ctypes drops the GIL: two threads on the same stream could interleave their launch sequences (the autograd
# engine's backward thread therefore gets a buffer of its own). Exceptions: while the stream is being captured into a HIP
# graph the buffer comes from the graph's own pool (a cached buffer could be replaced -- freed -- by a later, larger eager
# call while the graph still replays into it), and a caller-supplied workspace wins.
_tls = threading.local()     # per-thread: a thread's buffers are dropped with the thread (no entries of dead threads linger)


class _Scratch:
    """One thread's buffers: {(device, stream): [buffer, calls in a row that asked for less than a quarter of it]}."""
    __slots__ = ("buffers", "__weakref__")

    def __init__(self):
        self.buffers = {}


# every live thread's _Scratch (weak: a finished thread's entry goes with its thread-local storage), so that
# release_workspaces() can reach the gigabyte the autograd engine's backward thread holds (round-4 advisor finding)
_all_scratch = weakref.WeakSet()
_all_scratch_lock = threading.Lock()
_SHRINK_AFTER = 64           # calls in a row far below the buffer's size before it is given back


def _workspace(device, nbytes):
    if nbytes <= 0:
        return None
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((nbytes,), dtype=torch.uint8, device=device)
    scratch = getattr(_tls, "scratch", None)
    if scratch is None:
        scratch = _tls.scratch = _Scratch()
        with _all_scratch_lock:
            _all_scratch.add(scratch)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    entry = scratch.buffers.get(key)
    if entry is not None and entry[0].numel() >= nbytes:
        # retained memory is bounded in time, too: a buffer grown by one large call (1.1 GB for a 2^20-sample backward) is
        # given back once _SHRINK_AFTER calls in a row needed less than a quarter of it
        if nbytes * 4 < entry[0].numel():
            entry[1] += 1
            if entry[1] < _SHRINK_AFTER:
                return entry[0]
        else:
            entry[1] = 0
            return entry[0]
    scratch.buffers.pop(key, None)       # release the old buffer before the new one is allocated
    entry = None
    buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    scratch.buffers[key] = [buf, 0]
    return buf


def release_workspaces(all_threads=True):
    """Drop the cached scratch buffers (they go back to torch's caching allocator): the calling thread's, and -- by default
    -- every other thread's too (the autograd engine's backward thread keeps its own ~1 GB after a large backward). Call it
    between steps, not while another thread is inside an operator of this library. Returns the number of bytes released."""
    freed = 0
    with _all_scratch_lock:
        targets = list(_all_scratch) if all_threads else [getattr(_tls, "scratch", None)]
    for sc in targets:
        if sc is None:
            continue
        for key in list(sc.buffers):
            entry = sc.buffers.pop(key, None)
            if entry is not None:
                freed += entry[0].numel()
    return freed


def retained_workspace_bytes():
    """Bytes currently held in scratch buffers over all threads (what release_workspaces() would give back)."""
    with _all_scratch_lock:
        return sum(e[0].numel() for sc in list(_all_scratch) for e in list(sc.buffers.values()))


# workspace sizes are pure functions of the shape and the library's tunables: one C call per new shape, not per call
@functools.lru_cache(maxsize=4096)   # (NeRF steps change N every step: 256 entries thrashed)
def _ws_bytes(kind, dim, N, L, F, bw, res, T, dt, epoch):
    fn = (_lib.lib().shacira_hashgrid_forward_workspace_bytes, _lib.lib().shacira_hashgrid_backward_workspace_bytes,
          _lib.lib().shacira_hashgrid_plan_bytes, _lib.lib().shacira_hashgrid_backward_planned_workspace_bytes)[kind]
    return int(fn(dim, N, L, F, bw, _res_array(res), T, dt))


def hashgrid_plan_bytes(dim, num_coords, table_rows, table_dtype, resolution, codebook_bitwidth, feature_dim):
    """Size of the plan buffer the forward of this shape fills (0: it sorts nothing -- pass ``plan=None``)."""
    res = tuple(int(r) for r in resolution)
    return _ws_bytes(2, dim, int(num_coords), len(res), int(feature_dim), int(codebook_bitwidth), res, int(table_rows),
                     _DTYPES[table_dtype], _lib.options_epoch)


def hashgrid_backward_workspace_bytes(dim, num_coords, table_rows, table_dtype, resolution, codebook_bitwidth, feature_dim,
                                      planned=False):
    """Scratch bytes of the backward of this shape; ``planned=True``: of a whole call that brings the batch's plan."""
    res = tuple(int(r) for r in resolution)
    return _ws_bytes(3 if planned else 1, dim, int(num_coords), len(res), int(feature_dim), int(codebook_bitwidth), res,
                     int(table_rows), _DTYPES[table_dtype], _lib.options_epoch)


def hashgrid_plan_buffer(dim, coords, codebook, resolution, codebook_bitwidth):
    """A plan buffer for this batch and table (uint8 tensor), or None when the forward of this shape builds no plan. Pass it
    to the forward (``plan=``), keep it with the coordinates, pass it to the backward of the same batch."""
    n = hashgrid_plan_bytes(dim, coords.shape[0], codebook.shape[0], codebook.dtype, resolution, codebook_bitwidth,
                            codebook.shape[1])
    return torch.empty((n,), dtype=torch.uint8, device=codebook.device) if n > 0 else None


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (the context manager costs ~4 us a call)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        self.ctx = None if torch.cuda.current_device() == device.index else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f"shacira_amd: unsupported table dtype {t.dtype} (fp32, fp16 and fp64 are implemented)")


def _hashgrid_forward(dim, coords, codebook, codebook_first_idx, resolution, codebook_bitwidth, plan=None, plan_ready=False):
    _need_gpu(coords, codebook, codebook_first_idx)
    if coords.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float for coords")  # data_ptr<float>() in the reference
    _check_coords(dim, coords)
    res = tuple(int(r) for r in resolution)
    N, T, F = coords.shape[0], codebook.shape[0], codebook.shape[1]
    feats = torch.empty((N, F * len(res)), dtype=codebook.dtype, device=codebook.device)
    dt = _dtype_code(codebook)
    L = _lib.lib()
    with _on_device(codebook.device):
        nbytes = _ws_bytes(0, dim, N, len(res), F, int(codebook_bitwidth), res, T, dt, _lib.options_epoch)
        ws = _workspace(codebook.device, nbytes)
        if plan is None:
            rc = L.shacira_hashgrid_forward(dim, N, len(res), F, int(codebook_bitwidth), _res_array(res),
                                            _ptr(codebook_first_idx), T, _ptr(coords), _ptr(codebook), dt, _ptr(feats),
                                            _ptr(ws), nbytes, _stream(codebook))
        else:
            _need_gpu(plan)
            rc = L.shacira_hashgrid_forward_planned(dim, N, len(res), F, int(codebook_bitwidth), _res_array(res),
                                                    _ptr(codebook_first_idx), T, _ptr(coords), _ptr(codebook), dt,
                                                    _ptr(feats), _ptr(plan), plan.numel(),
                                                    _lib.PLAN_READY if plan_ready else 0, _ptr(ws), nbytes,
                                                    _stream(codebook))
    _lib.check(rc, "hashgrid_interpolate")
    return feats


def hashgrid_backward(dim, coords, grad_output, table_rows, table_dtype, codebook_first_idx, resolution,
                      codebook_bitwidth, feature_dim, levels=None, out=None, workspace=None, flags=0, plan=None):
    """grad_codebook [table_rows, feature_dim] of ``table_dtype`` (the codebook's values are not needed).

    ``plan``: the buffer the forward of the SAME coordinate batch filled (``hashgrid_plan_buffer``); whole calls only.

    ``levels=(begin, end)`` computes (and overwrites) only the rows of those levels inside ``out`` (an existing
    gradient buffer) -- used to overlap the all-reduce of finished rows with the remaining levels. A series of such
    calls can share ``workspace`` (see ``backward_workspace``) with ``flags`` BWD_STAGE_ALL_LEVELS on the first and
    BWD_REUSE_STAGED on the following ones, so the gradients are transposed once."""
    _need_gpu(coords, grad_output, codebook_first_idx)
    _check_coords(dim, coords)
    res = tuple(int(r) for r in resolution)
    N, T, F = coords.shape[0], int(table_rows), int(feature_dim)
    if table_dtype not in _DTYPES:
        raise RuntimeError(f"shacira_amd: unsupported table dtype {table_dtype} (fp32, fp16 and fp64 are implemented)")
    dt = _DTYPES[table_dtype]
    if grad_output.dtype != table_dtype:
        grad_output = grad_output.to(table_dtype)
    device = grad_output.device
    if out is not None:
        if tuple(out.shape) != (T, F) or out.dtype != table_dtype or not out.is_contiguous():
            raise RuntimeError("out must be a contiguous [table_rows, feature_dim] tensor of the table dtype")
        grad_codebook = out
    else:
        if levels is not None and tuple(levels) != (0, len(res)):
            raise RuntimeError("a level range needs an existing `out` buffer (other rows are left untouched)")
        grad_codebook = torch.empty((T, F), dtype=table_dtype, device=device)
    lb, le = (0, len(res)) if levels is None else (int(levels[0]), int(levels[1]))
    L = _lib.lib()
    with _on_device(device):
        planned = plan is not None and (lb, le) == (0, len(res)) and not flags
        # (a planned call whose coarse levels go through the brick pass writes fewer, smaller items: its own query -- valid for
        # 16-byte aligned gradients, which every freshly allocated tensor is)
        kind = 3 if planned and grad_output.data_ptr() % 16 == 0 else 1
        nbytes = _ws_bytes(kind, dim, N, len(res), F, int(codebook_bitwidth), res, T, dt, _lib.options_epoch)
        if workspace is not None:
            if workspace.numel() * workspace.element_size() < nbytes:
                raise RuntimeError("workspace too small")
            ws = workspace
        elif levels is not None and tuple(levels) != (0, len(res)):
            # a level-range call WITHOUT a shared workspace stages nothing for later calls: scratch of its own
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=device) if nbytes else None
        else:
            ws = _workspace(device, nbytes)
        if planned:
            _need_gpu(plan)
            rc = L.shacira_hashgrid_backward_planned(dim, N, len(res), F, int(codebook_bitwidth), _res_array(res),
                                                     _ptr(codebook_first_idx), T, _ptr(coords), _ptr(grad_output), dt,
                                                     _ptr(grad_codebook), _ptr(plan), plan.numel(), _ptr(ws), nbytes,
                                                     _stream(grad_output))
        else:
            rc = L.shacira_hashgrid_backward_levels(dim, N, len(res), F, int(codebook_bitwidth), _res_array(res),
                                                    _ptr(codebook_first_idx), T, _ptr(coords), _ptr(grad_output), dt,
                                                    _ptr(grad_codebook), lb, le, int(flags), _ptr(ws), nbytes,
                                                    _stream(grad_output))
    _lib.check(rc, "hashgrid_interpolate_backward")
    return grad_codebook


def hashgrid_debug_corners(dim, coords, resolution, codebook_bitwidth):
    """Test hook: (rows int32 [N, L, 2^dim], weights fp32 [N, L, 2^dim]) exactly as the kernels compute them."""
    _need_gpu(coords)
    if coords.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float for coords")
    _check_coords(dim, coords)      # [N, dim], contiguous: the kernel reads dim floats per sample
    res = tuple(int(r) for r in resolution)
    N = coords.shape[0]
    rows = torch.empty((N, len(res), 1 << dim), dtype=torch.int32, device=coords.device)
    w = torch.empty((N, len(res), 1 << dim), dtype=torch.float32, device=coords.device)
    with _on_device(coords.device):
        rc = _lib.lib().shacira_hashgrid_debug_corners(dim, N, len(res), int(codebook_bitwidth), _res_array(res),
                                                       _ptr(coords), _ptr(rows), _ptr(w), _stream(coords))
    _lib.check(rc, "hashgrid_debug_corners")
    return rows, w


def backward_workspace(dim, num_coords, table_rows, table_dtype, resolution, codebook_bitwidth, feature_dim, device,
                       planned=False):
    """Scratch buffer a caller can share between several ``hashgrid_backward(..., levels=...)`` calls; ``planned=True``: the
    (smaller) buffer of whole calls that bring the batch's plan."""
    res = tuple(int(r) for r in resolution)
    with _on_device(device):
        query = (_lib.lib().shacira_hashgrid_backward_planned_workspace_bytes if planned
                 else _lib.lib().shacira_hashgrid_backward_workspace_bytes)
        n = query(dim, int(num_coords), len(res), int(feature_dim), int(codebook_bitwidth), _res_array(res),
                  int(table_rows), _DTYPES[table_dtype])
    return torch.empty((max(n, 1),), dtype=torch.uint8, device=device)


def hashgrid_interpolate_cuda(coords, codebook, codebook_first_idx, resolution, codebook_bitwidth, plan=None,
                              plan_ready=False):
    """hashgrid_interpolate.h:18-23 -> feats [N, L*F] (3-D coords). ``plan``: see ``hashgrid_plan_buffer``."""
    return _hashgrid_forward(3, coords, codebook, codebook_first_idx, resolution, codebook_bitwidth, plan, plan_ready)


def hashgrid_interpolate2d_cuda(coords, codebook, codebook_first_idx, resolution, codebook_bitwidth, plan=None,
                                plan_ready=False):
    """hashgrid_interpolate.h:35-40 -> feats [N, L*F] (2-D coords)."""
    return _hashgrid_forward(2, coords, codebook, codebook_first_idx, resolution, codebook_bitwidth, plan, plan_ready)


def hashgrid_interpolate_backward_cuda(coords, grad_output, codebook, codebook_first_idx, resolution,
                                       codebook_bitwidth, feature_dim, require_grad_coords):
    """hashgrid_interpolate.h:25-33 -> grad_codebook [T, F]."""
    return hashgrid_backward(3, coords, grad_output, codebook.shape[0], codebook.dtype, codebook_first_idx, resolution,
                             codebook_bitwidth, feature_dim)


def hashgrid_interpolate2d_backward_cuda(coords, grad_output, codebook, codebook_first_idx, resolution,
                                         codebook_bitwidth, feature_dim, require_grad_coords):
    """hashgrid_interpolate.h:42-50 -> grad_codebook [T, F]."""
    return hashgrid_backward(2, coords, grad_output, codebook.shape[0], codebook.dtype, codebook_first_idx, resolution,
                             codebook_bitwidth, feature_dim)


# ------------------------------------------------------------------------------------------------ latent path
def latent_decode_supported(latent_dim, feature_dim):
    return latent_dim in (1, 2, 3, 4, 8) and feature_dim in (1, 2, 4, 8) and not (latent_dim == 8 and feature_dim == 1)


def _latent_workspace(device):
    """fp64 block partials of the table reductions: a fresh buffer per call (the caching allocator makes that cheap).
    A buffer cached per stream would be shared by two host threads working on that stream -- A.kernel, B.kernel,
    A.finish reduces B's partials -- and could be captured into a graph pool and then reused eagerly."""
    n = _lib.lib().shacira_entropy_bits_workspace_bytes(0, 1)
    return torch.empty((n,), dtype=torch.uint8, device=device)


def latent_decode_forward(latent, div, matrix, colscale, shift, clamp_weights):
    _need_gpu(latent, div, matrix, colscale, shift)
    T, ld = latent.shape
    F = matrix.shape[1]
    out = torch.empty((T, F), dtype=torch.float32, device=latent.device)
    with _on_device(latent.device):
        rc = _lib.lib().shacira_latent_decode_forward(T, ld, F, _ptr(latent), _ptr(div), _ptr(matrix), _ptr(colscale),
                                                      _ptr(shift), float(clamp_weights), _ptr(out), _stream(latent))
    _lib.check(rc, "latent_decode_forward")
    return out


def latent_decode_backward(latent, div, matrix, colscale, shift, clamp_weights, grad_decoded, need_colscale):
    _need_gpu(latent, grad_decoded)
    T, ld = latent.shape
    F = matrix.shape[1]
    dev = latent.device
    g_lat = torch.empty_like(latent)
    g_mat = torch.empty((ld, F), dtype=torch.float32, device=dev)
    g_cs = torch.empty((F,), dtype=torch.float32, device=dev) if need_colscale else None
    g_sh = torch.empty((F,), dtype=torch.float32, device=dev)
    with _on_device(dev):
        ws = _latent_workspace(dev)
        rc = _lib.lib().shacira_latent_decode_backward(T, ld, F, _ptr(latent), _ptr(div), _ptr(matrix), _ptr(colscale),
                                                       _ptr(shift), float(clamp_weights), _ptr(grad_decoded),
                                                       _ptr(g_lat), _ptr(g_mat), _ptr(g_cs), _ptr(g_sh), _ptr(ws),
                                                       ws.numel(), _stream(latent))
    _lib.check(rc, "latent_decode_backward")
    return g_lat, g_mat, g_cs, g_sh


def latent_decode_sga_forward(latent, uniforms, temperature, diff_sampling, div, matrix, colscale, shift, clamp_weights):
    """SGA sample (reference basic_latent_decoder.py:183-191) + decode in one kernel; ``uniforms`` [T, ld, 2]."""
    _need_gpu(latent, uniforms, div, matrix, colscale, shift)
    T, ld = latent.shape
    F = matrix.shape[1]
    if tuple(uniforms.shape) != (T, ld, 2) or uniforms.dtype != torch.float32 or not uniforms.is_contiguous():
        raise RuntimeError("uniforms must be a contiguous fp32 [rows, latent_dim, 2] tensor")
    out = torch.empty((T, F), dtype=torch.float32, device=latent.device)
    with _on_device(latent.device):
        if torch.is_tensor(temperature):     # one fp32 value in device memory (graph-captured steps anneal it between replays)
            _check_device_scalar(temperature, latent.device)
            rc = _lib.lib().shacira_latent_decode_sga_forward_tdev(
                T, ld, F, _ptr(latent), _ptr(uniforms), _ptr(temperature), int(bool(diff_sampling)), _ptr(div), _ptr(matrix),
                _ptr(colscale), _ptr(shift), float(clamp_weights), _ptr(out), _stream(latent))
        else:
            rc = _lib.lib().shacira_latent_decode_sga_forward(T, ld, F, _ptr(latent), _ptr(uniforms), float(temperature),
                                                              int(bool(diff_sampling)), _ptr(div), _ptr(matrix),
                                                              _ptr(colscale), _ptr(shift), float(clamp_weights), _ptr(out),
                                                              _stream(latent))
    _lib.check(rc, "latent_decode_sga_forward")
    return out


def _check_device_scalar(t, device):
    if t.device != device or t.dtype != torch.float32 or t.numel() != 1:
        raise RuntimeError("a device-side temperature must be one fp32 value on the table's device")


def latent_decode_sga_backward(latent, uniforms, temperature, diff_sampling, div, matrix, colscale, shift, clamp_weights,
                               grad_decoded, need_colscale):
    _need_gpu(latent, uniforms, grad_decoded)
    T, ld = latent.shape
    F = matrix.shape[1]
    dev = latent.device
    g_lat = torch.empty_like(latent)
    g_mat = torch.empty((ld, F), dtype=torch.float32, device=dev)
    g_cs = torch.empty((F,), dtype=torch.float32, device=dev) if need_colscale else None
    g_sh = torch.empty((F,), dtype=torch.float32, device=dev)
    with _on_device(dev):
        ws = _latent_workspace(dev)
        if torch.is_tensor(temperature):
            _check_device_scalar(temperature, dev)
            rc = _lib.lib().shacira_latent_decode_sga_backward_tdev(
                T, ld, F, _ptr(latent), _ptr(uniforms), _ptr(temperature), int(bool(diff_sampling)), _ptr(div),
                _ptr(matrix), _ptr(colscale), _ptr(shift), float(clamp_weights), _ptr(grad_decoded), _ptr(g_lat),
                _ptr(g_mat), _ptr(g_cs), _ptr(g_sh), _ptr(ws), ws.numel(), _stream(latent))
        else:
            rc = _lib.lib().shacira_latent_decode_sga_backward(
                T, ld, F, _ptr(latent), _ptr(uniforms), float(temperature), int(bool(diff_sampling)), _ptr(div),
                _ptr(matrix), _ptr(colscale), _ptr(shift), float(clamp_weights), _ptr(grad_decoded), _ptr(g_lat),
                _ptr(g_mat), _ptr(g_cs), _ptr(g_sh), _ptr(ws), ws.numel(), _stream(latent))
    _lib.check(rc, "latent_decode_sga_backward")
    return g_lat, g_mat, g_cs, g_sh


LATENT_ACTIVATIONS = {"none": 0, "sigmoid": 1, "tanh": 2, "relu": 3, "sine": 4}   # SHACIRA_ACT_*


def _widths_array(widths):
    return (ctypes.c_int32 * len(widths))(*[int(w) for w in widths])


def latent_mlp_supported(widths):
    """Hidden-layer latent decoder (shacira_latent_mlp_*): ``widths`` = (latent_dim, hidden..., feature_dim)."""
    return len(widths) >= 2 and bool(_lib.lib().shacira_latent_mlp_supported(len(widths) - 1, _widths_array(widths)))


def _check_mlp_operands(latent, uniforms, div, params, widths):
    _need_gpu(latent, uniforms, div, params)
    # the kernels read latent as float[rows * ld] and div as float[ld]: anything else would read garbage or out of bounds
    if latent.dim() != 2 or latent.dtype != torch.float32 or not latent.is_contiguous():
        raise RuntimeError("latent must be a contiguous fp32 [rows, latent_dim] tensor")
    if div.dtype != torch.float32 or not div.is_contiguous():
        raise RuntimeError("div must be a contiguous fp32 vector of latent_dim elements")
    T, ld = latent.shape
    if ld != widths[0] or div.numel() != ld:
        raise RuntimeError("latent / div do not match widths[0]")
    want = sum(a * b + b for a, b in zip(widths[:-1], widths[1:]))
    if params.numel() != want or params.dtype != torch.float32 or not params.is_contiguous():
        raise RuntimeError(f"params must be a contiguous fp32 vector of {want} elements (W_k then b_k per layer)")
    if uniforms is not None and (tuple(uniforms.shape) != (T, ld, 2) or uniforms.dtype != torch.float32
                                 or not uniforms.is_contiguous()):
        raise RuntimeError("uniforms must be a contiguous fp32 [rows, latent_dim, 2] tensor")
    return T


def latent_mlp_forward(latent, uniforms, temperature, diff_sampling, div, params, widths, activation, final_activation,
                       clamp_weights):
    """decoded [T, widths[-1]] = clamp(final_act(MLP(q(latent) / div))): one kernel (reference basic_latent_decoder.py:182-198
    with hidden layers). ``params``: per layer the effective matrix [in, out] row-major, then the shift [out]."""
    T = _check_mlp_operands(latent, uniforms, div, params, widths)
    out = torch.empty((T, widths[-1]), dtype=torch.float32, device=latent.device)
    with _on_device(latent.device):
        rc = _lib.lib().shacira_latent_mlp_forward(
            T, len(widths) - 1, _widths_array(widths), _ptr(latent), _ptr(uniforms), float(temperature),
            int(bool(diff_sampling)), _ptr(div), _ptr(params), LATENT_ACTIVATIONS[activation],
            LATENT_ACTIVATIONS[final_activation], float(clamp_weights), _ptr(out), _stream(latent))
    _lib.check(rc, "latent_mlp_forward")
    return out


def latent_mlp_backward(latent, uniforms, temperature, diff_sampling, div, params, widths, activation, final_activation,
                        clamp_weights, grad_decoded):
    """-> (grad_latent [T, ld], grad_params packed like ``params``)."""
    T = _check_mlp_operands(latent, uniforms, div, params, widths)
    _need_gpu(grad_decoded)
    if tuple(grad_decoded.shape) != (T, widths[-1]) or grad_decoded.dtype != torch.float32 or not grad_decoded.is_contiguous():
        raise RuntimeError("grad_decoded must be a contiguous fp32 [rows, feature_dim] tensor")
    dev = latent.device
    g_lat = torch.empty_like(latent)
    g_par = torch.empty_like(params)
    wa = _widths_array(widths)
    with _on_device(dev):
        n = _lib.lib().shacira_latent_mlp_backward_workspace_bytes(len(widths) - 1, wa)
        ws = torch.empty((n,), dtype=torch.uint8, device=dev)
        rc = _lib.lib().shacira_latent_mlp_backward(
            T, len(widths) - 1, wa, _ptr(latent), _ptr(uniforms), float(temperature), int(bool(diff_sampling)), _ptr(div),
            _ptr(params), LATENT_ACTIVATIONS[activation], LATENT_ACTIVATIONS[final_activation], float(clamp_weights),
            _ptr(grad_decoded), _ptr(g_lat), _ptr(g_par), _ptr(ws), ws.numel(), _stream(latent))
    _lib.check(rc, "latent_mlp_backward")
    return g_lat, g_par


def _offsets_array(offsets):
    return (ctypes.c_int64 * len(offsets))(*[int(o) for o in offsets])


def latent_decode_levels_forward(latent, offsets, uniforms, temperature, diff_sampling, div, matrix, colscale, shift,
                                 clamp_weights):
    """Per-level decoders in one launch: ``offsets`` (host ints, [L+1]) bound the levels' rows; ``div`` [L, ld],
    ``matrix`` [L, ld, F], ``colscale`` / ``shift`` [L, F] or None; ``uniforms`` [T, ld, 2] selects the SGA path."""
    _need_gpu(latent, uniforms, div, matrix, colscale, shift)
    T, ld = latent.shape
    F = matrix.shape[-1]
    out = torch.empty((T, F), dtype=torch.float32, device=latent.device)
    with _on_device(latent.device):
        rc = _lib.lib().shacira_latent_decode_levels_forward(
            len(offsets) - 1, _offsets_array(offsets), T, ld, F, _ptr(latent), _ptr(uniforms), float(temperature),
            int(bool(diff_sampling)), _ptr(div), _ptr(matrix), _ptr(colscale), _ptr(shift), float(clamp_weights),
            _ptr(out), _stream(latent))
    _lib.check(rc, "latent_decode_levels_forward")
    return out


def latent_decode_levels_backward(latent, offsets, uniforms, temperature, diff_sampling, div, matrix, colscale, shift,
                                  clamp_weights, grad_decoded):
    _need_gpu(latent, grad_decoded)
    T, ld = latent.shape
    L, F = len(offsets) - 1, matrix.shape[-1]
    dev = latent.device
    g_lat = torch.empty_like(latent)
    g_mat = torch.empty((L, ld, F), dtype=torch.float32, device=dev)
    g_cs = torch.empty((L, F), dtype=torch.float32, device=dev) if colscale is not None else None
    g_sh = torch.empty((L, F), dtype=torch.float32, device=dev)
    with _on_device(dev):
        ws = _latent_workspace(dev)
        rc = _lib.lib().shacira_latent_decode_levels_backward(
            L, _offsets_array(offsets), T, ld, F, _ptr(latent), _ptr(uniforms), float(temperature),
            int(bool(diff_sampling)), _ptr(div), _ptr(matrix), _ptr(colscale), _ptr(shift), float(clamp_weights),
            _ptr(grad_decoded), _ptr(g_lat), _ptr(g_mat), _ptr(g_cs), _ptr(g_sh), _ptr(ws), ws.numel(), _stream(latent))
    _lib.check(rc, "latent_decode_levels_backward")
    return g_lat, g_mat, g_cs, g_sh


def latent_multi_supported(latent_dim, feature_dim, num_decoders):
    return bool(_lib.lib().shacira_latent_multi_supported(int(latent_dim), int(feature_dim), int(num_decoders)))


def latent_multi_decode_forward(latent, alpha, uniforms, temperature, straight_through, diff_sampling, div, scale, dft,
                                shift, clamp_weights):
    """MultiLatentDecoder in one kernel: alpha [K, T], scale [K, S, F], dft [ld, F] or None, shift [K, F] or None."""
    _need_gpu(latent, alpha, uniforms, div, scale, dft, shift)
    T, ld = latent.shape
    K, F = scale.shape[0], scale.shape[-1]
    out = torch.empty((T, F), dtype=torch.float32, device=latent.device)
    with _on_device(latent.device):
        rc = _lib.lib().shacira_latent_multi_decode_forward(
            T, ld, F, K, _ptr(latent), _ptr(alpha), _ptr(uniforms), float(temperature), int(bool(straight_through)),
            int(bool(diff_sampling)), _ptr(div), _ptr(scale), _ptr(dft), _ptr(shift), float(clamp_weights), _ptr(out),
            _stream(latent))
    _lib.check(rc, "latent_multi_decode_forward")
    return out


def latent_multi_decode_backward(latent, alpha, uniforms, temperature, straight_through, diff_sampling, div, scale, dft,
                                 shift, clamp_weights, grad_decoded):
    _need_gpu(latent, alpha, grad_decoded)
    T, ld = latent.shape
    K, F = scale.shape[0], scale.shape[-1]
    dev = latent.device
    g_lat = torch.empty_like(latent)
    g_alpha = torch.empty_like(alpha)
    g_scale = torch.empty_like(scale)
    g_shift = torch.empty((K, F), dtype=torch.float32, device=dev) if shift is not None else None
    with _on_device(dev):
        ws = _latent_workspace(dev)
        rc = _lib.lib().shacira_latent_multi_decode_backward(
            T, ld, F, K, _ptr(latent), _ptr(alpha), _ptr(uniforms), float(temperature), int(bool(straight_through)),
            int(bool(diff_sampling)), _ptr(div), _ptr(scale), _ptr(dft), _ptr(shift), float(clamp_weights),
            _ptr(grad_decoded), _ptr(g_lat), _ptr(g_alpha), _ptr(g_scale), _ptr(g_shift), _ptr(ws), ws.numel(),
            _stream(latent))
    _lib.check(rc, "latent_multi_decode_backward")
    return g_lat, g_alpha, g_scale, g_shift


def entropy_supported(latent_dim):
    return latent_dim in (1, 2, 3, 4, 8)


def entropy_bits_forward(latent, noise, params, num_layers):
    _need_gpu(latent, noise, params)
    T, ld = latent.shape
    dev = latent.device
    total = torch.empty((), dtype=torch.float32, device=dev)
    with _on_device(dev):
        ws = _latent_workspace(dev)
        rc = _lib.lib().shacira_entropy_bits_forward(T, ld, int(num_layers), _ptr(latent), _ptr(noise), _ptr(params),
                                                     _ptr(total), _ptr(ws), ws.numel(), _stream(latent))
    _lib.check(rc, "entropy_bits_forward")
    return total


def entropy_bits_backward(latent, noise, params, num_layers, grad_total, need_latent=True):
    _need_gpu(latent, noise, params, grad_total)
    T, ld = latent.shape
    dev = latent.device
    g_lat = torch.empty_like(latent) if need_latent else None
    g_par = torch.empty((4, 3, ld), dtype=torch.float32, device=dev)
    with _on_device(dev):
        ws = _latent_workspace(dev)
        rc = _lib.lib().shacira_entropy_bits_backward(T, ld, int(num_layers), _ptr(latent), _ptr(noise), _ptr(params),
                                                      _ptr(grad_total), _ptr(g_lat), _ptr(g_par), _ptr(ws), ws.numel(),
                                                      _stream(latent))
    _lib.check(rc, "entropy_bits_backward")
    return g_lat, g_par


# ------------------------------------------------------------------------------------------------ decoder MLP (a14)
def mlp_supported(in_dim, hidden_dim, num_hidden, out_dim):
    return bool(_lib.lib().shacira_mlp_supported(int(in_dim), int(hidden_dim), int(num_hidden), int(out_dim)))


def mlp_forward(x, params, in_dim, hidden_dim, num_hidden, out_dim):
    _need_gpu(x, params)
    y = torch.empty((x.shape[0], out_dim), dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        rc = _lib.lib().shacira_mlp_forward(x.shape[0], in_dim, hidden_dim, num_hidden, out_dim, _ptr(x), _ptr(params),
                                            _ptr(y), _stream(x))
    _lib.check(rc, "mlp_forward")
    return y


def mlp_backward(x, params, grad_y, in_dim, hidden_dim, num_hidden, out_dim, need_grad_x=True):
    _need_gpu(x, params, grad_y)
    dev = x.device
    gx = torch.empty_like(x) if need_grad_x else None
    gp = torch.empty_like(params)
    L = _lib.lib()
    with _on_device(dev):
        n = L.shacira_mlp_backward_workspace_bytes(in_dim, hidden_dim, num_hidden, out_dim)
        ws = torch.empty((n,), dtype=torch.uint8, device=dev)
        rc = L.shacira_mlp_backward(x.shape[0], in_dim, hidden_dim, num_hidden, out_dim, _ptr(x), _ptr(params),
                                    _ptr(grad_y), _ptr(gx), _ptr(gp), _ptr(ws), n, _stream(x))
    _lib.check(rc, "mlp_backward")
    return gx, gp


# ---------------------------------------------------------------------------------------------- symbol statistics
def symbols_supported(latent_dim):
    return 1 <= latent_dim <= 16


def latent_symbol_counts(latent):
    """Per-channel histogram of ``round(latent)`` (half to even, as ``torch.round``).

    Returns ``(lo, counts)``: ``lo`` int64 [ld] on the host = smallest symbol per channel, ``counts`` int64
    [ld, nbins] on the device with ``counts[c, k]`` = rows whose rounded value in channel c is ``lo[c] + k``
    (trailing bins of narrower channels are 0). Replaces ``torch.unique(return_counts=True)`` of
    LatentGrid.size (reference latent_grid.py:141-143); one host read-back of 2*ld integers sizes the bins."""
    _need_gpu(latent)
    if latent.dtype != torch.float32 or latent.dim() != 2:
        raise RuntimeError("latent_symbol_counts: expected an fp32 [rows, latent_dim] tensor")
    latent = latent.detach().contiguous()
    rows, ld = latent.shape
    L = _lib.lib()
    with _on_device(latent.device):
        minmax = torch.empty((ld, 2), dtype=torch.int32, device=latent.device)
        _lib.check(L.shacira_latent_symbol_range(rows, ld, _ptr(latent), _ptr(minmax), _stream(latent)),
                   "shacira_latent_symbol_range")
        mm = minmax.cpu()
        if rows == 0:
            return torch.zeros(ld, dtype=torch.int64), torch.zeros((ld, 1), dtype=torch.int64, device=latent.device)
        nbins = int((mm[:, 1].long() - mm[:, 0].long()).max().item()) + 1
        counts = torch.empty((ld, nbins), dtype=torch.int64, device=latent.device)  # kernel writes uint64, < 2^63
        _lib.check(L.shacira_latent_symbol_histogram(rows, ld, _ptr(latent), _ptr(minmax), nbins, _ptr(counts),
                                                     _stream(latent)), "shacira_latent_symbol_histogram")
    return mm[:, 0].long(), counts
