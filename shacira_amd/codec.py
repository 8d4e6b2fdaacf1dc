"""Entropy-coded storage of a LatentGrid ("next" row f3 of SURVEY.md section 8).

The reference never writes a bitstream: LatentGrid.size (latent_grid.py:138-174) only *estimates* the size of the
rounded latents, or counts the bytes torchac would produce with the per-channel empirical distribution, and its
checkpoints are raw ``torch.save`` files (image_trainer.py:477-483). This module makes the size real:

  * symbols  = ``round(latent)`` per channel (``torch.round``, half to even) -- on the GPU the min/max and the
    histogram come from libshacira_hip.so (``shacira_latent_symbol_range/_histogram``), on the CPU from numpy;
  * model    = the empirical per-channel histogram, scaled to a total of 2^16 with every present symbol >= 1
    (stored in the container: 2 bytes per bin);
  * coder    = the library's static range coder (``shacira_rc_encode/_decode``, host side like the reference's use of
    torchac); ``decode(encode(x)) == round(x)`` exactly;
  * container = ``SHCR`` magic, JSON header, per channel: frequency table + payload.

Byte counts are those of THIS coder (torchac is an un-pinned third-party dependency of the reference, so its exact
counts cannot be reproduced); they sit within a fraction of a percent of the entropy estimate that *is* pinned by
the golden vectors (``LatentGrid.size(use_torchac=False)``).
"""
import ctypes
import json
import struct

import numpy as np
import torch

from . import _lib

MAGIC = b"SHCR\x01\x00"
RC_TOTAL = 1 << 16


def normalise_frequencies(counts):
    """int counts [nbins] -> uint32 freq [nbins], sum == 65536, freq >= 1 wherever counts > 0 (deterministic)."""
    counts = np.asarray(counts, dtype=np.int64)
    present = counts > 0
    k = int(present.sum())
    if k == 0:
        raise ValueError("cannot build a model for an empty histogram")
    if k > RC_TOTAL:
        raise ValueError(f"{k} distinct symbols exceed the coder's 16-bit frequency scale")
    total = int(counts.sum())
    freq = np.zeros(counts.shape, dtype=np.int64)
    freq[present] = np.maximum(1, (counts[present] * RC_TOTAL) // total)
    diff = RC_TOTAL - int(freq.sum())
    if diff != 0:
        # give / take the remainder to / from the most frequent symbols (largest first, stable order)
        order = np.argsort(-counts, kind="stable")
        i = 0
        while diff != 0:
            j = order[i % len(order)]
            if diff > 0:
                freq[j] += 1
                diff -= 1
            elif freq[j] > 1:
                freq[j] -= 1
                diff += 1
            i += 1
    assert int(freq.sum()) == RC_TOTAL and np.all(freq[present] >= 1)
    return freq.astype(np.uint32)


def rc_encode(symbols, freq):
    """symbols: int32 indices into ``freq``; returns the coded bytes."""
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    freq = np.ascontiguousarray(freq, dtype=np.uint32)
    L = _lib.lib()
    cap = L.shacira_rc_encode_bound(symbols.size)
    out = np.empty(cap, dtype=np.uint8)
    n_out = ctypes.c_size_t(0)
    _lib.check(L.shacira_rc_encode(symbols.ctypes.data_as(ctypes.c_void_p), symbols.size,
                                   freq.ctypes.data_as(ctypes.c_void_p), freq.size,
                                   out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n_out)), "shacira_rc_encode")
    return out[:n_out.value].tobytes()


def rc_decode(data, freq, n):
    freq = np.ascontiguousarray(freq, dtype=np.uint32)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(n, dtype=np.int32)
    _lib.check(_lib.lib().shacira_rc_decode(buf.ctypes.data_as(ctypes.c_void_p) if buf.size else None, buf.size,
                                            freq.ctypes.data_as(ctypes.c_void_p), freq.size, n,
                                            out.ctypes.data_as(ctypes.c_void_p)), "shacira_rc_decode")
    return out


def symbol_counts(latent):
    """(lo [ld] int64 numpy, counts [ld, nbins] int64 numpy) of round(latent); HIP kernels for device tensors."""
    if latent.is_cuda:
        from . import hip_ops
        lo, counts = hip_ops.latent_symbol_counts(latent.float())
        return lo.numpy(), counts.cpu().numpy()
    sym = torch.round(latent.detach().float()).long().numpy()
    if sym.shape[0] == 0:
        return np.zeros(sym.shape[1], np.int64), np.zeros((sym.shape[1], 1), np.int64)
    lo = sym.min(axis=0)
    nbins = int((sym.max(axis=0) - lo).max()) + 1
    counts = np.stack([np.bincount(sym[:, c] - lo[c], minlength=nbins) for c in range(sym.shape[1])])
    return lo.astype(np.int64), counts.astype(np.int64)


def entropy_bits(counts):
    """Reference estimate (latent_grid.py:150-152) from a histogram: sum clamp(-log2(p + 1e-10), 0, 1000) * count,
    evaluated in fp32 like the reference."""
    c = torch.as_tensor(np.asarray(counts)[np.asarray(counts) > 0])
    probs = c / torch.sum(c)
    bits = torch.clamp(-1.0 * torch.log(probs + 1e-10) / np.log(2.0), 0, 1000)
    return torch.sum(bits * c).item()


def compress_latents(latent):
    """[T, ld] float latents -> container bytes holding round(latent) exactly."""
    T, ld = latent.shape
    if not bool(torch.isfinite(latent).all()):
        raise ValueError("latents must be finite to be entropy coded")
    if T and float(latent.detach().abs().max()) >= 2.0 ** 30:
        raise ValueError("latents beyond +-2^30 are outside the coder's symbol range")
    lo, counts = symbol_counts(latent)
    sym = torch.round(latent.detach().float()).to(torch.int64).cpu().numpy()
    header = {"rows": int(T), "latent_dim": int(ld), "channels": []}
    blobs = []
    for c in range(ld):
        if T == 0:
            header["channels"].append({"lo": 0, "nbins": 0, "payload": 0})
            continue
        hi_bin = int(np.nonzero(counts[c])[0].max()) + 1
        freq = normalise_frequencies(counts[c, :hi_bin])
        payload = rc_encode((sym[:, c] - lo[c]).astype(np.int32), freq)
        # 2 bytes per bin; only a single-bin channel has freq 65536, stored as 0 (bin 0 and the last bin are present)
        table = (freq % RC_TOTAL).astype("<u2").tobytes()
        header["channels"].append({"lo": int(lo[c]), "nbins": hi_bin, "payload": len(payload)})
        blobs += [table, payload]
    hj = json.dumps(header, separators=(",", ":")).encode()
    return MAGIC + struct.pack("<I", len(hj)) + hj + b"".join(blobs)


def decompress_latents(data, device="cpu"):
    """Inverse of compress_latents: fp32 [T, ld] tensor of the rounded latents."""
    if data[:len(MAGIC)] != MAGIC:
        raise ValueError("not a SHCR container")
    (hl,) = struct.unpack_from("<I", data, len(MAGIC))
    off = len(MAGIC) + 4
    header = json.loads(data[off:off + hl].decode())
    off += hl
    T, ld = header["rows"], header["latent_dim"]
    out = np.zeros((T, ld), dtype=np.float32)
    for c, ch in enumerate(header["channels"]):
        if T == 0:
            continue
        nb = ch["nbins"]
        freq = np.frombuffer(data, dtype="<u2", count=nb, offset=off).astype(np.uint32)
        off += 2 * nb
        if nb == 1:                       # one symbol only: its frequency 65536 is stored as 0
            freq = np.array([RC_TOTAL], dtype=np.uint32)
        if int(freq.sum()) != RC_TOTAL:
            raise ValueError("corrupt frequency table")
        payload = data[off:off + ch["payload"]]
        off += ch["payload"]
        out[:, c] = rc_decode(payload, freq, T).astype(np.float32) + np.float32(ch["lo"])
    return torch.from_numpy(out).to(device)


def payload_bits(data):
    """Bits of the arithmetic-coded payloads only (what torchac's byte_stream length measures in the reference)."""
    (hl,) = struct.unpack_from("<I", data, len(MAGIC))
    header = json.loads(data[len(MAGIC) + 4:len(MAGIC) + 4 + hl].decode())
    return 8 * sum(ch["payload"] for ch in header["channels"])


# ------------------------------------------------------------------------------------------------- whole-model files
MODEL_MAGIC = b"SHCM\x01\x00"


def save_model(module, path=None, latent_key_suffix="grid.codebook"):
    """Serialise a neural field that owns a LatentGrid: the latent table goes through `compress_latents` (entropy
    coded, restores round(latent)); every other entry of the state dict is stored raw (fp32 / int as is).
    Returns the bytes (and writes them to `path` when given). The reference only ever `torch.save`s the fp32 state
    (image_trainer.py:477-483), so its reported sizes are estimates; this file IS the size."""
    state = module.state_dict()
    entries, blobs, off = [], [], 0
    for name, t in state.items():
        if name.endswith(latent_key_suffix) and t.dim() == 2 and t.is_floating_point():
            blob, kind = compress_latents(t), "latents"
            meta = {"shape": list(t.shape)}
        else:
            arr = t.detach().cpu().contiguous().numpy()
            blob, kind = arr.tobytes(), "raw"
            meta = {"shape": list(arr.shape), "dtype": str(arr.dtype)}
        entries.append({"name": name, "kind": kind, "offset": off, "nbytes": len(blob), **meta})
        blobs.append(blob)
        off += len(blob)
    hj = json.dumps({"entries": entries}, separators=(",", ":")).encode()
    data = MODEL_MAGIC + struct.pack("<I", len(hj)) + hj + b"".join(blobs)
    if path is not None:
        with open(path, "wb") as fh:
            fh.write(data)
    return data


def load_model(module, data):
    """Inverse of save_model (`data`: bytes or a path). Parameters are overwritten in place; the latent table receives
    the stored integers, i.e. exactly what the decoder sees at validation time (round(latent))."""
    if isinstance(data, (str, bytes)) and not isinstance(data, bytes):
        with open(data, "rb") as fh:
            data = fh.read()
    if data[:len(MODEL_MAGIC)] != MODEL_MAGIC:
        raise ValueError("not a SHCM model file")
    (hl,) = struct.unpack_from("<I", data, len(MODEL_MAGIC))
    base = len(MODEL_MAGIC) + 4 + hl
    header = json.loads(data[len(MODEL_MAGIC) + 4:base].decode())
    state = module.state_dict()
    seen = set()
    with torch.no_grad():
        for e in header["entries"]:
            if e["name"] not in state:
                raise KeyError(f"file entry {e['name']} has no counterpart in the module")
            dst = state[e["name"]]
            blob = data[base + e["offset"]:base + e["offset"] + e["nbytes"]]
            if e["kind"] == "latents":
                val = decompress_latents(blob, device=dst.device)
            else:
                val = torch.from_numpy(np.frombuffer(blob, dtype=np.dtype(e["dtype"])).reshape(e["shape"]).copy())
            if tuple(val.shape) != tuple(dst.shape):
                raise ValueError(f"{e['name']}: file holds {tuple(val.shape)}, module {tuple(dst.shape)}")
            dst.copy_(val.to(dst.device).to(dst.dtype))
            seen.add(e["name"])
    missing = [k for k in state if k not in seen]
    if missing:
        raise KeyError(f"module entries missing from the file: {missing[:5]}")
    return module
