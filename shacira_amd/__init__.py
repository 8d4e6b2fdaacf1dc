"""shacira_amd -- MI355X-native (gfx950) hash-grid interpolation + latent quantisation path of SHACIRA.

Layout:
  csrc/      hand-written HIP kernels + the C-ABI (include/shacira_hip.h) -> lib/libshacira_hip.so
  _lib.py    ctypes binding of the C-ABI (fails loudly when the library is missing; no CPU fallback)
  hip_ops.py tensor-level operators with the reference's ``wisp._C.ops`` signatures
  wisp/      host-side mirror of the reference interface for this path
             (wisp.ops.grid, wisp.models.grids.{BLASGrid,HashGrid,LatentGrid}, latent_decoders, prob_models)
  dist.py    data-parallel sharding of sample batches + one RCCL all-reduce of codebook gradients per step
"""
__version__ = "0.1.0"
