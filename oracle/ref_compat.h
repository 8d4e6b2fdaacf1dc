// oracle/ref_compat.h -- TEST INFRASTRUCTURE (force-included by oracle/ref_build.py, never by the product).
//
// The whole of what stands between the reference's hash-grid sources and this image's torch 2.10:
// `AT_DISPATCH_FLOATING_TYPES_AND_HALF(feats.type(), ...)` (reference
// wisp/csrc/ops/hashgrid_interpolate_cuda.cu:125,290 and hashgrid_interpolate2d_cuda.cu:115,251) hands the dispatch
// macro a `DeprecatedTypeProperties`. torch 1.12's ATen/Dispatch.h had the overload below (marked deprecated);
// torch 2.x dropped it while keeping `Tensor::type()` itself. It is restored here verbatim in meaning -- one
// accessor call, no arithmetic, nothing of the kernels -- so that the reference's four launchers compile unedited.
// Everything that computes (the eight `__global__` kernels, `hash_index`, `clamp`, the host loops over levels in
// hashgrid_interpolate.cpp) is the reference's own text, translated cuda->hip by torch's bundled hipify
// (torch.utils.hipify, the tool torch's own extension builder applies to every CUDA extension on ROCm).
#pragma once
#include <ATen/ATen.h>
#include <ATen/Dispatch.h>

namespace detail {
inline at::ScalarType scalar_type(const at::DeprecatedTypeProperties &t) { return t.scalarType(); }
}  // namespace detail
