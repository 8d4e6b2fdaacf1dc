// fixed_point.h -- 64-bit fixed-point accumulation helpers shared by the backward kernels (gfx950).
#pragma once

#include "hashgrid_device.h"

namespace shacira {

// ------------------------------------------------------------------------------------------------ fixed point
// LDS integer atomics run 1.6x faster than ds_add_f64 (2.1-2.5 vs 1.3-1.4 T op/s, profiles/r01_microbench2), so the
// accumulator images hold 64-bit fixed-point numbers. Scale per level: gmax[l] = max |grad_output| over the level's
// columns (bit pattern of the float, gathered by pass T for free; integer max on the bits orders
// finite < inf < NaN). Every contribution is |g * weight| <= gmax < 2^e, so with scale 2^(headroom - e) a contribution
// stays below 2^headroom and n_max of them below 2^62: headroom = min(50, 62 - ceil(log2(n_max))). Conversion is one
// fp64 fma with the 1.5 * 2^52 constant (the integer appears in the low mantissa bits) -- exact to the scale's LSB, i.e.
// 2^-headroom relative to gmax (>= 41 bits here vs 24 of the reference's fp32 atomics) and order-independent.
// A level whose gmax is inf / NaN falls back to the fp64 image so that non-finite gradients propagate as before.
struct FxScale {
    double scale, inv;   // 2^k, 2^-k
    bool fixed;          // false: accumulate in fp64 (non-finite gradients)
};
__device__ __forceinline__ FxScale fx_scale_of(uint32_t gmax_bits, int headroom) {
    FxScale f;
    f.fixed = gmax_bits < 0x7F800000u;
    int e = (int)((gmax_bits >> 23) & 0xFFu) - 126;   // |g| < 2^e for normal floats; denormals / zero: e = -126
    if (e < -126) e = -126;
    const int k = headroom - e;
    f.scale = __longlong_as_double((long long)(1023 + k) << 52);
    f.inv = __longlong_as_double((long long)(1023 - k) << 52);
    return f;
}
__device__ __forceinline__ unsigned long long fx_encode(float c, double scale) {
    const double magic = 6755399441055744.0;   // 1.5 * 2^52
    return (unsigned long long)(__double_as_longlong(fma((double)c, scale, magic)) - __double_as_longlong(magic));
}
__device__ __forceinline__ float fx_decode(unsigned long long v, double inv) { return (float)((double)(long long)v * inv); }
inline int fx_headroom(uint64_t n_max) {
    int bits = 0;
    while (((uint64_t)1 << bits) < n_max) ++bits;
    const int h = 62 - bits;
    return h > 50 ? 50 : (h < 24 ? 24 : h);
}

}  // namespace shacira
