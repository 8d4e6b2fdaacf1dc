// hashgrid_device.h -- device-side arithmetic shared by the hash-grid kernels (gfx950 only).
//
// Restates, instruction for instruction where rounding matters, the per-sample math of
//   wisp/csrc/ops/hashgrid_interpolate_cuda.cu:17-45, :68-94     (3-D)
//   wisp/csrc/ops/hashgrid_interpolate2d_cuda.cu:17-42, :65-88   (2-D)
// The file is compiled with -ffp-contract=off; every fused multiply-add below is an explicit fmaf.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/shacira_hip.h"

namespace shacira {

constexpr uint32_t kPrimeY = 2654435761u;  // .cu:25
constexpr uint32_t kPrimeZ = 805459861u;   // .cu:25

// Per-call level table, passed by value in the kernarg segment (SGPR-resident when indexed uniformly).
struct LevelTable {
    int32_t res[SHACIRA_MAX_LODS];    // resolution per level
    float hi[SHACIRA_MAX_LODS];       // (float)(res - 1 - 1e-5), the fp64->fp32 narrowed upper clamp
    uint32_t dense[SHACIRA_MAX_LODS]; // 1 = dense index rule, 0 = spatial hash (32-bit: a dynamically indexed byte of the
                                      // kernel arguments costs a vector load + s_waitcnt vmcnt(0), a dword is a scalar load)
    uint32_t mask;                    // codebook_size - 1   (uint32 % 2^bw == & mask)
    int32_t num_lods;
    int32_t feature_dim;
    int64_t table_rows;
    int32_t level_begin;              // backward only: levels [level_begin, level_end) are processed
    int32_t level_end;
    int32_t stage_flags;              // SHACIRA_BWD_STAGE_ALL_LEVELS / SHACIRA_BWD_REUSE_STAGED
};

// Host-side replica of the kernel's int32 dense test (.cu:27-29 / 2d.cu:26-27), short-circuit, wraparound.
inline bool level_is_dense(int dim, int32_t res, int32_t cs) {
    auto wmul = [](int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); };
    int32_t r2 = wmul(res, res);
    if (dim == 2) return res < cs && r2 < cs;
    return res < cs && r2 < cs && wmul(r2, res) < cs;
}

// .cu:68-75 for one axis. `t` = (double)c * 0.5 + 0.5 is level independent and hoisted by callers.
__device__ __forceinline__ void axis_transform(double t, int32_t res, float hi, int32_t &pos, float &fr,
                                               float &ifr) {
    float x = (float)((double)res * t);  // fp64 product narrowed to fp32 (implicit in the reference)
    x = fmaxf(0.0f, fminf(hi, x));       // clamp(): max(a, min(b, x)); NaN -> hi like CUDA's fminf
    float fl = floorf(x);
    pos = (int32_t)fl;
    fr = x - fl;
    ifr = 1.0f - fr;  // (float)(1.0 - (double)fr): a correctly rounded fp32 subtraction
}

__device__ __forceinline__ double axis_unit(float c) { return (double)c * 0.5 + 0.5; }

// Coordinates of sample i of [0, n): the loads are unconditional (index clamped into the batch) so that all of them are
// in flight together -- `live ? coords[..] : 0` compiles to one branch + load + s_waitcnt vmcnt(0) PER AXIS, i.e. DIM
// serialised memory latencies per sample. Callers ignore the result of samples >= n.
template <int DIM>
__device__ __forceinline__ void load_unit_coords(const float *__restrict__ coords, int64_t i, int64_t n, double (&t)[DIM]) {
    const float *p = coords + (i < n ? i : n - 1) * DIM;
    float c[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) c[a] = p[a];
#pragma unroll
    for (int a = 0; a < DIM; ++a) t[a] = axis_unit(c[a]);
}

// Scalar <-> storage conversions (fp32 math everywhere, like static_cast<float>(codebook[..]) in .cu:98)
template <typename T> struct Scalar;
template <> struct Scalar<float> {
    static __device__ __forceinline__ float load(const float *p) { return *p; }
    static __device__ __forceinline__ void store(float *p, float v) { *p = v; }
};
template <> struct Scalar<double> {   // fp64 tables: values narrowed to float at the load, results widened at the store (.cu:96-107)
    static __device__ __forceinline__ float load(const double *p) { return (float)*p; }
    static __device__ __forceinline__ void store(double *p, float v) { *p = (double)v; }
};
template <> struct Scalar<__half> {
    static __device__ __forceinline__ float load(const __half *p) { return __half2float(*p); }
    // The fp32 value is made opaque before it is narrowed: the reference rounds its fmaf chain to fp32 and THEN converts
    // (static_cast<scalar_t>(feat), .cu:106). Left visible, the compiler fuses the chain's last fma with the conversion into
    // v_fma_mixlo_f16 -- one rounding instead of two, a different half in ~1 of 10^4 values (seen in round 4 when the gathers
    // became unconditional; the hipcc build of the reference itself does it, tests/test_ref_kernel_vectors.py).
    static __device__ __forceinline__ float rounded_fp32(float v) {
        asm volatile("" : "+v"(v));
        return v;
    }
    static __device__ __forceinline__ void store(__half *p, float v) { *p = __float2half_rn(rounded_fp32(v)); }
};

// Corner bookkeeping for one (sample, level). NC = 2^DIM.
template <int DIM> struct Corners {
    static constexpr int NC = 1 << DIM;
    uint32_t row[NC];  // level-local row index
    float w[NC];       // interpolation weight
};

template <int DIM>
__device__ __forceinline__ void compute_corners(const double (&t)[DIM], int32_t res, float hi, bool dense,
                                                uint32_t mask, Corners<DIM> &c) {
    int32_t p[DIM];
    float f[DIM], g[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f[a], g[a]);
    if constexpr (DIM == 3) {
        // weights: (x ? f : g) * (y ? f : g) * (z ? f : g), left to right (.cu:77-84)
        float wxy[4] = {g[0] * g[1], g[0] * f[1], f[0] * g[1], f[0] * f[1]};
#pragma unroll
        for (int j = 0; j < 8; ++j) c.w[j] = wxy[j >> 1] * ((j & 1) ? f[2] : g[2]);
        uint32_t ux = (uint32_t)p[0], uy = (uint32_t)p[1], uz = (uint32_t)p[2];
        if (dense) {
            uint32_t r = (uint32_t)res;
            uint32_t by0 = uy * r, by1 = by0 + r;
            uint32_t rr = r * r;
            uint32_t bz0 = uz * rr, bz1 = bz0 + rr;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                c.row[j] = (ux + ((j >> 2) & 1)) + ((j & 2) ? by1 : by0) + ((j & 1) ? bz1 : bz0);
        } else {
            uint32_t hy0 = uy * kPrimeY, hy1 = hy0 + kPrimeY;
            uint32_t hz0 = uz * kPrimeZ, hz1 = hz0 + kPrimeZ;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                c.row[j] = ((ux + ((j >> 2) & 1)) ^ ((j & 2) ? hy1 : hy0) ^ ((j & 1) ? hz1 : hz0)) & mask;
        }
    } else {
        // 2d.cu:72-75: c0=_x.x*_x.y, c1=_x.x*x_.y, c2=x_.x*_x.y, c3=x_.x*x_.y ; corner j: bit1 -> x, bit0 -> y
        c.w[0] = g[0] * g[1];
        c.w[1] = g[0] * f[1];
        c.w[2] = f[0] * g[1];
        c.w[3] = f[0] * f[1];
        uint32_t ux = (uint32_t)p[0], uy = (uint32_t)p[1];
        if (dense) {
            uint32_t r = (uint32_t)res;
            uint32_t by0 = uy * r, by1 = by0 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) c.row[j] = (ux + ((j >> 1) & 1)) + ((j & 1) ? by1 : by0);
        } else {
            uint32_t hy0 = uy * kPrimeY, hy1 = hy0 + kPrimeY;
#pragma unroll
            for (int j = 0; j < 4; ++j) c.row[j] = ((ux + ((j >> 1) & 1)) ^ ((j & 1) ? hy1 : hy0)) & mask;
        }
    }
}

}  // namespace shacira
