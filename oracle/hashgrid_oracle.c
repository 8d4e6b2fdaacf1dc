/*
 * oracle/hashgrid_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Scalar CPU restatement of the reference's hash-grid kernels. Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * Follows (reference file:line, read-only at /root/reference):
 *   wisp/csrc/ops/hashgrid_interpolate_cuda.cu:17-39    hash_index   (3-D dense/hash rule, primes, uint32 wrap)
 *   wisp/csrc/ops/hashgrid_interpolate_cuda.cu:41-45    clamp = max(a, min(b, x))
 *   wisp/csrc/ops/hashgrid_interpolate_cuda.cu:47-109   forward kernel (3-D)
 *   wisp/csrc/ops/hashgrid_interpolate_cuda.cu:143-221  backward kernel (3-D, fp32 atomicAdd branch)
 *   wisp/csrc/ops/hashgrid_interpolate2d_cuda.cu:17-36, 44-99, 133-208   the 2-D twins
 *   wisp/csrc/ops/hashgrid_interpolate.cpp:44-66, 68-100, 130-186        host loops over levels
 *
 * PARITY STATUS: "parity unpinned" for this file. The reference ships no tests, golden vectors or
 * CPU path for these kernels (they are CUDA-only and there is no nvcc / NVIDIA GPU here), so this
 * restatement is pinned only by known answers derived by hand from the source (tests/test_oracle_kat.py).
 *
 * Arithmetic notes (all reproduced on purpose):
 *  - `resolution * (coords[i] * 0.5 + 0.5)` is evaluated in fp64 (0.5 is a double literal) and narrowed
 *    to fp32 when passed to clamp(float, float, float); the upper bound `resolution-1-1e-5` is a double
 *    narrowed to fp32 the same way (it rounds to exactly resolution-1 for resolution >= 258).
 *  - CUDA's min/max on floats are fminf/fmaxf (a NaN operand yields the other operand).
 *  - `1.0 - x_` is a double subtraction narrowed to fp32 == a correctly rounded fp32 subtraction.
 *  - weights are products evaluated left to right in fp32 (no FMA possible: pure products).
 *  - the feature sum `t0*c0 + t1*c1 + ...` is contracted by nvcc (-fmad=true default) into
 *    fma(t7,c7, ... fma(t1,c1, t0*c0)); restated with fmaf in that order.
 *  - dense-vs-hash test uses int32 products with short-circuit; restated with wraparound uint32
 *    products reinterpreted as int32 (what the hardware does where C leaves it undefined).
 *  - where the reference would touch memory outside the whole table (dense level, coord >= 1 exactly,
 *    res >= 258: corner pos+1 == res with weight 0) this restatement skips the corner; results agree
 *    wherever the reference is defined.
 *  - backward accumulates in double in sample order (the reference's atomicAdd order is unspecified).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static inline int32_t wrap_mul_i32(int32_t a, int32_t b) {
    return (int32_t)((uint32_t)a * (uint32_t)b);
}

/* hashgrid_interpolate_cuda.cu:17-39 */
ORACLE_API int32_t shacira_oracle_hash_index3(int32_t x, int32_t y, int32_t z, int32_t res, int32_t cs) {
    int32_t r2 = wrap_mul_i32(res, res);
    if (res < cs && r2 < cs && wrap_mul_i32(r2, res) < cs) {
        return (int32_t)((uint32_t)x + (uint32_t)y * (uint32_t)res + (uint32_t)z * (uint32_t)res * (uint32_t)res);
    }
    uint32_t h = ((uint32_t)x * 1u) ^ ((uint32_t)y * 2654435761u) ^ ((uint32_t)z * 805459861u);
    return (int32_t)(h % (uint32_t)cs);
}

/* hashgrid_interpolate2d_cuda.cu:17-36 */
ORACLE_API int32_t shacira_oracle_hash_index2(int32_t x, int32_t y, int32_t res, int32_t cs) {
    if (res < cs && wrap_mul_i32(res, res) < cs) {
        return (int32_t)((uint32_t)x + (uint32_t)y * (uint32_t)res);
    }
    uint32_t h = ((uint32_t)x * 1u) ^ ((uint32_t)y * 2654435761u);
    return (int32_t)(h % (uint32_t)cs);
}

/* .cu:41-45 with CUDA float min/max semantics */
static inline float clampf_ref(float x, float a, float b) { return fmaxf(a, fminf(b, x)); }

/* .cu:68-75: per-axis transform. Returns clamped position; *pos / *frac / *ifrac as the kernel has them. */
static inline void axis_transform(float c, int32_t res, int32_t *pos, float *x_, float *_x) {
    float hi = (float)((double)res - 1.0 - 1e-5);
    float x = clampf_ref((float)((double)res * ((double)c * 0.5 + 0.5)), 0.0f, hi);
    float fl = floorf(x);
    *pos = (int32_t)fl;
    *x_ = x - (float)(*pos);
    *_x = (float)(1.0 - (double)(*x_));
}

/* corner indices (level-local) and weights for one sample at one level. nc = 2^dim. */
static void corners(int dim, const float *c, int32_t res, int32_t cs, int32_t *idx, float *w) {
    int32_t p[3] = {0, 0, 0};
    float f[3], g[3];
    for (int a = 0; a < dim; ++a) axis_transform(c[a], res, &p[a], &f[a], &g[a]);
    if (dim == 3) {
        /* .cu:77-84 and :88-94: bit2 -> x, bit1 -> y, bit0 -> z */
        for (int j = 0; j < 8; ++j) {
            float wx = (j & 4) ? f[0] : g[0];
            float wy = (j & 2) ? f[1] : g[1];
            float wz = (j & 1) ? f[2] : g[2];
            float t = wx * wy; /* left to right */
            w[j] = t * wz;
            idx[j] = shacira_oracle_hash_index3(p[0] + ((j & 4) >> 2), p[1] + ((j & 2) >> 1), p[2] + (j & 1), res, cs);
        }
    } else {
        /* 2d.cu:72-75 and :83-88: bit1 -> x, bit0 -> y */
        for (int j = 0; j < 4; ++j) {
            float wx = (j & 2) ? f[0] : g[0];
            float wy = (j & 1) ? f[1] : g[1];
            w[j] = wx * wy;
            idx[j] = shacira_oracle_hash_index2(p[0] + ((j & 2) >> 1), p[1] + (j & 1), res, cs);
        }
    }
}

/*
 * Forward: feats[i, l*F + j] = sum_k table[(first_idx[l] + idx_k)*F + j] * w_k   (.cu:96-107, .cpp:58-61)
 * idx_out (optional, may be NULL): int32 [N, L, 2^dim] level-local corner rows.
 * w_out   (optional, may be NULL): float [N, L, 2^dim] weights.
 * T = total rows of the concatenated table (memory guard only).
 */
ORACLE_API void shacira_oracle_hashgrid_fwd(int dim, int64_t N, int L, int F, int bw, const int32_t *res,
                                            const int32_t *first_idx, int64_t T, const float *coords,
                                            const float *table, float *feats, int32_t *idx_out, float *w_out) {
    const int nc = 1 << dim;
    const int32_t cs = (int32_t)pow(2.0, (double)bw); /* .cpp:56 */
    for (int64_t i = 0; i < N; ++i) {
        for (int l = 0; l < L; ++l) {
            int32_t idx[8];
            float w[8];
            corners(dim, coords + i * dim, res[l], cs, idx, w);
            for (int k = 0; k < nc; ++k) {
                if (idx_out) idx_out[(i * L + l) * nc + k] = idx[k];
                if (w_out) w_out[(i * L + l) * nc + k] = w[k];
            }
            for (int j = 0; j < F; ++j) {
                float acc = 0.0f;
                for (int k = 0; k < nc; ++k) {
                    int64_t row = (int64_t)first_idx[l] + idx[k];
                    float t = (row >= 0 && row < T) ? table[row * F + j] : 0.0f;
                    acc = (k == 0) ? t * w[0] : fmaf(t, w[k], acc);
                }
                feats[i * (int64_t)L * F + (int64_t)l * F + j] = acc;
            }
        }
    }
}

/*
 * Backward: grad_table[(first_idx[l] + idx_k)*F + j] += grad_out[i, l*F + j] * w_k  (.cu:212-221, .cpp:81-95)
 * The fp32 product is formed first (as the kernel does), the sum is kept in double.
 * grad_table64 must be zeroed by the caller (at::zeros_like in the reference).
 */
ORACLE_API void shacira_oracle_hashgrid_bwd(int dim, int64_t N, int L, int F, int bw, const int32_t *res,
                                            const int32_t *first_idx, int64_t T, const float *coords,
                                            const float *grad_out, double *grad_table64) {
    const int nc = 1 << dim;
    const int32_t cs = (int32_t)pow(2.0, (double)bw);
    for (int64_t i = 0; i < N; ++i) {
        for (int l = 0; l < L; ++l) {
            int32_t idx[8];
            float w[8];
            corners(dim, coords + i * dim, res[l], cs, idx, w);
            for (int j = 0; j < F; ++j) {
                float g = grad_out[i * (int64_t)L * F + (int64_t)l * F + j];
                for (int k = 0; k < nc; ++k) {
                    int64_t row = (int64_t)first_idx[l] + idx[k];
                    if (row < 0 || row >= T) continue;
                    float prod = g * w[k];
                    grad_table64[row * F + j] += (double)prod;
                }
            }
        }
    }
}

/* scalar_t = double (AT_DISPATCH_FLOATING_TYPES_AND_HALF, .cu:290): `float grad = grad_output[..] * coeffs[k]` (.cu:215-217)
 * forms the product in double (double * float) and narrows it to float. The reference then adds that float through
 * `(float*)(grad_codebook + ...)`, i.e. into the LOW WORD of each double -- a bug of the reference; this restates the
 * intended sum (double accumulation of the float products, sample order). */
ORACLE_API void shacira_oracle_hashgrid_bwd_f64(int dim, int64_t N, int L, int F, int bw, const int32_t *res,
                                                const int32_t *first_idx, int64_t T, const float *coords,
                                                const double *grad_out, double *grad_table64) {
    const int nc = 1 << dim;
    const int32_t cs = (int32_t)pow(2.0, (double)bw);
    for (int64_t i = 0; i < N; ++i) {
        for (int l = 0; l < L; ++l) {
            int32_t idx[8];
            float w[8];
            corners(dim, coords + i * dim, res[l], cs, idx, w);
            for (int j = 0; j < F; ++j) {
                double g = grad_out[i * (int64_t)L * F + (int64_t)l * F + j];
                for (int k = 0; k < nc; ++k) {
                    int64_t row = (int64_t)first_idx[l] + idx[k];
                    if (row < 0 || row >= T) continue;
                    float prod = (float)(g * (double)w[k]);
                    grad_table64[row * F + j] += (double)prod;
                }
            }
        }
    }
}

/* Same as above but accumulates in fp32 in sample order: one admissible atomicAdd ordering. */
ORACLE_API void shacira_oracle_hashgrid_bwd_f32(int dim, int64_t N, int L, int F, int bw, const int32_t *res,
                                                const int32_t *first_idx, int64_t T, const float *coords,
                                                const float *grad_out, float *grad_table) {
    const int nc = 1 << dim;
    const int32_t cs = (int32_t)pow(2.0, (double)bw);
    for (int64_t i = 0; i < N; ++i) {
        for (int l = 0; l < L; ++l) {
            int32_t idx[8];
            float w[8];
            corners(dim, coords + i * dim, res[l], cs, idx, w);
            for (int j = 0; j < F; ++j) {
                float g = grad_out[i * (int64_t)L * F + (int64_t)l * F + j];
                for (int k = 0; k < nc; ++k) {
                    int64_t row = (int64_t)first_idx[l] + idx[k];
                    if (row < 0 || row >= T) continue;
                    grad_table[row * F + j] += g * w[k];
                }
            }
        }
    }
}

/* Exposes the per-axis transform for the clamp/floor edge-case KATs. */
ORACLE_API void shacira_oracle_axis(float c, int32_t res, int32_t *pos, float *frac, float *ifrac) {
    axis_transform(c, res, pos, frac, ifrac);
}
