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
 * PARITY STATUS: pinned (round 4) against outputs of the reference's own kernels: oracle/ref_build.py compiles the
 * reference's hashgrid_interpolate{,2d}_cuda.cu + hashgrid_interpolate.cpp for gfx950 (torch hipify + hipcc -O3) into
 * oracle/_ref/, tests/golden/make_ref_kernel_vectors.py ran them on an MI355X, tests/test_ref_kernel_vectors.py holds this
 * file to the committed vectors bit for bit (forward, every table type) -- see the contraction note below -- and within the
 * reference's own atomics spread (fp32 backward). The reference's fp16 / double BACKWARD is not covered (its `__half2`
 * branch is compiled under `__CUDA_ARCH__ >= 600`); there the hand-derived known answers (tests/test_oracle_kat.py) remain.
 *
 * Arithmetic notes (all reproduced on purpose):
 *  - `resolution * (coords[i] * 0.5 + 0.5)` is evaluated in fp64 (0.5 is a double literal) and narrowed
 *    to fp32 when passed to clamp(float, float, float); the upper bound `resolution-1-1e-5` is a double
 *    narrowed to fp32 the same way (it rounds to exactly resolution-1 for resolution >= 258).
 *  - CUDA's min/max on floats are fminf/fmaxf (a NaN operand yields the other operand).
 *  - `1.0 - x_` is a double subtraction narrowed to fp32 == a correctly rounded fp32 subtraction.
 *  - weights are products evaluated left to right in fp32 (no FMA possible: pure products).
 *  - the feature sum `t0*c0 + t1*c1 + ...` is contracted by nvcc (-fmad=true default) into
 *    fma(t7,c7, ... fma(t1,c1, t0*c0)); restated with fmaf in that order (contraction mode 0, the default).
 *    PINNED (round 4) through contraction mode 1: the reference's own kernels built for gfx950 (oracle/ref_build.py:
 *    torch hipify + hipcc -O3) contract the same expression as fma(t7,c7, ... fma(t2,c2, fma(t0,c0, t1*c1))) -- LLVM
 *    folds the LEFT product of `a*b + c*d` into the fma, nvcc the right one; only which of the first two products is
 *    rounded differs (<= 1 ulp of the result). In mode 1 this file reproduces those kernels' output BIT FOR BIT on every
 *    vector of tests/golden/ref_kernels.npz (fp32, fp16 and double tables, 2-D and 3-D; tests/test_ref_kernel_vectors.py),
 *    which pins everything else here -- transform, clamp, floor, hash, dense test, corner order, weights, layout --
 *    against reference-produced numbers. Mode 0 (what nvcc emits for the reference on the hardware it was written
 *    for: mul, fma, fma ...) stays the default the product is held to; the nvcc order itself cannot be executed here.
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

/* 0: nvcc's contraction of the feature sum (default); 1: LLVM's (the hipcc build of the reference, oracle/_ref);
 * 2: mode 0 with the third product added unfused (remainder loop of that build's Half kernel, see hashgrid_c.py). */
static int g_contraction = 0;
ORACLE_API void shacira_oracle_set_contraction(int mode) { g_contraction = mode; }
ORACLE_API int shacira_oracle_get_contraction(void) { return g_contraction; }

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
static void hashgrid_fwd_impl(int dim, int64_t N, int L, int F, int bw, const int32_t *res, const int32_t *first_idx,
                              int64_t T, const float *coords, const float *table, float *feats, double *feats_last64,
                              int32_t *idx_out, float *w_out) {
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
                float acc = 0.0f, t0 = 0.0f;
                for (int k = 0; k < nc; ++k) {
                    int64_t row = (int64_t)first_idx[l] + idx[k];
                    float t = (row >= 0 && row < T) ? table[row * F + j] : 0.0f;
                    if (k == 0) {
                        t0 = t;
                        acc = t * w[0];
                    } else if (k == 1 && g_contraction == 1) {
                        acc = fmaf(t0, w[0], t * w[1]); /* LLVM: fma(t0,c0, t1*c1) */
                    } else if (k == 2 && g_contraction == 2) {
                        acc = acc + t * w[2]; /* hipcc's remainder loop of the Half kernel: this product came out of a
                                                 packed multiply with t0*c0 and is ADDED, not fused */
                    } else {
                        /* the last step before its fp32 rounding: exact product (48 bits) + acc in double */
                        if (feats_last64 && k == nc - 1) {
                            /* s + e == t*w + acc EXACTLY (the product of two floats is exact in double; TwoSum) */
                            double p = (double)t * (double)w[k], a = (double)acc, s2 = p + a, bb = s2 - p;
                            double e = (p - (s2 - bb)) + (a - bb);
                            int64_t o = 2 * (i * (int64_t)L * F + (int64_t)l * F + j);
                            feats_last64[o] = s2;
                            feats_last64[o + 1] = e;
                        }
                        acc = fmaf(t, w[k], acc);
                    }
                }
                if (feats) feats[i * (int64_t)L * F + (int64_t)l * F + j] = acc;
            }
        }
    }
}

ORACLE_API void shacira_oracle_hashgrid_fwd(int dim, int64_t N, int L, int F, int bw, const int32_t *res,
                                            const int32_t *first_idx, int64_t T, const float *coords,
                                            const float *table, float *feats, int32_t *idx_out, float *w_out) {
    hashgrid_fwd_impl(dim, N, L, F, bw, res, first_idx, T, coords, table, feats, NULL, idx_out, w_out);
}

/*
 * The same sum with its LAST fma left unrounded, as an exact pair (sum, error) of doubles [N, L*F, 2]: what a
 * half-precision table's result is rounded FROM in the hipcc build of the reference wherever LLVM fuses the last fma and
 * the `static_cast<scalar_t>` of .cu:106 into one `v_fma_mixlo_f16` (one rounding; fp32 -> half ties go the other way in
 * ~1 of 10^4 values). Used only to explain tests/golden/ref_kernels.npz's fp16 cases bit for bit (contraction mode 1,
 * oracle/hashgrid_c.py: forward_half_llvm); nvcc rounds to fp32 first.
 */
ORACLE_API void shacira_oracle_hashgrid_fwd_last64(int dim, int64_t N, int L, int F, int bw, const int32_t *res,
                                                   const int32_t *first_idx, int64_t T, const float *coords,
                                                   const float *table, double *feats_last64) {
    hashgrid_fwd_impl(dim, N, L, F, bw, res, first_idx, T, coords, table, NULL, feats_last64, NULL, NULL);
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

/* ---------------------------------------------------------------------------------------------------------------------
 * fp16 tables: software model of the reference's `__half2` branch (.cu:198-211; what AMP training runs, grid.py:73).
 * Per sample, per level, per corner k (0..7 in the kernel's order), per feature pair:
 *     grad = __floats2half2_rn(__half2float(g.x) * coeffs[k], __half2float(g.y) * coeffs[k]);      (.cu:205-206)
 *     atomicAdd((__half2*)(grad_codebook + ...), grad);                                             (.cu:207)
 * i.e. the fp32 product is rounded to half (RNE), and the table entry -- a half -- takes `half(entry + grad)`, one RNE
 * rounding of the RUNNING SUM per add (an atomicAdd on __half2 is two independent half additions). The order of the adds
 * across samples is unspecified; this model takes sample order, one admissible schedule.
 * Besides the modelled table it returns, per entry,
 *     bound[e] = sum over the entry's adds of (half ulp of the rounded product) / 2 ... the products' roundings
 *              + sum over the adds of (half ulp of the new running sum) / 2       ... the running sum's roundings
 * = a rigorous bound on |model - exact sum of the fp32 products| for THIS schedule, and `maxabs[e]` = the largest |running
 * sum| (any other schedule's roundings are bounded by n * 2^-12 * its own largest running sum; the tests bound that by
 * the sum of |products|, `sumabs[e]`).
 * half values travel as floats holding exactly representable values. */
static float half_round(float x) {
    /* round-to-nearest-even of a float to IEEE binary16, returned as float (handles subnormals, overflow -> inf, NaN) */
    if (!(x == x)) return x;
    const float ax = fabsf(x);
    if (ax >= 65520.0f) return copysignf(INFINITY, x); /* ties to even at the top: 65520 rounds to inf */
    if (ax < 6.103515625e-05f) {                        /* below the smallest normal half: spacing 2^-24 */
        const float q = rintf(ax * 16777216.0f);        /* rintf: ties to even (default rounding mode) */
        return copysignf(q * (1.0f / 16777216.0f), x);
    }
    int e;
    (void)frexpf(ax, &e);                               /* ax = m * 2^e, m in [0.5, 1): ulp_half = 2^(e - 11) */
    const float ulp = ldexpf(1.0f, e - 11);
    return copysignf(rintf(ax / ulp) * ulp, x);
}
static float half_ulp(float x) {
    const float ax = fabsf(x);
    if (!(ax == ax) || ax >= 65520.0f) return 0.0f;
    if (ax < 6.103515625e-05f) return 1.0f / 16777216.0f;
    int e;
    (void)frexpf(ax, &e);
    return ldexpf(1.0f, e - 11);
}
ORACLE_API float shacira_oracle_half_round(float x) { return half_round(x); }

ORACLE_API void shacira_oracle_hashgrid_bwd_half_model(int dim, int64_t N, int L, int F, int bw, const int32_t *res,
                                                       const int32_t *first_idx, int64_t T, const float *coords,
                                                       const float *grad_out_half, float *grad_table_half, float *bound,
                                                       float *sumabs, float *sumg) {
    const int nc = 1 << dim;
    const int32_t cs = (int32_t)pow(2.0, (double)bw);
    for (int64_t i = 0; i < N; ++i) {
        for (int l = 0; l < L; ++l) {
            int32_t idx[8];
            float w[8];
            corners(dim, coords + i * dim, res[l], cs, idx, w);
            for (int j = 0; j < F; ++j) {
                const float g = grad_out_half[i * (int64_t)L * F + (int64_t)l * F + j];
                for (int k = 0; k < nc; ++k) {
                    const int64_t row = (int64_t)first_idx[l] + idx[k];
                    if (row < 0 || row >= T) continue;
                    const float prod = g * w[k];                /* fp32 product (.cu:205) */
                    const float c = half_round(prod);           /* __floats2half2_rn */
                    const float s = half_round(grad_table_half[row * F + j] + c);   /* half + half, one rounding */
                    grad_table_half[row * F + j] = s;
                    if (bound) bound[row * F + j] += 0.5f * half_ulp(c) + 0.5f * half_ulp(s);
                    if (sumabs) sumabs[row * F + j] += fabsf(prod);
                    if (sumg) sumg[row * F + j] += fabsf(g);   /* >= |g * any partial weight|: bounds a quantised-weight error */
                }
            }
        }
    }
}

/* Exposes the per-axis transform for the clamp/floor edge-case KATs. */
ORACLE_API void shacira_oracle_axis(float c, int32_t res, int32_t *pos, float *frac, float *ifrac) {
    axis_transform(c, res, pos, frac, ifrac);
}
