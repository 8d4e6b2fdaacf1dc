// latent.hip -- per-entry latent round/dequantise-decode and entropy-bottleneck (CDF) evaluation (gfx950).
//
// The reference evaluates these as chains of ~10-25 ATen elementwise kernels over the whole table on every
// step: LatentDecoder.forward (wisp/models/latent_decoders/basic_latent_decoder.py:182-198,
// DecoderLayer.forward :86-91, StraightThrough :28-36), LatentGrid.ent_loss (wisp/models/grids/latent_grid.py:122-136)
// and BitEstimator/Bitparm.forward (wisp/models/prob_models/bit_estimator.py:27-65). Here each is ONE pass over
// the table: HBM-bound streaming kernels, one thread per table row, fp32 math; reductions over the table are
// kept as fp64 block partials (workspace) and finished by one small kernel, so they are bitwise reproducible.
//
// The tiny per-channel parameter vectors are read through uniform (scalar) loads from their device pointers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "internal.h"

namespace shacira {

constexpr int kMaxPartialBlocks = 2048;   // 8 workgroups of 256 threads per CU
constexpr int kThreads = 256;
constexpr int kMaxRed = 96;  // largest reduction width of any kernel below

size_t latent_workspace_bytes() { return (size_t)kMaxPartialBlocks * kMaxRed * sizeof(double); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// NRED per-thread fp32 partials -> fp64 block partials: partials[block][NRED]
template <int NRED>
__device__ __forceinline__ void block_reduce_store(const float (&acc)[NRED], double *__restrict__ partials) {
    __shared__ double s_part[kThreads / 64][NRED];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NRED; ++q) {
        double v = wave_sum((double)acc[q]);
        if (lane == 0) s_part[wave][q] = v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < NRED; q += kThreads) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) v += s_part[w][q];
        partials[(size_t)blockIdx.x * NRED + q] = v;
    }
}

// out[q] = scale * sum_b partials[b][q]; one workgroup per q (fixed summation tree -> bitwise reproducible).
// The nred results are split over up to three fp32 outputs (NULL = skip).
__global__ __launch_bounds__(256) void finish_partials_kernel(const double *__restrict__ partials, int nblocks,
                                                              int nred, const float *__restrict__ scale,
                                                              float *__restrict__ out0, int n0,
                                                              float *__restrict__ out1, int n1,
                                                              float *__restrict__ out2, int n2) {
    __shared__ double s_w[4];
    const int q = blockIdx.x;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) v += partials[(size_t)b * nred + q];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        if (scale) v *= (double)scale[0];
        if (q < n0) {
            if (out0) out0[q] = (float)v;
        } else if (q < n0 + n1) {
            if (out1) out1[q - n0] = (float)v;
        } else if (q < n0 + n1 + n2) {
            if (out2) out2[q - n0 - n1] = (float)v;
        }
    }
}

static inline int grid_for(int64_t rows) {
    int64_t b = (rows + kThreads - 1) / kThreads;
    if (b > kMaxPartialBlocks) b = kMaxPartialBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

// =========================================================================================================
// latent decode
// =========================================================================================================
template <int LD, int F> struct DecodeConsts {
    float div[LD], mat[LD * F], cs[F], shift[F], clampw;
    __device__ __forceinline__ void load(const float *__restrict__ d, const float *__restrict__ m,
                                         const float *__restrict__ c, const float *__restrict__ s, float cw) {
        clampw = cw;
#pragma unroll
        for (int i = 0; i < LD; ++i) div[i] = d[i];
#pragma unroll
        for (int i = 0; i < LD * F; ++i) mat[i] = m[i];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            cs[j] = c ? c[j] : 1.0f;
            shift[j] = s ? s[j] : 0.0f;
        }
    }
};

// Stochastic Gumbel annealing (basic_latent_decoder.py:183-191): relaxed one-hot choice between floor(w) and floor(w)+1
// with logits -tanh(distance)/T, sampled as torch's RelaxedOneHotCategorical(T) does from two uniforms per latent:
// u = clamp(rand, eps, 1-eps); g = -log(-log u); score = (logit + g)/T; sample = exp(score - logsumexp(score)).
// dq = d(sample)/dw: rsample() path when `diff` (floor carries no gradient), else straight-through floor (s0 + s1).
struct SgaArgs {
    const float *uniforms;  // [rows, LD, 2]
    float temperature;
    int diff;
    const float *temperature_dev;   // non-NULL: the temperature is read from device memory (a step captured into a HIP graph
                                    // anneals it between replays without re-capturing), `temperature` is ignored
};
__device__ __forceinline__ float sga_temperature(const SgaArgs &sga) {
    return sga.temperature_dev ? *sga.temperature_dev : sga.temperature;
}
__device__ __forceinline__ void sga_quantise(float w, float u0, float u1, float T, bool diff, float &q, float &dq) {
    const float lim = 1.0f - 1e-6f, eps = 1.1920929e-07f;
    const float wf = floorf(w), wc = wf + 1.0f;
    const float a = w - wf, b = wc - w;
    const float tf_ = tanhf(fminf(fmaxf(a, -lim), lim)), tc = tanhf(fminf(fmaxf(b, -lim), lim));
    const float lf = -tf_ / T, lc = -tc / T;
    u0 = fminf(fmaxf(u0, eps), 1.0f - eps);
    u1 = fminf(fmaxf(u1, eps), 1.0f - eps);
    const float z0 = (lf + -logf(-logf(u0))) / T, z1 = (lc + -logf(-logf(u1))) / T;
    const float m = fmaxf(z0, z1);
    const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
    const float s0 = expf(z0 - lse), s1 = expf(z1 - lse);
    q = wf * s0 + wc * s1;
    if (diff) {
        const float in_f = (a > -lim && a < lim) ? 1.0f : 0.0f, in_c = (b > -lim && b < lim) ? 1.0f : 0.0f;
        dq = s0 * s1 * ((1.0f - tc * tc) * in_c + (1.0f - tf_ * tf_) * in_f) / (T * T);
    } else {
        dq = s0 + s1;
    }
}

template <int LD, int F, bool SGA>
__device__ __forceinline__ void decode_row(const DecodeConsts<LD, F> &p, const float *__restrict__ latent, int64_t r,
                                           const SgaArgs &sga, float (&z)[LD], float (&dq)[LD], float (&zm)[F],
                                           float (&y)[F]) {
#pragma unroll
    for (int c = 0; c < LD; ++c) {
        if constexpr (SGA) {
            float q;
            const float2 u = *reinterpret_cast<const float2 *>(sga.uniforms + (r * LD + c) * 2);
            sga_quantise(latent[r * LD + c], u.x, u.y, sga_temperature(sga), sga.diff != 0, q, dq[c]);
            z[c] = q / p.div[c];
        } else {
            z[c] = rintf(latent[r * LD + c]) / p.div[c];  // torch.round: half to even
            dq[c] = 1.0f;                                  // straight-through rounding
        }
    }
#pragma unroll
    for (int j = 0; j < F; ++j) {
        float s = z[0] * p.mat[j];
#pragma unroll
        for (int c = 1; c < LD; ++c) s = fmaf(z[c], p.mat[c * F + j], s);
        zm[j] = s;
        y[j] = s * p.cs[j] + p.shift[j];
    }
}

template <int LD, int F, bool SGA>
__global__ __launch_bounds__(kThreads) void latent_decode_fwd_kernel(
    const float *__restrict__ latent, const float *__restrict__ div, const float *__restrict__ matrix,
    const float *__restrict__ colscale, const float *__restrict__ shift, float clampw, float *__restrict__ decoded,
    int64_t rows, SgaArgs sga) {
    DecodeConsts<LD, F> p;
    p.load(div, matrix, colscale, shift, clampw);
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < rows; r += stride) {
        float z[LD], dq[LD], zm[F], y[F];
        decode_row<LD, F, SGA>(p, latent, r, sga, z, dq, zm, y);
#pragma unroll
        for (int j = 0; j < F; ++j) {
            float v = y[j];
            if (clampw > 0.0f) v = fminf(fmaxf(v, -clampw), clampw);
            decoded[r * F + j] = v;
        }
    }
}

// reductions: [LD*F] grad_matrix, [F] grad_colscale, [F] grad_shift
template <int LD, int F, bool SGA>
__global__ __launch_bounds__(kThreads) void latent_decode_bwd_kernel(
    const float *__restrict__ latent, const float *__restrict__ div, const float *__restrict__ matrix,
    const float *__restrict__ colscale, const float *__restrict__ shift, float clampw,
    const float *__restrict__ grad_decoded, float *__restrict__ grad_latent, double *__restrict__ partials,
    int64_t rows, SgaArgs sga) {
    constexpr int NRED = LD * F + 2 * F;
    DecodeConsts<LD, F> p;
    p.load(div, matrix, colscale, shift, clampw);
    float acc[NRED];
#pragma unroll
    for (int q = 0; q < NRED; ++q) acc[q] = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < rows; r += stride) {
        float z[LD], dq[LD], zm[F], y[F], gy[F];
        decode_row<LD, F, SGA>(p, latent, r, sga, z, dq, zm, y);
#pragma unroll
        for (int j = 0; j < F; ++j) {
            float g = grad_decoded[r * F + j];
            // torch.clamp passes the gradient where -c <= y <= c
            if (clampw > 0.0f && !(y[j] >= -clampw && y[j] <= clampw)) g = 0.0f;
            gy[j] = g;
            acc[LD * F + j] += g * zm[j];  // grad_colscale
            acc[LD * F + F + j] += g;      // grad_shift
        }
#pragma unroll
        for (int c = 0; c < LD; ++c) {
            float gl = 0.0f;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const float gs = gy[j] * p.cs[j];
                acc[c * F + j] += z[c] * gs;  // grad_matrix
                gl = fmaf(gs, p.mat[c * F + j], gl);
            }
            if (grad_latent) grad_latent[r * LD + c] = gl / p.div[c] * dq[c];  // dq = 1: straight-through rounding
        }
    }
    block_reduce_store<NRED>(acc, partials);
}

template <int LD, int F> static hipError_t decode_launch(bool bwd, const DecodeArgs &a, hipStream_t s) {
    const int blocks = grid_for(a.rows);
    const SgaArgs sga{a.uniforms, a.temperature, a.diff_sampling, a.temperature_dev};
    if (!bwd) {
        if (a.uniforms)
            hipLaunchKernelGGL((latent_decode_fwd_kernel<LD, F, true>), dim3(blocks), dim3(kThreads), 0, s, a.latent,
                               a.div, a.matrix, a.colscale, a.shift, a.clampw, a.decoded, a.rows, sga);
        else
            hipLaunchKernelGGL((latent_decode_fwd_kernel<LD, F, false>), dim3(blocks), dim3(kThreads), 0, s, a.latent,
                               a.div, a.matrix, a.colscale, a.shift, a.clampw, a.decoded, a.rows, sga);
        return hipGetLastError();
    }
    if (a.uniforms)
        hipLaunchKernelGGL((latent_decode_bwd_kernel<LD, F, true>), dim3(blocks), dim3(kThreads), 0, s, a.latent, a.div,
                           a.matrix, a.colscale, a.shift, a.clampw, a.grad_decoded, a.grad_latent, a.partials, a.rows,
                           sga);
    else
        hipLaunchKernelGGL((latent_decode_bwd_kernel<LD, F, false>), dim3(blocks), dim3(kThreads), 0, s, a.latent,
                           a.div, a.matrix, a.colscale, a.shift, a.clampw, a.grad_decoded, a.grad_latent, a.partials,
                           a.rows, sga);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finish_partials_kernel, dim3(LD * F + 2 * F), dim3(256), 0, s, a.partials, blocks,
                       LD * F + 2 * F, (const float *)nullptr, a.grad_matrix, LD * F, a.grad_colscale, F, a.grad_shift,
                       F);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Per-level decoders (HierarchicalLatentDecoder, reference hierarchical_latent_decoder.py:3-36 with the offsets of
// latent_grid.py:176-190): level l owns rows [lo[l], lo[l+1]) and has its own div / matrix / colscale / shift. ONE launch:
// grid.y = level, the level's parameters are rows of stacked arrays. The reference runs num_lods decoders (each the
// ~10-kernel ATen chain) and a torch.cat.
struct LevelRows {
    int64_t lo[SHACIRA_MAX_LODS + 1];
};

template <int LD, int F, bool SGA>
__global__ __launch_bounds__(kThreads) void latent_decode_levels_fwd_kernel(
    LevelRows lr, const float *__restrict__ latent, const float *__restrict__ div, const float *__restrict__ matrix,
    const float *__restrict__ colscale, const float *__restrict__ shift, float clampw, float *__restrict__ decoded,
    SgaArgs sga) {
    const int l = blockIdx.y;
    const int64_t lo = lr.lo[l], hi = lr.lo[l + 1];
    DecodeConsts<LD, F> p;
    p.load(div + l * LD, matrix + l * LD * F, colscale ? colscale + l * F : nullptr, shift ? shift + l * F : nullptr,
           clampw);
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = lo + (int64_t)blockIdx.x * kThreads + threadIdx.x; r < hi; r += stride) {
        float z[LD], dq[LD], zm[F], y[F];
        decode_row<LD, F, SGA>(p, latent, r, sga, z, dq, zm, y);
#pragma unroll
        for (int j = 0; j < F; ++j) {
            float v = y[j];
            if (clampw > 0.0f) v = fminf(fmaxf(v, -clampw), clampw);
            decoded[r * F + j] = v;
        }
    }
}

template <int LD, int F, bool SGA>
__global__ __launch_bounds__(kThreads) void latent_decode_levels_bwd_kernel(
    LevelRows lr, const float *__restrict__ latent, const float *__restrict__ div, const float *__restrict__ matrix,
    const float *__restrict__ colscale, const float *__restrict__ shift, float clampw,
    const float *__restrict__ grad_decoded, float *__restrict__ grad_latent, double *__restrict__ partials,
    SgaArgs sga) {
    constexpr int NRED = LD * F + 2 * F;
    const int l = blockIdx.y;
    const int64_t lo = lr.lo[l], hi = lr.lo[l + 1];
    DecodeConsts<LD, F> p;
    p.load(div + l * LD, matrix + l * LD * F, colscale ? colscale + l * F : nullptr, shift ? shift + l * F : nullptr,
           clampw);
    float acc[NRED];
#pragma unroll
    for (int q = 0; q < NRED; ++q) acc[q] = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = lo + (int64_t)blockIdx.x * kThreads + threadIdx.x; r < hi; r += stride) {
        float z[LD], dq[LD], zm[F], y[F], gy[F];
        decode_row<LD, F, SGA>(p, latent, r, sga, z, dq, zm, y);
#pragma unroll
        for (int j = 0; j < F; ++j) {
            float g = grad_decoded[r * F + j];
            if (clampw > 0.0f && !(y[j] >= -clampw && y[j] <= clampw)) g = 0.0f;
            gy[j] = g;
            acc[LD * F + j] += g * zm[j];
            acc[LD * F + F + j] += g;
        }
#pragma unroll
        for (int c = 0; c < LD; ++c) {
            float gl = 0.0f;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const float gs = gy[j] * p.cs[j];
                acc[c * F + j] += z[c] * gs;
                gl = fmaf(gs, p.mat[c * F + j], gl);
            }
            if (grad_latent) grad_latent[r * LD + c] = gl / p.div[c] * dq[c];
        }
    }
    // partials[level][block][NRED]
    block_reduce_store<NRED>(acc, partials + (size_t)l * gridDim.x * NRED);
}

// per level: out[l][q] = sum_b partials[l][b][q]; grid = (nred, levels)
__global__ __launch_bounds__(256) void finish_level_partials_kernel(const double *__restrict__ partials, int nblocks,
                                                                    int nred, float *__restrict__ out0, int n0,
                                                                    float *__restrict__ out1, int n1,
                                                                    float *__restrict__ out2, int n2) {
    __shared__ double s_w[4];
    const int q = blockIdx.x, l = blockIdx.y;
    const double *pl = partials + (size_t)l * nblocks * nred;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) v += pl[(size_t)b * nred + q];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        if (q < n0) {
            if (out0) out0[(size_t)l * n0 + q] = (float)v;
        } else if (q < n0 + n1) {
            if (out1) out1[(size_t)l * n1 + (q - n0)] = (float)v;
        } else if (q < n0 + n1 + n2) {
            if (out2) out2[(size_t)l * n2 + (q - n0 - n1)] = (float)v;
        }
    }
}

template <int LD, int F>
static hipError_t decode_levels_launch(bool bwd, int levels, const int64_t *offsets, const DecodeArgs &a, hipStream_t s) {
    LevelRows lr{};
    int64_t longest = 0;
    for (int l = 0; l <= levels; ++l) lr.lo[l] = offsets[l];
    for (int l = 0; l < levels; ++l) {
        if (lr.lo[l + 1] < lr.lo[l]) lr.lo[l + 1] = lr.lo[l];   // empty level (the reference's last-offset quirk)
        if (lr.lo[l + 1] - lr.lo[l] > longest) longest = lr.lo[l + 1] - lr.lo[l];
    }
    int blocks = grid_for(longest);
    const int cap = kMaxPartialBlocks / levels;   // the fp64 partials of all levels share the one workspace
    if (blocks > cap) blocks = cap < 1 ? 1 : cap;
    const SgaArgs sga{a.uniforms, a.temperature, a.diff_sampling, a.temperature_dev};
    const dim3 grid(blocks, levels);
    if (!bwd) {
        if (a.uniforms)
            hipLaunchKernelGGL((latent_decode_levels_fwd_kernel<LD, F, true>), grid, dim3(kThreads), 0, s, lr, a.latent,
                               a.div, a.matrix, a.colscale, a.shift, a.clampw, a.decoded, sga);
        else
            hipLaunchKernelGGL((latent_decode_levels_fwd_kernel<LD, F, false>), grid, dim3(kThreads), 0, s, lr, a.latent,
                               a.div, a.matrix, a.colscale, a.shift, a.clampw, a.decoded, sga);
        return hipGetLastError();
    }
    if (a.uniforms)
        hipLaunchKernelGGL((latent_decode_levels_bwd_kernel<LD, F, true>), grid, dim3(kThreads), 0, s, lr, a.latent,
                           a.div, a.matrix, a.colscale, a.shift, a.clampw, a.grad_decoded, a.grad_latent, a.partials, sga);
    else
        hipLaunchKernelGGL((latent_decode_levels_bwd_kernel<LD, F, false>), grid, dim3(kThreads), 0, s, lr, a.latent,
                           a.div, a.matrix, a.colscale, a.shift, a.clampw, a.grad_decoded, a.grad_latent, a.partials, sga);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finish_level_partials_kernel, dim3(LD * F + 2 * F, levels), dim3(256), 0, s, a.partials, blocks,
                       LD * F + 2 * F, a.grad_matrix, LD * F, a.grad_colscale, F, a.grad_shift, F);
    return hipGetLastError();
}

typedef hipError_t (*decode_levels_fn)(bool, int, const int64_t *, const DecodeArgs &, hipStream_t);

// ---------------------------------------------------------------------------------------------------------
// MultiLatentDecoder (reference wisp/models/latent_decoders/multi_latent_decoder.py:27-210, no hidden layers): K affine
// decoders mixed per table entry by a learned selector alpha [K, T]:
//     a_soft = softmax_k(alpha[:, r] / T);  a = straight_through ? onehot(argmax a_soft) : a_soft   (gradient: identity)
//     x = quantise(latent[r]) / div                                  (rounding, or SGA with the same temperature T)
//     'dft': per_k = (x @ dft) * scale_k + shift_k ;  y = sum_k per_k a_k
//     'sq' : mixed = sum_k (x @ scale_k) a_k ; per_k = mixed + shift_k ; y = sum_k per_k a_k    (the reference mixes TWICE)
// One pass each way; the reference evaluates ~20 ATen kernels over [K, T, F] temporaries. Backward: grad_latent, grad_alpha
// [K, T] (through the softmax), grad_scale [K, S, F], grad_shift [K, F] (fp64 block partials, S = LD for 'sq', 1 for 'dft').
constexpr int kMultiMaxK = 8;

struct MultiArgs {
    const float *latent, *alpha, *div, *scale, *dft, *shift, *uniforms, *grad_decoded;
    float *decoded, *grad_latent, *grad_alpha, *grad_scale, *grad_shift;
    double *partials;
    int64_t rows;
    int K, straight_through, diff_sampling;
    float temperature, clampw;
};

template <int LD, int F, bool DFT, bool SGA, bool BWD>
__global__ __launch_bounds__(kThreads) void multi_decode_kernel(MultiArgs a) {
    constexpr int S = DFT ? 1 : LD, SF = S * F, KM = kMultiMaxK;
    constexpr int NRED = KM * (SF + F);
    __shared__ float s_scale[KM * SF], s_shift[KM * F], s_dft[LD * F], s_div[LD];
    const int K = a.K;
    for (int e = threadIdx.x; e < K * SF; e += kThreads) s_scale[e] = a.scale[e];
    for (int e = threadIdx.x; e < K * F; e += kThreads) s_shift[e] = a.shift ? a.shift[e] : 0.0f;
    if (DFT)
        for (int e = threadIdx.x; e < LD * F; e += kThreads) s_dft[e] = a.dft[e];
    if (threadIdx.x < LD) s_div[threadIdx.x] = a.div[threadIdx.x];
    __syncthreads();
    float acc[BWD ? NRED : 1];
    if constexpr (BWD) {
#pragma unroll
        for (int q = 0; q < NRED; ++q) acc[q] = 0.0f;
    }
    const float T = a.temperature;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < a.rows; r += stride) {
        // selector
        float al[KM], mx = -INFINITY;
        int win = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            al[k] = (k < K) ? a.alpha[(int64_t)k * a.rows + r] / T : -INFINITY;
            if (al[k] > mx) { mx = al[k]; win = k; }          // first maximum, as torch.argmax
        }
        float asoft[KM], den = 0.0f;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            asoft[k] = (k < K) ? expf(al[k] - mx) : 0.0f;
            den += asoft[k];
        }
        float am[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            asoft[k] /= den;
            am[k] = a.straight_through ? ((k == win) ? 1.0f : 0.0f) : asoft[k];
        }
        // quantised, normalised latent
        float x[LD], dq[LD];
#pragma unroll
        for (int c = 0; c < LD; ++c) {
            if constexpr (SGA) {
                float q;
                const float2 u = *reinterpret_cast<const float2 *>(a.uniforms + (r * LD + c) * 2);
                sga_quantise(a.latent[r * LD + c], u.x, u.y, T, a.diff_sampling != 0, q, dq[c]);
                x[c] = q / s_div[c];
            } else {
                x[c] = rintf(a.latent[r * LD + c]) / s_div[c];
                dq[c] = 1.0f;
            }
        }
        float y[F], zm[F], mixed[F];
        float u[DFT ? 1 : KM][F];
        if constexpr (DFT) {
#pragma unroll
            for (int j = 0; j < F; ++j) {
                float sacc = x[0] * s_dft[j];
#pragma unroll
                for (int c = 1; c < LD; ++c) sacc = fmaf(x[c], s_dft[c * F + j], sacc);
                zm[j] = sacc;
                float yy = 0.0f;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) yy += (zm[j] * s_scale[k * F + j] + s_shift[k * F + j]) * am[k];
                y[j] = yy;
                mixed[j] = 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < F; ++j) {
                float mj = 0.0f;
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    float sacc = 0.0f;
                    if (k < K) {
                        sacc = x[0] * s_scale[k * SF + j];
#pragma unroll
                        for (int c = 1; c < LD; ++c) sacc = fmaf(x[c], s_scale[k * SF + c * F + j], sacc);
                        mj += sacc * am[k];
                    }
                    u[k][j] = sacc;
                }
                mixed[j] = mj;
                float yy = 0.0f;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) yy += (mj + s_shift[k * F + j]) * am[k];
                y[j] = yy;
                zm[j] = 0.0f;
            }
        }
        if constexpr (!BWD) {
#pragma unroll
            for (int j = 0; j < F; ++j) {
                float v = y[j];
                if (a.clampw > 0.0f) v = fminf(fmaxf(v, -a.clampw), a.clampw);
                a.decoded[r * F + j] = v;
            }
        } else {
            float gy[F], da[KM], dx[LD];
#pragma unroll
            for (int k = 0; k < KM; ++k) da[k] = 0.0f;
#pragma unroll
            for (int c = 0; c < LD; ++c) dx[c] = 0.0f;
            float asum = 0.0f;
#pragma unroll
            for (int k = 0; k < KM; ++k) asum += am[k];
#pragma unroll
            for (int j = 0; j < F; ++j) {
                float g = a.grad_decoded[r * F + j];
                if (a.clampw > 0.0f && !(y[j] >= -a.clampw && y[j] <= a.clampw)) g = 0.0f;
                gy[j] = g;
                if constexpr (DFT) {
                    float dzm = 0.0f;
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        if (k < K) {
                            const float sc = s_scale[k * F + j];
                            da[k] += g * (zm[j] * sc + s_shift[k * F + j]);
                            acc[k * SF + j] += g * am[k] * zm[j];            // grad_scale[k, 0, j]
                            acc[KM * SF + k * F + j] += g * am[k];           // grad_shift[k, j]
                            dzm += g * am[k] * sc;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < LD; ++c) dx[c] = fmaf(dzm, s_dft[c * F + j], dx[c]);
                } else {
                    const float dmixed = g * asum;
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        if (k < K) {
                            da[k] += g * (mixed[j] + s_shift[k * F + j]) + dmixed * u[k][j];
                            acc[KM * SF + k * F + j] += g * am[k];           // grad_shift[k, j]
                            const float du = dmixed * am[k];
#pragma unroll
                            for (int c = 0; c < LD; ++c) {
                                acc[k * SF + c * F + j] += x[c] * du;        // grad_scale[k, c, j]
                                dx[c] = fmaf(du, s_scale[k * SF + c * F + j], dx[c]);
                            }
                        }
                    }
                }
            }
            // selector gradient through the softmax (straight-through passes d(one-hot) to the soft weights unchanged)
            float dot = 0.0f;
#pragma unroll
            for (int k = 0; k < KM; ++k) dot += asoft[k] * da[k];
            if (a.grad_alpha) {
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) a.grad_alpha[(int64_t)k * a.rows + r] = asoft[k] * (da[k] - dot) / T;
            }
            if (a.grad_latent) {
#pragma unroll
                for (int c = 0; c < LD; ++c) a.grad_latent[r * LD + c] = dx[c] / s_div[c] * dq[c];
            }
        }
    }
    if constexpr (BWD) block_reduce_store<NRED>(acc, a.partials);
}

template <int LD, int F, bool DFT>
static hipError_t multi_launch(bool bwd, const MultiArgs &a, hipStream_t s) {
    constexpr int S = DFT ? 1 : LD, SF = S * F, NRED = kMultiMaxK * (SF + F);
    static_assert(NRED <= kMaxRed, "reduction width exceeds the partials workspace");
    const int blocks = grid_for(a.rows);
    const bool sga = a.uniforms != nullptr;
    if (!bwd) {
        if (sga) hipLaunchKernelGGL((multi_decode_kernel<LD, F, DFT, true, false>), dim3(blocks), dim3(kThreads), 0, s, a);
        else hipLaunchKernelGGL((multi_decode_kernel<LD, F, DFT, false, false>), dim3(blocks), dim3(kThreads), 0, s, a);
        return hipGetLastError();
    }
    if (sga) hipLaunchKernelGGL((multi_decode_kernel<LD, F, DFT, true, true>), dim3(blocks), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((multi_decode_kernel<LD, F, DFT, false, true>), dim3(blocks), dim3(kThreads), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // block partials: [KM][SF] scale part, then [KM][F] shift part; the first K entries of each are the outputs
    hipLaunchKernelGGL(finish_partials_kernel, dim3(a.K * SF), dim3(256), 0, s, a.partials, blocks, NRED,
                       (const float *)nullptr, a.grad_scale, a.K * SF, (float *)nullptr, 0, (float *)nullptr, 0);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.grad_shift) {
        hipLaunchKernelGGL(finish_partials_kernel, dim3(a.K * F), dim3(256), 0, s, a.partials + kMultiMaxK * SF, blocks,
                           NRED, (const float *)nullptr, a.grad_shift, a.K * F, (float *)nullptr, 0, (float *)nullptr, 0);
        e = hipGetLastError();
    }
    return e;
}

typedef hipError_t (*multi_fn)(bool, const MultiArgs &, hipStream_t);
static multi_fn multi_lookup(int ld, int f, bool dft) {
#define SHACIRA_MULTI(LD, F)                                    \
    if (ld == LD && f == F) return dft ? &multi_launch<LD, F, true> : &multi_launch<LD, F, false>;
    SHACIRA_MULTI(1, 2) SHACIRA_MULTI(2, 2) SHACIRA_MULTI(1, 4) SHACIRA_MULTI(2, 4) SHACIRA_MULTI(4, 2)
#undef SHACIRA_MULTI
    return nullptr;
}

bool latent_multi_supported(int ld, int f, int K) { return K >= 1 && K <= kMultiMaxK && multi_lookup(ld, f, false) != nullptr; }

hipError_t latent_multi_dispatch(bool bwd, int ld, int f, bool dft, const float *latent, const float *alpha, int K,
                                 const float *uniforms, float temperature, int straight_through, int diff_sampling,
                                 const float *div, const float *scale, const float *dftm, const float *shift,
                                 float clampw, int64_t rows, float *decoded, const float *grad_decoded,
                                 float *grad_latent, float *grad_alpha, float *grad_scale, float *grad_shift,
                                 double *partials, hipStream_t s) {
    MultiArgs a{};
    a.latent = latent; a.alpha = alpha; a.div = div; a.scale = scale; a.dft = dftm; a.shift = shift;
    a.uniforms = uniforms; a.grad_decoded = grad_decoded; a.decoded = decoded; a.grad_latent = grad_latent;
    a.grad_alpha = grad_alpha; a.grad_scale = grad_scale; a.grad_shift = grad_shift; a.partials = partials;
    a.rows = rows; a.K = K; a.straight_through = straight_through; a.diff_sampling = diff_sampling;
    a.temperature = temperature; a.clampw = clampw;
    return multi_lookup(ld, f, dft)(bwd, a, s);
}

typedef hipError_t (*decode_fn)(bool, const DecodeArgs &, hipStream_t);

static decode_fn decode_lookup(int ld, int f) {
#define SHACIRA_DEC(LD, F) \
    if (ld == LD && f == F) return &decode_launch<LD, F>;
    SHACIRA_DEC(1, 1) SHACIRA_DEC(1, 2) SHACIRA_DEC(1, 4) SHACIRA_DEC(1, 8)
    SHACIRA_DEC(2, 1) SHACIRA_DEC(2, 2) SHACIRA_DEC(2, 4) SHACIRA_DEC(2, 8)
    SHACIRA_DEC(3, 1) SHACIRA_DEC(3, 2) SHACIRA_DEC(3, 4) SHACIRA_DEC(3, 8)
    SHACIRA_DEC(4, 1) SHACIRA_DEC(4, 2) SHACIRA_DEC(4, 4) SHACIRA_DEC(4, 8)
    SHACIRA_DEC(8, 2) SHACIRA_DEC(8, 4) SHACIRA_DEC(8, 8)
#undef SHACIRA_DEC
    return nullptr;
}

static decode_levels_fn decode_levels_lookup(int ld, int f) {
#define SHACIRA_DECL(LD, F) \
    if (ld == LD && f == F) return &decode_levels_launch<LD, F>;
    SHACIRA_DECL(1, 1) SHACIRA_DECL(1, 2) SHACIRA_DECL(1, 4) SHACIRA_DECL(1, 8)
    SHACIRA_DECL(2, 1) SHACIRA_DECL(2, 2) SHACIRA_DECL(2, 4) SHACIRA_DECL(2, 8)
    SHACIRA_DECL(3, 1) SHACIRA_DECL(3, 2) SHACIRA_DECL(3, 4) SHACIRA_DECL(3, 8)
    SHACIRA_DECL(4, 1) SHACIRA_DECL(4, 2) SHACIRA_DECL(4, 4) SHACIRA_DECL(4, 8)
    SHACIRA_DECL(8, 2) SHACIRA_DECL(8, 4) SHACIRA_DECL(8, 8)
#undef SHACIRA_DECL
    return nullptr;
}

hipError_t latent_decode_levels_dispatch(bool bwd, int ld, int f, int levels, const int64_t *offsets,
                                         const DecodeArgs &a, hipStream_t s) {
    return decode_levels_lookup(ld, f)(bwd, levels, offsets, a, s);
}

bool latent_decode_supported(int ld, int f) { return decode_lookup(ld, f) != nullptr; }

hipError_t latent_decode_dispatch(bool bwd, int ld, int f, const DecodeArgs &a, hipStream_t s) {
    return decode_lookup(ld, f)(bwd, a, s);
}

// =========================================================================================================
// entropy bits
// =========================================================================================================
// params layout fp32 [4][3][LD]: layer k in {f1,f2,f3,f4}, slot {h,b,a}
template <int LD> struct CdfConsts {
    float sp[4][LD];   // softplus(h)
    float sgh[4][LD];  // d softplus / dh = sigmoid(h) (1 above torch's threshold 20)
    float b[4][LD];
    float ta[3][LD];   // tanh(a)
    __device__ __forceinline__ void load(const float *__restrict__ prm) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int c = 0; c < LD; ++c) {
                const float h = prm[(k * 3 + 0) * LD + c];
                // F.softplus(beta=1, threshold=20)
                sp[k][c] = (h > 20.0f) ? h : log1pf(expf(h));
                sgh[k][c] = (h > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-h));
                b[k][c] = prm[(k * 3 + 1) * LD + c];
                if (k < 3) ta[k][c] = tanhf(prm[(k * 3 + 2) * LD + c]);
            }
        }
    }
};

// 1 / (1 + exp(-x)) with the hardware reciprocal (1 ulp) instead of the IEEE division sequence; exp stays the accurate one
// (a probability is the DIFFERENCE of two of these: their absolute error is what the tails' gradients see)
__device__ __forceinline__ float sigmoidf_ref(float x) { return __frcp_rn(1.0f + expf(-x)); }

// tanh(u) = 1 - 2 / (exp(2u) + 1): one exp + one reciprocal instead of libm's ~120-instruction tanhf (the entropy backward was
// VALU-bound at 1 300 instructions per table row and 246 VGPRs). Absolute error ~1e-7; it enters the CDF through
// x = u + tanh(u) * tanh(a) and a sigmoid of slope <= 1/4, i.e. below the fp32 rounding of the CDF value itself.
__device__ __forceinline__ float tanhf_fast(float u) {
    const float e = __expf(2.0f * u);               // (hardware exp2: its error reaches tanh scaled by <= 1/2) inf -> 1, 0 -> -1
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}

// CDF with the chain's intermediates kept for the backward (NL = num_layers; layers used: first NL-1 of f1..f3, then f4)
template <int LD> struct CdfTrace {
    float xin[4];  // input of each applied layer (index = layer slot 0..2, 3 = final)
    float th[3];   // tanh(u) of each applied non-final layer
    float s;       // sigmoid output
};

template <int LD, int NL>
__device__ __forceinline__ float cdf_eval(const CdfConsts<LD> &p, int c, float x, CdfTrace<LD> &tr) {
    constexpr int nl = NL;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < nl - 1) {
            tr.xin[k] = x;
            const float u = x * p.sp[k][c] + p.b[k][c];
            const float th = tanhf_fast(u);
            tr.th[k] = th;
            x = u + th * p.ta[k][c];
        }
    }
    tr.xin[3] = x;
    tr.s = sigmoidf_ref(x * p.sp[3][c] + p.b[3][c]);
    return tr.s;
}

// back-propagates g = dL/d(cdf output); accumulates parameter grads into acc[(k*3+slot)*LD + c]; returns dL/dx
template <int LD, int NL>
__device__ __forceinline__ float cdf_backward(const CdfConsts<LD> &p, const float *__restrict__ prm, int c,
                                              const CdfTrace<LD> &tr, float g, float (&acc)[12 * LD]) {
    constexpr int nl = NL;
    float du = g * tr.s * (1.0f - tr.s);
    acc[(3 * 3 + 0) * LD + c] += du * tr.xin[3] * p.sgh[3][c];
    acc[(3 * 3 + 1) * LD + c] += du;
    float dx = du * p.sp[3][c];
#pragma unroll
    for (int k = 2; k >= 0; --k) {
        if (k < nl - 1) {
            const float th = tr.th[k];
            const float ta = p.ta[k][c];
            acc[(k * 3 + 2) * LD + c] += dx * th * (1.0f - ta * ta);
            du = dx * (1.0f + (1.0f - th * th) * ta);
            acc[(k * 3 + 0) * LD + c] += du * tr.xin[k] * p.sgh[k][c];
            acc[(k * 3 + 1) * LD + c] += du;
            dx = du * p.sp[k][c];
        }
    }
    (void)prm;
    return dx;
}

constexpr float kInvLn2 = 1.4426950408889634f;

template <int LD, int NL>
__global__ __launch_bounds__(kThreads) void entropy_fwd_kernel(const float *__restrict__ latent,
                                                               const float *__restrict__ noise,
                                                               const float *__restrict__ prm,
                                                               double *__restrict__ partials, int64_t rows) {
    CdfConsts<LD> p;
    p.load(prm);
    float acc[1] = {0.0f};
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < rows; r += stride) {
#pragma unroll
        for (int c = 0; c < LD; ++c) {
            const float v = latent[r * LD + c];
            const float w = noise ? (v + noise[r * LD + c]) : rintf(v);
            CdfTrace<LD> tp, tn;
            const float prob = cdf_eval<LD, NL>(p, c, w + 0.5f, tp) - cdf_eval<LD, NL>(p, c, w - 0.5f, tn);
            float bits = -__log2f(prob + 1e-10f);      // hardware log2, 1 ulp
            bits = fminf(fmaxf(bits, 0.0f), 50.0f);
            acc[0] += bits;
        }
    }
    block_reduce_store<1>(acc, partials);
}

template <int LD, int NL>
__global__ __launch_bounds__(kThreads) void entropy_bwd_kernel(const float *__restrict__ latent,
                                                               const float *__restrict__ noise,
                                                               const float *__restrict__ prm,
                                                               const float *__restrict__ grad_total,
                                                               float *__restrict__ grad_latent,
                                                               double *__restrict__ partials, int64_t rows) {
    CdfConsts<LD> p;
    p.load(prm);
    float acc[12 * LD];
#pragma unroll
    for (int q = 0; q < 12 * LD; ++q) acc[q] = 0.0f;
    const float gt = grad_total[0];
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < rows; r += stride) {
#pragma unroll
        for (int c = 0; c < LD; ++c) {
            const float v = latent[r * LD + c];
            const float w = noise ? (v + noise[r * LD + c]) : rintf(v);
            CdfTrace<LD> tp, tn;
            const float prob = cdf_eval<LD, NL>(p, c, w + 0.5f, tp) - cdf_eval<LD, NL>(p, c, w - 0.5f, tn);
            const float q = prob + 1e-10f;
            const float bits = -__log2f(q);
            // clamp(., 0, 50) passes the gradient on the closed interval; parameter grads are scaled by gt at the end
            const float gp = (bits >= 0.0f && bits <= 50.0f) ? -kInvLn2 * __frcp_rn(q) : 0.0f;
            const float dxp = cdf_backward<LD, NL>(p, prm, c, tp, gp, acc);
            const float dxn = cdf_backward<LD, NL>(p, prm, c, tn, -gp, acc);
            if (grad_latent) grad_latent[r * LD + c] = noise ? gt * (dxp + dxn) : 0.0f;  // round(): zero gradient
        }
    }
    block_reduce_store<12 * LD>(acc, partials);
}

template <int LD, int NL> static hipError_t entropy_launch_nl(bool bwd, const EntropyArgs &a, hipStream_t s) {
    const int blocks = grid_for(a.rows);
    if (!bwd) {
        hipLaunchKernelGGL((entropy_fwd_kernel<LD, NL>), dim3(blocks), dim3(kThreads), 0, s, a.latent, a.noise, a.params,
                           a.partials, a.rows);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(finish_partials_kernel, dim3(1), dim3(256), 0, s, a.partials, blocks, 1,
                           (const float *)nullptr, a.total_bits, 1, (float *)nullptr, 0, (float *)nullptr, 0);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((entropy_bwd_kernel<LD, NL>), dim3(blocks), dim3(kThreads), 0, s, a.latent, a.noise, a.params,
                       a.grad_total, a.grad_latent, a.partials, a.rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.grad_params) {
        hipLaunchKernelGGL(finish_partials_kernel, dim3(12 * LD), dim3(256), 0, s, a.partials, blocks, 12 * LD,
                           a.grad_total, a.grad_params, 12 * LD, (float *)nullptr, 0, (float *)nullptr, 0);
        e = hipGetLastError();
    }
    return e;
}

// the number of Bitparm layers is a template parameter: the unused layers' code (and registers) are gone at compile time
template <int LD> static hipError_t entropy_launch(bool bwd, const EntropyArgs &a, hipStream_t s) {
    switch (a.num_layers) {
        case 1: return entropy_launch_nl<LD, 1>(bwd, a, s);
        case 2: return entropy_launch_nl<LD, 2>(bwd, a, s);
        case 3: return entropy_launch_nl<LD, 3>(bwd, a, s);
        case 4: return entropy_launch_nl<LD, 4>(bwd, a, s);
        default: return hipErrorInvalidValue;
    }
}

bool entropy_supported(int ld) { return ld == 1 || ld == 2 || ld == 3 || ld == 4 || ld == 8; }

hipError_t entropy_dispatch(bool bwd, int ld, const EntropyArgs &a, hipStream_t s) {
    switch (ld) {
        case 1: return entropy_launch<1>(bwd, a, s);
        case 2: return entropy_launch<2>(bwd, a, s);
        case 3: return entropy_launch<3>(bwd, a, s);
        case 4: return entropy_launch<4>(bwd, a, s);
        default: return entropy_launch<8>(bwd, a, s);
    }
}

}  // namespace shacira
