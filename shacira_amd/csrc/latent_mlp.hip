// latent_mlp.hip -- latent quantise-and-decode with HIDDEN decoder layers and activations (gfx950).
//
// The reference's LatentDecoder with num_layers_dec > 0 and / or activations (wisp/models/latent_decoders/
// basic_latent_decoder.py:97-198: layers :139-147, forward :182-198, DecoderLayer.forward :86-91) is a per-row MLP over the
// whole table, evaluated there as a chain of ATen kernels per layer:
//     decoded = clamp(final_act(L_n(act(... act(L_1(q(latent) / div)) ...)))),   L_k(x) = x @ W_k + b_k
// with q = round (straight-through) or the SGA sample, W_k the layer's effective matrix (`scale`, or `dft * scale`) and
// b_k its `shift` (zeros without one). Here it is ONE pass over the table each way, like the affine decoder of latent.hip:
// one thread per row, the layer matrices in LDS (zero-padded to W x W, W = 4 / 8 / 16 = the widest layer rounded up), the
// row's activations in registers. The backward recomputes the forward, back-propagates in registers and reduces the
// parameter gradients layer by layer through LDS: the workgroup's 256 rows of (layer input, gradient at the layer output)
// are staged, thread (i, j) sums x[r][i] * g[r][j] over the rows in a fixed order and keeps the sum across the tiles it
// walks; fp64 block partials + one finishing kernel make the reduction over the table bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "internal.h"

namespace shacira {

namespace {

constexpr int kLmlpThreads = 256;
constexpr int kLmlpMaxBlocks = 512;

enum Act { kNone = 0, kSigmoid = 1, kTanh = 2, kRelu = 3, kSine = 4 };

// y = act(z), d = act'(z)
__device__ __forceinline__ void activate(int act, float z, float &y, float &d) {
    switch (act) {
    case kSigmoid: y = 1.0f / (1.0f + expf(-z)); d = y * (1.0f - y); break;
    case kTanh: y = tanhf(z); d = 1.0f - y * y; break;
    case kRelu: y = z > 0.0f ? z : 0.0f; d = z > 0.0f ? 1.0f : 0.0f; break;
    case kSine: y = sinf(30.0f * z); d = 30.0f * cosf(30.0f * z); break;   // SineScaled(30.0): activations.py
    default: y = z; d = 1.0f; break;
    }
}

// same SGA sample as latent.hip (basic_latent_decoder.py:183-191): see sga_quantise there
__device__ __forceinline__ void sga_sample(float w, float u0, float u1, float T, bool diff, float &q, float &dq) {
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

struct MlpShape {
    int32_t nl;                                   // layers (hidden + 1)
    int32_t w[SHACIRA_LATENT_MLP_MAX_LAYERS + 1]; // widths: latent_dim, hidden..., feature_dim
    int32_t off[SHACIRA_LATENT_MLP_MAX_LAYERS];   // offset of layer k's block {W_k [in x out], b_k [out]} in the packed parameters
    int32_t total;                                // packed parameter count
    int32_t act, final_act;
};

template <int W> struct MlpLds {
    float Wm[SHACIRA_LATENT_MLP_MAX_LAYERS][W * W];   // [i][j], zero beyond (in_k, out_k)
    float b[SHACIRA_LATENT_MLP_MAX_LAYERS][W];
    float div[W];
};

template <int W>
__device__ __forceinline__ void load_params(MlpLds<W> &s, const MlpShape &sh, const float *__restrict__ params,
                                            const float *__restrict__ div) {
    for (int k = 0; k < SHACIRA_LATENT_MLP_MAX_LAYERS; ++k) {
        const int in = k < sh.nl ? sh.w[k] : 0, out = k < sh.nl ? sh.w[k + 1] : 0;
        for (int e = threadIdx.x; e < W * W; e += kLmlpThreads) {
            const int i = e / W, j = e % W;
            s.Wm[k][e] = (i < in && j < out) ? params[sh.off[k] + i * out + j] : 0.0f;
        }
        for (int j = threadIdx.x; j < W; j += kLmlpThreads) s.b[k][j] = (j < out) ? params[sh.off[k] + in * out + j] : 0.0f;
    }
    for (int c = threadIdx.x; c < W; c += kLmlpThreads) s.div[c] = (c < sh.w[0]) ? div[c] : 1.0f;
    __syncthreads();
}

// quantised, normalised latents of row r -> x[0..W) (zero beyond latent_dim), dq = d q / d latent
template <int W, bool SGA>
__device__ __forceinline__ void load_row(const MlpLds<W> &s, const MlpShape &sh, const float *__restrict__ latent,
                                         const float *__restrict__ uniforms, float T, bool diff, int64_t r, bool live,
                                         float (&x)[W], float (&dq)[W]) {
    const int ld = sh.w[0];
#pragma unroll
    for (int c = 0; c < W; ++c) {
        x[c] = 0.0f;
        dq[c] = 0.0f;
        if (c < ld && live) {
            const float wv = latent[r * ld + c];
            float q;
            if constexpr (SGA) {
                const float2 u = *reinterpret_cast<const float2 *>(uniforms + (r * ld + c) * 2);
                sga_sample(wv, u.x, u.y, T, diff, q, dq[c]);
            } else {
                q = rintf(wv);   // torch.round: half to even; straight-through gradient
                dq[c] = 1.0f;
            }
            x[c] = q / s.div[c];
        }
    }
}

template <int W>
__device__ __forceinline__ void layer_forward(const float *__restrict__ Wm, const float *__restrict__ b,
                                              const float (&x)[W], float (&z)[W]) {
#pragma unroll
    for (int j = 0; j < W; ++j) z[j] = b[j];
#pragma unroll
    for (int i = 0; i < W; ++i) {
#pragma unroll
        for (int j = 0; j < W; ++j) z[j] = fmaf(x[i], Wm[i * W + j], z[j]);
    }
}

template <int W, bool SGA>
__global__ __launch_bounds__(kLmlpThreads) void latent_mlp_fwd_kernel(MlpShape sh, const float *__restrict__ latent,
                                                                     const float *__restrict__ uniforms, float T, int diff,
                                                                     const float *__restrict__ div,
                                                                     const float *__restrict__ params, float clampw,
                                                                     float *__restrict__ decoded, int64_t rows) {
    __shared__ MlpLds<W> s;
    load_params<W>(s, sh, params, div);
    const int F = sh.w[sh.nl];
    const int64_t stride = (int64_t)gridDim.x * kLmlpThreads;
    for (int64_t r = (int64_t)blockIdx.x * kLmlpThreads + threadIdx.x; r < rows; r += stride) {
        float x[W], dq[W], z[W];
        load_row<W, SGA>(s, sh, latent, uniforms, T, diff != 0, r, true, x, dq);
#pragma unroll
        for (int k = 0; k < SHACIRA_LATENT_MLP_MAX_LAYERS; ++k) {
            if (k < sh.nl) {
                layer_forward<W>(s.Wm[k], s.b[k], x, z);
                const int a = (k + 1 < sh.nl) ? sh.act : sh.final_act;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    float d;
                    activate(a, z[j], x[j], d);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (j < F) {
                float v = x[j];
                if (clampw > 0.0f) v = fminf(fmaxf(v, -clampw), clampw);
                decoded[r * F + j] = v;
            }
        }
    }
}

template <int W, bool SGA>
__global__ __launch_bounds__(kLmlpThreads) void latent_mlp_bwd_kernel(MlpShape sh, const float *__restrict__ latent,
                                                                     const float *__restrict__ uniforms, float T, int diff,
                                                                     const float *__restrict__ div,
                                                                     const float *__restrict__ params, float clampw,
                                                                     const float *__restrict__ grad_decoded,
                                                                     float *__restrict__ grad_latent,
                                                                     double *__restrict__ partials, int64_t rows) {
    constexpr int NL = SHACIRA_LATENT_MLP_MAX_LAYERS;
    constexpr int EPT = (W * W + kLmlpThreads - 1) / kLmlpThreads;   // matrix elements per thread (1 for W <= 16)
    static_assert(EPT == 1, "one matrix element per thread");
    __shared__ MlpLds<W> s;
    __shared__ float s_x[kLmlpThreads * W], s_g[kLmlpThreads * W];
    load_params<W>(s, sh, params, div);
    const int F = sh.w[sh.nl], ld = sh.w[0];
    float accW[NL], accB[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) accW[k] = accB[k] = 0.0f;
    const int ei = threadIdx.x / W, ej = threadIdx.x % W;   // this thread's matrix element (threads >= W * W idle in phase B)
    const int64_t ntiles = (rows + kLmlpThreads - 1) / kLmlpThreads;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r = tile * kLmlpThreads + threadIdx.x;
        const bool live = r < rows;
        // ---- forward, keeping every layer's input X[k] and the activation derivative D[k] at its output
        float X[NL][W], D[NL][W], dq[W], z[W];
        load_row<W, SGA>(s, sh, latent, uniforms, T, diff != 0, live ? r : 0, live, X[0], dq);
        float yout[W];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            if (k < sh.nl) {
                layer_forward<W>(s.Wm[k], s.b[k], X[k], z);
                const int a = (k + 1 < sh.nl) ? sh.act : sh.final_act;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    float y;
                    activate(a, z[j], y, D[k][j]);
                    if (k + 1 < NL) {
                        if (k + 1 < sh.nl) X[k + 1][j] = y;
                    }
                    if (k + 1 == sh.nl) yout[j] = y;
                }
            }
        }
        // ---- gradient at the last layer's output (pre-activation): clamp mask, final activation
        float G[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            float g = 0.0f;
            if (j < F && live) {
                g = grad_decoded[r * F + j];
                // torch.clamp passes the gradient where -c <= y <= c
                if (clampw > 0.0f && !(yout[j] >= -clampw && yout[j] <= clampw)) g = 0.0f;
            }
            G[j] = g;
        }
        // ---- layers in reverse: G = gradient wrt the layer's pre-activation output
#pragma unroll
        for (int kk = 0; kk < NL; ++kk) {
            const int k = NL - 1 - kk;
            if (k < sh.nl) {
#pragma unroll
                for (int j = 0; j < W; ++j) G[j] *= D[k][j];
                // parameter gradients of layer k through LDS: rows of the tile in a fixed order
                __syncthreads();   // the previous layer's sums are done with s_x / s_g
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    s_x[threadIdx.x * W + c] = live ? X[k][c] : 0.0f;
                    s_g[threadIdx.x * W + c] = G[c];
                }
                __syncthreads();
                if (threadIdx.x < W * W) {
                    float sw = 0.0f, sb = 0.0f;
                    for (int rr = 0; rr < kLmlpThreads; ++rr) {
                        const float gv = s_g[rr * W + ej];
                        sw = fmaf(s_x[rr * W + ei], gv, sw);
                        sb += gv;
                    }
                    accW[k] += sw;
                    if (ei == 0) accB[k] += sb;
                }
                // gradient wrt the layer's input
                float gx[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    float a = 0.0f;
#pragma unroll
                    for (int j = 0; j < W; ++j) a = fmaf(G[j], s.Wm[k][i * W + j], a);
                    gx[i] = a;
                }
#pragma unroll
                for (int i = 0; i < W; ++i) G[i] = gx[i];
            }
        }
        if (grad_latent != nullptr && live) {
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (c < ld) grad_latent[r * ld + c] = G[c] / s.div[c] * dq[c];
        }
    }
    // ---- block partials, packed like the parameters
    double *out = partials + (size_t)blockIdx.x * sh.total;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        if (k < sh.nl && threadIdx.x < W * W) {
            const int in = sh.w[k], o = sh.w[k + 1];
            if (ei < in && ej < o) out[sh.off[k] + ei * o + ej] = (double)accW[k];
            if (ei == 0 && ej < o) out[sh.off[k] + in * o + ej] = (double)accB[k];
        }
    }
}

// out[q] = sum over blocks of partials[b][q], fixed tree (one workgroup per parameter)
__global__ __launch_bounds__(256) void latent_mlp_finish_kernel(const double *__restrict__ partials, int nblocks, int total,
                                                                float *__restrict__ out) {
    __shared__ double s_w[4];
    const int q = blockIdx.x;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) v += partials[(size_t)b * total + q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[q] = (float)((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

bool make_shape(int num_layers, const int32_t *widths, int act, int final_act, MlpShape &sh, int &wmax) {
    if (widths == nullptr || num_layers < 1 || num_layers > SHACIRA_LATENT_MLP_MAX_LAYERS) return false;
    if (act < kNone || act > kSine || final_act < kNone || final_act > kSine) return false;
    sh.nl = num_layers;
    sh.act = act;
    sh.final_act = final_act;
    wmax = 0;
    int off = 0;
    for (int k = 0; k <= num_layers; ++k) {
        if (widths[k] < 1 || widths[k] > SHACIRA_LATENT_MLP_MAX_WIDTH) return false;
        sh.w[k] = widths[k];
        if (widths[k] > wmax) wmax = widths[k];
    }
    for (int k = num_layers + 1; k <= SHACIRA_LATENT_MLP_MAX_LAYERS; ++k) sh.w[k] = 0;
    for (int k = 0; k < SHACIRA_LATENT_MLP_MAX_LAYERS; ++k) {
        sh.off[k] = off;
        if (k < num_layers) off += widths[k] * widths[k + 1] + widths[k + 1];
    }
    sh.total = off;
    return true;
}

int grid_for_rows(int64_t rows) {
    int64_t b = (rows + kLmlpThreads - 1) / kLmlpThreads;
    if (b > kLmlpMaxBlocks) b = kLmlpMaxBlocks;
    return b < 1 ? 1 : (int)b;
}

}  // namespace

bool latent_mlp_supported(int num_layers, const int32_t *widths) {
    MlpShape sh;
    int wmax;
    return make_shape(num_layers, widths, 0, 0, sh, wmax);
}

size_t latent_mlp_workspace_bytes(int num_layers, const int32_t *widths) {
    MlpShape sh;
    int wmax;
    if (!make_shape(num_layers, widths, 0, 0, sh, wmax)) return 0;
    return (size_t)kLmlpMaxBlocks * sh.total * sizeof(double);
}

hipError_t latent_mlp_dispatch(bool bwd, const LatentMlpArgs &a, hipStream_t s) {
    MlpShape sh;
    int wmax;
    if (!make_shape(a.num_layers, a.widths, a.act, a.final_act, sh, wmax)) return hipErrorInvalidValue;
    if (a.rows <= 0) {
        if (bwd && a.grad_params != nullptr) return zero_fill_async(a.grad_params, sh.total, s);
        return hipSuccess;
    }
    const int blocks = grid_for_rows(a.rows);
    const bool sga = a.uniforms != nullptr;
#define SHACIRA_MLP_FWD(WW, SG)                                                                                          \
    hipLaunchKernelGGL((latent_mlp_fwd_kernel<WW, SG>), dim3(blocks), dim3(kLmlpThreads), 0, s, sh, a.latent, a.uniforms, \
                       a.temperature, a.diff_sampling, a.div, a.params, a.clampw, a.decoded, a.rows)
#define SHACIRA_MLP_BWD(WW, SG)                                                                                          \
    hipLaunchKernelGGL((latent_mlp_bwd_kernel<WW, SG>), dim3(blocks), dim3(kLmlpThreads), 0, s, sh, a.latent, a.uniforms, \
                       a.temperature, a.diff_sampling, a.div, a.params, a.clampw, a.grad_decoded, a.grad_latent,        \
                       a.partials, a.rows)
    if (!bwd) {
        if (wmax <= 4) { if (sga) SHACIRA_MLP_FWD(4, true); else SHACIRA_MLP_FWD(4, false); }
        else if (wmax <= 8) { if (sga) SHACIRA_MLP_FWD(8, true); else SHACIRA_MLP_FWD(8, false); }
        else { if (sga) SHACIRA_MLP_FWD(16, true); else SHACIRA_MLP_FWD(16, false); }
        return hipGetLastError();
    }
    if (wmax <= 4) { if (sga) SHACIRA_MLP_BWD(4, true); else SHACIRA_MLP_BWD(4, false); }
    else if (wmax <= 8) { if (sga) SHACIRA_MLP_BWD(8, true); else SHACIRA_MLP_BWD(8, false); }
    else { if (sga) SHACIRA_MLP_BWD(16, true); else SHACIRA_MLP_BWD(16, false); }
#undef SHACIRA_MLP_FWD
#undef SHACIRA_MLP_BWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(latent_mlp_finish_kernel, dim3(sh.total), dim3(256), 0, s, a.partials, blocks, sh.total,
                       a.grad_params);
    return hipGetLastError();
}

}  // namespace shacira
