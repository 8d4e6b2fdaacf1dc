// adam.hip -- fused Adam step over a flat fp32 parameter buffer (gfx950). "Next" row f1 of SURVEY.md section 8:
// the step that immediately follows the hash-grid backward in the reference's trainers is torch.optim.Adam over the
// whole table (wisp/trainers/base_trainer.py:206-266 builds the groups, image_trainer.py:355-359 steps), i.e. the
// multi-pass foreach implementation: ~10 elementwise kernels over 48.8 MB each. Here it is ONE streaming pass
// (reads p, g, m, v; writes p, m, v; optionally zeroes g for the next accumulation): 7-8 x 4 bytes per element,
// HBM-bound.
//
// Arithmetic follows torch.optim.Adam (amsgrad=False, maximize=False, L2 weight decay folded into the gradient):
//   g' = g + wd*p ; m = b1*m + (1-b1)*g' ; v = b2*v + (1-b2)*g'*g'
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <cmath>

#include "internal.h"

namespace shacira {

constexpr int kAdamMaxTensors = 32;

__global__ __launch_bounds__(256) void adam_step_kernel(float *__restrict__ p, float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                        float lr_over_bc1, float b1, float b2, float inv_sqrt_bc2,
                                                        float eps, float wd, int zero_grad, float lr,
                                                        const int32_t *__restrict__ step_dev) {
    if (step_dev) {  // graph-capturable form: the step count lives on the device, corrections computed here
        const double t = (double)step_dev[0];
        lr_over_bc1 = (float)((double)lr / (1.0 - pow((double)b1, t)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4 *>(p + i);
            float4 gg = *reinterpret_cast<float4 *>(g + i);
            float4 mm = *reinterpret_cast<float4 *>(m + i);
            float4 vv = *reinterpret_cast<float4 *>(v + i);
            float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gr = ga[k] + wd * pa[k];
                ma[k] = b1 * ma[k] + (1.0f - b1) * gr;
                va[k] = b2 * va[k] + (1.0f - b2) * gr * gr;
                const float denom = sqrtf(va[k]) * inv_sqrt_bc2 + eps;
                pa[k] = pa[k] - lr_over_bc1 * (ma[k] / denom);
            }
            *reinterpret_cast<float4 *>(p + i) = pp;
            *reinterpret_cast<float4 *>(m + i) = mm;
            *reinterpret_cast<float4 *>(v + i) = vv;
            if (zero_grad) *reinterpret_cast<float4 *>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int64_t j = i; j < n; ++j) {
                const float gr = g[j] + wd * p[j];
                const float mj = b1 * m[j] + (1.0f - b1) * gr;
                const float vj = b2 * v[j] + (1.0f - b2) * gr * gr;
                const float denom = sqrtf(vj) * inv_sqrt_bc2 + eps;
                p[j] = p[j] - lr_over_bc1 * (mj / denom);
                m[j] = mj;
                v[j] = vj;
                if (zero_grad) g[j] = 0.0f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Multi-tensor form: up to kAdamMaxTensors parameters (each with its own lr / weight decay) in ONE launch; the
// pointer table travels in the kernarg segment. Block b works on chunk (b - first_block[t]) of tensor t.
struct AdamMulti {
    float *p[kAdamMaxTensors], *g[kAdamMaxTensors], *m[kAdamMaxTensors], *v[kAdamMaxTensors];
    int64_t n[kAdamMaxTensors];
    float lr[kAdamMaxTensors], wd[kAdamMaxTensors];
    uint32_t first_block[kAdamMaxTensors + 1];
    int count;
};

constexpr int kAdamChunk = 256 * 4 * 8;  // elements per block

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMulti a, float b1, float b2, float eps, int step,
                                                         const int32_t *__restrict__ step_dev, int zero_grad) {
    int t = 0;
    while (t + 1 < a.count && a.first_block[t + 1] <= blockIdx.x) ++t;
    const double tt = step_dev ? (double)step_dev[0] : (double)step;
    const float lr_over_bc1 = (float)((double)a.lr[t] / (1.0 - pow((double)b1, tt)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, tt)));
    const float wd = a.wd[t];
    float *p = a.p[t], *g = a.g[t], *m = a.m[t], *v = a.v[t];
    const int64_t n = a.n[t];
    const int64_t lo = (int64_t)(blockIdx.x - a.first_block[t]) * kAdamChunk;
    const int64_t hi = (lo + kAdamChunk < n) ? lo + kAdamChunk : n;
    const bool aligned = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    int64_t j = lo;
    if (aligned) {  // full float4 groups of this chunk (lo is a multiple of the chunk size)
        const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
        for (j = lo + (int64_t)threadIdx.x * 4; j < hi4; j += 256 * 4) {
            float4 pp = *reinterpret_cast<float4 *>(p + j);
            float4 gg = *reinterpret_cast<float4 *>(g + j);
            float4 mm = *reinterpret_cast<float4 *>(m + j);
            float4 vv = *reinterpret_cast<float4 *>(v + j);
            float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gr = ga[k] + wd * pa[k];
                ma[k] = b1 * ma[k] + (1.0f - b1) * gr;
                va[k] = b2 * va[k] + (1.0f - b2) * gr * gr;
                pa[k] = pa[k] - lr_over_bc1 * (ma[k] / (sqrtf(va[k]) * inv_sqrt_bc2 + eps));
            }
            *reinterpret_cast<float4 *>(p + j) = pp;
            *reinterpret_cast<float4 *>(m + j) = mm;
            *reinterpret_cast<float4 *>(v + j) = vv;
            if (zero_grad) *reinterpret_cast<float4 *>(g + j) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        j = hi4;
    }
    for (j += threadIdx.x; j < hi; j += 256) {
        const float gr = g[j] + wd * p[j];
        const float mj = b1 * m[j] + (1.0f - b1) * gr;
        const float vj = b2 * v[j] + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vj) * inv_sqrt_bc2 + eps;
        p[j] = p[j] - lr_over_bc1 * (mj / denom);
        m[j] = mj;
        v[j] = vj;
        if (zero_grad) g[j] = 0.0f;
    }
}

hipError_t adam_multi_launch(int count, float *const *p, float *const *g, float *const *m, float *const *v,
                             const int64_t *n, const float *lr, const float *wd, float b1, float b2, float eps,
                             int step, const int32_t *step_dev, int zero_grad, hipStream_t s) {
    AdamMulti a;
    uint32_t blocks = 0;
    a.count = count;
    for (int t = 0; t < count; ++t) {
        a.p[t] = p[t]; a.g[t] = g[t]; a.m[t] = m[t]; a.v[t] = v[t];
        a.n[t] = n[t]; a.lr[t] = lr[t]; a.wd[t] = wd[t];
        a.first_block[t] = blocks;
        blocks += (uint32_t)((n[t] + kAdamChunk - 1) / kAdamChunk);
    }
    a.first_block[count] = blocks;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, s, a, b1, b2, eps, step, step_dev, zero_grad);
    return hipGetLastError();
}

hipError_t adam_step_launch(float *p, float *g, float *m, float *v, int64_t n, float lr, float b1, float b2, float eps,
                            float wd, int step, const int32_t *step_dev, int zero_grad, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const double bc1 = step_dev ? 1.0 : 1.0 - std::pow((double)b1, (double)step);
    const double bc2 = step_dev ? 1.0 : 1.0 - std::pow((double)b2, (double)step);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_step_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, p, g, m, v, n, (float)(lr / bc1), b1,
                       b2, (float)(1.0 / std::sqrt(bc2)), eps, wd, zero_grad, lr, step_dev);
    return hipGetLastError();
}

}  // namespace shacira
