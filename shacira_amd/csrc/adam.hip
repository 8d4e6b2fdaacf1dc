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
