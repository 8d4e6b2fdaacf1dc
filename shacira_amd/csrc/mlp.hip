// mlp.hip -- the tiny colour/density decoder MLP that consumes the hash-grid features, fused (gfx950).
//
// Row a14 of SURVEY.md section 8: BasicDecoder.forward (wisp/models/decoders/basic_decoders.py:74-101) as used by
// NeuralImage.rgb (wisp/models/nefs/image.py:127-154): NH hidden Linear(+bias)+ReLU layers of width H, then a linear
// `lout`. In the reference (and in torch on this GPU) it is 3 GEMMs + activations forward and 6 skinny GEMMs
// backward; the weight-gradient GEMMs have K = N (the whole pixel batch) and 16x32 outputs and run at 0.5-0.7 ms each
// through hipBLASLt -- 70 % of the whole image-fit step (profiles/r01_imagefit_step.txt). Here:
//
//   forward   one thread per sample, weights broadcast from LDS, activations in registers: reads x [N, IN], writes
//             y [N, OUT]                                                  -> HBM-bound, 4*(IN+OUT) bytes per sample
//   backward  per 256-sample tile: (1) one thread per sample recomputes the hidden activations, back-propagates
//             dy -> dh2 -> dh1 -> dx in registers and leaves the tile's (input, upstream-gradient) vectors in LDS
//             element-major; (2) one thread per weight element accumulates its dW over the tile from LDS (this is the
//             dense contraction: 851 outputs x 256-deep per tile, fp32 FMAs); block partials in fp64 -> finish kernel.
//             Reads x, dy; writes dx                                      -> 4*(2*IN+OUT) bytes per sample
// fp32 throughout (the reference runs this MLP in fp32: kodak.yaml disables AMP). MFMA is not used: fp32-input MFMA
// runs at the vector rate on gfx950 and the op is memory-bound; bf16 MFMA would break fp32 parity.
//
// params layout (one flat fp32 buffer, also the layout of grad_params):
//   W1 [H, IN] row-major (nn.Linear.weight), b1 [H], W2 [H, H], b2 [H], ... (NH hidden layers), Wout [OUT, H], bout [OUT]
#include <mutex>

#include "internal.h"

namespace shacira {

constexpr int kMlpTile = 128;  // samples per tile == threads per block

template <int IN, int H, int NH, int OUT> struct MlpShape {
    static constexpr int w_off(int layer) {  // offset of W of hidden layer `layer` (0-based); layer == NH: output layer
        int off = 0;
        for (int l = 0; l < layer; ++l) off += (l == 0 ? IN : H) * H + H;
        return off;
    }
    static constexpr int n_params = w_off(NH) + OUT * H + OUT;
};

// forward of one sample held in registers; weights read from LDS (wave-uniform addresses -> broadcast)
template <int IN, int H, int NH, int OUT>
__device__ __forceinline__ void mlp_forward_regs(const float *__restrict__ sw, const float (&x)[IN], float (&h)[NH][H],
                                                 float (&y)[OUT]) {
    using S = MlpShape<IN, H, NH, OUT>;
#pragma unroll
    for (int l = 0; l < NH; ++l) {
        const float *W = sw + S::w_off(l);
        const int fan_in = (l == 0) ? IN : H;
        const float *b = W + fan_in * H;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            float acc = b[j];
            if (l == 0) {
#pragma unroll
                for (int i = 0; i < IN; ++i) acc = fmaf(x[i], W[j * IN + i], acc);
            } else {
#pragma unroll
                for (int i = 0; i < H; ++i) acc = fmaf(h[l - 1][i], W[j * H + i], acc);
            }
            h[l][j] = fmaxf(acc, 0.0f);
        }
    }
    const float *Wo = sw + S::w_off(NH);
    const float *bo = Wo + OUT * H;
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        float acc = bo[o];
#pragma unroll
        for (int i = 0; i < H; ++i) acc = fmaf(h[NH - 1][i], Wo[o * H + i], acc);
        y[o] = acc;
    }
}

template <int IN> __device__ __forceinline__ void load_row_in(const float *__restrict__ p, float (&x)[IN]) {
    if constexpr (IN % 4 == 0) {
#pragma unroll
        for (int q = 0; q < IN / 4; ++q) {
            const float4 v = reinterpret_cast<const float4 *>(p)[q];
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < IN; ++i) x[i] = p[i];
    }
}

template <int IN, int H, int NH, int OUT>
__global__ __launch_bounds__(kMlpTile) void mlp_forward_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ params,
                                                               float *__restrict__ y, int64_t N) {
    using S = MlpShape<IN, H, NH, OUT>;
    __shared__ float sw[S::n_params];
    for (int e = threadIdx.x; e < S::n_params; e += kMlpTile) sw[e] = params[e];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kMlpTile;
    for (int64_t s = (int64_t)blockIdx.x * kMlpTile + threadIdx.x; s < N; s += stride) {
        float xr[IN], h[NH][H], yr[OUT];
        load_row_in<IN>(x + s * IN, xr);
        mlp_forward_regs<IN, H, NH, OUT>(sw, xr, h, yr);
#pragma unroll
        for (int o = 0; o < OUT; ++o) y[s * OUT + o] = yr[o];
    }
}

// Backward. LDS: weights + one element-major image per layer of (layer input, upstream gradient of its pre-activation).
template <int IN, int H, int NH, int OUT>
__global__ __launch_bounds__(kMlpTile) void mlp_backward_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ params,
                                                                const float *__restrict__ gy,
                                                                float *__restrict__ gx, double *__restrict__ partials,
                                                                int64_t N, int tiles_per_block) {
    using S = MlpShape<IN, H, NH, OUT>;
    constexpr int P = kMlpTile + 4;  // row pitch of the element-major images (conflict-free ds_read_b128)
    // images: input of every layer (x, h_0 .. h_{NH-1}) and pre-activation gradient of every layer (d_0 .. d_{NH-1}, dy)
    constexpr int kInRows = IN + NH * H;
    extern __shared__ __align__(16) float smem[];
    float *sw = smem;                                   // [n_params]
    float *s_in = smem + (S::n_params + 3) / 4 * 4;     // [kInRows][P]
    float *s_g = s_in + kInRows * P;                    // [NH * H + OUT][P]
    for (int e = threadIdx.x; e < S::n_params; e += kMlpTile) sw[e] = params[e];

    // each thread owns up to EPT weight/bias elements e = threadIdx.x + k*256
    constexpr int EPT = (S::n_params + kMlpTile - 1) / kMlpTile;
    float acc[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) acc[k] = 0.0f;
    __syncthreads();

    for (int t = 0; t < tiles_per_block; ++t) {
        const int64_t s = ((int64_t)blockIdx.x * tiles_per_block + t) * kMlpTile + threadIdx.x;
        const bool live = s < N;
        // ---- phase 1: per-sample forward recompute + backward chain in registers
        float xr[IN], h[NH][H], yr[OUT], d[OUT];
        if (live) {
            load_row_in<IN>(x + s * IN, xr);
#pragma unroll
            for (int o = 0; o < OUT; ++o) d[o] = gy[s * OUT + o];
        } else {
#pragma unroll
            for (int i = 0; i < IN; ++i) xr[i] = 0.0f;
#pragma unroll
            for (int o = 0; o < OUT; ++o) d[o] = 0.0f;
        }
        mlp_forward_regs<IN, H, NH, OUT>(sw, xr, h, yr);
        const int c = threadIdx.x;
#pragma unroll
        for (int i = 0; i < IN; ++i) s_in[i * P + c] = xr[i];
#pragma unroll
        for (int l = 0; l < NH; ++l)
#pragma unroll
            for (int j = 0; j < H; ++j) s_in[(IN + l * H + j) * P + c] = h[l][j];
#pragma unroll
        for (int o = 0; o < OUT; ++o) s_g[(NH * H + o) * P + c] = d[o];
        // output layer -> gradient of the last hidden activation
        float dh[H];
        {
            const float *Wo = sw + S::w_off(NH);
#pragma unroll
            for (int i = 0; i < H; ++i) {
                float a = 0.0f;
#pragma unroll
                for (int o = 0; o < OUT; ++o) a = fmaf(Wo[o * H + i], d[o], a);
                dh[i] = (h[NH - 1][i] > 0.0f) ? a : 0.0f;  // through the ReLU: gradient of the pre-activation
            }
        }
#pragma unroll
        for (int l = NH - 1; l >= 0; --l) {
#pragma unroll
            for (int j = 0; j < H; ++j) s_g[(l * H + j) * P + c] = dh[j];
            const float *W = sw + S::w_off(l);
            if (l > 0) {
                float dn[H];
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    float a = 0.0f;
#pragma unroll
                    for (int j = 0; j < H; ++j) a = fmaf(W[j * H + i], dh[j], a);
                    dn[i] = (h[l - 1][i] > 0.0f) ? a : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < H; ++i) dh[i] = dn[i];
            } else if (gx != nullptr && live) {
                float *dst = gx + s * IN;
#pragma unroll
                for (int i = 0; i < IN; ++i) {
                    float a = 0.0f;
#pragma unroll
                    for (int j = 0; j < H; ++j) a = fmaf(W[j * IN + i], dh[j], a);
                    dst[i] = a;
                }
            }
        }
        __syncthreads();
        // ---- phase 2: weight / bias gradients of this tile: element e = (layer, j, i): sum_s g[j][s] * in[i][s]
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = threadIdx.x + k * kMlpTile;
            if (e < S::n_params) {
                // decode e -> (gradient row, input row or -1 for a bias)
                int rem = e, grow = 0, irow = -1;
#pragma unroll
                for (int l = 0; l <= NH; ++l) {
                    const int fan_in = (l == 0) ? IN : H;
                    const int fan_out = (l == NH) ? OUT : H;
                    const int in_base = (l == 0) ? 0 : IN + (l - 1) * H;
                    const int g_base = l * H;
                    const int wsz = fan_in * fan_out;
                    if (rem >= 0 && rem < wsz) {
                        grow = g_base + rem / fan_in;
                        irow = in_base + rem % fan_in;
                        rem = -1;
                    } else if (rem >= wsz && rem < wsz + fan_out) {
                        grow = g_base + (rem - wsz);
                        irow = -1;
                        rem = -1;
                    } else if (rem >= 0) {
                        rem -= wsz + fan_out;
                    }
                }
                const float4 *gp = reinterpret_cast<const float4 *>(s_g + grow * P);
                float a = 0.0f;
                if (irow >= 0) {
                    const float4 *ip = reinterpret_cast<const float4 *>(s_in + irow * P);
#pragma unroll 8
                    for (int q = 0; q < kMlpTile / 4; ++q) {
                        const float4 gv = gp[q], iv = ip[q];
                        a = fmaf(gv.x, iv.x, a); a = fmaf(gv.y, iv.y, a);
                        a = fmaf(gv.z, iv.z, a); a = fmaf(gv.w, iv.w, a);
                    }
                } else {
#pragma unroll 8
                    for (int q = 0; q < kMlpTile / 4; ++q) {
                        const float4 gv = gp[q];
                        a += (gv.x + gv.y) + (gv.z + gv.w);
                    }
                }
                acc[k] += a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = threadIdx.x + k * kMlpTile;
        if (e < S::n_params) partials[(size_t)blockIdx.x * S::n_params + e] = (double)acc[k];
    }
}

// partial sums [nblocks][n] (fp64) -> grad_params [n] fp32, one workgroup per element (fixed tree: reproducible)
__global__ __launch_bounds__(256) void mlp_finish_kernel(const double *__restrict__ partials, int nblocks, int n,
                                                         float *__restrict__ out) {
    __shared__ double s_w[4];
    const int q = blockIdx.x;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) v += partials[(size_t)b * n + q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[q] = (float)((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}


// Same sums for long parameter vectors (the width-128 decoders: 22 K parameters x 256 partial rows): a workgroup owns 64
// consecutive parameters, 16 row groups x 64 columns, so every load instruction reads 512 contiguous bytes of one partial
// row (the kernel above reads one double per 180 KB stride). Fixed order: rows r, r + 16, ... per group, then the groups.
__global__ __launch_bounds__(1024) void mlp_finish_cols_kernel(const double *__restrict__ partials, int nblocks, int n,
                                                               float *__restrict__ out) {
    __shared__ double s_g[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + c;
    double v = 0.0;
    if (q < n)
        for (int b = g; b < nblocks; b += 16) v += partials[(size_t)b * n + q];
    s_g[g][c] = v;
    __syncthreads();
    if (g == 0 && q < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_g[k][c];
        out[q] = (float)t;
    }
}

hipError_t mlp_finish_launch(const double *partials, int nblocks, int n, float *out, hipStream_t s) {
    if (n >= 4096)
        hipLaunchKernelGGL(mlp_finish_cols_kernel, dim3((n + 63) / 64), dim3(1024), 0, s, partials, nblocks, n, out);
    else
        hipLaunchKernelGGL(mlp_finish_kernel, dim3(n), dim3(256), 0, s, partials, nblocks, n, out);
    return hipGetLastError();
}

// mlp_mfma.hip: the same decoders on the fp32 matrix cores (MB = 32: width-64 NeRF decoders, MB = 16: width-16 image
// decoders). Option "mlp_variant": -1 (default) = MFMA wherever instantiated, 0 = the VALU kernels of this file.
template <int MB, int IN, int H, int NH, int OUT>
hipError_t wide_mlp_run(bool bwd, int64_t N, const float *x, const float *params, float *y, const float *gy, float *gx,
                        float *gparams, double *partials, hipStream_t s);

template <int IN, int H, int NH, int OUT>
static hipError_t mlp_run(bool bwd, int64_t N, const float *x, const float *params, float *y, const float *gy,
                          float *gx, float *gparams, double *partials, hipStream_t s) {
    using S = MlpShape<IN, H, NH, OUT>;
    const int64_t tiles = (N + kMlpTile - 1) / kMlpTile;
    if (!bwd) {
        int64_t blocks = tiles < 2048 ? tiles : 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL((mlp_forward_kernel<IN, H, NH, OUT>), dim3((uint32_t)blocks), dim3(kMlpTile), 0, s, x, params,
                           y, N);
        return hipGetLastError();
    }
    int64_t blocks = tiles < kMlpMaxBlocks ? tiles : kMlpMaxBlocks;
    if (blocks < 1) blocks = 1;
    const int tpb = (int)((tiles + blocks - 1) / blocks);
    constexpr int P = kMlpTile + 4;
    const size_t shmem = ((size_t)(S::n_params + 3) / 4 * 4 + (size_t)(IN + NH * H + NH * H + OUT) * P) * sizeof(float);
    static PerDeviceOnce once;
    const hipError_t oe = once.run([]() -> hipError_t {
        constexpr size_t need = ((size_t)(S::n_params + 3) / 4 * 4 + (size_t)(IN + NH * H + NH * H + OUT) * P) * sizeof(float);
        if (need > 64 * 1024)
            return hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_backward_kernel<IN, H, NH, OUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
        return hipSuccess;
    });
    if (oe != hipSuccess) return oe;
    hipLaunchKernelGGL((mlp_backward_kernel<IN, H, NH, OUT>), dim3((uint32_t)blocks), dim3(kMlpTile), shmem, s, x,
                       params, gy, gx, partials, N, tpb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return mlp_finish_launch(partials, (int)blocks, S::n_params, gparams, s);
}

typedef hipError_t (*mlp_fn)(bool, int64_t, const float *, const float *, float *, const float *, float *, float *,
                             double *, hipStream_t);

static mlp_fn mlp_lookup(int in, int h, int nh, int out) {
    const bool valu_only = opt().mlp_variant == 0;
#define SHACIRA_WIDE(MB, IN, H, NH, OUT) \
    if (!valu_only && in == IN && h == H && nh == NH && out == OUT) return &wide_mlp_run<MB, IN, H, NH, OUT>;
    SHACIRA_WIDE(32, 32, 64, 1, 16)  // NeRF density decoder: 16 levels x F=2 -> 64 -> 16   (nerf.py:121-130, hidden_dim 64)
    SHACIRA_WIDE(32, 43, 64, 2, 3)   // NeRF colour decoder: 16 + 27 (view embedding) -> 64 -> 64 -> rgb   (nerf.py:132-140)
    SHACIRA_WIDE(32, 32, 64, 2, 3)   // image / 3-D field decoders with hidden_dim 64
    SHACIRA_WIDE(32, 16, 64, 2, 3)
    SHACIRA_WIDE(32, 32, 64, 1, 3)
    SHACIRA_WIDE(32, 96, 128, 1, 16) // nerf_lego.yaml (hidden_dim 128): 24 levels x F=4 -> 128 -> 16
    SHACIRA_WIDE(32, 43, 128, 2, 3)  // nerf_lego.yaml colour decoder: 16 + 27 -> 128 -> 128 -> rgb
    SHACIRA_WIDE(16, 32, 16, 2, 3)   // config B image decoder on 16x16x4 MFMA
    SHACIRA_WIDE(16, 24, 16, 2, 3)
    SHACIRA_WIDE(16, 16, 16, 2, 3)
    SHACIRA_WIDE(16, 48, 16, 2, 3)
    SHACIRA_WIDE(16, 32, 16, 1, 3)
    SHACIRA_WIDE(16, 32, 16, 2, 4)
#undef SHACIRA_WIDE
#define SHACIRA_MLP(IN, H, NH, OUT) \
    if (in == IN && h == H && nh == NH && out == OUT) return &mlp_run<IN, H, NH, OUT>;
    SHACIRA_MLP(32, 16, 2, 3)   // config B: 16 levels x F=2 -> 16 -> 16 -> rgb
    SHACIRA_MLP(24, 16, 2, 3)   // kodak.yaml: 24 levels x F=1
    SHACIRA_MLP(16, 16, 2, 3)   // config A: 8 levels x F=2
    SHACIRA_MLP(48, 16, 2, 3)   // 24 levels x F=2
    SHACIRA_MLP(32, 16, 1, 3)
    SHACIRA_MLP(32, 16, 3, 3)
    SHACIRA_MLP(32, 16, 2, 4)
#undef SHACIRA_MLP
    return nullptr;
}

bool mlp_supported(int in, int h, int nh, int out) { return mlp_lookup(in, h, nh, out) != nullptr; }

int mlp_num_params(int in, int h, int nh, int out) {
    int n = 0;
    for (int l = 0; l < nh; ++l) n += (l == 0 ? in : h) * h + h;
    return n + out * h + out;
}

size_t mlp_workspace_bytes(int in, int h, int nh, int out) {
    const int rows = h < 32 ? kMlpNarrowMaxBlocks : kMlpMaxBlocks;
    return (size_t)rows * mlp_num_params(in, h, nh, out) * sizeof(double);
}

hipError_t mlp_dispatch(bool bwd, int in, int h, int nh, int out, int64_t N, const float *x, const float *params,
                        float *y, const float *gy, float *gx, float *gparams, double *partials, hipStream_t s) {
    return mlp_lookup(in, h, nh, out)(bwd, N, x, params, y, gy, gx, gparams, partials, s);
}

}  // namespace shacira
