// mlp_mfma.hip -- the NeRF decoders (hidden width 64) on the fp32 matrix cores (gfx950).
//
// Row a14 of SURVEY.md section 8, NeRF half: NeuralRadianceField's density decoder 32 -> 64 -> 16 and colour decoder
// (16 + 27) -> 64 -> 64 -> 3 (wisp/models/nefs/nerf.py:121-147, BasicDecoder basic_decoders.py:74-101): Linear + bias +
// ReLU hidden layers, linear output. Through torch / hipBLASLt the weight-gradient GEMMs (K = the whole sample batch,
// 64x64 outputs) take ~2.2 ms of a 3.8 ms NeRF step. Here each way is one kernel built on v_mfma_f32_32x32x2f32 (fp32
// in, fp32 accumulate: same precision class as the reference's fp32 GEMMs; bf16 MFMA would break fp32 parity).
//
// A wave owns a tile of 32 samples and computes TRANSPOSED activations  H^T [features x samples] = W [out x in] . X^T :
//   A operand = weights from LDS (lane = output row), B operand = activations (lane = sample), C = 32x32 block whose
//   accumulator layout is  row = 8*(v/4) + 4*(lane/32) + v%4,  col = lane%32  (v = accumulator register 0..15).
// Chaining trick: the dot product may run over k in any order, so step (b, v) of the next layer takes k = 32b + row(v, h):
// then accumulator register v of the previous layer's block b IS the B operand of that step -- activations never leave
// the registers between layers, forward or backward (dH^T = W^T . dZ^T chains the same way).
// Weight gradients contract over SAMPLES (dW = dZ^T . In), which needs both operands with the lane on the other
// index: the wave stages dZ^T and In through a private LDS region per layer (bank-conflict-free pitches), 16 MFMA steps
// per 32x32 block of dW, accumulators persistent across the wave's tiles; bias gradients fall out of the same LDS reads.
// Block partials in fp64 + the finishing kernel of mlp.hip (reproducible for a fixed grid).
#include <mutex>

#include "internal.h"

namespace shacira {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMfmaWaves = 4;          // waves (= 32-sample tiles in flight) per workgroup
constexpr int kMfmaMaxBlocks = 256;    // persistent grid: one workgroup per CU

__device__ __forceinline__ int crow(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

template <int IN, int H, int NH, int OUT> struct WideShape {
    static_assert(H % 32 == 0 && OUT <= 32 && IN <= 64 && NH >= 1, "shape not covered by the MFMA kernels");
    static constexpr int KB0 = (IN + 31) / 32;   // 32-wide k blocks of the first layer
    static constexpr int HB = H / 32;
    static constexpr int fan_in(int l) { return l == 0 ? IN : H; }
    static constexpr int fan_out(int l) { return l == NH ? OUT : H; }
    static constexpr int kblocks(int l) { return l == 0 ? KB0 : HB; }
    static constexpr int oblocks(int l) { return l == NH ? 1 : HB; }
    static constexpr int pitch(int l) { return 32 * kblocks(l) + 4; }          // LDS row pitch of W_l (floats)
    static constexpr int lds_w_off(int l) {                                       // padded W_l, then its padded bias
        int off = 0;
        for (int q = 0; q < l; ++q) off += 32 * oblocks(q) * pitch(q) + 32 * oblocks(q);
        return off;
    }
    static constexpr int lds_b_off(int l) { return lds_w_off(l) + 32 * oblocks(l) * pitch(l); }
    static constexpr int lds_weights = lds_w_off(NH + 1);
    static constexpr int p_off(int l) {   // offset of W_l in the flat parameter buffer (mlp.hip layout)
        int off = 0;
        for (int q = 0; q < l; ++q) off += fan_in(q) * fan_out(q) + fan_out(q);
        return off;
    }
    static constexpr int n_params = p_off(NH + 1);
    static constexpr int max_kb = KB0 > HB ? KB0 : HB;
    static constexpr int stage_in_pitch = 32 * max_kb + 1;
    static constexpr int stage_floats = 32 * stage_in_pitch + 32 * HB * 33;     // In [sample][k] + dZ [o][sample]
};

// zero-padded copy of the parameters into LDS
template <class S, int NH> __device__ __forceinline__ void load_weights(float *sw, const float *__restrict__ params) {
    for (int e = threadIdx.x; e < S::lds_weights; e += 64 * kMfmaWaves) sw[e] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int l = 0; l <= NH; ++l) {
        const int fi = S::fan_in(l), fo = S::fan_out(l);
        const float *W = params + S::p_off(l);
        for (int e = threadIdx.x; e < fi * fo; e += 64 * kMfmaWaves)
            sw[S::lds_w_off(l) + (e / fi) * S::pitch(l) + (e % fi)] = W[e];
        for (int e = threadIdx.x; e < fo; e += 64 * kMfmaWaves) sw[S::lds_b_off(l) + e] = W[fi * fo + e];
    }
    __syncthreads();
}

// out^T[ob] = W[32ob.., :] . in^T + b   (A = W rows from LDS as float4 over 4 consecutive k, B = in[b][v])
template <int KB, int OB, bool RELU>
__device__ __forceinline__ void layer_forward(const float *__restrict__ sW, int pitch, const float *__restrict__ sB,
                                              const f32x16 (&in)[KB], f32x16 (&out)[OB], int i, int h) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = sB[32 * ob + crow(v, h)];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = *reinterpret_cast<const float4 *>(sW + (32 * ob + i) * pitch + 32 * b + 8 * g + 4 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, in[b][4 * g + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, in[b][4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, in[b][4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, in[b][4 * g + 3], acc, 0, 0, 0);
            }
        }
        if (RELU) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = fmaxf(acc[v], 0.0f);
        }
        out[ob] = acc;
    }
}

// din^T[kb] = W^T[32kb.., :] . dz^T   (A = W[o = 32ob + row(v, h)][k = 32kb + i] from LDS, B = dz[ob][v])
template <int KB, int OB>
__device__ __forceinline__ void layer_backward(const float *__restrict__ sW, int pitch, const f32x16 (&dz)[OB],
                                               f32x16 (&din)[KB], int i, int h, int out_rows) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                if (32 * ob + crow(v, 0) >= out_rows) continue;   // rows beyond fan_out are zero in both halves
                const float a = sW[(32 * ob + crow(v, h)) * pitch + 32 * kb + i];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dz[ob][v], acc, 0, 0, 0);
            }
        }
        din[kb] = acc;
    }
}

template <int IN, int KB0>
__device__ __forceinline__ void load_input(const float *__restrict__ x, int64_t s, bool live, f32x16 (&in)[KB0], int h) {
#pragma unroll
    for (int b = 0; b < KB0; ++b) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * b + 8 * g + 4 * h;
            if constexpr (IN % 4 == 0) {
                float4 q = {0.0f, 0.0f, 0.0f, 0.0f};
                if (live && f0 < IN) q = *reinterpret_cast<const float4 *>(x + s * IN + f0);
                in[b][4 * g] = q.x; in[b][4 * g + 1] = q.y; in[b][4 * g + 2] = q.z; in[b][4 * g + 3] = q.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) in[b][4 * g + r] = (live && f0 + r < IN) ? x[s * IN + f0 + r] : 0.0f;
            }
        }
    }
}

// rows [0, ROWS) of a transposed block set -> out[s, ROWS] (sample-major rows of the caller's tensor)
template <int ROWS, int NB>
__device__ __forceinline__ void store_rows(float *__restrict__ out, int64_t s, bool live, const f32x16 (&t)[NB], int h) {
    if (!live) return;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * b + 8 * g + 4 * h;
            if constexpr (ROWS % 4 == 0) {
                if (f0 < ROWS)
                    *reinterpret_cast<float4 *>(out + s * ROWS + f0) =
                        make_float4(t[b][4 * g], t[b][4 * g + 1], t[b][4 * g + 2], t[b][4 * g + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f0 + r < ROWS) out[s * ROWS + f0 + r] = t[b][4 * g + r];
            }
        }
    }
}

template <int IN, int H, int NH, int OUT>
__global__ __launch_bounds__(64 * kMfmaWaves) void wide_mlp_forward_kernel(const float *__restrict__ x,
                                                                            const float *__restrict__ params,
                                                                            float *__restrict__ y, int64_t N) {
    using S = WideShape<IN, H, NH, OUT>;
    extern __shared__ __align__(16) float smem[];
    load_weights<S, NH>(smem, params);
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles = (N + 31) / 32;
    for (int64_t t = (int64_t)blockIdx.x * kMfmaWaves + wave; t < tiles; t += (int64_t)gridDim.x * kMfmaWaves) {
        const int64_t s = t * 32 + i;
        const bool live = s < N;
        f32x16 in0[S::KB0], ha[S::HB], hb[S::HB], yo[1];
        load_input<IN, S::KB0>(x, s, live, in0, h);
        layer_forward<S::KB0, S::HB, true>(smem + S::lds_w_off(0), S::pitch(0), smem + S::lds_b_off(0), in0, ha, i, h);
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            layer_forward<S::HB, S::HB, true>(smem + S::lds_w_off(l), S::pitch(l), smem + S::lds_b_off(l), ha, hb, i, h);
#pragma unroll
            for (int b = 0; b < S::HB; ++b) ha[b] = hb[b];
        }
        layer_forward<S::HB, 1, false>(smem + S::lds_w_off(NH), S::pitch(NH), smem + S::lds_b_off(NH), ha, yo, i, h);
        store_rows<OUT, 1>(y, s, live, yo, h);
    }
}

// dW block (ob, kb) += dZ^T[32ob.., samples] . In[samples, 32kb..] from the wave's LDS stage; also the bias partial
template <int KB, int OB>
__device__ __forceinline__ void accumulate_dw(const float *__restrict__ s_in, int in_pitch, const float *__restrict__ s_dz,
                                              f32x16 (&dw)[OB][KB], float (&db)[OB], int i, int h) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
        float a[16];
        float bsum = 0.0f;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            a[st] = s_dz[(32 * ob + i) * 33 + 2 * st + h];
            bsum += a[st];
        }
        db[ob] += bsum;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const float b = s_in[(2 * st + h) * in_pitch + 32 * kb + i];
                dw[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b, dw[ob][kb], 0, 0, 0);
            }
        }
    }
}

template <int NB> __device__ __forceinline__ void stage_in(float *s_in, int in_pitch, const f32x16 (&t)[NB], int j, int h) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) s_in[j * in_pitch + 32 * b + crow(v, h)] = t[b][v];
}
template <int NB> __device__ __forceinline__ void stage_dz(float *s_dz, const f32x16 (&t)[NB], int j, int h) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) s_dz[(32 * b + crow(v, h)) * 33 + j] = t[b][v];
}

// dW block -> the block's fp32 parameter-gradient image in LDS
template <int KB, int OB>
__device__ __forceinline__ void flush_dw(float *s_gp, int p_off, int fan_in, int fan_out, const f32x16 (&dw)[OB][KB],
                                         const float (&db)[OB], int j, int h) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int o = 32 * ob + crow(v, h), k = 32 * kb + j;
                if (o < fan_out && k < fan_in) atomicAdd(&s_gp[p_off + o * fan_in + k], dw[ob][kb][v]);
            }
        }
        // bias: lane (i, h) holds the sum over its 16 samples of row 32ob + i
        const int o = 32 * ob + j;
        if (o < fan_out) atomicAdd(&s_gp[p_off + fan_in * fan_out + o], db[ob]);
    }
}

template <int IN, int H, int NH, int OUT>
__global__ __launch_bounds__(64 * kMfmaWaves) void wide_mlp_backward_kernel(
    const float *__restrict__ x, const float *__restrict__ params, const float *__restrict__ gy,
    float *__restrict__ gx, double *__restrict__ partials, int64_t N) {
    using S = WideShape<IN, H, NH, OUT>;
    static_assert(NH <= 2, "register plan covers one or two hidden layers");
    extern __shared__ __align__(16) float smem[];
    float *s_gp = smem + S::lds_weights;                        // [n_params] block gradient image
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    float *s_in = s_gp + (S::n_params + 3) / 4 * 4 + wave * S::stage_floats;   // wave-private stage
    float *s_dz = s_in + 32 * S::stage_in_pitch;
    for (int e = threadIdx.x; e < S::n_params; e += 64 * kMfmaWaves) s_gp[e] = 0.0f;
    load_weights<S, NH>(smem, params);

    // persistent weight-gradient accumulators
    f32x16 dw0[S::HB][S::KB0], dw1[S::HB][S::HB], dwo[1][S::HB];
    float db0[S::HB], db1[S::HB], dbo[1];
#pragma unroll
    for (int a = 0; a < S::HB; ++a) {
        db0[a] = 0.0f; db1[a] = 0.0f;
#pragma unroll
        for (int b = 0; b < S::KB0; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) dw0[a][b][v] = 0.0f;
#pragma unroll
        for (int b = 0; b < S::HB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) { dw1[a][b][v] = 0.0f; }
#pragma unroll
        for (int v = 0; v < 16; ++v) dwo[0][a][v] = 0.0f;
    }
    dbo[0] = 0.0f;

    const int64_t tiles = (N + 31) / 32;
    for (int64_t t = (int64_t)blockIdx.x * kMfmaWaves + wave; t < tiles; t += (int64_t)gridDim.x * kMfmaWaves) {
        const int64_t s = t * 32 + i;
        const bool live = s < N;
        // forward recompute, hidden activations kept in registers
        f32x16 in0[S::KB0], h0[S::HB], h1[S::HB];
        load_input<IN, S::KB0>(x, s, live, in0, h);
        layer_forward<S::KB0, S::HB, true>(smem + S::lds_w_off(0), S::pitch(0), smem + S::lds_b_off(0), in0, h0, i, h);
        if constexpr (NH == 2)
            layer_forward<S::HB, S::HB, true>(smem + S::lds_w_off(1), S::pitch(1), smem + S::lds_b_off(1), h0, h1, i, h);
        // upstream gradient in accumulator layout
        f32x16 dzo[1];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int o = crow(v, h);
            dzo[0][v] = (live && o < OUT) ? gy[s * OUT + o] : 0.0f;
        }
        const f32x16 (&hlast)[S::HB] = (NH == 2) ? h1 : h0;
        // output layer: dW_out, then dH_last
        stage_in<S::HB>(s_in, S::stage_in_pitch, hlast, i, h);
        stage_dz<1>(s_dz, dzo, i, h);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // stage written by all lanes before any lane reads it
        accumulate_dw<S::HB, 1>(s_in, S::stage_in_pitch, s_dz, dwo, dbo, i, h);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        f32x16 dcur[S::HB];
        layer_backward<S::HB, 1>(smem + S::lds_w_off(NH), S::pitch(NH), dzo, dcur, i, h, OUT);
#pragma unroll
        for (int b = 0; b < S::HB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) dcur[b][v] = (hlast[b][v] > 0.0f) ? dcur[b][v] : 0.0f;
        if constexpr (NH == 2) {
            // hidden layer 1: dW_1 = dZ_1^T . H_0, then dH_0
            stage_in<S::HB>(s_in, S::stage_in_pitch, h0, i, h);
            stage_dz<S::HB>(s_dz, dcur, i, h);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            accumulate_dw<S::HB, S::HB>(s_in, S::stage_in_pitch, s_dz, dw1, db1, i, h);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            f32x16 dprev[S::HB];
            layer_backward<S::HB, S::HB>(smem + S::lds_w_off(1), S::pitch(1), dcur, dprev, i, h, H);
#pragma unroll
            for (int b = 0; b < S::HB; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) dcur[b][v] = (h0[b][v] > 0.0f) ? dprev[b][v] : 0.0f;
        }
        // first layer: dW_0 = dZ_0^T . X, then dX
        stage_in<S::KB0>(s_in, S::stage_in_pitch, in0, i, h);
        stage_dz<S::HB>(s_dz, dcur, i, h);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        accumulate_dw<S::KB0, S::HB>(s_in, S::stage_in_pitch, s_dz, dw0, db0, i, h);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (gx != nullptr) {
            f32x16 dx[S::KB0];
            layer_backward<S::KB0, S::HB>(smem + S::lds_w_off(0), S::pitch(0), dcur, dx, i, h, H);
            store_rows<IN, S::KB0>(gx, s, live, dx, h);
        }
    }
    // wave accumulators -> block image (LDS float atomics: once per kernel) -> fp64 block partial
    flush_dw<S::KB0, S::HB>(s_gp, S::p_off(0), IN, H, dw0, db0, i, h);
    if constexpr (NH == 2) flush_dw<S::HB, S::HB>(s_gp, S::p_off(1), H, H, dw1, db1, i, h);
    flush_dw<S::HB, 1>(s_gp, S::p_off(NH), H, OUT, dwo, dbo, i, h);
    __syncthreads();
    for (int e = threadIdx.x; e < S::n_params; e += 64 * kMfmaWaves)
        partials[(size_t)blockIdx.x * S::n_params + e] = (double)s_gp[e];
}

// finishing pass (mlp.hip): fp64 block partials -> fp32 gradient, fixed summation tree
hipError_t mlp_finish_launch(const double *partials, int nblocks, int n, float *out, hipStream_t s);

template <int IN, int H, int NH, int OUT>
hipError_t wide_mlp_run(bool bwd, int64_t N, const float *x, const float *params, float *y, const float *gy, float *gx,
                        float *gparams, double *partials, hipStream_t s) {
    using S = WideShape<IN, H, NH, OUT>;
    const int64_t tiles = (N + 31) / 32;
    int64_t blocks = (tiles + kMfmaWaves - 1) / kMfmaWaves;
    if (blocks > kMfmaMaxBlocks) blocks = kMfmaMaxBlocks;
    if (blocks < 1) blocks = 1;
    constexpr size_t fwd_lds = (size_t)S::lds_weights * sizeof(float);
    constexpr size_t bwd_lds =
        ((size_t)S::lds_weights + (S::n_params + 3) / 4 * 4 + (size_t)kMfmaWaves * S::stage_floats) * sizeof(float);
    static_assert(bwd_lds <= 160 * 1024, "backward LDS plan exceeds the CU's 160 KiB");
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        hipError_t e = hipSuccess;
        if (fwd_lds > 64 * 1024)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_mlp_forward_kernel<IN, H, NH, OUT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds);
        if (e == hipSuccess && bwd_lds > 64 * 1024)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_mlp_backward_kernel<IN, H, NH, OUT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds);
        attr_err = e;
    });
    if (attr_err != hipSuccess) return attr_err;
    if (!bwd) {
        hipLaunchKernelGGL((wide_mlp_forward_kernel<IN, H, NH, OUT>), dim3((uint32_t)blocks), dim3(64 * kMfmaWaves),
                           fwd_lds, s, x, params, y, N);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((wide_mlp_backward_kernel<IN, H, NH, OUT>), dim3((uint32_t)blocks), dim3(64 * kMfmaWaves), bwd_lds,
                       s, x, params, gy, gx, partials, N);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return mlp_finish_launch(partials, (int)blocks, S::n_params, gparams, s);
}

// the instantiations mlp.hip's lookup table refers to
template hipError_t wide_mlp_run<32, 64, 1, 16>(bool, int64_t, const float *, const float *, float *, const float *,
                                                 float *, float *, double *, hipStream_t);
template hipError_t wide_mlp_run<43, 64, 2, 3>(bool, int64_t, const float *, const float *, float *, const float *,
                                                float *, float *, double *, hipStream_t);
template hipError_t wide_mlp_run<32, 64, 2, 3>(bool, int64_t, const float *, const float *, float *, const float *,
                                                float *, float *, double *, hipStream_t);
template hipError_t wide_mlp_run<16, 64, 2, 3>(bool, int64_t, const float *, const float *, float *, const float *,
                                                float *, float *, double *, hipStream_t);

}  // namespace shacira
