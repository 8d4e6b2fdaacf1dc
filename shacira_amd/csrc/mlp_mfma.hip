// mlp_mfma.hip -- the decoder MLPs on the fp32 matrix cores (gfx950): NeRF decoders (hidden width 64) on
// v_mfma_f32_32x32x2f32, image decoders (hidden width 16) on v_mfma_f32_16x16x4f32 -- one templated implementation.
//
// Row a14 of SURVEY.md section 8, NeRF half: NeuralRadianceField's density decoder 32 -> 64 -> 16 and colour decoder
// (16 + 27) -> 64 -> 64 -> 3 (wisp/models/nefs/nerf.py:121-147, BasicDecoder basic_decoders.py:74-101): Linear + bias +
// ReLU hidden layers, linear output. Through torch / hipBLASLt the weight-gradient GEMMs (K = the whole sample batch,
// 64x64 outputs) take ~2.2 ms of a 3.8 ms NeRF step. Here each way is one kernel built on v_mfma_f32_32x32x2f32 (fp32
// in, fp32 accumulate: same precision class as the reference's fp32 GEMMs; bf16 MFMA would break fp32 parity).
//
// A wave owns a tile of MB samples (MB = 32 or 16 = the MFMA block) and computes TRANSPOSED activations
// H^T [features x samples] = W [out x in] . X^T :
//   A operand = weights from LDS (lane = output row), B operand = activations (lane = sample), C = MB x MB block whose
//   accumulator layout is  row = (v/4)*(4*NQ) + 4*(lane/MB) + v%4,  col = lane%MB  (v = accumulator register, NQ = 64/MB;
//   both layouts probed on the chip: tools/mfma_probe.hip, tools/mfma_probe16.hip).
// Chaining trick: the dot product may run over k in any order, so step (b, v) of the next layer takes k = MB*b + row(v, q):
// then accumulator register v of the previous layer's block b IS the B operand of that step -- activations never leave
// the registers between layers, forward or backward (dH^T = W^T . dZ^T chains the same way).
// Weight gradients contract over SAMPLES (dW = dZ^T . In), which needs both operands with the lane on the other
// index: the wave stages dZ^T and In through a private LDS region per layer (bank-conflict-free pitches), MB/NQ MFMA
// steps per block of dW, accumulators persistent across the wave's tiles; bias gradients fall out of the same LDS reads.
// Block partials in fp64 + the finishing kernel of mlp.hip (reproducible for a fixed grid).
#include <mutex>

#include "internal.h"

namespace shacira {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaWaves = 4;          // waves (= sample tiles in flight) per workgroup
constexpr int kMfmaMaxBlocks = 256;    // persistent grid: one workgroup per CU

// MB = rows / columns of the MFMA block, NA = accumulator registers per lane, NQ = 64 / MB = k values per instruction
template <int MB> struct Mma;
template <> struct Mma<32> {
    typedef f32x16 acc_t;
    static constexpr int NA = 16, NQ = 2;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<16> {
    typedef f32x4 acc_t;
    static constexpr int NA = 4, NQ = 4;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
template <int MB> __device__ __forceinline__ constexpr int crow(int v, int q) {
    return (v >> 2) * (4 * Mma<MB>::NQ) + 4 * q + (v & 3);
}

template <int MB, int IN, int H, int NH, int OUT> struct WideShape {
    static_assert(H % MB == 0 && OUT <= MB && IN <= 128 && NH >= 1 && NH <= 2, "shape not covered by the MFMA kernels");
    static constexpr bool kSplit = H > 64;   // width 128: the weight gradients are split over the workgroup's waves
    static constexpr int KB0 = (IN + MB - 1) / MB;   // MB-wide k blocks of the first layer
    static constexpr int HB = H / MB;
    static constexpr int fan_in(int l) { return l == 0 ? IN : H; }
    static constexpr int fan_out(int l) { return l == NH ? OUT : H; }
    static constexpr int kblocks(int l) { return l == 0 ? KB0 : HB; }
    static constexpr int oblocks(int l) { return l == NH ? 1 : HB; }
    static constexpr int pitch(int l) { return MB * kblocks(l) + 4; }          // LDS row pitch of W_l (floats)
    static constexpr int lds_w_off(int l) {                                       // padded W_l, then its padded bias
        int off = 0;
        for (int q = 0; q < l; ++q) off += MB * oblocks(q) * pitch(q) + MB * oblocks(q);
        return off;
    }
    static constexpr int lds_b_off(int l) { return lds_w_off(l) + MB * oblocks(l) * pitch(l); }
    static constexpr int lds_weights = lds_w_off(NH + 1);
    static constexpr int p_off(int l) {   // offset of W_l in the flat parameter buffer (mlp.hip layout)
        int off = 0;
        for (int q = 0; q < l; ++q) off += fan_in(q) * fan_out(q) + fan_out(q);
        return off;
    }
    static constexpr int n_params = p_off(NH + 1);
    static constexpr int max_kb = KB0 > HB ? KB0 : HB;
    static constexpr int stage_in_pitch = MB * max_kb + 1;
    static constexpr int stage_floats = MB * stage_in_pitch + MB * HB * (MB + 1);   // In [sample][k] + dZ [o][sample]
    // backward LDS plan (floats): per-wave stages + the block's gradient image, or (split) ONE shared stage
    static constexpr size_t bwd_lds_floats =
        kSplit ? (size_t)lds_weights + stage_floats
               : (size_t)lds_weights + (n_params + 3) / 4 * 4 + (size_t)kMfmaWaves * stage_floats;
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

// out^T[ob] = W[MB*ob.., :] . in^T + b   (A = W rows from LDS as float4 over 4 consecutive k, B = in[b][v])
template <int MB, int KB, int OB, bool RELU>
__device__ __forceinline__ void layer_forward(const float *__restrict__ sW, int pitch, const float *__restrict__ sB,
                                              const typename Mma<MB>::acc_t (&in)[KB],
                                              typename Mma<MB>::acc_t (&out)[OB], int i, int q) {
    using M = Mma<MB>;
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
        typename M::acc_t acc;
#pragma unroll
        for (int v = 0; v < M::NA; ++v) acc[v] = sB[MB * ob + crow<MB>(v, q)];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
#pragma unroll
            for (int g = 0; g < M::NA / 4; ++g) {
                const float4 a = *reinterpret_cast<const float4 *>(sW + (MB * ob + i) * pitch + MB * b +
                                                                   4 * M::NQ * g + 4 * q);
                acc = M::mma(a.x, in[b][4 * g + 0], acc);
                acc = M::mma(a.y, in[b][4 * g + 1], acc);
                acc = M::mma(a.z, in[b][4 * g + 2], acc);
                acc = M::mma(a.w, in[b][4 * g + 3], acc);
            }
        }
        if (RELU) {
#pragma unroll
            for (int v = 0; v < M::NA; ++v) acc[v] = fmaxf(acc[v], 0.0f);
        }
        out[ob] = acc;
        // width 128: keep the scheduler from hoisting every block's LDS operand loads to the top (spills at 512 registers)
        if constexpr (KB * OB > 4) __builtin_amdgcn_sched_barrier(0);
    }
}

// din^T[kb] = W^T[MB*kb.., :] . dz^T   (A = W[o = MB*ob + row(v, q)][k = MB*kb + i] from LDS, B = dz[ob][v])
template <int MB, int KB, int OB>
__device__ __forceinline__ void layer_backward(const float *__restrict__ sW, int pitch,
                                               const typename Mma<MB>::acc_t (&dz)[OB],
                                               typename Mma<MB>::acc_t (&din)[KB], int i, int q, int out_rows) {
    using M = Mma<MB>;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        typename M::acc_t acc;
#pragma unroll
        for (int v = 0; v < M::NA; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
            for (int v = 0; v < M::NA; ++v) {
                if (MB * ob + crow<MB>(v, 0) >= out_rows) continue;   // rows beyond fan_out are zero for every lane
                const float a = sW[(MB * ob + crow<MB>(v, q)) * pitch + MB * kb + i];
                acc = M::mma(a, dz[ob][v], acc);
            }
        }
        din[kb] = acc;
        if constexpr (KB * OB > 4) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MB, int IN, int KB0>
__device__ __forceinline__ void load_input(const float *__restrict__ x, int64_t s, bool live,
                                           typename Mma<MB>::acc_t (&in)[KB0], int q) {
    using M = Mma<MB>;
#pragma unroll
    for (int b = 0; b < KB0; ++b) {
#pragma unroll
        for (int g = 0; g < M::NA / 4; ++g) {
            const int f0 = MB * b + 4 * M::NQ * g + 4 * q;
            if constexpr (IN % 4 == 0) {
                float4 t = {0.0f, 0.0f, 0.0f, 0.0f};
                if (live && f0 < IN) t = *reinterpret_cast<const float4 *>(x + s * IN + f0);
                in[b][4 * g] = t.x; in[b][4 * g + 1] = t.y; in[b][4 * g + 2] = t.z; in[b][4 * g + 3] = t.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) in[b][4 * g + r] = (live && f0 + r < IN) ? x[s * IN + f0 + r] : 0.0f;
            }
        }
    }
}

// rows [0, ROWS) of a transposed block set -> out[s, ROWS] (sample-major rows of the caller's tensor)
template <int MB, int ROWS, int NB>
__device__ __forceinline__ void store_rows(float *__restrict__ out, int64_t s, bool live,
                                           const typename Mma<MB>::acc_t (&t)[NB], int q) {
    using M = Mma<MB>;
    if (!live) return;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int g = 0; g < M::NA / 4; ++g) {
            const int f0 = MB * b + 4 * M::NQ * g + 4 * q;
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

template <int MB, int IN, int H, int NH, int OUT>
__global__ __launch_bounds__(64 * kMfmaWaves) void wide_mlp_forward_kernel(const float *__restrict__ x,
                                                                            const float *__restrict__ params,
                                                                            float *__restrict__ y, int64_t N) {
    using S = WideShape<MB, IN, H, NH, OUT>;
    using A = typename Mma<MB>::acc_t;
    extern __shared__ __align__(16) float smem[];
    load_weights<S, NH>(smem, params);
    const int lane = threadIdx.x & 63, i = lane % MB, q = lane / MB, wave = threadIdx.x >> 6;
    const int64_t tiles = (N + MB - 1) / MB;
    for (int64_t t = (int64_t)blockIdx.x * kMfmaWaves + wave; t < tiles; t += (int64_t)gridDim.x * kMfmaWaves) {
        const int64_t s = t * MB + i;
        const bool live = s < N;
        A in0[S::KB0], ha[S::HB], hb[S::HB], yo[1];
        load_input<MB, IN, S::KB0>(x, s, live, in0, q);
        layer_forward<MB, S::KB0, S::HB, true>(smem + S::lds_w_off(0), S::pitch(0), smem + S::lds_b_off(0), in0, ha, i, q);
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            layer_forward<MB, S::HB, S::HB, true>(smem + S::lds_w_off(l), S::pitch(l), smem + S::lds_b_off(l), ha, hb, i,
                                                  q);
#pragma unroll
            for (int b = 0; b < S::HB; ++b) ha[b] = hb[b];
        }
        layer_forward<MB, S::HB, 1, false>(smem + S::lds_w_off(NH), S::pitch(NH), smem + S::lds_b_off(NH), ha, yo, i, q);
        store_rows<MB, OUT, 1>(y, s, live, yo, q);
    }
}

// dW block (ob, kb) += dZ^T[MB*ob.., samples] . In[samples, MB*kb..] from the wave's LDS stage; also the bias partial
template <int MB, int KB, int OB>
__device__ __forceinline__ void accumulate_dw(const float *__restrict__ s_in, int in_pitch, const float *__restrict__ s_dz,
                                              typename Mma<MB>::acc_t (&dw)[OB][KB], float (&db)[OB], int i, int q) {
    using M = Mma<MB>;
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
        float a[M::NA];
        float bsum = 0.0f;
#pragma unroll
        for (int st = 0; st < M::NA; ++st) {
            a[st] = s_dz[(MB * ob + i) * (MB + 1) + M::NQ * st + q];
            bsum += a[st];
        }
        db[ob] += bsum;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int st = 0; st < M::NA; ++st) {
                const float b = s_in[(M::NQ * st + q) * in_pitch + MB * kb + i];
                dw[ob][kb] = M::mma(a[st], b, dw[ob][kb]);
            }
            if constexpr (KB > 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MB, int NB>
__device__ __forceinline__ void stage_in(float *s_in, int in_pitch, const typename Mma<MB>::acc_t (&t)[NB], int j, int q) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < Mma<MB>::NA; ++v) s_in[j * in_pitch + MB * b + crow<MB>(v, q)] = t[b][v];
}
template <int MB, int NB>
__device__ __forceinline__ void stage_dz(float *s_dz, const typename Mma<MB>::acc_t (&t)[NB], int j, int q) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < Mma<MB>::NA; ++v) s_dz[(MB * b + crow<MB>(v, q)) * (MB + 1) + j] = t[b][v];
}

// dW block -> the block's fp32 parameter-gradient image in LDS
template <int MB, int KB, int OB>
__device__ __forceinline__ void flush_dw(float *s_gp, int p_off, int fan_in, int fan_out,
                                         const typename Mma<MB>::acc_t (&dw)[OB][KB], const float (&db)[OB], int j, int q) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int v = 0; v < Mma<MB>::NA; ++v) {
                const int o = MB * ob + crow<MB>(v, q), k = MB * kb + j;
                if (o < fan_out && k < fan_in) atomicAdd(&s_gp[p_off + o * fan_in + k], dw[ob][kb][v]);
            }
        }
        // bias: lane (i, q) holds the sum over its MB / NQ samples of row MB*ob + i
        const int o = MB * ob + j;
        if (o < fan_out) atomicAdd(&s_gp[p_off + fan_in * fan_out + o], db[ob]);
    }
}

template <int MB, int IN, int H, int NH, int OUT>
__global__ __launch_bounds__(64 * kMfmaWaves) void wide_mlp_backward_kernel(
    const float *__restrict__ x, const float *__restrict__ params, const float *__restrict__ gy,
    float *__restrict__ gx, double *__restrict__ partials, int64_t N) {
    using S = WideShape<MB, IN, H, NH, OUT>;
    using M = Mma<MB>;
    using A = typename M::acc_t;
    extern __shared__ __align__(16) float smem[];
    float *s_gp = smem + S::lds_weights;                        // [n_params] block gradient image
    const int lane = threadIdx.x & 63, i = lane % MB, q = lane / MB, wave = threadIdx.x >> 6;
    float *s_in = s_gp + (S::n_params + 3) / 4 * 4 + wave * S::stage_floats;   // wave-private stage
    float *s_dz = s_in + MB * S::stage_in_pitch;
    for (int e = threadIdx.x; e < S::n_params; e += 64 * kMfmaWaves) s_gp[e] = 0.0f;
    load_weights<S, NH>(smem, params);

    // persistent weight-gradient accumulators
    A dw0[S::HB][S::KB0], dw1[S::HB][S::HB], dwo[1][S::HB];
    float db0[S::HB], db1[S::HB], dbo[1];
#pragma unroll
    for (int a = 0; a < S::HB; ++a) {
        db0[a] = 0.0f; db1[a] = 0.0f;
#pragma unroll
        for (int b = 0; b < S::KB0; ++b)
#pragma unroll
            for (int v = 0; v < M::NA; ++v) dw0[a][b][v] = 0.0f;
#pragma unroll
        for (int b = 0; b < S::HB; ++b)
#pragma unroll
            for (int v = 0; v < M::NA; ++v) { dw1[a][b][v] = 0.0f; }
#pragma unroll
        for (int v = 0; v < M::NA; ++v) dwo[0][a][v] = 0.0f;
    }
    dbo[0] = 0.0f;

    const int64_t tiles = (N + MB - 1) / MB;
    for (int64_t t = (int64_t)blockIdx.x * kMfmaWaves + wave; t < tiles; t += (int64_t)gridDim.x * kMfmaWaves) {
        const int64_t s = t * MB + i;
        const bool live = s < N;
        // forward recompute, hidden activations kept in registers
        A in0[S::KB0], h0[S::HB], h1[S::HB];
        load_input<MB, IN, S::KB0>(x, s, live, in0, q);
        layer_forward<MB, S::KB0, S::HB, true>(smem + S::lds_w_off(0), S::pitch(0), smem + S::lds_b_off(0), in0, h0, i, q);
        if constexpr (NH == 2)
            layer_forward<MB, S::HB, S::HB, true>(smem + S::lds_w_off(1), S::pitch(1), smem + S::lds_b_off(1), h0, h1, i,
                                                  q);
        // upstream gradient in accumulator layout
        A dzo[1];
#pragma unroll
        for (int v = 0; v < M::NA; ++v) {
            const int o = crow<MB>(v, q);
            dzo[0][v] = (live && o < OUT) ? gy[s * OUT + o] : 0.0f;
        }
        const A (&hlast)[S::HB] = (NH == 2) ? h1 : h0;
        // output layer: dW_out, then dH_last
        stage_in<MB, S::HB>(s_in, S::stage_in_pitch, hlast, i, q);
        stage_dz<MB, 1>(s_dz, dzo, i, q);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // stage written by all lanes before any lane reads it
        accumulate_dw<MB, S::HB, 1>(s_in, S::stage_in_pitch, s_dz, dwo, dbo, i, q);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        A dcur[S::HB];
        layer_backward<MB, S::HB, 1>(smem + S::lds_w_off(NH), S::pitch(NH), dzo, dcur, i, q, OUT);
#pragma unroll
        for (int b = 0; b < S::HB; ++b)
#pragma unroll
            for (int v = 0; v < M::NA; ++v) dcur[b][v] = (hlast[b][v] > 0.0f) ? dcur[b][v] : 0.0f;
        if constexpr (NH == 2) {
            // hidden layer 1: dW_1 = dZ_1^T . H_0, then dH_0
            stage_in<MB, S::HB>(s_in, S::stage_in_pitch, h0, i, q);
            stage_dz<MB, S::HB>(s_dz, dcur, i, q);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            accumulate_dw<MB, S::HB, S::HB>(s_in, S::stage_in_pitch, s_dz, dw1, db1, i, q);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            A dprev[S::HB];
            layer_backward<MB, S::HB, S::HB>(smem + S::lds_w_off(1), S::pitch(1), dcur, dprev, i, q, H);
#pragma unroll
            for (int b = 0; b < S::HB; ++b)
#pragma unroll
                for (int v = 0; v < M::NA; ++v) dcur[b][v] = (h0[b][v] > 0.0f) ? dprev[b][v] : 0.0f;
        }
        // first layer: dW_0 = dZ_0^T . X, then dX
        stage_in<MB, S::KB0>(s_in, S::stage_in_pitch, in0, i, q);
        stage_dz<MB, S::HB>(s_dz, dcur, i, q);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        accumulate_dw<MB, S::KB0, S::HB>(s_in, S::stage_in_pitch, s_dz, dw0, db0, i, q);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (gx != nullptr) {
            A dx[S::KB0];
            layer_backward<MB, S::KB0, S::HB>(smem + S::lds_w_off(0), S::pitch(0), dcur, dx, i, q, H);
            store_rows<MB, IN, S::KB0>(gx, s, live, dx, q);
        }
    }
    // wave accumulators -> block image (LDS float atomics: once per kernel) -> fp64 block partial
    flush_dw<MB, S::KB0, S::HB>(s_gp, S::p_off(0), IN, H, dw0, db0, i, q);
    if constexpr (NH == 2) flush_dw<MB, S::HB, S::HB>(s_gp, S::p_off(1), H, H, dw1, db1, i, q);
    flush_dw<MB, S::HB, 1>(s_gp, S::p_off(NH), H, OUT, dwo, dbo, i, q);
    __syncthreads();
    for (int e = threadIdx.x; e < S::n_params; e += 64 * kMfmaWaves)
        partials[(size_t)blockIdx.x * S::n_params + e] = (double)s_gp[e];
}


// ---------------------------------------------------------------------------------------------------------------------
// Width-128 decoders (nerf_lego.yaml: hidden_dim 128 -> density 96 -> 128 -> 16, colour 43 -> 128 -> 128 -> 3).
// One layer's weight gradient is 4 x 4 MFMA blocks = 256 accumulator registers, so a wave cannot keep all of dW like the
// width-64 kernel does. Here every wave still chains ITS OWN 32-sample tile through the network in registers, but the
// weight gradients are split by OUTPUT-ROW block: wave w owns rows [32w, 32w + 32) of dW_0 and dW_1 and k-block w of
// dW_out. The four tiles of an iteration take turns in ONE shared LDS stage (In [32][129] + dZ [128][33] = 33 KiB next to
// 118 KiB of padded weights): the owner wave writes its tile's operands, all four waves accumulate their rows from it.
// No block-level gradient image: every element of dW has exactly one owner, which writes it to the fp64 partial row.
template <int MB, int KB>
__device__ __forceinline__ void flush_owned(double *__restrict__ part, int p_off, int fan_in, int fan_out, int ob, int kb0,
                                            const typename Mma<MB>::acc_t (&dw)[1][KB], int j, int q) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int v = 0; v < Mma<MB>::NA; ++v) {
            const int o = MB * ob + crow<MB>(v, q), k = MB * (kb0 + kb) + j;
            if (o < fan_out && k < fan_in) part[p_off + o * fan_in + k] = (double)dw[0][kb][v];
        }
    }
}
template <int MB>
__device__ __forceinline__ void flush_owned_bias(double *__restrict__ part, int p_off, int fan_in, int fan_out, int ob,
                                                 float db, int j, int q) {
    // lane (j, q) holds the sum over its MB / NQ samples of row MB*ob + j: add the NQ lanes of the row
    float t = db;
#pragma unroll
    for (int m = MB; m < 64; m <<= 1) t += __shfl_xor(t, m, 64);
    const int o = MB * ob + j;
    if (q == 0 && o < fan_out) part[p_off + fan_in * fan_out + o] = (double)t;
}

template <int MB, int IN, int H, int NH, int OUT>
__global__ __launch_bounds__(64 * kMfmaWaves) void split_mlp_backward_kernel(
    const float *__restrict__ x, const float *__restrict__ params, const float *__restrict__ gy,
    float *__restrict__ gx, double *__restrict__ partials, int64_t N) {
    using S = WideShape<MB, IN, H, NH, OUT>;
    using M = Mma<MB>;
    using A = typename M::acc_t;
    static_assert(S::HB == kMfmaWaves, "one output-row block of the hidden layers per wave");
    extern __shared__ __align__(16) float smem[];
    float *s_in = smem + S::lds_weights;             // shared stage: In [MB samples][stage_in_pitch]
    float *s_dz = s_in + MB * S::stage_in_pitch;     //               dZ [H rows][MB + 1]
    const int lane = threadIdx.x & 63, i = lane % MB, q = lane / MB, wave = threadIdx.x >> 6;
    load_weights<S, NH>(smem, params);

    A dw0[1][S::KB0], dw1[1][S::HB], dwo[1][1];
    float db0[1] = {0.0f}, db1[1] = {0.0f}, dbo[1] = {0.0f};
#pragma unroll
    for (int v = 0; v < M::NA; ++v) {
#pragma unroll
        for (int b = 0; b < S::KB0; ++b) dw0[0][b][v] = 0.0f;
#pragma unroll
        for (int b = 0; b < S::HB; ++b) dw1[0][b][v] = 0.0f;
        dwo[0][0][v] = 0.0f;
    }
    const float *s_dz_own = s_dz + (MB * wave) * (MB + 1);   // this wave's rows of dZ

    const int64_t tiles = (N + MB - 1) / MB;
    // uniform trip count for the whole workgroup (barriers inside): a wave past the last tile runs on zeros
    for (int64_t base = (int64_t)blockIdx.x * kMfmaWaves; base < tiles; base += (int64_t)gridDim.x * kMfmaWaves) {
        const int64_t s = (base + wave) * MB + i;
        const bool live = s < N;
        A in0[S::KB0], h0[S::HB], h1[S::HB];
        load_input<MB, IN, S::KB0>(x, s, live, in0, q);
        layer_forward<MB, S::KB0, S::HB, true>(smem + S::lds_w_off(0), S::pitch(0), smem + S::lds_b_off(0), in0, h0, i, q);
        if constexpr (NH == 2)
            layer_forward<MB, S::HB, S::HB, true>(smem + S::lds_w_off(1), S::pitch(1), smem + S::lds_b_off(1), h0, h1, i,
                                                  q);
        A dzo[1];
#pragma unroll
        for (int v = 0; v < M::NA; ++v) {
            const int o = crow<MB>(v, q);
            dzo[0][v] = (live && o < OUT) ? gy[s * OUT + o] : 0.0f;
        }
        const A (&hlast)[S::HB] = (NH == 2) ? h1 : h0;
        // output layer: wave w accumulates k-block w of dW_out (and everyone the bias; wave 0's copy is written)
#pragma unroll 1
        for (int tw = 0; tw < kMfmaWaves; ++tw) {
            if (wave == tw) {
                stage_in<MB, S::HB>(s_in, S::stage_in_pitch, hlast, i, q);
                stage_dz<MB, 1>(s_dz, dzo, i, q);
            }
            __syncthreads();
            accumulate_dw<MB, 1, 1>(s_in + MB * wave, S::stage_in_pitch, s_dz, dwo, dbo, i, q);
            __syncthreads();
        }
        A dcur[S::HB];
        layer_backward<MB, S::HB, 1>(smem + S::lds_w_off(NH), S::pitch(NH), dzo, dcur, i, q, OUT);
#pragma unroll
        for (int b = 0; b < S::HB; ++b)
#pragma unroll
            for (int v = 0; v < M::NA; ++v) dcur[b][v] = (hlast[b][v] > 0.0f) ? dcur[b][v] : 0.0f;
        if constexpr (NH == 2) {
#pragma unroll 1
            for (int tw = 0; tw < kMfmaWaves; ++tw) {
                if (wave == tw) {
                    stage_in<MB, S::HB>(s_in, S::stage_in_pitch, h0, i, q);
                    stage_dz<MB, S::HB>(s_dz, dcur, i, q);
                }
                __syncthreads();
                accumulate_dw<MB, S::HB, 1>(s_in, S::stage_in_pitch, s_dz_own, dw1, db1, i, q);
                __syncthreads();
            }
            A dprev[S::HB];
            layer_backward<MB, S::HB, S::HB>(smem + S::lds_w_off(1), S::pitch(1), dcur, dprev, i, q, H);
#pragma unroll
            for (int b = 0; b < S::HB; ++b)
#pragma unroll
                for (int v = 0; v < M::NA; ++v) dcur[b][v] = (h0[b][v] > 0.0f) ? dprev[b][v] : 0.0f;
        }
        // two hidden layers: the inputs are read again here instead of being kept in 32 registers through both chains
        // (the compiler spilled 216 B / lane otherwise); x2 is opaque so that the two reads are not merged
        A in0b[S::KB0];
        if constexpr (NH == 2) {
            const float *x2 = x;
            asm volatile("" : "+s"(x2));
            load_input<MB, IN, S::KB0>(x2, s, live, in0b, q);
        }
        const A (&in_l0)[S::KB0] = (NH == 2) ? in0b : in0;
#pragma unroll 1
        for (int tw = 0; tw < kMfmaWaves; ++tw) {
            if (wave == tw) {
                stage_in<MB, S::KB0>(s_in, S::stage_in_pitch, in_l0, i, q);
                stage_dz<MB, S::HB>(s_dz, dcur, i, q);
            }
            __syncthreads();
            accumulate_dw<MB, S::KB0, 1>(s_in, S::stage_in_pitch, s_dz_own, dw0, db0, i, q);
            __syncthreads();
        }
        if (gx != nullptr) {
            A dx[S::KB0];
            layer_backward<MB, S::KB0, S::HB>(smem + S::lds_w_off(0), S::pitch(0), dcur, dx, i, q, H);
            store_rows<MB, IN, S::KB0>(gx, s, live, dx, q);
        }
    }
    double *part = partials + (size_t)blockIdx.x * S::n_params;
    flush_owned<MB, S::KB0>(part, S::p_off(0), IN, H, wave, 0, dw0, i, q);
    flush_owned_bias<MB>(part, S::p_off(0), IN, H, wave, db0[0], i, q);
    if constexpr (NH == 2) {
        flush_owned<MB, S::HB>(part, S::p_off(1), H, H, wave, 0, dw1, i, q);
        flush_owned_bias<MB>(part, S::p_off(1), H, H, wave, db1[0], i, q);
    }
    flush_owned<MB, 1>(part, S::p_off(NH), H, OUT, 0, wave, dwo, i, q);
    if (wave == 0) flush_owned_bias<MB>(part, S::p_off(NH), H, OUT, 0, dbo[0], i, q);
}

// finishing pass (mlp.hip): fp64 block partials -> fp32 gradient, fixed summation tree
hipError_t mlp_finish_launch(const double *partials, int nblocks, int n, float *out, hipStream_t s);

template <int MB, int IN, int H, int NH, int OUT>
hipError_t wide_mlp_run(bool bwd, int64_t N, const float *x, const float *params, float *y, const float *gy, float *gx,
                        float *gparams, double *partials, hipStream_t s) {
    using S = WideShape<MB, IN, H, NH, OUT>;
    const int64_t tiles = (N + MB - 1) / MB;
    int64_t blocks = (tiles + kMfmaWaves - 1) / kMfmaWaves;
    // narrow decoders are memory-bound: several workgroups per CU hide the load latency; wide ones fill the registers
    // forward: no per-block epilogue, so many small workgroups; backward: every workgroup pays the weight load, the
    // gradient flush and one partial row for the finishing pass
    const int64_t max_blocks = (MB == 16) ? (bwd ? kMlpNarrowMaxBlocks : 2048) : kMfmaMaxBlocks;
    static_assert(kMfmaMaxBlocks <= kMlpMaxBlocks, "the partial-gradient workspace has kMlpMaxBlocks rows");
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    constexpr size_t fwd_lds = (size_t)S::lds_weights * sizeof(float);
    constexpr size_t bwd_lds = S::bwd_lds_floats * sizeof(float);
    static_assert(bwd_lds <= 160 * 1024, "backward LDS plan exceeds the CU's 160 KiB");
    static PerDeviceOnce once;
    const hipError_t attr_err = once.run([]() -> hipError_t {
        hipError_t e = hipSuccess;
        if (fwd_lds > 64 * 1024)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_mlp_forward_kernel<MB, IN, H, NH, OUT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds);
        if (e == hipSuccess && bwd_lds > 64 * 1024) {
            if constexpr (S::kSplit)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&split_mlp_backward_kernel<MB, IN, H, NH, OUT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds);
            else
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_mlp_backward_kernel<MB, IN, H, NH, OUT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds);
        }
        return e;
    });
    if (attr_err != hipSuccess) return attr_err;
    if (!bwd) {
        hipLaunchKernelGGL((wide_mlp_forward_kernel<MB, IN, H, NH, OUT>), dim3((uint32_t)blocks), dim3(64 * kMfmaWaves),
                           fwd_lds, s, x, params, y, N);
        return hipGetLastError();
    }
    if constexpr (S::kSplit)
        hipLaunchKernelGGL((split_mlp_backward_kernel<MB, IN, H, NH, OUT>), dim3((uint32_t)blocks), dim3(64 * kMfmaWaves),
                           bwd_lds, s, x, params, gy, gx, partials, N);
    else
        hipLaunchKernelGGL((wide_mlp_backward_kernel<MB, IN, H, NH, OUT>), dim3((uint32_t)blocks), dim3(64 * kMfmaWaves),
                           bwd_lds, s, x, params, gy, gx, partials, N);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return mlp_finish_launch(partials, (int)blocks, S::n_params, gparams, s);
}

// the instantiations mlp.hip's lookup table refers to
#define SHACIRA_WIDE_INST(MB, IN, H, NH, OUT)                                                                        \
    template hipError_t wide_mlp_run<MB, IN, H, NH, OUT>(bool, int64_t, const float *, const float *, float *,       \
                                                         const float *, float *, float *, double *, hipStream_t);
SHACIRA_WIDE_INST(32, 32, 64, 1, 16)
SHACIRA_WIDE_INST(32, 43, 64, 2, 3)
SHACIRA_WIDE_INST(32, 32, 64, 2, 3)
SHACIRA_WIDE_INST(32, 16, 64, 2, 3)
SHACIRA_WIDE_INST(32, 32, 64, 1, 3)
SHACIRA_WIDE_INST(32, 96, 128, 1, 16)
SHACIRA_WIDE_INST(32, 43, 128, 2, 3)
SHACIRA_WIDE_INST(16, 32, 16, 2, 3)
SHACIRA_WIDE_INST(16, 24, 16, 2, 3)
SHACIRA_WIDE_INST(16, 16, 16, 2, 3)
SHACIRA_WIDE_INST(16, 48, 16, 2, 3)
SHACIRA_WIDE_INST(16, 32, 16, 1, 3)
SHACIRA_WIDE_INST(16, 32, 16, 2, 4)
#undef SHACIRA_WIDE_INST

}  // namespace shacira
