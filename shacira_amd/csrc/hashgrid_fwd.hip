// hashgrid_fwd.hip -- multi-resolution hash-grid lookup + bi/trilinear interpolation, forward (gfx950).
//
// Replaces hashgrid_interpolate_cuda / hashgrid_interpolate2d_cuda
// (wisp/csrc/ops/hashgrid_interpolate.cpp:44-66, :130-152; kernels hashgrid_interpolate_cuda.cu:47-109 and
// hashgrid_interpolate2d_cuda.cu:44-99). One launch covers every level (the reference launches L kernels).
//
// Work decomposition ("lane = (sample, level)"): work item w = sample * L + level, one per lane, so that
//   - the feats store is one contiguous F*sizeof(T)-byte piece per lane, consecutive lanes consecutive addresses
//     (a wave writes 64*F*sizeof(T) contiguous bytes; the reference strides lanes by L*F*sizeof(T));
//   - coords are fetched once per sample by L neighbouring lanes (same address -> one request);
//   - each corner row (F scalars) is fetched by ONE vector load of F*sizeof(T) bytes.
// HBM-bound: algorithmic bytes per sample = 4*DIM + L*2^DIM*F*s + L*F*s.
#include "internal.h"

namespace shacira {

template <typename T, int F> struct RowVec;  // one table row as a single vector access
template <> struct RowVec<float, 2> { using type = float2; };
template <> struct RowVec<float, 4> { using type = float4; };
template <> struct RowVec<__half, 2> { using type = uint32_t; };
template <> struct RowVec<__half, 4> { using type = uint2; };
template <> struct RowVec<__half, 8> { using type = uint4; };

template <typename T, int F> __device__ __forceinline__ void load_row(const T *p, float (&v)[F]) {
    if constexpr (sizeof(T) == 4 && F == 2) {
        float2 r = *reinterpret_cast<const float2 *>(p);
        v[0] = r.x; v[1] = r.y;
    } else if constexpr (sizeof(T) == 4 && F == 4) {
        float4 r = *reinterpret_cast<const float4 *>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    } else if constexpr (sizeof(T) == 2 && F == 2) {
        __half2 r = *reinterpret_cast<const __half2 *>(p);
        v[0] = __low2float(r); v[1] = __high2float(r);
    } else {
#pragma unroll
        for (int j = 0; j < F; ++j) v[j] = Scalar<T>::load(p + j);
    }
}

template <typename T, int F> __device__ __forceinline__ void store_row(T *p, const float (&v)[F]) {
    if constexpr (sizeof(T) == 4 && F == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else if constexpr (sizeof(T) == 4 && F == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (sizeof(T) == 2 && F == 2) {
        *reinterpret_cast<__half2 *>(p) = __floats2half2_rn(v[0], v[1]);
    } else {
#pragma unroll
        for (int j = 0; j < F; ++j) Scalar<T>::store(p + j, v[j]);
    }
}

// F > 0: compile-time feature_dim; F == 0: runtime feature_dim (any even value), scalar row access.
template <int DIM, typename T, int F>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(LevelTable lt, const int32_t *__restrict__ first_idx,
                                                           const float *__restrict__ coords,
                                                           const T *__restrict__ table, T *__restrict__ feats,
                                                           int64_t sample0, uint32_t num_items) {
    constexpr int NC = 1 << DIM;
    const uint32_t L = (uint32_t)lt.num_lods;
    __shared__ int32_t s_res[SHACIRA_MAX_LODS];
    __shared__ float s_hi[SHACIRA_MAX_LODS];
    __shared__ int32_t s_first[SHACIRA_MAX_LODS];
    __shared__ uint8_t s_dense[SHACIRA_MAX_LODS];
    if (threadIdx.x < L) {
        s_res[threadIdx.x] = lt.res[threadIdx.x];
        s_hi[threadIdx.x] = lt.hi[threadIdx.x];
        s_dense[threadIdx.x] = lt.dense[threadIdx.x];
        s_first[threadIdx.x] = first_idx[threadIdx.x];
    }
    __syncthreads();

    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < num_items; w += stride) {
        const uint32_t s = w / L;
        const uint32_t lvl = w - s * L;
        const int64_t i = sample0 + s;
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
        Corners<DIM> c;
        compute_corners<DIM>(t, s_res[lvl], s_hi[lvl], s_dense[lvl] != 0, lt.mask, c);
        const int64_t base = (int64_t)s_first[lvl];
        if constexpr (F > 0) {
            float acc[F];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int64_t row = base + (int64_t)c.row[k];
                float v[F];
                if ((uint64_t)row < (uint64_t)lt.table_rows) {
                    load_row<T, F>(table + row * F, v);
                } else {
#pragma unroll
                    for (int j = 0; j < F; ++j) v[j] = 0.0f;
                }
#pragma unroll
                for (int j = 0; j < F; ++j) acc[j] = (k == 0) ? v[j] * c.w[0] : fmaf(v[j], c.w[k], acc[j]);
            }
            store_row<T, F>(feats + (i * L + lvl) * F, acc);
        } else {
            const int Fr = lt.feature_dim;
            for (int j = 0; j < Fr; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int64_t row = base + (int64_t)c.row[k];
                    float v = ((uint64_t)row < (uint64_t)lt.table_rows) ? Scalar<T>::load(table + row * Fr + j) : 0.0f;
                    acc = (k == 0) ? v * c.w[0] : fmaf(v, c.w[k], acc);
                }
                Scalar<T>::store(feats + (i * L + lvl) * Fr + j, acc);
            }
        }
    }
}

template <int DIM, typename T, int F>
static hipError_t launch_fwd(const LevelTable &lt, const int32_t *first_idx, const float *coords, const void *table,
                             void *feats, int64_t num_coords, hipStream_t stream) {
    // items are indexed with 32 bits inside a launch; chunk the samples so that samples*L < 2^31
    const int64_t L = lt.num_lods;
    const int64_t max_samples = ((int64_t)1 << 31) / L - 1;
    for (int64_t s0 = 0; s0 < num_coords; s0 += max_samples) {
        const int64_t ns = (num_coords - s0 < max_samples) ? (num_coords - s0) : max_samples;
        const uint32_t items = (uint32_t)(ns * L);
        const uint32_t blocks = (items + 255u) / 256u;
        hipLaunchKernelGGL((hashgrid_fwd_kernel<DIM, T, F>), dim3(blocks), dim3(256), 0, stream, lt, first_idx, coords,
                           static_cast<const T *>(table), static_cast<T *>(feats), s0, items);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int DIM, typename T>
static hipError_t dispatch_f(const LevelTable &lt, const int32_t *first_idx, const float *coords, const void *table,
                             void *feats, int64_t n, hipStream_t s) {
    switch (lt.feature_dim) {
        case 2: return launch_fwd<DIM, T, 2>(lt, first_idx, coords, table, feats, n, s);
        case 4: return launch_fwd<DIM, T, 4>(lt, first_idx, coords, table, feats, n, s);
        default: return launch_fwd<DIM, T, 0>(lt, first_idx, coords, table, feats, n, s);
    }
}

hipError_t hashgrid_forward_dispatch(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx,
                                     const float *coords, const void *table, void *feats, int64_t n,
                                     hipStream_t s) {
    if (dim == 3) {
        return dtype == SHACIRA_F32 ? dispatch_f<3, float>(lt, first_idx, coords, table, feats, n, s)
                                    : dispatch_f<3, __half>(lt, first_idx, coords, table, feats, n, s);
    }
    return dtype == SHACIRA_F32 ? dispatch_f<2, float>(lt, first_idx, coords, table, feats, n, s)
                                : dispatch_f<2, __half>(lt, first_idx, coords, table, feats, n, s);
}

}  // namespace shacira
