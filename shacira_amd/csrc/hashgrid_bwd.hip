// hashgrid_bwd.hip -- gradient scatter-add into the codebook, backward of the hash-grid lookup (gfx950).
//
// Replaces hashgrid_interpolate_backward_cuda / hashgrid_interpolate2d_backward_cuda
// (wisp/csrc/ops/hashgrid_interpolate.cpp:68-100, :154-186; kernels hashgrid_interpolate_cuda.cu:143-221 and
// hashgrid_interpolate2d_cuda.cu:133-208):  grad_codebook[first[l] + row_k, j] += grad_output[i, l*F + j] * w_k.
//
// Variant 0 ("atomic"): lane = (sample, level); per corner one global_atomic_add_f32 per feature. The
//   accumulator is always fp32: for fp16 tables the sums are kept in an fp32 workspace and rounded once at the
//   end (the reference rounds every __half2 atomicAdd, .cu:198-211; ours is the more accurate of the two).
#include "internal.h"

namespace shacira {

template <int DIM, typename T, int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_atomic_kernel(LevelTable lt, const int32_t *__restrict__ first_idx,
                                                                  const float *__restrict__ coords,
                                                                  const T *__restrict__ grad_out,
                                                                  float *__restrict__ grad_table, int64_t sample0,
                                                                  uint32_t num_items) {
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
    const int Fr = (F > 0) ? F : lt.feature_dim;

    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < num_items; w += stride) {
        const uint32_t s = w / L;
        const uint32_t lvl = w - s * L;
        if ((int32_t)lvl < lt.level_begin || (int32_t)lvl >= lt.level_end) continue;
        const int64_t i = sample0 + s;
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
        Corners<DIM> c;
        compute_corners<DIM>(t, s_res[lvl], s_hi[lvl], s_dense[lvl] != 0, lt.mask, c);
        const int64_t base = (int64_t)s_first[lvl];
        const T *g = grad_out + (i * L + lvl) * Fr;
        for (int j = 0; j < Fr; ++j) {
            const float gj = Scalar<T>::load(g + j);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int64_t row = base + (int64_t)c.row[k];
                if ((uint64_t)row < (uint64_t)lt.table_rows)
                    unsafeAtomicAdd(grad_table + row * Fr + j, gj * c.w[k]);
            }
        }
    }
}

// F = 2, lane PAIR per (sample, level) (round 5): lane parity = feature, so the two atomics of a row sit in adjacent lanes
// of ONE instruction and leave the CU as one 8-byte request -- the memory-side atomic units retire ~41 G such requests per
// second against ~10 G/s for the two-instructions-per-lane form above (profiles/r01_microbench_rates.txt: "atomic f32 pair
// (lane pair)" vs "(2/thread)"). Both lanes compute the corners (the arithmetic is nothing beside the atomics). This is what
// small batches on large tables run (a few thousand ray points on the 48.8 MB NeRF table), where the binned form's
// table-sized passes cost more than the atomics.
template <int DIM, typename T, int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_atomic_pair_kernel(LevelTable lt, const int32_t *__restrict__ first_idx,
                                                                       const float *__restrict__ coords,
                                                                       const T *__restrict__ grad_out,
                                                                       float *__restrict__ grad_table, int64_t sample0,
                                                                       uint32_t num_items) {
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
    static_assert(F == 2 || F == 4, "2 F lanes per (sample, level)");
    constexpr uint32_t LOGF = (F == 2) ? 1u : 2u;
    // the grid covers every lane (launch_bwd_atomic): ONE guarded pass -- a grid-stride loop on a 32-bit index wrapped for the
    // last threads of a full-size chunk (items * 2 F close to 2^31) and added the first items twice (round-5 advisor finding)
    const uint32_t t2 = blockIdx.x * blockDim.x + threadIdx.x;
    if ((t2 >> (LOGF + 1u)) < num_items) {
        // four lanes per (sample, level): (x offset, feature). The rows x and x + 1 of a corner pair are neighbours (dense
        // levels) or differ in a few low bits (hashed levels: x ^ (x + 1)), so the four lanes' atomics of one instruction
        // mostly fall into one 64-byte piece and travel as ONE request: 8 192 ray points 67 -> ~45 us (pairs only: 121 -> 67)
        // (F = 4: eight lanes, a row's four features = one 16-byte piece)
        const uint32_t w = t2 >> (LOGF + 1u), dx = (t2 >> LOGF) & 1u, j = t2 & (uint32_t)(F - 1);
        const uint32_t s = w / L;
        const uint32_t lvl = w - s * L;
        if ((int32_t)lvl < lt.level_begin || (int32_t)lvl >= lt.level_end) return;
        const int64_t i = sample0 + s;
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
        Corners<DIM> c;
        compute_corners<DIM>(t, s_res[lvl], s_hi[lvl], s_dense[lvl] != 0, lt.mask, c);
        const int64_t base = (int64_t)s_first[lvl];
        const float gj = Scalar<T>::load(grad_out + (i * L + lvl) * F + j);
#pragma unroll
        for (int q = 0; q < NC / 2; ++q) {   // corner bit (DIM - 1) is the x offset (.cu:88-94, 2d.cu:83-88)
            const int k0 = q, k1 = q | (NC / 2);
            const uint32_t r = dx ? c.row[k1] : c.row[k0];
            const float wk = dx ? c.w[k1] : c.w[k0];
            const int64_t row = base + (int64_t)r;
            if ((uint64_t)row < (uint64_t)lt.table_rows) unsafeAtomicAdd(grad_table + row * F + j, gj * wk);
        }
    }
}

// fp64 tables: contribution = (float)(grad * weight) with the product formed in double (`float grad = grad_output[..] *
// coeffs[k]`, .cu:215-217, scalar_t = double), accumulated with atomicAdd(double). The reference then adds that float
// through `(float*)(grad_codebook + ...)` -- into the low word of each double, a bug; this is the intended gradient.
template <int DIM>
__global__ __launch_bounds__(256) void hashgrid_bwd_atomic_f64_kernel(LevelTable lt, const int32_t *__restrict__ first_idx,
                                                                      const float *__restrict__ coords,
                                                                      const double *__restrict__ grad_out,
                                                                      double *__restrict__ grad_table, int64_t sample0,
                                                                      uint32_t num_items) {
    constexpr int NC = 1 << DIM;
    const uint32_t L = (uint32_t)lt.num_lods;
    const int Fr = lt.feature_dim;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < num_items; w += stride) {
        const uint32_t s = w / L;
        const uint32_t lvl = w - s * L;
        const int64_t i = sample0 + s;
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
        Corners<DIM> c;
        compute_corners<DIM>(t, lt.res[lvl], lt.hi[lvl], lt.dense[lvl] != 0, lt.mask, c);
        const int64_t base = (int64_t)first_idx[lvl];
        const double *g = grad_out + (i * L + lvl) * Fr;
        for (int j = 0; j < Fr; ++j) {
            const double gj = g[j];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int64_t row = base + (int64_t)c.row[k];
                if ((uint64_t)row < (uint64_t)lt.table_rows)
                    unsafeAtomicAdd(grad_table + row * Fr + j, (double)(float)(gj * (double)c.w[k]));
            }
        }
    }
}

template <int DIM>
static hipError_t launch_bwd_atomic_f64(const LevelTable &lt, const int32_t *first_idx, const float *coords,
                                        const void *grad_out, void *grad_table, int64_t num_coords, hipStream_t stream) {
    const int64_t L = lt.num_lods;
    const int64_t max_samples = ((int64_t)1 << 31) / L - 1;
    for (int64_t s0 = 0; s0 < num_coords; s0 += max_samples) {
        const int64_t ns = (num_coords - s0 < max_samples) ? (num_coords - s0) : max_samples;
        const uint32_t items = (uint32_t)(ns * L);
        hipLaunchKernelGGL(hashgrid_bwd_atomic_f64_kernel<DIM>, dim3((items + 255u) / 256u), dim3(256), 0, stream, lt,
                           first_idx, coords, static_cast<const double *>(grad_out), static_cast<double *>(grad_table), s0,
                           items);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float *__restrict__ src, __half *__restrict__ dst,
                                                         int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = __float2half_rn(src[i]);
}

// fp32 image -> half table for the rows of levels [level_begin, level_end) only: a level-range call (the all-reduce overlap of
// shacira_hashgrid_backward_levels) used to convert the WHOLE table every time
__global__ __launch_bounds__(256) void f32_to_f16_levels_kernel(const float *__restrict__ src, __half *__restrict__ dst,
                                                               const int32_t *__restrict__ first_idx, int level_begin,
                                                               int level_end, int num_lods, int64_t table_rows, int F) {
    const int64_t lo = (int64_t)first_idx[level_begin] * F;
    const int64_t hi = ((level_end < num_lods) ? (int64_t)first_idx[level_end] : table_rows) * F;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; e < hi; e += stride) dst[e] = __float2half_rn(src[e]);
}

// zeroes the rows of levels [level_begin, level_end) (first_idx lives on the device)
__global__ __launch_bounds__(256) void zero_level_rows_kernel(float *__restrict__ acc,
                                                              const int32_t *__restrict__ first_idx, int level_begin,
                                                              int level_end, int num_lods, int64_t table_rows, int F) {
    const int64_t lo = (int64_t)first_idx[level_begin] * F;
    const int64_t hi = ((level_end < num_lods) ? (int64_t)first_idx[level_end] : table_rows) * F;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; e < hi; e += stride) acc[e] = 0.0f;
}

// Streaming zero fill as a kernel of the library (16 bytes per lane, scalar head / tail). Used instead of hipMemsetAsync on
// every path that may be captured into a HIP graph: a captured memset node was seen not to take effect on replay (round 3).
__global__ __launch_bounds__(256) void zero_fill_kernel(float *__restrict__ p, int64_t n) {
    const int64_t head = (int64_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4;   // floats up to 16-byte alignment
    const int64_t h = head < n ? head : n;
    const int64_t nv = (n - h) / 4;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t0 < h) p[t0] = 0.0f;
    float4 *v = reinterpret_cast<float4 *>(p + h);
    for (int64_t e = t0; e < nv; e += stride) v[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int64_t tail0 = h + nv * 4;
    if (tail0 + t0 < n) p[tail0 + t0] = 0.0f;
}

hipError_t zero_fill_async(float *p, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, p, n);
    return hipGetLastError();
}

template <int DIM, typename T, int F>
static hipError_t launch_bwd_atomic(const LevelTable &lt, const int32_t *first_idx, const float *coords,
                                    const void *grad_out, float *acc, int64_t num_coords, hipStream_t stream) {
    const int64_t L = lt.num_lods;
    const int64_t max_samples = ((int64_t)1 << (F == 2 ? 29 : F == 4 ? 28 : 31)) / L - 1;
    for (int64_t s0 = 0; s0 < num_coords; s0 += max_samples) {
        const int64_t ns = (num_coords - s0 < max_samples) ? (num_coords - s0) : max_samples;
        const uint32_t items = (uint32_t)(ns * L);
        if constexpr (F == 2 || F == 4) {   // 2 F lanes per (sample, level) (items * 2 F < 2^31: see max_samples)
            const uint32_t blocks2 = (uint32_t)(((uint64_t)items * (2u * F) + 255u) / 256u);
            hipLaunchKernelGGL((hashgrid_bwd_atomic_pair_kernel<DIM, T, F>), dim3(blocks2), dim3(256), 0, stream, lt, first_idx,
                               coords, static_cast<const T *>(grad_out), acc, s0, items);
            hipError_t e2 = hipGetLastError();
            if (e2 != hipSuccess) return e2;
        } else {   // any other (even) feature count: the reference's shape, one thread per (sample, level)
            const uint32_t blocks = (items + 255u) / 256u;
            hipLaunchKernelGGL((hashgrid_bwd_atomic_kernel<DIM, T, F>), dim3(blocks), dim3(256), 0, stream, lt, first_idx,
                               coords, static_cast<const T *>(grad_out), acc, s0, items);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

template <int DIM, typename T>
static hipError_t bwd_atomic_f(const LevelTable &lt, const int32_t *first_idx, const float *coords,
                               const void *grad_out, float *acc, int64_t n, hipStream_t s) {
    if (lt.feature_dim == 2) return launch_bwd_atomic<DIM, T, 2>(lt, first_idx, coords, grad_out, acc, n, s);
    if (lt.feature_dim == 4) return launch_bwd_atomic<DIM, T, 4>(lt, first_idx, coords, grad_out, acc, n, s);
    return launch_bwd_atomic<DIM, T, 0>(lt, first_idx, coords, grad_out, acc, n, s);
}

// hashgrid_bwd_bin.hip
bool bin_supported(int dim, const LevelTable &lt);
bool bin_all_direct(int dim, const LevelTable &lt);
size_t bin_workspace_bytes(int dim, int dtype, const LevelTable &lt, int64_t n);
size_t bin_workspace_bytes_planned(int dim, int dtype, const LevelTable &lt, int64_t n, bool whole, bool grad_aligned);
float *bin_acc32(int dim, int dtype, const LevelTable &lt, int64_t n, void *workspace);
hipError_t bin_backward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                        const void *grad_out, float *acc, void *workspace, int64_t n, hipStream_t s, bool zero_table,
                        void *half_table, bool *converted, const SortedBatch *sb);

// variant 1 ("bin") whenever the shape allows it and the batch is big enough to amortise its fixed passes
static bool use_bin(int dim, const LevelTable &lt, int64_t n) {
    const int v = opt().bwd_variant;
    if (v == 0 || !bin_supported(dim, lt)) return false;
    if (v == 1) return true;
    // tables whose levels all fit LDS images need no partitioning pass at all: one kernel, ahead of the scattered atomics
    // from ~2 K samples (Kodak table: 21.7 vs 49.8 us at 4 096 samples, 21.4 vs 18.6 us at 1 024)
    // (config D's 48.8 MB table: binned 62 us at 4 096 - 8 192 samples against 66 / 122 us for the atomics; 64 vs 40 us at 2 048)
    // Round 5, with the lane-group atomics (2 F lanes per (sample, level): one request per x-pair of rows instead of four):
    // config D's table 20 / 29 / 47 / 82 us at 2 048 / 4 096 / 8 192 / 16 384 samples against 58-64 us for the binned form
    // (64 vs 62 at 12 288); 2-D bw-19 table 36 / 38 / 65 / 118 us at 4 096 / 8 192 / 16 384 / 32 768 against 65-69
    if (bin_all_direct(dim, lt)) return n >= 2048;
    // (nerf_lego.yaml's F = 4 table: 54 / 77 / 126 us at 4 096 / 8 192 / 16 384 against 115-121)
    if (lt.feature_dim == 2) return n >= (dim == 3 ? 12288 : 16384);
    if (lt.feature_dim == 4) return n >= 16384;
    return n >= 4096;
}

size_t hashgrid_backward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n) {
    if (dtype == SHACIRA_F64) return 0;   // atomicAdd(double) straight into the caller's table
    size_t need = (dtype == SHACIRA_F16) ? (size_t)lt.table_rows * lt.feature_dim * sizeof(float) : 0;
    if (bin_supported(dim, lt) && n > 0) {
        const size_t b = bin_workspace_bytes(dim, dtype, lt, n);
        if (b > need) need = b;
    }
    return need;
}

size_t hashgrid_backward_workspace_planned(int dim, int dtype, const LevelTable &lt, int64_t n, bool grad_aligned) {
    if (dtype == SHACIRA_F64) return 0;
    size_t need = (dtype == SHACIRA_F16) ? (size_t)lt.table_rows * lt.feature_dim * sizeof(float) : 0;
    if (bin_supported(dim, lt) && n > 0) {
        const bool whole = lt.level_begin == 0 && lt.level_end == lt.num_lods;
        // (a batch below the binned form's threshold runs the atomic form: no items at all; sized like the plain query, which
        // does not look at the threshold either)
        const size_t b = (n < ((int64_t)1 << 31)) ? bin_workspace_bytes_planned(dim, dtype, lt, n, whole, grad_aligned)
                                                  : bin_workspace_bytes(dim, dtype, lt, n);
        if (b > need) need = b;
    }
    return need;
}

hipError_t hashgrid_backward_dispatch(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx,
                                      const float *coords, const void *grad_out, void *grad_table, void *workspace,
                                      size_t workspace_bytes, int64_t n, hipStream_t s, const void *plan) {
    (void)workspace_bytes;
    // the batch's plan (hashgrid_tiled.hip), when the caller kept the one its forward call built
    SortedBatch sbv{};
    const SortedBatch *sb = nullptr;
    if (plan != nullptr && n > 0 && n < ((int64_t)1 << 31)) {
        sample_plan_view(dim, n, plan, sbv);
        sb = &sbv;
    }
    const int64_t numel = lt.table_rows * lt.feature_dim;
    if (dtype == SHACIRA_F64) {   // the reference-shaped form only: zeros_like, then one atomicAdd(double) per corner and feature
        hipError_t e64 = zero_fill_async(static_cast<float *>(grad_table), 2 * numel, s);
        if (e64 != hipSuccess || n <= 0) return e64;
        return dim == 3 ? launch_bwd_atomic_f64<3>(lt, first_idx, coords, grad_out, grad_table, n, s)
                        : launch_bwd_atomic_f64<2>(lt, first_idx, coords, grad_out, grad_table, n, s);
    }
    const bool bin = n > 0 && use_bin(dim, lt, n);
    // fp16 tables accumulate in an fp32 image: the tail of the bin workspace, or the whole workspace (atomic variant)
    float *acc = static_cast<float *>(grad_table);
    if (dtype == SHACIRA_F16) {
        acc = bin ? bin_acc32(dim, dtype, lt, n, workspace) : static_cast<float *>(workspace);
    }
    hipError_t e = hipSuccess;
    const bool full = lt.level_begin == 0 && lt.level_end == lt.num_lods;
    if (full && bin) {
        // zeroed inside bin_backward (on its side stream, next to the transpose, when it forks)
    } else if (full) {
        e = zero_fill_async(acc, numel, s);  // at::zeros_like, .cpp:81/:167
    } else {
        hipLaunchKernelGGL(zero_level_rows_kernel, dim3(2048), dim3(256), 0, s, acc, first_idx, lt.level_begin,
                           lt.level_end, lt.num_lods, lt.table_rows, lt.feature_dim);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return e;
    bool converted = false;   // fp16 tables: the binned path may write the half table itself (single-unit buckets + a skipping conversion)
    if (bin) {
        e = bin_backward(dim, dtype, lt, first_idx, coords, grad_out, acc, workspace, n, s, full,
                         (dtype == SHACIRA_F16 && full) ? grad_table : nullptr, &converted, sb);
        if (e != hipSuccess) return e;
    } else if (n > 0) {
        if (dim == 3) {
            e = (dtype == SHACIRA_F32) ? bwd_atomic_f<3, float>(lt, first_idx, coords, grad_out, acc, n, s)
                                       : bwd_atomic_f<3, __half>(lt, first_idx, coords, grad_out, acc, n, s);
        } else {
            e = (dtype == SHACIRA_F32) ? bwd_atomic_f<2, float>(lt, first_idx, coords, grad_out, acc, n, s)
                                       : bwd_atomic_f<2, __half>(lt, first_idx, coords, grad_out, acc, n, s);
        }
        if (e != hipSuccess) return e;
    }
    if (dtype == SHACIRA_F16 && !converted) {
        int64_t blocks = (numel + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks > 0 && full)
            hipLaunchKernelGGL(f32_to_f16_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, acc,
                               static_cast<__half *>(grad_table), numel);
        else if (blocks > 0)   // only the rows this call computed (the other levels' rows of the half table stay as they are)
            hipLaunchKernelGGL(f32_to_f16_levels_kernel, dim3(2048), dim3(256), 0, s, acc, static_cast<__half *>(grad_table),
                               first_idx, lt.level_begin, lt.level_end, lt.num_lods, lt.table_rows, lt.feature_dim);
        return hipGetLastError();
    }
    return hipSuccess;
}

}  // namespace shacira
