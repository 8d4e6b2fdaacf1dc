// cell_sort.hip -- orders the samples of a batch by the Morton code of a coarse spatial cell (gfx950).
//
// Why: a wave-level gather costs ~20 clk + 2 clk per DISTINCT 128-byte line. Uniformly random
// samples give every lane its own line at every level; samples that sit in the same coarse cell share table lines at
// the coarse and middle levels (a cell of the 32^3 sort grid spans < 2 grid cells of a level up to res ~ 60 and the
// lines of a level run along x). The sort is a one-pass counting sort with LDS histograms:
//   count    per tile of 8192 samples: histogram of the 2^15 cell keys in LDS -> cnt[tile][key]
//   scan     per key: exclusive scan over the tiles; then exclusive scan over the keys -> base[key]
//   scatter  per tile: rank of every sample inside its (tile, key) run via LDS atomics; writes perm[pos] = sample and
//            the coordinates in sorted order
// No ordering is imposed inside a cell (not needed). Everything is recomputed per call: no state survives the call.
#include "internal.h"

namespace shacira {

constexpr int kSortKeys = 1 << 15;
constexpr int kSortTile = 8192;
constexpr int kSortThreads = 1024;

template <int DIM> __device__ __forceinline__ uint32_t cell_key(const float *__restrict__ c) {
    // cell = floor(unit * G) clamped; NaN -> 0. 3-D: 32^3 cells, 2-D: 128 x 256 cells (x finest so lines stay together)
    auto cell = [](float v, int g) {
        float u = (v * 0.5f + 0.5f) * (float)g;
        int q = (u >= 0.0f) ? (int)u : 0;       // NaN compares false -> 0
        return (uint32_t)(q < g ? q : g - 1);
    };
    if constexpr (DIM == 3) {
        const uint32_t x = cell(c[0], 32), y = cell(c[1], 32), z = cell(c[2], 32);
        uint32_t k = 0;  // Morton: z,y,x bit-interleaved, x lowest
#pragma unroll
        for (int b = 0; b < 5; ++b)
            k |= (((x >> b) & 1u) << (3 * b)) | (((y >> b) & 1u) << (3 * b + 1)) | (((z >> b) & 1u) << (3 * b + 2));
        return k;
    } else {
        const uint32_t x = cell(c[0], 256), y = cell(c[1], 128);
        return (y << 8) | x;   // row-major cells: coords[0] is the fast axis of the tables (row = x + y*res)
    }
}

template <int DIM>
__global__ __launch_bounds__(kSortThreads) void sort_count_kernel(const float *__restrict__ coords, int64_t N,
                                                                  uint32_t *__restrict__ cnt) {
    extern __shared__ uint32_t s_hist[];  // [kSortKeys]
    for (int k = threadIdx.x; k < kSortKeys; k += kSortThreads) s_hist[k] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * kSortTile;
    for (int k = threadIdx.x; k < kSortTile; k += kSortThreads) {
        const int64_t i = s0 + k;
        if (i < N) atomicAdd(&s_hist[cell_key<DIM>(coords + i * DIM)], 1u);
    }
    __syncthreads();
    uint32_t *row = cnt + (size_t)blockIdx.x * kSortKeys;
    for (int k = threadIdx.x; k < kSortKeys; k += kSortThreads) row[k] = s_hist[k];
}

// thread per key: exclusive scan over the tiles, total -> totals[key]
__global__ __launch_bounds__(256) void sort_scan_tiles_kernel(uint32_t *__restrict__ cnt, uint32_t *__restrict__ totals,
                                                              int num_tiles) {
    const int key = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int t = 0; t < num_tiles; ++t) {
        uint32_t *p = cnt + (size_t)t * kSortKeys + key;
        const uint32_t c = *p;
        *p = acc;
        acc += c;
    }
    totals[key] = acc;
}

// one block: exclusive scan of the 2^15 totals (32 per thread)
__global__ __launch_bounds__(1024) void sort_scan_keys_kernel(const uint32_t *__restrict__ totals,
                                                              uint32_t *__restrict__ base) {
    __shared__ uint32_t s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v[32], sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v[k] = totals[t * 32 + k];
        sum += v[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - sum;
    for (int w = 0; w < wave; ++w) pre += s_wave[w];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        base[t * 32 + k] = pre;
        pre += v[k];
    }
}

template <int DIM>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const float *__restrict__ coords, int64_t N,
                                                                    const uint32_t *__restrict__ tile_off,
                                                                    const uint32_t *__restrict__ base,
                                                                    uint32_t *__restrict__ perm,
                                                                    float *__restrict__ sorted_coords) {
    extern __shared__ uint32_t s_hist[];
    for (int k = threadIdx.x; k < kSortKeys; k += kSortThreads) s_hist[k] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * kSortTile;
    const uint32_t *off = tile_off + (size_t)blockIdx.x * kSortKeys;
    for (int k = threadIdx.x; k < kSortTile; k += kSortThreads) {
        const int64_t i = s0 + k;
        if (i >= N) break;
        float c[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[a] = coords[i * DIM + a];
        const uint32_t key = cell_key<DIM>(c);
        const uint32_t rank = atomicAdd(&s_hist[key], 1u);
        const uint32_t pos = base[key] + off[key] + rank;
        perm[pos] = (uint32_t)i;
#pragma unroll
        for (int a = 0; a < DIM; ++a) sorted_coords[(size_t)pos * DIM + a] = c[a];
    }
}

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

size_t cell_sort_workspace_bytes(int dim, int64_t n) {
    const size_t tiles = (size_t)((n + kSortTile - 1) / kSortTile);
    return up256((size_t)n * sizeof(uint32_t)) + up256((size_t)n * dim * sizeof(float)) +
           up256(tiles * kSortKeys * sizeof(uint32_t)) + 2 * up256(kSortKeys * sizeof(uint32_t));
}

// Carves [perm | sorted coords | cnt | totals | base] out of `ws` and runs the sort on stream s.
hipError_t cell_sort(int dim, const float *coords, int64_t n, void *ws, uint32_t **perm_out, float **sorted_out,
                     hipStream_t s) {
    if (n >= ((int64_t)1 << 32)) return hipErrorInvalidValue;
    unsigned char *p = static_cast<unsigned char *>(ws);
    const size_t tiles = (size_t)((n + kSortTile - 1) / kSortTile);
    uint32_t *perm = reinterpret_cast<uint32_t *>(p);
    p += up256((size_t)n * sizeof(uint32_t));
    float *sorted = reinterpret_cast<float *>(p);
    p += up256((size_t)n * dim * sizeof(float));
    uint32_t *cnt = reinterpret_cast<uint32_t *>(p);
    p += up256(tiles * kSortKeys * sizeof(uint32_t));
    uint32_t *totals = reinterpret_cast<uint32_t *>(p);
    p += up256(kSortKeys * sizeof(uint32_t));
    uint32_t *base = reinterpret_cast<uint32_t *>(p);
    *perm_out = perm;
    *sorted_out = sorted;
    static PerDeviceOnce once;
    const hipError_t attr_err = once.run([]() -> hipError_t {
        const int bytes = kSortKeys * (int)sizeof(uint32_t);
        const void *fns[] = {reinterpret_cast<const void *>(&sort_count_kernel<2>),
                             reinterpret_cast<const void *>(&sort_count_kernel<3>),
                             reinterpret_cast<const void *>(&sort_scatter_kernel<2>),
                             reinterpret_cast<const void *>(&sort_scatter_kernel<3>)};
        hipError_t err = hipSuccess;
        for (const void *fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) err = e;
        }
        return err;
    });
    if (attr_err != hipSuccess) return attr_err;
    const size_t shmem = kSortKeys * sizeof(uint32_t);
    if (dim == 3)
        hipLaunchKernelGGL(sort_count_kernel<3>, dim3((uint32_t)tiles), dim3(kSortThreads), shmem, s, coords, n, cnt);
    else
        hipLaunchKernelGGL(sort_count_kernel<2>, dim3((uint32_t)tiles), dim3(kSortThreads), shmem, s, coords, n, cnt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sort_scan_tiles_kernel, dim3(kSortKeys / 256), dim3(256), 0, s, cnt, totals, (int)tiles);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sort_scan_keys_kernel, dim3(1), dim3(1024), 0, s, totals, base);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (dim == 3)
        hipLaunchKernelGGL(sort_scatter_kernel<3>, dim3((uint32_t)tiles), dim3(kSortThreads), shmem, s, coords, n, cnt,
                           base, perm, sorted);
    else
        hipLaunchKernelGGL(sort_scatter_kernel<2>, dim3((uint32_t)tiles), dim3(kSortThreads), shmem, s, coords, n, cnt,
                           base, perm, sorted);
    return hipGetLastError();
}

}  // namespace shacira
