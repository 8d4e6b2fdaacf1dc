// hashgrid_tiled.hip -- cell-sorted ("tiled") hash-grid forward for large batches (gfx950).
//
// Replaces, for batches large enough to amortise one counting sort, the level loop of hashgrid_interpolate_cuda
// (wisp/csrc/ops/hashgrid_interpolate.cpp:44-66, kernel hashgrid_interpolate_cuda.cu:47-109; 2-D:
// hashgrid_interpolate2d_cuda.cu:44-99).
//
// Why: a wave-level gather costs ~20 clk + 2 clk per distinct 128-byte line it pulls from L2, but only
// ~35 clk when its lines sit in the CU's L1 (profiles/r01_microbench2_lds_gather.txt). On uniformly random samples every
// x-pair of corners is its own line on every level. Samples that sit in the same spatial BLOCK (a box of the unit cube
// holding a few hundred samples) share their table lines on every level whose cells are not much smaller than the
// block ("coarse" levels: 0-7 of S1), so walking the samples block by block turns those levels' gathers into L1 hits.
// Levels finer than that (one sample per cell: no reuse possible in any order) keep the level-per-XCD pair kernel.
//
//   sort      counting sort of the samples by block id -> 16-byte records {coordinates (exact copies), sample index}
//   fine      hashgrid_fwd_level_pair_kernel over the sorted coordinates, levels [lc, L) -> staging [L][N][F]
//   rows      hashgrid_fwd_rows_kernel: coarse levels [0, lc) over the sorted coordinates (lane pairs, L1-resident
//             lines), then whole feature rows (coarse from the wave's LDS + fine from the staging) are scattered back
//             through perm as full contiguous rows
//
// Tried and dropped in round 2 (profiles/r02_tiled_experiments.md, git 9ad8299): loading each block's sub-volume of a
// coarse level into LDS and interpolating from there, forward and backward (LDS accumulation + float-atomic flush). Per
// level it cost as much as the kernels it replaced: the unit kernels were VALU-bound on the per-cell index arithmetic
// of the sub-volume loops and on the barriers of the level chain, the backward's flush on memory-side float atomics
// (104 MB per call), and the existing compact-item pipeline handles the coarse levels of the backward at 1/3 of that.
//
// Arithmetic is the reference's (hashgrid_device.h): the result is bit-identical to every other variant, in any order.
#include <cmath>
#include <mutex>

#include "internal.h"

namespace shacira {

constexpr int kTargetPerBlock = 352;  // mean samples per block
constexpr int kSortTileS = 4096;      // samples per workgroup in the sort passes
constexpr int kSortThreadsS = 1024;
constexpr int kMaxBlocksS = 4096;     // blocks (LDS histogram of the sort passes: 16 KiB)
constexpr int kMaxAxisBlocks = 64;    // blocks per axis
constexpr uint32_t kCtxMagic = 0x53484354u;   // "SHCT"

struct TilePlan {
    int32_t nb[3];          // blocks per axis (x, y, z); block id = qx + nb[0] * (qy + nb[1] * qz)
    uint32_t num_blocks;
    uint32_t num_tiles;     // workgroups of the sort passes
    int32_t lc;             // coarse levels: [0, lc)
};

struct TileCtx {            // device pointers into the sort's scratch / outputs
    uint32_t *header;       // [0] magic, [1] num_blocks, [2] n
    float4 *sorted4;        // [n] sample records in sorted order: {x, y, z (0 in 2-D) -- exact copies --, bits of the sample index}
    uint32_t *block_start;  // [num_blocks + 1]
    uint32_t *cnt;          // [num_blocks][num_tiles] sort scratch
    uint32_t *totals;       // [num_blocks]
};

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// ----------------------------------------------------------------------------------------------- block geometry
// block index of a coordinate along one axis with nb blocks: floor((c + 1) * nb / 2), clamped; NaN -> last block (the
// clamp of the reference sends NaN to the last cell). Only locality depends on it, never the result.
__host__ __device__ inline uint32_t axis_block(float c, int nb) {
    const float u = (c + 1.0f) * (0.5f * (float)nb);
    if (!(u < (float)nb)) return (uint32_t)(nb - 1);
    if (!(u >= 0.0f)) return 0u;
    return (uint32_t)(int)u;
}

template <int DIM> __device__ __forceinline__ uint32_t block_key(const float (&c)[DIM], const TilePlan &tp) {
    uint32_t k = axis_block(c[0], tp.nb[0]) + (uint32_t)tp.nb[0] * axis_block(c[1], tp.nb[1]);
    if constexpr (DIM == 3) k += (uint32_t)(tp.nb[0] * tp.nb[1]) * axis_block(c[2], tp.nb[2]);
    return k;
}

static void make_tile_plan(int dim, const LevelTable &lt, int64_t n, TilePlan &tp) {
    // blocks: ~n / kTargetPerBlock boxes; twice as many along the slowest axis as along x and y (3-D) resp. along y as
    // along x (2-D), so that x -- the direction table lines run in -- stays long
    const double want = (double)n / kTargetPerBlock;
    tp.nb[0] = tp.nb[1] = tp.nb[2] = 1;
    auto clampi = [](double v) { int q = (int)(v + 0.5); return q < 1 ? 1 : (q > kMaxAxisBlocks ? kMaxAxisBlocks : q); };
    if (dim == 3) {
        int m = clampi(std::cbrt(want / 2.0));
        if (m > 12) m = 12;   // 12 * 12 * 28 < kMaxBlocksS
        tp.nb[0] = tp.nb[1] = m;
        int z = clampi(want / ((double)m * m));
        while ((int64_t)m * m * z > kMaxBlocksS) --z;
        tp.nb[2] = z;
    } else {
        const int m = clampi(std::sqrt(want / 2.0));
        tp.nb[0] = m;
        int y = clampi(want / m);
        while ((int64_t)m * y > kMaxBlocksS) --y;
        tp.nb[1] = y;
    }
    tp.num_blocks = (uint32_t)(tp.nb[0] * tp.nb[1] * tp.nb[2]);
    tp.num_tiles = (uint32_t)((n + kSortTileS - 1) / kSortTileS);
    // coarse levels (rows kernel) vs fine levels (level-per-XCD kernel), measured (tools/tiled_sweep.py): 3-D -- the levels
    // whose cell count does not exceed ~4x the batch keep enough reuse inside a block (S1: 8 or 9 of 16 are equally good,
    // 12 costs +30 %); 2-D -- every level (lines are shared along x at any resolution: 0.140 ms against 0.157 ms with one
    // fine level and 0.235 ms for the unsorted kernels on 2^20 samples)
    int lc = 0;
    while (lc < lt.num_lods) {
        double cells = 1.0;
        for (int a = 0; a < dim; ++a) cells *= (double)lt.res[lc];
        if (dim == 3 && cells > 4.0 * (double)n) break;
        ++lc;
    }
    const int lc_opt = opt().tiled_lc_fwd;
    if (lc_opt >= 0 && lc_opt <= lt.num_lods) lc = lc_opt;
    tp.lc = lc;
}

static TileCtx carve_ctx(int dim, int64_t n, void *buf, size_t *bytes) {
    TilePlan tp;
    LevelTable none{};
    none.num_lods = 0;
    none.feature_dim = 2;
    make_tile_plan(dim, none, n, tp);
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = up256(off + b); return o; };
    const size_t o_hdr = take(256);
    const size_t o_sorted = take((size_t)n * sizeof(float4));
    const size_t o_bs = take((size_t)(kMaxBlocksS + 1) * 4);
    const size_t o_cnt = take((size_t)tp.num_blocks * tp.num_tiles * 4);
    const size_t o_tot = take((size_t)kMaxBlocksS * 4);
    TileCtx c{};
    unsigned char *p = static_cast<unsigned char *>(buf);
    if (p) {
        c.header = reinterpret_cast<uint32_t *>(p + o_hdr);
        c.sorted4 = reinterpret_cast<float4 *>(p + o_sorted);
        c.block_start = reinterpret_cast<uint32_t *>(p + o_bs);
        c.cnt = reinterpret_cast<uint32_t *>(p + o_cnt);
        c.totals = reinterpret_cast<uint32_t *>(p + o_tot);
    }
    if (bytes) *bytes = off;
    return c;
}

static size_t sort_bytes(int dim, int64_t n) {
    size_t b = 0;
    carve_ctx(dim, n, nullptr, &b);
    return b;
}

// ----------------------------------------------------------------------------------------------- sort
template <int DIM>
__global__ __launch_bounds__(kSortThreadsS) void ctx_count_kernel(TilePlan tp, const float *__restrict__ coords,
                                                                  int64_t N, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t s_hist[kMaxBlocksS];
    for (uint32_t k = threadIdx.x; k < tp.num_blocks; k += kSortThreadsS) s_hist[k] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * kSortTileS;
    constexpr int U = kSortTileS / kSortThreadsS;
    // every coordinate load of the thread first, unconditional from a clamped index (round 4: under `if (i < N)` each
    // iteration's loads were closed by their own s_waitcnt -- U serialised round trips per thread)
    float c[U][DIM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        const int64_t ic = i < N ? i : N - 1;
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[u][a] = coords[ic * DIM + a];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        if (i < N) atomicAdd(&s_hist[block_key<DIM>(c[u], tp)], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < tp.num_blocks; k += kSortThreadsS)
        cnt[(size_t)k * tp.num_tiles + blockIdx.x] = s_hist[k];
}

// one wave per block id: exclusive scan of cnt[key][0..num_tiles) in place, total -> totals[key]
__global__ __launch_bounds__(256) void ctx_scan_tiles_kernel(uint32_t *__restrict__ cnt, uint32_t *__restrict__ totals,
                                                             uint32_t num_tiles, uint32_t num_keys) {
    const uint32_t key = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key >= num_keys) return;
    uint32_t *row = cnt + (size_t)key * num_tiles;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_tiles; base += 256) {
        const uint32_t idx = base + lane * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (idx + k < num_tiles) ? row[idx + k] : 0u;
        const uint32_t sum = (v[0] + v[1]) + (v[2] + v[3]);
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nbr = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += nbr;
        }
        uint32_t run = carry + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (idx + k < num_tiles) row[idx + k] = run;
            run += v[k];
        }
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) totals[key] = carry;
}

// one workgroup: block offsets (exclusive scan of the totals)
__global__ __launch_bounds__(1024) void ctx_scan_blocks_kernel(const uint32_t *__restrict__ totals,
                                                               uint32_t *__restrict__ block_start,
                                                               uint32_t *__restrict__ header, uint32_t num_blocks,
                                                               uint32_t n) {
    constexpr int PER = kMaxBlocksS / 1024;
    __shared__ uint32_t s_wc[16];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t c[PER], cs = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t b = t * PER + k;
        c[k] = (b < num_blocks) ? totals[b] : 0u;
        cs += c[k];
    }
    uint32_t ci = cs;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t nc = __shfl_up(ci, off, 64);
        if (lane >= (uint32_t)off) ci += nc;
    }
    if (lane == 63) s_wc[wave] = ci;
    __syncthreads();
    uint32_t pc = ci - cs;
    for (uint32_t w = 0; w < wave; ++w) pc += s_wc[w];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t b = t * PER + k;
        if (b < num_blocks) block_start[b] = pc;
        pc += c[k];
    }
    if (t == 1023) {
        block_start[num_blocks] = pc;
        header[0] = kCtxMagic;
        header[1] = num_blocks;
        header[2] = n;
    }
}

template <int DIM>
__global__ __launch_bounds__(kSortThreadsS) void ctx_scatter_kernel(TilePlan tp, const float *__restrict__ coords,
                                                                    int64_t N, const uint32_t *__restrict__ tile_off,
                                                                    const uint32_t *__restrict__ block_start,
                                                                    float4 *__restrict__ sorted4) {
    __shared__ uint32_t s_hist[kMaxBlocksS];
    for (uint32_t k = threadIdx.x; k < tp.num_blocks; k += kSortThreadsS) s_hist[k] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * kSortTileS;
    constexpr int U = kSortTileS / kSortThreadsS;
    // three batches of memory operations instead of U dependent chains (see ctx_count_kernel): coordinates, then the two
    // offsets of every sample, then the records
    float c[U][DIM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        const int64_t ic = i < N ? i : N - 1;
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[u][a] = coords[ic * DIM + a];
    }
    uint32_t key[U], rank[U], b0[U], t0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        key[u] = block_key<DIM>(c[u], tp);
        rank[u] = (i < N) ? atomicAdd(&s_hist[key[u]], 1u) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        b0[u] = block_start[key[u]];
        t0[u] = tile_off[(size_t)key[u] * tp.num_tiles + blockIdx.x];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        if (i < N)
            sorted4[b0[u] + t0[u] + rank[u]] =
                make_float4(c[u][0], c[u][1], DIM == 3 ? c[u][DIM - 1] : 0.0f, __uint_as_float((uint32_t)i));
    }
}

#define SHACIRA_CHECK_LAUNCH()                 \
    do {                                       \
        hipError_t e_ = hipGetLastError();     \
        if (e_ != hipSuccess) return e_;       \
    } while (0)

static hipError_t sort_samples(int dim, const TilePlan &tp, const float *coords, int64_t n, const TileCtx &c,
                               hipStream_t s) {
    if (dim == 3)
        hipLaunchKernelGGL(ctx_count_kernel<3>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt);
    else
        hipLaunchKernelGGL(ctx_count_kernel<2>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt);
    SHACIRA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctx_scan_tiles_kernel, dim3((tp.num_blocks + 3) / 4), dim3(256), 0, s, c.cnt, c.totals,
                       tp.num_tiles, tp.num_blocks);
    SHACIRA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctx_scan_blocks_kernel, dim3(1), dim3(1024), 0, s, c.totals, c.block_start, c.header,
                       tp.num_blocks, (uint32_t)n);
    SHACIRA_CHECK_LAUNCH();
    if (dim == 3)
        hipLaunchKernelGGL(ctx_scatter_kernel<3>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt,
                           c.block_start, c.sorted4);
    else
        hipLaunchKernelGGL(ctx_scatter_kernel<2>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt,
                           c.block_start, c.sorted4);
    SHACIRA_CHECK_LAUNCH();
    return hipSuccess;
}

// ----------------------------------------------------------------------------------------------- host side
bool tiled_supported(int dim, int dtype, const LevelTable &lt, int64_t n) {
    const int t_opt = opt().tiled;
    if (t_opt == 0) return false;
    // explicit algorithm selectors win: forward variant 8 forces this path, any other explicit variant excludes it
    const int v = opt().fwd_variant;
    if (v >= 0 && v != 8) return false;
    if (lt.feature_dim != 2 && lt.feature_dim != 4) return false;
    if (n < 1 || n >= ((int64_t)1 << 31)) return false;
    if (t_opt == 1 || v == 8) return true;
    // measured rule (tools/tiled_check.py, tools/lego_fwd_check.py): tables that do not fit an XCD's L2 (the Kodak tables of
    // configs B / C are L1 / LDS resident: sorting only costs there); 3-D F = 2 batches from 2^18 samples (equal there, -6 %
    // at 320 K, -20 % at 2^20), 3-D F = 4 (nerf_lego.yaml's 24-level table: 16-byte rows) from 80 K (-15 % at 96 K, -35 %
    // at 400 K), 2-D from 192 K (-15 % at 2^18, -40 % at 2^19)
    const size_t table_bytes = (size_t)lt.table_rows * lt.feature_dim * (dtype == SHACIRA_F32 ? 4 : 2);
    if (table_bytes < ((size_t)8 << 20)) return false;
    if (dim == 2) return n >= ((int64_t)3 << 16);
    return n >= (lt.feature_dim == 4 ? (int64_t)80 << 10 : (int64_t)1 << 18);
}

static size_t staged_bytes(int dtype, const LevelTable &lt, int64_t n) {
    return up256((size_t)n * lt.num_lods * lt.feature_dim * (dtype == SHACIRA_F32 ? 4 : 2));
}

size_t tiled_forward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n) {
    return staged_bytes(dtype, lt, n) + sort_bytes(dim, n);
}

hipError_t tiled_forward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                         const void *table, void *feats, void *workspace, int64_t n, hipStream_t s) {
    TilePlan tp;
    make_tile_plan(dim, lt, n, tp);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    void *staged = ws;
    const TileCtx ctx = carve_ctx(dim, n, ws + staged_bytes(dtype, lt, n), nullptr);
    hipError_t e = sort_samples(dim, tp, coords, n, ctx, s);
    if (e != hipSuccess) return e;
    const int L = lt.num_lods;
    if (tp.lc < L) {   // fine levels: level-per-XCD pair kernel over the sorted coordinates -> staging [L][N][F]
        LevelTable fine = lt;
        fine.level_begin = tp.lc;
        fine.level_end = L;
        e = hashgrid_forward_levels_staged(dim, dtype, fine, first_idx, reinterpret_cast<const float *>(ctx.sorted4), table,
                                           staged, n, s);
        if (e != hipSuccess) return e;
    }
    return hashgrid_forward_rows(dim, dtype, lt, first_idx, reinterpret_cast<const float *>(ctx.sorted4), table, staged,
                                 feats, n, tp.lc, s);
}

}  // namespace shacira
