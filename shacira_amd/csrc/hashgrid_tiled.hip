// hashgrid_tiled.hip -- cell-sorted ("tiled") hash-grid forward / backward for large batches (gfx950).
//
// Replaces, for batches large enough to amortise one counting sort, the level loops of
// hashgrid_interpolate_cuda / hashgrid_interpolate_backward_cuda (wisp/csrc/ops/hashgrid_interpolate.cpp:44-100,
// kernels hashgrid_interpolate_cuda.cu:47-109, :143-221; 2-D: hashgrid_interpolate2d_cuda.cu:44-99, :133-208).
//
// Why: on uniformly random samples every x-pair of corners is its own 128-byte line, so the
// forward is bound by L2 line requests (4 per sample and level) and the backward by the item stream it needs to avoid
// scattered atomics. Samples that sit in the same spatial BLOCK (a box of the unit cube holding ~1024 samples) share
// their table rows on every level whose cells are not much smaller than the block: the rows a block can touch on such
// a "coarse" level form a small sub-volume (<= 6 K cells on S1's levels 0-7) that one workgroup loads into LDS once
// (forward: ~700 line requests instead of 4096) or accumulates into in LDS and flushes once (backward: no items at
// all). Levels finer than that ("fine": one sample per cell, no reuse possible) keep the level-per-XCD pair kernel
// (forward) and the bin pipeline (backward), walking the samples in the sorted order.
//
//   context   counting sort of the samples by block id -> perm, sorted coordinates, block offsets, work units
//             (unit = <= 1024 consecutive sorted samples of ONE block); built by the forward, reusable by the backward
//             of the same coordinates (same batch, same step)
//   forward   fine levels: hashgrid_fwd_level_pair_kernel over sorted coords -> staging [L][N][F]
//             tiled_fwd_kernel: coarse levels from LDS sub-volumes (results in registers), then whole feature rows
//             (coarse from registers + fine from the staging) are scattered back through perm
//   backward  tiled_bwd_kernel: gathers whole gradient rows through perm; coarse levels accumulate in LDS sub-volumes
//             of 64-bit fixed-point / fp64 sums (float-atomic flush); the fine levels' gradients leave level-major
//             (gT) with their max |g| for the bin pipeline of hashgrid_bwd_bin.hip, which runs on the sorted coordinates
//
// Arithmetic is the reference's (hashgrid_device.h): the forward stays bit-identical, the backward differs by
// summation order only. A sample whose cell falls outside its block's precomputed sub-volume (cannot happen for finite
// coordinates; NaNs are keyed to the last block like the clamp sends them to the last cell) takes a per-sample global
// path, so correctness never depends on the bounds being tight.
#include <cmath>
#include <functional>
#include <mutex>
#include <utility>

#include "fixed_point.h"
#include "internal.h"

namespace shacira {

constexpr int kUnit = 512;            // threads per workgroup of the unit kernels = most samples of one work unit
constexpr int kTargetPerBlock = 352;  // mean samples per block: ~8 sigma of a uniform batch below kUnit -> one unit per block
constexpr int kZBatch = 6;            // sub-volume rows a thread keeps in flight
constexpr int kSortTileS = 4096;      // samples per workgroup in the sort passes
constexpr int kSortThreadsS = 1024;
constexpr int kMaxBlocksS = 4096;     // blocks (LDS histogram of the sort passes: 16 KiB)
constexpr int kMaxAxisBlocks = 64;    // blocks per axis
constexpr uint32_t kCtxMagic = 0x53484354u;   // "SHCT"
constexpr size_t kFwdRegionBytes = 40 * 1024; // LDS sub-volume budget, forward (rows of the table's scalar type)
constexpr size_t kBwdRegionBytes = 40 * 1024; // backward (64-bit sums, replicated for the small sub-volumes)

struct TilePlan {
    int32_t nb[3];          // blocks per axis (x, y, z); block id = qx + nb[0] * (qy + nb[1] * qz)
    uint32_t num_blocks;
    uint32_t num_tiles;     // workgroups of the sort passes
    uint32_t max_units;
    int32_t lc_fwd, lc_bwd; // coarse levels: [0, lc)
    uint32_t cells_fwd, cells_bwd;   // largest sub-volume (cells) over the coarse levels and all blocks
    uint32_t dbg;           // timing-only ablation mask (option "tiled_dbg"; 0 in production)
};

struct TileCtx {            // device pointers into the context buffer
    uint32_t *header;       // [0] magic, [1] num_units, [2] num_blocks, [3] n
    uint32_t *perm;         // [n]     sorted position -> sample
    float *sorted;          // [n*dim] coordinates in sorted order (exact copies)
    uint32_t *block_start;  // [num_blocks + 1]
    uint32_t *unit_block;   // [max_units]
    uint32_t *unit_off;     // [max_units] first sample of the unit, relative to its block
    uint32_t *unit_q;       // [max_units] block coordinates qx | qy << 8 | qz << 16
    int2 *ranges;           // [L][3][kMaxAxisBlocks] {first base cell, extent} of a block's sub-volume per level and axis
    uint32_t *cnt;          // [num_blocks][num_tiles] sort scratch
    uint32_t *totals;       // [num_blocks]
};

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// ----------------------------------------------------------------------------------------------- block geometry
// Same arithmetic as axis_transform (hashgrid_device.h), position only; host + device so the planner sees what the
// kernels see. Monotonic non-decreasing in c.
__host__ __device__ inline int32_t axis_cell(float c, int32_t res, float hi) {
    float x = (float)((double)res * ((double)c * 0.5 + 0.5));
    x = fmaxf(0.0f, fminf(hi, x));
    return (int32_t)floorf(x);
}

// block index of a coordinate along one axis with nb blocks: floor((c + 1) * nb / 2), clamped; NaN -> last block (the
// clamp of the reference sends NaN to the last cell). Monotonic in c.
__host__ __device__ inline uint32_t axis_block(float c, int nb) {
    const float u = (c + 1.0f) * (0.5f * (float)nb);
    if (!(u < (float)nb)) return (uint32_t)(nb - 1);
    if (!(u >= 0.0f)) return 0u;
    return (uint32_t)(int)u;
}

// cells [p_lo, p_hi] a sample of block q can have as its BASE cell on a level (block bounds widened by 4e-6, far more
// than the roundings inside axis_block); the sub-volume also holds the +1 corners: extent = p_hi - p_lo + 2
__host__ __device__ inline void block_axis_range(int q, int nb, int32_t res, float hi, int32_t &p_lo, int32_t &p_hi) {
    const float w = 2.0f / (float)nb;
    p_lo = (q == 0) ? 0 : axis_cell(-1.0f + (float)q * w - 4e-6f, res, hi);
    p_hi = (q == nb - 1) ? (int32_t)floorf(fmaxf(0.0f, hi)) : axis_cell(-1.0f + (float)(q + 1) * w + 4e-6f, res, hi);
    if (p_hi < p_lo) p_hi = p_lo;
}

template <int DIM> __device__ __forceinline__ uint32_t block_key(const float (&c)[DIM], const TilePlan &tp) {
    uint32_t k = axis_block(c[0], tp.nb[0]) + (uint32_t)tp.nb[0] * axis_block(c[1], tp.nb[1]);
    if constexpr (DIM == 3) k += (uint32_t)(tp.nb[0] * tp.nb[1]) * axis_block(c[2], tp.nb[2]);
    return k;
}

static void make_tile_plan(int dim, int dtype, const LevelTable &lt, int64_t n, TilePlan &tp) {
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
    tp.max_units = (uint32_t)(n / kUnit) + tp.num_blocks + 1;
    const size_t row_fwd = (size_t)lt.feature_dim * (dtype == SHACIRA_F32 ? 4 : 2);
    const size_t row_bwd = (size_t)lt.feature_dim * 8;
    tp.lc_fwd = tp.lc_bwd = 0;
    tp.cells_fwd = tp.cells_bwd = 1;
    bool fwd_open = true, bwd_open = true;
    for (int l = 0; l < lt.num_lods && (fwd_open || bwd_open); ++l) {
        uint64_t cells = 1;
        for (int a = 0; a < dim; ++a) {
            int32_t ext = 0;
            for (int q = 0; q < tp.nb[a]; ++q) {
                int32_t lo, hi;
                block_axis_range(q, tp.nb[a], lt.res[l], lt.hi[l], lo, hi);
                if (hi - lo + 2 > ext) ext = hi - lo + 2;
            }
            cells *= (uint64_t)ext;
        }
        if (fwd_open && cells * row_fwd <= kFwdRegionBytes && cells < 65536) {
            tp.lc_fwd = l + 1;
            if (cells > tp.cells_fwd) tp.cells_fwd = (uint32_t)cells;
        } else {
            fwd_open = false;
        }
        if (bwd_open && cells * row_bwd <= kBwdRegionBytes && cells < 65536) {
            tp.lc_bwd = l + 1;
            if (cells > tp.cells_bwd) tp.cells_bwd = (uint32_t)cells;
        } else {
            bwd_open = false;
        }
    }
    const int maxc = (dim == 3) ? 8 : 16;   // MaxCoarse<DIM>: per-level results / gradients live in registers
    if (tp.lc_fwd > maxc) tp.lc_fwd = maxc;
    if (tp.lc_bwd > maxc) tp.lc_bwd = maxc;
    // options: cap the coarse prefix (A/B, and "0" = every level through the fine path)
    tp.dbg = (uint32_t)g_tiled_dbg.load();
    const int of = g_tiled_lc_fwd.load(), ob = g_tiled_lc_bwd.load();
    if (of >= 0 && of < tp.lc_fwd) tp.lc_fwd = of;
    if (ob >= 0 && ob < tp.lc_bwd) tp.lc_bwd = ob;
}

static TileCtx carve_ctx(int dim, int64_t n, void *buf, size_t *bytes) {
    TilePlan tp;
    LevelTable none{};
    none.num_lods = 0;
    none.feature_dim = 2;
    make_tile_plan(dim, SHACIRA_F32, none, n, tp);
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = up256(off + b); return o; };
    const size_t o_hdr = take(256);
    const size_t o_perm = take((size_t)n * 4);
    const size_t o_sorted = take((size_t)n * dim * 4);
    const size_t o_bs = take((size_t)(kMaxBlocksS + 1) * 4);
    const size_t o_ub = take((size_t)tp.max_units * 4);
    const size_t o_uo = take((size_t)tp.max_units * 4);
    const size_t o_uq = take((size_t)tp.max_units * 4);
    const size_t o_rg = take((size_t)SHACIRA_MAX_LODS * 3 * kMaxAxisBlocks * sizeof(int2));
    const size_t o_cnt = take((size_t)tp.num_blocks * tp.num_tiles * 4);
    const size_t o_tot = take((size_t)kMaxBlocksS * 4);
    TileCtx c{};
    unsigned char *p = static_cast<unsigned char *>(buf);
    if (p) {
        c.header = reinterpret_cast<uint32_t *>(p + o_hdr);
        c.perm = reinterpret_cast<uint32_t *>(p + o_perm);
        c.sorted = reinterpret_cast<float *>(p + o_sorted);
        c.block_start = reinterpret_cast<uint32_t *>(p + o_bs);
        c.unit_block = reinterpret_cast<uint32_t *>(p + o_ub);
        c.unit_off = reinterpret_cast<uint32_t *>(p + o_uo);
        c.unit_q = reinterpret_cast<uint32_t *>(p + o_uq);
        c.ranges = reinterpret_cast<int2 *>(p + o_rg);
        c.cnt = reinterpret_cast<uint32_t *>(p + o_cnt);
        c.totals = reinterpret_cast<uint32_t *>(p + o_tot);
    }
    if (bytes) *bytes = off;
    return c;
}

size_t tiled_context_bytes(int dim, int64_t n) {
    size_t b = 0;
    carve_ctx(dim, n, nullptr, &b);
    return b;
}

// ----------------------------------------------------------------------------------------------- context (sort)
template <int DIM>
__global__ __launch_bounds__(kSortThreadsS) void ctx_count_kernel(TilePlan tp, const float *__restrict__ coords,
                                                                  int64_t N, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t s_hist[kMaxBlocksS];
    for (uint32_t k = threadIdx.x; k < tp.num_blocks; k += kSortThreadsS) s_hist[k] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * kSortTileS;
#pragma unroll
    for (int u = 0; u < kSortTileS / kSortThreadsS; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        if (i < N) {
            float c[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) c[a] = coords[i * DIM + a];
            atomicAdd(&s_hist[block_key<DIM>(c, tp)], 1u);
        }
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

// one workgroup: block offsets (exclusive scan of the totals) and the work-unit list
__global__ __launch_bounds__(1024) void ctx_scan_blocks_kernel(TilePlan tp, const uint32_t *__restrict__ totals,
                                                               uint32_t *__restrict__ block_start,
                                                               uint32_t *__restrict__ unit_block,
                                                               uint32_t *__restrict__ unit_off,
                                                               uint32_t *__restrict__ unit_q,
                                                               uint32_t *__restrict__ header, uint32_t n) {
    constexpr int PER = kMaxBlocksS / 1024;
    __shared__ uint32_t s_wc[16], s_wu[16];
    const uint32_t num_blocks = tp.num_blocks;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t c[PER], u[PER], cs = 0, us = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t b = t * PER + k;
        c[k] = (b < num_blocks) ? totals[b] : 0u;
        u[k] = (c[k] + kUnit - 1) / kUnit;
        cs += c[k];
        us += u[k];
    }
    uint32_t ci = cs, ui = us;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t nc = __shfl_up(ci, off, 64), nu = __shfl_up(ui, off, 64);
        if (lane >= (uint32_t)off) { ci += nc; ui += nu; }
    }
    if (lane == 63) { s_wc[wave] = ci; s_wu[wave] = ui; }
    __syncthreads();
    uint32_t pc = ci - cs, pu = ui - us;
    for (uint32_t w = 0; w < wave; ++w) { pc += s_wc[w]; pu += s_wu[w]; }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t b = t * PER + k;
        if (b < num_blocks) {
            block_start[b] = pc;
            const uint32_t qx = b % (uint32_t)tp.nb[0], rest = b / (uint32_t)tp.nb[0];
            const uint32_t qy = rest % (uint32_t)tp.nb[1], qz = rest / (uint32_t)tp.nb[1];
            for (uint32_t q = 0; q < u[k]; ++q) {
                unit_block[pu + q] = b;
                unit_off[pu + q] = q * kUnit;
                unit_q[pu + q] = qx | (qy << 8) | (qz << 16);
            }
        }
        pc += c[k];
        pu += u[k];
    }
    if (t == 1023) {
        block_start[num_blocks] = pc;
        header[0] = kCtxMagic;
        header[1] = pu;
        header[2] = num_blocks;
        header[3] = n;
    }
}

// sub-volume of every (level, axis, block index): {first base cell, extent}
__global__ __launch_bounds__(kMaxAxisBlocks) void ctx_ranges_kernel(LevelTable lt, TilePlan tp, int dim,
                                                                    int2 *__restrict__ ranges) {
    const int l = blockIdx.x / 3, a = blockIdx.x % 3, q = threadIdx.x;
    int2 r = make_int2(0, 1);
    if (a < dim && q < tp.nb[a]) {
        int32_t lo, hi;
        block_axis_range(q, tp.nb[a], lt.res[l], lt.hi[l], lo, hi);
        r = make_int2(lo, hi - lo + 2);
    }
    ranges[(size_t)blockIdx.x * kMaxAxisBlocks + q] = r;
}

template <int DIM>
__global__ __launch_bounds__(kSortThreadsS) void ctx_scatter_kernel(TilePlan tp, const float *__restrict__ coords,
                                                                    int64_t N, const uint32_t *__restrict__ tile_off,
                                                                    const uint32_t *__restrict__ block_start,
                                                                    uint32_t *__restrict__ perm,
                                                                    float *__restrict__ sorted) {
    __shared__ uint32_t s_hist[kMaxBlocksS];
    for (uint32_t k = threadIdx.x; k < tp.num_blocks; k += kSortThreadsS) s_hist[k] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * kSortTileS;
#pragma unroll
    for (int u = 0; u < kSortTileS / kSortThreadsS; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        if (i < N) {
            float c[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) c[a] = coords[i * DIM + a];
            const uint32_t key = block_key<DIM>(c, tp);
            const uint32_t rank = atomicAdd(&s_hist[key], 1u);
            const uint32_t pos = block_start[key] + tile_off[(size_t)key * tp.num_tiles + blockIdx.x] + rank;
            perm[pos] = (uint32_t)i;
#pragma unroll
            for (int a = 0; a < DIM; ++a) sorted[(size_t)pos * DIM + a] = c[a];
        }
    }
}

#define SHACIRA_CHECK_LAUNCH()                 \
    do {                                       \
        hipError_t e_ = hipGetLastError();     \
        if (e_ != hipSuccess) return e_;       \
    } while (0)

static hipError_t build_context(int dim, const LevelTable &lt, const TilePlan &tp, const float *coords, int64_t n,
                                const TileCtx &c, hipStream_t s) {
    if (dim == 3)
        hipLaunchKernelGGL(ctx_count_kernel<3>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt);
    else
        hipLaunchKernelGGL(ctx_count_kernel<2>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt);
    SHACIRA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctx_ranges_kernel, dim3(3 * lt.num_lods), dim3(kMaxAxisBlocks), 0, s, lt, tp, dim, c.ranges);
    SHACIRA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctx_scan_tiles_kernel, dim3((tp.num_blocks + 3) / 4), dim3(256), 0, s, c.cnt, c.totals,
                       tp.num_tiles, tp.num_blocks);
    SHACIRA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctx_scan_blocks_kernel, dim3(1), dim3(1024), 0, s, tp, c.totals, c.block_start, c.unit_block,
                       c.unit_off, c.unit_q, c.header, (uint32_t)n);
    SHACIRA_CHECK_LAUNCH();
    if (dim == 3)
        hipLaunchKernelGGL(ctx_scatter_kernel<3>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt,
                           c.block_start, c.perm, c.sorted);
    else
        hipLaunchKernelGGL(ctx_scatter_kernel<2>, dim3(tp.num_tiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt,
                           c.block_start, c.perm, c.sorted);
    SHACIRA_CHECK_LAUNCH();
    return hipSuccess;
}

// ----------------------------------------------------------------------------------------------- sub-volume helpers
// The unit kernels are VALU-bound (every wave instruction costs 4 issue cycles, ~130 of them per sample and level are
// the reference arithmetic itself), so the loops over a sub-volume's cells must not decode a linear cell index per cell:
// a thread owns one in-plane cell (lx, ly) -- decoded once per level -- and walks z with a stride.
struct Region {
    int32_t lo[3];
    uint32_t ext[3];      // cells per axis (base cells + the +1 corners)
    uint32_t exy, cells;
};

// All coarse levels' ranges of one block are fetched together at kernel start into LDS (one global latency instead of
// one per level; registers would not hold them: the level table already fills the scalar file).
template <int DIM, int MAXC>
__device__ __forceinline__ void load_block_ranges(const int2 *__restrict__ ranges, uint32_t uq, int lc, int2 *s_rg) {
    if (threadIdx.x < (unsigned)(MAXC * 3)) {
        const int l = threadIdx.x / 3, a = threadIdx.x - 3 * l;
        int2 v = make_int2(0, 1);
        if (l < lc && a < DIM) v = ranges[(size_t)(l * 3 + a) * kMaxAxisBlocks + ((uq >> (8 * a)) & 0xFFu)];
        s_rg[threadIdx.x] = v;
    }
}

template <int DIM>
__device__ __forceinline__ Region region_of(const int2 *s_rg, int l) {
    Region r;
    r.lo[2] = 0;
    r.ext[2] = 1;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        const int2 v = s_rg[l * 3 + a];
        r.lo[a] = v.x;
        r.ext[a] = (uint32_t)v.y;
    }
    r.exy = r.ext[0] * r.ext[1];
    r.cells = r.exy * r.ext[2];
    return r;
}

// n / d for n < 2^16, d < 2^16: float estimate from below (never above the true quotient, at most 1 below) + fix-up
__device__ __forceinline__ uint32_t small_div(uint32_t n, uint32_t d, uint32_t &rem) {
    const float rcp = __uint_as_float(__float_as_uint(1.0f / (float)d) - 2u);
    uint32_t q = (uint32_t)((float)n * rcp);
    rem = n - q * d;
    if (rem >= d) { rem -= d; ++q; }
    return q;
}

// How the threads of a unit cover a sub-volume whose plane holds `plane` items (cells, or cells * F words):
//   plane <= kUnit: G = kUnit / plane planes side by side; thread -> (z slot g, in-plane item q), one pass
//   plane >  kUnit: one plane at a time, the thread strides through it
struct PlaneWalk {
    uint32_t q0, qstep, g, G;
};
__device__ __forceinline__ PlaneWalk plane_walk(uint32_t plane) {
    PlaneWalk w;
    if (plane <= (uint32_t)kUnit) {
        uint32_t unused;
        w.G = small_div((uint32_t)kUnit, plane, unused);
        w.g = small_div(threadIdx.x, plane, w.q0);
        if (w.g >= w.G) w.q0 = plane;     // idle thread: no in-plane item
        w.qstep = plane;                  // single pass
    } else {
        w.G = 1;
        w.g = 0;
        w.q0 = threadIdx.x;
        w.qstep = kUnit;
    }
    return w;
}

// level-local row of a cell, exactly the reference's rule (.cu:27-36, 2d.cu:26-33), uint32 wraparound included
template <int DIM>
__device__ __forceinline__ uint32_t cell_row(uint32_t x, uint32_t y, uint32_t z, uint32_t r, bool dense, uint32_t mask) {
    if (dense) {
        uint32_t row = x + y * r;
        if constexpr (DIM == 3) row += z * r * r;
        return row;
    }
    uint32_t row = x ^ (y * kPrimeY);
    if constexpr (DIM == 3) row ^= z * kPrimeZ;
    return row & mask;
}

template <typename T, int F> struct RowOf { T v[F]; };

#ifdef SHACIRA_TILED_STAMPS   // diagnostic build only (tools/): per-phase cycle sums of wave 0 of every workgroup
__device__ unsigned long long g_tiled_stamps[32];
#define STAMP_DECL unsigned long long st_t = clock64();
#define STAMP(k)                                                                          \
    do {                                                                                  \
        const unsigned long long st_n = clock64();                                        \
        if (threadIdx.x == 0) atomicAdd(&g_tiled_stamps[k], st_n - st_t);                 \
        st_t = st_n;                                                                      \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#endif

template <int... Is, typename Fn>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, Fn &&fn) {
    (fn(std::integral_constant<int, Is>{}), ...);
}

template <int DIM> struct MaxCoarse { static constexpr int value = (DIM == 3) ? 8 : 16; };

struct RowStage {          // epilogue / prologue staging of whole rows, computed on the host
    uint32_t rows;         // rows per round (power of two, multiple of 64, <= kUnit)
    uint32_t pitch_bytes;  // row pitch in LDS: row bytes rounded up to 16, + 16
    uint32_t row_bytes;    // L * F * sizeof(scalar)
    uint32_t wide;         // 1: rows move in 16-byte chunks, 0: one F-piece per lane
};

static RowStage make_row_stage(int L, int F, size_t scalar_bytes, size_t budget) {
    RowStage rs;
    rs.row_bytes = (uint32_t)((size_t)L * F * scalar_bytes);
    rs.pitch_bytes = (rs.row_bytes + 15u) / 16u * 16u + 16u;
    rs.wide = (rs.row_bytes % 16u == 0) ? 1u : 0u;
    rs.rows = kUnit;
    while (rs.rows > 64 && (size_t)rs.rows * rs.pitch_bytes > budget) rs.rows >>= 1;
    return rs;
}

// per-sample global gather, the reference's formula verbatim: only for samples whose cell lies outside their block's
// sub-volume (never for finite coordinates). noinline: keeps the hot loop's LDS reads from being merged with it.
template <int DIM, typename T, int F>
__device__ __forceinline__ void slow_corners(const int32_t (&p)[DIM], const float (&f)[DIM], const float (&g)[DIM],
                                          uint32_t r, bool dense, uint32_t mask, int64_t base, int64_t table_rows,
                                          const RowOf<T, F> *__restrict__ rows, float (&acc)[F]) {
    constexpr int NC = 1 << DIM;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int dx = (DIM == 3) ? ((k >> 2) & 1) : ((k >> 1) & 1);
        const int dy = (DIM == 3) ? ((k >> 1) & 1) : (k & 1);
        const int dz = (DIM == 3) ? (k & 1) : 0;
        float w = (dx ? f[0] : g[0]) * (dy ? f[1] : g[1]);
        if constexpr (DIM == 3) w = w * (dz ? f[2] : g[2]);
        const uint32_t row = cell_row<DIM>((uint32_t)p[0] + dx, (uint32_t)p[1] + dy,
                                           (DIM == 3) ? (uint32_t)p[DIM - 1] + dz : 0u, r, dense, mask);
        const int64_t grow = base + (int64_t)row;
        RowOf<T, F> v;
#pragma unroll
        for (int j = 0; j < F; ++j) Scalar<T>::store(&v.v[j], 0.0f);
        if ((uint64_t)grow < (uint64_t)table_rows) v = rows[grow];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const float tv = Scalar<T>::load(&v.v[j]);
            acc[j] = (k == 0) ? tv * w : fmaf(tv, w, acc[j]);
        }
    }
}

// ----------------------------------------------------------------------------------------------- forward
// One workgroup = one work unit (<= 1024 sorted samples of one block), thread = sample.
//   coarse levels [0, lc): sub-volume of the level into LDS, 2^DIM LDS reads per sample, result kept in registers
//   epilogue: whole feature rows [L*F] are assembled in LDS (coarse part from registers, fine part from the level-major
//   staging buffer the pair kernel wrote) and stored to feats[perm[i]] as full contiguous rows (16-byte chunks).
template <int DIM, typename T, int F>
__global__ __launch_bounds__(kUnit) void tiled_fwd_kernel(LevelTable lt, TilePlan tp, RowStage rs,
                                                          const int32_t *__restrict__ first_idx, TileCtx ctx,
                                                          const T *__restrict__ table, const T *__restrict__ staged,
                                                          T *__restrict__ feats, int64_t N) {
    constexpr int NC = 1 << DIM;
    constexpr int MAXC = MaxCoarse<DIM>::value;
    using Row = RowOf<T, F>;
    extern __shared__ __align__(16) unsigned char s_raw_f[];
    Row *s_reg = reinterpret_cast<Row *>(s_raw_f);
    const uint32_t unit = blockIdx.x;
    if (unit >= ctx.header[1]) return;
    const uint32_t blk = ctx.unit_block[unit];
    const uint32_t uq = ctx.unit_q[unit];
    const uint32_t begin = ctx.block_start[blk] + ctx.unit_off[unit];
    uint32_t end = ctx.block_start[blk + 1];
    if (end > begin + kUnit) end = begin + kUnit;
    const uint32_t i = begin + threadIdx.x;
    const bool live = i < end;
    double t[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) t[a] = axis_unit(live ? ctx.sorted[(size_t)i * DIM + a] : 0.0f);
    const Row *rows = reinterpret_cast<const Row *>(table);
    const Row *fine = reinterpret_cast<const Row *>(staged);
    const int L = lt.num_lods;
    Row out[MAXC];
    __shared__ int2 s_rg[MAXC * 3];
    load_block_ranges<DIM, MAXC>(ctx.ranges, uq, tp.lc_fwd, s_rg);
    // the fine levels' pieces of this thread's own sample (written by the pair kernel): fetched now, used by the epilogue
    constexpr int kFinePre = 8;
    Row fpre[kFinePre];
#pragma unroll
    for (int k = 0; k < kFinePre; ++k) {
#pragma unroll
        for (int j = 0; j < F; ++j) Scalar<T>::store(&fpre[k].v[j], 0.0f);
        if (live && tp.lc_fwd + k < L && !(tp.dbg & 4u)) fpre[k] = fine[(int64_t)(tp.lc_fwd + k) * N + i];
    }
    const uint32_t my_perm = live ? ctx.perm[i] : 0u;
    __syncthreads();   // s_rg
    STAMP_DECL

    // The sub-volume of level l+1 is fetched into registers while level l is evaluated from LDS (software pipeline):
    // issue(l) starts the loads of this thread's first kZBatch cells, commit(l) stores them to LDS and fetches what a
    // larger sub-volume has beyond them.
    Row pre[kZBatch];
    uint32_t pre_q = 0;
    auto row_of = [&](const Region &rg, uint32_t xy, uint32_t lz, uint32_t r, bool dense) -> uint32_t {
        uint32_t row = xy;
        if constexpr (DIM == 3) {
            const uint32_t cz = (uint32_t)rg.lo[2] + lz;
            row = dense ? xy + cz * r * r : (xy ^ (cz * kPrimeZ));
        }
        return dense ? row : (row & lt.mask);
    };
    auto xy_of = [&](const Region &rg, uint32_t q, uint32_t r, bool dense) -> uint32_t {
        uint32_t lx;
        const uint32_t ly = small_div(q, rg.ext[0], lx);
        const uint32_t cx = (uint32_t)rg.lo[0] + lx, cy = (uint32_t)rg.lo[1] + ly;
        return dense ? cx + cy * r : (cx ^ (cy * kPrimeY));
    };
    auto issue = [&](auto lc_) __attribute__((always_inline)) {
        constexpr int l = decltype(lc_)::value;
        const Region rg = region_of<DIM>(s_rg, l);
        const uint32_t r = (uint32_t)lt.res[l];
        const bool dense = lt.dense[l] != 0;
        const int64_t base = (int64_t)first_idx[l];
        const PlaneWalk pw = plane_walk(rg.exy);
        pre_q = (tp.dbg & 1u) ? rg.exy : pw.q0;
#pragma unroll
        for (int u = 0; u < kZBatch; ++u)
#pragma unroll
            for (int j = 0; j < F; ++j) Scalar<T>::store(&pre[u].v[j], 0.0f);
        if (pre_q < rg.exy) {
            const uint32_t xy = xy_of(rg, pre_q, r, dense);
#pragma unroll
            for (int u = 0; u < kZBatch; ++u) {
                const uint32_t lz = pw.g + u * pw.G;
                if (lz < rg.ext[2]) {
                    const int64_t grow = base + (int64_t)row_of(rg, xy, lz, r, dense);
                    if ((uint64_t)grow < (uint64_t)lt.table_rows) pre[u] = rows[grow];
                }
            }
        }
    };
    auto commit = [&](auto lc_) __attribute__((always_inline)) {
        constexpr int l = decltype(lc_)::value;
        const Region rg = region_of<DIM>(s_rg, l);
        const uint32_t r = (uint32_t)lt.res[l];
        const bool dense = lt.dense[l] != 0;
        const int64_t base = (int64_t)first_idx[l];
        const PlaneWalk pw = plane_walk(rg.exy);
        if (pre_q < rg.exy) {
#pragma unroll
            for (int u = 0; u < kZBatch; ++u) {
                const uint32_t lz = pw.g + u * pw.G;
                if (lz < rg.ext[2]) s_reg[lz * rg.exy + pre_q] = pre[u];
            }
        }
        // what lies beyond the prefetched batch (sub-volumes of more than kUnit * kZBatch cells): fetched here
        for (uint32_t q = pre_q; q < rg.exy; q += pw.qstep) {
            const uint32_t xy = xy_of(rg, q, r, dense);
            for (uint32_t z0 = pw.g + ((q == pre_q) ? pw.G * kZBatch : 0u); z0 < rg.ext[2]; z0 += pw.G * kZBatch) {
                Row v[kZBatch];
#pragma unroll
                for (int u = 0; u < kZBatch; ++u) {
                    const uint32_t lz = z0 + u * pw.G;
#pragma unroll
                    for (int j = 0; j < F; ++j) Scalar<T>::store(&v[u].v[j], 0.0f);
                    if (lz < rg.ext[2]) {
                        const int64_t grow = base + (int64_t)row_of(rg, xy, lz, r, dense);
                        if ((uint64_t)grow < (uint64_t)lt.table_rows) v[u] = rows[grow];
                    }
                }
#pragma unroll
                for (int u = 0; u < kZBatch; ++u) {
                    const uint32_t lz = z0 + u * pw.G;
                    if (lz < rg.ext[2]) s_reg[lz * rg.exy + q] = v[u];
                }
            }
        }
    };
    if (tp.lc_fwd > 0) issue(std::integral_constant<int, 0>{});
    static_for(std::make_integer_sequence<int, MAXC>{}, [&](auto lc_) __attribute__((always_inline)) {
        constexpr int l = decltype(lc_)::value;
        if (l >= tp.lc_fwd) return;   // uniform
        const int32_t res = lt.res[l];
        const float hi = lt.hi[l];
        const bool dense = lt.dense[l] != 0;
        const uint32_t r = (uint32_t)res;
        const int64_t base = (int64_t)first_idx[l];
        const Region rg = region_of<DIM>(s_rg, l);
        commit(lc_);
        __syncthreads();
        STAMP(1);
        if constexpr (l + 1 < MAXC) {
            if (l + 1 < tp.lc_fwd) issue(std::integral_constant<int, l + 1>{});
        }
        if (live && !(tp.dbg & 2u)) {
            int32_t p[DIM];
            float f[DIM], g[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f[a], g[a]);
            bool inside = true;
#pragma unroll
            for (int a = 0; a < DIM; ++a)
                inside = inside && (uint32_t)(p[a] - rg.lo[a]) + 1u < rg.ext[a];
            const uint32_t sy = rg.ext[0], sz = rg.exy;
            uint32_t local = (uint32_t)(p[0] - rg.lo[0]) + (uint32_t)(p[1] - rg.lo[1]) * sy;
            if constexpr (DIM == 3) local += (uint32_t)(p[2] - rg.lo[2]) * sz;
            float acc[F];
            // reference corner order: 3-D k = dx*4 + dy*2 + dz, 2-D k = dx*2 + dy; weight (wx * wy) * wz
            if (inside) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int dx = (DIM == 3) ? ((k >> 2) & 1) : ((k >> 1) & 1);
                    const int dy = (DIM == 3) ? ((k >> 1) & 1) : (k & 1);
                    const int dz = (DIM == 3) ? (k & 1) : 0;
                    float w = (dx ? f[0] : g[0]) * (dy ? f[1] : g[1]);
                    if constexpr (DIM == 3) w = w * (dz ? f[2] : g[2]);
                    const Row v = s_reg[local + dx + dy * sy + dz * sz];
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        const float tv = Scalar<T>::load(&v.v[j]);
                        acc[j] = (k == 0) ? tv * w : fmaf(tv, w, acc[j]);
                    }
                }
            } else {
                slow_corners<DIM, T, F>(p, f, g, r, dense, lt.mask, base, lt.table_rows, rows, acc);
            }
#pragma unroll
            for (int j = 0; j < F; ++j) Scalar<T>::store(&out[l].v[j], acc[j]);
        }
        __syncthreads();   // the sub-volume is overwritten by the next level / the epilogue
        STAMP(2);
    });
    // epilogue: rows of rs.rows samples at a time through LDS (the sub-volume's memory)
    const uint32_t count = end - begin;
    const uint32_t nf = (uint32_t)(L - tp.lc_fwd);
    for (uint32_t r0 = 0; r0 < count; r0 += rs.rows) {
        const uint32_t nr = (count - r0 < rs.rows) ? (count - r0) : rs.rows;
        if (live && threadIdx.x >= r0 && threadIdx.x < r0 + nr) {
            Row *dst = reinterpret_cast<Row *>(s_raw_f + (size_t)(threadIdx.x - r0) * rs.pitch_bytes);
            static_for(std::make_integer_sequence<int, MAXC>{}, [&](auto lc_) __attribute__((always_inline)) {
                constexpr int l = decltype(lc_)::value;
                if (l < tp.lc_fwd) dst[l] = out[l];
            });
#pragma unroll
            for (int k = 0; k < kFinePre; ++k)
                if (tp.lc_fwd + k < L) dst[tp.lc_fwd + k] = fpre[k];
            // the row's destination travels with it (last 16 bytes of the pitch are padding)
            *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(dst) + rs.pitch_bytes - 16u) = my_perm;
        }
        if (nf > (uint32_t)kFinePre) {   // more fine levels than the prefetch holds: the rest through the staging now
            const uint32_t nrest = nf - kFinePre, l0 = (uint32_t)tp.lc_fwd + kFinePre;
            for (uint32_t e = threadIdx.x; e < rs.rows * nrest; e += kUnit) {
                const uint32_t rr = e & (rs.rows - 1), lf = e / rs.rows;
                if (rr < nr)
                    reinterpret_cast<Row *>(s_raw_f + (size_t)rr * rs.pitch_bytes)[l0 + lf] =
                        fine[(int64_t)(l0 + lf) * N + begin + r0 + rr];
            }
        }
        __syncthreads();
        STAMP(3);
        if (tp.dbg & 8u) {
        } else if (rs.wide) {
            const uint32_t cpr = rs.row_bytes / 16u;
            for (uint32_t e = threadIdx.x; e < nr * cpr; e += kUnit) {
                const uint32_t rr = e / cpr, q = e - rr * cpr;
                const unsigned char *src = s_raw_f + (size_t)rr * rs.pitch_bytes;
                const uint4 v = *reinterpret_cast<const uint4 *>(src + q * 16u);
                const uint32_t pr = *reinterpret_cast<const uint32_t *>(src + rs.pitch_bytes - 16u);
                unsigned char *dst = reinterpret_cast<unsigned char *>(feats) + (size_t)pr * rs.row_bytes + q * 16u;
                *reinterpret_cast<uint4 *>(dst) = v;
            }
        } else {
            for (uint32_t e = threadIdx.x; e < nr * (uint32_t)L; e += kUnit) {
                const uint32_t rr = e / (uint32_t)L, l = e - rr * (uint32_t)L;
                const unsigned char *src = s_raw_f + (size_t)rr * rs.pitch_bytes;
                const uint32_t pr = *reinterpret_cast<const uint32_t *>(src + rs.pitch_bytes - 16u);
                reinterpret_cast<Row *>(feats)[(size_t)pr * L + l] = reinterpret_cast<const Row *>(src)[l];
            }
        }
        __syncthreads();
        STAMP(4);
    }
}

// ----------------------------------------------------------------------------------------------- backward
// One workgroup = one work unit, thread = sample.
//   prologue: the unit's gradient rows grad_output[perm[i]] (whole rows, 16-byte chunks) pass through LDS; every thread
//   keeps the coarse levels of its own sample in registers, the fine levels are written level-major to gT (what the bin
//   pipeline reads) and the per-level max |g| is gathered on the way (workgroup-local for the coarse levels, global for
//   the fine ones);
//   coarse levels [0, lc): zero the sub-volume of 64-bit sums, every sample adds its 2^DIM corners with LDS atomics (fixed
//   point scaled by the WORKGROUP's max |g| of the level -- each workgroup rounds its own sums to fp32 before the flush;
//   fp64 if that max is inf / NaN), then the non-zero sums are added to the (zeroed) table with float atomics, lanes
//   consecutive in (cell, feature) so that a request covers neighbouring rows.
template <int DIM, int F>
__global__ __launch_bounds__(kUnit) void tiled_bwd_kernel(LevelTable lt, TilePlan tp, RowStage rs,
                                                          const int32_t *__restrict__ first_idx, TileCtx ctx,
                                                          const float *__restrict__ go, float *__restrict__ gT,
                                                          float *__restrict__ grad_table, int64_t N,
                                                          uint32_t *__restrict__ gmax, int headroom,
                                                          uint32_t lds_bytes) {
    constexpr int NC = 1 << DIM;
    constexpr int MAXC = MaxCoarse<DIM>::value;
    extern __shared__ double s_acc_t[];
    unsigned char *s_raw = reinterpret_cast<unsigned char *>(s_acc_t);
    unsigned long long *s_fix = reinterpret_cast<unsigned long long *>(s_acc_t);
    __shared__ uint32_t s_max[SHACIRA_MAX_LODS];
    const uint32_t unit = blockIdx.x;
    if (unit >= ctx.header[1]) return;
    const uint32_t blk = ctx.unit_block[unit];
    const uint32_t uq = ctx.unit_q[unit];
    const uint32_t begin = ctx.block_start[blk] + ctx.unit_off[unit];
    uint32_t end = ctx.block_start[blk + 1];
    if (end > begin + kUnit) end = begin + kUnit;
    const uint32_t i = begin + threadIdx.x;
    const bool live = i < end;
    const int L = lt.num_lods;
    const uint32_t count = end - begin;
    const uint32_t lane = threadIdx.x & 63;
    if (threadIdx.x < SHACIRA_MAX_LODS) s_max[threadIdx.x] = 0;
    double t[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) t[a] = axis_unit(live ? ctx.sorted[(size_t)i * DIM + a] : 0.0f);
    __shared__ int2 s_rg[MAXC * 3];
    load_block_ranges<DIM, MAXC>(ctx.ranges, uq, tp.lc_bwd, s_rg);
    float gc[MAXC][F];
#pragma unroll
    for (int l = 0; l < MAXC; ++l)
#pragma unroll
        for (int j = 0; j < F; ++j) gc[l][j] = 0.0f;
    const uint32_t nf = (uint32_t)(L - tp.lc_bwd);
    STAMP_DECL
    // whole gradient rows, rs.rows samples per round; the loads of round k+1 are in flight while round k is distributed
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int UB = 5;   // rs.rows * chunks-per-row <= 5 * kUnit for every row size (make_row_stage's budget)
    const uint32_t cpr = rs.row_bytes / 16u;
    u32x4 pv[UB];
    auto issue_rows = [&](uint32_t r0) __attribute__((always_inline)) {
        const uint32_t nr = (count - r0 < rs.rows) ? (count - r0) : rs.rows;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const uint32_t e = u * kUnit + threadIdx.x;
            const uint32_t rr = e / cpr, q = e - rr * cpr;
            pv[u] = u32x4{0u, 0u, 0u, 0u};
            if (e < nr * cpr && !(tp.dbg & 16u))
                pv[u] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(go) +
                                                         (size_t)ctx.perm[begin + r0 + rr] * rs.row_bytes + q * 16u);
        }
    };
    if (rs.wide && count > 0) issue_rows(0);
    for (uint32_t r0 = 0; r0 < count; r0 += rs.rows) {
        const uint32_t nr = (count - r0 < rs.rows) ? (count - r0) : rs.rows;
        __syncthreads();   // previous round's readers are done (and s_max is zeroed)
        if (rs.wide) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const uint32_t e = u * kUnit + threadIdx.x;
                const uint32_t rr = e / cpr, q = e - rr * cpr;
                if (e < nr * cpr) *reinterpret_cast<u32x4 *>(s_raw + (size_t)rr * rs.pitch_bytes + q * 16u) = pv[u];
            }
        } else {
            const uint32_t epr = (uint32_t)(L * F);
            for (uint32_t e = threadIdx.x; e < nr * epr; e += kUnit) {
                const uint32_t rr = e / epr, q = e - rr * epr;
                reinterpret_cast<float *>(s_raw + (size_t)rr * rs.pitch_bytes)[q] =
                    go[(size_t)ctx.perm[begin + r0 + rr] * epr + q];
            }
        }
        __syncthreads();
        STAMP(8);
        if (rs.wide && r0 + rs.rows < count) issue_rows(r0 + rs.rows);
        if (live && threadIdx.x >= r0 && threadIdx.x < r0 + nr) {
            const float *src = reinterpret_cast<const float *>(s_raw + (size_t)(threadIdx.x - r0) * rs.pitch_bytes);
            static_for(std::make_integer_sequence<int, MAXC>{}, [&](auto lc_) __attribute__((always_inline)) {
                constexpr int l = decltype(lc_)::value;
                if (l < tp.lc_bwd) {
#pragma unroll
                    for (int j = 0; j < F; ++j) gc[l][j] = src[l * F + j];
                }
            });
        }
        // fine levels: level-major, coalesced along the samples; a wave stays inside one level (rs.rows % 64 == 0)
        for (uint32_t e = threadIdx.x; e < rs.rows * nf; e += kUnit) {
            const uint32_t rr = e & (rs.rows - 1), lf = e / rs.rows;
            const uint32_t l = (uint32_t)tp.lc_bwd + lf;
            uint32_t m = 0;
            if (rr < nr) {
                const float *src = reinterpret_cast<const float *>(s_raw + (size_t)rr * rs.pitch_bytes) + l * F;
                float v[F];
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    v[j] = src[j];
                    const uint32_t b = __float_as_uint(fabsf(v[j]));
                    m = b > m ? b : m;
                }
                float *dst = gT + ((int64_t)l * N + begin + r0 + rr) * F;
                if constexpr (F == 2) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 o = {v[0], v[1]};
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x2 *>(dst));
                } else {
#pragma unroll
                    for (int j = 0; j < F; ++j) __builtin_nontemporal_store(v[j], dst + j);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(m, off, 64);
                m = o > m ? o : m;
            }
            if (lane == 0 && m) atomicMax(&s_max[l], m);
        }
        STAMP(9);
    }
    // workgroup-local max |g| of the coarse levels
    static_for(std::make_integer_sequence<int, MAXC>{}, [&](auto lc_) __attribute__((always_inline)) {
        constexpr int l = decltype(lc_)::value;
        if (l >= tp.lc_bwd) return;
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const uint32_t b = __float_as_uint(fabsf(gc[l][j]));
            m = b > m ? b : m;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(m, off, 64);
            m = o > m ? o : m;
        }
        if (lane == 0 && m) atomicMax(&s_max[l], m);
    });
    __syncthreads();
    if ((int)threadIdx.x >= tp.lc_bwd && (int)threadIdx.x < L && s_max[threadIdx.x])
        atomicMax(&gmax[threadIdx.x], s_max[threadIdx.x]);   // the consume pass scales the fine levels by the global max
    STAMP(10);

    static_for(std::make_integer_sequence<int, MAXC>{}, [&](auto lc_) __attribute__((always_inline)) {
        constexpr int l = decltype(lc_)::value;
        if (l >= tp.lc_bwd) return;   // uniform
        const int32_t res = lt.res[l];
        const float hi = lt.hi[l];
        const bool dense = lt.dense[l] != 0;
        const uint32_t r = (uint32_t)res;
        const int64_t base = (int64_t)first_idx[l];
        const Region rg = region_of<DIM>(s_rg, l);
        // Small sub-volumes (the coarsest levels: ~100 cells for ~700 samples) are REPLICATED: lane k of a wave adds into
        // copy k % R, so lanes of one LDS atomic instruction that hit the same cell do not serialise on one address.
        // R = largest power of two <= 64 that fits the LDS budget.
        const uint32_t words = rg.cells * F;
        uint32_t R = 1;
        while (R < 64u && (size_t)words * (2u * R) * sizeof(double) <= lds_bytes) R <<= 1;
        for (uint32_t e = threadIdx.x; e < words * R; e += kUnit) s_acc_t[e] = 0.0;   // all-zero bits either way
        const FxScale fx = fx_scale_of(s_max[l], headroom);
        __syncthreads();
        STAMP(11);
        if (live && !(tp.dbg & 32u)) {
            int32_t p[DIM];
            float f[DIM], g[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f[a], g[a]);
            bool inside = true;
#pragma unroll
            for (int a = 0; a < DIM; ++a)
                inside = inside && (uint32_t)(p[a] - rg.lo[a]) + 1u < rg.ext[a];
            const uint32_t sy = rg.ext[0], sz = rg.exy;
            uint32_t local = (uint32_t)(p[0] - rg.lo[0]) + (uint32_t)(p[1] - rg.lo[1]) * sy;
            if constexpr (DIM == 3) local += (uint32_t)(p[2] - rg.lo[2]) * sz;
            const uint32_t copy = (lane & (R - 1u)) * words;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int dx = (DIM == 3) ? ((k >> 2) & 1) : ((k >> 1) & 1);
                const int dy = (DIM == 3) ? ((k >> 1) & 1) : (k & 1);
                const int dz = (DIM == 3) ? (k & 1) : 0;
                float w = (dx ? f[0] : g[0]) * (dy ? f[1] : g[1]);
                if constexpr (DIM == 3) w = w * (dz ? f[2] : g[2]);
                if (inside) {
                    const uint32_t slot = copy + (local + dx + dy * sy + dz * sz) * F;
                    if (fx.fixed) {
#pragma unroll
                        for (int j = 0; j < F; ++j) atomicAdd(s_fix + slot + j, fx_encode(gc[l][j] * w, fx.scale));
                    } else {
#pragma unroll
                        for (int j = 0; j < F; ++j) atomicAdd(s_acc_t + slot + j, (double)(gc[l][j] * w));
                    }
                } else {
                    // per-sample global path (never taken for finite coordinates)
                    const uint32_t cx = (uint32_t)p[0] + dx, cy = (uint32_t)p[1] + dy;
                    const uint32_t cz = (DIM == 3) ? (uint32_t)p[DIM - 1] + dz : 0u;
                    if (dense && (cx >= r || cy >= r || (DIM == 3 && cz >= r))) continue;
                    const int64_t grow = base + (int64_t)cell_row<DIM>(cx, cy, cz, r, dense, lt.mask);
                    if ((uint64_t)grow < (uint64_t)lt.table_rows) {
#pragma unroll
                        for (int j = 0; j < F; ++j) unsafeAtomicAdd(grad_table + grow * F + j, gc[l][j] * w);
                    }
                }
            }
        }
        __syncthreads();
        STAMP(12);
        // flush: a thread owns one in-plane (cell, feature) word and walks z; lanes consecutive in (x, feature)
        const uint32_t pwords = rg.exy * F;
        const PlaneWalk pw = plane_walk(pwords);
        for (uint32_t qq = (tp.dbg & 64u) ? pwords : pw.q0; qq < pwords; qq += pw.qstep) {
            const uint32_t j = qq & (F - 1), q = qq / F;
            uint32_t lx;
            const uint32_t ly = small_div(q, rg.ext[0], lx);
            const uint32_t cx = (uint32_t)rg.lo[0] + lx, cy = (uint32_t)rg.lo[1] + ly;
            // corners with a coordinate == res lie outside a dense level (weight 0 in the reference): dropped
            if (dense && (cx >= r || cy >= r)) continue;
            const uint32_t xy = dense ? cx + cy * r : (cx ^ (cy * kPrimeY));
            for (uint32_t lz = pw.g; lz < rg.ext[2]; lz += pw.G) {
                const uint32_t e = lz * pwords + qq;
                float v;
                if (fx.fixed) {
                    unsigned long long sum = s_fix[e];
                    for (uint32_t c = 1; c < R; ++c) sum += s_fix[e + c * words];
                    v = fx_decode(sum, fx.inv);
                } else {
                    double sum = s_acc_t[e];
                    for (uint32_t c = 1; c < R; ++c) sum += s_acc_t[e + c * words];
                    v = (float)sum;
                }
                if (v == 0.0f) continue;
                uint32_t row = xy;
                if constexpr (DIM == 3) {
                    const uint32_t cz = (uint32_t)rg.lo[2] + lz;
                    if (dense && cz >= r) continue;
                    row = dense ? xy + cz * r * r : (xy ^ (cz * kPrimeZ));
                }
                if (!dense) row &= lt.mask;
                const int64_t grow = base + (int64_t)row;
                if ((uint64_t)grow >= (uint64_t)lt.table_rows) continue;
                unsafeAtomicAdd(grad_table + grow * F + j, v);
            }
        }
        __syncthreads();
        STAMP(13);
    });
}

// ----------------------------------------------------------------------------------------------- host side
bool tiled_supported(int dim, int dtype, const LevelTable &lt, int64_t n, bool backward) {
    const int opt = g_tiled.load();
    if (opt == 0) return false;
    // explicit algorithm selectors win: forward variant 8 / backward variant 2 force this path, others exclude it
    const int v = backward ? g_bwd_variant.load() : g_fwd_variant.load();
    const int mine = backward ? 2 : 8;
    if (v >= 0 && v != mine) return false;
    if (backward && dtype != SHACIRA_F32) return false;   // fp16 tables: the bin pipeline keeps its fp32 image
    if (lt.feature_dim != 2 && lt.feature_dim != 4) return false;
    if (n < 1 || n >= ((int64_t)1 << 31)) return false;
    if (!bin_supported(dim, lt)) return false;
    if (opt == 1 || v == mine) return true;
    // measured rule: the sort pays for itself once blocks hold a few hundred samples each at >= 64 blocks
    return n >= ((int64_t)1 << 18) && dim == 3;
}

static size_t staged_bytes(int dtype, const LevelTable &lt, int64_t n) {
    return up256((size_t)n * lt.num_lods * lt.feature_dim * (dtype == SHACIRA_F32 ? 4 : 2));
}

size_t tiled_forward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n) {
    return staged_bytes(dtype, lt, n) + tiled_context_bytes(dim, n);
}

// the level table the bin pipeline sees: fine levels only, their gradients already staged by tiled_bwd_kernel
static LevelTable fine_levels(const LevelTable &lt, const TilePlan &tp) {
    LevelTable fine = lt;
    fine.level_begin = tp.lc_bwd < lt.num_lods ? tp.lc_bwd : lt.num_lods;
    fine.level_end = lt.num_lods;
    fine.stage_flags = SHACIRA_BWD_REUSE_STAGED;
    return fine;
}

size_t tiled_backward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n) {
    TilePlan tp;
    make_tile_plan(dim, dtype, lt, n, tp);
    return up256(bin_workspace_bytes(dim, dtype, fine_levels(lt, tp), n)) + tiled_context_bytes(dim, n);
}

static constexpr size_t kRowStageBudget = 20 * 1024;       // forward epilogue (4 workgroups per CU)
static constexpr size_t kRowStageBudgetBwd = 40 * 1024;    // backward prologue: the accumulators' memory

static hipError_t opt_in_lds() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        auto set = [](const void *fn, size_t bytes) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) err = e;
        };
#define SHACIRA_TILED_ATTR(D, FF)                                                                   \
        set(reinterpret_cast<const void *>(&tiled_fwd_kernel<D, float, FF>), kFwdRegionBytes);      \
        set(reinterpret_cast<const void *>(&tiled_fwd_kernel<D, __half, FF>), kFwdRegionBytes);     \
        set(reinterpret_cast<const void *>(&tiled_bwd_kernel<D, FF>), kBwdRegionBytes);
        SHACIRA_TILED_ATTR(2, 2) SHACIRA_TILED_ATTR(2, 4) SHACIRA_TILED_ATTR(3, 2) SHACIRA_TILED_ATTR(3, 4)
#undef SHACIRA_TILED_ATTR
    });
    return err;
}

template <int DIM, typename T, int F>
static hipError_t launch_fwd_tiles(const LevelTable &lt, const TilePlan &tp, const int32_t *first_idx,
                                   const TileCtx &ctx, const void *table, const void *staged, void *feats, int64_t n,
                                   hipStream_t s) {
    const RowStage rs = make_row_stage(lt.num_lods, F, sizeof(T), kRowStageBudget);
    size_t shmem = (size_t)tp.cells_fwd * F * sizeof(T);
    const size_t stage = (size_t)rs.rows * rs.pitch_bytes;
    if (stage > shmem) shmem = stage;
    hipLaunchKernelGGL((tiled_fwd_kernel<DIM, T, F>), dim3(tp.max_units), dim3(kUnit), shmem, s, lt, tp, rs, first_idx,
                       ctx, static_cast<const T *>(table), static_cast<const T *>(staged), static_cast<T *>(feats), n);
    return hipGetLastError();
}

hipError_t tiled_forward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                         const void *table, void *feats, void *workspace, int64_t n, void *context, int ctx_flags,
                         hipStream_t s) {
    hipError_t e = opt_in_lds();
    if (e != hipSuccess) return e;
    TilePlan tp;
    make_tile_plan(dim, dtype, lt, n, tp);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    void *staged = ws;
    void *ctx_buf = context ? context : ws + staged_bytes(dtype, lt, n);
    const TileCtx ctx = carve_ctx(dim, n, ctx_buf, nullptr);
    (void)ctx_flags;
    if ((e = build_context(dim, lt, tp, coords, n, ctx, s)) != hipSuccess) return e;
    const int L = lt.num_lods, F = lt.feature_dim;
    if (tp.lc_fwd < L) {   // fine levels: level-per-XCD pair kernel over the sorted coordinates -> staging [L][N][F]
        LevelTable fine = lt;
        fine.level_begin = tp.lc_fwd;
        fine.level_end = L;
        e = hashgrid_forward_levels_staged(dim, dtype, fine, first_idx, ctx.sorted, table, staged, n, s);
        if (e != hipSuccess) return e;
    }
    if (g_tiled_rows.load() != 0)
        return hashgrid_forward_rows(dim, dtype, lt, first_idx, ctx.sorted, ctx.perm, table, staged, feats, n,
                                     tp.lc_fwd, s);
    if (dim == 3 && dtype == SHACIRA_F32)
        return F == 2 ? launch_fwd_tiles<3, float, 2>(lt, tp, first_idx, ctx, table, staged, feats, n, s)
                      : launch_fwd_tiles<3, float, 4>(lt, tp, first_idx, ctx, table, staged, feats, n, s);
    if (dim == 3)
        return F == 2 ? launch_fwd_tiles<3, __half, 2>(lt, tp, first_idx, ctx, table, staged, feats, n, s)
                      : launch_fwd_tiles<3, __half, 4>(lt, tp, first_idx, ctx, table, staged, feats, n, s);
    if (dtype == SHACIRA_F32)
        return F == 2 ? launch_fwd_tiles<2, float, 2>(lt, tp, first_idx, ctx, table, staged, feats, n, s)
                      : launch_fwd_tiles<2, float, 4>(lt, tp, first_idx, ctx, table, staged, feats, n, s);
    return F == 2 ? launch_fwd_tiles<2, __half, 2>(lt, tp, first_idx, ctx, table, staged, feats, n, s)
                  : launch_fwd_tiles<2, __half, 4>(lt, tp, first_idx, ctx, table, staged, feats, n, s);
}

template <int DIM, int F>
static hipError_t launch_bwd_tiles(const LevelTable &lt, const TilePlan &tp, const int32_t *first_idx,
                                   const TileCtx &ctx, const float *go, float *gT, float *acc, int64_t n,
                                   uint32_t *gmax, hipStream_t s) {
    const RowStage rs = make_row_stage(lt.num_lods, F, sizeof(float), kRowStageBudgetBwd);
    (void)tp.cells_bwd;
    const size_t shmem = kBwdRegionBytes;   // whole budget: small sub-volumes are replicated to spread the LDS atomics
    const int headroom = fx_headroom((uint64_t)kUnit << DIM);   // a cell receives <= kUnit * 2^DIM contributions
    hipLaunchKernelGGL((tiled_bwd_kernel<DIM, F>), dim3(tp.max_units), dim3(kUnit), shmem, s, lt, tp, rs, first_idx,
                       ctx, go, gT, acc, n, gmax, headroom, (uint32_t)shmem);
    return hipGetLastError();
}

// `acc` is the fp32 gradient table ([table_rows, F]); fp32 tables only (tiled_supported).
hipError_t tiled_backward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                          const void *grad_out, float *acc, void *workspace, int64_t n, void *context, int ctx_flags,
                          hipStream_t s) {
    hipError_t e = opt_in_lds();
    if (e != hipSuccess) return e;
    TilePlan tp;
    make_tile_plan(dim, dtype, lt, n, tp);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    const LevelTable fine = fine_levels(lt, tp);
    void *ctx_buf = context ? context : ws + up256(bin_workspace_bytes(dim, dtype, fine, n));
    const TileCtx ctx = carve_ctx(dim, n, ctx_buf, nullptr);
    const bool reuse = context != nullptr && (ctx_flags & SHACIRA_CTX_REUSE) != 0;
    if (!reuse && (e = build_context(dim, lt, tp, coords, n, ctx, s)) != hipSuccess) return e;
    const int F = lt.feature_dim;
    e = hipMemsetAsync(acc, 0, (size_t)lt.table_rows * F * sizeof(float), s);   // at::zeros_like, .cpp:81/:167
    if (e != hipSuccess) return e;
    float *gT;
    uint32_t *gmax;
    bin_staged_pointers(dim, dtype, fine, n, workspace, &gT, &gmax);
    if ((e = hipMemsetAsync(gmax, 0, SHACIRA_MAX_LODS * sizeof(uint32_t), s)) != hipSuccess) return e;
    // the unit kernel (coarse levels + staging of the fine levels' gradients) is issued by the bin pipeline at the point
    // where it would transpose, i.e. after it has forked its count + scan passes onto the side stream
    const float *go = static_cast<const float *>(grad_out);
    const std::function<hipError_t(hipStream_t)> stage = [&](hipStream_t st) -> hipError_t {
        if (dim == 3)
            return F == 2 ? launch_bwd_tiles<3, 2>(lt, tp, first_idx, ctx, go, gT, acc, n, gmax, st)
                          : launch_bwd_tiles<3, 4>(lt, tp, first_idx, ctx, go, gT, acc, n, gmax, st);
        return F == 2 ? launch_bwd_tiles<2, 2>(lt, tp, first_idx, ctx, go, gT, acc, n, gmax, st)
                      : launch_bwd_tiles<2, 4>(lt, tp, first_idx, ctx, go, gT, acc, n, gmax, st);
    };
    return bin_backward(dim, dtype, fine, first_idx, ctx.sorted, grad_out, acc, workspace, n, s, false, ctx.perm, &stage);
}

#ifdef SHACIRA_TILED_STAMPS
hipError_t tiled_read_stamps(unsigned long long *out32, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_tiled_stamps), 32 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_tiled_stamps), z, sizeof(z));
    }
    return e;
}
#endif

}  // namespace shacira
