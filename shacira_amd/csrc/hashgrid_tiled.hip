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
//   sort      counting sort of the samples by block id (two partitioning passes, three launches: psort_*) -> 16-byte records
//             {coordinates (exact copies), sample index} + block offsets = the batch's PLAN, which the backward can reuse
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
#ifndef SHACIRA_SORT_TILE
#define SHACIRA_SORT_TILE 4096
#endif
constexpr int kSortTileS = SHACIRA_SORT_TILE;      // samples per workgroup in the sort passes
constexpr int kSortThreadsS = 1024;
constexpr int kMaxBlocksS = 4096;     // blocks (LDS histogram of the sort passes: 16 KiB)
constexpr int kMaxAxisBlocks = 64;    // blocks per axis
constexpr uint32_t kCtxMagic = 0x53484354u;   // "SHCT"

constexpr int kMaxCoarse = 256;       // coarse bins of the partition sort (first pass)
constexpr int kMaxSortTiles = 512;    // workgroups of its first two passes

struct TilePlan {
    int32_t nb[3];          // blocks per axis (x, y, z); block id = qx + nb[0] * (qy + nb[1] * qz)
    uint32_t num_blocks;
    int32_t lc;             // coarse levels: [0, lc)
    // partition sort: coarse bin = block id >> coarse_shift (a contiguous range of <= 16 blocks); tiles of whole chunks
    uint32_t coarse_shift, num_coarse;
    uint32_t ptiles, chunks_per_tile;
};

struct TileCtx {            // device pointers into the sort's outputs (the plan) and its scratch
    uint32_t *header;       // [0] magic, [1] num_blocks, [2] n
    float4 *sorted4;        // [n] sample records in sorted order: {x, y, z (0 in 2-D) -- exact copies --, bits of the sample index}
    uint32_t *block_start;  // [num_blocks + 1]
    uint32_t *cnt;          // scratch: [ptiles][kMaxCoarse] samples per (tile, coarse bin)
    float4 *inter4;         // scratch: [n] records grouped by coarse bin
    uint32_t *gcursor;      // scratch: [2][kMaxBlocksS] per-block cursors | per-block counts of over-full bins
    uint32_t *cbase;        // scratch: [kMaxCoarse + 1] first record of every coarse bin
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

// block grid + sort geometry: a function of (dim, n) alone, so that a plan built by one call can be read by another
static void make_sort_plan(int dim, int64_t n, TilePlan &tp) {
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
    tp.lc = 0;
    tp.coarse_shift = 0;
    while (((tp.num_blocks + (1u << tp.coarse_shift) - 1) >> tp.coarse_shift) > (uint32_t)kMaxCoarse) ++tp.coarse_shift;
    tp.num_coarse = (tp.num_blocks + (1u << tp.coarse_shift) - 1) >> tp.coarse_shift;
    const int64_t chunks = (n + kSortTileS - 1) / kSortTileS;
    tp.chunks_per_tile = (uint32_t)((chunks + kMaxSortTiles - 1) / kMaxSortTiles);
    if (tp.chunks_per_tile < 1) tp.chunks_per_tile = 1;
    tp.ptiles = (uint32_t)((chunks + tp.chunks_per_tile - 1) / tp.chunks_per_tile);
}

static void make_tile_plan(int dim, const LevelTable &lt, int64_t n, TilePlan &tp) {
    make_sort_plan(dim, n, tp);
    // coarse levels (rows kernel) vs fine levels (level-per-XCD kernel). 3-D: the levels whose cell count does not exceed ~32x
    // the batch. Round 6 (tools/fwd_lc_ab.py, profiles/r06_experiments.md 6): with the records ordered by sub-cell inside a block
    // the rows kernel takes a level for less than the fine kernel's staging round trip costs, until the hashed tables of too
    // many levels compete for the XCD's L2 -- best split S1 (2^20): 10 levels (0.299 -> 0.282 ms; 12: 0.323), 2^19: 9, 2^18: 8,
    // nerf_lego.yaml's F = 4 table at 409 600 samples: 18 of 24 (0.300 -> 0.265 ms), at 102 400: 15-16; the last level taken
    // has 14-25x the batch's cells, the first one left 35-64x. (Rounds 2-5, block order only: 4x the batch.)
    // 2-D: every level (lines are shared along x at any resolution).
    int lc = 0;
    while (lc < lt.num_lods) {
        double cells = 1.0;
        for (int a = 0; a < dim; ++a) cells *= (double)lt.res[lc];
        if (dim == 3 && cells > 32.0 * (double)n) break;
        ++lc;
    }
    const int lc_opt = opt().tiled_lc_fwd;
    if (lc_opt >= 0 && lc_opt <= lt.num_lods) lc = lc_opt;
    tp.lc = lc;
}

// The PLAN of a coordinate batch (what a later call may reuse: the sorted records and the block offsets) and the sort's
// SCRATCH are carved separately: a caller-owned plan buffer outlives the forward call, the scratch does not.
static size_t carve_plan(int64_t n, void *buf, TileCtx &c) {
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = up256(off + b); return o; };
    const size_t o_hdr = take(256);
    const size_t o_sorted = take((size_t)n * sizeof(float4));
    const size_t o_bs = take((size_t)(kMaxBlocksS + 1) * 4);
    unsigned char *p = static_cast<unsigned char *>(buf);
    if (p) {
        c.header = reinterpret_cast<uint32_t *>(p + o_hdr);
        c.sorted4 = reinterpret_cast<float4 *>(p + o_sorted);
        c.block_start = reinterpret_cast<uint32_t *>(p + o_bs);
    }
    return off;
}
static size_t carve_scratch(int dim, int64_t n, void *buf, TileCtx &c) {
    (void)dim;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = up256(off + b); return o; };
    const size_t o_cnt = take((size_t)kMaxSortTiles * kMaxCoarse * 4);
    const size_t o_inter = take((size_t)n * sizeof(float4));
    const size_t o_cb = take((size_t)(kMaxCoarse + 1) * 4);
    const size_t o_gc = take((size_t)2 * kMaxBlocksS * 4);   // per-block cursors | per-block counts of over-full bins
    unsigned char *p = static_cast<unsigned char *>(buf);
    if (p) {
        c.cnt = reinterpret_cast<uint32_t *>(p + o_cnt);
        c.inter4 = reinterpret_cast<float4 *>(p + o_inter);
        c.cbase = reinterpret_cast<uint32_t *>(p + o_cb);
        c.gcursor = reinterpret_cast<uint32_t *>(p + o_gc);
    }
    return off;
}
static size_t plan_bytes_of(int64_t n) {
    TileCtx c{};
    return carve_plan(n, nullptr, c);
}
static size_t scratch_bytes_of(int dim, int64_t n) {
    TileCtx c{};
    return carve_scratch(dim, n, nullptr, c);
}

// ----------------------------------------------------------------------------------------------- partition sort (round 6)
// Counting sort by block id in two partitioning passes, every write stream private to one workgroup:
//   count       per tile of whole 4 096-sample chunks: samples per COARSE bin (<= 256 contiguous ranges of <= 16 block ids)
//   partition   the same tiles: bin bases + this tile's offsets from the count matrix (column sums, no atomics, no scan
//               launch), then each record goes to its bin's next slot (LDS returning atomic = final position)
//   local       one workgroup per coarse bin keeps the bin's records in registers (<= 8 192; ~5 500 on a uniform batch): the
//               returning LDS atomic that counts a record per (block, sub-cell of the block) is its rank; offsets = scan of
//               <= 1 024 counts; the records go out run by run (a few KB each, written by one workgroup) and the bin's
//               block offsets with them
//   over-full bins (a batch concentrated in few blocks): `partition` also counts such a bin's records per block (global
//               counters; one LDS flag decides, a uniform batch skips it); in `local` workgroup `bin` takes the bin's first
//               8 192 records and one more workgroup per 8 192-aligned window of the partitioned batch takes what the bin holds
//               beyond (surplus ones leave at once); each ranks its records in LDS and reserves one range per block from
//               global cursors. Block order only in such bins. A Gaussian blob / a thin slab / one block of 2^20 samples
//               sort as fast as a uniform batch (tools/skew_check.py: forward 0.29 / 0.26 / 0.22 ms against 0.29)
// The round-5 sort it replaces (four launches) wrote every record straight to its block's slot: 2^20 scattered 16-byte stores into lines
// shared by 256 workgroups (25 us of its 47). Order inside a block is arbitrary in both (ranks come from atomics).
template <int DIM>
__device__ __forceinline__ void load_chunk_coords(const float *__restrict__ coords, int64_t s0, int64_t N,
                                                  float (&c)[kSortTileS / kSortThreadsS][DIM]) {
    constexpr int U = kSortTileS / kSortThreadsS;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
        const int64_t ic = i < N ? i : N - 1;   // unconditional loads from a clamped index (see ctx_count_kernel)
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[u][a] = coords[ic * DIM + a];
    }
}

constexpr int kLocalR = 8;                                   // records a thread of the local pass keeps
constexpr uint32_t kLocalCap = kLocalR * kSortThreadsS;      // a bin of <= 8 192 records is one chunk: read once, ranked in LDS

template <int DIM>
__global__ __launch_bounds__(kSortThreadsS) void psort_count_kernel(TilePlan tp, const float *__restrict__ coords,
                                                                    int64_t N, uint32_t *__restrict__ cnt,
                                                                    uint32_t *__restrict__ gcursor) {
    __shared__ uint32_t s_hist[kMaxCoarse];
    if (threadIdx.x < kMaxCoarse) s_hist[threadIdx.x] = 0;
    // (the per-block cursors | counts of over-full bins start at zero: cleared here, one launch ahead of their first use)
    if (blockIdx.x == 0)
        for (uint32_t k = threadIdx.x; k < 2u * kMaxBlocksS; k += kSortThreadsS) gcursor[k] = 0u;
    __syncthreads();
    constexpr int U = kSortTileS / kSortThreadsS;
    for (uint32_t ch = 0; ch < tp.chunks_per_tile; ++ch) {
        const int64_t s0 = ((int64_t)blockIdx.x * tp.chunks_per_tile + ch) * kSortTileS;
        if (s0 >= N) break;
        float c[U][DIM];
        load_chunk_coords<DIM>(coords, s0, N, c);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s0 + u * kSortThreadsS + threadIdx.x < N) atomicAdd(&s_hist[block_key<DIM>(c[u], tp) >> tp.coarse_shift], 1u);
    }
    __syncthreads();
    // bin-major: a bin's counts over the tiles are one contiguous row (the partition pass sums it with one load per lane)
    if (threadIdx.x < kMaxCoarse) cnt[(size_t)threadIdx.x * kMaxSortTiles + blockIdx.x] = s_hist[threadIdx.x];
}

template <int DIM>
__global__ __launch_bounds__(kSortThreadsS) void psort_partition_kernel(TilePlan tp, const float *__restrict__ coords,
                                                                        int64_t N, const uint32_t *__restrict__ cnt,
                                                                        uint32_t *__restrict__ cbase,
                                                                        uint32_t *__restrict__ gcursor,
                                                                        float4 *__restrict__ inter4) {
    __shared__ uint32_t s_cursor[kMaxCoarse], s_tot[kMaxCoarse], s_before[kMaxCoarse];
    __shared__ uint32_t s_wave[kMaxCoarse / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int U = kSortTileS / kSortThreadsS;
    // this tile's first chunk: in flight while the offsets below are computed
    float c0[U][DIM];
    const int64_t first_s0 = (int64_t)blockIdx.x * tp.chunks_per_tile * kSortTileS;
    load_chunk_coords<DIM>(coords, first_s0 < N ? first_s0 : 0, N, c0);
    // bin totals and this tile's offset inside every bin: a wave sums one bin's row of the count matrix (16-byte loads, four
    // tiles per lane), 16 bins per wave, eight rows in flight
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kWaves = kSortThreadsS / 64, kBinsPerWave = kMaxCoarse / kWaves;
    const bool wide = tp.ptiles > 256u;   // a second 256-tile half
#ifndef SHACIRA_SORT_RB
#define SHACIRA_SORT_RB 8
#endif
    constexpr int RB = SHACIRA_SORT_RB;
#pragma unroll 1
    for (int g = 0; g < kBinsPerWave; g += RB) {
        u32x4 v[RB], w[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const u32x4 *row = reinterpret_cast<const u32x4 *>(cnt + (size_t)(wave * kBinsPerWave + g + k) * kMaxSortTiles);
            v[k] = row[lane];
            w[k] = wide ? row[64 + lane] : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            uint32_t tot = 0, before = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t t0 = lane * 4u + (uint32_t)e, t1 = 256u + t0;
                const uint32_t a = t0 < tp.ptiles ? v[k][e] : 0u, b2 = t1 < tp.ptiles ? w[k][e] : 0u;
                tot += a + b2;
                before += (t0 < blockIdx.x ? a : 0u) + (t1 < blockIdx.x ? b2 : 0u);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                tot += __shfl_xor(tot, off, 64);
                before += __shfl_xor(before, off, 64);
            }
            if (lane == 0) {
                s_tot[wave * kBinsPerWave + g + k] = tot;
                s_before[wave * kBinsPerWave + g + k] = before;
            }
        }
    }
    __syncthreads();
    const bool owner = threadIdx.x < kMaxCoarse;   // (whole waves)
    uint32_t tot = 0;
    if (owner) {
        tot = s_tot[threadIdx.x];
        uint32_t incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nbr = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += nbr;
        }
        if (lane == 63) s_wave[wave] = incl;
        s_cursor[threadIdx.x] = incl - tot;   // exclusive inside the wave; the wave bases are added behind the barrier
    }
    __syncthreads();
    uint32_t base = 0;
    if (owner) {
        for (uint32_t wv = 0; wv < wave; ++wv) base += s_wave[wv];
        base += s_cursor[threadIdx.x];
        if (blockIdx.x == 0) {
            cbase[threadIdx.x] = base;
            if (threadIdx.x == kMaxCoarse - 1) cbase[kMaxCoarse] = base + tot;
        }
    }
    // Over-full bins (more records than the last pass keeps in one workgroup's registers: a batch concentrated in few blocks):
    // their records are also counted per BLOCK here, so that the last pass can place every 8 192-record chunk of such a bin with
    // its own workgroup straight away. A uniform batch has none and skips all of it (one LDS flag).
    __shared__ uint32_t s_over_any;
    __shared__ uint32_t s_blk[kMaxBlocksS];
    if (threadIdx.x == 0) s_over_any = 0u;
    __syncthreads();
    if (owner) {
        s_cursor[threadIdx.x] = base + s_before[threadIdx.x];
        if (tot > kLocalCap) s_over_any = 1u;
        s_tot[threadIdx.x] = tot > kLocalCap ? 1u : 0u;       // (from here on: the bin is over-full)
    }
    __syncthreads();
    const bool over_any = s_over_any != 0u;
    if (over_any) {
        for (uint32_t k = threadIdx.x; k < tp.num_blocks; k += kSortThreadsS) s_blk[k] = 0u;
        __syncthreads();
    }
    for (uint32_t ch = 0; ch < tp.chunks_per_tile; ++ch) {
        const int64_t s0 = first_s0 + (int64_t)ch * kSortTileS;
        if (s0 >= N) break;
        float c[U][DIM];
        if (ch == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int a = 0; a < DIM; ++a) c[u][a] = c0[u][a];
        } else {
            load_chunk_coords<DIM>(coords, s0, N, c);
        }
        uint32_t pos[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
            const uint32_t key = block_key<DIM>(c[u], tp);
            pos[u] = (i < N) ? atomicAdd(&s_cursor[key >> tp.coarse_shift], 1u) : 0xFFFFFFFFu;
            if (over_any && i < N && s_tot[key >> tp.coarse_shift]) atomicAdd(&s_blk[key], 1u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = s0 + u * kSortThreadsS + threadIdx.x;
            if (pos[u] != 0xFFFFFFFFu) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 rec = {c[u][0], c[u][1], DIM == 3 ? c[u][DIM - 1] : 0.0f, __uint_as_float((uint32_t)i)};
#ifdef SHACIRA_SORT_NT
                __builtin_nontemporal_store(rec, reinterpret_cast<f32x4 *>(inter4 + pos[u]));
#else
                *reinterpret_cast<f32x4 *>(inter4 + pos[u]) = rec;
#endif
            }
        }
    }
    if (over_any) {
        __syncthreads();
        uint32_t *gcount = gcursor + kMaxBlocksS;
        for (uint32_t k = threadIdx.x; k < tp.num_blocks; k += kSortThreadsS)
            if (s_blk[k]) atomicAdd(&gcount[k], s_blk[k]);
    }
}

// A wave adds `one` per lane to counters[j] and learns its lanes' ranks; when every lane names the same counter (a batch
// concentrated in one block) lane 0 adds for all of them instead of 64 serialised same-address atomics.
__device__ __forceinline__ uint32_t wave_rank_add(uint32_t *counters, uint32_t j, bool live, uint32_t lane) {
    const uint64_t lm = __ballot(live);
    if (lm == 0ull) return 0u;
    const uint32_t leader = (uint32_t)__builtin_ctzll(lm);
    const uint32_t j0 = (uint32_t)__shfl((int)j, (int)leader, 64);
    const uint64_t same = __ballot(live && j == j0);
    if (same == lm) {
        uint32_t b = 0;
        if (lane == leader) b = atomicAdd(&counters[j0], (uint32_t)__popcll(lm));
        b = (uint32_t)__shfl((int)b, (int)leader, 64);
        return b + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull));
    }
    return live ? atomicAdd(&counters[j], 1u) : 0u;
}

// Workgroups of the last pass. w < num_coarse: bin w -- all of it when it holds <= 8 192 records (every bin of a uniform batch:
// nothing but the bin's two offsets is read before its records), its first 8 192 records otherwise. What an over-full bin (a
// batch concentrated in few blocks) holds beyond is cut at the 8 192-aligned windows of the partitioned batch, and workgroup
// num_coarse + v takes window v: the bin holding the window's first record is the only one whose remainder can reach into that
// window (the next bin's remainder starts >= 8 192 records behind its own start). So such a batch gets as many workgroups as it
// has 8 192-record pieces, whatever bin they fall into (round 6: the first form gave an over-full bin four workgroups that each
// counted ALL of it -- 0.8 ms for a Gaussian blob of 2^20 samples, tools/skew_check.py), and no workgroup searches a list.
template <int DIM>
__global__ __launch_bounds__(kSortThreadsS) void psort_local_kernel(TilePlan tp, const float4 *__restrict__ inter4,
                                                                    const uint32_t *__restrict__ cbase,
                                                                    const uint32_t *__restrict__ gcount,
                                                                    uint32_t *__restrict__ gcursor,
                                                                    float4 *__restrict__ sorted4,
                                                                    uint32_t *__restrict__ block_start,
                                                                    uint32_t *__restrict__ header, uint32_t n) {
    constexpr int kBpc = kMaxBlocksS / kMaxCoarse;   // blocks per coarse bin (<= 16)
    constexpr int R = kLocalR;
    uint32_t bin, lo, hi, c_lo, c_hi;
    const bool first = blockIdx.x < tp.num_coarse;
    if (first) {
        bin = blockIdx.x;
        lo = cbase[bin];
        hi = cbase[bin + 1];
        c_lo = lo;
        c_hi = (lo + kLocalCap < hi) ? lo + kLocalCap : hi;
    } else {
        const uint32_t p0 = (blockIdx.x - tp.num_coarse) * kLocalCap;      // window's first record
        if (p0 >= n) return;
        __shared__ uint32_t s_le;
        if (threadIdx.x == 0) s_le = 0u;
        __syncthreads();
        uint32_t c = (threadIdx.x < tp.num_coarse && cbase[threadIdx.x] <= p0) ? 1u : 0u;   // (whole waves: kMaxCoarse = 256)
        if (threadIdx.x < kMaxCoarse) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            if ((threadIdx.x & 63u) == 0u && c) atomicAdd(&s_le, c);
        }
        __syncthreads();
        bin = s_le - 1u;                                                    // (cbase[0] = 0 <= p0)
        lo = cbase[bin];
        hi = cbase[bin + 1];
        c_lo = (lo + kLocalCap > p0) ? lo + kLocalCap : p0;
        c_hi = (p0 + kLocalCap < hi) ? p0 + kLocalCap : hi;
        if (c_lo >= c_hi) return;
    }
    const uint32_t key0 = bin << tp.coarse_shift;
    const uint32_t bpc = 1u << tp.coarse_shift;
    const uint32_t lane = threadIdx.x & 63;
    auto key_of = [&](const float4 &r) {
        float cc[DIM];
        cc[0] = r.x;
        cc[1] = r.y;
        if constexpr (DIM == 3) cc[2] = r.z;
        return block_key<DIM>(cc, tp) - key0;
    };
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        header[0] = kCtxMagic;
        header[1] = tp.num_blocks;
        header[2] = n;
    }
    if (first && hi - lo <= kLocalCap) {
        // the usual case: every record of the bin in registers -- counted, ranked (the returning LDS atomic of the count IS the
        // rank) and written to its block's run. Inside a block the records are ordered by sub-cell of the block (SHACIRA_SORT_SUB
        // bits per axis; the key is block * cells + cell, block_start still addresses whole blocks): consecutive samples then
        // share a box of 1/2 or 1/4 of the block's side -- more lanes of one gather instruction fall into one table line on the
        // levels around the batch's density, at no extra pass.
#ifndef SHACIRA_SORT_SUB
#define SHACIRA_SORT_SUB 2          // bits per axis: 0 = block order only, 1 = octants, 2 = 4 x 4 x 4 cells of a block
#endif
        constexpr int kSubBits = SHACIRA_SORT_SUB;
        constexpr int kSub = 1 << (kSubBits * DIM);
        constexpr int kKeys = kBpc * kSub;                 // <= 16 * 64
        __shared__ uint32_t s_tot8[kKeys], s_wtot[16];
        uint32_t *s_off8 = s_tot8;                         // the scan below turns the counts into offsets in place
        for (uint32_t k = threadIdx.x; k < (uint32_t)kKeys; k += kSortThreadsS) s_tot8[k] = 0;
        __syncthreads();
        auto sub_of = [&](const float4 &r) -> uint32_t {
            if constexpr (kSub == 1) return 0u;
            uint32_t sb = 0;
            const float cc[3] = {r.x, r.y, r.z};
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                const float u = (cc[a] + 1.0f) * (0.5f * (float)tp.nb[a]);
                const float fr = u - floorf(u);            // (NaN / out-of-range coordinates: any cell is as good)
                uint32_t q = (uint32_t)(int)(fr * (float)(1 << kSubBits));
                q = q < (1u << kSubBits) ? q : (1u << kSubBits) - 1u;
                sb |= q << (a * kSubBits);
            }
            return sb;
        };
        float4 r[R];
        uint32_t j[R], rk[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const uint32_t p = lo + (uint32_t)u * kSortThreadsS + threadIdx.x;
            r[u] = inter4[p < hi ? p : (hi > 0u ? hi - 1u : 0u)];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const bool live = lo + (uint32_t)u * kSortThreadsS + threadIdx.x < hi;
            j[u] = live ? key_of(r[u]) * (uint32_t)kSub + sub_of(r[u]) : 0u;
            rk[u] = wave_rank_add(s_tot8, j[u], live, lane);
        }
        __syncthreads();
        // exclusive scan of the counts, IN PLACE (a thread reads and rewrites only its own KPT consecutive keys; the ranks are
        // in registers already): wave scans + wave totals
        {
            constexpr int KPT = (kKeys + kSortThreadsS - 1) / kSortThreadsS;
            uint32_t cnt[KPT], sum = 0;
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint32_t k = threadIdx.x * KPT + q;
                cnt[q] = k < (uint32_t)kKeys ? s_tot8[k] : 0u;
                sum += cnt[q];
            }
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t nbr = __shfl_up(incl, off, 64);
                if (lane >= (uint32_t)off) incl += nbr;
            }
            if (lane == 63) s_wtot[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += s_wtot[w];
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint32_t k = threadIdx.x * KPT + q;
                if (k < (uint32_t)kKeys) s_off8[k] = run;
                run += cnt[q];
            }
        }
        __syncthreads();
        if (threadIdx.x < bpc && key0 + threadIdx.x < tp.num_blocks) block_start[key0 + threadIdx.x] = lo + s_off8[threadIdx.x * kSub];
        if (bin + 1u == tp.num_coarse && threadIdx.x == 0) block_start[tp.num_blocks] = hi;
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (lo + (uint32_t)u * kSortThreadsS + threadIdx.x < hi) sorted4[lo + s_off8[j[u]] + rk[u]] = r[u];
        return;
    }
    // a chunk of an over-full bin (a batch concentrated in few blocks; the partition pass counted such bins' records per block):
    // block offsets of the bin from the global counts, then the chunk's <= 8 192 records, held in registers, ranked inside the
    // workgroup with LDS atomics (a wave whose lanes all name one block draws once) and ONE global reservation per block of the
    // bin (per-lane returning atomics on <= 16 global words serialised: 0.7 ms for a Gaussian blob of 2^20 samples). Order inside
    // a block is arrival order: these bins keep block order only (no sub-cell order).
    __shared__ uint32_t s_off[kBpc + 1], s_cnt[kBpc], s_base[kBpc];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < bpc; ++b) {
            s_off[b] = run;
            run += (key0 + b < tp.num_blocks) ? gcount[key0 + b] : 0u;
        }
        s_off[bpc] = run;
    }
    if (threadIdx.x < kBpc) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    if (first) {
        if (threadIdx.x < bpc && key0 + threadIdx.x < tp.num_blocks) block_start[key0 + threadIdx.x] = lo + s_off[threadIdx.x];
        if (bin + 1u == tp.num_coarse && threadIdx.x == 0) block_start[tp.num_blocks] = hi;
    }
    float4 r[R];
    uint32_t j[R], rk[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const uint32_t p = c_lo + (uint32_t)u * kSortThreadsS + threadIdx.x;
        r[u] = inter4[p < c_hi ? p : c_hi - 1u];
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const bool live = c_lo + (uint32_t)u * kSortThreadsS + threadIdx.x < c_hi;
        j[u] = live ? key_of(r[u]) : 0u;
        rk[u] = wave_rank_add(s_cnt, j[u], live, lane);
    }
    __syncthreads();
    if (threadIdx.x < bpc) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&gcursor[key0 + threadIdx.x], s_cnt[threadIdx.x]) : 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < R; ++u)
        if (c_lo + (uint32_t)u * kSortThreadsS + threadIdx.x < c_hi) sorted4[lo + s_off[j[u]] + s_base[j[u]] + rk[u]] = r[u];
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
    // measured rule (tools/attic/tiled_check.py, tools/attic/lego_fwd_check.py): tables that do not fit an XCD's L2 (the Kodak tables of
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

#define SHACIRA_CHECK_LAUNCH()                 \
    do {                                       \
        hipError_t e_ = hipGetLastError();     \
        if (e_ != hipSuccess) return e_;       \
    } while (0)

size_t sample_plan_bytes(int dim, int64_t n) {
    (void)dim;
    return n > 0 ? plan_bytes_of(n) : 0;
}
size_t sample_plan_scratch_bytes(int dim, int64_t n) { return n > 0 ? scratch_bytes_of(dim, n) : 0; }

// the sort: coordinates -> plan (sorted records + block offsets); three launches on `s`
hipError_t sample_plan_build(int dim, const float *coords, int64_t n, void *plan, void *scratch, hipStream_t s) {
    TilePlan tp;
    make_sort_plan(dim, n, tp);
    TileCtx c{};
    carve_plan(n, plan, c);
    carve_scratch(dim, n, scratch, c);
    // workgroups of the last pass: one per coarse bin + one per 8 192-record window (the remainders of over-full bins)
    const uint32_t chunks = tp.num_coarse + (uint32_t)((n + kLocalCap - 1) / kLocalCap);
    uint32_t *gcount = c.gcursor + kMaxBlocksS;
#define SHACIRA_SORT_LAUNCHES(D)                                                                                          \
    hipLaunchKernelGGL(psort_count_kernel<D>, dim3(tp.ptiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt, c.gcursor);  \
    SHACIRA_CHECK_LAUNCH();                                                                                                  \
    hipLaunchKernelGGL(psort_partition_kernel<D>, dim3(tp.ptiles), dim3(kSortThreadsS), 0, s, tp, coords, n, c.cnt, c.cbase, \
                       c.gcursor, c.inter4);                                                                                 \
    SHACIRA_CHECK_LAUNCH();                                                                                                  \
    hipLaunchKernelGGL(psort_local_kernel<D>, dim3(chunks), dim3(kSortThreadsS), 0, s, tp, c.inter4, c.cbase, gcount,        \
                       c.gcursor, c.sorted4, c.block_start, c.header, (uint32_t)n);
    if (dim == 3) {
        SHACIRA_SORT_LAUNCHES(3)
    } else {
        SHACIRA_SORT_LAUNCHES(2)
    }
#undef SHACIRA_SORT_LAUNCHES
    SHACIRA_CHECK_LAUNCH();
    return hipSuccess;
}

// block grid of the plan of (dim, n), no buffer at hand (workspace queries)
void sample_plan_grid(int dim, int64_t n, SortedBatch &out) {
    TilePlan tp;
    make_sort_plan(dim, n, tp);
    out.sorted4 = nullptr;
    out.block_start = nullptr;
    out.num_blocks = tp.num_blocks;
    for (int a = 0; a < 3; ++a) out.nb[a] = tp.nb[a];
}

// host-side view of a plan buffer built for (dim, n): pointers + block grid (no device read)
void sample_plan_view(int dim, int64_t n, const void *plan, SortedBatch &out) {
    TilePlan tp;
    make_sort_plan(dim, n, tp);
    TileCtx c{};
    carve_plan(n, const_cast<void *>(plan), c);
    out.sorted4 = c.sorted4;
    out.block_start = c.block_start;
    out.num_blocks = tp.num_blocks;
    for (int a = 0; a < 3; ++a) out.nb[a] = tp.nb[a];
}

// workspace of the cell-sorted forward: the fine levels' staging, the sort's scratch, and -- when the caller keeps no plan
// buffer of its own -- the plan
size_t tiled_forward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n) {
    return staged_bytes(dtype, lt, n) + scratch_bytes_of(dim, n) + plan_bytes_of(n);
}

hipError_t tiled_forward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                         const void *table, void *feats, void *workspace, int64_t n, hipStream_t s, void *plan,
                         bool plan_ready) {
    TilePlan tp;
    make_tile_plan(dim, lt, n, tp);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    void *staged = ws;
    unsigned char *scratch = ws + staged_bytes(dtype, lt, n);
    if (plan == nullptr) plan = scratch + scratch_bytes_of(dim, n);
    if (!plan_ready) {
        hipError_t e = sample_plan_build(dim, coords, n, plan, scratch, s);
        if (e != hipSuccess) return e;
    }
    TileCtx ctx{};
    carve_plan(n, plan, ctx);
    const int L = lt.num_lods;
    if (tp.lc < L) {   // fine levels: level-per-XCD pair kernel over the sorted coordinates -> staging [L][N][F]
        LevelTable fine = lt;
        fine.level_begin = tp.lc;
        fine.level_end = L;
        hipError_t e = hashgrid_forward_levels_staged(dim, dtype, fine, first_idx, reinterpret_cast<const float *>(ctx.sorted4),
                                                      table, staged, n, s);
        if (e != hipSuccess) return e;
    }
    return hashgrid_forward_rows(dim, dtype, lt, first_idx, reinterpret_cast<const float *>(ctx.sorted4), table, staged,
                                 feats, n, tp.lc, s);
}

}  // namespace shacira
