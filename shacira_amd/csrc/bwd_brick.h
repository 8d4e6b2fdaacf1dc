// bwd_brick.h -- the brick pass of the binned backward (round 6): coarse levels accumulated block by block from the batch's PLAN
// Part of the translation unit hashgrid_bwd_bin.hip (included there after bwd_bin_types.h).
//
// Replaces, for the levels it takes, the item stream of hashgrid_interpolate_backward_cuda_kernel's scatter-add
// (wisp/csrc/ops/hashgrid_interpolate_cuda.cu:143-221): same fp32 products grad * weight, summed on chip.
//
// The cell-sorted forward leaves the batch counting-sorted by spatial block (hashgrid_tiled.hip: 16-byte records {x, y, z,
// sample index} + block offsets -- the batch's plan). On a level whose cells are not much smaller than a block, the samples of
// one block touch only the vertices of a small box of that level (the block's BRICK: at most a few hundred to a thousand rows),
// many times each. A workgroup therefore walks a GROUP of x-adjacent blocks of the sorted records (up to 1 024 records at a time),
// adds every corner contribution into a brick-local 64-bit fixed-point LDS image per level, and flushes the image with coalesced
// float atomics: x-runs of adjacent rows on dense levels, rows x ^ h inside one or two 128-byte lines on hashed ones. Those
// levels leave the item stream altogether (S1: the dense levels 0-4 = 5 x 32 of the 864 item bytes per sample, written once and
// read once before; the hashed levels 5 and 6 fit, too, but cost more here than as items: profiles/r06_experiments.md 3).
//   * scale: the level's max |grad_output| from the transposing pass (front16_kernel), as in the consume pass; a level whose
//     maximum is not finite accumulates in fp64
//   * a sample whose cell falls outside its block's brick (cannot happen for coordinates the block rule and the clamp agree
//     on; kept as the safety net for the few-ulp cases) adds its corners with global atomics directly
#pragma once

#include "bwd_bin_types.h"

namespace shacira {

constexpr int kBrickMaxLevels = 6;
#ifndef SHACIRA_BRICK_THREADS
#define SHACIRA_BRICK_THREADS 512
#endif
#ifndef SHACIRA_BRICK_UNIT
#define SHACIRA_BRICK_UNIT 1024
#endif
constexpr int kBrickThreads = SHACIRA_BRICK_THREADS;
constexpr int kBrickUnit = SHACIRA_BRICK_UNIT;     // samples per unit (one zero / accumulate / flush round of a workgroup)
constexpr double kBrickSlack = 1e-3;               // cells: the block rule (fp32) and the cell rule (fp64 -> fp32) agree to ~1e-5

struct BrickLevel {
    int32_t res;
    float hi;
    uint32_t dense;
    uint32_t level;      // index into the table's levels
    uint32_t w[3];       // vertices of the image along x, y, z (the largest brick of any block)
    uint32_t row0;       // first row of this level's image inside the workgroup's image set
    uint32_t m_wx;       // ceil(2^20 / w[0]): row / w[0] == (row * m_wx) >> 20 for row < rows (checked by the host)
    uint32_t m_wy;       // same for the line index / w[1]
};

struct BrickPlan {
    uint32_t nlev;
    uint32_t level0;     // the brick levels are [level0, level0 + nlev)
    uint32_t rows_total;
    int32_t nb[3];
    uint32_t num_blocks;
    uint32_t span;       // a workgroup's unit = `span` consecutive blocks along x (one box: block ids are x-fastest)
    uint32_t groups_x;   // ceil(nb[0] / span); grid.x = groups_x * nb[1] * nb[2]
    BrickLevel lv[kBrickMaxLevels];
};

// first cell along one axis that a sample of block q (of nb) can fall into on a level of resolution res
__host__ __device__ inline int32_t brick_lo(int32_t res, int32_t q, int32_t nb) {
    const int32_t v = (int32_t)floor((double)res * (double)q / (double)nb - kBrickSlack);
    return v < 0 ? 0 : (v > res - 2 ? res - 2 : v);   // (blocks thinner than a cell: the last ones start behind the last cell)
}
__host__ __device__ inline int32_t brick_hi(int32_t res, int32_t q, int32_t nb) {
    const int32_t v = (int32_t)floor((double)res * (double)(q + 1) / (double)nb + kBrickSlack);
    return v > res - 2 ? res - 2 : v;    // the clamp keeps positions <= res - 2 (res < 512: hi = res - 1 - 1e-5 is not rounded up)
}

// vertices along one axis of the largest brick of `span` consecutive blocks (q0 = 0, span, 2 span, ...): cells lo..hi own
// vertices lo..hi + 1
static inline uint32_t brick_extent(int32_t res, int32_t nb, int32_t span) {
    int32_t w = 2;
    for (int32_t q = 0; q < nb; q += span) {
        const int32_t ql = (q + span - 1 < nb) ? q + span - 1 : nb - 1;
        const int32_t e = brick_hi(res, ql, nb) - brick_lo(res, q, nb) + 2;
        w = e > w ? e : w;
    }
    return (uint32_t)w;
}

// Which levels the brick pass takes, and their images. A level qualifies when a block's samples revisit its brick often
// enough to pay for the flush (>= kReuse contributions per vertex) and the brick is not so small that the lanes of an LDS
// atomic pile onto a handful of addresses (>= kMinVerts vertices; the compact items stay the better form there). The levels
// must form one contiguous range and their images fit `lds_budget`.
static bool make_brick_plan(const LevelTable &lt, int64_t n, const SortedBatch &sb, int lo_opt, int hi_opt, int span_opt,
                            size_t lds_budget, BrickPlan &bp) {
    bp.nlev = 0;
    bp.level0 = 0;
    bp.rows_total = 0;
    bp.num_blocks = sb.num_blocks;
    for (int a = 0; a < 3; ++a) bp.nb[a] = sb.nb[a];
    if (sb.num_blocks == 0 || n <= 0) return false;
    const int F = lt.feature_dim;
    const double spb1 = (double)n / (double)sb.num_blocks;      // samples per block
    // blocks per unit: ~3/4 of a unit's capacity on a uniform batch (fewer, larger bricks: less halo to flush, and the
    // zero / flush phases of a workgroup are amortised over more samples)
    int span = (int)(0.75 * kBrickUnit / spb1);
    if (span_opt > 0) span = span_opt;
    span = span < 1 ? 1 : (span > sb.nb[0] ? sb.nb[0] : span);
    bp.span = (uint32_t)span;
    bp.groups_x = (uint32_t)((sb.nb[0] + span - 1) / span);
    const double spb = spb1 * span;
    constexpr double kReuse = 3.0, kMinVerts = 0.0;
    int first = -1, last = -1;
    for (int l = lt.level_begin; l < lt.level_end; ++l) {
        const int32_t r = lt.res[l];
        if (r < 4 || r > 256) {
            if (first >= 0) break;
            continue;
        }
        double verts = 1.0;
        for (int a = 0; a < 3; ++a) verts *= (double)r * (a == 0 ? span : 1) / (double)sb.nb[a] + 1.0;
        bool ok = verts >= kMinVerts && 8.0 * spb >= kReuse * verts;
        if (lo_opt >= 0) ok = l >= lo_opt && l < hi_opt;        // explicit range (options bwd_brick_lo / bwd_brick_hi)
        if (!ok) {
            if (first >= 0) break;
            continue;
        }
        if (first < 0) first = l;
        last = l;
        if (last - first + 1 == kBrickMaxLevels) break;
    }
    if (first < 0) return false;
    // images; levels are dropped from the fine end until the set fits
    for (;;) {
        uint32_t rows = 0;
        bool magic_ok = true;
        bp.nlev = (uint32_t)(last - first + 1);
        bp.level0 = (uint32_t)first;
        for (uint32_t q = 0; q < bp.nlev; ++q) {
            BrickLevel &b = bp.lv[q];
            const int l = first + (int)q;
            b.res = lt.res[l];
            b.hi = lt.hi[l];
            b.dense = lt.dense[l];
            b.level = (uint32_t)l;
            for (int a = 0; a < 3; ++a) b.w[a] = brick_extent(b.res, sb.nb[a], a == 0 ? span : 1);
            b.row0 = rows;
            const uint32_t lvl_rows = b.w[0] * b.w[1] * b.w[2];
            b.m_wx = ((1u << 20) + b.w[0] - 1u) / b.w[0];
            b.m_wy = ((1u << 20) + b.w[1] - 1u) / b.w[1];
            if (lvl_rows >= (1u << 12) || b.w[1] * b.w[2] >= (1u << 12)) magic_ok = false;
            for (uint32_t v = 0; v < lvl_rows && magic_ok; ++v) magic_ok = ((v * b.m_wx) >> 20) == v / b.w[0];
            for (uint32_t v = 0; v < b.w[1] * b.w[2] && magic_ok; ++v) magic_ok = ((v * b.m_wy) >> 20) == v / b.w[1];
            rows += lvl_rows;
        }
        bp.rows_total = rows;
        if (magic_ok && (size_t)rows * F * sizeof(double) <= lds_budget) return true;
        if (last == first) {
            bp.nlev = 0;
            return false;
        }
        --last;
    }
}

// One sample's corners on one brick level, straight to the table (the safety net, see the header)
template <int F>
__device__ __forceinline__ void brick_direct_add(const BrickLevel &b, uint32_t mask, const int32_t (&p)[3], const float (&f)[3],
                                                 const float (&g)[F], float *__restrict__ acc, int64_t level_row0,
                                                 int64_t table_rows) {
    const float gx = 1.0f - f[0], gy = 1.0f - f[1], gz = 1.0f - f[2];
    const uint32_t r = (uint32_t)b.res;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
        const uint32_t ux = (uint32_t)p[0] + bx, uy = (uint32_t)p[1] + by, uz = (uint32_t)p[2] + bz;
        const float w = ((bx ? f[0] : gx) * (by ? f[1] : gy)) * (bz ? f[2] : gz);
        uint32_t row;
        if (b.dense) {
            if (ux >= r || uy >= r || uz >= r) continue;
            row = ux + uy * r + uz * r * r;
        } else {
            row = (ux ^ (uy * kPrimeY) ^ (uz * kPrimeZ)) & mask;
        }
        const int64_t grow = level_row0 + (int64_t)row;
        if ((uint64_t)grow >= (uint64_t)table_rows) continue;
#pragma unroll
        for (int j = 0; j < F; ++j) unsafeAtomicAdd(acc + grow * F + j, g[j] * w);
    }
}

// The pass's work: one unit = up to kBrickUnit consecutive records of one GROUP (`span` x-adjacent blocks). Workgroup g < groups
// takes group g's FIRST unit straight from the plan's block offsets (all of a uniform batch). What an over-full group holds
// beyond its first unit is cut at the kBrickUnit-aligned windows of the sorted batch, and workgroup groups + v takes window v:
// the group holding the window's first record is the only one whose remainder can reach into that window (the next group's
// remainder starts >= kBrickUnit records behind its own start), so the workgroup finds its unit with one pass over the block
// offsets and no list. A batch concentrated in few blocks thereby becomes many units on many workgroups instead of a loop inside
// a few (round 6: with grid.y = 4 workgroups per group a 2^20-sample batch inside one block took 4.4 ms, tools/skew_check.py).
__device__ __forceinline__ void brick_group_range(const BrickPlan &bp, const uint32_t *__restrict__ block_start, uint32_t g,
                                                  uint32_t &lo, uint32_t &hi) {
    const uint32_t nbx = (uint32_t)bp.nb[0];
    const uint32_t gx = g % bp.groups_x, gyz = g / bp.groups_x;
    const uint32_t blk0 = gx * bp.span + nbx * gyz;
    const uint32_t blk1 = (gx * bp.span + bp.span < nbx) ? blk0 + bp.span : nbx * (gyz + 1u);
    lo = block_start[blk0];
    hi = block_start[blk1];
}

// gT: the transposed gradients [L][gpitch][F] fp32 in SORTED order (front16_kernel<.., SORTED>): sample i of the sorted batch
template <int F>
__global__ __launch_bounds__(kBrickThreads) void brick_accumulate_kernel(LevelTable lt, BrickPlan bp,
                                                                         const int32_t *__restrict__ first_idx,
                                                                         const float4 *__restrict__ sorted4,
                                                                         const uint32_t *__restrict__ block_start,
                                                                         const float *__restrict__ gT, int64_t gpitch,
                                                                         float *__restrict__ acc,
                                                                         const uint32_t *__restrict__ gmax, int headroom) {
    extern __shared__ double s_img[];                 // [rows_total][F]: 64-bit fixed point (or fp64)
    unsigned long long *s_fix = reinterpret_cast<unsigned long long *>(s_img);
    // the unit: up to kBrickUnit records of one group = `span` consecutive blocks along x at (qy, qz)
    const uint32_t groups = bp.groups_x * (uint32_t)(bp.nb[1] * bp.nb[2]);
    uint4 unit;
    if (blockIdx.x < groups) {
        uint32_t lo, hi;
        brick_group_range(bp, block_start, blockIdx.x, lo, hi);
        if (lo >= hi) return;
        unit = make_uint4(lo, (lo + (uint32_t)kBrickUnit < hi) ? lo + (uint32_t)kBrickUnit : hi, blockIdx.x, 0u);
    } else {
        // window v of the sorted batch: the block holding its first record = (offsets <= that record) - 1
        __shared__ uint32_t s_le;
        const uint32_t p0 = (blockIdx.x - groups) * (uint32_t)kBrickUnit;
        if (threadIdx.x == 0) s_le = 0u;
        __syncthreads();
        uint32_t c = 0;
        for (uint32_t k = threadIdx.x; k < bp.num_blocks; k += kBrickThreads) c += block_start[k] <= p0 ? 1u : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if ((threadIdx.x & 63u) == 0u && c) atomicAdd(&s_le, c);
        __syncthreads();
        const uint32_t blk = s_le - 1u;                      // (block_start[0] = 0 <= p0)
        const uint32_t nbx = (uint32_t)bp.nb[0];
        const uint32_t g = (blk % nbx) / bp.span + bp.groups_x * (blk / nbx);
        uint32_t lo, hi;
        brick_group_range(bp, block_start, g, lo, hi);
        const uint32_t u0 = (lo + (uint32_t)kBrickUnit > p0) ? lo + (uint32_t)kBrickUnit : p0;
        const uint32_t u1 = (p0 + (uint32_t)kBrickUnit < hi) ? p0 + (uint32_t)kBrickUnit : hi;
        if (u0 >= u1) return;
        unit = make_uint4(u0, u1, g, 0u);
    }
    const uint32_t nby = (uint32_t)bp.nb[1];
    const uint32_t gx = unit.z % bp.groups_x, gyz = unit.z / bp.groups_x;
    const int32_t qx = (int32_t)(gx * bp.span), qy = (int32_t)(gyz % nby), qz = (int32_t)(gyz / nby);
    const int rot = (int)(threadIdx.x & (F - 1));
    const int rotc = (int)((threadIdx.x >> 1) & 7u);
    constexpr int SPT = kBrickUnit / kBrickThreads;   // records a thread keeps per unit
    {
        const uint32_t u0 = unit.x, u1 = unit.y;
        // the thread's records: unconditional loads from clamped positions, in flight while the image is zeroed
        float4 rec[SPT];
        bool live[SPT];
        uint32_t pos[SPT];
        // Slot k of the unit takes sample (k & 7) * S + (k >> 3), S = ceil(samples / 8): the eight neighbouring lanes of an LDS
        // atomic come from eight distant stretches of the unit (the plan orders a block's records by octant, so neighbours in
        // memory sit in the same cells and would pile onto the same image rows); a bijection onto the unit's samples.
#ifndef SHACIRA_BRICK_INTERLEAVE
#define SHACIRA_BRICK_INTERLEAVE 1
#endif
        const uint32_t nloc = u1 - u0, S8 = (nloc + 7u) >> 3;
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const uint32_t k = (uint32_t)u * kBrickThreads + threadIdx.x;
            uint32_t i = u0 + k;
            live[u] = i < u1;
            if (SHACIRA_BRICK_INTERLEAVE) {
                const uint32_t kk = (k & 7u) * S8 + (k >> 3);
                live[u] = (k >> 3) < S8 && kk < nloc;
                i = u0 + kk;
            }
            pos[u] = live[u] ? i : u1 - 1u;
            rec[u] = sorted4[pos[u]];
        }
        // gradient pieces of one level (F floats per sample, coalesced: gT is in sorted order), loaded one level AHEAD of
        // their use: a level's LDS atomics run while the next level's gradients are in flight
        auto load_g = [&](uint32_t level, float (&g)[SPT][F]) {
#pragma unroll
            for (int u = 0; u < SPT; ++u) {
                const float *gp = gT + ((int64_t)level * gpitch + pos[u]) * F;
                if constexpr (F == 2) {
                    const float2 v = *reinterpret_cast<const float2 *>(gp);
                    g[u][0] = v.x;
                    g[u][1] = v.y;
                } else if constexpr (F == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(gp);
                    g[u][0] = v.x; g[u][1] = v.y; g[u][2] = v.z; g[u][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < F; ++j) g[u][j] = gp[j];
                }
            }
        };
        float gn[SPT][F];
        load_g(bp.lv[0].level, gn);
        for (uint32_t e = threadIdx.x; e < bp.rows_total * F; e += kBrickThreads) s_fix[e] = 0ull;
        lds_barrier();
        // levels outside, samples inside: one level's parameters are wave-uniform scalars
#pragma unroll 1
        for (uint32_t q = 0; q < bp.nlev; ++q) {
            const BrickLevel b = bp.lv[q];
            float g[SPT][F];
#pragma unroll
            for (int u = 0; u < SPT; ++u)
#pragma unroll
                for (int j = 0; j < F; ++j) g[u][j] = gn[u][j];
            load_g(bp.lv[q + 1u < bp.nlev ? q + 1u : q].level, gn);
            const FxScale fx = fx_scale_of(gmax[b.level], headroom);
            const int32_t lo0 = brick_lo(b.res, qx, bp.nb[0]), lo1 = brick_lo(b.res, qy, bp.nb[1]),
                          lo2 = brick_lo(b.res, qz, bp.nb[2]);
            const uint32_t wx = b.w[0], wxy = b.w[0] * b.w[1];
#pragma unroll
            for (int u = 0; u < SPT; ++u) {
                if (!live[u]) continue;
                int32_t p[3];
                float f[3], gg[3];
                axis_transform(axis_unit(rec[u].x), b.res, b.hi, p[0], f[0], gg[0]);
                axis_transform(axis_unit(rec[u].y), b.res, b.hi, p[1], f[1], gg[1]);
                axis_transform(axis_unit(rec[u].z), b.res, b.hi, p[2], f[2], gg[2]);
                const uint32_t lx = (uint32_t)(p[0] - lo0), ly = (uint32_t)(p[1] - lo1), lz = (uint32_t)(p[2] - lo2);
                if (lx + 1u >= b.w[0] || ly + 1u >= b.w[1] || lz + 1u >= b.w[2]) {   // (unsigned: cells in front of the brick too)
                    brick_direct_add<F>(b, lt.mask, p, f, g[u], acc, (int64_t)first_idx[b.level], lt.table_rows);
                    continue;
                }
                const uint32_t base = b.row0 + lx + ly * wx + lz * wxy;
                float gr[F];
                rotate_features<F>(g[u], rot, gr);
                // corners in a lane-dependent order (LDS bank / same-address spreading: the samples of one cell would otherwise
                // add the same corner in the same instruction), weights as the reference forms them: (x * y) * z
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int cc = (c + rotc) & 7;
                    const bool bx = (cc & 4) != 0, by = (cc & 2) != 0, bz = (cc & 1) != 0;
                    const float w = ((bx ? f[0] : gg[0]) * (by ? f[1] : gg[1])) * (bz ? f[2] : gg[2]);
                    const uint32_t row = base + (bx ? 1u : 0u) + (by ? wx : 0u) + (bz ? wxy : 0u);
#pragma unroll
                    for (int jj = 0; jj < F; ++jj) {
                        const int j = (jj + rot) & (F - 1);
                        const float cv = gr[jj] * w;
                        if (fx.fixed) atomicAdd(&s_fix[row * F + j], fx_encode(cv, fx.scale));
                        else atomicAdd(&s_img[row * F + j], (double)cv);
                    }
                }
            }
        }
        lds_barrier();
        // flush: lanes run along x (and over the F features of a row): adjacent rows of a dense level, rows x ^ h of a hashed one
        // (walking a hashed level's aligned 16-row groups in ADDRESS order instead changed nothing: measured, round 6)
#pragma unroll 1
        for (uint32_t q = 0; q < bp.nlev; ++q) {
            const BrickLevel b = bp.lv[q];
            const FxScale fx = fx_scale_of(gmax[b.level], headroom);
            const int32_t lo0 = brick_lo(b.res, qx, bp.nb[0]), lo1 = brick_lo(b.res, qy, bp.nb[1]),
                          lo2 = brick_lo(b.res, qz, bp.nb[2]);
            const uint32_t rows = b.w[0] * b.w[1] * b.w[2];
            const uint32_t r = (uint32_t)b.res;
            const int64_t level_row0 = (int64_t)first_idx[b.level];
            for (uint32_t e = threadIdx.x; e < rows * F; e += kBrickThreads) {
                const uint32_t le = b.row0 * F + e;
                const float v = fx.fixed ? fx_decode(s_fix[le], fx.inv) : (float)s_img[le];
                if (v == 0.0f) continue;
                const uint32_t row = e / F, j = e & (F - 1);
                const uint32_t line = (row * b.m_wx) >> 20, ix = row - line * b.w[0];
                const uint32_t iz = (line * b.m_wy) >> 20, iy = line - iz * b.w[1];
                const uint32_t ux = (uint32_t)lo0 + ix, uy = (uint32_t)lo1 + iy, uz = (uint32_t)lo2 + iz;
                uint32_t grow_l;
                if (b.dense) {
                    if (ux >= r || uy >= r || uz >= r) continue;
                    grow_l = ux + uy * r + uz * r * r;
                } else {
                    grow_l = (ux ^ (uy * kPrimeY) ^ (uz * kPrimeZ)) & lt.mask;
                }
                const int64_t grow = level_row0 + (int64_t)grow_l;
                if ((uint64_t)grow < (uint64_t)lt.table_rows) unsafeAtomicAdd(acc + grow * F + j, v);
            }
        }
    }
}

}  // namespace shacira
