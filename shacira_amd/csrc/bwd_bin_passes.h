// bwd_bin_passes.h -- the item passes of the binned backward: scatter (pass B), consume (pass C), direct levels (pass D)
// Part of the translation unit hashgrid_bwd_bin.hip (included there, in this order: bwd_bin_types.h, bwd_bin_front.h,
// bwd_bin_passes.h); see that file's header for the pipeline.
#pragma once

#include "bwd_bin_types.h"

namespace shacira {

// ------------------------------------------------------------------------------------------------- pass B
// Gradients from the transposed image gT [L][NP][F], grid (tiles, binned levels). A (tile, bucket) run is reserved with one
// returning atomic on the bucket's cursor (set to the bucket's base by the bucket scan): runs of different tiles land in
// the bucket in arrival order -- the consumer's fixed-point sums do not depend on it.
// H: half-precision item stream (fp16 tables, F = 2): 8-byte pair items, 16-byte compact items.
#ifdef SHACIRA_SCATTER_WAVES   // A/B builds: cap the registers so that this many waves share a SIMD
#define SHACIRA_SCATTER_ATTR __attribute__((amdgpu_waves_per_eu(SHACIRA_SCATTER_WAVES)))
#else
#define SHACIRA_SCATTER_ATTR
#endif
template <int DIM, int F, int FMT, bool STREAM = true>
__global__ __launch_bounds__((ScatterThreads<DIM, F, FMT>::value)) SHACIRA_SCATTER_ATTR void bin_scatter_kernel(LevelTable lt, BinPlan plan,
                                                                  const float *__restrict__ coords,
                                                                  const float *__restrict__ gT,
                                                                  unsigned long long *__restrict__ cursor,
                                                                  const uint32_t *__restrict__ cnt, uint32_t cps,
                                                                  uint32_t cnt_rows,
                                                                  typename ItemSel<F, FMT>::type *__restrict__ items,
                                                                  int64_t sample0, int64_t N, int64_t gpitch,
                                                                  float *__restrict__ zero_acc,
                                                                  const int32_t *__restrict__ first_idx,
                                                                  const uint32_t *__restrict__ unit_first,
                                                                  int cstride = DIM) {   // floats per coordinate record
    typedef typename ItemSel<F, FMT>::type ItemT;
    constexpr bool H = FMT == 1;      // half-precision stream (fp16 tables)
    constexpr bool P12 = FMT == 2;    // 12-byte units (fp32 tables, 3-D, F = 2, large batches)
    static_assert(!P12 || (DIM == 3 && F == 2), "12-byte units: 3-D, F = 2");
    constexpr int NP = 1 << (DIM - 1);
    constexpr int kTileD = TileOf<DIM>::value;
    constexpr int kThreads = ScatterThreads<DIM, F, FMT>::value;
    constexpr int SPT = kTileD / kThreads;  // samples per thread
    // staged items per PASS: the tile's kTileD * NP items (sorted by bucket) go through the LDS buffer in kScatterSplit windows
    // of their sorted order, so the buffer is 1 / kScatterSplit of the tile and more workgroups share a CU (round 4: F = 2
    // 69 -> 35 KiB = four instead of two, F = 4 100 -> 51 KiB = three instead of one) while a (tile, bucket) run keeps its
    // length -- it is written in one window, or in two parts where a window boundary falls inside it
    constexpr int kScatterSplit = ScatterSplit<ItemT>::value;
    constexpr int kStage = kTileD * NP / kScatterSplit;
    static_assert(kStage % 2 == 0, "two-slot items never straddle a window");
    extern __shared__ __align__(16) unsigned char s_raw[];
    ItemT *s_items = reinterpret_cast<ItemT *>(s_raw);
    // (one window: the buffer holds the tile's items plus the pad units of its runs, plan.stage_cap >= kStage; windows: exact runs)
    uint8_t *s_bucket = reinterpret_cast<uint8_t *>(s_items + (kScatterSplit == 1 ? plan.stage_cap : (uint32_t)kStage));
    __shared__ uint32_t s_hist[kMaxLevelBuckets];
    __shared__ uint32_t s_start[kMaxLevelBuckets + 1];
    __shared__ uint64_t s_gbase[kMaxLevelBuckets];

    // (tile-fastest numbering; level-fastest -- a tile's levels back to back -- measured 3 % slower, round 3)
#ifdef SCATTER_LEVEL_FAST   // A/B builds: a tile's levels back to back (grid = (levels, tiles))
    const uint32_t tile = blockIdx.y, bi = blockIdx.x;
#else
    const uint32_t tile = blockIdx.x, bi = blockIdx.y;
#endif
    const uint32_t lvl = plan.blevel[bi];
    const BinLevel bl = plan.lv[lvl];
    if (threadIdx.x < kMaxLevelBuckets) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int32_t res = lt.res[lvl];
    const float hi = lt.hi[lvl];
    const bool dense = lt.dense[lvl] != 0;

    PairSlot ps[SPT][NP];
    uint32_t rank[SPT][NP];
    float fx[SPT];
    float g[SPT][F];
    float fyz[SPT][2];   // compact levels: y / z fractions travel with the item
    const bool compact = bl.compact != 0;   // (2-D: fp32 item stream only -- the plan never marks a level otherwise)
    constexpr uint32_t kCompactSlots = (DIM == 3) ? 2u : 1u;
    // every global load of the workgroup up front, unconditional (indices clamped into the batch): coordinates and
    // gradients of the thread's samples, then -- waves 1 and 2 -- the tile's bucket counts (rows of cnt[tile][bucket]
    // written by the counting pass) and straight away the returning atomic that reserves the bucket's run: it is the
    // YOUNGEST memory operation of the wave, so nothing below waits for it until the run offsets are needed (after the
    // staging phase); issued after the ranking instead, its round trip cost 23 us on S1
    float craw[SPT][DIM], graw[SPT][F];
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        int64_t i = sample0 + (int64_t)tile * kTileD + threadIdx.x + u * kThreads;
        i = i < N ? i : N - 1;
#ifdef ABL_NO_LOAD    // every thread reads the same few samples: cached loads, same arithmetic downstream
        i = (i * 2654435761ll) & 1023;
#endif
#pragma unroll
        for (int a = 0; a < DIM; ++a) craw[u][a] = coords[i * cstride + a];
        const float *gp = gT + ((int64_t)lvl * gpitch + i) * F;
        if constexpr (F == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(gp);
            graw[u][0] = v.x; graw[u][1] = v.y;
        } else {
#pragma unroll
            for (int j = 0; j < F; ++j) graw[u][j] = gp[j];
        }
    }
    // (round 5, tried and dropped: wave 1 laying out the tile's runs from the counting pass's numbers up front -- scan, pad fill
    // and final staging positions drawn by the ranking atomics themselves, two barriers and the scan phase fewer -- made this
    // kernel 75 us SLOWER on S1: every wave then waits behind one wave's global loads; profiles/r05_experiments.md)
    const bool reserver = threadIdx.x >= 64 && threadIdx.x - 64 < bl.nb;
    unsigned long long run_base = 0ull;
    if (reserver) {
        // cnt[counting tile][bucket] holds the ACTUAL numbers of item units; the run is reserved in multiples of plan.pad
        const uint32_t gb = bl.bucket0 + threadIdx.x - 64;
        const uint32_t *row = cnt + (size_t)tile * cps * plan.total_buckets + gb;
        uint32_t c = 0;
        for (uint32_t k = 0; k < cps && tile * cps + k < cnt_rows; ++k) c += row[(size_t)k * plan.total_buckets];
        c = (c + plan.pad - 1u) & ~(plan.pad - 1u);
#ifdef ABL_NO_CURSOR   // ablation: no reservation (runs overlap: wrong results on purpose)
        if (c) run_base = cursor[gb] + (unsigned long long)tile * c;
#else
        if (c) run_base = atomicAdd(&cursor[gb], (unsigned long long)c);
#endif
    }
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const int k = threadIdx.x + u * kThreads;
        const int64_t i = sample0 + (int64_t)tile * kTileD + k;
        const bool live = i < N;
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(craw[u][a]);
#pragma unroll
        for (int j = 0; j < F; ++j) g[u][j] = graw[u][j];
        if (compact) {
            if constexpr (DIM == 3) {
                // slot key = local row of the base corner inside the bucket's image (slab + halo planes) | valid bit
                int32_t pp[3];
                float ff[3], gg[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) axis_transform(t[a], res, hi, pp[a], ff[a], gg[a]);
                const uint32_t r = (uint32_t)res, b = (uint32_t)pp[2] / bl.slab;
                const uint32_t local = ((uint32_t)pp[2] - b * bl.slab) * r * r + (uint32_t)pp[1] * r + (uint32_t)pp[0];
                fx[u] = ff[0];
                fyz[u][0] = ff[1];
                fyz[u][1] = ff[2];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    ps[u][q].bucket = b;
                    ps[u][q].key = 0;
                    ps[u][q].wrest = 0.0f;
                }
                ps[u][0].key = local | (1u << 26);
            } else {
                // 2-D: local row of the base corner inside the bucket's image (slab of lines + one halo line)
                int32_t pp[2];
                float ff[2], gg[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) axis_transform(t[a], res, hi, pp[a], ff[a], gg[a]);
                const uint32_t r = (uint32_t)res, b = compact2d_line((uint32_t)pp[1], r) / bl.slab;
                const uint32_t local = ((uint32_t)pp[1] - b * bl.slab) * r + (uint32_t)pp[0];
                fx[u] = ff[0];
                fyz[u][0] = ff[1];
                fyz[u][1] = 0.0f;
                ps[u][0].bucket = b;
                ps[u][0].key = local | (1u << 26);
                ps[u][0].wrest = 0.0f;
                ps[u][1].bucket = b;
                ps[u][1].key = 0;
                ps[u][1].wrest = 0.0f;
            }
        } else {
            enumerate_pairs<DIM>(t, res, hi, dense, lt.mask, bl, plan.BR, fx[u], ps[u]);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (!live) ps[u][q].key = 0;
            rank[u][q] = (ps[u][q].key >> 26) ? atomicAdd(&s_hist[ps[u][q].bucket], compact ? kCompactSlots : 1u) : 0u;
        }
    }
    lds_barrier();   // (not __syncthreads(): its vmcnt(0) would wait for the reservation)
    if (threadIdx.x < 64) {  // wave 0: exclusive scan of the <= 128 bucket counts, two per lane
        const uint32_t lane = threadIdx.x;
        // run lengths in multiples of plan.pad (line-aligned runs; what the reservation above took)
        const uint32_t pm = plan.pad - 1u;
        const uint32_t c0 = (2 * lane < bl.nb) ? ((s_hist[2 * lane] + pm) & ~pm) : 0u;
        const uint32_t c1 = (2 * lane + 1 < bl.nb) ? ((s_hist[2 * lane + 1] + pm) & ~pm) : 0u;
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nbr = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += nbr;
        }
        const uint32_t excl = incl - (c0 + c1);
        if (2 * lane < bl.nb) s_start[2 * lane] = excl;
        if (2 * lane + 1 < bl.nb) s_start[2 * lane + 1] = excl + c0;
        if (lane == 63) s_start[bl.nb] = incl;
    }
    lds_barrier();
    // one window (kScatterSplit == 1: the loop and the window test vanish at compile time -- the formats that do not gain
    // from windows keep the code they had) or ceil(staged / kStage) of them; an item's position in the tile's sorted order is
    // re-read per window (an array of positions cost ten registers and 1-3 % on every format)
    if (kScatterSplit == 1 && plan.pad > 1u && threadIdx.x < bl.nb) {
        // pad units behind the bucket's items: all-zero units (no valid corner in any item format), written out with the run
        const uint32_t b = threadIdx.x;
        for (uint32_t pos = s_start[b] + s_hist[b]; pos < s_start[b + 1]; ++pos) {
            uint32_t *w = reinterpret_cast<uint32_t *>(&s_items[pos]);
#pragma unroll
            for (int k = 0; k < (int)(sizeof(ItemT) / 4); ++k) w[k] = 0u;
            s_bucket[pos] = (uint8_t)b;
        }
    }
    const uint32_t staged = s_start[bl.nb];
    const uint32_t n_win = (kScatterSplit == 1) ? 1u : (staged + (uint32_t)kStage - 1u) / (uint32_t)kStage;
    for (uint32_t wi = 0; wi < n_win; ++wi) {
    const uint32_t w0 = wi * (uint32_t)kStage;
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (ps[u][q].key >> 26) {
                const uint32_t gp = s_start[ps[u][q].bucket] + rank[u][q];
                if (kScatterSplit > 1 && gp - w0 >= (uint32_t)kStage) continue;   // (unsigned: other windows' items)
                const uint32_t pos = gp - w0;
                if constexpr (H && F == 4) {
                    ItemH4 it;
                    it.key = ps[u][q].key;
                    it.fx = fx[u];
                    if (compact) {   // two 16-byte units: {key, fx, fy, fz} {g01, g23, -, -}
                        it.p2 = __float_as_uint(fyz[u][0]);
                        it.p3 = __float_as_uint(fyz[u][1]);
                        s_items[pos] = it;
                        ItemH4 it2;
                        it2.key = 0;
                        it2.fx = 0.0f;
                        it2.p2 = float2_to_half2_bits(g[u][0], g[u][1]);
                        it2.p3 = float2_to_half2_bits(g[u][2], g[u][3]);
                        s_items[pos + 1] = it2;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        s_bucket[pos + 1] = (uint8_t)ps[u][q].bucket;
                    } else {
                        const float w = ps[u][q].wrest;
                        it.p2 = float2_to_half2_bits(g[u][0] * w, g[u][1] * w);
                        it.p3 = float2_to_half2_bits(g[u][2] * w, g[u][3] * w);
                        s_items[pos] = it;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                    }
                    continue;
                }
                if constexpr (P12) {
                    if (compact) {   // two 12-byte units {key, fx, fy} {fz, g0, g1}
                        Item12 u0, u1;
                        u0.w[0] = ps[u][q].key;
                        u0.w[1] = __float_as_uint(fx[u]);
                        u0.w[2] = __float_as_uint(fyz[u][0]);
                        u1.w[0] = __float_as_uint(fyz[u][1]);
                        u1.w[1] = __float_as_uint(g[u][0]);
                        u1.w[2] = __float_as_uint(g[u][1]);
                        s_items[pos] = u0;
                        s_items[pos + 1] = u1;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        s_bucket[pos + 1] = (uint8_t)ps[u][q].bucket;
                    } else {
                        s_items[pos] = pack_item12(ps[u][q].key, fx[u], dense, g[u][0] * ps[u][q].wrest, g[u][1] * ps[u][q].wrest);
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                    }
                    continue;
                }
                if constexpr (H && F == 2) {
                    if (compact) {   // one 16-byte item = two 8-byte units (pos is even: every count of the level is)
                        ItemHC c;
                        c.key = ps[u][q].key;
                        c.fx = (uint16_t)(fx[u] * 65536.0f);
                        c.fy = (uint16_t)(fyz[u][0] * 65536.0f);
                        c.fz = (uint16_t)(fyz[u][1] * 65536.0f);
                        c.pad = 0;
                        c.g = __floats2half2_rn(g[u][0], g[u][1]);
                        *reinterpret_cast<ItemHC *>(&s_items[pos]) = c;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        s_bucket[pos + 1] = (uint8_t)ps[u][q].bucket;
                    } else {
                        ItemH it;
                        it.key = pack_half_key(ps[u][q].key, fx[u], dense);
                        it.a = __floats2half2_rn(g[u][0] * ps[u][q].wrest, g[u][1] * ps[u][q].wrest);
                        s_items[pos] = it;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                    }
                    continue;
                }
                if constexpr (!H && !P12) {
                Item<F> it;
                it.key = ps[u][q].key;
                it.fx = fx[u];
                if constexpr (DIM == 2 && !H) {
                    if (compact) {   // one slot: {local | valid | fx, fy (25-bit fixed point), g[F]}
                        uint32_t w0, w1;
                        pack_compact2d(ps[u][q].key, fx[u], fyz[u][0], w0, w1);
                        it.key = w0;
                        it.fx = __uint_as_float(w1);
#pragma unroll
                        for (int j = 0; j < F; ++j) it.a[j] = g[u][j];
                        *reinterpret_cast<Item<F> *>(&s_items[pos]) = it;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        continue;
                    }
                }
                if (compact) {
                    if constexpr (F == 2) {   // two slots: {key, fx, fy, fz} {0, g0, g1, 0}
                        it.a[0] = fyz[u][0];
                        it.a[1] = fyz[u][1];
                        *reinterpret_cast<Item<F> *>(&s_items[pos]) = it;
                        Item<F> it2;
                        it2.key = 0;
                        it2.fx = g[u][0];
                        it2.a[0] = g[u][1];
                        it2.a[1] = 0.0f;
                        *reinterpret_cast<Item<F> *>(&s_items[pos + 1]) = it2;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        s_bucket[pos + 1] = (uint8_t)ps[u][q].bucket;
                    } else if constexpr (F == 4) {   // two 24-byte slots: {key, fx, fy, fz, g0, g1} {0, g2, g3, -, -, -}
                        it.a[0] = fyz[u][0];
                        it.a[1] = fyz[u][1];
                        it.a[2] = g[u][0];
                        it.a[3] = g[u][1];
                        *reinterpret_cast<Item<F> *>(&s_items[pos]) = it;
                        Item<F> it2;
                        it2.key = 0;
                        it2.fx = g[u][2];
                        it2.a[0] = g[u][3];
                        it2.a[1] = 0.0f; it2.a[2] = 0.0f; it2.a[3] = 0.0f;
                        *reinterpret_cast<Item<F> *>(&s_items[pos + 1]) = it2;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        s_bucket[pos + 1] = (uint8_t)ps[u][q].bucket;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < F; ++j) it.a[j] = g[u][j] * ps[u][q].wrest;
                    *reinterpret_cast<Item<F> *>(&s_items[pos]) = it;
                    s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                }
                }
            }
        }
    }
    if (wi == 0) {
        if (reserver) s_gbase[threadIdx.x - 64] = run_base;   // (waits for the reservation's return: the first use of it)
        __syncthreads();
    } else {
        lds_barrier();
    }
    const uint32_t in_window = (kScatterSplit == 1 || staged - w0 < (uint32_t)kStage) ? (staged - w0) : (uint32_t)kStage;
#ifndef ABL_NO_STORE   // (ablation builds: make variant ... EXTRA=-DABL_*; wrong results on purpose)
    for (uint32_t pos = threadIdx.x; pos < in_window; pos += kThreads) {
        const uint32_t b = s_bucket[pos];
        // write-once / read-once stream: non-temporal stores (measured -7 % on the whole backward)
        store_item_nt<STREAM>(items + s_gbase[b] + (w0 + pos - s_start[b]), s_items[pos]);
    }
#endif
    if (kScatterSplit > 1 && wi + 1u < n_win) lds_barrier();   // the buffer is refilled by the next window
    }
    // selective table zeroing, second half (zero_unowned_rows_kernel did the rows no bucket covers): a hashed bucket with
    // exactly one work unit is overwritten by the consume pass; one with none is never written and one with several is
    // added to atomically -- those are zeroed here, by the workgroups of the level's first tiles, behind their own item
    // stores (the consume pass runs after this kernel either way).
    if (zero_acc != nullptr && !dense) {
        for (uint32_t b = tile; b < bl.nb; b += plan.num_tiles) {
            const uint32_t gb = bl.bucket0 + b;
            if (unit_first[gb + 1] - unit_first[gb] == 1u) continue;
            const uint32_t row0 = b * bl.rows_pb;
            const uint32_t nrows = (bl.used - row0 < bl.rows_pb) ? (bl.used - row0) : bl.rows_pb;
            float *dst = zero_acc + ((int64_t)first_idx[lvl] + row0) * F;
            for (uint32_t e = threadIdx.x; e < nrows * (uint32_t)F; e += kThreads) dst[e] = 0.0f;
        }
    }
}

#ifdef CONSUME_TRACE
// Instrumented build only (make variant ... EXTRA=-DCONSUME_TRACE; tools/attic/consume_trace.py): per work unit, wave 0's clock
// (100 MHz) at the phase boundaries of consume_unit. Never compiled into the shipped library.
__device__ unsigned long long g_consume_trace[16384 * 8];
__device__ unsigned int g_consume_trace_n;
#define CTRACE(k) do { if (threadIdx.x == 0) ctr[k] = wall_clock64(); } while (0)
#else
#define CTRACE(k) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------------- pass C
// One work unit (a bucket, or a chunk of an over-full one) on the calling workgroup. Every item format streams the same way
// (`stream` below): the first round of 16-byte loads goes out BEFORE the image is zeroed, so the zeroing and its barrier sit
// inside the loads' latency instead of in front of it; `hook()` runs once after the first round has been added (the
// persistent kernel fetches its next unit's descriptor there). The fixed-point scale comes with the descriptor.
// (FX also selects the item loads' cache policy: fixed-point images <=> batches from 2^17 samples <=> streaming loads; the
// fp64 form of small batches reads its items with plain loads -- the scatter pass wrote them with plain stores, see
// store_item_nt)
template <int F, bool FX, int FMT, class Hook>
__device__ __forceinline__ void consume_unit(const LevelTable &lt, const BinPlan &plan, const int32_t *__restrict__ first_idx,
                                             const UnitDesc d, const typename ItemSel<F, FMT>::type *__restrict__ items,
                                             float *__restrict__ grad_table, int force_atomic, int headroom, double *s_acc,
                                             Hook &&hook, __half *__restrict__ half_out = nullptr) {
    constexpr bool H = FMT == 1;      // half-precision stream (fp16 tables)
    constexpr bool P12 = FMT == 2;    // 12-byte units (Item12)
    // (uniform over the workgroup -- every thread copied the same descriptor out of LDS; as scalars the level's record comes
    // out of the kernel arguments by scalar loads instead of one vector load per field)
    const uint32_t gb = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.bucket);
    const uint32_t lvl = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.level);
    const BinLevel bl = plan.lv[lvl];
    const uint32_t b = gb - bl.bucket0;
    const uint32_t r1 = (uint32_t)lt.res[lvl];
    // compact levels: the image starts at the bucket's first base plane and includes one halo plane
    const bool two_d = plan.pairs == 2u;
    const uint32_t row0 = bl.compact ? b * bl.slab * (two_d ? r1 : r1 * r1) : b * bl.rows_pb;
    const uint32_t nrows = (bl.used - row0 < bl.rows_pb) ? (bl.used - row0) : bl.rows_pb;

    FxScale fx{1.0, 1.0, false};
    if constexpr (FX) fx = fx_scale_of(d.gmax_bits, headroom);
    unsigned long long *s_fix = reinterpret_cast<unsigned long long *>(s_acc);
#ifdef CONSUME_TRACE
    unsigned long long ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    struct TraceOut {
        unsigned long long *c; UnitDesc dd;
        __device__ ~TraceOut() {
            if (threadIdx.x == 0) {
                c[5] = wall_clock64();
                const unsigned int slot = atomicAdd(&g_consume_trace_n, 1u);
                if (slot < 16384u) {
                    unsigned long long *o = g_consume_trace + (size_t)slot * 8;
                    o[0] = ((unsigned long long)dd.level << 32) | dd.bucket;
                    o[1] = dd.end - dd.begin;
                    for (int k = 0; k < 6; ++k) o[2 + k] = c[k];
                }
            }
        }
    } trace_out{ctr, d};
#endif
    CTRACE(0);
    auto zero_image = [&]() {
        for (uint32_t e = threadIdx.x; e < nrows * F; e += kConsumeThreads) s_acc[e] = 0.0;   // all-zero bits either way
        lds_barrier();
    };
    const uint64_t begin = d.begin, end = d.end;
    // load(p0): the thread's round of items starting at p0 (elements past the unit's end come back with no valid corner);
    // proc(): add the round to the image
    auto stream = [&](uint64_t first, uint64_t lim, uint64_t stride, auto &&load, auto &&proc) {
        uint64_t p0 = first;
        load(p0);
        CTRACE(1);      // first round issued
        zero_image();
        CTRACE(2);      // image zeroed (barrier passed)
        proc();
        CTRACE(3);      // first round added (its loads have arrived)
        hook();
        for (p0 += stride; p0 < lim; p0 += stride) {
            load(p0);
            proc();
        }
        CTRACE(4);      // every round of this wave added
    };
    const int rot = (int)(threadIdx.x & (F - 1));
    // feature order rotated by lane: the F slots of a row are consecutive 8-byte words, so with every lane adding feature j
    // in the same instruction only 1 / F of the LDS banks were addressed (half of the pass's LDS cycles were bank conflicts)
    // `vr` = the gradient vector ALREADY rotated by the lane's `rot` (rotate_features, once per item): step jj adds feature
    // (jj + rot) mod F
    auto add_row = [&](uint32_t row, const float (&vr)[F], float w) {
#pragma unroll
        for (int jj = 0; jj < F; ++jj) {
            const int j = (jj + rot) & (F - 1);
            const float c = vr[jj] * w;
            if (FX && fx.fixed) atomicAdd(&s_fix[row * F + j], fx_encode(c, fx.scale));
            else atomicAdd(&s_acc[row * F + j], (double)c);
        }
    };
#ifdef ABL_CONSUME_STREAM_ONLY   // ablation: the item loads alone (everything loaded is folded into one word so nothing is dropped)
    uint32_t abl_sink = 0;
#endif
    auto add_pair = [&](uint32_t ra, uint32_t rb, bool va, bool vb, float fxv, const float (&a)[F]) {
#ifdef ABL_CONSUME_STREAM_ONLY
        abl_sink ^= ra ^ rb ^ (uint32_t)va ^ (uint32_t)vb ^ __float_as_uint(fxv) ^ __float_as_uint(a[0]) ^ __float_as_uint(a[F - 1]);
        if (abl_sink == 0x12345678u) s_fix[ra] = abl_sink;
        return;
#endif
        float ar[F];
        rotate_features<F>(a, rot, ar);
        if (va) add_row(ra, ar, 1.0f - fxv);
        if (vb) add_row(rb, ar, fxv);
    };
    const uint32_t r2 = r1 * r1;
    auto add_compact = [&](uint32_t base_row, float fxx, float fyy, float fzz, const float (&gg)[F]) {
        const float gxx = 1.0f - fxx, gyy = 1.0f - fyy, gzz = 1.0f - fzz;
        const float wxy[4] = {gxx * gyy, gxx * fyy, fxx * gyy, fxx * fyy};   // reference order: (x * y) * z
        float gr[F];
        rotate_features<F>(gg, rot, gr);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t row = base_row + ((c >> 2) & 1) + ((c >> 1) & 1) * r1 + (c & 1) * r2;
            if (row >= nrows) continue;   // cannot happen for in-range cells; keeps the image safe
            add_row(row, gr, wxy[c >> 1] * ((c & 1) ? fzz : gzz));
        }
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if constexpr ((F == 2 || F == 4) && !H && !P12) {
        if (bl.compact && two_d) {
            // 2-D: one sample per item {local base row | valid | fx, fy (25-bit fixed point), g[F]}; 4 corners
            const Item<F> *itf = reinterpret_cast<const Item<F> *>(items);
            constexpr int UD = 8;
            Item<F> it[UD];
            stream(begin + threadIdx.x, end, (uint64_t)kConsumeThreads * UD,
                   [&](uint64_t p0) {
#pragma unroll
                       for (int u = 0; u < UD; ++u) {
                           const uint64_t pp = p0 + (uint64_t)u * kConsumeThreads;
                           if (pp < end) it[u] = load_item_nt<F, FX>(itf + pp);
                           else it[u].key = 0;
                       }
                   },
                   [&]() {
#pragma unroll
                       for (int u = 0; u < UD; ++u) {
                           if (!(it[u].key & (1u << 13))) continue;
                           uint32_t local, qx, qy;
                           unpack_compact2d(it[u].key, __float_as_uint(it[u].fx), local, qx, qy);
                           const float q = 1.0f / 33554432.0f;
                           const float fxx = (float)qx * q, fyy = (float)qy * q;
                           const float gxx = 1.0f - fxx, gyy = 1.0f - fyy;
                           float ar[F];
                           rotate_features<F>(it[u].a, rot, ar);
                           // a zero fraction = a corner of weight 0 (it lies outside the level when the coordinate was clamped
                           // onto the last line / column): skipped, like the pair items' validity bits
                           if (local + r1 + 1u < nrows) {
                               add_row(local, ar, gxx * gyy);
                               if (qx) add_row(local + 1u, ar, fxx * gyy);
                               if (qy) {
                                   add_row(local + r1, ar, gxx * fyy);
                                   if (qx) add_row(local + r1 + 1u, ar, fxx * fyy);
                               }
                           } else {   // the level's last line: only rows inside the image
                               if (local < nrows) add_row(local, ar, gxx * gyy);
                               if (qx && local + 1u < nrows) add_row(local + 1u, ar, fxx * gyy);
                               if (qy && local + r1 < nrows) add_row(local + r1, ar, gxx * fyy);
                           }
                       }
                   });
            lds_barrier();
            const int64_t grow0d = (int64_t)first_idx[lvl] + row0;
            for (uint32_t e = threadIdx.x; e < nrows * F; e += kConsumeThreads) {
                const int64_t grow = grow0d + e / F;
                if ((uint64_t)grow >= (uint64_t)lt.table_rows) continue;
                const float v = (FX && fx.fixed) ? fx_decode(s_fix[e], fx.inv) : (float)s_acc[e];
                if (v != 0.0f) unsafeAtomicAdd(grad_table + grow * F + (e % F), v);
            }
            return;
        }
    }
    if constexpr (F == 2 || F == 4) {
        if (bl.compact) {
            constexpr int UC = 2;
            const uint64_t first = begin + 2ull * threadIdx.x, stride = 2ull * kConsumeThreads * UC;
            if constexpr (P12) {
                // one sample per two 12-byte units {local base row | valid, fx, fy} {fz, g0, g1}
                // (loads unconditional from a clamped unit pair, validity kept beside them: a load under `if (p + 1 < end)` with the
                // key cleared in the else branch is closed by s_waitcnt vmcnt(0) -- every load of the round waited for on its own)
                Item12 ia[UC], ib[UC];
                uint64_t pcur = 0;
                stream(first, end, stride,
                       [&](uint64_t p0) {
                           pcur = p0;
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               const uint64_t p = p0 + 2ull * u * kConsumeThreads;
                               const uint64_t pc = (p + 1 < end) ? p : begin;
                               ia[u] = load_item12_nt(items + pc);
                               ib[u] = load_item12_nt(items + pc + 1);
                           }
                       },
                       [&]() {
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               if (pcur + 2ull * u * kConsumeThreads + 1 >= end) continue;
                               if (!(ia[u].w[0] & (1u << 26))) continue;
                               const float gg[F] = {__uint_as_float(ib[u].w[1]), __uint_as_float(ib[u].w[2])};
                               add_compact(ia[u].w[0] & 0x1FFFu, __uint_as_float(ia[u].w[1]), __uint_as_float(ia[u].w[2]),
                                           __uint_as_float(ib[u].w[0]), gg);
                           }
                       });
            } else if constexpr (H && F == 4) {
                // one sample per two 16-byte units: {local base row | valid, fx, fy, fz (fp32)} {half2 g01, half2 g23, -, -}
                u32x4 va[UC], vb2[UC];
                stream(first, end, stride,
                       [&](uint64_t p0) {
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               const uint64_t p = p0 + 2ull * u * kConsumeThreads;
                               va[u] = u32x4{0u, 0u, 0u, 0u};
                               vb2[u] = va[u];
                               if (p + 1 < end) {
                                   va[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + p));
                                   vb2[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + p + 1));
                               }
                           }
                       },
                       [&]() {
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               if (!(va[u][0] & (1u << 26))) continue;
                               const float2 g01 = half2_bits_to_float2(vb2[u][2]);
                               const float2 g23 = half2_bits_to_float2(vb2[u][3]);
                               const float gg[F] = {g01.x, g01.y, g23.x, g23.y};
                               add_compact(va[u][0] & 0x1FFFu, __uint_as_float(va[u][1]), __uint_as_float(va[u][2]),
                                           __uint_as_float(va[u][3]), gg);
                           }
                       });
            } else if constexpr (H) {
                // one sample per 16-byte record (two 8-byte units): {local base row | valid, fx, fy, fz (u16), half2 g}
                ItemHC rec[UC];
                stream(first, end, stride,
                       [&](uint64_t p0) {
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               const uint64_t p = p0 + 2ull * u * kConsumeThreads;
                               rec[u].key = 0;
                               if (p + 1 < end) {
                                   const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + p));
                                   __builtin_memcpy(&rec[u], &v, 16);
                               }
                           }
                       },
                       [&]() {
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               if (!(rec[u].key & (1u << 26))) continue;
                               const float2 gf = __half22float2(rec[u].g);
                               const float gg[F] = {gf.x, gf.y};
                               const float q = 1.0f / 65536.0f;
                               add_compact(rec[u].key & 0x1FFFu, ((float)rec[u].fx + 0.5f) * q, ((float)rec[u].fy + 0.5f) * q,
                                           ((float)rec[u].fz + 0.5f) * q, gg);
                           }
                       });
            } else {
                // one sample per two slots: F = 2 {local base row | valid, fx, fy, fz} {-, g0, g1, -}; F = 4 {.., fx, fy, fz,
                // g0, g1} {-, g2, g3, ...}; all 8 corners land here
                const Item<F> *itf = reinterpret_cast<const Item<F> *>(items);
                Item<F> ia[UC], ib[UC];
                stream(first, end, stride,
                       [&](uint64_t p0) {
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               const uint64_t p = p0 + 2ull * u * kConsumeThreads;
                               if constexpr (sizeof(Item<F>) == 16) {   // (one load per slot: the conditional form batches fine, and
                                                                        // measured 2.6 % better on S1 than the clamped one)
                                   if (p + 1 < end) {
                                       ia[u] = load_item_nt<F, FX>(itf + p);
                                       ib[u] = load_item_nt<F, FX>(itf + p + 1);
                                   } else {
                                       ia[u].key = 0;
                                   }
                               } else {   // 24-byte slots: unconditional from a clamped slot pair, masked below (see the pair items)
                                   const uint64_t pc = (p + 1 < end) ? p : begin;
                                   ia[u] = load_item_nt<F, FX>(itf + pc);
                                   ib[u] = load_item_nt<F, FX>(itf + pc + 1);
                               }
                           }
                           if constexpr (sizeof(Item<F>) != 16) {
#pragma unroll
                               for (int u = 0; u < UC; ++u)
                                   if (p0 + 2ull * u * kConsumeThreads + 1 >= end) ia[u].key = 0;
                           }
                       },
                       [&]() {
#pragma unroll
                           for (int u = 0; u < UC; ++u) {
                               if (!(ia[u].key & (1u << 26))) continue;
                               float gg[F];
                               if constexpr (F == 2) {
                                   gg[0] = ib[u].fx; gg[1] = ib[u].a[0];
                               } else {
                                   gg[0] = ia[u].a[2]; gg[1] = ia[u].a[3]; gg[2] = ib[u].fx; gg[3] = ib[u].a[0];
                               }
                               add_compact(ia[u].key & 0x1FFFu, ia[u].fx, ia[u].a[0], ia[u].a[1], gg);
                           }
                       });
            }
            lds_barrier();
            // neighbouring buckets share their boundary plane: everything is added atomically (the table is zeroed)
            const int64_t grow0c = (int64_t)first_idx[lvl] + row0;
            for (uint32_t e = threadIdx.x; e < nrows * F; e += kConsumeThreads) {
                const int64_t grow = grow0c + e / F;
                if ((uint64_t)grow >= (uint64_t)lt.table_rows) continue;
                const float v = (FX && fx.fixed) ? fx_decode(s_fix[e], fx.inv) : (float)s_acc[e];
                if (v != 0.0f) unsafeAtomicAdd(grad_table + grow * F + (e % F), v);
            }
            return;
        }
    }
    constexpr int UN = 8;  // 16-byte loads in flight per thread (4: -1 %, 16: +3 % with the fixed-point atomics)
    if constexpr (P12) {
        Item12 it[UN];
        uint64_t pcur = 0;
        stream(begin + threadIdx.x, end, (uint64_t)kConsumeThreads * UN,
               [&](uint64_t p0) {
                   pcur = p0;
#pragma unroll
                   for (int u = 0; u < UN; ++u) {   // unconditional, clamped; validity checked at use (see the compact form)
                       const uint64_t pp = p0 + (uint64_t)u * kConsumeThreads;
                       it[u] = load_item12_nt(items + (pp < end ? pp : end - 1));
                   }
               },
               [&]() {
#pragma unroll
                   for (int u = 0; u < UN; ++u) {
                       uint32_t ra, rb;
                       bool va, vb;
                       float fxv, a[F];
                       unpack_item12(it[u].w, ra, rb, va, vb, fxv, a);
                       const bool in = pcur + (uint64_t)u * kConsumeThreads < end;
                       add_pair(ra, rb, va && in, vb && in, fxv, a);
                   }
               });
    } else if constexpr (H && F == 4) {
        u32x4 v[UN];
        stream(begin + threadIdx.x, end, (uint64_t)kConsumeThreads * UN,
               [&](uint64_t p0) {
#pragma unroll
                   for (int u = 0; u < UN; ++u) {
                       const uint64_t pp = p0 + (uint64_t)u * kConsumeThreads;
                       v[u] = u32x4{0u, 0u, 0u, 0u};
                       if (pp < end) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + pp));
                   }
               },
               [&]() {
#pragma unroll
                   for (int u = 0; u < UN; ++u) {
                       const float2 a01 = half2_bits_to_float2(v[u][2]);
                       const float2 a23 = half2_bits_to_float2(v[u][3]);
                       const float a[F] = {a01.x, a01.y, a23.x, a23.y};
                       add_pair(v[u][0] & 0x1FFFu, (v[u][0] >> 13) & 0x1FFFu, (v[u][0] >> 26) & 1u, (v[u][0] >> 27) & 1u,
                                __uint_as_float(v[u][1]), a);
                   }
               });
    } else if constexpr (H) {
        // 8-byte items read two at a time (16-byte loads from even unit indices); a unit's odd first / last item goes alone
        auto consume8 = [&](uint32_t key, uint32_t payload) {
            uint32_t ra, rb;
            bool va, vb;
            float fxv;
            unpack_half_key(key, ra, rb, va, vb, fxv);
            __half2 h;
            __builtin_memcpy(&h, &payload, 4);
            const float2 af = __half22float2(h);
            const float a[F] = {af.x, af.y};
            add_pair(ra, rb, va, vb, fxv, a);
        };
        uint64_t p = begin;
        const bool head = (p & 1ull) && p < end;
        if (head) ++p;
        const uint64_t even_end = end & ~1ull;
        u32x4 v[UN];
        stream(p + 2ull * threadIdx.x, even_end, 2ull * kConsumeThreads * UN,
               [&](uint64_t p0) {
#pragma unroll
                   for (int u = 0; u < UN; ++u) {
                       const uint64_t q = p0 + 2ull * u * kConsumeThreads;
                       v[u] = u32x4{0u, 0u, 0u, 0u};   // key 0: no valid corner
                       if (q < even_end) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + q));
                   }
               },
               [&]() {
#pragma unroll
                   for (int u = 0; u < UN; ++u) {
                       consume8(v[u][0], v[u][1]);
                       consume8(v[u][2], v[u][3]);
                   }
               });
        if (head && threadIdx.x == 0) {
            const u32x2 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(items + begin));
            consume8(v1[0], v1[1]);
        }
        if ((end & 1ull) && end - 1 >= p && threadIdx.x == 64) {
            const u32x2 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(items + end - 1));
            consume8(v1[0], v1[1]);
        }
    } else {
        const Item<F> *itf = reinterpret_cast<const Item<F> *>(items);
        Item<F> it[UN];
        stream(begin + threadIdx.x, end, (uint64_t)kConsumeThreads * UN,
               [&](uint64_t p0) {
                   if constexpr (sizeof(Item<F>) == 16) {
#pragma unroll
                       for (int u = 0; u < UN; ++u) {
                           const uint64_t pp = p0 + (uint64_t)u * kConsumeThreads;
                           if (pp < end) it[u] = load_item_nt<F, FX>(itf + pp);
                           else it[u].key = 0;
                       }
                   } else {
                       // 24-byte items are two loads each: under `if (pp < end)` the compiler waited for every item before
                       // it issued the next one (ISA, round 4). Unconditional loads from a clamped index, masked afterwards.
#pragma unroll
                       for (int u = 0; u < UN; ++u) {
                           const uint64_t pp = p0 + (uint64_t)u * kConsumeThreads;
                           it[u] = load_item_nt<F, FX>(itf + (pp < end ? pp : end - 1));
                       }
#pragma unroll
                       for (int u = 0; u < UN; ++u)
                           if (p0 + (uint64_t)u * kConsumeThreads >= end) it[u].key = 0;
                   }
               },
               [&]() {
#pragma unroll
                   for (int u = 0; u < UN; ++u)
                       add_pair(it[u].key & 0x1FFFu, (it[u].key >> 13) & 0x1FFFu, (it[u].key >> 26) & 1u,
                                (it[u].key >> 27) & 1u, it[u].fx, it[u].a);
               });
    }
    lds_barrier();

    const bool single = d.single != 0 && !force_atomic;
    const int64_t grow0 = (int64_t)first_idx[lvl] + row0;
    // Single-unit buckets overwrite their rows: 16-byte stores of four consecutive elements per lane (round 4: the
    // one-dword-per-lane form below cost 37 us of S1's backward -- 44 MB at 1.2 TB/s -- and 7-11 us of the 65 536- / 8 192-sample
    // calls; ablation in profiles/r04_experiments.md)
    {
        float *dst0 = grad_table + grow0 * F;
        const uint32_t nelem = nrows * F;
        if (single && grow0 + (int64_t)nrows <= lt.table_rows) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            auto value = [&](uint32_t e) { return (FX && fx.fixed) ? fx_decode(s_fix[e], fx.inv) : (float)s_acc[e]; };
            if constexpr (H) {
                if (half_out != nullptr) {
                    // fp16 table, single-unit bucket: the rounded sums go straight into the caller's half table (round 4)
                    // instead of through the fp32 accumulation image and the conversion pass behind it (which skips this
                    // bucket: f32_to_f16_skip_kernel). 8-byte stores of four consecutive halves, scalar head / tail.
                    __half *h0 = half_out + grow0 * F;
                    uint32_t hh = (uint32_t)(((8u - (uint32_t)(reinterpret_cast<uintptr_t>(h0) & 7u)) & 7u) / 2u);
                    if (hh > nelem) hh = nelem;
                    const uint32_t hb = (nelem - hh) & ~3u;
                    if (threadIdx.x < hh) h0[threadIdx.x] = __float2half_rn(value(threadIdx.x));
                    for (uint32_t e = hh + threadIdx.x * 4u; e < hh + hb; e += kConsumeThreads * 4u) {
                        const __half2 lo = __floats2half2_rn(value(e), value(e + 1));
                        const __half2 hi2 = __floats2half2_rn(value(e + 2), value(e + 3));
                        uint2 pk;
                        __builtin_memcpy(&pk.x, &lo, 4);
                        __builtin_memcpy(&pk.y, &hi2, 4);
                        *reinterpret_cast<uint2 *>(h0 + e) = pk;
                    }
                    if (hh + hb + threadIdx.x < nelem) h0[hh + hb + threadIdx.x] = __float2half_rn(value(hh + hb + threadIdx.x));
                    return;
                }
            }
            // elements in front of the first 16-byte boundary (a level may start on an odd row: the dense levels in front of
            // it have odd sizes), whole vectors, then the elements behind the last one
            uint32_t head = (uint32_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst0) & 15u)) & 15u) / 4u);
            if (head > nelem) head = nelem;
            const uint32_t body = (nelem - head) & ~3u;
            if (threadIdx.x < head) dst0[threadIdx.x] = value(threadIdx.x);
            for (uint32_t e = head + threadIdx.x * 4u; e < head + body; e += kConsumeThreads * 4u) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = value(e + k);
                *reinterpret_cast<f32x4 *>(dst0 + e) = v;
            }
            if (head + body + threadIdx.x < nelem) dst0[head + body + threadIdx.x] = value(head + body + threadIdx.x);
            return;
        }
    }
    for (uint32_t e = threadIdx.x; e < nrows * F; e += kConsumeThreads) {
        const int64_t grow = grow0 + e / F;
        if ((uint64_t)grow >= (uint64_t)lt.table_rows) continue;
        const float v = (FX && fx.fixed) ? fx_decode(s_fix[e], fx.inv) : (float)s_acc[e];
        float *dst = grad_table + grow * F + (e % F);
        if constexpr (H) {
            if (single && half_out != nullptr) {   // (a table shorter than this bucket: element-wise, same destination rule)
                half_out[grow * F + (e % F)] = __float2half_rn(v);
                continue;
            }
        }
        if (single) *dst = v;
        else if (v != 0.0f) unsafeAtomicAdd(dst, v);
    }
}

// Persistent form: `work_counter` non-NULL -> every workgroup keeps fetching units from it until they run out (grid = the
// number of workgroups the chip holds, not the number of units). A unit of a small batch is ~10 us of work between a launch,
// a 128 KiB image to zero and a flush whose stores s_endpgm would wait for: as separate workgroups (one per CU at a time)
// nerf_lego.yaml's 1 800 units of 8 K items took 228 us; here the flush of unit k drains behind unit k + 1 (all barriers in
// consume_unit are LDS-only). The fetch is pipelined one unit ahead: thread 0 draws the NEXT unit's number when a unit
// starts and loads its descriptor once the first round of items is in (consume_unit's hook), so that neither the returning
// atomic nor the descriptor load stands between two units. `work_counter` NULL: one unit per workgroup (small batches).
template <int F, bool FX, int FMT>
__global__ __launch_bounds__(kConsumeThreads) void bin_consume_kernel(LevelTable lt, BinPlan plan,
                                                                      const int32_t *__restrict__ first_idx,
                                                                      const uint32_t *__restrict__ unit_first,
                                                                      const UnitDesc *__restrict__ unit_desc,
                                                                      const typename ItemSel<F, FMT>::type *__restrict__ items,
                                                                      float *__restrict__ grad_table,
                                                                      int force_atomic, int headroom,
                                                                      uint32_t *__restrict__ work_counter,
                                                                      __half *__restrict__ half_out = nullptr) {
    extern __shared__ double s_acc[];  // [rows_pb][F]: fp64, or 64-bit fixed point (same size)
    __shared__ UnitDesc s_desc;
    __shared__ uint32_t s_unit;
    if (work_counter == nullptr) {
        // (the descriptor array is sized for the grid: the load needs no bound check and goes out with the unit count's)
        const UnitDesc d = unit_desc[blockIdx.x];
        const uint32_t unit_end = unit_first[plan.total_buckets];
        if (blockIdx.x >= unit_end) return;
        consume_unit<F, FX, FMT>(lt, plan, first_idx, d, items, grad_table, force_atomic, headroom, s_acc, []() {}, half_out);
        return;
    }
    const uint32_t unit_end = unit_first[plan.total_buckets];
    if (unit_end == 0u) return;
    uint32_t nxt = 0;          // thread 0: the unit after the current one
    __shared__ UnitDesc s_desc_nxt;   // its descriptor (in LDS: as a local of thread 0 it lived in scratch memory)
    if (threadIdx.x == 0) {
        const uint32_t u = atomicAdd(work_counter, 1u);
        s_unit = u;
        s_desc = unit_desc[u < unit_end ? u : unit_end - 1u];
    }
    lds_barrier();
    for (;;) {
        const uint32_t unit = s_unit;
        if (unit >= unit_end) return;
        const UnitDesc d = s_desc;
        if (threadIdx.x == 0) nxt = atomicAdd(work_counter, 1u);
        consume_unit<F, FX, FMT>(lt, plan, first_idx, d, items, grad_table, force_atomic, headroom, s_acc, [&]() {
            if (threadIdx.x == 0) s_desc_nxt = unit_desc[nxt < unit_end ? nxt : unit_end - 1u];
        }, half_out);
        // (every thread read s_unit / s_desc before the barriers inside consume_unit: thread 0 may overwrite them now)
        if (threadIdx.x == 0) {
            s_unit = nxt;
            s_desc = s_desc_nxt;
        }
        lds_barrier();   // the next unit is known and the image is free again; the flush stores keep draining
    }
}

// fp16 tables: fp32 accumulation image -> the caller's half table, MINUS the row-partitioning buckets with exactly one work unit, whose
// rows the consume pass wrote into the half table itself. grid (x, num_lods): level l = rows [first_idx[l], first_idx[l + 1])
// (level 0 from row 0, the last one to the end of the table), walked in chunks of one bucket (hashed binned levels) or 4 096 rows.
__global__ __launch_bounds__(256) void f32_to_f16_skip_kernel(const float *__restrict__ acc, __half *__restrict__ dst,
                                                             const int32_t *__restrict__ first_idx, LevelTable lt,
                                                             BinPlan plan, const uint32_t *__restrict__ unit_first, int F) {
    const int l = (int)blockIdx.y;
    const int64_t lo = (l == 0) ? 0 : (int64_t)first_idx[l];
    const int64_t hi = (l + 1 < lt.num_lods) ? (int64_t)first_idx[l + 1] : lt.table_rows;
    const BinLevel bl = plan.lv[l];
    // buckets that partition the level's rows (hashed levels: 2^k consecutive rows; dense levels binned by whole x-lines): a
    // single-unit one is overwritten by the consume pass. Compact levels share halo planes and direct levels have no buckets.
    const bool hashed_binned = bl.bucket0 != 0xFFFFFFFFu && bl.nb > 0 && bl.compact == 0;
    const int64_t chunk = hashed_binned ? (int64_t)bl.rows_pb : 4096;
    const int64_t level0 = (int64_t)first_idx[l];   // buckets count from the level's first row
    for (int64_t c = blockIdx.x;; c += gridDim.x) {
        // chunk c of the level proper; rows in front of first_idx[0] (l == 0, lo < level0) ride with chunk 0
        int64_t r0 = level0 + c * chunk, r1 = r0 + chunk;
        if (c == 0 && lo < r0) r0 = lo;
        if (r0 >= hi) break;
        if (r1 > hi) r1 = hi;
        if (hashed_binned && c < (int64_t)bl.nb) {
            const uint32_t gb = bl.bucket0 + (uint32_t)c;
            if (unit_first[gb + 1] - unit_first[gb] == 1u) {
                // written by the consume pass -- exactly the bucket's rows INSIDE the level, [w0, w1): what the chunk holds
                // besides them (rows in front of the level's start that ride with chunk 0, padding rows behind a level whose
                // last bucket is partial) is still converted here, or the caller's `empty` half table would keep garbage there
                // (round-4 advisor finding: the fp32 path and the reference's zeros_like both leave zeros)
                const int64_t w0 = level0 + c * chunk;
                int64_t w1 = level0 + (int64_t)bl.used;
                if (w1 > w0 + chunk) w1 = w0 + chunk;
                if (w1 > hi) w1 = hi;
                for (int64_t e = r0 * F + threadIdx.x; e < w0 * F; e += 256) dst[e] = __float2half_rn(acc[e]);
                for (int64_t e = w1 * F + threadIdx.x; e < r1 * F; e += 256) dst[e] = __float2half_rn(acc[e]);
                continue;
            }
        }
        for (int64_t e = r0 * F + threadIdx.x; e < r1 * F; e += 256) dst[e] = __float2half_rn(acc[e]);
    }
}

// ------------------------------------------------------------------------------------------------- direct levels
// Levels whose whole (used) row range fits one LDS image need no partitioning at all: a workgroup keeps a private
// fp64 image of a GROUP of such levels, walks its share of the samples adding every corner with ds_add_f64, and
// adds the image to the (zeroed) gradient table with coalesced float atomics at the end.
// GT = float: gradients come from the transposed image gT [L][N][F]; otherwise (T = table scalar) straight from
// grad_output [N, L*F] -- used when no level needs binning, which makes the transposing pass unnecessary.
template <int DIM, int F, typename GT, bool TRANSPOSED, bool FX>
__global__ __launch_bounds__(kConsumeThreads) void direct_accumulate_kernel(LevelTable lt, BinPlan plan,
                                                                            const int32_t *__restrict__ first_idx,
                                                                            const float *__restrict__ coords,
                                                                            const GT *__restrict__ gT,
                                                                            float *__restrict__ grad_table,
                                                                            int64_t N, int64_t gpitch,
                                                                            const uint32_t *__restrict__ gmax,
                                                                            int headroom, int cstride = DIM) {
    constexpr int NC = 1 << DIM;
    extern __shared__ double s_acc[];
    __shared__ double s_scale[SHACIRA_MAX_LODS], s_inv[SHACIRA_MAX_LODS];
    __shared__ int s_all_fixed;
    const uint32_t grp = blockIdx.y;
    const uint32_t rows = plan.grows[grp];
    const uint32_t mask = plan.gmask[grp];
    __shared__ int s_lv[SHACIRA_MAX_LODS], s_nl;   // the group's levels
    if (threadIdx.x == 0) {
        s_all_fixed = 1;
        int n = 0;
        for (int l = 0; l < lt.num_lods; ++l)
            if ((mask >> l) & 1u) s_lv[n++] = l;
        s_nl = n;
    }
    for (uint32_t e = threadIdx.x; e < rows * F; e += kConsumeThreads) s_acc[e] = 0.0;
    __syncthreads();
    __shared__ float s_limit[SHACIRA_MAX_LODS];   // largest |gradient| the level's fixed-point scale is good for
    if constexpr (FX) {
        if (gmax != nullptr) {
            if ((int)threadIdx.x < lt.num_lods && ((mask >> threadIdx.x) & 1u)) {
                const FxScale f = fx_scale_of(gmax[threadIdx.x], headroom);
                s_scale[threadIdx.x] = f.scale;
                s_inv[threadIdx.x] = f.inv;
                s_limit[threadIdx.x] = __uint_as_float(0x7F800000u);   // the true maximum: every finite value fits
                if (!f.fixed) s_all_fixed = 0;     // one non-finite level: the whole group accumulates in fp64
            }
        } else {
            // No max |gradient| from a transposing pass (all-direct tables, small batches): the workgroup takes a PILOT
            // maximum per level over 1 024 of its own samples (one per thread, spread over its whole walk), and scales for
            // 2^12 times that. A contribution beyond the limit -- or non-finite -- goes straight to the table with a float
            // atomic (the image is flushed that way too), so the limit only has to be right for almost all of them; a level
            // whose pilot saw nothing but zeros keeps the fp64 image (with its group). Consequences to know: (1) the result is
            // bitwise reproducible (order-independent fixed-point sums) only while no gradient exceeds a workgroup's limit --
            // out-of-range contributions take float atomics in arrival order; (2) heavy-tailed or mostly-zero gradients (masked
            // pixels, a loss spike outside the pilot) send a larger share down that slow path: correct (tested), but slower.
            __shared__ uint32_t s_pm[SHACIRA_MAX_LODS];
            if ((int)threadIdx.x < SHACIRA_MAX_LODS) s_pm[threadIdx.x] = 0u;
            __syncthreads();
            const int64_t stride0 = (int64_t)gridDim.x * kConsumeThreads;
            const int64_t base = (int64_t)blockIdx.x * kConsumeThreads + threadIdx.x;
            const int64_t iters = (N + stride0 - 1) / stride0;
            int64_t ip = base + (int64_t)(threadIdx.x % (uint32_t)(iters > 0 ? iters : 1)) * stride0;
            if (ip >= N) ip = base;
            const int nlp = s_nl;
            const int64_t ipc = ip < N ? ip : N - 1;
            for (int q0 = 0; q0 < nlp; q0 += 4) {     // four levels' loads in flight
                float gp4[4][F];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int l = s_lv[(q0 + k < nlp) ? q0 + k : nlp - 1];
                    const GT *gp = TRANSPOSED ? gT + ((int64_t)l * gpitch + ipc) * F : gT + (ipc * lt.num_lods + l) * F;
#pragma unroll
                    for (int j = 0; j < F; ++j) gp4[k][j] = Scalar<GT>::load(gp + j);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float m = 0.0f;
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        const float v = fabsf(gp4[k][j]);
                        if (ip < N && v < __uint_as_float(0x7F800000u) && v > m) m = v;    // finite values only
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                    if (q0 + k < nlp && (threadIdx.x & 63) == 0 && m > 0.0f)
                        atomicMax(&s_pm[s_lv[q0 + k]], __float_as_uint(m));
                }
            }
            __syncthreads();
            if ((int)threadIdx.x < lt.num_lods && ((mask >> threadIdx.x) & 1u)) {
                const uint32_t bits = s_pm[threadIdx.x];
                if (bits == 0u) {
                    s_all_fixed = 0;
                } else {
                    uint32_t ef = ((bits >> 23) & 0xFFu) + 13u;     // limit = 2^(exponent + 13 - 127) >= 2^12 x pilot max
                    if (ef > 254u) ef = 254u;
                    const uint32_t lim_bits = ef << 23;
                    const FxScale f = fx_scale_of(lim_bits, headroom);   // every accepted |contribution| <= limit < 2^(ef - 126)
                    s_scale[threadIdx.x] = f.scale;
                    s_inv[threadIdx.x] = f.inv;
                    s_limit[threadIdx.x] = __uint_as_float(lim_bits);
                }
            }
        }
        __syncthreads();
    }
    const bool fixed = FX && s_all_fixed != 0;
    unsigned long long *s_fix = reinterpret_cast<unsigned long long *>(s_acc);
    const int64_t stride = (int64_t)gridDim.x * kConsumeThreads;
    const int rotd = (int)(threadIdx.x & (F - 1));
    // The walk is a chain of memory latencies at 16 waves per CU (one 128 KiB image per CU), so every load goes out as early
    // as it can: the NEXT sample's coordinates while this one is added, and the gradients of kLv levels of the group at a
    // time, unconditionally (the group's level list sits in LDS; slots past its end repeat the last level and are skipped).
    constexpr int kLv = 4;
    const int nl = s_nl;
    int64_t i = (int64_t)blockIdx.x * kConsumeThreads + threadIdx.x;
    float cn[DIM];
    {
        const int64_t ic = i < N ? i : N - 1;
#pragma unroll
        for (int a = 0; a < DIM; ++a) cn[a] = coords[ic * cstride + a];
    }
    for (; i < N; i += stride) {
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(cn[a]);
        {
            const int64_t in = (i + stride < N) ? i + stride : N - 1;
#pragma unroll
            for (int a = 0; a < DIM; ++a) cn[a] = coords[in * cstride + a];
        }
        for (int q0 = 0; q0 < nl; q0 += kLv) {
            float g[kLv][F];
#pragma unroll
            for (int k = 0; k < kLv; ++k) {
                const int l = s_lv[(q0 + k < nl) ? q0 + k : nl - 1];
                const GT *gp = TRANSPOSED ? gT + ((int64_t)l * gpitch + i) * F : gT + (i * lt.num_lods + l) * F;
#pragma unroll
                for (int j = 0; j < F; ++j) g[k][j] = Scalar<GT>::load(gp + j);
            }
#pragma unroll
            for (int k = 0; k < kLv; ++k) {
                if (q0 + k >= nl) continue;
                const int l = s_lv[q0 + k];
                const BinLevel bl = plan.lv[l];
                Corners<DIM> c;
                compute_corners<DIM>(t, lt.res[l], lt.hi[l], lt.dense[l] != 0, lt.mask, c);
                const double scale = FX ? s_scale[l] : 1.0;
                const float lim = FX ? s_limit[l] : 0.0f;
                bool in_range = true;
#pragma unroll
                for (int j = 0; j < F; ++j) in_range = in_range && (fabsf(g[k][j]) <= lim);
                float gr[F];   // g[k] in the lane's rotated feature order
                rotate_features<F>(g[k], rotd, gr);
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    if (c.row[kc] < bl.used) {
                        const size_t slot = (size_t)(bl.drow0 + c.row[kc]) * F;
                        if (fixed && in_range) {
#pragma unroll
                            for (int jj = 0; jj < F; ++jj) {   // feature order rotated by lane (LDS bank spreading)
                                const int j = (jj + rotd) & (F - 1);
                                atomicAdd(s_fix + slot + j, fx_encode(gr[jj] * c.w[kc], scale));
                            }
                        } else if (fixed) {
                            // a gradient beyond the pilot's limit, or non-finite (rare): that feature goes straight to the table
#pragma unroll
                            for (int j = 0; j < F; ++j) {
                                if (fabsf(g[k][j]) <= lim) {
                                    atomicAdd(s_fix + slot + j, fx_encode(g[k][j] * c.w[kc], scale));
                                } else {
                                    const int64_t grow = (int64_t)first_idx[l] + c.row[kc];
                                    if ((uint64_t)grow < (uint64_t)lt.table_rows)
                                        unsafeAtomicAdd(grad_table + grow * F + j, g[k][j] * c.w[kc]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int jj = 0; jj < F; ++jj) {
                                const int j = (jj + rotd) & (F - 1);
                                atomicAdd(s_acc + slot + j, (double)(gr[jj] * c.w[kc]));
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int l = 0; l < lt.num_lods; ++l) {
        if (!((mask >> l) & 1u)) continue;
        const BinLevel bl = plan.lv[l];
        const int64_t grow0 = (int64_t)first_idx[l];
        for (uint32_t e = threadIdx.x; e < bl.used * F; e += kConsumeThreads) {
            const int64_t grow = grow0 + e / F;
            if ((uint64_t)grow >= (uint64_t)lt.table_rows) continue;
            const float v = fixed ? fx_decode(s_fix[(size_t)bl.drow0 * F + e], s_inv[l])
                                  : (float)s_acc[(size_t)bl.drow0 * F + e];
            if (v != 0.0f) unsafeAtomicAdd(grad_table + grow * F + (e % F), v);
        }
    }
}


}  // namespace shacira
