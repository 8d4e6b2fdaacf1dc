// bwd_bin_front.h -- the front of the binned backward: transpose (+ fused bucket counting), standalone counting, bucket scan, table zeroing
// Part of the translation unit hashgrid_bwd_bin.hip (included there, in this order: bwd_bin_types.h, bwd_bin_front.h,
// bwd_bin_passes.h); see that file's header for the pipeline.
#pragma once

#include "bwd_bin_types.h"

namespace shacira {

// ------------------------------------------------------------------------------------------------- pass T
// grad_output [N, L*F] (T) -> gT [L][NP][F] fp32 (NP = N rounded up to even), through LDS, F scalars per lane per access.
// Block: 256 samples. Generic fallback: rows that are not whole 16-byte vectors (odd level counts), unaligned input.
template <typename T, int F, bool GMAX>
__global__ __launch_bounds__(256) void transpose_grad_kernel(const T *__restrict__ go, float *__restrict__ gT,
                                                             int64_t N, int64_t NP, int L, int lb, int le,
                                                             uint32_t *__restrict__ gmax) {
    __shared__ uint32_t s_max[SHACIRA_MAX_LODS];
    if (GMAX && threadIdx.x < SHACIRA_MAX_LODS) s_max[threadIdx.x] = 0;
    struct alignas(sizeof(T) * F) PieceIn { T v[F]; };
    struct alignas(sizeof(float) * F) PieceOut { float v[F]; };
    extern __shared__ __align__(16) unsigned char s_raw_g[];
    PieceOut *s_tile = reinterpret_cast<PieceOut *>(s_raw_g);  // [256][L + 1]
    const int pitch = L + 1;
    const int64_t s0 = (int64_t)blockIdx.x * 256;
    const int ns = (int)((N - s0 < 256) ? (N - s0) : 256);
    const PieceIn *in = reinterpret_cast<const PieceIn *>(go) + s0 * L;
    const int total = ns * L;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int sm = e / L, l = e - sm * L;
        const PieceIn p = in[e];
        PieceOut q;
#pragma unroll
        for (int j = 0; j < F; ++j) q.v[j] = Scalar<T>::load(&p.v[j]);
        s_tile[sm * pitch + l] = q;
    }
    __syncthreads();
    PieceOut *out = reinterpret_cast<PieceOut *>(gT);
    for (int l = lb; l < le; ++l) {
        uint32_t m = 0;
        if ((int)threadIdx.x < ns) {
            const PieceOut q = s_tile[threadIdx.x * pitch + l];
            if constexpr (GMAX) {
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const uint32_t b = __float_as_uint(fabsf(q.v[j]));
                    m = b > m ? b : m;
                }
            }
            float *dst = reinterpret_cast<float *>(out + (int64_t)l * NP + s0 + threadIdx.x);
            if constexpr (F == 2) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 v = {q.v[0], q.v[1]};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x2 *>(dst));
            } else {
#pragma unroll
                for (int j = 0; j < F; ++j) __builtin_nontemporal_store(q.v[j], dst + j);
            }
        }
        if constexpr (GMAX) {   // wave max -> one LDS atomic per wave and level
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(m, off, 64);
                m = o > m ? o : m;
            }
            if ((threadIdx.x & 63) == 0 && m) atomicMax(&s_max[l], m);
        }
    }
    if constexpr (GMAX) {
        __syncthreads();
        if ((int)threadIdx.x >= lb && (int)threadIdx.x < le && s_max[threadIdx.x])
            atomicMax(&gmax[threadIdx.x], s_max[threadIdx.x]);
    }
}

// 16-byte form of pass T, fused with pass A (round 3). The 8-byte kernel above ran at 2.5 TB/s, half the chip's copy rate
// (the guide prices 8-byte accesses at 0.54-0.70x the 16-byte rate), and the bucket counting ran as a separate kernel on a
// side stream that slowed it down further (105 us together on S1). Here a workgroup of 512 threads walks `rounds` tiles of
// TS samples: it issues every 16-byte load of a tile's gradient rows (K = 16 / (sizeof(T) * F) level pieces of one sample
// per lane), COUNTs the buckets of that tile's samples while the rows are in flight (corner hashing: pure ALU + LDS
// atomics; COUNT = true), parks the rows LEVEL-major in LDS and writes them out as 16-byte non-temporal vectors of
// M = 16 / (4 F) consecutive samples of one level. The staging image gT is [L][NP][F] with the level pitch NP = N rounded
// up to even, so that every vector is 16-byte aligned for any batch size. Bucket counts leave as ONE global atomic per
// (workgroup, non-empty bucket) into totals[] -- the per-(tile, bucket) matrix and its scan are gone: the scatter pass
// reserves its runs with returning atomics on per-bucket cursors instead.
constexpr int kFrontThreads = 512;
// SORTED (round 6): the batch's plan is at hand -- `sorted4` = its 16-byte records {x, y, z, sample index} in block order. The
// tile's samples are then records s0 .. s0 + TS: their gradient rows are GATHERED (row = the record's index; a row is whole
// 128-byte lines, so the gather moves the same bytes), the counting takes its coordinates from the records, and gT comes out
// in SORTED order -- every later pass (scatter, brick, direct) walks sorted samples with coalesced gradient loads.
template <int DIM, typename T, int F, bool GMAX, bool COUNT, bool SORTED = false>
__global__ __launch_bounds__(kFrontThreads) void front16_kernel(LevelTable lt, BinPlan plan, const T *__restrict__ go,
                                                                float *__restrict__ gT, const float *__restrict__ coords,
                                                                uint32_t *__restrict__ totals, uint32_t *__restrict__ cnt,
                                                                int64_t N, int64_t NP, int lb, int le, int ts_log2,
                                                                int rounds, uint32_t *__restrict__ gmax, uint32_t cps,
                                                                const float4 *__restrict__ sorted4 = nullptr) {
    constexpr int K = 16 / (int)(sizeof(T) * F);   // level pieces per 16-byte input vector
    constexpr int HE = SHACIRA_MAX_LODS * kMaxLevelBuckets / kFrontThreads;   // histogram words per thread (<= 8)
    constexpr int M = 16 / (4 * F);                // samples per 16-byte output vector
    constexpr int M_LOG2 = (M == 2) ? 1 : 0;
    constexpr int UL = 8;                          // 16-byte loads in flight per thread and round
    static_assert(K >= 1 && (M == 1 || M == 2), "16-byte transpose: F = 2 or 4");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    struct alignas(sizeof(float) * F) PieceOut { float v[F]; };
    __shared__ uint32_t s_max[SHACIRA_MAX_LODS];
    __shared__ uint32_t s_idx[SORTED ? kFrontThreads : 1];   // SORTED: row (= original sample index) of the tile's k-th record
    extern __shared__ __align__(16) unsigned char s_raw_g[];
    const int L = lt.num_lods;
    const int TS = 1 << ts_log2;                   // samples per tile: a power of two in [128, 512]
    const int pitch = TS + 2;                      // even: 16-byte LDS reads stay aligned; 2-way conflicts on the writes only
    PieceOut *s_tile = reinterpret_cast<PieceOut *>(s_raw_g);                                    // [L][TS + 2]
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_raw_g + (size_t)L * pitch * sizeof(PieceOut));   // [nbl][128]
    if (GMAX && threadIdx.x < SHACIRA_MAX_LODS) s_max[threadIdx.x] = 0;
    if constexpr (COUNT)
        for (uint32_t k = threadIdx.x; k < plan.nbl * (uint32_t)kMaxLevelBuckets; k += kFrontThreads) s_hist[k] = 0;
    __syncthreads();
    const int VPR = L / K;                         // input vectors per row
    const int nvec_log2 = ts_log2 - M_LOG2;        // output vectors per level and tile (a multiple of 64)
    // input vector e = tid + u * 512 of a tile belongs to sample e / VPR, level group e % VPR: divided once here, then
    // stepped (no integer division inside the rounds -- the kernel is bound by its vector ALU work, not by memory)
    const int q512 = kFrontThreads / VPR, r512 = kFrontThreads % VPR;
    const int sm_first = (int)threadIdx.x / VPR, v_first = (int)threadIdx.x % VPR;
    // counting: thread = (sample of the tile, level slot); tiles smaller than the workgroup split a sample's levels over
    // 512 / TS threads. A wave's threads share the slot (TS >= 128): readfirstlane keeps the level loop uniform.
    const int cslots = kFrontThreads >> ts_log2, csm = (int)threadIdx.x & (TS - 1);
    const int cslot = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> ts_log2);
    const int64_t tile0 = (int64_t)blockIdx.x * rounds;
    const int64_t tiles = (N + TS - 1) >> ts_log2;
    // the histogram is per ROUND: after each tile a thread moves its words (k = tid + j * 512 <-> (level k / 128, bucket
    // k % 128)) to the tile's row of cnt[tile][bucket] -- what lets the scatter pass reserve its runs before it has ranked
    // anything -- and keeps the workgroup's sums in registers for the totals
    uint32_t hsum[HE];
    uint32_t hacc[HE];   // items of the current SCATTER tile (cps counting tiles) so far: its runs are reserved in multiples of plan.pad
    // word k = tid + j * 512 <-> (level slot k / 128, bucket k % 128): the slot is uniform over a wave (128 = two waves), so
    // it is made a scalar and the two plan words come by scalar loads. (Round 4: with a lane-dependent index into the
    // by-value plan the compiler read the kernel arguments through VECTOR loads -- blevel[li] -> lv[..].nb -> bstart[li],
    // 3 dependent global loads x 8 words, each behind its own s_waitcnt: 24 serialised round trips at the start of
    // every workgroup, most of this kernel's time on the small NeRF batches.) The column is recomputed where it is used
    // (round 6: kept in an array it cost eight registers, the sorted form's margin to two workgroups per CU).
    static_assert(!COUNT || kFrontThreads % kMaxLevelBuckets == 0, "a wave maps to one level slot");
    const uint32_t hli0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kMaxLevelBuckets));
    const uint32_t hb = threadIdx.x % kMaxLevelBuckets;
    auto hcol_of = [&](int j) -> int {
        const uint32_t li = hli0 + (uint32_t)j * (kFrontThreads / kMaxLevelBuckets);
        const uint32_t lc = li < (uint32_t)SHACIRA_MAX_LODS ? li : 0u;
        const uint32_t nbq = li < plan.nbl ? plan.bnb[lc] : 0u, bsq = plan.bstart[lc];
        return (hb < nbq) ? (int)(bsq + hb) : -1;
    };
    if constexpr (COUNT) {
#pragma unroll
        for (int j = 0; j < HE; ++j) {
            hsum[j] = 0;
            hacc[j] = 0;
        }
    }
    // The counting thread's coordinates, loaded ONE ROUND AHEAD as a loop-carried value. (Round 4, ISA: loaded at the top of
    // the round the compiler sank them into the `csm < ns` block below -- behind the eight row loads -- and the counting began
    // with s_waitcnt vmcnt(0), i.e. after every row had arrived instead of while they were in flight. A value carried over
    // the back edge cannot be sunk; being older than the round's row loads it costs vmcnt(8), not vmcnt(0).)
    float cc[DIM];
    uint32_t cidx = 0;   // SORTED: the record's sample index, loaded with the coordinates
    auto load_coords = [&](int r) {
        int64_t t = tile0 + r;
        if (t >= tiles) t = tiles - 1;
        const int64_t s0r = t << ts_log2;
        const int nsr = (int)((N - s0r < TS) ? (N - s0r) : TS);
        const int64_t ci = s0r + (csm < nsr ? csm : nsr - 1);
        if constexpr (SORTED) {
            const float4 rc = sorted4[ci];
            cc[0] = rc.x;
            cc[1] = rc.y;
            if constexpr (DIM == 3) cc[2] = rc.z;
            cidx = __float_as_uint(rc.w);
        } else {
#pragma unroll
            for (int a = 0; a < DIM; ++a) cc[a] = coords[ci * DIM + a];
        }
    };
    if constexpr (COUNT || SORTED) {
        if (tile0 < tiles) load_coords(0);
    }
    for (int r = 0; r < rounds && tile0 + r < tiles; ++r) {
        const int64_t s0 = (tile0 + r) << ts_log2;
        const int ns = (int)((N - s0 < TS) ? (N - s0) : TS);
        const int total = ns * VPR;
        // row vectors: all loads unconditional with clamped indices (a branch around a load makes the compiler wait with
        // vmcnt(0) in front of the counting, which would serialise it with the row loads)
        // (the sample's coordinates are already in flight or in: loaded one round ahead, see the end of the loop body)
        const u32x4 *in = reinterpret_cast<const u32x4 *>(go) + (SORTED ? 0 : s0 * VPR);
        if constexpr (SORTED) {   // the tile's row numbers (the records arrived one round ahead, with the coordinates)
            if (cslot == 0) s_idx[csm] = cidx;
            lds_barrier();
        }
        // element e of the tile -> (sample e / VPR, vector e % VPR): stepped like the parking loop below
        auto src_of = [&](int e, int smq, int vq) -> const u32x4 * {
            if constexpr (SORTED) return in + (uint32_t)(s_idx[smq] * (uint32_t)VPR + (uint32_t)vq);   // (host: N * VPR < 2^32)
            else return in + e;
        };
        u32x4 raw[UL];
        {
            int smq = sm_first, vq = v_first;
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int e = (int)threadIdx.x + u * kFrontThreads;
                const bool in_tile = e < total;
                // idle lanes: one merged request (the tile's last element)
                raw[u] = __builtin_nontemporal_load(in_tile ? src_of(e, smq, vq) : src_of(total - 1, ns - 1, VPR - 1));
                smq += q512;
                vq += r512;
                if (vq >= VPR) { vq -= VPR; ++smq; }
            }
        }
        if constexpr (COUNT || SORTED) {
            double t[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) t[a] = axis_unit(cc[a]);
            load_coords(r + 1);   // next round's (clamped to the last tile): in flight behind this round's rows
            if (COUNT && csm < ns) {
#pragma unroll 2
                for (uint32_t li = (uint32_t)cslot; li < plan.nbl; li += (uint32_t)cslots)
                    count_level<DIM>(t, plan.cl[li], lt.mask, s_hist + li * kMaxLevelBuckets);
            }
        }
        // rows -> LEVEL-major LDS image
        auto park1 = [&](const u32x4 &rv, int sm, int v) {
            T tv[K * F];
            __builtin_memcpy(tv, &rv, 16);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                PieceOut q;
#pragma unroll
                for (int j = 0; j < F; ++j) q.v[j] = Scalar<T>::load(&tv[k * F + j]);
                s_tile[(v * K + k) * pitch + sm] = q;
            }
        };
        int sm = sm_first, v = v_first;
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            if ((int)threadIdx.x + u * kFrontThreads < total) park1(raw[u], sm, v);
            sm += q512;
            v += r512;
            if (v >= VPR) { v -= VPR; ++sm; }
        }
        for (int e = (int)threadIdx.x + kFrontThreads * UL; e < total; e += kFrontThreads) {   // rows wider than 8 vectors
            park1(__builtin_nontemporal_load(src_of(e, sm, v)), sm, v);
            sm += q512;
            v += r512;
            if (v >= VPR) { v -= VPR; ++sm; }
        }
        lds_barrier();
        if constexpr (COUNT) {
            // Counts leave as one row of cnt[counting tile][bucket] with the ACTUAL number of item units; the totals -- the bucket
            // bases -- take every (scatter tile, bucket) run rounded up to plan.pad units (line-aligned runs, plan.pad > 1: the
            // scatter pass reserves its runs in such multiples; a workgroup's counting tiles are then whole scatter tiles, the
            // host makes `rounds` a multiple of cps)
            uint32_t *row = cnt + (size_t)(tile0 + r) * plan.total_buckets;
            const bool close = plan.pad > 1u && ((uint32_t)((tile0 + r + 1) % (int64_t)cps) == 0u || tile0 + r + 1 == tiles);
#pragma unroll
            for (int j = 0; j < HE; ++j) {
                const int hc = hcol_of(j);
                if (hc >= 0) {
                    const uint32_t h = s_hist[threadIdx.x + j * kFrontThreads];
                    s_hist[threadIdx.x + j * kFrontThreads] = 0;
                    row[hc] = h;
                    hacc[j] += h;
                    hsum[j] += h;
                    if (close) {
                        hsum[j] += (0u - hacc[j]) & (plan.pad - 1u);
                        hacc[j] = 0;
                    }
                }
            }
        }
        // LDS image -> gT: (level, vector) pairs over all threads; a wave stays inside one level per trip
        const int work = (le - lb) << nvec_log2;
        for (int idx = threadIdx.x; idx < work; idx += kFrontThreads) {
            const int l = lb + (idx >> nvec_log2), smo = (idx & ((1 << nvec_log2) - 1)) << M_LOG2;
            uint32_t m = 0;
            if (smo < ns) {
                const f32x4 val = *reinterpret_cast<const f32x4 *>(&s_tile[l * pitch + smo]);
                float *dst = gT + ((int64_t)l * NP + s0 + smo) * F;
                if (smo + M <= ns) {
                    // fp32 gradients: PLAIN stores (round 4) -- the scatter pass reads these 134 MB back within ~100 us, and the
                    // Infinity Cache keeps what a streaming store would have sent to HBM (S1 backward -2.8 %,
                    // profiles/r04_experiments.md 9); fp16 gradients (half the input bytes in flight) measured 2 % better streaming
                    if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4 *>(dst) = val;
                    else __builtin_nontemporal_store(val, reinterpret_cast<f32x4 *>(dst));
                    if constexpr (GMAX) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t b = __float_as_uint(fabsf(val[j]));
                            m = b > m ? b : m;
                        }
                    }
                } else {   // last sample of an odd tail (M == 2)
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        dst[j] = val[j];
                        if constexpr (GMAX) {
                            const uint32_t b = __float_as_uint(fabsf(val[j]));
                            m = b > m ? b : m;
                        }
                    }
                }
            }
            if constexpr (GMAX) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t o = __shfl_xor(m, off, 64);
                    m = o > m ? o : m;
                }
                if ((threadIdx.x & 63) == 0 && m) atomicMax(&s_max[l], m);
            }
        }
        lds_barrier();   // s_tile is refilled by the next round (the stores keep draining)
    }
    __syncthreads();
    if constexpr (COUNT) {
        // lanes = consecutive buckets: contiguous atomics into one of kTotalShards copies of the totals (512 workgroups adding to the same word serialise at the memory side: 18 us of the 30 this kernel took
        // on 65 536 samples); the bucket scan adds the copies up
        uint32_t *mine = totals + (size_t)(blockIdx.x % kTotalShards) * kMaxBuckets;
#pragma unroll
        for (int j = 0; j < HE; ++j) {
            const int hc = hcol_of(j);
            if (hc >= 0 && hsum[j]) atomicAdd(&mine[hc], hsum[j]);
        }
    }
    if constexpr (GMAX) {
        if ((int)threadIdx.x >= lb && (int)threadIdx.x < le && s_max[threadIdx.x])
            atomicMax(&gmax[threadIdx.x], s_max[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------------------- pass S
// single block: bucket bases (exclusive scan of totals) and the consumer work list
//   base[b]         first item of bucket b in the item array (base[nb] = total)
//   unit_first[b]   first work unit of bucket b; unit_first[nb] = number of units
//   unit_desc[u]    item range, bucket and level of work unit u
__global__ __launch_bounds__(1024) void bin_scan_buckets_kernel(const uint32_t *__restrict__ totals,
                                                                uint64_t *__restrict__ base,
                                                                uint32_t *__restrict__ unit_first,
                                                                UnitDesc *__restrict__ unit_desc, uint32_t nb,
                                                                BinPlan plan,
                                                                uint32_t *__restrict__ work_counter,
                                                                unsigned long long *__restrict__ cursor,
                                                                const uint32_t *__restrict__ gmax) {
    __shared__ uint64_t s_items[kMaxBuckets + 2];
    __shared__ uint32_t s_units[kMaxBuckets + 2];
    __shared__ uint64_t s_wave_items[16];
    __shared__ uint32_t s_wave_units[16];
    // each thread owns buckets 2t, 2t+1 (kMaxBuckets = 2 * 1024)
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) work_counter[0] = 0;   // the persistent consume pass fetches its units from here
    uint64_t c[2];
    uint32_t u[2], lv_of[2], ck[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t b = 2 * t + k;
        c[k] = 0u;
        if (b < nb)
            for (int sh = 0; sh < kTotalShards; ++sh) c[k] += totals[(size_t)sh * kMaxBuckets + b];
        // the bucket's level: a fixed-length scan with uniform indices (scalar loads of the whole arrays in one go; a
        // `q < plan.nbl` loop bound and the lane-dependent blevel[lq] / lv[..].chunk lookups were three dependent memory
        // round trips in a one-workgroup kernel that sits on every backward's critical path)
        uint32_t lvl = plan.blevel[0];
#pragma unroll
        for (uint32_t q = 1; q < SHACIRA_MAX_LODS; ++q)
            if (q < plan.nbl && plan.bstart[q] <= b) lvl = plan.blevel[q];
        lv_of[k] = lvl;
        ck[k] = plan.chunk;   // (one unit size for every level of a plan)
        u[k] = (c[k] >> 32) ? (uint32_t)((c[k] + ck[k] - 1) / ck[k]) : (c[k] ? ((uint32_t)c[k] - 1u) / ck[k] + 1u : 0u);
    }
    uint64_t ci = c[0] + c[1];
    uint32_t ui = u[0] + u[1];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t nc = __shfl_up(ci, off, 64);
        const uint32_t nu = __shfl_up(ui, off, 64);
        if (lane >= (uint32_t)off) { ci += nc; ui += nu; }
    }
    if (lane == 63) { s_wave_items[wave] = ci; s_wave_units[wave] = ui; }
    __syncthreads();
    uint64_t wc = 0;
    uint32_t wu = 0;
    for (uint32_t w = 0; w < wave; ++w) { wc += s_wave_items[w]; wu += s_wave_units[w]; }
    const uint64_t ex_items = wc + ci - (c[0] + c[1]);
    const uint32_t ex_units = wu + ui - (u[0] + u[1]);
    s_items[2 * t] = ex_items;
    s_items[2 * t + 1] = ex_items + c[0];
    s_units[2 * t] = ex_units;
    s_units[2 * t + 1] = ex_units + u[0];
    if (t == 1023) {  // grand totals for nb == kMaxBuckets
        s_items[kMaxBuckets] = ex_items + c[0] + c[1];
        s_units[kMaxBuckets] = ex_units + u[0] + u[1];
    }
    __syncthreads();
    for (uint32_t b = t; b <= nb; b += 1024) {
        base[b] = s_items[b];          // entries >= nb hold the grand totals (zero counts beyond nb)
        cursor[b] = s_items[b];        // the scatter pass reserves its runs from here (returning atomics)
        unit_first[b] = s_units[b];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t b = 2 * t + k;
        if (b < nb) {
            const uint32_t lvl = lv_of[k];
            const uint32_t gm = (gmax != nullptr && u[k] > 0) ? gmax[lvl] : 0u;
            for (uint32_t q = 0; q < u[k]; ++q) {
                UnitDesc d;
                d.begin = s_items[b] + (uint64_t)q * ck[k];
                const uint64_t bucket_end = s_items[b] + c[k];
                d.end = (d.begin + ck[k] < bucket_end) ? (d.begin + ck[k]) : bucket_end;
                d.bucket = b;
                d.level = lvl;
                d.single = u[k] == 1 ? 1u : 0u;
                d.gmax_bits = gm;
                unit_desc[s_units[b] + q] = d;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- pass A
// Standalone counting pass (calls that do not transpose: sub-batches, level-range calls on staged gradients, rows the
// 16-byte front kernel cannot read). One workgroup per TILE counting every binned level: the coordinates are loaded once;
// counts leave as one global atomic per (workgroup, non-empty bucket) into totals[].
template <int DIM>
__global__ __launch_bounds__(kBinThreads) void bin_count_levels_kernel(LevelTable lt, BinPlan plan,
                                                                       const float *__restrict__ coords,
                                                                       uint32_t *__restrict__ totals,
                                                                       uint32_t *__restrict__ cnt, int64_t sample0,
                                                                       int64_t N) {
    __shared__ uint32_t s_hist[SHACIRA_MAX_LODS][kMaxLevelBuckets];
    constexpr int kTileD = TileOf<DIM>::value;
    constexpr int SPT = kTileD / kBinThreads;
    const uint32_t tile = blockIdx.x;
    // gridDim.y workgroups share a tile's levels (bi = blockIdx.y, blockIdx.y + gridDim.y, ...): small batches keep the chip
    // busy with one level each, large ones load the coordinates once for all levels
    for (uint32_t e = threadIdx.x; e < plan.nbl * kMaxLevelBuckets; e += kBinThreads) (&s_hist[0][0])[e] = 0;
    double t[SPT][DIM];
    bool live[SPT];
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const int64_t i = sample0 + (int64_t)tile * kTileD + threadIdx.x + u * kBinThreads;
        live[u] = i < N;
        load_unit_coords<DIM>(coords, i, N, t[u]);
    }
    __syncthreads();
#pragma unroll 1
    for (uint32_t bi = blockIdx.y; bi < plan.nbl; bi += gridDim.y) {
        const uint32_t lvl = plan.blevel[bi];
        const BinLevel bl = plan.lv[lvl];
        const int32_t res = lt.res[lvl];
        const float hi = lt.hi[lvl];
        const bool dense = lt.dense[lvl] != 0;
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            if (!live[u]) continue;
            if (bl.compact) {
                if constexpr (DIM == 3) {
                    atomicAdd(&s_hist[bi][axis_pos(t[u][2], res, hi) / bl.slab], 2u);
                } else {
                    atomicAdd(&s_hist[bi][compact2d_line(axis_pos(t[u][1], res, hi), (uint32_t)res) / bl.slab], 1u);
                }
                continue;
            }
            uint32_t bk[1 << (DIM - 1)];
            bool ok[1 << (DIM - 1)];
            enumerate_buckets<DIM>(t[u], res, hi, dense, lt.mask, bl, bk, ok);
#pragma unroll
            for (int q = 0; q < (1 << (DIM - 1)); ++q)
                if (ok[q]) atomicAdd(&s_hist[bi][bk[q]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t bi = blockIdx.y; bi < plan.nbl; bi += gridDim.y) {
        const BinLevel bl = plan.lv[plan.blevel[bi]];
        for (uint32_t b = threadIdx.x; b < bl.nb; b += kBinThreads)
        {
            const uint32_t c = (s_hist[bi][b] + plan.pad - 1u) & ~(plan.pad - 1u);   // (tile = scatter tile: padded run length)
            cnt[(size_t)tile * plan.total_buckets + bl.bucket0 + b] = s_hist[bi][b];     // (rows: actual counts; totals: padded)
            if (c)
                atomicAdd(&totals[(size_t)((blockIdx.x + blockIdx.y) % kTotalShards) * kMaxBuckets + bl.bucket0 + b], c);
        }
    }
}

// ------------------------------------------------------------------------------------------------- table zeroing
// at::zeros_like of the reference, minus what the consume pass overwrites anyway: the rows of a HASHED binned level are
// covered by its buckets, and a bucket with exactly one work unit writes all its rows with plain stores. So only the
// other rows are zeroed up front (S1: 6.6 of 48.8 MB; the table-sized memset was 13 of config D's 93 us and 40 MB of the
// write-bound traffic of every call) and the buckets that turn out to have 0 or several units are zeroed once the bucket
// scan knows them. grid (x, num_lods): segment l = rows [first_idx[l], first_idx[l + 1]) (segment 0 starts at row 0).
// Slice blockIdx.y == num_lods (when launched with one extra slice) zeroes the call's control words instead: one launch
// fewer on the small-batch path.
__global__ __launch_bounds__(256) void zero_unowned_rows_kernel(float *__restrict__ acc,
                                                                const int32_t *__restrict__ first_idx, LevelTable lt,
                                                                BinPlan plan, uint32_t *__restrict__ words,
                                                                uint32_t nwords) {
    if ((int)blockIdx.y == lt.num_lods) {
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nwords; i += gridDim.x * 256) words[i] = 0u;
        return;
    }
    const int l = blockIdx.y, F = lt.feature_dim;
    const int64_t level0 = first_idx[l];
    const int64_t start = (l == 0) ? 0 : level0;
    const int64_t end = (l + 1 < lt.num_lods) ? (int64_t)first_idx[l + 1] : lt.table_rows;
    const BinLevel bl = plan.lv[l];
    const bool covered = bl.nb > 0 && bl.dgroup < 0 && lt.dense[l] == 0;   // hashed + binned: rows [level0, level0 + used)
    // two plain ranges around the covered rows: [start, hole_lo) and [hole_hi, end)
    const int64_t hole_lo = covered ? level0 : end;
    int64_t hole_hi = covered ? level0 + (int64_t)bl.used : end;
    if (hole_hi > end) hole_hi = end;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t e = start * F + t0; e < hole_lo * F; e += stride) acc[e] = 0.0f;
    for (int64_t e = hole_hi * F + t0; e < end * F; e += stride) acc[e] = 0.0f;
}

// control words of a call (bucket totals, per-level max |grad_output|): a kernel of our own rather than hipMemsetAsync -- a
// memset node captured into a HIP graph after an eager call on ANOTHER stream was seen not to take effect on replay
// (round 3: stale totals -> wrong bucket bases)
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *__restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0u;
}

// Feature order rotated by a lane-dependent amount WITHOUT a private array: r[k] = a[(k + rot) mod F]. (Round 4: the previous
// form, a select chain `pick(a, j)` per atomic, was turned back into a dynamically indexed stack array by the compiler for
// F = 4 -- `scratch_store_dwordx4` + one `scratch_load_dword` per LDS atomic, 1 254 scratch instructions in the consume
// kernel: nerf_lego.yaml's table paid 2x on its consume pass for it.) A butterfly of conditional swaps on the bits of `rot`,
// once per gradient vector; the opaque barrier on the conditions keeps the optimiser from recognising an indexed array again.
template <int F> __device__ __forceinline__ void rotate_features(const float (&a)[F], int rot, float (&r)[F]) {
    static_assert(F == 1 || F == 2 || F == 4 || F == 8, "power-of-two feature counts");
#pragma unroll
    for (int k = 0; k < F; ++k) r[k] = a[k];
#pragma unroll
    for (int bit = 1; bit < F; bit <<= 1) {
        int on = rot & bit;
        asm volatile("" : "+v"(on));   // opaque: a plain v_cndmask per element, never a table lookup
        float t[F];
#pragma unroll
        for (int k = 0; k < F; ++k) t[k] = on ? r[(k + bit) & (F - 1)] : r[k];
#pragma unroll
        for (int k = 0; k < F; ++k) r[k] = t[k];
    }
}

}  // namespace shacira
