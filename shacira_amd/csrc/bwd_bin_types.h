// bwd_bin_types.h -- plan / item / unit types, corner and bucket enumeration, fixed-point helpers of the binned backward
// Part of the translation unit hashgrid_bwd_bin.hip (included there, in this order: bwd_bin_types.h, bwd_bin_front.h,
// bwd_bin_passes.h); see that file's header for the pipeline.
#pragma once

#include "internal.h"

namespace shacira {

#ifndef SHACIRA_KTILE
#define SHACIRA_KTILE 1024
#endif
#ifndef SHACIRA_KBIN
#define SHACIRA_KBIN 512
#endif
constexpr int kTile = SHACIRA_KTILE;       // samples per (level, tile) block in passes A and B, 3-D
// 2-D samples have half as many x-pairs: tiles of twice as many samples fill the same LDS staging buffer and halve the number
// of scatter workgroups (each pays the same latencies and barriers whatever it carries: 16 us per level either way before)
#ifndef SHACIRA_TILE2D_MUL
#define SHACIRA_TILE2D_MUL 2
#endif
template <int DIM> struct TileOf { static constexpr int value = (DIM == 2) ? SHACIRA_TILE2D_MUL * kTile : kTile; };
static inline int tile_samples(int dim) { return dim == 2 ? SHACIRA_TILE2D_MUL * kTile : kTile; }
constexpr int kBinThreads = SHACIRA_KBIN;  // threads of passes A and B
// Scatter pass, 3-D with F = 2: 1 024 threads = one sample per thread (38 instead of 64 VGPRs, 32 waves per CU with the same two
// workgroups' worth of staging): S1 backward -2 % (0.509 -> 0.498 ms, 2^19 samples 0.281 -> 0.275); 2-D and config D equal;
// the F = 4 kernels (staging windows) lose 25 % with it and keep 512 (profiles/r05_experiments.md 4).
#ifndef SHACIRA_KSCATTER3D
#define SHACIRA_KSCATTER3D 1024
#endif
// (fp32 item streams only: the 8-byte half-precision stream of fp16 tables loses 8 % with it, S1 fp16 0.435 -> 0.470 ms)
template <int DIM, int F, int FMT> struct ScatterThreads {
    static constexpr int value = (DIM == 3 && F == 2 && FMT != 1) ? SHACIRA_KSCATTER3D : kBinThreads;
};
constexpr int kConsumeThreads = 1024;
constexpr int kMaxBuckets = 2048;     // over all levels
constexpr int kMaxLevelBuckets = 128; // per level (LDS histogram size)
constexpr int kTotalShards = 16;      // bucket totals are accumulated in this many copies (same-address atomic contention)

struct BinLevel {
    uint32_t nb;        // buckets in this level
    uint32_t bucket0;   // global index of its first bucket
    uint32_t rows_pb;   // rows per bucket (hashed: BR; dense: G*res)
    uint32_t G;         // dense: x-lines per bucket
    uint64_t magicG;    // ceil(2^40 / G): line / G == (line * magicG) >> 40 for line < 2^20
    uint32_t used;      // rows of the level the kernels can touch: dense res^d, hashed 2^bw
    uint32_t shift;     // hashed: log2(BR)
    int32_t dgroup;     // >= 0: "direct" level (fits one LDS image): index of its group; -1: binned level
    uint32_t drow0;     // direct: first row of the level inside its group's LDS image
    uint32_t compact;   // 1: dense 3-D level binned by z-slab with ONE two-slot item per sample (all 8 corners): 32 B (F = 2), 48 B (F = 4)
    uint32_t slab;      // compact: base-cell planes per bucket (its image holds slab + 1 planes)
    uint32_t chunk;     // items per consumer work unit of this level
};

// What the counting needs to know about binned level q, dense in q (one unchained scalar load per level: reading the fields
// through blevel[q] -> lv[lvl] / lt.res[lvl] chained four scalar-load round trips per level and made the fused front kernel
// latency bound).
struct CountLevel {
    int32_t res;
    float hi;
    uint32_t kind;      // 0 hashed, 1 compact (z slab), 2 dense x-lines
    uint32_t shift;     // hashed: log2(rows per bucket)
    uint32_t m_lo, m_hi; // compact: m_lo = ceil(2^18 / slab) (pz / slab == (pz * m_lo) >> 18, checked by make_plan); dense: magicG
};

struct BinPlan {
    BinLevel lv[SHACIRA_MAX_LODS];
    CountLevel cl[SHACIRA_MAX_LODS];
    uint32_t total_buckets;
    uint32_t BR;
    uint32_t num_tiles;
    uint32_t pairs;     // items per (sample, level) = 2^(dim-1)
    uint32_t chunk;     // items per consumer work unit
    uint32_t chunk_min; // smallest unit size of the plan (sizes the consume pass's unit descriptors)
    uint32_t nbl;       // number of binned levels
    uint32_t pad;       // a (tile, bucket) run is reserved in multiples of this many item units (power of two; 1 = exact): with
                        // SHACIRA_RUN_ALIGN / unit bytes every run starts and ends on a 64-byte boundary, see make_plan
    uint32_t stage_cap; // item units the scatter pass's LDS staging buffer holds: the tile's items + the pad units of its runs
    uint32_t blevel[SHACIRA_MAX_LODS];  // their level indices (grid.y of passes A/B); 32-bit = scalar loads
    uint32_t bstart[SHACIRA_MAX_LODS];  // first global bucket of binned level q (= lv[blevel[q]].bucket0)
    uint32_t bnb[SHACIRA_MAX_LODS];     // its number of buckets (= lv[blevel[q]].nb): one unchained scalar load in the front kernel
    uint32_t ngroups;   // groups of direct levels
    uint32_t gmask[SHACIRA_MAX_LODS];   // levels of each group (bit l)
    uint32_t grows[SHACIRA_MAX_LODS];   // rows of each group's LDS image
};

template <int F> struct alignas(F == 2 ? 16 : 8) Item {
    uint32_t key;
    float fx;
    float a[F];
};

// fp16 tables with F = 2 (the reference's NeRF mode: AMP on, grid.py:73, .cu:198-211): the gradient is stored as fp16 anyway,
// so the item stream may carry half-precision payloads -- HALF the bytes of the pass that bounds the backward:
//   pair item, 8 B:     key = rowA (13) | kx (4) | validA | validB | fx quantised to 13 bits;  a = half2(g0 w, g1 w)
//                       rowB = kx ? rowA ^ (2^kx - 1) : rowA + 1   (hashed: x ^ (x + 1) = 2^kx - 1; dense: the next row)
//   compact item, 16 B: key = local base row | valid;  fx, fy, fz as 16-bit fixed point;  g = half2(g0, g1)  (two 8-byte units)
// Weight error <= 2^-14, payload rounding 2^-11 relative per term (the reference's own fp16 atomics round the running SUM to
// 11 bits at every add); sums are still accumulated in the 64-bit fixed-point / fp64 LDS images.
struct alignas(8) ItemH {
    uint32_t key;
    __half2 a;
};
struct alignas(16) ItemHC {
    uint32_t key;
    uint16_t fx, fy, fz, pad;
    __half2 g;
};
// F = 4 (nerf_lego.yaml under AMP): 16-byte pair items {key, fx (fp32: exact weights), half2 a01, half2 a23} instead of
// 24 bytes moved as 8-byte pieces; compact items = two 16-byte units {key, fx, fy, fz} {half2 g01, half2 g23, -, -} (32
// instead of 48 bytes). Keys as in the fp32 stream.
struct alignas(16) ItemH4 {
    uint32_t key;
    float fx;
    uint32_t p2, p3;   // two half2 as raw bits (pair item: a01, a23; compact unit 0: fy, fz as fp32 bits)
};
// fp32 tables, 3-D, F = 2, large batches (round 5): 12-byte units. Both item passes move nothing but this stream at the
// memory system's rate, so its bytes are the backward's time: 16 -> 12 bytes per x-pair, 32 -> 24 per compact sample.
//   pair item:     w0 = rowA (13) | code (4) << 13 | fx bits [6, 21) << 17
//                  w1 = a0 rounded to a 21-bit mantissa | fx bits [3, 6);   w2 = a1 likewise | fx bits [0, 3)
//                  code: 0 no corner (pad unit), 1..13 hashed level: rowB = rowA ^ (2^code - 1), 14 dense: rowB = rowA + 1,
//                  15 dense: rowA only (x + 1 outside the level). fx = 21-bit fixed point.
//   compact item:  two units {local base row | valid << 26, fx, fy} {fz, g0, g1}: exact, as in the 16-byte stream.
// Error of a pair item: |a| * (2^-22 from the mantissa + 2^-22 from fx) per contribution, against the 1e-5 bar on the level's
// largest gradient (measured ~5e-7 of it where the 16-byte stream measures 1e-7); fp16 / F = 4 / 2-D streams are unchanged.
struct alignas(4) Item12 {
    uint32_t w[3];
};
// FMT: 0 = fp32 payloads (16- / 24-byte units), 1 = half-precision stream of fp16 tables, 2 = 12-byte units (Item12)
template <int F, int FMT> struct ItemSel { typedef Item<F> type; };
template <> struct ItemSel<2, 1> { typedef ItemH type; };
template <> struct ItemSel<4, 1> { typedef ItemH4 type; };
template <> struct ItemSel<2, 2> { typedef Item12 type; };

__device__ __forceinline__ Item12 pack_item12(uint32_t key, float fx, bool dense, float a0, float a1) {
    const uint32_t ra = key & 0x1FFFu, rb = (key >> 13) & 0x1FFFu;
    const uint32_t vb = (key >> 27) & 1u;
    const uint32_t code = dense ? (vb ? 14u : 15u) : (32u - (uint32_t)__clz((int)(ra ^ rb)));
    uint32_t q = (uint32_t)(fx * 2097152.0f + 0.5f);   // fx in [0, 1): 21 bits, round to nearest
    q = q > 2097151u ? 2097151u : q;
    Item12 it;
    it.w[0] = ra | (code << 13) | ((q >> 6) << 17);
    it.w[1] = ((__float_as_uint(a0) + 4u) & ~7u) | ((q >> 3) & 7u);   // (mantissa rounded half-up in magnitude; inf stays inf)
    it.w[2] = ((__float_as_uint(a1) + 4u) & ~7u) | (q & 7u);
    return it;
}
__device__ __forceinline__ void unpack_item12(const uint32_t (&w)[3], uint32_t &ra, uint32_t &rb, bool &va, bool &vb, float &fx,
                                              float (&a)[2]) {
    const uint32_t code = (w[0] >> 13) & 15u;
    ra = w[0] & 0x1FFFu;
    rb = code >= 14u ? ((ra + 1u) & 0x1FFFu) : (ra ^ ((1u << code) - 1u));
    va = code != 0u;
    vb = code != 0u && code != 15u;
    const uint32_t q = ((w[0] >> 17) << 6) | ((w[1] & 7u) << 3) | (w[2] & 7u);
    fx = (float)q * (1.0f / 2097152.0f);
    a[0] = __uint_as_float(w[1] & ~7u);
    a[1] = __uint_as_float(w[2] & ~7u);
}
template <bool STREAM = true> __device__ __forceinline__ void store_item_nt(Item12 *p, const Item12 &it) {
    // three dword stores that the compiler merges into one global_store_dwordx3 nt (a 3-vector store would be widened to four
    // elements by the front end); 12-byte lane accesses stream as fast as 16-byte ones (tools/microbench3.hip)
    __builtin_nontemporal_store(it.w[0], &p->w[0]);
    __builtin_nontemporal_store(it.w[1], &p->w[1]);
    __builtin_nontemporal_store(it.w[2], &p->w[2]);
}
__device__ __forceinline__ Item12 load_item12_nt(const Item12 *p) {
    Item12 it;
    it.w[0] = __builtin_nontemporal_load(&p->w[0]);
    it.w[1] = __builtin_nontemporal_load(&p->w[1]);
    it.w[2] = __builtin_nontemporal_load(&p->w[2]);
    return it;
}

// LDS staging windows per scatter tile (bwd_bin_passes.h, pass B), by item format; measured same-box (profiles/
// r04_experiments.md 14): the 16-byte fp32 and 8-byte half items of F = 2 and the 16-byte half items of F = 4 lose 1-3 % with windows; the
// 24-byte items of F = 4 win 8 % with four (their 100 KiB of staging allowed ONE workgroup per CU). -DSHACIRA_SCATTER_SPLIT=n forces one value for every format (A/B builds).
#ifdef SHACIRA_SCATTER_SPLIT
template <class ItemT> struct ScatterSplit { static constexpr int value = SHACIRA_SCATTER_SPLIT; };
#else
template <class ItemT> struct ScatterSplit { static constexpr int value = 1; };
template <> struct ScatterSplit<Item<4>> { static constexpr int value = 4; };
#endif

__device__ __forceinline__ uint32_t pack_half_key(uint32_t key, float fx, bool dense) {
    const uint32_t ra = key & 0x1FFFu, rb = (key >> 13) & 0x1FFFu;
    const uint32_t kx = dense ? 0u : (32u - (uint32_t)__clz((int)(ra ^ rb)));
    const uint32_t fq = (uint32_t)(fx * 8192.0f);   // fx in [0, 1)
    return ra | (kx << 13) | (((key >> 26) & 3u) << 17) | ((fq > 8191u ? 8191u : fq) << 19);
}
__device__ __forceinline__ void unpack_half_key(uint32_t k, uint32_t &ra, uint32_t &rb, bool &va, bool &vb, float &fx) {
    ra = k & 0x1FFFu;
    const uint32_t kx = (k >> 13) & 15u;
    rb = kx ? (ra ^ ((1u << kx) - 1u)) : ((ra + 1u) & 0x1FFFu);
    va = (k >> 17) & 1u;
    vb = (k >> 18) & 1u;
    fx = ((float)(k >> 19) + 0.5f) * (1.0f / 8192.0f);
}
// (raw 16-bit halves on purpose: __builtin_bit_cast between uint32_t and __half2 miscompiled here -- both halves came out
// as the low one)
__device__ __forceinline__ uint32_t float2_to_half2_bits(float a, float b) {
    return (uint32_t)__half_as_ushort(__float2half_rn(a)) | ((uint32_t)__half_as_ushort(__float2half_rn(b)) << 16);
}
__device__ __forceinline__ float2 half2_bits_to_float2(uint32_t bits) {
    return make_float2(__half2float(__ushort_as_half((unsigned short)(bits & 0xFFFFu))),
                       __half2float(__ushort_as_half((unsigned short)(bits >> 16))));
}
template <bool STREAM = true> __device__ __forceinline__ void store_item_nt(ItemH4 *p, const ItemH4 &it) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    __builtin_memcpy(&v, &it, 16);
    *reinterpret_cast<u32x4 *>(p) = v;     // plain, like the 8-byte items below (nerf_lego table, fp16: backward -2.5 %)
}
template <bool STREAM = true> __device__ __forceinline__ void store_item_nt(ItemH *p, const ItemH &it) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    __builtin_memcpy(&v, &it, 8);
    // PLAIN 8-byte stores (round 4): as non-temporal stores the half-size items of fp16 tables cost the S1 backward 7 % (a wave's
    // store covers 512 bytes: partial lines at both ends of every run, which a streaming store does not merge in L2)
#ifdef SHACIRA_HALF_NT   // A/B builds
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p));
#else
    *reinterpret_cast<u32x2 *>(p) = v;
#endif
}

// One consumer work unit, written by the bucket scan: everything a consume workgroup needs in ONE 32-byte load (it used to
// chase unit -> bucket -> base / unit_first -> level through four dependent loads and a 15-step scalar search: ~8 us per
// unit before the first item arrived).
struct alignas(16) UnitDesc {
    uint64_t begin, end;   // item range
    uint32_t bucket;       // global bucket index
    uint32_t level;
    uint32_t single;       // 1: the bucket's only unit (rows are written with plain stores)
    uint32_t gmax_bits;    // bit pattern of the level's max |grad_output| (fixed-point scale; 0 when the call keeps fp64)
};

// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains the wave's global loads AND stores
// (s_waitcnt vmcnt(0)), which serialises a block's write-out with its next phase.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Items are written once and read once: stream them past the caches (non-temporal).
// STREAM (compile time: a run-time choice between the two load instructions made the compiler serialise the round's loads):
// the call's item array is larger than the caches hold between the two passes, so the write-once / read-once accesses go past
// them. Batches below 2^17 samples (<= ~60 MB of items; they are the ones that accumulate in fp64 images) are written and read
// with PLAIN accesses instead and the consume pass finds its items in L2 / the Infinity Cache (round 4: -4 % on the
// 65 536-sample fwd+bwd pair; from 2^18 samples on plain accesses lose).
template <bool STREAM = true, int F> __device__ __forceinline__ void store_item_nt(Item<F> *p, const Item<F> &it) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (sizeof(Item<F>) == 16) {
        u32x4 v;
        __builtin_memcpy(&v, &it, 16);
        // (large batches: non-temporal -- plain stores change nothing on S1 and trade -3 % on the 2-D backward against a slower
        // forward right behind it; profiles/r04_experiments.md 9)
        if constexpr (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
        else *reinterpret_cast<u32x4 *>(p) = v;
    } else {   // 24-byte items (F = 4, 8-byte aligned): three 8-byte stores instead of six dwords
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        static_assert(sizeof(Item<F>) % 8 == 0, "item size");
        u32x2 *q = reinterpret_cast<u32x2 *>(p);
        u32x2 d[sizeof(Item<F>) / 8];
        __builtin_memcpy(d, &it, sizeof(Item<F>));
#pragma unroll
        // (non-temporal: plain stores measured +8 % on the nerf_lego table's backward)
        for (int k = 0; k < (int)(sizeof(Item<F>) / 8); ++k) __builtin_nontemporal_store(d[k], q + k);
    }
}

template <int F, bool STREAM = true> __device__ __forceinline__ Item<F> load_item_nt(const Item<F> *p) {
    Item<F> it;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (sizeof(Item<F>) == 16) {
        u32x4 v;
        if constexpr (STREAM) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        else v = *reinterpret_cast<const u32x4 *>(p);
        __builtin_memcpy(&it, &v, 16);
    } else {   // 24-byte items: three 8-byte loads. PLAIN loads, not non-temporal ones (round 4): the 16- and 8-byte pieces of
               // neighbouring items share lines, and a streaming load does not keep them (nerf_lego table backward -5 %)
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 *q = reinterpret_cast<const u32x2 *>(p);
        u32x2 d[sizeof(Item<F>) / 8];
#pragma unroll
        for (int k = 0; k < (int)(sizeof(Item<F>) / 8); ++k) d[k] = q[k];
        __builtin_memcpy(&it, d, sizeof(Item<F>));
    }
    return it;
}

// One x-pair of corners of a (sample, level), in bucket coordinates.
struct PairSlot {
    uint32_t bucket;  // level-local bucket index
    uint32_t key;     // rowA | rowB << 13 | validA << 26 | validB << 27   (0 valid bits -> nothing to add)
    float wrest;      // product of the non-x weights
};

// Enumerates the 2^(DIM-1) x-pairs of one (sample, level). fx/gx are the x-axis weights (corner x+1 / corner x).
template <int DIM>
__device__ __forceinline__ void enumerate_pairs(const double (&t)[DIM], int32_t res, float hi, bool dense,
                                                uint32_t mask, const BinLevel &bl, uint32_t BR, float &fx,
                                                PairSlot (&out)[1 << (DIM - 1)]) {
    int32_t p[DIM];
    float f[DIM], g[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f[a], g[a]);
    fx = f[0];
    const uint32_t ux = (uint32_t)p[0];
    const uint32_t r = (uint32_t)res;
    constexpr int NP = 1 << (DIM - 1);
    // hashed levels: the two products per axis once ((y + 1) * P = y * P + P mod 2^32) -- written per pair, each pair's
    // v_mul_lo_u32 (a quarter-rate instruction) stayed inside its own uniform branch: eight per sample instead of two
    uint32_t hyv[2] = {0u, 0u}, hzv[2] = {0u, 0u};
    if (!dense) {
        hyv[0] = (uint32_t)p[1] * kPrimeY;
        hyv[1] = hyv[0] + kPrimeY;
        if constexpr (DIM == 3) {
            hzv[0] = (uint32_t)p[2] * kPrimeZ;
            hzv[1] = hzv[0] + kPrimeZ;
        }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        // q bit (DIM-2) -> y offset, bit 0 -> z offset (3-D); q -> y offset (2-D): same order as the corner bits
        const int dy = (DIM == 3) ? ((q >> 1) & 1) : (q & 1);
        const int dz = (DIM == 3) ? (q & 1) : 0;
        float w = dy ? f[1] : g[1];
        if constexpr (DIM == 3) w = w * (dz ? f[2] : g[2]);
        out[q].wrest = w;
        const uint32_t uy = (uint32_t)p[1] + dy;
        uint32_t uz = 0;
        if constexpr (DIM == 3) uz = (uint32_t)p[2] + dz;
        if (dense) {
            // corners with a coordinate == res lie outside the level (weight 0 in the reference): dropped
            bool ok = uy < r;
            uint32_t line = uy;
            if constexpr (DIM == 3) {
                ok = ok && uz < r;
                line += uz * r;
            }
            const uint32_t b = (uint32_t)(((uint64_t)line * bl.magicG) >> 40);
            const uint32_t ra = (line - b * bl.G) * r + ux;
            const uint32_t va = ok ? 1u : 0u;
            const uint32_t vb = (ok && (ux + 1u) < r) ? 1u : 0u;
            out[q].bucket = ok ? b : 0u;
            out[q].key = (ra & 0x1FFFu) | (((ra + 1u) & 0x1FFFu) << 13) | (va << 26) | (vb << 27);
        } else {
            uint32_t h = hyv[dy];
            if constexpr (DIM == 3) h ^= hzv[dz];
            const uint32_t rowA = (ux ^ h) & mask;
            const uint32_t rowB = ((ux + 1u) ^ h) & mask;
            out[q].bucket = rowA >> bl.shift;
            out[q].key = (rowA & (BR - 1u)) | ((rowB & (BR - 1u)) << 13) | (3u << 26);
        }
    }
}

// Buckets of the 2^(DIM-1) x-pairs of one (sample, level) WITHOUT the x axis: a pair's bucket and validity depend on its
// (y[, z]) line only (hashed: x < 2^shift never reaches the bucket bits; dense: buckets hold whole x-lines, and a pair is
// dropped only when its line lies outside the level). Same result as enumerate_pairs(...).bucket / (key >> 26 != 0) at a
// third of the arithmetic: this is what the counting passes run.
template <int DIM>
__device__ __forceinline__ void enumerate_buckets(const double (&t)[DIM], int32_t res, float hi, bool dense, uint32_t mask,
                                                  const BinLevel &bl, uint32_t (&bucket)[1 << (DIM - 1)],
                                                  bool (&valid)[1 << (DIM - 1)]) {
    int32_t p[DIM];
    float f, g;
#pragma unroll
    for (int a = 1; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f, g);
    const uint32_t r = (uint32_t)res;
    constexpr int NP = 1 << (DIM - 1);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int dy = (DIM == 3) ? ((q >> 1) & 1) : (q & 1);
        const int dz = (DIM == 3) ? (q & 1) : 0;
        const uint32_t uy = (uint32_t)p[1] + dy;
        uint32_t uz = 0;
        if constexpr (DIM == 3) uz = (uint32_t)p[2] + dz;
        if (dense) {
            bool ok = uy < r;
            uint32_t line = uy;
            if constexpr (DIM == 3) {
                ok = ok && uz < r;
                line += uz * r;
            }
            valid[q] = ok;
            bucket[q] = ok ? (uint32_t)(((uint64_t)line * bl.magicG) >> 40) : 0u;
        } else {
            uint32_t h = uy * kPrimeY;
            if constexpr (DIM == 3) h ^= uz * kPrimeZ;
            valid[q] = true;
            bucket[q] = (h & mask) >> bl.shift;
        }
    }
}

// 2-D compact levels: the line that picks a sample's bucket. Base cells sit on lines [0, res - 2]; a coordinate clamped onto
// the last line (y == hi == res - 1 exactly, possible from res = 513 on) stays in the last slab: its y + 1 corner has weight 0.
__device__ __forceinline__ uint32_t compact2d_line(uint32_t py, uint32_t res) { return py < res - 2u ? py : res - 2u; }

// 2-D compact item, first 8 bytes: local base row (13 bits) | valid (bit 13) | fx and fy as 25-bit fixed point, rounded to
// nearest: exact for fractions >= 0.5 (their ulp is 2^-24) and wherever the position is >= 2 (fraction = a multiple of 2^-22),
// within 2^-26 absolute otherwise -- below the 2^-25 rounding of the reference's own `1 - fx`. Not bit-identical to the
// pair-item path, far inside the 1e-5 bar.
__device__ __forceinline__ void pack_compact2d(uint32_t local, float fx, float fy, uint32_t &w0, uint32_t &w1) {
    uint32_t qx = __float2uint_rn(fx * 33554432.0f), qy = __float2uint_rn(fy * 33554432.0f);   // fractions are < 1
    qx = qx > 33554431u ? 33554431u : qx;    // (a fraction within 2^-26 of 1 rounds up to 2^25: clamped)
    qy = qy > 33554431u ? 33554431u : qy;
    w0 = (local & 0x1FFFu) | (1u << 13) | ((qx >> 7) << 14);
    w1 = ((qx & 127u) << 25) | qy;
}
__device__ __forceinline__ void unpack_compact2d(uint32_t w0, uint32_t w1, uint32_t &local, uint32_t &qx, uint32_t &qy) {
    local = w0 & 0x1FFFu;
    qx = ((w0 >> 14) << 7) | (w1 >> 25);
    qy = w1 & 0x1FFFFFFu;
}

// position along one axis only (axis_transform without the fractions)
__device__ __forceinline__ uint32_t axis_pos(double t, int32_t res, float hi) {
    float x = (float)((double)res * t);
    x = fmaxf(0.0f, fminf(hi, x));
    return (uint32_t)(int32_t)floorf(x);
}

// bucket counts of one (sample, binned level) into the level's LDS histogram -- the same buckets / validity as
// enumerate_pairs (hashed, dense) and the compact scatter path, from the dense per-level record
template <int DIM>
__device__ __forceinline__ void count_level(const double (&t)[DIM], const CountLevel cl, uint32_t mask, uint32_t *hist) {
    if (cl.kind == 1u) {
        if constexpr (DIM == 3) {
            const uint32_t pz = axis_pos(t[2], cl.res, cl.hi);
            const uint32_t b = cl.m_lo ? (__umul24(pz, cl.m_lo) >> 18) : pz / cl.m_hi;
            atomicAdd(hist + b, 2u);
        } else {   // 2-D compact: one 16-byte item per sample, bucket = slab of base lines (compact2d_line)
            const uint32_t py = compact2d_line(axis_pos(t[1], cl.res, cl.hi), (uint32_t)cl.res);
            const uint32_t b = cl.m_lo ? (__umul24(py, cl.m_lo) >> 18) : py / cl.m_hi;
            atomicAdd(hist + b, 1u);
        }
    } else if (cl.kind == 0u) {
        const uint32_t hy0 = axis_pos(t[1], cl.res, cl.hi) * kPrimeY, hy1 = hy0 + kPrimeY;
        if constexpr (DIM == 3) {
            const uint32_t hz0 = axis_pos(t[2], cl.res, cl.hi) * kPrimeZ, hz1 = hz0 + kPrimeZ;
            atomicAdd(hist + (((hy0 ^ hz0) & mask) >> cl.shift), 1u);
            atomicAdd(hist + (((hy0 ^ hz1) & mask) >> cl.shift), 1u);
            atomicAdd(hist + (((hy1 ^ hz0) & mask) >> cl.shift), 1u);
            atomicAdd(hist + (((hy1 ^ hz1) & mask) >> cl.shift), 1u);
        } else {
            atomicAdd(hist + ((hy0 & mask) >> cl.shift), 1u);
            atomicAdd(hist + ((hy1 & mask) >> cl.shift), 1u);
        }
    } else {
        const uint64_t magic = ((uint64_t)cl.m_hi << 32) | cl.m_lo;
        const uint32_t r = (uint32_t)cl.res;
        const uint32_t py = axis_pos(t[1], cl.res, cl.hi);
        uint32_t pz = 0;
        if constexpr (DIM == 3) pz = axis_pos(t[2], cl.res, cl.hi);
#pragma unroll
        for (int q = 0; q < (1 << (DIM - 1)); ++q) {
            const uint32_t uy = py + ((DIM == 3) ? ((q >> 1) & 1) : (q & 1));
            bool ok = uy < r;
            uint32_t line = uy;
            if constexpr (DIM == 3) {
                const uint32_t uz = pz + (q & 1);
                ok = ok && uz < r;
                line += uz * r;
            }
            if (ok) atomicAdd(hist + (uint32_t)(((uint64_t)line * magic) >> 40), 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------------ fixed point
// LDS integer atomics run 1.6x faster than ds_add_f64 (2.1-2.5 vs 1.3-1.4 T op/s, profiles/r01_microbench2), so the
// accumulator images hold 64-bit fixed-point numbers. Scale per level: gmax[l] = max |grad_output| over the level's
// columns (bit pattern of the float, gathered by pass T for free; integer max on the bits orders
// finite < inf < NaN). Every contribution is |g * weight| <= gmax < 2^e, so with scale 2^(headroom - e) a contribution
// stays below 2^headroom and n_max of them below 2^62: headroom = min(50, 62 - ceil(log2(n_max))). Conversion is one
// fp64 fma with the 1.5 * 2^52 constant (the integer appears in the low mantissa bits) -- exact to the scale's LSB, i.e.
// 2^-headroom relative to gmax (>= 41 bits here vs 24 of the reference's fp32 atomics) and order-independent.
// A level whose gmax is inf / NaN falls back to the fp64 image so that non-finite gradients propagate as before.
struct FxScale {
    double scale, inv;   // 2^k, 2^-k
    bool fixed;          // false: accumulate in fp64 (non-finite gradients)
};
__device__ __forceinline__ FxScale fx_scale_of(uint32_t gmax_bits, int headroom) {
    FxScale f;
    f.fixed = gmax_bits < 0x7F800000u;
    int e = (int)((gmax_bits >> 23) & 0xFFu) - 126;   // |g| < 2^e for normal floats; denormals / zero: e = -126
    if (e < -126) e = -126;
    const int k = headroom - e;
    f.scale = __longlong_as_double((long long)(1023 + k) << 52);
    f.inv = __longlong_as_double((long long)(1023 - k) << 52);
    return f;
}
__device__ __forceinline__ unsigned long long fx_encode(float c, double scale) {
    const double magic = 6755399441055744.0;   // 1.5 * 2^52
    return (unsigned long long)(__double_as_longlong(fma((double)c, scale, magic)) - __double_as_longlong(magic));
}
__device__ __forceinline__ float fx_decode(unsigned long long v, double inv) { return (float)((double)(long long)v * inv); }
static inline int fx_headroom(uint64_t n_max) {
    int bits = 0;
    while (((uint64_t)1 << bits) < n_max) ++bits;
    const int h = 62 - bits;
    return h > 50 ? 50 : (h < 24 ? 24 : h);
}


}  // namespace shacira
