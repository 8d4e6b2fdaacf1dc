// hashgrid_bwd_bin.hip -- backward of the hash-grid lookup WITHOUT scattered global atomics (gfx950).
//
// Why: on MI355X a scattered global float atomic costs one memory-side request (~21 G requests/s chip-wide,
// profiles/r01_microbench_rates.txt) and LDS ds_add_f32 runs at 0.33 op/clk/CU, while LDS ds_add_f64 sustains
// ~1.3 T adds/s (profiles/r01_microbench2_lds_gather.txt). The reference's design (one atomicAdd per corner per
// feature, hashgrid_interpolate_cuda.cu:212-221) therefore runs ~30x under the HBM roof here. This file replaces
// it by "partition, then accumulate on chip":
//
//   pass F  front       grad_output [N, L*F] -> gT [L][NP][F] with 16-byte accesses both ways, FUSED with the bucket
//                       counting of the same samples (front16_kernel): cnt[tile][bucket] rows + per-bucket totals
//   pass S  scan        one workgroup: exclusive scan of the totals -> first item of every bucket, the cursors the
//                       scatter pass reserves its runs from, and the consumer work list
//   pass B  bin         recompute the corners, stage the tile's items in LDS sorted by bucket, reserve each (tile, bucket)
//                       run with one returning atomic on the bucket's cursor, write it with coalesced 16-byte stores
//   pass C  consume     persistent workgroups fetch (bucket, chunk) units: accumulate the items into an LDS-resident
//                       64-bit fixed-point (or fp64) image of the bucket's rows, then write the rows out (plain coalesced
//                       stores when the bucket has a single unit, coalesced float atomics otherwise)
//   pass D  direct      levels whose rows fit one LDS image and that do not travel as compact items: no items at all
//   (pass T / pass A: the 8-byte transpose and the standalone counting pass remain for rows that are not whole 16-byte
//    vectors, sub-batches and level-range calls on staged gradients)
//
// A *bucket* is a range of <= BR consecutive rows of one level (BR*F*8 B = 128 KiB of LDS). An *item* is one
// x-pair of corners (x, x+1) at fixed (y[,z]) offsets: both rows always share a bucket (hashed levels: the rows
// differ only in the low bits x ^ (x+1); dense levels: buckets hold whole x-lines), so an item is 8 + 4F bytes:
//   { key = rowA | rowB << 13 | validA << 26 | validB << 27,  fx,  a_j = grad_j * w_rest }   (rows bucket-local)
// and the consumer adds a_j*(1-fx) to rowA and a_j*fx to rowB. The sum is kept in 64 bits and rounded once. fp16 tables
// carry half-precision payloads (8- / 16-byte items, ItemH / ItemH4 below).
//
// Results differ from the reference only by summation order / two fp32 roundings per term (the reference's own
// atomicAdd order is unspecified); tests hold them to 1e-5 relative against the fp64-accumulating oracle.
#include <mutex>

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
template <int DIM> struct TileOf { static constexpr int value = (DIM == 2) ? 2 * kTile : kTile; };
static inline int tile_samples(int dim) { return dim == 2 ? 2 * kTile : kTile; }
constexpr int kBinThreads = SHACIRA_KBIN;  // threads of passes A and B
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
    uint32_t chunk_min; // smallest unit size of the plan (sizes the unit list)
    uint32_t nbl;       // number of binned levels
    uint32_t blevel[SHACIRA_MAX_LODS];  // their level indices (grid.y of passes A/B); 32-bit = scalar loads
    uint32_t bstart[SHACIRA_MAX_LODS];  // first global bucket of binned level q (= lv[blevel[q]].bucket0)
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
template <int F, bool H> struct ItemSel { typedef Item<F> type; };
template <> struct ItemSel<2, true> { typedef ItemH type; };
template <> struct ItemSel<4, true> { typedef ItemH4 type; };

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
__device__ __forceinline__ void store_item_nt(ItemH4 *p, const ItemH4 &it) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    __builtin_memcpy(&v, &it, 16);
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
}
__device__ __forceinline__ void store_item_nt(ItemH *p, const ItemH &it) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    __builtin_memcpy(&v, &it, 8);
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p));
}

// One consumer work unit, written by the bucket scan: everything a consume workgroup needs in ONE 32-byte load (it used to
// chase unit -> bucket -> base / unit_first -> level through four dependent loads and a 15-step scalar search: ~8 us per
// unit before the first item arrived).
struct alignas(16) UnitDesc {
    uint64_t begin, end;   // item range
    uint32_t bucket;       // global bucket index
    uint32_t level;
    uint32_t single;       // 1: the bucket's only unit (rows are written with plain stores)
    uint32_t pad;
};

// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains the wave's global loads AND stores
// (s_waitcnt vmcnt(0)), which serialises a block's write-out with its next phase.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Items are written once and read once: stream them past the caches (non-temporal).
template <int F> __device__ __forceinline__ void store_item_nt(Item<F> *p, const Item<F> &it) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (sizeof(Item<F>) == 16) {
        u32x4 v;
        __builtin_memcpy(&v, &it, 16);
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    } else {   // 24-byte items (F = 4, 8-byte aligned): three 8-byte stores instead of six dwords
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        static_assert(sizeof(Item<F>) % 8 == 0, "item size");
        u32x2 *q = reinterpret_cast<u32x2 *>(p);
        u32x2 d[sizeof(Item<F>) / 8];
        __builtin_memcpy(d, &it, sizeof(Item<F>));
#pragma unroll
        for (int k = 0; k < (int)(sizeof(Item<F>) / 8); ++k) __builtin_nontemporal_store(d[k], q + k);
    }
}

template <int F> __device__ __forceinline__ Item<F> load_item_nt(const Item<F> *p) {
    Item<F> it;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (sizeof(Item<F>) == 16) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        __builtin_memcpy(&it, &v, 16);
    } else {   // 24-byte items: three 8-byte loads
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 *q = reinterpret_cast<const u32x2 *>(p);
        u32x2 d[sizeof(Item<F>) / 8];
#pragma unroll
        for (int k = 0; k < (int)(sizeof(Item<F>) / 8); ++k) d[k] = __builtin_nontemporal_load(q + k);
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
            uint32_t h = uy * kPrimeY;
            if constexpr (DIM == 3) h ^= uz * kPrimeZ;
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
template <int DIM, typename T, int F, bool GMAX, bool COUNT>
__global__ __launch_bounds__(kFrontThreads) void front16_kernel(LevelTable lt, BinPlan plan, const T *__restrict__ go,
                                                                float *__restrict__ gT, const float *__restrict__ coords,
                                                                uint32_t *__restrict__ totals, uint32_t *__restrict__ cnt,
                                                                int64_t N, int64_t NP, int lb, int le, int ts_log2,
                                                                int rounds, uint32_t *__restrict__ gmax) {
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
    int hcol[HE];
    uint32_t hsum[HE];
    if constexpr (COUNT) {
#pragma unroll
        for (int j = 0; j < HE; ++j) {
            const uint32_t k = threadIdx.x + j * kFrontThreads, li = k / kMaxLevelBuckets, b = k % kMaxLevelBuckets;
            hcol[j] = (li < plan.nbl && b < plan.lv[plan.blevel[li < plan.nbl ? li : 0]].nb) ? (int)(plan.bstart[li] + b) : -1;
            hsum[j] = 0;
        }
    }
    for (int r = 0; r < rounds && tile0 + r < tiles; ++r) {
        const int64_t s0 = (tile0 + r) << ts_log2;
        const int ns = (int)((N - s0 < TS) ? (N - s0) : TS);
        const int total = ns * VPR;
        // the sample's coordinates FIRST (vmcnt counts in order: the counting then waits for them only), then the row
        // vectors; all loads unconditional with clamped indices (a branch around a load makes the compiler wait with
        // vmcnt(0) in front of the counting, which would serialise it with the row loads)
        float cc[DIM];
        if constexpr (COUNT) {
            const int64_t ci = s0 + (csm < ns ? csm : ns - 1);
#pragma unroll
            for (int a = 0; a < DIM; ++a) cc[a] = coords[ci * DIM + a];
        }
        const u32x4 *in = reinterpret_cast<const u32x4 *>(go) + s0 * VPR;
        u32x4 raw[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int e = (int)threadIdx.x + u * kFrontThreads;
            raw[u] = __builtin_nontemporal_load(in + (e < total ? e : total - 1));   // idle lanes: one merged request
        }
        if constexpr (COUNT) {
            if (csm < ns) {
                double t[DIM];
#pragma unroll
                for (int a = 0; a < DIM; ++a) t[a] = axis_unit(cc[a]);
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
            park1(__builtin_nontemporal_load(in + e), sm, v);
            sm += q512;
            v += r512;
            if (v >= VPR) { v -= VPR; ++sm; }
        }
        lds_barrier();
        if constexpr (COUNT) {
            uint32_t *row = cnt + (size_t)(tile0 + r) * plan.total_buckets;
#pragma unroll
            for (int j = 0; j < HE; ++j) {
                if (hcol[j] >= 0) {
                    const uint32_t h = s_hist[threadIdx.x + j * kFrontThreads];
                    s_hist[threadIdx.x + j * kFrontThreads] = 0;
                    row[hcol[j]] = h;
                    hsum[j] += h;
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
                    __builtin_nontemporal_store(val, reinterpret_cast<f32x4 *>(dst));
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
        for (int j = 0; j < HE; ++j)
            if (hcol[j] >= 0 && hsum[j]) atomicAdd(&mine[hcol[j]], hsum[j]);
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
                                                                unsigned long long *__restrict__ cursor) {
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
        uint32_t lq = 0;
        for (uint32_t q = 1; q < plan.nbl; ++q)
            if (plan.bstart[q] <= b) lq = q;
        lv_of[k] = plan.blevel[lq];
        ck[k] = plan.lv[lv_of[k]].chunk;
        u[k] = (uint32_t)((c[k] + ck[k] - 1) / ck[k]);
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
            for (uint32_t q = 0; q < u[k]; ++q) {
                UnitDesc d;
                d.begin = s_items[b] + (uint64_t)q * ck[k];
                const uint64_t bucket_end = s_items[b] + c[k];
                d.end = (d.begin + ck[k] < bucket_end) ? (d.begin + ck[k]) : bucket_end;
                d.bucket = b;
                d.level = lvl;
                d.single = u[k] == 1 ? 1u : 0u;
                d.pad = 0;
                unit_desc[s_units[b] + q] = d;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- pass B
// Gradients from the transposed image gT [L][NP][F], grid (tiles, binned levels). A (tile, bucket) run is reserved with one
// returning atomic on the bucket's cursor (set to the bucket's base by the bucket scan): runs of different tiles land in
// the bucket in arrival order -- the consumer's fixed-point sums do not depend on it.
// H: half-precision item stream (fp16 tables, F = 2): 8-byte pair items, 16-byte compact items.
template <int DIM, int F, bool H>
__global__ __launch_bounds__(kBinThreads) void bin_scatter_kernel(LevelTable lt, BinPlan plan,
                                                                  const float *__restrict__ coords,
                                                                  const float *__restrict__ gT,
                                                                  unsigned long long *__restrict__ cursor,
                                                                  const uint32_t *__restrict__ cnt, uint32_t cps,
                                                                  uint32_t cnt_rows,
                                                                  typename ItemSel<F, H>::type *__restrict__ items,
                                                                  int64_t sample0, int64_t N, int64_t gpitch) {
    typedef typename ItemSel<F, H>::type ItemT;
    constexpr int NP = 1 << (DIM - 1);
    constexpr int kTileD = TileOf<DIM>::value;
    constexpr int SPT = kTileD / kBinThreads;  // samples per thread
    constexpr int kStage = kTileD * NP;        // staged items per block
    extern __shared__ __align__(16) unsigned char s_raw[];
    ItemT *s_items = reinterpret_cast<ItemT *>(s_raw);
    uint8_t *s_bucket = reinterpret_cast<uint8_t *>(s_items + kStage);
    __shared__ uint32_t s_hist[kMaxLevelBuckets];
    __shared__ uint32_t s_start[kMaxLevelBuckets + 1];
    __shared__ uint64_t s_gbase[kMaxLevelBuckets];

    // (tile-fastest numbering; level-fastest -- a tile's levels back to back -- measured 3 % slower, round 3)
    const uint32_t tile = blockIdx.x, bi = blockIdx.y;
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
    const bool compact = (DIM == 3) && bl.compact != 0;
    // every global load of the workgroup up front, unconditional (indices clamped into the batch): coordinates and
    // gradients of the thread's samples, then -- waves 1 and 2 -- the tile's bucket counts (rows of cnt[tile][bucket]
    // written by the counting pass) and straight away the returning atomic that reserves the bucket's run: it is the
    // YOUNGEST memory operation of the wave, so nothing below waits for it until the run offsets are needed (after the
    // staging phase); issued after the ranking instead, its round trip cost 23 us on S1
    float craw[SPT][DIM], graw[SPT][F];
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        int64_t i = sample0 + (int64_t)tile * kTileD + threadIdx.x + u * kBinThreads;
        i = i < N ? i : N - 1;
#pragma unroll
        for (int a = 0; a < DIM; ++a) craw[u][a] = coords[i * DIM + a];
        const float *gp = gT + ((int64_t)lvl * gpitch + i) * F;
        if constexpr (F == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(gp);
            graw[u][0] = v.x; graw[u][1] = v.y;
        } else {
#pragma unroll
            for (int j = 0; j < F; ++j) graw[u][j] = gp[j];
        }
    }
    const bool reserver = threadIdx.x >= 64 && threadIdx.x - 64 < bl.nb;
    unsigned long long run_base = 0ull;
    if (reserver) {
        const uint32_t gb = bl.bucket0 + threadIdx.x - 64;
        const uint32_t *row = cnt + (size_t)tile * cps * plan.total_buckets + gb;
        uint32_t c = 0;
        for (uint32_t k = 0; k < cps && tile * cps + k < cnt_rows; ++k) c += row[(size_t)k * plan.total_buckets];
        if (c) run_base = atomicAdd(&cursor[gb], (unsigned long long)c);
    }
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const int k = threadIdx.x + u * kBinThreads;
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
            }
        } else {
            enumerate_pairs<DIM>(t, res, hi, dense, lt.mask, bl, plan.BR, fx[u], ps[u]);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (!live) ps[u][q].key = 0;
            rank[u][q] = (ps[u][q].key >> 26) ? atomicAdd(&s_hist[ps[u][q].bucket], compact ? 2u : 1u) : 0u;
        }
    }
    lds_barrier();   // (not __syncthreads(): its vmcnt(0) would wait for the reservation)
    if (threadIdx.x < 64) {  // wave 0: exclusive scan of the <= 128 bucket counts, two per lane
        const uint32_t lane = threadIdx.x;
        const uint32_t c0 = (2 * lane < bl.nb) ? s_hist[2 * lane] : 0u;
        const uint32_t c1 = (2 * lane + 1 < bl.nb) ? s_hist[2 * lane + 1] : 0u;
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
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (ps[u][q].key >> 26) {
                const uint32_t pos = s_start[ps[u][q].bucket] + rank[u][q];
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
                Item<F> it;
                it.key = ps[u][q].key;
                it.fx = fx[u];
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
    if (reserver) s_gbase[threadIdx.x - 64] = run_base;
    __syncthreads();
    const uint32_t staged = s_start[bl.nb];
    for (uint32_t pos = threadIdx.x; pos < staged; pos += kBinThreads) {
        const uint32_t b = s_bucket[pos];
        // write-once / read-once stream: non-temporal stores (measured -7 % on the whole backward)
        store_item_nt(items + s_gbase[b] + (pos - s_start[b]), s_items[pos]);
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
            if constexpr (DIM == 3) {
                if (bl.compact) {
                    int32_t pz;
                    float fz, gz;
                    axis_transform(t[u][2], res, hi, pz, fz, gz);
                    atomicAdd(&s_hist[bi][(uint32_t)pz / bl.slab], 2u);
                    continue;
                }
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
            cnt[(size_t)tile * plan.total_buckets + bl.bucket0 + b] = s_hist[bi][b];
            if (s_hist[bi][b])
                atomicAdd(&totals[(size_t)((blockIdx.x + blockIdx.y) % kTotalShards) * kMaxBuckets + bl.bucket0 + b], s_hist[bi][b]);
        }
    }
}

// ------------------------------------------------------------------------------------------------- table zeroing
// at::zeros_like of the reference, minus what the consume pass overwrites anyway: the rows of a HASHED binned level are
// covered by its buckets, and a bucket with exactly one work unit writes all its rows with plain stores. So only the
// other rows are zeroed up front (S1: 6.6 of 48.8 MB; the table-sized memset was 13 of config D's 93 us and 40 MB of the
// write-bound traffic of every call) and the buckets that turn out to have 0 or several units are zeroed once the bucket
// scan knows them. grid (x, num_lods): segment l = rows [first_idx[l], first_idx[l + 1]) (segment 0 starts at row 0).
__global__ __launch_bounds__(256) void zero_unowned_rows_kernel(float *__restrict__ acc,
                                                                const int32_t *__restrict__ first_idx, LevelTable lt,
                                                                BinPlan plan) {
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

// after the bucket scan: hashed buckets with 0 units (never written) or several (they add atomically) are zeroed now.
// grid (kMaxLevelBuckets, nbl)
__global__ __launch_bounds__(256) void zero_odd_buckets_kernel(float *__restrict__ acc, const int32_t *__restrict__ first_idx,
                                                               const uint32_t *__restrict__ unit_first, LevelTable lt,
                                                               BinPlan plan) {
    const uint32_t lvl = plan.blevel[blockIdx.y];
    const BinLevel bl = plan.lv[lvl];
    const uint32_t b = blockIdx.x;
    if (b >= bl.nb || lt.dense[lvl] != 0) return;
    const uint32_t gb = bl.bucket0 + b;
    if (unit_first[gb + 1] - unit_first[gb] == 1u) return;
    const uint32_t row0 = b * bl.rows_pb;
    const uint32_t nrows = (bl.used - row0 < bl.rows_pb) ? (bl.used - row0) : bl.rows_pb;
    float *dst = acc + ((int64_t)first_idx[lvl] + row0) * lt.feature_dim;
    for (uint32_t e = threadIdx.x; e < nrows * (uint32_t)lt.feature_dim; e += 256) dst[e] = 0.0f;
}

// a[j] for a lane-dependent j without a scratch array (select chain)
template <int F> __device__ __forceinline__ float pick(const float (&a)[F], int j) {
    float v = a[0];
#pragma unroll
    for (int k = 1; k < F; ++k) v = (j == k) ? a[k] : v;
    return v;
}

// ------------------------------------------------------------------------------------------------- pass C
// one work unit (a bucket, or a chunk of an over-full one) on the calling workgroup
template <int F, bool FX, bool H>
__device__ __forceinline__ void consume_unit(const LevelTable &lt, const BinPlan &plan, const int32_t *__restrict__ first_idx,
                                             const UnitDesc d, const typename ItemSel<F, H>::type *__restrict__ items,
                                             float *__restrict__ grad_table, int force_atomic,
                                             const uint32_t *__restrict__ gmax, int headroom, double *s_acc) {
    const uint32_t gb = d.bucket, lvl = d.level;
    const BinLevel bl = plan.lv[lvl];
    const uint32_t b = gb - bl.bucket0;
    const uint32_t r1 = (uint32_t)lt.res[lvl];
    // compact levels: the image starts at the bucket's first base plane and includes one halo plane
    const uint32_t row0 = bl.compact ? b * bl.slab * r1 * r1 : b * bl.rows_pb;
    const uint32_t nrows = (bl.used - row0 < bl.rows_pb) ? (bl.used - row0) : bl.rows_pb;

    for (uint32_t e = threadIdx.x; e < nrows * F; e += kConsumeThreads) s_acc[e] = 0.0;   // all-zero bits either way
    lds_barrier();
    FxScale fx{1.0, 1.0, false};
    if constexpr (FX) fx = fx_scale_of(gmax[lvl], headroom);
    unsigned long long *s_fix = reinterpret_cast<unsigned long long *>(s_acc);

    const uint64_t begin = d.begin, end = d.end;
    const int rot = (int)(threadIdx.x & (F - 1));
    // feature order rotated by lane: the F slots of a row are consecutive 8-byte words, so with every lane adding feature j
    // in the same instruction only 1 / F of the LDS banks were addressed (half of the pass's LDS cycles were bank conflicts)
    auto add_row = [&](uint32_t row, const float (&v)[F], float w) {
#pragma unroll
        for (int jj = 0; jj < F; ++jj) {
            const int j = (jj + rot) & (F - 1);
            const float c = pick<F>(v, j) * w;
            if (FX && fx.fixed) atomicAdd(&s_fix[row * F + j], fx_encode(c, fx.scale));
            else atomicAdd(&s_acc[row * F + j], (double)c);
        }
    };
    auto add_pair = [&](uint32_t ra, uint32_t rb, bool va, bool vb, float fxv, const float (&a)[F]) {
        if (va) add_row(ra, a, 1.0f - fxv);
        if (vb) add_row(rb, a, fxv);
    };
    const uint32_t r2 = r1 * r1;
    auto add_compact = [&](uint32_t base_row, float fxx, float fyy, float fzz, const float (&gg)[F]) {
        const float gxx = 1.0f - fxx, gyy = 1.0f - fyy, gzz = 1.0f - fzz;
        const float wxy[4] = {gxx * gyy, gxx * fyy, fxx * gyy, fxx * fyy};   // reference order: (x * y) * z
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t row = base_row + ((c >> 2) & 1) + ((c >> 1) & 1) * r1 + (c & 1) * r2;
            if (row >= nrows) continue;   // cannot happen for in-range cells; keeps the image safe
            add_row(row, gg, wxy[c >> 1] * ((c & 1) ? fzz : gzz));
        }
    };
    if constexpr (F == 2 || F == 4) {
        if (bl.compact) {
            constexpr int UC = 2;
            if constexpr (H && F == 4) {
                // one sample per two 16-byte units: {local base row | valid, fx, fy, fz (fp32)} {half2 g01, half2 g23, -, -}
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                for (uint64_t p0 = begin + 2ull * threadIdx.x; p0 < end; p0 += 2ull * kConsumeThreads * UC) {
                    u32x4 va[UC], vb2[UC];
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
#pragma unroll
                    for (int u = 0; u < UC; ++u) {
                        if (!(va[u][0] & (1u << 26))) continue;
                        const float2 g01 = half2_bits_to_float2(vb2[u][2]);
                        const float2 g23 = half2_bits_to_float2(vb2[u][3]);
                        const float gg[F] = {g01.x, g01.y, g23.x, g23.y};
                        add_compact(va[u][0] & 0x1FFFu, __uint_as_float(va[u][1]), __uint_as_float(va[u][2]),
                                    __uint_as_float(va[u][3]), gg);
                    }
                }
            } else if constexpr (H) {
                // one sample per 16-byte record (two 8-byte units): {local base row | valid, fx, fy, fz (u16), half2 g}
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                for (uint64_t p0 = begin + 2ull * threadIdx.x; p0 < end; p0 += 2ull * kConsumeThreads * UC) {
                    ItemHC rec[UC];
#pragma unroll
                    for (int u = 0; u < UC; ++u) {
                        const uint64_t p = p0 + 2ull * u * kConsumeThreads;
                        rec[u].key = 0;
                        if (p + 1 < end) {
                            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + p));
                            __builtin_memcpy(&rec[u], &v, 16);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UC; ++u) {
                        if (!(rec[u].key & (1u << 26))) continue;
                        const float2 gf = __half22float2(rec[u].g);
                        const float gg[F] = {gf.x, gf.y};
                        const float q = 1.0f / 65536.0f;
                        add_compact(rec[u].key & 0x1FFFu, ((float)rec[u].fx + 0.5f) * q, ((float)rec[u].fy + 0.5f) * q,
                                    ((float)rec[u].fz + 0.5f) * q, gg);
                    }
                }
            } else {
                // one sample per two slots: F = 2 {local base row | valid, fx, fy, fz} {-, g0, g1, -}; F = 4 {.., fx, fy, fz,
                // g0, g1} {-, g2, g3, ...}; all 8 corners land here
                const Item<F> *itf = reinterpret_cast<const Item<F> *>(items);
                for (uint64_t p0 = begin + 2ull * threadIdx.x; p0 < end; p0 += 2ull * kConsumeThreads * UC) {
                    Item<F> ia[UC], ib[UC];
#pragma unroll
                    for (int u = 0; u < UC; ++u) {
                        const uint64_t p = p0 + 2ull * u * kConsumeThreads;
                        if (p + 1 < end) {
                            ia[u] = load_item_nt<F>(itf + p);
                            ib[u] = load_item_nt<F>(itf + p + 1);
                        } else {
                            ia[u].key = 0;
                        }
                    }
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
                }
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
    if constexpr (H && F == 4) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        for (uint64_t p0 = begin + threadIdx.x; p0 < end; p0 += (uint64_t)kConsumeThreads * UN) {
            u32x4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint64_t pp = p0 + (uint64_t)u * kConsumeThreads;
                v[u] = u32x4{0u, 0u, 0u, 0u};
                if (pp < end) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + pp));
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float2 a01 = half2_bits_to_float2(v[u][2]);
                const float2 a23 = half2_bits_to_float2(v[u][3]);
                const float a[F] = {a01.x, a01.y, a23.x, a23.y};
                add_pair(v[u][0] & 0x1FFFu, (v[u][0] >> 13) & 0x1FFFu, (v[u][0] >> 26) & 1u, (v[u][0] >> 27) & 1u,
                         __uint_as_float(v[u][1]), a);
            }
        }
    } else if constexpr (H) {
        // 8-byte items read two at a time (16-byte loads from even unit indices); a unit's odd first / last item goes alone
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
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
        if ((p & 1ull) && p < end) {
            if (threadIdx.x == 0) {
                const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(items + p));
                consume8(v[0], v[1]);
            }
            ++p;
        }
        const uint64_t even_end = end & ~1ull;
        for (uint64_t p0 = p + 2ull * threadIdx.x; p0 < even_end; p0 += 2ull * kConsumeThreads * UN) {
            u32x4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint64_t q = p0 + 2ull * u * kConsumeThreads;
                v[u] = u32x4{0u, 0u, 0u, 0u};   // key 0: no valid corner
                if (q < even_end) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + q));
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                consume8(v[u][0], v[u][1]);
                consume8(v[u][2], v[u][3]);
            }
        }
        if ((end & 1ull) && end - 1 >= p && threadIdx.x == 64) {
            const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(items + end - 1));
            consume8(v[0], v[1]);
        }
    } else {
        const Item<F> *itf = reinterpret_cast<const Item<F> *>(items);
        for (uint64_t p0 = begin + threadIdx.x; p0 < end; p0 += (uint64_t)kConsumeThreads * UN) {
            Item<F> it[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint64_t pp = p0 + (uint64_t)u * kConsumeThreads;
                if (pp < end) it[u] = load_item_nt<F>(itf + pp);
                else it[u].key = 0;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                add_pair(it[u].key & 0x1FFFu, (it[u].key >> 13) & 0x1FFFu, (it[u].key >> 26) & 1u, (it[u].key >> 27) & 1u,
                         it[u].fx, it[u].a);
        }
    }
    lds_barrier();

    const bool single = d.single != 0 && !force_atomic;
    const int64_t grow0 = (int64_t)first_idx[lvl] + row0;
    for (uint32_t e = threadIdx.x; e < nrows * F; e += kConsumeThreads) {
        const int64_t grow = grow0 + e / F;
        if ((uint64_t)grow >= (uint64_t)lt.table_rows) continue;
        const float v = (FX && fx.fixed) ? fx_decode(s_fix[e], fx.inv) : (float)s_acc[e];
        float *dst = grad_table + grow * F + (e % F);
        if (single) *dst = v;
        else if (v != 0.0f) unsafeAtomicAdd(dst, v);
    }
}


// Persistent form: `work_counter` non-NULL -> every workgroup keeps fetching units from it until they run out (grid = the
// number of workgroups the chip holds, not the number of units). A unit of a small batch is ~10 us of work between a launch,
// a 128 KiB image to zero and a flush whose stores s_endpgm would wait for: as separate workgroups (one per CU at a time)
// nerf_lego.yaml's 1 800 units of 8 K items took 228 us; here the flush of unit k drains behind unit k + 1 (all barriers in
// consume_unit are LDS-only). `work_counter` NULL: one unit per workgroup (small batches).
template <int F, bool FX, bool H>
__global__ __launch_bounds__(kConsumeThreads) void bin_consume_kernel(LevelTable lt, BinPlan plan,
                                                                      const int32_t *__restrict__ first_idx,
                                                                      const uint64_t *__restrict__ base,
                                                                      const uint32_t *__restrict__ unit_first,
                                                                      const UnitDesc *__restrict__ unit_desc,
                                                                      const typename ItemSel<F, H>::type *__restrict__ items,
                                                                      float *__restrict__ grad_table,
                                                                      int force_atomic,
                                                                      const uint32_t *__restrict__ gmax,
                                                                      int headroom,
                                                                      uint32_t *__restrict__ work_counter) {
    extern __shared__ double s_acc[];  // [rows_pb][F]: fp64, or 64-bit fixed point (same size)
    __shared__ uint32_t s_unit;
    const uint32_t unit0 = 0u, unit_end = unit_first[plan.total_buckets];
    if (work_counter == nullptr) {
        const uint32_t unit = blockIdx.x + unit0;
        if (unit >= unit_end) return;
        consume_unit<F, FX, H>(lt, plan, first_idx, unit_desc[unit], items, grad_table, force_atomic, gmax, headroom, s_acc);
        return;
    }
    for (;;) {
        if (threadIdx.x == 0) s_unit = atomicAdd(work_counter, 1u);
        lds_barrier();
        const uint32_t unit = s_unit + unit0;
        if (unit >= unit_end) return;
        consume_unit<F, FX, H>(lt, plan, first_idx, unit_desc[unit], items, grad_table, force_atomic, gmax, headroom, s_acc);
        lds_barrier();   // the image and s_unit are free again; the flush stores keep draining
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
                                                                            int headroom) {
    constexpr int NC = 1 << DIM;
    extern __shared__ double s_acc[];
    __shared__ double s_scale[SHACIRA_MAX_LODS], s_inv[SHACIRA_MAX_LODS];
    __shared__ int s_all_fixed;
    const uint32_t grp = blockIdx.y;
    const uint32_t rows = plan.grows[grp];
    const uint32_t mask = plan.gmask[grp];
    if (threadIdx.x == 0) s_all_fixed = 1;
    for (uint32_t e = threadIdx.x; e < rows * F; e += kConsumeThreads) s_acc[e] = 0.0;
    __syncthreads();
    if constexpr (FX) {
        if ((int)threadIdx.x < lt.num_lods && ((mask >> threadIdx.x) & 1u)) {
            const FxScale f = fx_scale_of(gmax[threadIdx.x], headroom);
            s_scale[threadIdx.x] = f.scale;
            s_inv[threadIdx.x] = f.inv;
            if (!f.fixed) s_all_fixed = 0;     // one non-finite level: the whole group accumulates in fp64
        }
        __syncthreads();
    }
    const bool fixed = FX && s_all_fixed != 0;
    unsigned long long *s_fix = reinterpret_cast<unsigned long long *>(s_acc);
    const int64_t stride = (int64_t)gridDim.x * kConsumeThreads;
    const int rotd = (int)(threadIdx.x & (F - 1));
    for (int64_t i = (int64_t)blockIdx.x * kConsumeThreads + threadIdx.x; i < N; i += stride) {
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
        for (int l = 0; l < lt.num_lods; ++l) {
            if (!((mask >> l) & 1u)) continue;
            const BinLevel bl = plan.lv[l];
            Corners<DIM> c;
            compute_corners<DIM>(t, lt.res[l], lt.hi[l], lt.dense[l] != 0, lt.mask, c);
            const GT *gp = TRANSPOSED ? gT + ((int64_t)l * gpitch + i) * F : gT + (i * lt.num_lods + l) * F;
            float g[F];
#pragma unroll
            for (int j = 0; j < F; ++j) g[j] = Scalar<GT>::load(gp + j);
            const double scale = FX ? s_scale[l] : 1.0;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (c.row[k] < bl.used) {
                    const size_t slot = (size_t)(bl.drow0 + c.row[k]) * F;
                    if (fixed) {
#pragma unroll
                        for (int jj = 0; jj < F; ++jj) {   // feature order rotated by lane (LDS bank spreading)
                            const int j = (jj + rotd) & (F - 1);
                            atomicAdd(s_fix + slot + j, fx_encode(pick<F>(g, j) * c.w[k], scale));
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < F; ++jj) {
                            const int j = (jj + rotd) & (F - 1);
                            atomicAdd(s_acc + slot + j, (double)(pick<F>(g, j) * c.w[k]));
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

// ------------------------------------------------------------------------------------------------- host side
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// `one_image_compact`: a dense 3-D level that fits ONE image travels as compact items too (one bucket whose units flush
// atomically) instead of the direct pass in front of the scatter pass; decided per CALL from the total batch (below), so
// that every plan of a call classifies the levels alike
static void make_plan(int dim, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib, bool one_image_compact);
static inline bool one_image_compact_rule(int64_t n_total) { return n_total >= ((int64_t)1 << 17); }

// can the table be partitioned with an LDS accumulator image of `acc_kib` KiB per consumer workgroup?
static bool bin_feasible(int dim, const LevelTable &lt, int acc_kib) {
    const int F = lt.feature_dim;
    if (F != 2 && F != 4) return false;
    const uint32_t BR = (uint32_t)acc_kib * 128u / (uint32_t)F;  // rows of the fp64 image
    for (int l = 0; l < lt.num_lods; ++l) {
        const uint32_t res = (uint32_t)lt.res[l];
        if (lt.dense[l]) {
            if (res > BR) return false;  // a bucket must hold at least one x-line
        } else {
            // x ^ (x+1) must stay below BR so that both rows of a pair share a bucket
            uint32_t bits = 0;
            while ((1u << bits) <= res) ++bits;
            if ((1u << bits) > BR && (lt.mask + 1u) > BR) return false;
        }
    }
    BinPlan plan;
    make_plan(dim, lt, kTile, plan, acc_kib, false);
    if (plan.total_buckets + (uint32_t)lt.num_lods > (uint32_t)kMaxBuckets) return false;   // (+ one-image compact levels)
    for (int l = 0; l < lt.num_lods; ++l)
        if (plan.lv[l].nb > (uint32_t)kMaxLevelBuckets) return false;
    return true;
}

// Image size per call, from the TOTAL batch (one choice per call so that every plan of the call classifies the levels
// alike). Option "bin_acc_kib": 64 / 128 force it, 0 (default) = measured rule: 64 KiB images (two consumer workgroups
// per CU overlap their zero / stream / flush phases) win up to 2^19 3-D samples, 128 KiB (half as many buckets) beyond.
static int choose_acc_kib(int dim, const LevelTable &lt, int64_t n) {
    const int kib = opt().bin_acc_kib;
    if (kib != 0) return kib;
    const int64_t pairs = (int64_t)1 << (dim - 1);
    if (n * pairs > ((int64_t)1 << 21) || !bin_feasible(dim, lt, 64)) return 128;
    // tables whose levels are all "direct" (config B: every level fits an LDS image) want the big image: fewer level
    // groups, hence fewer walks over the samples (measured 82 vs 124 us on the 393 216-pixel batch)
    BinPlan big;
    make_plan(dim, lt, kTile, big, 128, false);
    return big.nbl == 0 ? 128 : 64;
}

bool bin_supported(int dim, const LevelTable &lt) {
    const int kib = opt().bin_acc_kib;
    return bin_feasible(dim, lt, kib ? kib : 128);
}

static void make_plan(int dim, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib, bool one_image_compact) {
    if (one_image_compact) {   // tables whose levels are ALL direct stay that way (no transposing pass at all)
        make_plan(dim, lt, n_batch, plan, acc_kib, false);
        if (plan.nbl == 0) return;
    }
    const int F = lt.feature_dim;
    const uint32_t BR = (uint32_t)acc_kib * 128u / (uint32_t)F;
    uint32_t shift = 0;
    while ((1u << shift) < BR) ++shift;
    uint32_t nbk = 0;
    plan.nbl = 0;
    plan.ngroups = 0;
    for (int l = 0; l < lt.num_lods; ++l) {
        BinLevel &bl = plan.lv[l];
        const uint64_t res = (uint64_t)lt.res[l];
        bl.bucket0 = nbk;
        bl.shift = shift;
        bl.dgroup = -1;
        bl.drow0 = 0;
        bl.compact = 0;
        bl.slab = 0;
        if (l < lt.level_begin || l >= lt.level_end) {  // not part of this call
            bl.nb = 0;
            bl.used = 0;
            bl.rows_pb = 0;
            bl.G = 1;
            bl.magicG = 0;
            bl.bucket0 = 0xFFFFFFFFu;
            continue;
        }
        if (lt.dense[l]) {
            const uint64_t lines = (dim == 3) ? res * res : res;
            bl.used = (uint32_t)(lines * res);
            bl.G = (uint32_t)(BR / res);
            if (bl.G > lines) bl.G = (uint32_t)lines;
            bl.rows_pb = (uint32_t)(bl.G * res);
            bl.nb = (uint32_t)((lines + bl.G - 1) / bl.G);
            bl.magicG = (((uint64_t)1 << 40) + bl.G - 1) / bl.G;
            // Compact mode (3-D, F = 2): when an image holds at least two z-planes of the level, bucket = slab of base
            // cells in z and the image = that slab plus one halo plane, so all 8 corners of a sample land in ONE
            // bucket and the sample travels as one 32-byte item instead of four 16-byte pair items.
            const uint64_t planes = (dim == 3 && (F == 2 || F == 4) && opt().bwd_compact != 0) ? BR / (res * res) : 0;
            // (exp2: a level that fits ONE image may also travel as compact items -- one bucket, its units flush atomically --
            // instead of the direct pass that walks the whole batch in front of the scatter pass)
            if (planes >= 2 && res >= 3 && (bl.used > BR || one_image_compact)) {
                const uint32_t slab = (uint32_t)planes - 1;
                const uint32_t nbz = ((uint32_t)res - 2u) / slab + 1u;      // base cells: z in [0, res - 2]
                if (nbz <= (uint32_t)kMaxLevelBuckets) {
                    bl.compact = 1;
                    bl.slab = slab;
                    bl.rows_pb = (slab + 1u) * (uint32_t)(res * res);
                    bl.nb = nbz;
                }
            }
        } else {
            bl.used = lt.mask + 1u;
            bl.rows_pb = (bl.used < BR) ? bl.used : BR;
            bl.nb = (bl.used + BR - 1) / BR;
            bl.G = 1;
            bl.magicG = 0;
        }
        if (bl.nb == 1 && bl.used <= BR && !bl.compact) {
            // direct level: first group with room (greedy); groups hold <= BR rows
            uint32_t gi = 0;
            while (gi < plan.ngroups && plan.grows[gi] + bl.used > BR) ++gi;
            if (gi == plan.ngroups) {
                plan.gmask[gi] = 0;
                plan.grows[gi] = 0;
                ++plan.ngroups;
            }
            bl.dgroup = (int32_t)gi;
            bl.drow0 = plan.grows[gi];
            plan.grows[gi] += bl.used;
            plan.gmask[gi] |= 1u << l;
            bl.bucket0 = 0xFFFFFFFFu;
            bl.nb = 0;
        } else {
            plan.blevel[plan.nbl++] = (uint32_t)l;
            nbk += bl.nb;
        }
    }
    // Re-deal the direct levels over the same number of groups so that every group carries about the same NUMBER of
    // levels: a group's workgroups walk all samples once per level they own, so the greedy fill (6 / 4 / 4 / 2 levels on
    // the Kodak tables) left the slowest group with 1.5x the average work. Kept only if it fits the same group count.
    if (plan.ngroups > 1) {
        int dl[SHACIRA_MAX_LODS], ndl = 0;
        for (int l = 0; l < lt.num_lods; ++l)
            if (plan.lv[l].dgroup >= 0) dl[ndl++] = l;
        uint32_t nmask[SHACIRA_MAX_LODS] = {0}, nrows[SHACIRA_MAX_LODS] = {0}, row0[SHACIRA_MAX_LODS] = {0};
        int grp_of[SHACIRA_MAX_LODS];
        uint32_t g = 0;
        int in_group = 0, k = 0;
        bool ok = true;
        for (; k < ndl; ++k) {
            const uint32_t used = plan.lv[dl[k]].used;
            const int left_levels = ndl - k, left_groups = (int)plan.ngroups - (int)g;
            const int quota = (left_levels + in_group + left_groups - 1) / left_groups;   // ceil of what is left per group
            if (in_group > 0 && (nrows[g] + used > BR || in_group >= quota)) {
                ++g;
                in_group = 0;
                if (g >= plan.ngroups) { ok = false; break; }
            }
            if (nrows[g] + used > BR) { ok = false; break; }
            grp_of[k] = (int)g;
            row0[k] = nrows[g];
            nrows[g] += used;
            nmask[g] |= 1u << dl[k];
            ++in_group;
        }
        if (ok) {
            for (uint32_t q = 0; q < plan.ngroups; ++q) {
                plan.gmask[q] = nmask[q];
                plan.grows[q] = nrows[q];
            }
            for (int q = 0; q < ndl; ++q) {
                plan.lv[dl[q]].dgroup = grp_of[q];
                plan.lv[dl[q]].drow0 = row0[q];
            }
        }
    }
    for (uint32_t q = 0; q < plan.nbl; ++q) plan.bstart[q] = plan.lv[plan.blevel[q]].bucket0;
    plan.total_buckets = nbk;
    plan.BR = BR;
    plan.num_tiles = (uint32_t)((n_batch + tile_samples(dim) - 1) / tile_samples(dim));
    plan.pairs = 1u << (dim - 1);
    // work-unit size: 1/48 of ONE level's items, so that an evenly loaded hashed bucket (1/64 of a level) is ONE unit
    // (plain-store flush) with 33 % slack, while over-full coarse buckets split into equal chunks that keep all CUs busy
    uint64_t chunk = (uint64_t)n_batch * plan.pairs / 48 + 1024;
    if (chunk < 8192) chunk = 8192;
    if (chunk > (1u << 22)) chunk = 1u << 22;
    plan.chunk = (uint32_t)chunk & ~1u;   // even: a compact item (two 16-byte slots) never straddles two work units
    // Unit order = bucket order (dense compact levels first, then the hashed levels, coarse to fine) is the measured best for
    // the persistent consume pass: hashed levels first or reverse order cost +45 us on S1, smaller units for the dense levels
    // or for the last hashed levels changed nothing (round 3, tools/r3_ab.py).
    for (uint32_t q = 0; q < plan.nbl; ++q) {
        BinLevel &bl = plan.lv[plan.blevel[q]];
        bl.chunk = plan.chunk;
        CountLevel &c = plan.cl[q];
        c.res = lt.res[plan.blevel[q]];
        c.hi = lt.hi[plan.blevel[q]];
        c.shift = bl.shift;
        c.m_lo = (uint32_t)bl.magicG;
        c.m_hi = (uint32_t)(bl.magicG >> 32);
        c.kind = lt.dense[plan.blevel[q]] ? 2u : 0u;
        if (bl.compact) {
            const uint32_t m = ((1u << 18) + bl.slab - 1u) / bl.slab;
            bool exact = m < (1u << 18) || bl.slab == 1;
            for (uint32_t pz = 0; pz < (uint32_t)c.res && exact; ++pz) exact = ((pz * m) >> 18) == pz / bl.slab;
            c.kind = 1u;
            c.m_lo = exact ? m : 0u;   // 0: divide
            c.m_hi = bl.slab;
        }
    }
    plan.chunk_min = plan.chunk;
}

// sub-batch so that the item array stays below the cap (default 1.5 GiB, option "bin_batch_mib")
// bytes of one item unit: 8 + 4 F (fp32 payloads), 8 with the half-precision stream (fp16 tables, F = 2)
static inline bool half_items(int dtype, const LevelTable &lt) {
    return dtype == SHACIRA_F16 && (lt.feature_dim == 2 || lt.feature_dim == 4);
}
static inline size_t item_unit_bytes(int dtype, const LevelTable &lt) {
    if (half_items(dtype, lt)) return lt.feature_dim == 2 ? 8 : 16;
    return 8 + 4 * (size_t)lt.feature_dim;
}

static int64_t bin_batch_samples(int dim, int dtype, const LevelTable &lt, int64_t n) {
    const size_t item = item_unit_bytes(dtype, lt);
    BinPlan plan;
    make_plan(dim, lt, kTile, plan, choose_acc_kib(dim, lt, n), one_image_compact_rule(n));
    const size_t per_sample = (size_t)(plan.nbl ? plan.nbl : 1) * (1u << (dim - 1)) * item;
    int64_t cap = (int64_t)(((size_t)opt().bin_batch_mib << 20) / per_sample);
    cap = cap / tile_samples(dim) * tile_samples(dim);
    if (cap < tile_samples(dim)) cap = tile_samples(dim);
    return n < cap ? n : cap;
}

struct BinWorkspace {
    float *gT;                    // [L][NP][F] transposed gradients, NP = n rounded up to even
    unsigned char *items;
    uint32_t *totals;             // [kTotalShards][kMaxBuckets] items per bucket (global atomics of the counting pass)
    uint32_t *gmax;               // [SHACIRA_MAX_LODS] bit patterns of max |grad_output| per level (right behind totals)
    uint64_t *base;
    unsigned long long *cursor;   // [kMaxBuckets + 2] next free item slot of each bucket (scatter pass)
    uint32_t *cnt;                // [tiles of the counting pass][total_buckets] items per (tile, bucket)
    uint32_t *unit_first;
    UnitDesc *unit_desc;
    uint32_t *work_counter;       // next unit of the persistent consume pass (zeroed by the bucket scan)
    float *acc32;                 // fp32 accumulation image for fp16 tables
    size_t bytes;
};

static inline int64_t level_pitch(int64_t n) { return (n + 1) & ~(int64_t)1; }

static BinWorkspace carve(int dim, int dtype, const LevelTable &lt, int64_t n, void *ws) {
    BinPlan plan;
    const int64_t nb = bin_batch_samples(dim, dtype, lt, n);
    make_plan(dim, lt, nb, plan, choose_acc_kib(dim, lt, n), one_image_compact_rule(n));
    const size_t item = item_unit_bytes(dtype, lt);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // gT and gmax first: their offsets must not depend on the level range of the call (REUSE_STAGED calls share them)
    const size_t o_gT = take((size_t)level_pitch(n) * lt.num_lods * lt.feature_dim * sizeof(float));
    const size_t o_ctrl = take((size_t)(kTotalShards * kMaxBuckets + SHACIRA_MAX_LODS) * sizeof(uint32_t));   // totals | gmax: one memset
    const size_t o_items = take((size_t)nb * plan.nbl * plan.pairs * item);
    const size_t o_base = take((size_t)(kMaxBuckets + 2) * sizeof(uint64_t));
    const size_t o_cur = take((size_t)(kMaxBuckets + 2) * sizeof(uint64_t));
    const size_t o_cnt = take((size_t)(nb / 128 + 8) * plan.total_buckets * sizeof(uint32_t));   // smallest counting tile: 128
    const size_t o_unit = take((size_t)(kMaxBuckets + 2) * sizeof(uint32_t));
    const uint64_t max_items_ws = (uint64_t)nb * plan.nbl * plan.pairs;
    const size_t o_ub = take((size_t)(max_items_ws / plan.chunk_min + plan.total_buckets + 2) * sizeof(UnitDesc));
    const size_t o_wc = take(256);
    const size_t o_acc = take(dtype == SHACIRA_F16 ? (size_t)lt.table_rows * lt.feature_dim * sizeof(float) : 0);
    BinWorkspace w{};
    unsigned char *p = static_cast<unsigned char *>(ws);
    if (p) {
        w.gT = reinterpret_cast<float *>(p + o_gT);
        w.items = p + o_items;
        w.totals = reinterpret_cast<uint32_t *>(p + o_ctrl);
        w.gmax = w.totals + (size_t)kTotalShards * kMaxBuckets;
        w.base = reinterpret_cast<uint64_t *>(p + o_base);
        w.cursor = reinterpret_cast<unsigned long long *>(p + o_cur);
        w.cnt = reinterpret_cast<uint32_t *>(p + o_cnt);
        w.unit_first = reinterpret_cast<uint32_t *>(p + o_unit);
        w.unit_desc = reinterpret_cast<UnitDesc *>(p + o_ub);
        w.work_counter = reinterpret_cast<uint32_t *>(p + o_wc);
        w.acc32 = reinterpret_cast<float *>(p + o_acc);
    }
    w.bytes = off;
    return w;
}

size_t bin_workspace_bytes(int dim, int dtype, const LevelTable &lt, int64_t n) {
    return carve(dim, dtype, lt, n, nullptr).bytes;
}

float *bin_acc32(int dim, int dtype, const LevelTable &lt, int64_t n, void *workspace) {
    return carve(dim, dtype, lt, n, workspace).acc32;
}

#define SHACIRA_CHECK_LAUNCH()                 \
    do {                                       \
        hipError_t e_ = hipGetLastError();     \
        if (e_ != hipSuccess) return e_;       \
    } while (0)
#define SHACIRA_CHECK(expr)                    \
    do {                                       \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return e_;       \
    } while (0)

// Side stream for what does not sit on the critical path transpose -> bucket scan -> scatter -> consume: the table
// zeroing and the direct levels (LDS bound; they run beside the bucket scan and the write-bound scatter pass). One per host
// thread and device; fork/join with events keeps the caller's stream semantics (and is capturable in a HIP graph once the
// objects exist -- they are created on the first eager call).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, staged = nullptr;
};
static hipError_t side_stream(SideStream **out) {
    static thread_local SideStream per_device[kMaxDevices];
    int dev = 0;
    SHACIRA_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    SideStream &ss = per_device[dev];
    if (!ss.stream) {
        hipStream_t st;
        hipEvent_t a, b, c;
        SHACIRA_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        SHACIRA_CHECK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        SHACIRA_CHECK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        SHACIRA_CHECK(hipEventCreateWithFlags(&c, hipEventDisableTiming));
        ss.fork = a;
        ss.join = b;
        ss.staged = c;
        ss.stream = st;
    }
    *out = &ss;
    return hipSuccess;
}

// standalone pass A grid: tiles x level shares, at least ~1024 workgroups when the batch is small
static dim3 count_grid(const BinPlan &plan) {
    uint32_t shares = plan.num_tiles >= 1024 ? 1u : (1024u + plan.num_tiles - 1) / plan.num_tiles;
    if (shares > plan.nbl) shares = plan.nbl;
    return dim3(plan.num_tiles, shares < 1 ? 1 : shares);
}

// tile size of the 16-byte front kernel: the largest multiple of 128 samples (<= 512) whose staging image + bucket
// histograms leave room for two workgroups per CU
static int front_tile(int L, int F, uint32_t nbl, int64_t n, size_t *shmem) {
    int ts = 512;
    for (;;) {
        const size_t bytes = (size_t)L * (ts + 2) * F * sizeof(float) + (size_t)nbl * kMaxLevelBuckets * sizeof(uint32_t);
        // small batches: smaller tiles, so that at least ~512 workgroups share the counting (one sample per thread)
        const bool enough = (n + ts - 1) / ts >= 512;
        if ((bytes <= (size_t)78 * 1024 && enough) || ts == 128) {
            *shmem = bytes;
            return bytes <= (size_t)156 * 1024 ? ts : 0;
        }
        ts /= 2;
    }
}

template <int DIM, int F>
static hipError_t run_bin(int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                          const void *grad_out, float *acc, const BinWorkspace &w, int64_t n, hipStream_t s,
                          bool zero_table) {
    const int L = lt.num_lods;
    const int64_t NP = level_pitch(n);
    BinPlan whole;
    const int acc_kib = choose_acc_kib(DIM, lt, n);
    const bool oic = one_image_compact_rule(n);
    make_plan(DIM, lt, n, whole, acc_kib, oic);
    const int64_t nb = bin_batch_samples(DIM, dtype, lt, n);
    const bool multi = nb < n;
    const bool stage_all = (lt.stage_flags & SHACIRA_BWD_STAGE_ALL_LEVELS) != 0;
    const bool staged = (lt.stage_flags & SHACIRA_BWD_REUSE_STAGED) != 0;
    // only binned levels consume the transposed gradients (a later call on this workspace may, too: stage_all)
    const bool need_T = whole.nbl > 0 || stage_all || staged;
    // fixed-point images pay off once the accumulation itself dominates; small batches are bound by fixed costs and
    // keep the fp64 image (and skip the gmax bookkeeping): measured 100 vs 107 us at 65 536 samples
    const bool use_fx = need_T && n >= (1 << 17);
    // the 16-byte front kernel needs rows of whole 16-byte vectors and a 16-byte aligned input
    const size_t esz = dtype == SHACIRA_F32 ? 4 : 2;
    const int kvec = (int)(16 / (esz * F));
    size_t front_shmem = 0;
    const int ts16 = front_tile(L, F, whole.nbl, n, &front_shmem);
    const bool t16 = ts16 > 0 && (L % kvec) == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0;
    // counting fused into the front kernel: a call that transposes, single sub-batch
    const bool front_counts = need_T && !staged && !multi && t16 && whole.nbl > 0;
    // side stream (option bwd_fork): table zeroing + direct levels beside the critical path. Worth an event pair once the
    // batch is large (threshold in units of n * L * F: heavier tables fork earlier)
    const int64_t fork_work = n * lt.num_lods * lt.feature_dim;
    const int64_t fork_min = DIM == 3 ? ((int64_t)7 << 21) : ((int64_t)1 << 23);
    // (an event pair costs ~10-20 us of cross-stream latency here: only worth it with direct levels to hide)
    const bool fork = whole.nbl > 0 && whole.ngroups > 0 && !multi && opt().bwd_fork != 0 && fork_work >= fork_min;
    // selective zeroing (see zero_unowned_rows_kernel): a single sub-batch whose plan has hashed binned levels
    bool any_hashed = false;
    for (uint32_t q = 0; q < whole.nbl; ++q) any_hashed = any_hashed || lt.dense[whole.blevel[q]] == 0;
    const bool selective = zero_table && !multi && any_hashed && opt().bwd_selective_zero != 0;
    SideStream *ss = nullptr;
    hipStream_t zs = s;   // stream of the table zeroing and the direct levels
    if (fork) {
        SHACIRA_CHECK(side_stream(&ss));
        SHACIRA_CHECK(hipEventRecord(ss->fork, s));
        SHACIRA_CHECK(hipStreamWaitEvent(ss->stream, ss->fork, 0));
        zs = ss->stream;
    }
    if (zero_table) {   // at::zeros_like of the reference
        if (!selective) {
            SHACIRA_CHECK(zero_fill_async(acc, (int64_t)lt.table_rows * lt.feature_dim, zs));
        } else {
            hipLaunchKernelGGL(zero_unowned_rows_kernel, dim3(256, (uint32_t)L), dim3(256), 0, zs, acc, first_idx, lt, whole);
            SHACIRA_CHECK_LAUNCH();
        }
    }
    // bucket totals (and, when this call transposes, the per-level max |grad_output| right behind them) start at zero
    if (whole.nbl > 0 || (!staged && use_fx)) {
        const uint32_t words = (uint32_t)kTotalShards * kMaxBuckets + ((!staged && use_fx) ? SHACIRA_MAX_LODS : 0);
        hipLaunchKernelGGL(zero_words_kernel, dim3(32), dim3(256), 0, s, w.totals, words);
        SHACIRA_CHECK_LAUNCH();
    }
    if (need_T && !staged) {
        const int t_lb = stage_all ? 0 : lt.level_begin, t_le = stage_all ? L : lt.level_end;
        if (t16) {
            // ~512 workgroups (two per CU, all resident) when the batch allows; never fewer than one tile per workgroup
            const int64_t tiles = (n + ts16 - 1) / ts16;
            int ts_log2 = 7;
            while ((1 << ts_log2) < ts16) ++ts_log2;
            const int rounds = (int)((tiles + 511) / 512);
            const uint32_t blocks = (uint32_t)((tiles + rounds - 1) / rounds);
#define SHACIRA_FRONT(TT, GM, CN)                                                                                         \
            hipLaunchKernelGGL((front16_kernel<DIM, TT, F, GM, CN>), dim3(blocks), dim3(kFrontThreads), front_shmem, s, lt, \
                               whole, static_cast<const TT *>(grad_out), w.gT, coords, w.totals, w.cnt, n, NP, t_lb, t_le, ts_log2, \
                               rounds, GM ? w.gmax : nullptr)
            if (dtype == SHACIRA_F32) {
                if (use_fx && front_counts) SHACIRA_FRONT(float, true, true);
                else if (use_fx) SHACIRA_FRONT(float, true, false);
                else if (front_counts) SHACIRA_FRONT(float, false, true);
                else SHACIRA_FRONT(float, false, false);
            } else {
                if (use_fx && front_counts) SHACIRA_FRONT(__half, true, true);
                else if (use_fx) SHACIRA_FRONT(__half, true, false);
                else if (front_counts) SHACIRA_FRONT(__half, false, true);
                else SHACIRA_FRONT(__half, false, false);
            }
#undef SHACIRA_FRONT
        } else {
            const uint32_t blocks = (uint32_t)((n + 255) / 256);
            const size_t shmem = (size_t)256 * (L + 1) * F * sizeof(float);
            if (dtype == SHACIRA_F32 && use_fx)
                hipLaunchKernelGGL((transpose_grad_kernel<float, F, true>), dim3(blocks), dim3(256), shmem, s,
                                   static_cast<const float *>(grad_out), w.gT, n, NP, L, t_lb, t_le, w.gmax);
            else if (dtype == SHACIRA_F32)
                hipLaunchKernelGGL((transpose_grad_kernel<float, F, false>), dim3(blocks), dim3(256), shmem, s,
                                   static_cast<const float *>(grad_out), w.gT, n, NP, L, t_lb, t_le, nullptr);
            else if (use_fx)
                hipLaunchKernelGGL((transpose_grad_kernel<__half, F, true>), dim3(blocks), dim3(256), shmem, s,
                                   static_cast<const __half *>(grad_out), w.gT, n, NP, L, t_lb, t_le, w.gmax);
            else
                hipLaunchKernelGGL((transpose_grad_kernel<__half, F, false>), dim3(blocks), dim3(256), shmem, s,
                                   static_cast<const __half *>(grad_out), w.gT, n, NP, L, t_lb, t_le, nullptr);
        }
        SHACIRA_CHECK_LAUNCH();
    }
    if (fork) {   // the direct levels need the staged gradients (and gmax)
        SHACIRA_CHECK(hipEventRecord(ss->staged, s));
        SHACIRA_CHECK(hipStreamWaitEvent(ss->stream, ss->staged, 0));
    }
    // When nothing is transposed (every level is direct: the image configs) gmax would cost an extra read of grad_output
    // (tried: a streaming abs-max kernel); measured on config B it costs more than the faster atomics return (0.103 vs
    // 0.082 ms for the whole backward), so those calls keep the fp64 image.
    // direct levels: one pass over the whole batch, no items (they add into the zeroed table)
    if (whole.ngroups > 0) {
        const BinPlan &plan = whole;
        // one level group (S1's level 0): ~512 workgroups measured best; several groups (the all-direct image tables,
        // 128 KiB images = one resident workgroup per CU): 256 in total = one wave of workgroups, no tail
        // (config B backward 65 vs 77 us, tools/direct_blocks.py)
        uint32_t bpg = (plan.ngroups > 1 ? 256u : 512u) / plan.ngroups;
        const uint32_t need = (uint32_t)((n + 2047) / 2048);      // at least ~2 samples per thread each
        if (bpg > need) bpg = need;
        if (bpg < 1) bpg = 1;
        const size_t acc_bytes = (size_t)plan.BR * F * sizeof(double);
        const dim3 grid(bpg, plan.ngroups);
        // a row receives at most (samples walked by one workgroup) x (corners) contributions
        const int headroom = (use_fx && need_T) ? fx_headroom(((uint64_t)n / bpg + kConsumeThreads) * (1u << DIM)) : -1;
        if (need_T && use_fx)
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, float, true, true>), grid, dim3(kConsumeThreads),
                               acc_bytes, zs, lt, plan, first_idx, coords, w.gT, acc, n, NP, w.gmax, headroom);
        else if (need_T)
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, float, true, false>), grid, dim3(kConsumeThreads),
                               acc_bytes, zs, lt, plan, first_idx, coords, w.gT, acc, n, NP, nullptr, headroom);
        else if (dtype == SHACIRA_F32)
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, float, false, false>), grid, dim3(kConsumeThreads),
                               acc_bytes, zs, lt, plan, first_idx, coords, static_cast<const float *>(grad_out), acc, n,
                               NP, nullptr, headroom);
        else
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, __half, false, false>), grid, dim3(kConsumeThreads),
                               acc_bytes, zs, lt, plan, first_idx, coords, static_cast<const __half *>(grad_out), acc, n,
                               NP, nullptr, headroom);
        SHACIRA_CHECK_LAUNCH();
    }
    if (fork) SHACIRA_CHECK(hipEventRecord(ss->join, ss->stream));
    if (whole.nbl == 0) return hipSuccess;   // (fork implies binned levels)
    constexpr int NPAIR = 1 << (DIM - 1);
    // half-precision item stream: fp16 tables with F = 2 (8-byte units)
    const bool half = half_items(dtype, lt);
    const size_t stage = (size_t)TileOf<DIM>::value * NPAIR * ((half ? sizeof(typename ItemSel<F, true>::type) : sizeof(Item<F>)) + 1);
    bool first_batch = true;
    for (int64_t s0 = 0; s0 < n; s0 += nb) {
        const int64_t hi = (s0 + nb < n) ? (s0 + nb) : n;
        BinPlan plan;
        make_plan(DIM, lt, hi - s0, plan, acc_kib, oic);
        if (!first_batch) {
            hipLaunchKernelGGL(zero_words_kernel, dim3(32), dim3(256), 0, s, w.totals, (uint32_t)kTotalShards * kMaxBuckets);
            SHACIRA_CHECK_LAUNCH();
        }
        const bool fused_now = front_counts && first_batch;
        if (!fused_now) {
            hipLaunchKernelGGL((bin_count_levels_kernel<DIM>), count_grid(plan), dim3(kBinThreads), 0, s, lt, plan, coords,
                               w.totals, w.cnt, s0, hi);
            SHACIRA_CHECK_LAUNCH();
        }
        first_batch = false;
        hipLaunchKernelGGL(bin_scan_buckets_kernel, dim3(1), dim3(1024), 0, s, w.totals, w.base, w.unit_first, w.unit_desc,
                           plan.total_buckets, plan, w.work_counter, w.cursor);
        SHACIRA_CHECK_LAUNCH();
        if (selective) {   // hashed buckets with 0 or several units are zeroed now (the others are overwritten)
            hipLaunchKernelGGL(zero_odd_buckets_kernel, dim3(kMaxLevelBuckets, plan.nbl), dim3(256), 0, s, acc, first_idx,
                               w.unit_first, lt, plan);
            SHACIRA_CHECK_LAUNCH();
        }
        const uint32_t cps = fused_now ? (uint32_t)(TileOf<DIM>::value / ts16) : 1u;
        const uint32_t cnt_rows = fused_now ? (uint32_t)((n + ts16 - 1) / ts16) : plan.num_tiles;
        if (half)
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, true>), dim3(plan.num_tiles, plan.nbl), dim3(kBinThreads),
                               stage, s, lt, plan, coords, w.gT, w.cursor, w.cnt, cps, cnt_rows,
                               reinterpret_cast<typename ItemSel<F, true>::type *>(w.items), s0, hi, NP);
        else
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, false>), dim3(plan.num_tiles, plan.nbl), dim3(kBinThreads),
                               stage, s, lt, plan, coords, w.gT, w.cursor, w.cnt, cps, cnt_rows,
                               reinterpret_cast<Item<F> *>(w.items), s0, hi, NP);
        SHACIRA_CHECK_LAUNCH();
        if (fork) SHACIRA_CHECK(hipStreamWaitEvent(s, ss->join, 0));   // table zeroed, direct levels in
        const uint64_t max_items = (uint64_t)(hi - s0) * plan.nbl * NPAIR;
        uint32_t grid_units = (uint32_t)(max_items / plan.chunk_min) + plan.total_buckets + 1;
        const size_t acc_bytes = (size_t)plan.BR * F * sizeof(double);
        // persistent: as many workgroups as the chip holds fetch units from the work counter (measured: S1 backward
        // 0.611 -> 0.595 ms, 2-D 0.375 -> 0.369; at 65 536 samples the hardware's own dispatch of 1 400 tiny workgroups
        // is 5 us faster, so small batches keep it)
        uint32_t *wc = (opt().bwd_persistent != 0 && n >= (1 << 17)) ? w.work_counter : nullptr;
        if (wc != nullptr && grid_units > 512u) grid_units = 512u;
        const int headroom = use_fx ? fx_headroom((uint64_t)plan.chunk + 1) : -1;   // a unit streams <= chunk items
        const uint32_t *gm = use_fx ? w.gmax : nullptr;
        const int fa = multi ? 1 : 0;
        if (half && use_fx)
            hipLaunchKernelGGL((bin_consume_kernel<F, true, true>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s,
                               lt, plan, first_idx, w.base, w.unit_first, w.unit_desc,
                               reinterpret_cast<const typename ItemSel<F, true>::type *>(w.items), acc, fa, gm, headroom, wc);
        else if (half)
            hipLaunchKernelGGL((bin_consume_kernel<F, false, true>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s,
                               lt, plan, first_idx, w.base, w.unit_first, w.unit_desc,
                               reinterpret_cast<const typename ItemSel<F, true>::type *>(w.items), acc, fa, gm, headroom, wc);
        else if (use_fx)
            hipLaunchKernelGGL((bin_consume_kernel<F, true, false>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s, lt,
                               plan, first_idx, w.base, w.unit_first, w.unit_desc,
                               reinterpret_cast<const Item<F> *>(w.items), acc, fa, gm, headroom, wc);
        else if (!half)
            hipLaunchKernelGGL((bin_consume_kernel<F, false, false>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s, lt,
                               plan, first_idx, w.base, w.unit_first, w.unit_desc,
                               reinterpret_cast<const Item<F> *>(w.items), acc, fa, gm, headroom, wc);
        SHACIRA_CHECK_LAUNCH();
    }
    return hipSuccess;
}

hipError_t bin_backward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                        const void *grad_out, float *acc, void *workspace, int64_t n, hipStream_t s, bool zero_table) {
    const BinWorkspace w = carve(dim, dtype, lt, n, workspace);
    static PerDeviceOnce once;  // kernels that use more than 64 KiB of dynamic LDS must opt in once per device
    const hipError_t attr_err = once.run([]() -> hipError_t {
        hipError_t attr_err = hipSuccess;
        // dynamic LDS actually requested (static LDS of the kernels comes on top and must fit in 160 KiB too)
        auto set = [&attr_err](const void *fn, size_t bytes) {
            if (bytes <= 64 * 1024) return;
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) attr_err = e;
        };
#define SHACIRA_T_ATTR(TT, FF)                                                                           \
        set(reinterpret_cast<const void *>(&transpose_grad_kernel<TT, FF, true>), 140 * 1024);          \
        set(reinterpret_cast<const void *>(&transpose_grad_kernel<TT, FF, false>), 140 * 1024);         \
        set(reinterpret_cast<const void *>(&front16_kernel<2, TT, FF, true, true>), 156 * 1024);        \
        set(reinterpret_cast<const void *>(&front16_kernel<2, TT, FF, true, false>), 156 * 1024);       \
        set(reinterpret_cast<const void *>(&front16_kernel<2, TT, FF, false, true>), 156 * 1024);       \
        set(reinterpret_cast<const void *>(&front16_kernel<2, TT, FF, false, false>), 156 * 1024);      \
        set(reinterpret_cast<const void *>(&front16_kernel<3, TT, FF, true, true>), 156 * 1024);        \
        set(reinterpret_cast<const void *>(&front16_kernel<3, TT, FF, true, false>), 156 * 1024);       \
        set(reinterpret_cast<const void *>(&front16_kernel<3, TT, FF, false, true>), 156 * 1024);       \
        set(reinterpret_cast<const void *>(&front16_kernel<3, TT, FF, false, false>), 156 * 1024);
        SHACIRA_T_ATTR(float, 2) SHACIRA_T_ATTR(float, 4) SHACIRA_T_ATTR(__half, 2) SHACIRA_T_ATTR(__half, 4)
#undef SHACIRA_T_ATTR
#define SHACIRA_DIRECT_ATTR(D, FF)                                                                              \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, true, true>), 16384 * sizeof(double));   \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, true, false>), 16384 * sizeof(double));  \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, false, false>), 16384 * sizeof(double)); \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, __half, false, false>), 16384 * sizeof(double));
        SHACIRA_DIRECT_ATTR(2, 2) SHACIRA_DIRECT_ATTR(2, 4) SHACIRA_DIRECT_ATTR(3, 2) SHACIRA_DIRECT_ATTR(3, 4)
#undef SHACIRA_DIRECT_ATTR
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, true, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, true, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, false, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, false, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, true, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, false, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, true, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, false, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 4, true>), (size_t)TileOf<2>::value * 2 * (sizeof(ItemH4) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 4, true>), (size_t)kTile * 4 * (sizeof(ItemH4) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 2, false>), (size_t)TileOf<2>::value * 2 * (sizeof(Item<2>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 4, false>), (size_t)TileOf<2>::value * 2 * (sizeof(Item<4>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 2, false>), (size_t)kTile * 4 * (sizeof(Item<2>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 4, false>), (size_t)kTile * 4 * (sizeof(Item<4>) + 1));
        return attr_err;
    });
    if (attr_err != hipSuccess) return attr_err;
    if (dim == 3) {
        return lt.feature_dim == 2 ? run_bin<3, 2>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table)
                                   : run_bin<3, 4>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table);
    }
    return lt.feature_dim == 2 ? run_bin<2, 2>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table)
                               : run_bin<2, 4>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table);
}

}  // namespace shacira
