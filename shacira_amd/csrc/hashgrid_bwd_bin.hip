// hashgrid_bwd_bin.hip -- backward of the hash-grid lookup WITHOUT scattered global atomics (gfx950).
//
// Why: on MI355X a scattered global float atomic costs one memory-side request (~21 G requests/s chip-wide,
// profiles/r01_microbench_rates.txt) and LDS ds_add_f32 runs at 0.33 op/clk/CU, while LDS ds_add_f64 sustains
// ~1.3 T adds/s (profiles/r01_microbench2_lds_gather.txt). The reference's design (one atomicAdd per corner per
// feature, hashgrid_interpolate_cuda.cu:212-221) therefore runs ~30x under the HBM roof here. This file replaces
// it by "partition, then accumulate on chip":
//
//   pass T  transpose   grad_output [N, L*F] -> gT [L][N][F]            (coalesced both ways through LDS)
//   pass A  count       per (level, tile of samples): how many items fall into each bucket  -> cnt[tile][bucket]
//   pass S  scan        exclusive scans: per bucket over tiles, then over buckets; builds the consumer work list
//   pass B  bin         recompute the corners, stage the tile's items in LDS sorted by bucket, write each bucket's
//                       run to its exact slot in HBM with coalesced 16-byte stores
//   pass C  consume     one workgroup per (bucket, chunk): accumulate the items into an LDS-resident fp64 image of
//                       the bucket's rows (ds_add_f64), then write the rows out (plain coalesced stores when the
//                       bucket has a single chunk, coalesced float atomics otherwise)
//
// A *bucket* is a range of <= BR consecutive rows of one level (BR*F*8 B = 128 KiB of LDS). An *item* is one
// x-pair of corners (x, x+1) at fixed (y[,z]) offsets: both rows always share a bucket (hashed levels: the rows
// differ only in the low bits x ^ (x+1); dense levels: buckets hold whole x-lines), so an item is 8 + 4F bytes:
//   { key = rowA | rowB << 13 | validA << 26 | validB << 27,  fx,  a_j = grad_j * w_rest }   (rows bucket-local)
// and the consumer adds a_j*(1-fx) to rowA and a_j*fx to rowB. The sum is kept in fp64 and rounded once.
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
constexpr int kTile = SHACIRA_KTILE;       // samples per (level, tile) block in passes A and B
constexpr int kBinThreads = SHACIRA_KBIN;  // threads of passes A and B
constexpr int kConsumeThreads = 1024;
constexpr int kMaxBuckets = 2048;     // over all levels
constexpr int kMaxLevelBuckets = 128; // per level (LDS histogram size)

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

struct BinPlan {
    BinLevel lv[SHACIRA_MAX_LODS];
    uint32_t total_buckets;
    uint32_t BR;
    uint32_t num_tiles;
    uint32_t pairs;     // items per (sample, level) = 2^(dim-1)
    uint32_t chunk;     // items per consumer work unit
    uint32_t chunk_c;   // same for compact levels (smaller: they are consumed last and even out the tail)
    uint32_t rotf;      // 1: consumers rotate the feature order by lane (LDS bank spreading)
    uint32_t rot_bucket; // first bucket of the first HASHED binned level: the persistent consume pass starts there
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
// grad_output [N, L*F] (T) -> gT [L][N][F] fp32, through LDS, F scalars per lane per access. Block: 256 samples.
template <typename T, int F, bool GMAX>
__global__ __launch_bounds__(256) void transpose_grad_kernel(const T *__restrict__ go, float *__restrict__ gT,
                                                             int64_t N, int L, int lb, int le,
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
            float *dst = reinterpret_cast<float *>(out + (int64_t)l * N + s0 + threadIdx.x);
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

// 16-byte form of pass T (round 3): the 8-byte accesses above ran at 2.5 TB/s, half the chip's copy rate (the guide prices
// 8-byte accesses at 0.54-0.70x the 16-byte rate). Rows whose byte size is a multiple of 16 are read as 16-byte vectors
// (K = 16 / (sizeof(T) * F) level pieces of one sample per lane, every load of the tile issued before the first LDS write),
// kept LEVEL-major in LDS, and leave as 16-byte non-temporal vectors of M = 16 / (4 F) consecutive samples of one level.
// Tile = TS samples (a multiple of 128); pitch = TS + 2 pieces keeps the 16-byte LDS reads aligned (2-way conflicts on the
// writes only). Requires (l * N + tile start) * F * 4 to be 16-byte aligned for every level: N even when F == 2.
template <typename T, int F, bool GMAX>
__global__ __launch_bounds__(256) void transpose_grad16_kernel(const T *__restrict__ go, float *__restrict__ gT,
                                                               int64_t N, int L, int lb, int le, int TS,
                                                               uint32_t *__restrict__ gmax) {
    constexpr int K = 16 / (int)(sizeof(T) * F);   // level pieces per 16-byte input vector
    constexpr int M = 16 / (4 * F);                // samples per 16-byte output vector
    static_assert(K >= 1 && M >= 1, "16-byte transpose: F <= 4");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    struct alignas(sizeof(float) * F) PieceOut { float v[F]; };
    __shared__ uint32_t s_max[SHACIRA_MAX_LODS];
    extern __shared__ __align__(16) unsigned char s_raw_g[];
    PieceOut *s_tile = reinterpret_cast<PieceOut *>(s_raw_g);   // [L][TS + 2]
    if (GMAX && threadIdx.x < SHACIRA_MAX_LODS) s_max[threadIdx.x] = 0;
    const int pitch = TS + 2;
    const int VPR = L / K;                                      // input vectors per row
    const int64_t s0 = (int64_t)blockIdx.x * TS;
    const int ns = (int)((N - s0 < TS) ? (N - s0) : TS);
    const u32x4 *in = reinterpret_cast<const u32x4 *>(go) + s0 * VPR;
    const int total = ns * VPR;
    constexpr int UL = 8;                                       // loads in flight per thread
    for (int e0 = threadIdx.x; e0 < total; e0 += 256 * UL) {
        u32x4 raw[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int e = e0 + u * 256;
            if (e < total) raw[u] = __builtin_nontemporal_load(in + e);
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int e = e0 + u * 256;
            if (e >= total) continue;
            const int sm = e / VPR, v = e - sm * VPR;
            T tv[K * F];
            __builtin_memcpy(tv, &raw[u], 16);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                PieceOut q;
#pragma unroll
                for (int j = 0; j < F; ++j) q.v[j] = Scalar<T>::load(&tv[k * F + j]);
                s_tile[(v * K + k) * pitch + sm] = q;
            }
        }
    }
    __syncthreads();
    const int nvec = TS / M;
    for (int l = lb; l < le; ++l) {
        uint32_t m = 0;
        for (int q = threadIdx.x; q < nvec; q += 256) {
            const int sm = q * M;
            if (sm >= ns) break;
            const f32x4 val = *reinterpret_cast<const f32x4 *>(&s_tile[l * pitch + sm]);
            float *dst = gT + ((int64_t)l * N + s0 + sm) * F;
            if (sm + M <= ns) {
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
    if constexpr (GMAX) {
        __syncthreads();
        if ((int)threadIdx.x >= lb && (int)threadIdx.x < le && s_max[threadIdx.x])
            atomicMax(&gmax[threadIdx.x], s_max[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------------------- passes T + A fused
// One workgroup = one 1024-sample tile of passes A / B: four 256-sample rounds of the transpose (memory bound) with
// the bucket counting of the same samples (ALU bound: corner hashing of every binned level) in between, so the two
// overlap inside every CU -- issued as two kernels on two streams they mostly ran one after the other.
constexpr int kFuseSlots = 4;                    // level slots per sample in the fused transpose + count kernel
constexpr int kFuseThreads = 256 * kFuseSlots;
template <int DIM, typename T, int F, bool GMAX>
__global__ __launch_bounds__(kFuseThreads) void transpose_count_kernel(LevelTable lt, BinPlan plan, const T *__restrict__ go,
                                                              float *__restrict__ gT, const float *__restrict__ coords,
                                                              uint32_t *__restrict__ cnt, int64_t N, int lb, int le,
                                                              uint32_t *__restrict__ gmax) {
    __shared__ uint32_t s_max[SHACIRA_MAX_LODS];
    __shared__ uint32_t s_hist[SHACIRA_MAX_LODS][kMaxLevelBuckets];
    struct alignas(sizeof(T) * F) PieceIn { T v[F]; };
    struct alignas(sizeof(float) * F) PieceOut { float v[F]; };
    extern __shared__ __align__(16) unsigned char s_raw_g[];
    PieceOut *s_tile = reinterpret_cast<PieceOut *>(s_raw_g);  // [256][L + 1]
    const int L = lt.num_lods, pitch = L + 1;
    const uint32_t tile = blockIdx.x;
    // thread = (sample sm of the round, level slot q): the transpose stores and the counting loop take every
    // kFuseSlots-th level, so a round keeps 16 waves busy
    const int sm_t = threadIdx.x & 255, q_t = threadIdx.x >> 8;
    if (threadIdx.x < SHACIRA_MAX_LODS) s_max[threadIdx.x] = 0;
    for (uint32_t k = threadIdx.x; k < plan.nbl * (uint32_t)kMaxLevelBuckets; k += kFuseThreads) (&s_hist[0][0])[k] = 0;
    __syncthreads();
    PieceOut *out = reinterpret_cast<PieceOut *>(gT);
    for (int sub = 0; sub < kTile / 256; ++sub) {
        const int64_t s0 = (int64_t)tile * kTile + sub * 256;
        if (s0 >= N) break;
        const int ns = (int)((N - s0 < 256) ? (N - s0) : 256);
        const PieceIn *in = reinterpret_cast<const PieceIn *>(go) + s0 * L;
        const int total = ns * L;
        for (int e = threadIdx.x; e < total; e += kFuseThreads) {
            const int sm = e / L, l = e - sm * L;
            const PieceIn p = in[e];
            PieceOut q;
#pragma unroll
            for (int j = 0; j < F; ++j) q.v[j] = Scalar<T>::load(&p.v[j]);
            s_tile[sm * pitch + l] = q;
        }
        // bucket counts of this round's sample while the rows above are in flight
        if (sm_t < ns) {
            const int64_t i = s0 + sm_t;
            double t[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
            for (uint32_t li = (uint32_t)q_t; li < plan.nbl; li += kFuseSlots) {
                const uint32_t lvl = plan.blevel[li];
                const BinLevel bl = plan.lv[lvl];
                bool done = false;
                if constexpr (DIM == 3) {
                    if (bl.compact) {
                        int32_t pz;
                        float fz, gz;
                        axis_transform(t[2], lt.res[lvl], lt.hi[lvl], pz, fz, gz);
                        atomicAdd(&s_hist[li][(uint32_t)pz / bl.slab], 2u);
                        done = true;
                    }
                }
                if (!done) {
                    float fx;
                    PairSlot ps[1 << (DIM - 1)];
                    enumerate_pairs<DIM>(t, lt.res[lvl], lt.hi[lvl], lt.dense[lvl] != 0, lt.mask, bl, plan.BR, fx, ps);
#pragma unroll
                    for (int q = 0; q < (1 << (DIM - 1)); ++q)
                        if (ps[q].key >> 26) atomicAdd(&s_hist[li][ps[q].bucket], 1u);
                }
            }
        }
        __syncthreads();
        for (int l = lb + q_t; l < le; l += kFuseSlots) {
            uint32_t m = 0;
            if (sm_t < ns) {
                const PieceOut q = s_tile[sm_t * pitch + l];
                if constexpr (GMAX) {
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        const uint32_t b = __float_as_uint(fabsf(q.v[j]));
                        m = b > m ? b : m;
                    }
                }
                float *dst = reinterpret_cast<float *>(out + (int64_t)l * N + s0 + sm_t);
                if constexpr (F == 2) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 v = {q.v[0], q.v[1]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x2 *>(dst));
                } else {
#pragma unroll
                    for (int j = 0; j < F; ++j) __builtin_nontemporal_store(q.v[j], dst + j);
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
        __syncthreads();   // s_tile is refilled by the next round
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < plan.nbl * (uint32_t)kMaxLevelBuckets; k += kFuseThreads) {
        const uint32_t li = k / kMaxLevelBuckets, b = k % kMaxLevelBuckets;
        const BinLevel &bl = plan.lv[plan.blevel[li]];
        if (b < bl.nb) cnt[(size_t)tile * plan.total_buckets + bl.bucket0 + b] = s_hist[li][b];
    }
    if constexpr (GMAX) {
        if ((int)threadIdx.x >= lb && (int)threadIdx.x < le && s_max[threadIdx.x])
            atomicMax(&gmax[threadIdx.x], s_max[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------------------- pass S
// cnt is TILE-major, cnt[tile][bucket] (a tile's counts of one level are contiguous: the count pass writes and the scatter
// pass reads them as one 256-byte piece instead of one dword per 128-byte line). One workgroup per 16 buckets: thread =
// (bucket, chunk of tiles), 64 chunks: per-chunk sums, prefix over the chunks, exclusive offsets in place; total ->
// totals[bucket]. Every access is a 64-byte row piece shared by 16 lanes.
constexpr int kScanWaves = 16, kScanBuckets = 16, kScanChunks = 64 * kScanWaves / kScanBuckets;
__global__ __launch_bounds__(64 * kScanWaves) void bin_scan_tiles_kernel(uint32_t *__restrict__ cnt,
                                                                          uint32_t *__restrict__ totals, uint32_t num_tiles,
                                                                          uint32_t nbuckets) {
    __shared__ uint32_t s_sum[kScanChunks][kScanBuckets];
    const uint32_t bl = threadIdx.x % kScanBuckets, chunk = threadIdx.x / kScanBuckets;
    const uint32_t gb = blockIdx.x * kScanBuckets + bl;
    const bool ok = gb < nbuckets;
    const uint32_t tpc = (num_tiles + kScanChunks - 1) / kScanChunks;
    const uint32_t t0 = chunk * tpc < num_tiles ? chunk * tpc : num_tiles;
    const uint32_t t1 = (t0 + tpc < num_tiles) ? t0 + tpc : num_tiles;
    uint32_t sum = 0;
    if (ok) {
#pragma unroll 8
        for (uint32_t t = t0; t < t1; ++t) sum += cnt[(size_t)t * nbuckets + gb];
    }
    s_sum[chunk][bl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t c = 0; c < (uint32_t)kScanChunks; ++c) {
        const uint32_t v = s_sum[c][bl];
        if (c < chunk) run += v;
        total += v;
    }
    if (ok) {
        for (uint32_t t = t0; t < t1; ++t) {
            uint32_t *p = cnt + (size_t)t * nbuckets + gb;
            const uint32_t v = *p;
            *p = run;
            run += v;
        }
        if (chunk == 0) totals[gb] = total;
    }
}

// single block: bucket bases (exclusive scan of totals) and the consumer work list
//   base[b]         first item of bucket b in the item array (base[nb] = total)
//   unit_first[b]   first work unit of bucket b; unit_first[nb] = number of units
//   unit_desc[u]    item range, bucket and level of work unit u
__global__ __launch_bounds__(1024) void bin_scan_buckets_kernel(const uint32_t *__restrict__ totals,
                                                                uint64_t *__restrict__ base,
                                                                uint32_t *__restrict__ unit_first,
                                                                UnitDesc *__restrict__ unit_desc, uint32_t nb,
                                                                uint32_t chunk_items, BinPlan plan,
                                                                uint32_t *__restrict__ work_counter) {
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
        c[k] = (b < nb) ? totals[b] : 0u;
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
// ROWS = false: gradients from the transposed image gT [L][N][F], grid (tiles, levels).
// ROWS = true:  gradients straight from grad_output [N, L*F] (8 or 16 bytes of every 128-byte row per level) -- no
//   transposing pass. Only worth it when the rows are fetched from HBM once: the 1-D grid is numbered so that ALL levels
//   of a tile run back to back on ONE XCD (workgroup b runs on XCD b % 8): the first level pulls the tile's rows (and
//   coordinates) into that XCD's L2, the others hit there. Each workgroup also folds max |g| of its level into gmax[]
//   (the consumer's fixed-point scale, otherwise a by-product of the transpose).
template <int DIM, int F, bool ROWS>
__global__ __launch_bounds__(kBinThreads) void bin_scatter_kernel(LevelTable lt, BinPlan plan,
                                                                  const float *__restrict__ coords,
                                                                  const float *__restrict__ gT,
                                                                  const uint32_t *__restrict__ tile_off,
                                                                  const uint64_t *__restrict__ base,
                                                                  Item<F> *__restrict__ items, int64_t sample0,
                                                                  int64_t N, int64_t Ntotal, uint32_t lvl_off,
                                                                  uint32_t nlev, uint32_t *__restrict__ gmax) {
    constexpr int NP = 1 << (DIM - 1);
    constexpr int SPT = kTile / kBinThreads;   // samples per thread
    constexpr int kStage = kTile * NP;         // staged items per block
    extern __shared__ __align__(16) unsigned char s_raw[];
    Item<F> *s_items = reinterpret_cast<Item<F> *>(s_raw);
    uint8_t *s_bucket = reinterpret_cast<uint8_t *>(s_items + kStage);
    __shared__ uint32_t s_hist[kMaxLevelBuckets];
    __shared__ uint32_t s_start[kMaxLevelBuckets + 1];
    __shared__ uint64_t s_gbase[kMaxLevelBuckets];

    uint32_t tile = blockIdx.x, bi = blockIdx.y;
    if constexpr (ROWS) {   // b -> (XCD, slot); slot -> (tile of that XCD, level)
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        tile = (slot / nlev) * 8u + xcd;
        bi = slot % nlev;
        if (tile >= plan.num_tiles) return;
    }
    const uint32_t lvl = plan.blevel[bi + lvl_off];
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
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const int k = threadIdx.x + u * kBinThreads;
        const int64_t i = sample0 + (int64_t)tile * kTile + k;
        const bool live = i < N;
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(live ? coords[i * DIM + a] : 0.0f);
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
        if (live) {
            const float *gp = ROWS ? gT + (i * lt.num_lods + lvl) * F : gT + ((int64_t)lvl * Ntotal + i) * F;
            if constexpr (F == 2) {
                const float2 v = *reinterpret_cast<const float2 *>(gp);
                g[u][0] = v.x; g[u][1] = v.y;
            } else {
#pragma unroll
                for (int j = 0; j < F; ++j) g[u][j] = gp[j];
            }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (!live) ps[u][q].key = 0;
            rank[u][q] = (ps[u][q].key >> 26) ? atomicAdd(&s_hist[ps[u][q].bucket], compact ? 2u : 1u) : 0u;
        }
    }
    __syncthreads();
    if constexpr (ROWS) {
        if (gmax != nullptr) {   // max |g| of the level as an integer max on the float bit patterns (finite < inf < NaN)
            uint32_t m = 0;
#pragma unroll
            for (int u = 0; u < SPT; ++u) {
                if (sample0 + (int64_t)tile * kTile + threadIdx.x + u * kBinThreads < N) {
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        const uint32_t b = __float_as_uint(fabsf(g[u][j]));
                        m = b > m ? b : m;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(m, off, 64);
                m = o > m ? o : m;
            }
            if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(&gmax[lvl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(&gmax[lvl], m);
        }
    }
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
    if (threadIdx.x < bl.nb) {
        const size_t gb = bl.bucket0 + threadIdx.x;
        s_gbase[threadIdx.x] = base[gb] + tile_off[(size_t)tile * plan.total_buckets + gb];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (ps[u][q].key >> 26) {
                const uint32_t pos = s_start[ps[u][q].bucket] + rank[u][q];
                Item<F> it;
                it.key = ps[u][q].key;
                it.fx = fx[u];
                if (compact) {
                    if constexpr (F == 2) {   // two slots: {key, fx, fy, fz} {0, g0, g1, 0}
                        it.a[0] = fyz[u][0];
                        it.a[1] = fyz[u][1];
                        s_items[pos] = it;
                        Item<F> it2;
                        it2.key = 0;
                        it2.fx = g[u][0];
                        it2.a[0] = g[u][1];
                        it2.a[1] = 0.0f;
                        s_items[pos + 1] = it2;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        s_bucket[pos + 1] = (uint8_t)ps[u][q].bucket;
                    } else if constexpr (F == 4) {   // two 24-byte slots: {key, fx, fy, fz, g0, g1} {0, g2, g3, -, -, -}
                        it.a[0] = fyz[u][0];
                        it.a[1] = fyz[u][1];
                        it.a[2] = g[u][0];
                        it.a[3] = g[u][1];
                        s_items[pos] = it;
                        Item<F> it2;
                        it2.key = 0;
                        it2.fx = g[u][2];
                        it2.a[0] = g[u][3];
                        it2.a[1] = 0.0f; it2.a[2] = 0.0f; it2.a[3] = 0.0f;
                        s_items[pos + 1] = it2;
                        s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                        s_bucket[pos + 1] = (uint8_t)ps[u][q].bucket;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < F; ++j) it.a[j] = g[u][j] * ps[u][q].wrest;
                    s_items[pos] = it;
                    s_bucket[pos] = (uint8_t)ps[u][q].bucket;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t staged = s_start[bl.nb];
    for (uint32_t pos = threadIdx.x; pos < staged; pos += kBinThreads) {
        const uint32_t b = s_bucket[pos];
        // write-once / read-once stream: non-temporal stores (measured -7 % on the whole backward)
        store_item_nt<F>(items + s_gbase[b] + (pos - s_start[b]), s_items[pos]);
    }
}

// ------------------------------------------------------------------------------------------------- pass A
// One workgroup per TILE counting every binned level: the coordinates are loaded once (all loads of the block in flight
// together) and there are 10-15x fewer workgroups than one per (tile, level): 55 vs 88 us on S1. The same restructuring of
// pass B measured slower (283 vs 247 us): profiles/r02_bwd_ablation.md.
template <int DIM>
__global__ __launch_bounds__(kBinThreads) void bin_count_levels_kernel(LevelTable lt, BinPlan plan,
                                                                       const float *__restrict__ coords,
                                                                       uint32_t *__restrict__ cnt, int64_t sample0,
                                                                       int64_t N) {
    __shared__ uint32_t s_hist[SHACIRA_MAX_LODS][kMaxLevelBuckets];
    constexpr int SPT = kTile / kBinThreads;
    const uint32_t tile = blockIdx.x;
    // gridDim.y workgroups share a tile's levels (bi = blockIdx.y, blockIdx.y + gridDim.y, ...): small batches keep the chip
    // busy with one level each, large ones load the coordinates once for all levels
    for (uint32_t e = threadIdx.x; e < plan.nbl * kMaxLevelBuckets; e += kBinThreads) (&s_hist[0][0])[e] = 0;
    double t[SPT][DIM];
    bool live[SPT];
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const int64_t i = sample0 + (int64_t)tile * kTile + threadIdx.x + u * kBinThreads;
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
            float fx;
            PairSlot ps[1 << (DIM - 1)];
            enumerate_pairs<DIM>(t[u], res, hi, dense, lt.mask, bl, plan.BR, fx, ps);
#pragma unroll
            for (int q = 0; q < (1 << (DIM - 1)); ++q)
                if (ps[q].key >> 26) atomicAdd(&s_hist[bi][ps[q].bucket], 1u);
        }
    }
    __syncthreads();
    for (uint32_t bi = blockIdx.y; bi < plan.nbl; bi += gridDim.y) {
        const BinLevel bl = plan.lv[plan.blevel[bi]];
        for (uint32_t b = threadIdx.x; b < bl.nb; b += kBinThreads)
            cnt[(size_t)tile * plan.total_buckets + bl.bucket0 + b] = s_hist[bi][b];
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
__device__ __forceinline__ bool g_rot_enabled(const BinPlan &plan) { return plan.rotf != 0; }

// ------------------------------------------------------------------------------------------------- pass C
// one work unit (a bucket, or a chunk of an over-full one) on the calling workgroup
template <int F, bool FX>
__device__ __forceinline__ void consume_unit(const LevelTable &lt, const BinPlan &plan, const int32_t *__restrict__ first_idx,
                                             const UnitDesc d, const Item<F> *__restrict__ items,
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
    if constexpr (F == 2 || F == 4) {
        if (bl.compact) {
            // one sample per two slots: F = 2 {local base row | valid, fx, fy, fz} {-, g0, g1, -}; F = 4 {.., fx, fy, fz, g0, g1}
            // {-, g2, g3, ...}; all 8 corners land here
            constexpr int UC = 2;
            const uint32_t r2 = r1 * r1;
            const int rotc = plan.rotf ? (int)(threadIdx.x & (F - 1)) : 0;
            for (uint64_t p0 = begin + 2ull * threadIdx.x; p0 < end; p0 += 2ull * kConsumeThreads * UC) {
                Item<F> ia[UC], ib[UC];
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const uint64_t p = p0 + 2ull * u * kConsumeThreads;
                    if (p + 1 < end) {
                        ia[u] = load_item_nt<F>(items + p);
                        ib[u] = load_item_nt<F>(items + p + 1);
                    } else {
                        ia[u].key = 0;
                    }
                }
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    if (!(ia[u].key & (1u << 26))) continue;
                    const uint32_t base_row = ia[u].key & 0x1FFFu;
                    const float fxx = ia[u].fx, fyy = ia[u].a[0], fzz = ia[u].a[1];
                    const float gxx = 1.0f - fxx, gyy = 1.0f - fyy, gzz = 1.0f - fzz;
                    float gg[F];
                    if constexpr (F == 2) {
                        gg[0] = ib[u].fx; gg[1] = ib[u].a[0];
                    } else {
                        gg[0] = ia[u].a[2]; gg[1] = ia[u].a[3]; gg[2] = ib[u].fx; gg[3] = ib[u].a[0];
                    }
                    const float wxy[4] = {gxx * gyy, gxx * fyy, fxx * gyy, fxx * fyy};   // reference order: (x * y) * z
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const uint32_t row = base_row + ((c >> 2) & 1) + ((c >> 1) & 1) * r1 + (c & 1) * r2;
                        const float w = wxy[c >> 1] * ((c & 1) ? fzz : gzz);
                        if (row >= nrows) continue;   // cannot happen for in-range cells; keeps the image safe
#pragma unroll
                        for (int jj = 0; jj < F; ++jj) {
                            const int j = (jj + rotc) & (F - 1);
                            const float gj = pick<F>(gg, j);
                            if (FX && fx.fixed) atomicAdd(&s_fix[row * F + j], fx_encode(gj * w, fx.scale));
                            else atomicAdd(&s_acc[row * F + j], (double)(gj * w));
                        }
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
    constexpr int UN = 8;  // items in flight per thread (4: -1 %, 16: +3 % with the fixed-point atomics)
    const int rotf = g_rot_enabled(plan) ? (int)(threadIdx.x & (F - 1)) : 0;
    for (uint64_t p0 = begin + threadIdx.x; p0 < end; p0 += (uint64_t)kConsumeThreads * UN) {
        Item<F> it[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint64_t p = p0 + (uint64_t)u * kConsumeThreads;
            if (p < end) it[u] = load_item_nt<F>(items + p);
            else it[u].key = 0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t ra = it[u].key & 0x1FFFu, rb = (it[u].key >> 13) & 0x1FFFu;
            const float gx = 1.0f - it[u].fx;
            if (FX && fx.fixed) {
                // feature order rotated by lane: the F slots of a row are consecutive 8-byte words, so with every lane adding
                // feature j in the same instruction only 1 / F of the LDS banks were addressed (half of the pass's LDS
                // cycles were bank conflicts)
                if (it[u].key & (1u << 26)) {
#pragma unroll
                    for (int jj = 0; jj < F; ++jj) {
                        const int j = (jj + rotf) & (F - 1);
                        atomicAdd(&s_fix[ra * F + j], fx_encode(pick<F>(it[u].a, j) * gx, fx.scale));
                    }
                }
                if (it[u].key & (1u << 27)) {
#pragma unroll
                    for (int jj = 0; jj < F; ++jj) {
                        const int j = (jj + rotf) & (F - 1);
                        atomicAdd(&s_fix[rb * F + j], fx_encode(pick<F>(it[u].a, j) * it[u].fx, fx.scale));
                    }
                }
            } else {
                if (it[u].key & (1u << 26)) {
#pragma unroll
                    for (int j = 0; j < F; ++j) atomicAdd(&s_acc[ra * F + j], (double)(it[u].a[j] * gx));
                }
                if (it[u].key & (1u << 27)) {
#pragma unroll
                    for (int j = 0; j < F; ++j) atomicAdd(&s_acc[rb * F + j], (double)(it[u].a[j] * it[u].fx));
                }
            }
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
// consume_unit are LDS-only). `work_counter` NULL: one unit per workgroup (launches over a bucket range: option bwd_groups).
template <int F, bool FX>
__global__ __launch_bounds__(kConsumeThreads) void bin_consume_kernel(LevelTable lt, BinPlan plan,
                                                                      const int32_t *__restrict__ first_idx,
                                                                      const uint64_t *__restrict__ base,
                                                                      const uint32_t *__restrict__ unit_first,
                                                                      const UnitDesc *__restrict__ unit_desc,
                                                                      const Item<F> *__restrict__ items,
                                                                      float *__restrict__ grad_table,
                                                                      int force_atomic,
                                                                      const uint32_t *__restrict__ gmax,
                                                                      int headroom, uint32_t bucket_lo,
                                                                      uint32_t bucket_hi,
                                                                      uint32_t *__restrict__ work_counter) {
    extern __shared__ double s_acc[];  // [rows_pb][F]: fp64, or 64-bit fixed point (same size)
    __shared__ uint32_t s_unit;
    // this launch consumes the work units of buckets [bucket_lo, bucket_hi) (one group of levels, or all of them)
    const uint32_t unit0 = bucket_lo ? unit_first[bucket_lo] : 0u;
    const uint32_t unit_end = unit_first[bucket_hi];
    if (work_counter == nullptr) {
        const uint32_t unit = blockIdx.x + unit0;
        if (unit >= unit_end) return;
        consume_unit<F, FX>(lt, plan, first_idx, unit_desc[unit], items, grad_table, force_atomic, gmax, headroom, s_acc);
        return;
    }
    // fetch order: the hashed levels' units (one whole bucket each: equal, large) first, the dense levels' smaller chunks
    // last, so that the last round of units is made of small ones (persistent launches always cover the whole plan)
    const uint32_t rot = (plan.rot_bucket == 0xFFFFFFFFu ? 0u : unit_first[plan.rot_bucket]) - unit0, nunits = unit_end - unit0;
    for (;;) {
        if (threadIdx.x == 0) s_unit = atomicAdd(work_counter, 1u);
        lds_barrier();
        if (s_unit >= nunits) return;
        const uint32_t unit = plan.rot_bucket == 0xFFFFFFFFu
                                  ? unit_end - 1u - s_unit
                                  : unit0 + (s_unit + rot < nunits ? s_unit + rot : s_unit + rot - nunits);
        consume_unit<F, FX>(lt, plan, first_idx, unit_desc[unit], items, grad_table, force_atomic, gmax, headroom, s_acc);
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
                                                                            int64_t N,
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
    const int rotd = plan.rotf ? (int)(threadIdx.x & (F - 1)) : 0;
    for (int64_t i = (int64_t)blockIdx.x * kConsumeThreads + threadIdx.x; i < N; i += stride) {
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
        for (int l = 0; l < lt.num_lods; ++l) {
            if (!((mask >> l) & 1u)) continue;
            const BinLevel bl = plan.lv[l];
            Corners<DIM> c;
            compute_corners<DIM>(t, lt.res[l], lt.hi[l], lt.dense[l] != 0, lt.mask, c);
            const GT *gp = TRANSPOSED ? gT + ((int64_t)l * N + i) * F : gT + (i * lt.num_lods + l) * F;
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
                        for (int j = 0; j < F; ++j) atomicAdd(s_acc + slot + j, (double)(g[j] * c.w[k]));
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

static void make_plan(int dim, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib);

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
    make_plan(dim, lt, kTile, plan, acc_kib);
    if (plan.total_buckets > (uint32_t)kMaxBuckets) return false;
    for (int l = 0; l < lt.num_lods; ++l)
        if (plan.lv[l].nb > (uint32_t)kMaxLevelBuckets) return false;
    return true;
}

// Image size per call, from the TOTAL batch (one choice per call so that every plan of the call classifies the levels
// alike). Option "bin_acc_kib": 64 / 128 force it, 0 (default) = measured rule: 64 KiB images (two consumer workgroups
// per CU overlap their zero / stream / flush phases) win up to 2^19 3-D samples, 128 KiB (half as many buckets) beyond.
static int choose_acc_kib(int dim, const LevelTable &lt, int64_t n) {
    const int opt = g_bin_acc_kib.load();
    if (opt != 0) return opt;
    const int64_t pairs = (int64_t)1 << (dim - 1);
    if (n * pairs > ((int64_t)1 << 21) || !bin_feasible(dim, lt, 64)) return 128;
    // tables whose levels are all "direct" (config B: every level fits an LDS image) want the big image: fewer level
    // groups, hence fewer walks over the samples (measured 82 vs 124 us on the 393 216-pixel batch)
    BinPlan big;
    make_plan(dim, lt, kTile, big, 128);
    return big.nbl == 0 ? 128 : 64;
}

bool bin_supported(int dim, const LevelTable &lt) {
    const int opt = g_bin_acc_kib.load();
    return bin_feasible(dim, lt, opt ? opt : 128);
}

static void make_plan(int dim, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib) {
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
            const uint64_t planes = (dim == 3 && (F == 2 || F == 4) && g_bwd_compact.load() != 0) ? BR / (res * res) : 0;
            if (planes >= 2 && res >= 3 && bl.used > BR) {
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
        if (bl.nb == 1 && bl.used <= BR) {
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
    plan.num_tiles = (uint32_t)((n_batch + kTile - 1) / kTile);
    plan.pairs = 1u << (dim - 1);
    // work-unit size: 1/48 of ONE level's items, so that an evenly loaded hashed bucket (1/64 of a level) is ONE unit
    // (plain-store flush) with 33 % slack, while over-full coarse buckets split into equal chunks that keep all CUs busy
    uint64_t chunk = (uint64_t)n_batch * plan.pairs / 48 + 1024;
    if (chunk < 8192) chunk = 8192;
    if (chunk > (1u << 22)) chunk = 1u << 22;
    plan.chunk = (uint32_t)chunk & ~1u;   // even: a compact item (two 16-byte slots) never straddles two work units
    const int cdiv = g_exp[1].load() > 0 ? g_exp[1].load() : 1;
    uint64_t cc = chunk / (uint64_t)cdiv;
    if (cc < 8192) cc = 8192;
    if (cc > chunk) cc = chunk;
    plan.chunk_c = (uint32_t)cc & ~1u;
    // The consume pass hands out units in bucket order to one persistent workgroup per CU: the units of the LAST levels are
    // its tail. An evenly loaded hashed bucket is one unit (~45 us on S1); the last `tail_levels` levels are cut into
    // `tail_div` chunks per bucket instead (atomic flush onto rows zeroed by zero_odd_buckets_kernel), so that the
    // workgroups run out of work within ~10 us of each other.
    {
        const int tail_levels = g_exp[3].load() >= 0 ? g_exp[3].load() : 2;
        const int tail_div = g_exp[4].load() > 0 ? g_exp[4].load() : 4;
        for (uint32_t q = 0; q < plan.nbl; ++q) {
            BinLevel &bl = plan.lv[plan.blevel[q]];
            bl.chunk = bl.compact ? plan.chunk_c : plan.chunk;
            if (q + (uint32_t)tail_levels >= plan.nbl && !bl.compact && n_batch >= (1 << 17)) {
                uint64_t c = ((uint64_t)n_batch * plan.pairs / bl.nb) / (uint64_t)tail_div + 1024;   // per-bucket mean / div, + slack
                if (c < 8192) c = 8192;
                if (c < bl.chunk) bl.chunk = (uint32_t)c & ~1u;
            }
            if (bl.chunk < plan.chunk_c) plan.chunk_c = bl.chunk;   // smallest unit size of the plan (sizes the unit list)
        }
    }
    plan.rot_bucket = 0;
    plan.rotf = g_exp[5].load() != 0 ? 1u : 0u;
    if (g_exp[2].load() == 2) plan.rot_bucket = 0xFFFFFFFFu;   // reverse order: the items written last are read first
    if (g_exp[2].load() == 1)
        for (uint32_t q = 0; q < plan.nbl; ++q)
            if (lt.dense[plan.blevel[q]] == 0) { plan.rot_bucket = plan.bstart[q]; break; }
}

// sub-batch so that the item array stays below the cap (default 1.5 GiB, option "bin_batch_mib")
static int64_t bin_batch_samples(int dim, const LevelTable &lt, int64_t n) {
    const size_t item = 8 + 4 * (size_t)lt.feature_dim;
    BinPlan plan;
    make_plan(dim, lt, kTile, plan, choose_acc_kib(dim, lt, n));
    const size_t per_sample = (size_t)(plan.nbl ? plan.nbl : 1) * (1u << (dim - 1)) * item;
    int64_t cap = (int64_t)(((size_t)g_bin_batch_mib.load() << 20) / per_sample);
    cap = cap / kTile * kTile;
    if (cap < kTile) cap = kTile;
    return n < cap ? n : cap;
}

struct BinWorkspace {
    float *gT;
    unsigned char *items;
    uint32_t *cnt;
    uint32_t *totals;
    uint64_t *base;
    uint32_t *unit_first;
    UnitDesc *unit_desc;
    uint32_t *work_counter;   // next unit of the persistent consume pass (zeroed by the bucket scan)
    uint32_t *gmax;  // [SHACIRA_MAX_LODS] bit patterns of max |grad_output| per level
    float *acc32;  // fp32 accumulation image for fp16 tables
    size_t bytes;
};

static BinWorkspace carve(int dim, int dtype, const LevelTable &lt, int64_t n, void *ws) {
    BinPlan plan;
    const int64_t nb = bin_batch_samples(dim, lt, n);
    make_plan(dim, lt, nb, plan, choose_acc_kib(dim, lt, n));
    const size_t item = 8 + 4 * (size_t)lt.feature_dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // gT and gmax first: their offsets must not depend on the level range of the call (REUSE_STAGED calls share them)
    const size_t o_gT = take((size_t)n * lt.num_lods * lt.feature_dim * sizeof(float));
    const size_t o_gmax = take(SHACIRA_MAX_LODS * sizeof(uint32_t));
    const size_t o_items = take((size_t)nb * plan.nbl * plan.pairs * item);
    const size_t o_cnt = take((size_t)plan.total_buckets * plan.num_tiles * sizeof(uint32_t));
    const size_t o_tot = take((size_t)(plan.total_buckets + 1) * sizeof(uint32_t));
    const size_t o_base = take((size_t)(kMaxBuckets + 2) * sizeof(uint64_t));
    const size_t o_unit = take((size_t)(kMaxBuckets + 2) * sizeof(uint32_t));
    const uint64_t max_items_ws = (uint64_t)nb * plan.nbl * plan.pairs;
    const size_t o_ub = take((size_t)(max_items_ws / plan.chunk_c + plan.total_buckets + 2) * sizeof(UnitDesc));
    const size_t o_wc = take(256);
    const size_t o_acc = take(dtype == SHACIRA_F16 ? (size_t)lt.table_rows * lt.feature_dim * sizeof(float) : 0);
    BinWorkspace w{};
    unsigned char *p = static_cast<unsigned char *>(ws);
    if (p) {
        w.gT = reinterpret_cast<float *>(p + o_gT);
        w.items = p + o_items;
        w.cnt = reinterpret_cast<uint32_t *>(p + o_cnt);
        w.totals = reinterpret_cast<uint32_t *>(p + o_tot);
        w.base = reinterpret_cast<uint64_t *>(p + o_base);
        w.unit_first = reinterpret_cast<uint32_t *>(p + o_unit);
        w.unit_desc = reinterpret_cast<UnitDesc *>(p + o_ub);
        w.work_counter = reinterpret_cast<uint32_t *>(p + o_wc);
        w.gmax = reinterpret_cast<uint32_t *>(p + o_gmax);
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

// Side stream for the passes that do not depend on the transposed gradient (count + scans): they are compute/LDS
// bound while the transpose and the direct levels are memory/LDS bound, so they share the chip well. One per host
// thread and device; fork/join with events keeps the caller's stream semantics (and is capturable in a HIP graph once
// the objects exist -- they are created on the first eager call).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, zeroed = nullptr, staged = nullptr, group = nullptr, done = nullptr;
};
static hipError_t side_stream(SideStream **out) {
    static thread_local SideStream per_device[16];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    SideStream &ss = per_device[dev];
    if (!ss.stream) {
        hipStream_t st;
        hipEvent_t a, b, c, d, g2, d2;
        if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&a, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&b, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&c, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&d, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&g2, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&d2, hipEventDisableTiming)) != hipSuccess) return e;
        ss.fork = a;
        ss.join = b;
        ss.zeroed = c;
        ss.staged = d;
        ss.group = g2;
        ss.done = d2;
        ss.stream = st;
    }
    *out = &ss;
    return hipSuccess;
}

// pass A grid: tiles x level shares, at least ~1024 workgroups when the batch is small
static dim3 count_grid(const BinPlan &plan) {
    uint32_t shares = plan.num_tiles >= 1024 ? 1u : (1024u + plan.num_tiles - 1) / plan.num_tiles;
    if (shares > plan.nbl) shares = plan.nbl;
    return dim3(plan.num_tiles, shares < 1 ? 1 : shares);
}

template <int DIM, int F>
static hipError_t run_bin(int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                          const void *grad_out, float *acc, const BinWorkspace &w, int64_t n, hipStream_t s,
                          bool zero_table) {
    const int L = lt.num_lods;
    BinPlan whole;
    const int acc_kib = choose_acc_kib(DIM, lt, n);
    make_plan(DIM, lt, n, whole, acc_kib);
    const int64_t nb = bin_batch_samples(DIM, lt, n);
    const bool multi = nb < n;
    const bool stage_all = (lt.stage_flags & SHACIRA_BWD_STAGE_ALL_LEVELS) != 0;
    const bool staged = (lt.stage_flags & SHACIRA_BWD_REUSE_STAGED) != 0;
    // only binned levels consume the transposed gradients (a later call on this workspace may, too: stage_all)
    // "rows" mode: the scatter pass reads grad_output itself (XCD-affine grid), no transposed image is built at all
    const int rows_opt = g_bwd_rows.load();
    const bool rows_mode = rows_opt != 0 && whole.nbl > 0 && dtype == SHACIRA_F32 && !stage_all && !staged && !multi &&
                           (rows_opt == 2 || n >= (1 << 18));
    const bool need_T = !rows_mode && (whole.nbl > 0 || stage_all || staged);
    // single sub-batch (the usual case): a side stream takes what is off the critical path. With a transpose to do,
    // counting is FUSED into it (transpose_count_kernel) and the side stream zeroes the table and runs the direct
    // levels; a call that reuses staged gradients counts + scans on the side stream instead.
    SideStream *ss = nullptr;
    // measured (tools/bwd_rules_sweep.py, S1 table): 3-D: the one-stream order is 4-5 % faster at 256 K - 320 K samples, equal at
    // 400 K, 3-5 % slower from 2^19; 2-D: the fork wins from 2^18 (equal at 192 K); a loss at 64 K and 128 K (also fused)
    // (thresholds in units of n * L * F so that heavier tables fork earlier: nerf_lego.yaml's 24-level F = 4 table gains 7 %
    // from the fork at 256 K samples, where the 16-level F = 2 table loses 5 %)
    const int64_t fork_work = n * lt.num_lods * lt.feature_dim;
    const bool can_fork = whole.nbl > 0 && !multi && g_bwd_fork.load() != 0 &&
                          fork_work >= (DIM == 3 ? ((int64_t)7 << 21) : ((int64_t)1 << 23));
    // measured (tools/fuse_check.py, re-measured at the end of round 2): 3-D: fused wins by 2-4 % up to 786 K samples and
    // loses 3 % at 2^20; 2-D: fused wins by 2-8 % at every size tried (2^18 ... 2^21) ("bwd_fuse": 0 = never, 1 = by that
    // rule, 2 = always)
    const int fuse_opt = g_bwd_fuse.load();
    const bool fuse_rule = (DIM == 2) ? n <= ((int64_t)1 << 22) : n <= ((int64_t)3 << 18);
    const bool fuse = can_fork && need_T && !staged && (fuse_opt == 2 || (fuse_opt == 1 && fuse_rule));
    // selective zeroing (see zero_unowned_rows_kernel): a single sub-batch whose plan has hashed binned levels
    bool any_hashed = false;
    for (uint32_t q = 0; q < whole.nbl; ++q) any_hashed = any_hashed || lt.dense[whole.blevel[q]] == 0;
    const bool selective = zero_table && !multi && any_hashed && g_bwd_selective_zero.load() != 0;
    auto zero_acc = [&](hipStream_t zs) -> hipError_t {
        if (!selective)
            return hipMemsetAsync(acc, 0, (size_t)lt.table_rows * lt.feature_dim * sizeof(float), zs);
        hipLaunchKernelGGL(zero_unowned_rows_kernel, dim3(256, (uint32_t)L), dim3(256), 0, zs, acc, first_idx, lt, whole);
        return hipGetLastError();
    };
    auto zero_odd_buckets = [&](const BinPlan &plan, hipStream_t zs) -> hipError_t {   // after bin_scan_buckets_kernel
        if (!selective) return hipSuccess;
        hipLaunchKernelGGL(zero_odd_buckets_kernel, dim3(kMaxLevelBuckets, plan.nbl), dim3(256), 0, zs, acc, first_idx,
                           w.unit_first, lt, plan);
        return hipGetLastError();
    };
    if (can_fork) {
        hipError_t e = side_stream(&ss);
        if (e != hipSuccess) return e;
        if ((e = hipEventRecord(ss->fork, s)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(ss->stream, ss->fork, 0)) != hipSuccess) return e;
        if (zero_table) {   // at::zeros_like of the reference: off the critical path, next to the transpose
            if ((e = zero_acc(ss->stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(ss->zeroed, ss->stream)) != hipSuccess) return e;
        }
        if (!fuse) {
            BinPlan plan;
            make_plan(DIM, lt, n, plan, acc_kib);
            const dim3 grid(plan.num_tiles, plan.nbl);
            hipLaunchKernelGGL((bin_count_levels_kernel<DIM>), count_grid(plan), dim3(kBinThreads), 0, ss->stream, lt,
                               plan, coords, w.cnt, (int64_t)0, n);
            SHACIRA_CHECK_LAUNCH();
            hipLaunchKernelGGL(bin_scan_tiles_kernel, dim3((plan.total_buckets + kScanBuckets - 1) / kScanBuckets), dim3(64 * kScanWaves), 0,
                               ss->stream, w.cnt, w.totals, plan.num_tiles, plan.total_buckets);
            SHACIRA_CHECK_LAUNCH();
            hipLaunchKernelGGL(bin_scan_buckets_kernel, dim3(1), dim3(1024), 0, ss->stream, w.totals, w.base,
                               w.unit_first, w.unit_desc, plan.total_buckets, plan.chunk, plan, w.work_counter);
            SHACIRA_CHECK_LAUNCH();
            if ((e = zero_odd_buckets(plan, ss->stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(ss->join, ss->stream)) != hipSuccess) return e;
        }
    }
    if (zero_table && !ss) {
        hipError_t e = zero_acc(s);
        if (e != hipSuccess) return e;
    }
    // fixed-point images pay off once the accumulation itself dominates; small batches are bound by fixed costs and
    // keep the fp64 image (and skip the gmax bookkeeping): measured 100 vs 107 us at 65 536 samples
    const bool use_fx = (need_T || rows_mode) && n >= (1 << 17);
    if (!staged && use_fx) {   // per-level max |grad_output| for the scales (kept in the workspace for REUSE_STAGED)
        hipError_t e = hipMemsetAsync(w.gmax, 0, SHACIRA_MAX_LODS * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    if (fuse) {
        // passes T + A in one kernel, then the scans, all on the caller's stream
        const int t_lb = stage_all ? 0 : lt.level_begin, t_le = stage_all ? L : lt.level_end;
        const size_t shmem = (size_t)256 * (L + 1) * F * sizeof(float);
        BinPlan plan;
        make_plan(DIM, lt, n, plan, acc_kib);
        const dim3 grid(plan.num_tiles);
        if (dtype == SHACIRA_F32 && use_fx)
            hipLaunchKernelGGL((transpose_count_kernel<DIM, float, F, true>), grid, dim3(kFuseThreads), shmem, s, lt, plan,
                               static_cast<const float *>(grad_out), w.gT, coords, w.cnt, n, t_lb, t_le, w.gmax);
        else if (dtype == SHACIRA_F32)
            hipLaunchKernelGGL((transpose_count_kernel<DIM, float, F, false>), grid, dim3(kFuseThreads), shmem, s, lt, plan,
                               static_cast<const float *>(grad_out), w.gT, coords, w.cnt, n, t_lb, t_le, nullptr);
        else if (use_fx)
            hipLaunchKernelGGL((transpose_count_kernel<DIM, __half, F, true>), grid, dim3(kFuseThreads), shmem, s, lt, plan,
                               static_cast<const __half *>(grad_out), w.gT, coords, w.cnt, n, t_lb, t_le, w.gmax);
        else
            hipLaunchKernelGGL((transpose_count_kernel<DIM, __half, F, false>), grid, dim3(kFuseThreads), shmem, s, lt, plan,
                               static_cast<const __half *>(grad_out), w.gT, coords, w.cnt, n, t_lb, t_le, nullptr);
        SHACIRA_CHECK_LAUNCH();
        hipError_t e;
        if ((e = hipEventRecord(ss->staged, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(bin_scan_tiles_kernel, dim3((plan.total_buckets + kScanBuckets - 1) / kScanBuckets), dim3(64 * kScanWaves), 0, s,
                           w.cnt, w.totals, plan.num_tiles, plan.total_buckets);
        SHACIRA_CHECK_LAUNCH();
        hipLaunchKernelGGL(bin_scan_buckets_kernel, dim3(1), dim3(1024), 0, s, w.totals, w.base, w.unit_first,
                           w.unit_desc, plan.total_buckets, plan.chunk, plan, w.work_counter);
        SHACIRA_CHECK_LAUNCH();
        if ((e = zero_odd_buckets(plan, s)) != hipSuccess) return e;
    } else if (need_T && !staged) {
        const int t_lb = stage_all ? 0 : lt.level_begin, t_le = stage_all ? L : lt.level_end;
        // pass T over the whole batch (also gathers gmax)
        const uint32_t blocks = (uint32_t)((n + 255) / 256);
        const size_t shmem = (size_t)256 * (L + 1) * F * sizeof(float);
        // 16-byte form: rows of whole 16-byte vectors, 16-byte aligned input, N even for the two-sample output vectors
        const size_t esz = dtype == SHACIRA_F32 ? 4 : 2;
        const int kvec = (int)(16 / (esz * F));
        int ts16 = g_exp[0].load();   // samples per tile (exp0: 0 = 8-byte kernel, else tile size)
        if (ts16 < 0) ts16 = 512;
        while (ts16 >= 128 && (size_t)L * (ts16 + 2) * F * sizeof(float) > (size_t)72 * 1024) ts16 /= 2;
        const bool t16 = ts16 >= 128 && ts16 % 128 == 0 && (L % kvec) == 0 && (F == 4 || (n % 2) == 0) &&
                         (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0;
        if (t16) {
            const uint32_t blocks16 = (uint32_t)((n + ts16 - 1) / ts16);
            const size_t shmem16 = (size_t)L * (ts16 + 2) * F * sizeof(float);
            if (dtype == SHACIRA_F32 && use_fx)
                hipLaunchKernelGGL((transpose_grad16_kernel<float, F, true>), dim3(blocks16), dim3(256), shmem16, s,
                                   static_cast<const float *>(grad_out), w.gT, n, L, t_lb, t_le, ts16, w.gmax);
            else if (dtype == SHACIRA_F32)
                hipLaunchKernelGGL((transpose_grad16_kernel<float, F, false>), dim3(blocks16), dim3(256), shmem16, s,
                                   static_cast<const float *>(grad_out), w.gT, n, L, t_lb, t_le, ts16, nullptr);
            else if (use_fx)
                hipLaunchKernelGGL((transpose_grad16_kernel<__half, F, true>), dim3(blocks16), dim3(256), shmem16, s,
                                   static_cast<const __half *>(grad_out), w.gT, n, L, t_lb, t_le, ts16, w.gmax);
            else
                hipLaunchKernelGGL((transpose_grad16_kernel<__half, F, false>), dim3(blocks16), dim3(256), shmem16, s,
                                   static_cast<const __half *>(grad_out), w.gT, n, L, t_lb, t_le, ts16, nullptr);
        } else if (dtype == SHACIRA_F32 && use_fx)
            hipLaunchKernelGGL((transpose_grad_kernel<float, F, true>), dim3(blocks), dim3(256), shmem, s,
                               static_cast<const float *>(grad_out), w.gT, n, L, t_lb, t_le, w.gmax);
        else if (dtype == SHACIRA_F32)
            hipLaunchKernelGGL((transpose_grad_kernel<float, F, false>), dim3(blocks), dim3(256), shmem, s,
                               static_cast<const float *>(grad_out), w.gT, n, L, t_lb, t_le, nullptr);
        else if (use_fx)
            hipLaunchKernelGGL((transpose_grad_kernel<__half, F, true>), dim3(blocks), dim3(256), shmem, s,
                               static_cast<const __half *>(grad_out), w.gT, n, L, t_lb, t_le, w.gmax);
        else
            hipLaunchKernelGGL((transpose_grad_kernel<__half, F, false>), dim3(blocks), dim3(256), shmem, s,
                               static_cast<const __half *>(grad_out), w.gT, n, L, t_lb, t_le, nullptr);
        SHACIRA_CHECK_LAUNCH();
    }
    // When nothing is transposed (every level is direct: the image configs) gmax would cost an extra read of grad_output
    // (tried: a streaming abs-max kernel); measured on config B it costs more than the faster atomics return (0.103 vs
    // 0.082 ms for the whole backward), so those calls keep the fp64 image.
    // direct levels: one pass over the whole batch, no items (they add into the zeroed table)
    hipStream_t ds = s;   // stream of the direct levels
    // two-stream form without the fused count: the direct levels (LDS bound) follow the count + scans on the side
    // stream and run beside the scatter pass (write bound) instead of in front of it
    const bool direct_side = ss && !fuse && need_T && !staged && whole.ngroups > 0 && whole.nbl > 0 &&
                             g_bwd_direct_side.load() != 0 && g_bwd_groups.load() <= 1;
    if (direct_side) {
        hipError_t e = hipEventRecord(ss->staged, s);
        if (e != hipSuccess) return e;
    }
    if (fuse || direct_side) {   // side stream: after its memset, once the staged gradients (and gmax) exist
        hipError_t e = hipStreamWaitEvent(ss->stream, ss->staged, 0);
        if (e != hipSuccess) return e;
        ds = ss->stream;
    }
    if (whole.ngroups > 0) {
        if (zero_table && ss && !fuse && !direct_side) {
            hipError_t e = hipStreamWaitEvent(s, ss->zeroed, 0);
            if (e != hipSuccess) return e;
        }
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
                               acc_bytes, ds, lt, plan, first_idx, coords, w.gT, acc, n, w.gmax, headroom);
        else if (need_T)
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, float, true, false>), grid, dim3(kConsumeThreads),
                               acc_bytes, ds, lt, plan, first_idx, coords, w.gT, acc, n, nullptr, headroom);
        else if (dtype == SHACIRA_F32)
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, float, false, false>), grid, dim3(kConsumeThreads),
                               acc_bytes, ds, lt, plan, first_idx, coords, static_cast<const float *>(grad_out), acc, n,
                               nullptr, headroom);
        else
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, __half, false, false>), grid, dim3(kConsumeThreads),
                               acc_bytes, ds, lt, plan, first_idx, coords, static_cast<const __half *>(grad_out), acc, n,
                               nullptr, headroom);
        SHACIRA_CHECK_LAUNCH();
    }
    if (fuse) {   // direct levels (and the table zeroing before them) join the caller's stream before the consume pass
        hipError_t e = hipEventRecord(ss->join, ss->stream);
        if (e != hipSuccess) return e;
    }
    if (direct_side) {
        hipError_t e = hipEventRecord(ss->done, ss->stream);
        if (e != hipSuccess) return e;
    }
    if (whole.nbl == 0) return hipSuccess;
    if (ss && !fuse) {
        hipError_t e = hipStreamWaitEvent(s, ss->join, 0);
        if (e != hipSuccess) return e;
    }
    constexpr int NP = 1 << (DIM - 1);
    const size_t stage = (size_t)kTile * NP * (sizeof(Item<F>) + 1);
    auto consume = [&](const BinPlan &plan, uint32_t grid_units, uint32_t b_lo, uint32_t b_hi, int force_atomic,
                       hipStream_t cs) -> hipError_t {
        const size_t acc_bytes = (size_t)plan.BR * F * sizeof(double);
        // whole-plan launches are persistent: as many workgroups as the chip holds fetch units from the work counter
        // (measured: S1 backward 0.611 -> 0.595 ms, 2-D 0.375 -> 0.369; at 65 536 samples the hardware's own dispatch of
        // 1 400 tiny workgroups is 5 us faster, so small batches keep it)
        uint32_t *wc = (b_lo == 0 && b_hi == plan.total_buckets && g_bwd_persistent.load() != 0 && n >= (1 << 17))
                           ? w.work_counter : nullptr;
        if (wc != nullptr && grid_units > 512u) grid_units = 512u;
        if (use_fx)   // a unit streams <= chunk items
            hipLaunchKernelGGL((bin_consume_kernel<F, true>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, cs, lt,
                               plan, first_idx, w.base, w.unit_first, w.unit_desc,
                               reinterpret_cast<const Item<F> *>(w.items), acc, force_atomic, w.gmax,
                               fx_headroom((uint64_t)plan.chunk + 1), b_lo, b_hi, wc);
        else
            hipLaunchKernelGGL((bin_consume_kernel<F, false>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, cs, lt,
                               plan, first_idx, w.base, w.unit_first, w.unit_desc,
                               reinterpret_cast<const Item<F> *>(w.items), acc, force_atomic, nullptr, -1, b_lo, b_hi, wc);
        return hipGetLastError();
    };
    // Level groups: the scatter pass is bound by the chip's WRITE rate (3.4 TB/s: a kernel that only stores the items
    // takes 195 of its 254 us), the consume pass by reads and LDS atomics -- opposite directions of the fabric. With the
    // binned levels cut into groups, the caller's stream scatters group g + 1 while the side stream consumes group g.
    const int groups_opt = g_bwd_groups.load();
    if (ss && !multi && groups_opt > 1 && whole.nbl >= 2) {
        const BinPlan &plan = whole;
        const uint32_t G = (uint32_t)groups_opt < plan.nbl ? (uint32_t)groups_opt : plan.nbl;
        // equal item bytes per group (a compact level moves half of what a pair level does)
        uint32_t weight[SHACIRA_MAX_LODS], total = 0;
        for (uint32_t q = 0; q < plan.nbl; ++q) total += (weight[q] = plan.lv[plan.blevel[q]].compact ? 1u : 2u);
        uint32_t q0 = 0, spent = 0;
        for (uint32_t g = 0; g < G && q0 < plan.nbl; ++g) {
            uint32_t q1 = q0, acc_w = 0;
            const uint32_t target = (total - spent + (G - g) - 1) / (G - g);
            while (q1 < plan.nbl && (acc_w < target || g + 1 == G)) acc_w += weight[q1++];
            spent += acc_w;
            const dim3 grid(plan.num_tiles, q1 - q0);
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, false>), grid, dim3(kBinThreads), stage, s, lt, plan, coords,
                               w.gT, w.cnt, w.base, reinterpret_cast<Item<F> *>(w.items), (int64_t)0, n, n, q0, q1 - q0,
                               (uint32_t *)nullptr);
            SHACIRA_CHECK_LAUNCH();
            hipError_t e = hipEventRecord(ss->group, s);
            if (e != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(ss->stream, ss->group, 0)) != hipSuccess) return e;
            const BinLevel &first = plan.lv[plan.blevel[q0]], &last = plan.lv[plan.blevel[q1 - 1]];
            const uint32_t b_lo = first.bucket0, b_hi = last.bucket0 + last.nb;
            const uint64_t max_items = (uint64_t)n * NP * (q1 - q0);
            const uint32_t max_units = (uint32_t)(max_items / plan.chunk_c) + (b_hi - b_lo) + 1;
            if ((e = consume(plan, max_units, b_lo, b_hi, 0, ss->stream)) != hipSuccess) return e;
            q0 = q1;
        }
        hipError_t e = hipEventRecord(ss->done, ss->stream);   // the side stream carries everything the table waits for:
        if (e != hipSuccess) return e;                         // its zeroing, (fused mode) the direct levels, the consumes
        return hipStreamWaitEvent(s, ss->done, 0);
    }
    for (int64_t s0 = 0; s0 < n; s0 += nb) {
        const int64_t hi = (s0 + nb < n) ? (s0 + nb) : n;
        BinPlan plan;
        make_plan(DIM, lt, hi - s0, plan, acc_kib);
        const dim3 grid(plan.num_tiles, plan.nbl);
        if (!ss) {
            hipLaunchKernelGGL((bin_count_levels_kernel<DIM>), count_grid(plan), dim3(kBinThreads), 0, s, lt, plan,
                               coords, w.cnt, s0, hi);
            SHACIRA_CHECK_LAUNCH();
            hipLaunchKernelGGL(bin_scan_tiles_kernel, dim3((plan.total_buckets + kScanBuckets - 1) / kScanBuckets), dim3(64 * kScanWaves), 0, s,
                               w.cnt, w.totals, plan.num_tiles, plan.total_buckets);
            SHACIRA_CHECK_LAUNCH();
            hipLaunchKernelGGL(bin_scan_buckets_kernel, dim3(1), dim3(1024), 0, s, w.totals, w.base, w.unit_first,
                               w.unit_desc, plan.total_buckets, plan.chunk, plan, w.work_counter);
            SHACIRA_CHECK_LAUNCH();
            hipError_t ze = zero_odd_buckets(plan, s);
            if (ze != hipSuccess) return ze;
        }
        if (rows_mode) {   // 1-D grid, XCD-affine numbering (see the kernel)
            const uint32_t per_xcd = (plan.num_tiles + 7) / 8;
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, true>), dim3(8u * per_xcd * plan.nbl), dim3(kBinThreads), stage,
                               s, lt, plan, coords, static_cast<const float *>(grad_out), w.cnt, w.base,
                               reinterpret_cast<Item<F> *>(w.items), s0, hi, n, 0u, plan.nbl, use_fx ? w.gmax : nullptr);
        } else {
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, false>), grid, dim3(kBinThreads), stage, s, lt, plan, coords,
                               w.gT, w.cnt, w.base, reinterpret_cast<Item<F> *>(w.items), s0, hi, n, 0u, plan.nbl,
                               (uint32_t *)nullptr);
        }
        SHACIRA_CHECK_LAUNCH();
        const uint64_t max_items = (uint64_t)(hi - s0) * plan.nbl * NP;
        const uint32_t max_units = (uint32_t)(max_items / plan.chunk_c) + plan.total_buckets + 1;
        if (fuse) {
            hipError_t e = hipStreamWaitEvent(s, ss->join, 0);
            if (e != hipSuccess) return e;
        }
        if (direct_side) {
            hipError_t e = hipStreamWaitEvent(s, ss->done, 0);
            if (e != hipSuccess) return e;
        }
        hipError_t e = consume(plan, max_units, 0u, plan.total_buckets, multi ? 1 : 0, s);
        if (e != hipSuccess) return e;
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
        set(reinterpret_cast<const void *>(&transpose_grad16_kernel<TT, FF, true>), 72 * 1024);         \
        set(reinterpret_cast<const void *>(&transpose_grad16_kernel<TT, FF, false>), 72 * 1024);        \
        set(reinterpret_cast<const void *>(&transpose_count_kernel<2, TT, FF, true>), 140 * 1024);      \
        set(reinterpret_cast<const void *>(&transpose_count_kernel<2, TT, FF, false>), 140 * 1024);     \
        set(reinterpret_cast<const void *>(&transpose_count_kernel<3, TT, FF, true>), 140 * 1024);      \
        set(reinterpret_cast<const void *>(&transpose_count_kernel<3, TT, FF, false>), 140 * 1024);
        SHACIRA_T_ATTR(float, 2) SHACIRA_T_ATTR(float, 4) SHACIRA_T_ATTR(__half, 2) SHACIRA_T_ATTR(__half, 4)
#undef SHACIRA_T_ATTR
#define SHACIRA_DIRECT_ATTR(D, FF)                                                                              \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, true, true>), 16384 * sizeof(double));   \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, true, false>), 16384 * sizeof(double));  \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, false, false>), 16384 * sizeof(double)); \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, __half, false, false>), 16384 * sizeof(double));
        SHACIRA_DIRECT_ATTR(2, 2) SHACIRA_DIRECT_ATTR(2, 4) SHACIRA_DIRECT_ATTR(3, 2) SHACIRA_DIRECT_ATTR(3, 4)
#undef SHACIRA_DIRECT_ATTR
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 2, false>), (size_t)kTile * 2 * (sizeof(Item<2>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 2, true>), (size_t)kTile * 2 * (sizeof(Item<2>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 4, false>), (size_t)kTile * 2 * (sizeof(Item<4>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 4, true>), (size_t)kTile * 2 * (sizeof(Item<4>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 2, false>), (size_t)kTile * 4 * (sizeof(Item<2>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 2, true>), (size_t)kTile * 4 * (sizeof(Item<2>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 4, false>), (size_t)kTile * 4 * (sizeof(Item<4>) + 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 4, true>), (size_t)kTile * 4 * (sizeof(Item<4>) + 1));
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
