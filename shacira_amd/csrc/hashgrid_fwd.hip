// hashgrid_fwd.hip -- multi-resolution hash-grid lookup + bi/trilinear interpolation, forward (gfx950).
//
// Replaces hashgrid_interpolate_cuda / hashgrid_interpolate2d_cuda
// (wisp/csrc/ops/hashgrid_interpolate.cpp:44-66, :130-152; kernels hashgrid_interpolate_cuda.cu:47-109 and
// hashgrid_interpolate2d_cuda.cu:44-99). One launch covers every level (the reference launches L kernels).
//
// What bounds it (measured): a wave-level gather costs ~20 clk + 2 clk per distinct 128-byte line
// it touches, and lines fetched from beyond the XCD's L2 cost a full Infinity-Cache request. The kernels below differ
// in how they keep that count down; all of them produce bit-identical results (same fp64 coordinate scale, same
// corner order, same fmaf chain as the reference):
//   variant 0  lane = (sample, level), one gather per corner: the reference's own kernel shape (any even F; kept for A/B)
//   variant 3  lane pair = sample (x / x+1 corners merge into one request), sample-major   [default: 2-D, small N]
//   variant 6  one level per XCD at a time + lane pairing, level-major staging + transposing copy [default: 3-D, N >= 16 K]
//   variant 8  cell-sorted forward (hashgrid_tiled.hip) over this file's rows / level-pair kernels [default: large batches]
//   variant 9  small tables (every level fits an LDS image: the Kodak tables): groups of consecutive levels held in LDS,
//              lane = (sample, level) like variant 0 -> corner reads are ds_read, outputs coalesce    [default: such tables, large N]
// (option "fwd_variant"; -1 = the measured rule. Variants 1, 2, 4, 5, 7 of rounds 1-2 lost and were removed: git 4a7dfa7)
// HBM-bound: algorithmic bytes per sample = 4*DIM + L*2^DIM*F*s + L*F*s.
#include <mutex>

#include "internal.h"

namespace shacira {

static bool use_staged(int dim, const LevelTable &lt, int64_t n);
static bool use_lds_tables(int dim, int esz, const LevelTable &lt, int64_t n);

template <typename T, int F> struct RowVec;  // one table row as a single vector access
template <> struct RowVec<float, 2> { using type = float2; };
template <> struct RowVec<float, 4> { using type = float4; };
template <> struct RowVec<__half, 2> { using type = uint32_t; };
template <> struct RowVec<__half, 4> { using type = uint2; };
template <> struct RowVec<__half, 8> { using type = uint4; };

template <typename T, int F> __device__ __forceinline__ void load_row(const T *p, float (&v)[F]) {
    if constexpr (sizeof(T) == 4 && F == 2) {
        float2 r = *reinterpret_cast<const float2 *>(p);
        v[0] = r.x; v[1] = r.y;
    } else if constexpr (sizeof(T) == 4 && F == 4) {
        float4 r = *reinterpret_cast<const float4 *>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    } else if constexpr (sizeof(T) == 2 && F == 2) {
        __half2 r = *reinterpret_cast<const __half2 *>(p);
        v[0] = __low2float(r); v[1] = __high2float(r);
    } else {
#pragma unroll
        for (int j = 0; j < F; ++j) v[j] = Scalar<T>::load(p + j);
    }
}

// One table row for a lane that may have none (sample beyond the batch, row beyond the table): the load is UNCONDITIONAL from
// a clamped row and the result is masked afterwards. (Round 4: as `if (ok) load else 0` every gather sat in its own
// lane-dependent branch and the compiler closed each branch with `s_waitcnt vmcnt(0)` -- the four corner gathers of a sample
// went out one round trip after the other instead of together; ISA of hashgrid_fwd_level_pair_kernel.)
template <typename T, int F>
__device__ __forceinline__ void gather_row(const T *__restrict__ table, int64_t grow, bool ok, float (&v)[F]) {
    const int64_t safe = ok ? grow : 0;
    load_row<T, F>(table + safe * F, v);
#pragma unroll
    for (int j = 0; j < F; ++j) v[j] = ok ? v[j] : 0.0f;
}

template <typename T, int F> __device__ __forceinline__ void store_row(T *p, const float (&v)[F]) {
    if constexpr (sizeof(T) == 4 && F == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else if constexpr (sizeof(T) == 4 && F == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (sizeof(T) == 2 && F == 2) {
        *reinterpret_cast<__half2 *>(p) = __floats2half2_rn(Scalar<__half>::rounded_fp32(v[0]), Scalar<__half>::rounded_fp32(v[1]));
    } else {
#pragma unroll
        for (int j = 0; j < F; ++j) Scalar<T>::store(p + j, v[j]);
    }
}

// F > 0: compile-time feature_dim; F == 0: runtime feature_dim (any even value), scalar row access.
template <int DIM, typename T, int F>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(LevelTable lt, const int32_t *__restrict__ first_idx,
                                                           const float *__restrict__ coords,
                                                           const T *__restrict__ table, T *__restrict__ feats,
                                                           int64_t sample0, uint32_t num_items) {
    constexpr int NC = 1 << DIM;
    const uint32_t L = (uint32_t)lt.num_lods;
    __shared__ int32_t s_res[SHACIRA_MAX_LODS];
    __shared__ float s_hi[SHACIRA_MAX_LODS];
    __shared__ int32_t s_first[SHACIRA_MAX_LODS];
    __shared__ uint8_t s_dense[SHACIRA_MAX_LODS];
    if (threadIdx.x < L) {
        s_res[threadIdx.x] = lt.res[threadIdx.x];
        s_hi[threadIdx.x] = lt.hi[threadIdx.x];
        s_dense[threadIdx.x] = lt.dense[threadIdx.x];
        s_first[threadIdx.x] = first_idx[threadIdx.x];
    }
    __syncthreads();

    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < num_items; w += stride) {
        const uint32_t s = w / L;
        const uint32_t lvl = w - s * L;
        const int64_t i = sample0 + s;
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
        Corners<DIM> c;
        compute_corners<DIM>(t, s_res[lvl], s_hi[lvl], s_dense[lvl] != 0, lt.mask, c);
        const int64_t base = (int64_t)s_first[lvl];
        if constexpr (F > 0) {
            float acc[F];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int64_t row = base + (int64_t)c.row[k];
                float v[F];
                gather_row<T, F>(table, row, (uint64_t)row < (uint64_t)lt.table_rows, v);
#pragma unroll
                for (int j = 0; j < F; ++j) acc[j] = (k == 0) ? v[j] * c.w[0] : fmaf(v[j], c.w[k], acc[j]);
            }
            store_row<T, F>(feats + (i * L + lvl) * F, acc);
        } else {
            const int Fr = lt.feature_dim;
            for (int j = 0; j < Fr; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int64_t row = base + (int64_t)c.row[k];
                    float v = ((uint64_t)row < (uint64_t)lt.table_rows) ? Scalar<T>::load(table + row * Fr + j) : 0.0f;
                    acc = (k == 0) ? v * c.w[0] : fmaf(v, c.w[k], acc);
                }
                Scalar<T>::store(feats + (i * L + lvl) * Fr + j, acc);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Variant 9 (round 3): LDS-resident tables. The image configs' tables are a few hundred KB (configs B / C: 26 704 rows =
// 0.2 MB; kodak.yaml: 0.3 MB): too large for the 32 KiB L1, so every corner gather of variants 0 / 3 is an L2 -> L1 line
// transfer (2 clk per 128-byte line for 8-16 useful bytes) -- the forward of the 24-image batch ran at the line rate, not at
// any byte rate. Here a workgroup copies a GROUP of G consecutive levels (<= 128 KiB of rows) into LDS once and then walks
// its share of the samples: corner reads are ds_read_b64 / b128. A sample is served by P = G / LPC adjacent lanes, lane p
// owning the LPC levels that make the p-th 16-byte piece of the group's part of the output row (its level parameters stay
// in registers for the whole walk): a wave's store instruction writes whole contiguous runs (P = 4: 16 samples x 64 bytes)
// instead of 64 pieces in 64 lines. grid = (workgroups per group, groups). Arithmetic is variant 0's (same corner order,
// same fmaf chain): bit-identical.
constexpr int kLdsFwdThreads = 1024;
constexpr int kLdsFwdMaxGroups = 8;
constexpr size_t kLdsFwdBytes = 128 * 1024;

struct LdsFwdPlan {
    int32_t ngroups;
    int32_t G;                                     // levels per group (a power of two times LPC; the last group may hold fewer)
    int32_t P;                                     // lanes per sample = 16-byte pieces per group = G / LPC (power of two)
    uint32_t loff[SHACIRA_MAX_LODS];               // first row of the level inside its group's LDS image
    uint32_t grows[kLdsFwdMaxGroups];              // rows of the group
};

static uint64_t level_rows(int dim, const LevelTable &lt, int l) {
    if (!lt.dense[l]) return (uint64_t)lt.mask + 1u;
    uint64_t rows = 1;
    for (int a = 0; a < dim; ++a) rows *= (uint64_t)lt.res[l];
    return rows;
}

// The largest G = LPC * 2^k (<= 32 levels) whose blocks of G consecutive levels each fit one image, in at most 8 groups.
// Dense levels must have res <= 256: below that the clamp keeps every corner inside its level (hi < res - 1), so the kernel
// reads LDS rows without a bound check.
static bool make_lds_fwd_plan(int dim, const LevelTable &lt, size_t esz, LdsFwdPlan &p) {
    const size_t row_bytes = (size_t)lt.feature_dim * esz;
    if (row_bytes > 16 || (16 % row_bytes) != 0) return false;
    const int lpc = (int)(16 / row_bytes);
    const uint64_t cap = kLdsFwdBytes / row_bytes;
    for (int l = 0; l < lt.num_lods; ++l)
        if (lt.dense[l] && lt.res[l] > 256) return false;
    for (int G = lpc * 8; G >= lpc; G /= 2) {
        const int ng = (lt.num_lods + G - 1) / G;
        if (ng > kLdsFwdMaxGroups) break;
        bool ok = true;
        for (int g = 0; g < ng && ok; ++g) {
            uint64_t used = 0;
            for (int l = g * G; l < (g + 1) * G && l < lt.num_lods; ++l) {
                p.loff[l] = (uint32_t)used;
                used += level_rows(dim, lt, l);
            }
            ok = used <= cap;
            p.grows[g] = (uint32_t)used;
        }
        if (ok) {
            p.ngroups = ng;
            p.G = G;
            p.P = G / lpc;
            return true;
        }
    }
    return false;
}

template <int DIM, typename T, int F>
__global__ __launch_bounds__(kLdsFwdThreads) void hashgrid_fwd_lds_kernel(LevelTable lt, LdsFwdPlan plan,
                                                                          const int32_t *__restrict__ first_idx,
                                                                          const float *__restrict__ coords,
                                                                          const T *__restrict__ table, T *__restrict__ feats,
                                                                          int64_t n) {
    constexpr int NC = 1 << DIM;
    constexpr int LPC = 16 / (F * (int)sizeof(T));     // levels per 16-byte piece of an output row (1, 2 or 4)
    extern __shared__ __align__(16) unsigned char s_raw[];
    T *s_tab = reinterpret_cast<T *>(s_raw);
    const int g = blockIdx.y;
    const int lb = g * plan.G;
    const int le = (lb + plan.G < lt.num_lods) ? lb + plan.G : lt.num_lods;
    const int L = lt.num_lods;
    // the group's rows, level by level: level l starts at row codebook_first_idx[l] of the table (the header's contract: any
    // layout with ascending starts, e.g. padded or aligned level starts of an external caller -- not only the packed layout
    // the in-tree modules build) and lands at row plan.loff[l] of the image. 16-byte global loads from the first aligned
    // element on; the LDS side is written element-wise, its offset need not be 16-byte aligned.
    for (int l = lb; l < le; ++l) {
        const int64_t row0 = first_idx[l];
        const int64_t want = (int64_t)((l + 1 < le ? plan.loff[l + 1] : plan.grows[g]) - plan.loff[l]);
        int64_t rows = want;
        if (row0 + rows > lt.table_rows) rows = lt.table_rows - row0;     // (a table shorter than its last level: zeros behind)
        if (rows < 0) rows = 0;
        const T *src = table + row0 * F;
        T *dst = s_tab + (size_t)plan.loff[l] * F;
        const int64_t elems = rows * F, total = want * F;
        constexpr int VE = 16 / (int)sizeof(T);
        const int64_t head = (int64_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u)) & 15u) / sizeof(T));
        const int64_t h = head < elems ? head : elems;
        for (int64_t e = threadIdx.x; e < h; e += kLdsFwdThreads) dst[e] = src[e];
        const int64_t nvec = (elems - h) / VE;
        for (int64_t v = threadIdx.x; v < nvec; v += kLdsFwdThreads) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src + h + v * VE);
            T tmp[VE];
            __builtin_memcpy(tmp, &q, 16);
#pragma unroll
            for (int k = 0; k < VE; ++k) dst[h + v * VE + k] = tmp[k];
        }
        for (int64_t e = h + nvec * VE + threadIdx.x; e < elems; e += kLdsFwdThreads) dst[e] = src[e];
        for (int64_t e = elems + threadIdx.x; e < total; e += kLdsFwdThreads) Scalar<T>::store(&dst[e], 0.0f);
    }
    // this lane's piece of the group and the parameters of its LPC levels (constant for the whole walk)
    const uint32_t P = (uint32_t)plan.P;
    const uint32_t piece = threadIdx.x & (P - 1u);
    int32_t res[LPC];
    float hi[LPC];
    bool dense[LPC], have[LPC];
    uint32_t off[LPC];
    bool all = true;
#pragma unroll
    for (int u = 0; u < LPC; ++u) {
        const int l = lb + (int)piece * LPC + u;
        have[u] = l < le;
        all = all && have[u];
        const int lc = have[u] ? l : lb;
        res[u] = lt.res[lc];
        hi[u] = lt.hi[lc];
        dense[u] = lt.dense[lc] != 0;
        off[u] = plan.loff[lc];
    }
    __syncthreads();

    const uint32_t spw = kLdsFwdThreads / P;                       // samples per workgroup and iteration
    const int64_t stride = (int64_t)gridDim.x * spw;
    int64_t i = (int64_t)blockIdx.x * spw + threadIdx.x / P;
    float cn[DIM];
    {
        const int64_t ic = i < n ? i : n - 1;
#pragma unroll
        for (int a = 0; a < DIM; ++a) cn[a] = coords[ic * DIM + a];
    }
    for (; i < n; i += stride) {
        double t[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) t[a] = axis_unit(cn[a]);
        {   // next sample's coordinates while this one is interpolated
            const int64_t in = (i + stride < n) ? i + stride : n - 1;
#pragma unroll
            for (int a = 0; a < DIM; ++a) cn[a] = coords[in * DIM + a];
        }
        float acc[LPC][F];
#pragma unroll
        for (int u = 0; u < LPC; ++u) {
            Corners<DIM> c;
            compute_corners<DIM>(t, res[u], hi[u], dense[u], lt.mask, c);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                float v[F];
                // (no bound check: hashed rows are masked, dense levels have res <= 256 -- the planner's condition)
                load_row<T, F>(s_tab + (size_t)(off[u] + c.row[k]) * F, v);
#pragma unroll
                for (int j = 0; j < F; ++j) acc[u][j] = (k == 0) ? v[j] * c.w[0] : fmaf(v[j], c.w[k], acc[u][j]);
            }
        }
        T *out = feats + ((int64_t)i * L + lb + (int64_t)piece * LPC) * F;
        if (all) {
            T pc[LPC * F];
#pragma unroll
            for (int u = 0; u < LPC; ++u) {
#pragma unroll
                for (int j = 0; j < F; ++j) Scalar<T>::store(&pc[u * F + j], acc[u][j]);
            }
            uint4 pv;
            __builtin_memcpy(&pv, pc, 16);
            *reinterpret_cast<uint4 *>(out) = pv;
        } else {
#pragma unroll
            for (int u = 0; u < LPC; ++u)
                if (have[u]) store_row<T, F>(out + (size_t)u * F, acc[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Variant 3 ("x-pair per lane pair"): two adjacent lanes serve one sample; lane dx in {0,1} gathers the corners
// (x+dx, y+dy, z+dz). Corners x and x+1 of a cell are neighbouring rows (dense levels) or rows that differ only in
// low bits (hashed levels: row = (x ^ h) & mask), i.e. almost always the same 128-byte line -- and lanes of ONE wave
// instruction that hit one line are merged into one L2 request (profiles/r01_microbench2_lds_gather.txt), so this
// halves the L2 request count, the resource the forward is bound by. The even lane pulls its partner's values with
// DPP (no LDS) and runs the reference's fmaf chain in the reference's corner order (k = dx*4 + dy*2 + dz), so the
// result stays bit-identical. Levels are walked in a rolled loop (wave-uniform parameters), results staged in
// wave-private LDS and written as contiguous float4 rows, like variant 1.
template <int DIM, typename T, int F>
__global__ __launch_bounds__(256) void hashgrid_fwd_pair_kernel(LevelTable lt, const int32_t *__restrict__ first_idx,
                                                                const float *__restrict__ coords,
                                                                const T *__restrict__ table, T *__restrict__ feats,
                                                                int64_t N) {
    constexpr int NH = 1 << (DIM - 1);              // corners per lane
    extern __shared__ __align__(16) float s_out[];  // [4 waves][32 samples][LFP]
    const int L = lt.num_lods;
    const int LF = L * F;
    const int LFP = (LF + 3) / 4 * 4 + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dx = lane & 1, sl = lane >> 1;        // x offset of this lane, sample slot within the wave
    float *my = s_out + (size_t)wave * 32 * LFP;
    const int64_t wave_s0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t i = wave_s0 + sl;
    const bool live = i < N;
    double t[DIM];
    load_unit_coords<DIM>(coords, i, N, t);
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const int32_t res = lt.res[l];
        const float hi = lt.hi[l];
        const bool dense = lt.dense[l] != 0;
        int32_t p[DIM];
        float f[DIM], g[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f[a], g[a]);
        const uint32_t ux = (uint32_t)p[0] + (uint32_t)dx;
        const uint32_t r = (uint32_t)res;
        const int64_t base = (int64_t)first_idx[l];
        float v[NH][F];
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int dy = (DIM == 3) ? (q >> 1) : q;
            const int dz = (DIM == 3) ? (q & 1) : 0;
            const uint32_t uy = (uint32_t)p[1] + dy;
            uint32_t row;
            if (dense) {
                row = ux + uy * r;
                if constexpr (DIM == 3) row += ((uint32_t)p[2] + dz) * r * r;
            } else {
                row = ux ^ (uy * kPrimeY);
                if constexpr (DIM == 3) row ^= ((uint32_t)p[2] + dz) * kPrimeZ;
                row &= lt.mask;
            }
            const int64_t grow = base + (int64_t)row;
            gather_row<T, F>(table, grow, live && (uint64_t)grow < (uint64_t)lt.table_rows, v[q]);
        }
        // partner's values: lane ^ 1 (DPP quad_perm [1,0,3,2])
        float pv[NH][F];
#pragma unroll
        for (int q = 0; q < NH; ++q)
#pragma unroll
            for (int j = 0; j < F; ++j)
                pv[q][j] = __builtin_bit_cast(
                    float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[q][j]), 0xB1, 0xF, 0xF, true));
        if (dx == 0) {
            // weights in the reference's order: (wx*wy)*wz, corner k = dx*4 + dy*2 + dz (3-D) / dx*2 + dy (2-D)
            float acc[F];
#pragma unroll
            for (int k = 0; k < 2 * NH; ++k) {
                const int kx = k / NH, q = k % NH;
                const int dy = (DIM == 3) ? (q >> 1) : q;
                const int dz = (DIM == 3) ? (q & 1) : 0;
                float w = (kx ? f[0] : g[0]) * (dy ? f[1] : g[1]);
                if constexpr (DIM == 3) w = w * (dz ? f[2] : g[2]);
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const float tv = kx ? pv[q][j] : v[q][j];
                    acc[j] = (k == 0) ? tv * w : fmaf(tv, w, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < F; ++j) my[sl * LFP + l * F + j] = acc[j];
        }
    }
    const int64_t rows = (N - wave_s0 < 32) ? (N - wave_s0) : 32;
    if (rows <= 0) return;
    T *dst = feats + wave_s0 * LF;
    const int total = (int)rows * LF;
    if constexpr (sizeof(T) == 4) {
        if ((LF & 3) == 0) {
            const int q4 = LF >> 2;
            for (int e = lane; e < total / 4; e += 64) {
                const int rr = e / q4, c4 = e - rr * q4;
                reinterpret_cast<float4 *>(dst)[e] = *reinterpret_cast<const float4 *>(my + rr * LFP + c4 * 4);
            }
            return;
        }
    }
    for (int e = lane; e < total; e += 64) {
        const int rr = e / LF, cc = e - rr * LF;
        Scalar<T>::store(dst + e, my[rr * LFP + cc]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Tiled forward, rows kernel (hashgrid_tiled.hip): variant 3's lane pairing over CELL-SORTED sample records for the coarse
// levels [0, lc) -- consecutive samples sit in the same spatial block, so their corner rows come out of L1 (measured:
// a gather instruction whose lines hit L1 costs ~35 clk against ~84-148 from L2) -- then the fine levels' pieces are
// read from the level-major staging buffer the level-per-XCD kernel wrote, and whole feature rows leave through the
// permutation: feats[perm[i]] = row i (full contiguous rows, 16-byte chunks when the row size allows).
template <int DIM, typename T, int F>
__global__ __launch_bounds__(256) void hashgrid_fwd_rows_kernel(LevelTable lt, const int32_t *__restrict__ first_idx,
                                                                const float4 *__restrict__ sorted4,
                                                                const T *__restrict__ table,
                                                                const T *__restrict__ staged, T *__restrict__ feats,
                                                                int64_t N, int lc, uint32_t xcd_affine) {
    constexpr int NH = 1 << (DIM - 1);
    struct alignas(sizeof(T) * F) Piece { T v[F]; };
    extern __shared__ __align__(16) unsigned char s_rows_raw[];   // [4 waves][32 samples][pitch]
    // XCD-affine numbering (round-robin dispatch, workgroup b on XCD b % 8 -- a speed assumption only): XCD k walks the
    // k-th eighth of the SORTED samples, i.e. one slab of space, so its L2 holds that slab's share of the coarse tables
    // instead of every XCD pulling every table through the fabric (the kernel fetched 390 MB for ~100 MB of inputs)
    uint32_t vb = blockIdx.x;
    if (xcd_affine) {
        const uint32_t per = gridDim.x >> 3;
        vb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    const int L = lt.num_lods;
    const uint32_t row_bytes = (uint32_t)(L * F * sizeof(T));
    const uint32_t pitch = (row_bytes + 15u) / 16u * 16u + 16u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dx = lane & 1, sl = lane >> 1;
    unsigned char *my = s_rows_raw + (size_t)wave * 32 * pitch;
    const int64_t wave_s0 = (int64_t)vb * 128 + wave * 32;
    const int64_t i = wave_s0 + sl;
    const bool live = i < N;
    // one 16-byte record per sample: {x, y, z (0 in 2-D), bit pattern of the sample's original index}
    const float4 c4 = sorted4[live ? i : N - 1];   // (clamped, unconditional: a dead lane's values are never stored)
    double t[DIM];
    t[0] = axis_unit(c4.x);
    t[1] = axis_unit(c4.y);
    if constexpr (DIM == 3) t[2] = axis_unit(c4.z);
#pragma unroll 1
    for (int l = 0; l < lc; ++l) {
        const int32_t res = lt.res[l];
        const float hi = lt.hi[l];
        const bool dense = lt.dense[l] != 0;
        int32_t p[DIM];
        float f[DIM], g[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f[a], g[a]);
        const uint32_t ux = (uint32_t)p[0] + (uint32_t)dx;
        const uint32_t r = (uint32_t)res;
        const int64_t base = (int64_t)first_idx[l];
        if constexpr (DIM == 3 && F == 2 && sizeof(T) == 4) {
            if (dense) {
                // Dense level: rows x and x + 1 are neighbours in memory, so ONE 16-byte load (8-byte aligned) fetches the
                // x-pair. The lane pair splits y instead of x: lane dx takes y + dx, two loads (z, z + 1) instead of four;
                // the partner's four pieces come over DPP and lane 0 runs the corner sum in the reference's order.
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
                float w4[2][4];
                // both 16-byte loads first (unconditional, from a clamped row: see gather_row), masks and the rare single-row
                // fallback afterwards -- with the fallback between them the second load waited for the first (round 4)
                int64_t grow2[2];
                bool ok16[2];
                f32x4u q4[2];
#pragma unroll
                for (int dz = 0; dz < 2; ++dz) {
                    const uint32_t row = (uint32_t)p[0] + ((uint32_t)p[1] + (uint32_t)dx) * r + ((uint32_t)p[2] + dz) * r * r;
                    grow2[dz] = base + (int64_t)row;
                    ok16[dz] = live && (uint64_t)(grow2[dz] + 1) < (uint64_t)lt.table_rows;
                    q4[dz] = *reinterpret_cast<const f32x4u *>(table + (ok16[dz] ? grow2[dz] : 0) * 2);
                }
#pragma unroll
                for (int dz = 0; dz < 2; ++dz) {
                    w4[dz][0] = ok16[dz] ? q4[dz].x : 0.0f; w4[dz][1] = ok16[dz] ? q4[dz].y : 0.0f;
                    w4[dz][2] = ok16[dz] ? q4[dz].z : 0.0f; w4[dz][3] = ok16[dz] ? q4[dz].w : 0.0f;
                }
#pragma unroll
                for (int dz = 0; dz < 2; ++dz) {
                    if (!ok16[dz] && live && (uint64_t)grow2[dz] < (uint64_t)lt.table_rows) {   // last row of the table: x only
                        w4[dz][0] = Scalar<T>::load(table + grow2[dz] * 2);
                        w4[dz][1] = Scalar<T>::load(table + grow2[dz] * 2 + 1);
                    }
                }
                float o4[2][4];
#pragma unroll
                for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        o4[dz][c] = __builtin_bit_cast(
                            float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, w4[dz][c]), 0xB1, 0xF, 0xF, true));
                if (dx == 0) {
                    float acc[F];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {   // corner k: bit 2 -> x, bit 1 -> y, bit 0 -> z (the reference's order)
                        const int kx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
                        const float w = ((kx ? f[0] : g[0]) * (dy ? f[1] : g[1])) * (dz ? f[2] : g[2]);
#pragma unroll
                        for (int j = 0; j < F; ++j) {
                            const float tv = dy ? o4[dz][2 * kx + j] : w4[dz][2 * kx + j];
                            acc[j] = (k == 0) ? tv * w : fmaf(tv, w, acc[j]);
                        }
                    }
                    store_row<T, F>(reinterpret_cast<T *>(my + (size_t)sl * pitch) + l * F, acc);
                }
                continue;
            }
        }
        float v[NH][F];
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int dy = (DIM == 3) ? (q >> 1) : q;
            const int dz = (DIM == 3) ? (q & 1) : 0;
            const uint32_t uy = (uint32_t)p[1] + dy;
            uint32_t row;
            if (dense) {
                row = ux + uy * r;
                if constexpr (DIM == 3) row += ((uint32_t)p[2] + dz) * r * r;
            } else {
                row = ux ^ (uy * kPrimeY);
                if constexpr (DIM == 3) row ^= ((uint32_t)p[2] + dz) * kPrimeZ;
                row &= lt.mask;
            }
            const int64_t grow = base + (int64_t)row;
            gather_row<T, F>(table, grow, live && (uint64_t)grow < (uint64_t)lt.table_rows, v[q]);
        }
        float pv[NH][F];
#pragma unroll
        for (int q = 0; q < NH; ++q)
#pragma unroll
            for (int j = 0; j < F; ++j)
                pv[q][j] = __builtin_bit_cast(
                    float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[q][j]), 0xB1, 0xF, 0xF, true));
        if (dx == 0) {
            float acc[F];
#pragma unroll
            for (int k = 0; k < 2 * NH; ++k) {
                const int kx = k / NH, q = k % NH;
                const int dy = (DIM == 3) ? (q >> 1) : q;
                const int dz = (DIM == 3) ? (q & 1) : 0;
                float w = (kx ? f[0] : g[0]) * (dy ? f[1] : g[1]);
                if constexpr (DIM == 3) w = w * (dz ? f[2] : g[2]);
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const float tv = kx ? pv[q][j] : v[q][j];
                    acc[j] = (k == 0) ? tv * w : fmaf(tv, w, acc[j]);
                }
            }
            store_row<T, F>(reinterpret_cast<T *>(my + (size_t)sl * pitch) + l * F, acc);
        }
    }
    const int64_t rows = (N - wave_s0 < 32) ? (N - wave_s0) : 32;
    if (rows <= 0) return;
    // fine levels: pieces of the wave's 32 samples, lanes consecutive along the samples
    const Piece *fine = reinterpret_cast<const Piece *>(staged);
    const int nf = L - lc;
    // (four loads in flight per lane, then their LDS writes: one load -> wait -> write per trip was a chain of nf / 2 round trips)
    for (int e0 = lane; e0 < 32 * nf; e0 += 256) {
        Piece pc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + 64 * k;
            const int rr = e & 31, lf = e >> 5;
            const bool ok = e < 32 * nf && rr < rows;
            pc[k] = fine[ok ? (int64_t)(lc + lf) * N + wave_s0 + rr : (int64_t)lc * N + wave_s0];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + 64 * k;
            const int rr = e & 31, lf = e >> 5;
            if (e < 32 * nf && rr < rows) reinterpret_cast<Piece *>(my + (size_t)rr * pitch)[lc + lf] = pc[k];
        }
    }
    if (dx == 0 && sl < rows) *reinterpret_cast<uint32_t *>(my + (size_t)sl * pitch + pitch - 16u) = __float_as_uint(c4.w);
    // wave-private staging: no workgroup barrier (the wave's own LDS writes are visible to it after the wait below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (row_bytes % 16u == 0) {
        const uint32_t cpr = row_bytes / 16u;
        for (uint32_t e = lane; e < (uint32_t)rows * cpr; e += 64) {
            const uint32_t rr = e / cpr, q = e - rr * cpr;
            const unsigned char *src = my + (size_t)rr * pitch;
            const uint32_t pr = *reinterpret_cast<const uint32_t *>(src + pitch - 16u);
            *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(feats) + (size_t)pr * row_bytes + q * 16u) =
                *reinterpret_cast<const uint4 *>(src + q * 16u);
        }
    } else {
        for (uint32_t e = lane; e < (uint32_t)rows * (uint32_t)L; e += 64) {
            const uint32_t rr = e / (uint32_t)L, l = e - rr * (uint32_t)L;
            const unsigned char *src = my + (size_t)rr * pitch;
            const uint32_t pr = *reinterpret_cast<const uint32_t *>(src + pitch - 16u);
            reinterpret_cast<Piece *>(feats)[(size_t)pr * L + l] = reinterpret_cast<const Piece *>(src)[l];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Level-per-XCD pair kernel (variant 6, and the fine levels of the cell-sorted forward): one level per XCD at a time, so
// the XCD's 4 MiB L2 holds exactly that level's table and every table is pulled from the Infinity Cache by one XCD only
// (measured: fabric reads 50 M -> 8 M), + variant 3's lane pairing (corners x / x+1 in one instruction -> one L2
// request). The per-level feature piece of a sample (F scalars) goes to a level-major staging image [L][N][F] -- coalesced,
// where an 8-byte store into [N, L*F] allocates a whole 128-byte line per piece (0.2 ms on S1) -- with a PLAIN store since
// round 4: the rows kernel right behind reads the image back from the Infinity Cache (non-temporal: S1 forward +7 %).
// PACKED: `coords` is an array of 16-byte records {x, y, z (0 in 2-D), bits} (the cell-sorted copy of hashgrid_tiled.hip):
// one dwordx4 load per sample instead of DIM dword loads -- the kernel is bound by vector-memory instructions.
// (Round 3 pruned the unstaged forms of this kernel and the per-sample / per-level variants 1, 2, 4, 5, 7: git 4a7dfa7.)
#ifndef SHACIRA_LEVEL_PAIR_U
#define SHACIRA_LEVEL_PAIR_U 1
#endif
constexpr int kLevelPairU = SHACIRA_LEVEL_PAIR_U;
#ifndef SHACIRA_FWD_DIRECT_MAX
#define SHACIRA_FWD_DIRECT_MAX 16384          // batches up to this many samples write the output rows straight from the level kernel
#endif
template <int DIM, typename T, int F, bool PACKED = false>
__global__ __launch_bounds__(256) void hashgrid_fwd_level_pair_kernel(LevelTable lt,
                                                                      const int32_t *__restrict__ first_idx,
                                                                      const float *__restrict__ coords,
                                                                      const T *__restrict__ table,
                                                                      T *__restrict__ feats, int64_t N,
                                                                      uint32_t tiles, int64_t level_stride,
                                                                      int64_t sample_stride) {
    // destination of (level, sample): feats + level * level_stride + sample * sample_stride (elements). Staging [L][N][F]:
    // (N F, F); the caller's [N][L F] rows directly: (F, L F) -- small batches, see launch_fwd
    constexpr int NH = 1 << (DIM - 1);
    constexpr int U = kLevelPairU;   // samples per lane pair
    // work list = (level, tile) pairs, level-major, of levels [level_begin, level_end); XCD k (round-robin dispatch,
    // blockIdx % 8 -- a speed assumption only) takes the k-th eighth of it and walks it in order, so that at any moment its
    // L2 holds the table of one level (two at a slice boundary), whatever the number of levels
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t slot = blockIdx.x >> 3;
    const uint32_t per = gridDim.x >> 3;
    const uint32_t work = (uint32_t)(lt.level_end - lt.level_begin) * tiles;
    const uint32_t wi = xcd * per + slot;
    if (wi >= work) return;
    const uint32_t lvl = (uint32_t)lt.level_begin + wi / tiles;
    const uint32_t tile = wi % tiles;
    const int dx = threadIdx.x & 1;
    const int32_t res = lt.res[lvl];
    const float hi = lt.hi[lvl];
    const bool dense = lt.dense[lvl] != 0;
    const uint32_t r = (uint32_t)res;
    const int64_t base = (int64_t)first_idx[lvl];
    float v[U][NH][F];
    float f[U][DIM], g[U][DIM];
    bool live[U];
    int64_t idx[U];
    // issue all U*NH gathers before consuming any (memory-level parallelism)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = ((int64_t)tile * U + u) * 128 + (threadIdx.x >> 1);
        idx[u] = i;
        live[u] = i < N;
        double t[DIM];
        if constexpr (PACKED) {
            const float4 c4 = reinterpret_cast<const float4 *>(coords)[live[u] ? i : N - 1];   // clamped, unconditional
            t[0] = axis_unit(c4.x);
            t[1] = axis_unit(c4.y);
            if constexpr (DIM == 3) t[2] = axis_unit(c4.z);
        } else {
            load_unit_coords<DIM>(coords, i, N, t);
        }
        int32_t p[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) axis_transform(t[a], res, hi, p[a], f[u][a], g[u][a]);
        const uint32_t ux = (uint32_t)p[0] + (uint32_t)dx;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int dy = (DIM == 3) ? (q >> 1) : q;
            const int dz = (DIM == 3) ? (q & 1) : 0;
            const uint32_t uy = (uint32_t)p[1] + dy;
            uint32_t row;
            if (dense) {
                row = ux + uy * r;
                if constexpr (DIM == 3) row += ((uint32_t)p[2] + dz) * r * r;
            } else {
                row = ux ^ (uy * kPrimeY);
                if constexpr (DIM == 3) row ^= ((uint32_t)p[2] + dz) * kPrimeZ;
                row &= lt.mask;
            }
            const int64_t grow = base + (int64_t)row;
            gather_row<T, F>(table, grow, live[u] && (uint64_t)grow < (uint64_t)lt.table_rows, v[u][q]);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float pv[NH][F];
#pragma unroll
        for (int q = 0; q < NH; ++q)
#pragma unroll
            for (int j = 0; j < F; ++j)
                pv[q][j] = __builtin_bit_cast(
                    float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[u][q][j]), 0xB1, 0xF, 0xF, true));
        if (dx == 0 && live[u]) {
            float acc[F];
#pragma unroll
            for (int k = 0; k < 2 * NH; ++k) {
                const int kx = k / NH, q = k % NH;
                const int dy = (DIM == 3) ? (q >> 1) : q;
                const int dz = (DIM == 3) ? (q & 1) : 0;
                float w = (kx ? f[u][0] : g[u][0]) * (dy ? f[u][1] : g[u][1]);
                if constexpr (DIM == 3) w = w * (dz ? f[u][2] : g[u][2]);
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const float tv = kx ? pv[q][j] : v[u][q][j];
                    acc[j] = (k == 0) ? tv * w : fmaf(tv, w, acc[j]);
                }
            }
            T *dstT = feats + (int64_t)lvl * level_stride + idx[u] * sample_stride;
            if constexpr (sizeof(T) == 4 && F == 2) {   // staging stream: written once, read once (soon)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 o = {acc[0], acc[1]};
                // PLAIN 8-byte stores (round 4; they were non-temporal): the rows kernel reads the staged features back right
                // behind this kernel, from L2 / the Infinity Cache when the stores were allowed to stay there (S1 forward -7 %)
                *reinterpret_cast<f32x2 *>(dstT) = o;
            } else {
                store_row<T, F>(dstT, acc);
            }
        }
    }
}

// [L][N][F] -> [N][L*F] through LDS: both sides coalesced. Block: 256 samples; every access is one 8-byte
// (fp32, F == 2) or F*sizeof(T)-byte piece per lane, lanes consecutive.
template <typename T, int F>
__global__ __launch_bounds__(256) void untranspose_feats_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                                int64_t N, int L) {
    struct alignas(sizeof(T) * F) Piece { T v[F]; };
    extern __shared__ __align__(16) unsigned char s_raw_t[];
    Piece *s_tile = reinterpret_cast<Piece *>(s_raw_t);  // [256][L + 1] pieces
    const int pitch = L + 1;
    const int64_t s0 = (int64_t)blockIdx.x * 256;
    const int ns = (int)((N - s0 < 256) ? (N - s0) : 256);
    const Piece *in = reinterpret_cast<const Piece *>(src);
    for (int l = 0; l < L; ++l)
        if ((int)threadIdx.x < ns) s_tile[threadIdx.x * pitch + l] = in[(int64_t)l * N + s0 + threadIdx.x];
    __syncthreads();
    const int total = ns * L;
    Piece *out = reinterpret_cast<Piece *>(dst) + s0 * L;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int sm = e / L, l = e - sm * L;
        out[e] = s_tile[sm * pitch + l];
    }
}

template <typename T> static size_t staged_bytes(const LevelTable &lt, int64_t n) {
    return ((size_t)n * lt.num_lods * lt.feature_dim * sizeof(T) + 255) / 256 * 256;
}

template <int DIM, typename T, int F>
static hipError_t launch_fwd(const LevelTable &lt, const int32_t *first_idx, const float *coords, const void *table,
                             void *feats, void *workspace, int64_t num_coords, hipStream_t stream) {
    const int variant = opt().fwd_variant;
    if constexpr (F > 0) {
        // (its 16-byte output pieces need output rows of whole 16-byte units and an aligned base)
        if (use_lds_tables(DIM, (int)sizeof(T), lt, num_coords) && ((size_t)lt.num_lods * F * sizeof(T)) % 16 == 0 &&
            (reinterpret_cast<uintptr_t>(feats) & 15u) == 0) {
            LdsFwdPlan plan;
            make_lds_fwd_plan(DIM, lt, sizeof(T), plan);
            static PerDeviceOnce once;  // per instantiation and device: 128 KiB of dynamic LDS
            hipError_t e = once.run([]() -> hipError_t {
                return hipFuncSetAttribute(reinterpret_cast<const void *>(&hashgrid_fwd_lds_kernel<DIM, T, F>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsFwdBytes);
            });
            if (e != hipSuccess) return e;
            // one resident workgroup per CU (the image fills its LDS): 256 in total, never more than the items ask for
            uint32_t bpg = 256u / (uint32_t)plan.ngroups;
            const uint64_t spw = (uint64_t)kLdsFwdThreads / (uint64_t)plan.P;
            const uint64_t need = ((uint64_t)num_coords + spw - 1) / spw;
            if (bpg > need) bpg = (uint32_t)need;
            if (bpg < 1) bpg = 1;
            hipLaunchKernelGGL((hashgrid_fwd_lds_kernel<DIM, T, F>), dim3(bpg, (uint32_t)plan.ngroups), dim3(kLdsFwdThreads),
                               kLdsFwdBytes, stream, lt, plan, first_idx, coords, static_cast<const T *>(table),
                               static_cast<T *>(feats), num_coords);
            return hipGetLastError();
        }
        if (use_staged(DIM, lt, num_coords) && workspace) {
            // variant 6: level-per-XCD schedule with lane pairing, features staged level-major (coalesced stores),
            // then one transposing copy into the caller's [N, L*F] layout
            const uint32_t tiles = (uint32_t)((num_coords + 128 * kLevelPairU - 1) / (128 * kLevelPairU));
            const uint32_t grid_v6 = 8u * (uint32_t)(((uint64_t)lt.num_lods * tiles + 7) / 8);
            hipError_t e;
            // The smallest batches write the caller's rows themselves: F-element pieces, 8 XCDs x 2 levels into each 128-byte row of
            // S1's shape -- partial-line stores that the memory side merges -- and the staging round trip with its second launch
            // goes away. Measured (tools/small_ab.py, option fwd_direct, config D's table, ray points): 8 192 samples (the E
            // shard) 17.3 -> 14.0 us, 16 384: 17.9 -> 16.2; beyond that the partial lines cost more than the launch saves
            // (32 768: 23.3 -> 25.3, 65 536: 36.2 -> 44.4, 2^17: 61 -> 82)
            const int d_opt = opt().fwd_direct;
            if (d_opt == 1 || (d_opt < 0 && num_coords <= SHACIRA_FWD_DIRECT_MAX)) {
                hipLaunchKernelGGL((hashgrid_fwd_level_pair_kernel<DIM, T, F>), dim3(grid_v6), dim3(256), 0, stream, lt,
                                   first_idx, coords, static_cast<const T *>(table), static_cast<T *>(feats), num_coords, tiles,
                                   (int64_t)F, (int64_t)lt.num_lods * F);
                return hipGetLastError();
            }
            hipLaunchKernelGGL((hashgrid_fwd_level_pair_kernel<DIM, T, F>), dim3(grid_v6),
                               dim3(256), 0, stream, lt, first_idx, coords, static_cast<const T *>(table),
                               static_cast<T *>(workspace), num_coords, tiles, num_coords * F, (int64_t)F);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            const size_t shmem = (size_t)256 * (lt.num_lods + 1) * F * sizeof(T);
            static PerDeviceOnce once;  // per instantiation and device: allow > 64 KiB of dynamic LDS (L = 32, F = 4, fp32)
            e = once.run([]() -> hipError_t {
                return hipFuncSetAttribute(reinterpret_cast<const void *>(&untranspose_feats_kernel<T, F>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
            });
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((untranspose_feats_kernel<T, F>), dim3((uint32_t)((num_coords + 255) / 256)), dim3(256),
                               shmem, stream, static_cast<const T *>(workspace), static_cast<T *>(feats), num_coords,
                               lt.num_lods);
            return hipGetLastError();
        }
        if (variant == 3 || variant < 0) {
            const uint32_t blocks = (uint32_t)((num_coords + 127) / 128);
            const int LFP = (lt.num_lods * F + 3) / 4 * 4 + 4;
            const size_t shmem = (size_t)128 * LFP * sizeof(float);
            if (shmem <= 64 * 1024) {
                hipLaunchKernelGGL((hashgrid_fwd_pair_kernel<DIM, T, F>), dim3(blocks), dim3(256), shmem, stream, lt,
                                   first_idx, coords, static_cast<const T *>(table), static_cast<T *>(feats),
                                   num_coords);
                return hipGetLastError();
            }
        }
    }
    // items are indexed with 32 bits inside a launch; chunk the samples so that samples*L < 2^31
    const int64_t L = lt.num_lods;
    const int64_t max_samples = ((int64_t)1 << 31) / L - 1;
    for (int64_t s0 = 0; s0 < num_coords; s0 += max_samples) {
        const int64_t ns = (num_coords - s0 < max_samples) ? (num_coords - s0) : max_samples;
        const uint32_t items = (uint32_t)(ns * L);
        const uint32_t blocks = (items + 255u) / 256u;
        hipLaunchKernelGGL((hashgrid_fwd_kernel<DIM, T, F>), dim3(blocks), dim3(256), 0, stream, lt, first_idx, coords,
                           static_cast<const T *>(table), static_cast<T *>(feats), s0, items);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int DIM, typename T>
static hipError_t dispatch_f(const LevelTable &lt, const int32_t *first_idx, const float *coords, const void *table,
                             void *feats, void *ws, int64_t n, hipStream_t s) {
    switch (lt.feature_dim) {
        case 2: return launch_fwd<DIM, T, 2>(lt, first_idx, coords, table, feats, ws, n, s);
        case 4: return launch_fwd<DIM, T, 4>(lt, first_idx, coords, table, feats, ws, n, s);
        default: return launch_fwd<DIM, T, 0>(lt, first_idx, coords, table, feats, ws, n, s);
    }
}

// variant 9: every level fits an LDS image (and the levels make at most 8 groups). Measured on the Kodak tables: 0.0194 vs
// 0.0335 ms at 393 216 samples, 0.47 vs 0.85 ms at 9.4 M, still ahead at 4 096 (13.7 vs 16.3 us, both launch-bound)
static bool use_lds_tables(int dim, int esz, const LevelTable &lt, int64_t n) {
    const int v = opt().fwd_variant;
    if (lt.feature_dim != 2 && lt.feature_dim != 4) return false;
    if (v >= 0 && v != 9) return false;
    if (n < 1 || n >= ((int64_t)1 << 31)) return false;
    LdsFwdPlan p;
    if (!make_lds_fwd_plan(dim, lt, (size_t)esz, p)) return false;
    return v == 9 || n >= 4096;
}

// level-major staging buffer [L][N][F] of the table's scalar type (variant 6)
static bool use_staged(int dim, const LevelTable &lt, int64_t n) {
    const int v = opt().fwd_variant;
    if (lt.feature_dim != 2 && lt.feature_dim != 4) return false;
    if (v == 6) return true;
    // measured: 3-D batches from 8 192 samples (config E's per-GPU shard: 16.7 vs 19.0 us; 16 384: 19.9 vs 22.5);
    // 2-D and smaller batches: variant 3
    return v < 0 && dim == 3 && n >= 8192;
}

template <int DIM, typename T, int F>
static hipError_t launch_levels_staged(const LevelTable &lt, const int32_t *first_idx, const float *coords,
                                       const void *table, void *staged, int64_t n, hipStream_t s) {
    const uint32_t nl = (uint32_t)(lt.level_end - lt.level_begin);
    const uint32_t tiles = (uint32_t)((n + 128 * kLevelPairU - 1) / (128 * kLevelPairU));
    hipLaunchKernelGGL((hashgrid_fwd_level_pair_kernel<DIM, T, F, true>),
                       dim3(8u * (uint32_t)(((uint64_t)nl * tiles + 7) / 8)), dim3(256), 0, s,
                       lt, first_idx, coords, static_cast<const T *>(table), static_cast<T *>(staged), n, tiles, n * F,
                       (int64_t)F);
    return hipGetLastError();
}

hipError_t hashgrid_forward_levels_staged(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx,
                                          const float *coords, const void *table, void *staged, int64_t n,
                                          hipStream_t s) {
    const int F = lt.feature_dim;
    if (F != 2 && F != 4) return hipErrorInvalidValue;
    if (dim == 3 && dtype == SHACIRA_F32)
        return F == 2 ? launch_levels_staged<3, float, 2>(lt, first_idx, coords, table, staged, n, s)
                      : launch_levels_staged<3, float, 4>(lt, first_idx, coords, table, staged, n, s);
    if (dim == 3)
        return F == 2 ? launch_levels_staged<3, __half, 2>(lt, first_idx, coords, table, staged, n, s)
                      : launch_levels_staged<3, __half, 4>(lt, first_idx, coords, table, staged, n, s);
    if (dtype == SHACIRA_F32)
        return F == 2 ? launch_levels_staged<2, float, 2>(lt, first_idx, coords, table, staged, n, s)
                      : launch_levels_staged<2, float, 4>(lt, first_idx, coords, table, staged, n, s);
    return F == 2 ? launch_levels_staged<2, __half, 2>(lt, first_idx, coords, table, staged, n, s)
                  : launch_levels_staged<2, __half, 4>(lt, first_idx, coords, table, staged, n, s);
}

template <int DIM, typename T, int F>
static hipError_t launch_rows(const LevelTable &lt, const int32_t *first_idx, const float *sorted4,
                              const void *table, const void *staged, void *feats, int64_t n, int lc, hipStream_t s) {
    const uint32_t row_bytes = (uint32_t)(lt.num_lods * F * sizeof(T));
    const size_t shmem = (size_t)128 * ((row_bytes + 15u) / 16u * 16u + 16u);
    static PerDeviceOnce once;
    const hipError_t oe = once.run([]() -> hipError_t {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&hashgrid_fwd_rows_kernel<DIM, T, F>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    });
    if (oe != hipSuccess) return oe;
    const uint32_t blocks = ((uint32_t)((n + 127) / 128) + 7u) & ~7u;   // a multiple of 8 (surplus workgroups find no samples)
    hipLaunchKernelGGL((hashgrid_fwd_rows_kernel<DIM, T, F>), dim3(blocks), dim3(256), shmem, s, lt,
                       first_idx, reinterpret_cast<const float4 *>(sorted4), static_cast<const T *>(table),
                       static_cast<const T *>(staged), static_cast<T *>(feats), n, lc, 1u);
    return hipGetLastError();
}

// coarse levels [0, lc) over cell-sorted coordinates + assembly of whole rows through perm (hashgrid_tiled.hip)
hipError_t hashgrid_forward_rows(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *sorted4,
                                 const void *table, const void *staged, void *feats, int64_t n, int lc, hipStream_t s) {
    const int F = lt.feature_dim;
    if (F != 2 && F != 4) return hipErrorInvalidValue;
    if (dim == 3 && dtype == SHACIRA_F32)
        return F == 2 ? launch_rows<3, float, 2>(lt, first_idx, sorted4, table, staged, feats, n, lc, s)
                      : launch_rows<3, float, 4>(lt, first_idx, sorted4, table, staged, feats, n, lc, s);
    if (dim == 3)
        return F == 2 ? launch_rows<3, __half, 2>(lt, first_idx, sorted4, table, staged, feats, n, lc, s)
                      : launch_rows<3, __half, 4>(lt, first_idx, sorted4, table, staged, feats, n, lc, s);
    if (dtype == SHACIRA_F32)
        return F == 2 ? launch_rows<2, float, 2>(lt, first_idx, sorted4, table, staged, feats, n, lc, s)
                      : launch_rows<2, float, 4>(lt, first_idx, sorted4, table, staged, feats, n, lc, s);
    return F == 2 ? launch_rows<2, __half, 2>(lt, first_idx, sorted4, table, staged, feats, n, lc, s)
                  : launch_rows<2, __half, 4>(lt, first_idx, sorted4, table, staged, feats, n, lc, s);
}

// Test hook: the level-local corner rows and interpolation weights exactly as the kernels compute them
// (compute_corners of hashgrid_device.h = hash_index / hash_index2d and the weight products of the reference,
// hashgrid_interpolate_cuda.cu:17-39, :68-94), one (sample, level) per thread.
template <int DIM>
__global__ __launch_bounds__(256) void debug_corners_kernel(LevelTable lt, const float *__restrict__ coords, int64_t N,
                                                            int32_t *__restrict__ idx, float *__restrict__ w) {
    constexpr int NC = 1 << DIM;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int L = lt.num_lods;
    if (e >= N * L) return;
    const int64_t i = e / L;
    const int l = (int)(e - i * L);
    double t[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) t[a] = axis_unit(coords[i * DIM + a]);
    Corners<DIM> c;
    compute_corners<DIM>(t, lt.res[l], lt.hi[l], lt.dense[l] != 0, lt.mask, c);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if (idx) idx[e * NC + k] = (int32_t)c.row[k];
        if (w) w[e * NC + k] = c.w[k];
    }
}

hipError_t hashgrid_debug_corners(int dim, const LevelTable &lt, const float *coords, int64_t n, int32_t *idx, float *w,
                                  hipStream_t s) {
    const int64_t items = n * lt.num_lods;
    if (items <= 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((items + 255) / 256);
    if (dim == 3)
        hipLaunchKernelGGL(debug_corners_kernel<3>, dim3(blocks), dim3(256), 0, s, lt, coords, n, idx, w);
    else
        hipLaunchKernelGGL(debug_corners_kernel<2>, dim3(blocks), dim3(256), 0, s, lt, coords, n, idx, w);
    return hipGetLastError();
}

// fp64 tables: the reference-shaped kernel (one gather per corner, runtime feature_dim) with double loads / stores
template <int DIM>
static hipError_t launch_fwd_f64(const LevelTable &lt, const int32_t *first_idx, const float *coords, const void *table,
                                 void *feats, int64_t num_coords, hipStream_t stream) {
    const int64_t L = lt.num_lods;
    const int64_t max_samples = ((int64_t)1 << 31) / L - 1;
    for (int64_t s0 = 0; s0 < num_coords; s0 += max_samples) {
        const int64_t ns = (num_coords - s0 < max_samples) ? (num_coords - s0) : max_samples;
        const uint32_t items = (uint32_t)(ns * L);
        hipLaunchKernelGGL((hashgrid_fwd_kernel<DIM, double, 0>), dim3((items + 255u) / 256u), dim3(256), 0, stream, lt,
                           first_idx, coords, static_cast<const double *>(table), static_cast<double *>(feats), s0, items);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t hashgrid_forward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n) {
    if (dtype == SHACIRA_F64) return 0;
    if (tiled_supported(dim, dtype, lt, n)) return tiled_forward_workspace(dim, dtype, lt, n);
    if (lt.feature_dim != 2 && lt.feature_dim != 4) return 0;
    if (use_lds_tables(dim, dtype == SHACIRA_F32 ? 4 : 2, lt, n)) return 0;   // variant 9 stages nothing
    size_t b = ((size_t)n * lt.num_lods * lt.feature_dim * (dtype == SHACIRA_F32 ? 4 : 2) + 255) / 256 * 256;
    return b;
}

hipError_t hashgrid_forward_dispatch(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx,
                                     const float *coords, const void *table, void *feats, void *ws, int64_t n,
                                     hipStream_t s, void *plan, bool plan_ready) {
    if (lt.table_rows == 0) {
        // an empty table has no corner inside it: every feature is zero (and gather_row, which reads row 0 on behalf of lanes
        // that have no row, must not run; the dense 16-byte form that reads rows 0-1 is only reached by tables >= 8 MB)
        const size_t bytes = (size_t)n * lt.num_lods * lt.feature_dim * (dtype == SHACIRA_F64 ? 8 : dtype == SHACIRA_F32 ? 4 : 2);
        return zero_fill_async(static_cast<float *>(feats), (int64_t)(bytes / 4), s);   // (F is even: a multiple of 4 bytes)
    }
    if (dtype == SHACIRA_F64)
        return dim == 3 ? launch_fwd_f64<3>(lt, first_idx, coords, table, feats, n, s)
                        : launch_fwd_f64<2>(lt, first_idx, coords, table, feats, n, s);
    if (tiled_supported(dim, dtype, lt, n)) return tiled_forward(dim, dtype, lt, first_idx, coords, table, feats, ws, n, s, plan, plan_ready);
    if (dim == 3) {
        return dtype == SHACIRA_F32 ? dispatch_f<3, float>(lt, first_idx, coords, table, feats, ws, n, s)
                                    : dispatch_f<3, __half>(lt, first_idx, coords, table, feats, ws, n, s);
    }
    return dtype == SHACIRA_F32 ? dispatch_f<2, float>(lt, first_idx, coords, table, feats, ws, n, s)
                                : dispatch_f<2, __half>(lt, first_idx, coords, table, feats, ws, n, s);
}

}  // namespace shacira
