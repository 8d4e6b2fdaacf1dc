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
//                       run with one returning atomic on the bucket's cursor, write it with coalesced 16-byte stores. Runs are
//                       reserved in whole 64-byte pieces (round 5, BinPlan::pad: pad units = all-zero items): scattered runs
//                       stream at 5.3 TB/s when they start and end on 64-byte boundaries, at 2.5-3.6 TB/s when they do not
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

#include <cstring>

#include "bwd_bin_types.h"
#include "bwd_bin_front.h"
#include "bwd_bin_passes.h"
#include "bwd_brick.h"
#include <hip/hip_ext.h>

namespace shacira {

// ------------------------------------------------------------------------------------------------- host side
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// `one_image_compact`: a dense 3-D level that fits ONE image travels as compact items too (one bucket whose units flush
// atomically) instead of the direct pass in front of the scatter pass; decided per CALL from the total batch (below), so
// that every plan of a call classifies the levels alike
// `skip_mask`: levels (bit l) another pass accumulates (the brick pass): neither binned nor direct here
// `n_call`: samples of the whole CALL (0 = n_batch). The item format is chosen per call (run_bin, carve), so a sub-batch's plan
// must size its pad units and staging for THAT format, not for the one its own sample count would pick (round-5 advisor finding)
static void make_plan(int dim, int dtype, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib,
                      bool one_image_compact, uint32_t skip_mask = 0u, int64_t n_call = 0);
static inline bool half_items(int dtype, const LevelTable &lt);
static inline int item_format(int dim, int dtype, const LevelTable &lt, int64_t n);
static inline size_t item_unit_bytes(int fmt, const LevelTable &lt);
#ifndef SHACIRA_RUN_ALIGN
#define SHACIRA_RUN_ALIGN 64           // bytes a (tile, bucket) run is padded to: the memory system's write atom (measured:
                                       // runs on 64-byte boundaries stream at 4.7-5.0 TB/s, on 128-byte ones at 5.2-5.4, on
                                       // 16 / 32 / 48-byte ones at 2.5-3.5; 64 costs half the pad units of 128)
#endif
#ifndef SHACIRA_PAD_HALF
#define SHACIRA_PAD_HALF 0             // A/B builds: pad the 8-byte half-precision units' runs too
#endif
#ifndef SHACIRA_FX_MIN
#define SHACIRA_FX_MIN (1 << 17)       // fixed-point images from this batch size (below: fp64 images)
#endif
#ifndef SHACIRA_PERSIST_MIN
#define SHACIRA_PERSIST_MIN (1 << 16)  // persistent consume workgroups from this batch size (measured: -2.5 % at 65 536
                                       // samples, equal at 32 768, +2 % at 16 384: below, the hardware's own dispatch of the
                                       // ~1 400 tiny workgroups is as good)
#endif
static inline bool one_image_compact_rule(int64_t n_total) { return n_total >= ((int64_t)1 << 17); }

// can the table be partitioned with an LDS accumulator image of `acc_kib` KiB per consumer workgroup?
static bool bin_feasible(int dim, int dtype, const LevelTable &lt, int acc_kib) {
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
    make_plan(dim, dtype, lt, kTile, plan, acc_kib, false);
    if (plan.total_buckets + (uint32_t)lt.num_lods > (uint32_t)kMaxBuckets) return false;   // (+ one-image compact levels)
    for (int l = 0; l < lt.num_lods; ++l)
        if (plan.lv[l].nb > (uint32_t)kMaxLevelBuckets) return false;
    return true;
}

// Image size per call, from the TOTAL batch (one choice per call so that every plan of the call classifies the levels
// alike). Option "bin_acc_kib": 64 / 128 force it, 0 (default) = measured rule: 64 KiB images for small batches (below 2^17
// samples in 3-D, up to 2^20 in 2-D), 128 KiB (half as many buckets) beyond.
static int choose_acc_kib(int dim, int dtype, const LevelTable &lt, int64_t n) {
    const int kib = opt().bin_acc_kib;
    if (kib != 0) return kib;
    const int64_t pairs = (int64_t)1 << (dim - 1);
    // (3-D, re-measured after the consume pass learned to fetch one unit ahead: 128 KiB wins from 2^17 samples -- 0.124 vs
    // 0.130 ms at 2^17, 0.180 vs 0.196 at 2^18, 0.315 vs 0.333 at 2^19 -- the fixed-point consume kernel needs 66 VGPRs, so
    // two 64 KiB workgroups do not share a CU anyway; 65 536 samples, fp64 images: 64 KiB 0.0925 vs 0.0960. 2-D: 64 KiB up to
    // 2^20 samples, 0.125 vs 0.134 ms at 2^18)
    const bool large = dim == 3 ? n >= ((int64_t)1 << 17) : n * pairs > ((int64_t)1 << 21);
    if (large || !bin_feasible(dim, dtype, lt, 64)) return 128;
    // tables whose levels are all "direct" (config B: every level fits an LDS image) want the big image: fewer level
    // groups, hence fewer walks over the samples (measured 82 vs 124 us on the 393 216-pixel batch)
    BinPlan big;
    make_plan(dim, dtype, lt, kTile, big, 128, false);
    return big.nbl == 0 ? 128 : 64;
}

bool bin_supported(int dim, const LevelTable &lt);

// every level fits an LDS image (no level is binned): the backward is the direct-level kernel alone
bool bin_all_direct(int dim, const LevelTable &lt) {
    if (!bin_supported(dim, lt)) return false;
    BinPlan plan;
    make_plan(dim, SHACIRA_F32, lt, kTile, plan, 128, false);
    return plan.nbl == 0 && plan.ngroups > 0;
}

bool bin_supported(int dim, const LevelTable &lt) {
    const int kib = opt().bin_acc_kib;
    // (fp32 plan: its 2-D compact levels need a few more buckets than the half-precision stream's pair items)
    return bin_feasible(dim, SHACIRA_F32, lt, kib ? kib : 128);
}

static void make_plan_uncached(int dim, int dtype, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib,
                               bool one_image_compact, uint32_t skip_mask, int64_t n_call);

// A backward call plans several times (workspace query, carving, image-size rule, the run itself), and a plan costs a few
// microseconds of host time (the magic-division checks of the compact levels walk every line): small batches became
// HOST-bound (2-D bw-19 table at 2^17 samples: 0.123 ms per call against 0.094 ms of GPU time). Plans are pure functions of
// their arguments and two options, so each thread keeps its last few.
static void make_plan(int dim, int dtype, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib,
                      bool one_image_compact, uint32_t skip_mask, int64_t n_call) {
    if (n_call <= 0) n_call = n_batch;
    struct Key {
        int dim, dtype, acc_kib, oic, compact, run_pad, item12, fmt;
        uint32_t skip_mask;
        int64_t n_batch;
        LevelTable lt;
    };
    struct Entry {
        bool valid = false;
        Key key;
        BinPlan plan;
    };
    constexpr int kEntries = 8;
    thread_local Entry cache[kEntries];
    thread_local int next = 0;
    Key k;
    std::memset(&k, 0, sizeof(k));     // (padding bytes take part in the comparison)
    k.dim = dim; k.dtype = dtype; k.acc_kib = acc_kib; k.oic = one_image_compact ? 1 : 0; k.compact = opt().bwd_compact;
    k.run_pad = opt().bwd_run_pad;
    k.item12 = opt().bwd_item12;
    k.skip_mask = skip_mask;
    k.fmt = item_format(dim, dtype, lt, n_call);
    k.n_batch = n_batch;
    std::memcpy(&k.lt, &lt, sizeof(LevelTable));
    for (int e = 0; e < kEntries; ++e) {
        if (cache[e].valid && std::memcmp(&cache[e].key, &k, sizeof(Key)) == 0) {
            plan = cache[e].plan;
            return;
        }
    }
    make_plan_uncached(dim, dtype, lt, n_batch, plan, acc_kib, one_image_compact, skip_mask, n_call);
    Entry &slot = cache[next];
    next = (next + 1) % kEntries;
    slot.key = k;
    slot.plan = plan;
    slot.valid = true;
}

static void make_plan_uncached(int dim, int dtype, const LevelTable &lt, int64_t n_batch, BinPlan &plan, int acc_kib,
                               bool one_image_compact, uint32_t skip_mask, int64_t n_call) {
    if (one_image_compact) {   // tables whose levels are ALL direct stay that way (no transposing pass at all)
        make_plan_uncached(dim, dtype, lt, n_batch, plan, acc_kib, false, skip_mask, n_call);
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
        if (l < lt.level_begin || l >= lt.level_end || ((skip_mask >> l) & 1u)) {  // not part of this call / the brick pass's
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
            // 2-D (fp32 item stream): the same with slabs of LINES -- all 4 corners in one bucket, one 8 + 4 F byte item per
            // sample (fractions as 25-bit fixed point) instead of two pair items
            const bool cm = (F == 2 || F == 4) && opt().bwd_compact != 0 && (dim == 3 || !half_items(dtype, lt));
            const uint64_t plane_rows = (dim == 3) ? res * res : res;
            const uint64_t planes = cm ? BR / plane_rows : 0;
            // (exp2: a level that fits ONE image may also travel as compact items -- one bucket, its units flush atomically --
            // instead of the direct pass that walks the whole batch in front of the scatter pass; 3-D only: on the 2-D bw-19
            // table it turns six more levels into items, 0.314 -> 0.352 ms at 2^20 samples)
            if (planes >= 2 && res >= 3 && (bl.used > BR || (one_image_compact && dim == 3))) {
                const uint32_t slab = (uint32_t)planes - 1;
                const uint32_t nbz = ((uint32_t)res - 2u) / slab + 1u;      // base cells: z (2-D: y) in [0, res - 2]
                if (nbz <= (uint32_t)kMaxLevelBuckets) {
                    bl.compact = 1;
                    bl.slab = slab;
                    bl.rows_pb = (slab + 1u) * (uint32_t)plane_rows;
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
    for (uint32_t q = 0; q < (uint32_t)SHACIRA_MAX_LODS; ++q) {
        plan.bstart[q] = q < plan.nbl ? plan.lv[plan.blevel[q]].bucket0 : 0u;
        plan.bnb[q] = q < plan.nbl ? plan.lv[plan.blevel[q]].nb : 0u;
    }
    plan.total_buckets = nbk;
    plan.BR = BR;
    plan.num_tiles = (uint32_t)((n_batch + tile_samples(dim) - 1) / tile_samples(dim));
    plan.pairs = 1u << (dim - 1);
    // work-unit size: 1/48 of ONE level's items, so that an evenly loaded hashed bucket (1/64 of a level) is ONE unit
    // (plain-store flush) with 33 % slack, while over-full coarse buckets split into equal chunks that keep all CUs busy
    uint64_t chunk = (uint64_t)n_batch * plan.pairs / 48 + 1024;
    if (chunk < 8192) chunk = 8192;
    if (chunk > (1u << 22)) chunk = 1u << 22;
    // Line-aligned runs (round 5). A (tile, bucket) run used to start wherever the bucket's cursor stood: every 128-byte line of
    // the item array was then written in two pieces by different 16-lane groups of a store (and the lines at a run's ends by
    // different workgroups), and scattered runs written that way reach 3.2 TB/s at 1 KB per run against 5.3 TB/s for the same
    // runs on line boundaries -- where the run LENGTH stops mattering at all (tools/microbench3.hip, profiles/r05_experiments.md).
    // So runs are reserved in multiples of SHACIRA_RUN_ALIGN = 64 bytes (4 units of 16 bytes, or 16 units of 12 bytes = 192 bytes;
    // pad units = all-zero items, which no consumer adds): large batches only
    // (below 2^17 samples the item array lives in the caches, which merge the pieces), 16-byte units only (a 24-byte unit would
    // need 192-byte multiples and goes through staging windows; the half-precision streams of fp16 tables -- 8-byte units, and
    // the 16-byte units of F = 4 -- are written with plain stores, which the L2 merges, and lose with pads: S1 fp16 backward
    // 0.438 -> 0.461 ms, nerf_lego table fp16 0.349 -> 0.363), and only while two scatter workgroups still share a CU with the
    // pad slots staged.
    plan.pad = 1;
    {
        const int fmt_p = item_format(dim, dtype, lt, n_call);   // the CALL's format (see the declaration)
        const size_t unit = item_unit_bytes(fmt_p, lt);
        const uint32_t tile_units = (uint32_t)tile_samples(dim) * plan.pairs;
        uint32_t maxnb = 0;
        for (uint32_t q = 0; q < plan.nbl; ++q) maxnb = plan.lv[plan.blevel[q]].nb > maxnb ? plan.lv[plan.blevel[q]].nb : maxnb;
        bool windows = false;
#ifdef SHACIRA_SCATTER_SPLIT
        windows = SHACIRA_SCATTER_SPLIT > 1;
#endif
        if (opt().bwd_run_pad != 0 && !windows && plan.nbl > 0 && n_call >= SHACIRA_FX_MIN &&
            ((fmt_p != 1 && (unit == 16 || unit == 12)) || (SHACIRA_PAD_HALF && fmt_p == 1 && (unit == 8 || unit == 16)))) {
            const uint32_t P = unit == 12 ? (uint32_t)(SHACIRA_RUN_ALIGN / 4) : (uint32_t)(SHACIRA_RUN_ALIGN / unit);   // 12-byte units: lcm(12, 64) = 16 of them
            const size_t staged = (size_t)(tile_units + maxnb * (P - 1u)) * (unit + 1);
            if (staged <= (size_t)78 * 1024 || (size_t)tile_units * (unit + 1) > (size_t)78 * 1024) plan.pad = P;
        }
        plan.stage_cap = tile_units + maxnb * (plan.pad - 1u);
    }
    const uint32_t cmask = plan.pad > 2u ? plan.pad - 1u : 1u;
    plan.chunk = (uint32_t)chunk & ~cmask;   // even: a compact item (two 16-byte slots) never straddles two work units; a
                                             // multiple of the run pad: every unit starts on a 64-byte boundary
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
// item stream format of a call: 0 = fp32 payloads (8 + 4 F bytes per unit), 1 = half-precision stream (fp16 tables), 2 = 12-byte
// units (fp32 tables, 3-D, F = 2, batches that accumulate in fixed-point images; option "bwd_item12")
static inline int item_format(int dim, int dtype, const LevelTable &lt, int64_t n) {
    if (half_items(dtype, lt)) return 1;
    // (bwd_item12 = -1, the automatic setting, is resolved per call by bin_backward before anything is sized; a workspace query
    // sees -1 and sizes for the 16-byte stream, the larger of the two)
    if (dim == 3 && lt.feature_dim == 2 && dtype == SHACIRA_F32 && n >= SHACIRA_FX_MIN && opt().bwd_item12 == 1) return 2;
    return 0;
}
static inline size_t item_unit_bytes(int fmt, const LevelTable &lt) {
    if (fmt == 1) return lt.feature_dim == 2 ? 8 : 16;
    if (fmt == 2) return 12;
    return 8 + 4 * (size_t)lt.feature_dim;
}

static int64_t bin_batch_samples(int dim, int dtype, const LevelTable &lt, int64_t n) {
    const size_t item = item_unit_bytes(item_format(dim, dtype, lt, n), lt);
    BinPlan plan;
    make_plan(dim, dtype, lt, kTile, plan, choose_acc_kib(dim, dtype, lt, n), one_image_compact_rule(n));
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

// every level of the table is "direct" (fits an LDS image) whatever level range a call names: such tables never need the
// transposed gradients nor items (the image configs B / C / kodak.yaml: the direct kernel reads grad_output itself)
static bool table_all_direct(int dim, int dtype, const LevelTable &lt, int64_t n) {
    LevelTable full = lt;
    full.level_begin = 0;
    full.level_end = lt.num_lods;
    BinPlan plan;
    make_plan(dim, dtype, full, kTile, plan, choose_acc_kib(dim, dtype, full, n), one_image_compact_rule(n));
    return plan.nbl == 0;
}

// 16-byte (item-unit) slots one sample can occupy over all binned levels of the plan: x-pair items 2^(dim-1) per level,
// compact 3-D levels two slots, compact 2-D levels one
static uint64_t slots_per_sample(const BinPlan &plan) {
    uint64_t slots = 0;
    for (uint32_t q = 0; q < plan.nbl; ++q) {
        const BinLevel &bl = plan.lv[plan.blevel[q]];
        slots += bl.compact ? (plan.pairs == 4u ? 2u : 1u) : plan.pairs;
    }
    return slots;
}

static int front_tile(int L, int F, uint32_t nbl, int64_t n, size_t *shmem);

// The brick pass's levels for a call (bwd_brick.h), or none: with the batch's plan at hand (the cell-sorted forward's records;
// only its block grid `sb->nb` is read here), coarse 3-D levels are accumulated block by block in LDS and leave the item stream.
// Whole calls over one sub-batch that accumulate in fixed point (>= 2^17 samples: the regime the item stream binds in); never for
// tables whose levels are all direct. One function for the workspace query, the carving and the launch sequence.
// Measured rule (option bwd_brick = -1; tools/brick_cfg_ab.py, profiles/r06_experiments.md 3 + 7): fp32 item streams of
// F = 2 tables, at every batch size that has a plan (the forward sorts from 2^18 samples) -- with the 12-byte item units
// the pass brings along, S1 backward at 2^20 / 2^19 / 2^18 samples: 0.468 -> 0.417 / 0.270 -> 0.252 / 0.166 -> 0.161 ms.
// Not taken: half-precision item streams (items are half the bytes already: S1 fp16 0.415 -> 0.435) and F = 4
// (nerf_lego.yaml's table: 32 LDS atomics per sample and level, six of its eleven dense levels do not fit the images
// and would stay compact items in sorted order: 0.612 -> 0.617, fp16 0.504 -> 0.508). bwd_brick = 1 takes it wherever
// the shape allows.
static void brick_levels_of_call(int dim, int dtype, const LevelTable &lt, int64_t n, const SortedBatch *sb, bool zero_table,
                                 bool grad_aligned, BrickPlan &brick, uint32_t &skip_mask) {
    brick.nlev = 0;
    skip_mask = 0u;
    if (dim != 3 || sb == nullptr) return;
    const int L = lt.num_lods, F = lt.feature_dim;
    const bool multi = bin_batch_samples(dim, dtype, lt, n) < n;
    // (sorted mode rides on the fused 16-byte front kernel: rows of whole 16-byte vectors, 16-byte aligned input)
    const int kvec0 = (int)(16 / ((dtype == SHACIRA_F32 ? 4 : 2) * F));
    const int bopt = opt().bwd_brick;
    const bool wanted = bopt == 1 || (bopt < 0 && F == 2 && dtype == SHACIRA_F32);
    if (!wanted || multi || lt.stage_flags != 0 || !zero_table || n < SHACIRA_FX_MIN || n * L >= ((int64_t)1 << 31) ||
        table_all_direct(dim, dtype, lt, n) || (L % kvec0) != 0 || !grad_aligned)
        return;
    // beside the consume pass (mode 2) a brick workgroup must fit the LDS its 128 KiB image leaves
    if (!make_brick_plan(lt, n, *sb, opt().bwd_brick_lo, opt().bwd_brick_hi, opt().bwd_brick_span,
                         (size_t)(opt().bwd_brick_fork == 2 ? 30 : 64) * 1024, brick)) {
        brick.nlev = 0;
        return;
    }
    for (uint32_t q = 0; q < brick.nlev; ++q) skip_mask |= 1u << brick.lv[q].level;
    // (nothing left for the item passes, or no tile size for the fused front kernel: keep the plain pipeline)
    BinPlan rest;
    make_plan(dim, dtype, lt, n, rest, choose_acc_kib(dim, dtype, lt, n), one_image_compact_rule(n), skip_mask, n);
    size_t sh0 = 0;
    if (rest.nbl == 0 || front_tile(L, F, rest.nbl, n, &sh0) <= 0) {
        skip_mask = 0u;
        brick.nlev = 0;
    }
}

// skip_mask: the levels the call's brick pass takes (brick_levels_of_call): they emit no items
static BinWorkspace carve(int dim, int dtype, const LevelTable &lt, int64_t n, void *ws, uint32_t skip_mask = 0u) {
    BinPlan plan;
    const int64_t nb = bin_batch_samples(dim, dtype, lt, n);
    make_plan(dim, dtype, lt, nb, plan, choose_acc_kib(dim, dtype, lt, n), one_image_compact_rule(n), skip_mask, n);
    const size_t item = item_unit_bytes(item_format(dim, dtype, lt, n), lt);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // gT and gmax first: their offsets must not depend on the level range of the call (REUSE_STAGED calls share them).
    // Sized by what the selected path uses (round 4): an all-direct table stages nothing -- config C's query returned 1.21 GB
    // for a call that ran one kernel on grad_output -- and the item array holds the slots the plan's levels can emit
    // (compact levels: 2 of the 4 pair slots), S1: 1.23 -> 1.05 GB.
    const bool no_stage = table_all_direct(dim, dtype, lt, n);
    const size_t o_gT = take(no_stage ? 0 : (size_t)level_pitch(n) * lt.num_lods * lt.feature_dim * sizeof(float));
    const size_t o_ctrl = take((size_t)(kTotalShards * kMaxBuckets + SHACIRA_MAX_LODS) * sizeof(uint32_t));   // totals | gmax: one memset
    // (+ the pad units of line-aligned runs: at most pad - 1 per (scatter tile, bucket))
    const uint64_t pad_units = (uint64_t)plan.num_tiles * plan.total_buckets * (plan.pad - 1u);
    const size_t o_items = take(((size_t)nb * slots_per_sample(plan) + (size_t)pad_units) * item);
    const size_t o_base = take((size_t)(kMaxBuckets + 2) * sizeof(uint64_t));
    const size_t o_cur = take((size_t)(kMaxBuckets + 2) * sizeof(uint64_t));
    const size_t o_cnt = take((size_t)(nb / 128 + 8) * plan.total_buckets * sizeof(uint32_t));   // smallest counting tile: 128
    const size_t o_unit = take((size_t)(kMaxBuckets + 2) * sizeof(uint32_t));
    const uint64_t max_items_ws = (uint64_t)nb * plan.nbl * plan.pairs + pad_units;
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

// Item format and brick levels of one call (the thread's option snapshot of this C-ABI call is updated in place).
// 12-byte item units, automatic setting: with the brick pass, i.e. on planned calls of the shapes its rule takes. In sorted order
// with the dense levels out of the item stream the consume pass has the instruction slots the unpacking costs (S1 backward
// 0.440 -> 0.417 ms; on the plain path the format returns 1-3 %; profiles/r06_experiments.md 7). Decided before the workspace is
// sized or carved: every plan and pass of the call then sees one format.
static void resolve_call(int dim, int dtype, const LevelTable &lt, int64_t n, const SortedBatch *sb, bool zero_table,
                         bool grad_aligned, BrickPlan &brick, uint32_t &skip_mask) {
    const bool auto12 = opt().bwd_item12 < 0;
    if (auto12) {
        const bool brick_shape = sb != nullptr && dim == 3 && lt.feature_dim == 2 && dtype == SHACIRA_F32 &&
                                 lt.level_begin == 0 && lt.level_end == lt.num_lods && lt.stage_flags == 0 && zero_table &&
                                 opt().bwd_brick != 0;
        options_resolve_item12(brick_shape ? 1 : 0);
    }
    brick_levels_of_call(dim, dtype, lt, n, sb, zero_table, grad_aligned, brick, skip_mask);
    if (auto12 && brick.nlev == 0) options_resolve_item12(0);
}
// (fp16 tables keep the full layout: the offset of their fp32 accumulation image is asked for before the call is resolved)
static inline uint32_t carve_skip(int dtype, uint32_t skip_mask) { return dtype == SHACIRA_F32 ? skip_mask : 0u; }

size_t bin_workspace_bytes(int dim, int dtype, const LevelTable &lt, int64_t n) {
    return carve(dim, dtype, lt, n, nullptr).bytes;
}

// workspace of a PLANNED call (a plan of the batch exists; `whole`: level range = all levels, table zeroed by the call;
// `grad_aligned`: 16-byte aligned grad_output): the levels its brick pass takes emit no items, the rest 12-byte units. A call
// the brick rule does not take is sized like the plain one.
size_t bin_workspace_bytes_planned(int dim, int dtype, const LevelTable &lt, int64_t n, bool whole, bool grad_aligned) {
    SortedBatch sb{};
    sample_plan_grid(dim, n, sb);
    BrickPlan brick;
    uint32_t skip_mask = 0;
    resolve_call(dim, dtype, lt, n, &sb, whole, grad_aligned, brick, skip_mask);
    return carve(dim, dtype, lt, n, nullptr, carve_skip(dtype, skip_mask)).bytes;
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
    hipEvent_t bfork = nullptr, bjoin = nullptr;   // brick pass beside the item passes
};
static hipError_t side_stream(SideStream **out) {
    static thread_local SideStream per_device[kMaxDevices];
    int dev = 0;
    SHACIRA_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    SideStream &ss = per_device[dev];
    if (!ss.stream) {
        hipStream_t st;
        hipEvent_t a, b, c, d, e;
        SHACIRA_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        SHACIRA_CHECK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        SHACIRA_CHECK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        SHACIRA_CHECK(hipEventCreateWithFlags(&c, hipEventDisableTiming));
        SHACIRA_CHECK(hipEventCreateWithFlags(&d, hipEventDisableTiming));
        SHACIRA_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ss.fork = a;
        ss.join = b;
        ss.staged = c;
        ss.bfork = d;
        ss.bjoin = e;
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
                          bool zero_table, __half *half_table, bool *converted, const SortedBatch *sb) {
    const int L = lt.num_lods;
    const int64_t NP = level_pitch(n);
    BinPlan whole;
    const int acc_kib = choose_acc_kib(DIM, dtype, lt, n);
    const bool oic = one_image_compact_rule(n);
    const int64_t nb = bin_batch_samples(DIM, dtype, lt, n);
    const bool multi = nb < n;
    const bool stage_all = (lt.stage_flags & SHACIRA_BWD_STAGE_ALL_LEVELS) != 0;
    const bool staged = (lt.stage_flags & SHACIRA_BWD_REUSE_STAGED) != 0;
    // Brick pass (bwd_brick.h): with the batch's plan at hand (the cell-sorted forward's records), coarse 3-D levels are
    // accumulated block by block in LDS and leave the item stream. Whole calls over one sub-batch that accumulate in
    // fixed point (>= 2^17 samples: the regime the item stream binds in); never for tables whose levels are all direct.
    BrickPlan brick;
    uint32_t skip_mask = 0;
    brick_levels_of_call(DIM, dtype, lt, n, sb, zero_table, (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0, brick, skip_mask);
    make_plan(DIM, dtype, lt, n, whole, acc_kib, oic, skip_mask, n);
    // SORTED mode: the whole call walks the batch in the plan's block order -- the front kernel gathers the gradient rows in
    // that order, so gT and every later pass's sample index mean "k-th sorted sample" and coordinates come from the records
    const bool sorted = brick.nlev > 0;
    const float *cptr = sorted ? reinterpret_cast<const float *>(sb->sorted4) : coords;
    const int cstride = sorted ? 4 : DIM;
    // Where the brick pass runs (it is bound by LDS atomics and touches HBM hardly at all; it needs the zeroed table, the
    // sorted gT and gmax, i.e. the front pass): option bwd_brick_fork 0 = last on the caller's stream (behind the scatter pass,
    // which wants the transposed gradients still in the Infinity Cache); 1 / 2 = on the side stream, forked behind the front pass /
    // behind the scatter pass, joined at the end -- beside the consume pass, whose workgroups (128 KiB image, 1 024 threads) leave
    // room for brick workgroups on every CU and wait on HBM while the LDS atomic units idle
    int brick_mode = -1;   // -1: no brick pass
    SideStream *bss = nullptr;
    if (brick.nlev > 0) {
        brick_mode = opt().bwd_brick_fork;
        if (brick_mode > 0) SHACIRA_CHECK(side_stream(&bss));
    }
    // `done`: an event the launch signals on completion (hipExtLaunchKernelGGL's stop event), or nullptr
    auto launch_brick = [&](hipStream_t bs, hipEvent_t done = nullptr) -> hipError_t {
        if constexpr (DIM == 3) {
            const size_t img = (size_t)brick.rows_total * F * sizeof(double);
            const int hr = fx_headroom((uint64_t)kBrickUnit * 8u);
            // one workgroup per group (its first unit) + one per kBrickUnit-record window of the batch (what over-full groups
            // hold beyond; all of them leave at once on a uniform batch)
            const uint32_t groups = brick.groups_x * (uint32_t)(brick.nb[1] * brick.nb[2]);
            const uint32_t windows = (uint32_t)((n + kBrickUnit - 1) / kBrickUnit);
            if (done != nullptr)
                hipExtLaunchKernelGGL((brick_accumulate_kernel<F>), dim3(groups + windows), dim3(kBrickThreads), (uint32_t)img, bs,
                                      nullptr, done, 0u, lt, brick, first_idx, sb->sorted4, sb->block_start, (const float *)w.gT,
                                      NP, acc, (const uint32_t *)w.gmax, hr);
            else
                hipLaunchKernelGGL((brick_accumulate_kernel<F>), dim3(groups + windows), dim3(kBrickThreads), img, bs, lt, brick,
                                   first_idx, sb->sorted4, sb->block_start, w.gT, NP, acc, w.gmax, hr);
            return hipGetLastError();
        } else {
            (void)bs;
            return hipSuccess;
        }
    };
    auto fork_brick = [&]() -> hipError_t {
        SHACIRA_CHECK(hipEventRecord(bss->bfork, s));
        SHACIRA_CHECK(hipStreamWaitEvent(bss->stream, bss->bfork, 0));
        SHACIRA_CHECK(launch_brick(bss->stream));
        return hipEventRecord(bss->bjoin, bss->stream);
    };
    // only binned levels consume the transposed gradients (a later call on this workspace may, too: stage_all) -- a table
    // whose levels are all direct has none in any level range, so its calls never stage (and carve() reserves no gT)
    const bool need_T = whole.nbl > 0 || ((stage_all || staged) && !table_all_direct(DIM, dtype, lt, n));
    // fixed-point images pay off once the accumulation itself dominates; small batches are bound by fixed costs and
    // keep the fp64 image (and skip the gmax bookkeeping): measured 100 vs 107 us at 65 536 samples
    const bool use_fx = need_T && n >= SHACIRA_FX_MIN;
    // the 16-byte front kernel needs rows of whole 16-byte vectors and a 16-byte aligned input
    const size_t esz = dtype == SHACIRA_F32 ? 4 : 2;
    const int kvec = (int)(16 / (esz * F));
    size_t front_shmem = 0;
    const int ts16 = front_tile(L, F, whole.nbl, n, &front_shmem);
    const bool t16 = ts16 > 0 && (L % kvec) == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0;
    // counting fused into the front kernel: a call that transposes, single sub-batch
    const bool front_counts = need_T && !staged && !multi && t16 && whole.nbl > 0;
    // fp16 tables, one sub-batch over all levels: single-unit hashed buckets flush straight into the caller's half table and
    // the conversion of the fp32 accumulation image skips them (f32_to_f16_skip_kernel) -- most of the table never makes the
    // round trip through the image (round 4)
    const bool direct_half = half_table != nullptr && half_items(dtype, lt) && !multi && !stage_all && !staged &&
                             whole.nbl > 0 && lt.level_begin == 0 && lt.level_end == L;
    // side stream (option bwd_fork): table zeroing + direct levels beside the critical path. Worth an event pair once the
    // batch is large (threshold in units of n * L * F: heavier tables fork earlier)
    const int64_t fork_work = n * lt.num_lods * lt.feature_dim;
    const int64_t fork_min = DIM == 3 ? ((int64_t)7 << 21) : ((int64_t)1 << 23);
    // (an event pair costs ~10-20 us of cross-stream latency here: only worth it with direct levels to hide)
    const bool fork = whole.nbl > 0 && whole.ngroups > 0 && !multi && opt().bwd_fork != 0 && fork_work >= fork_min &&
                      brick_mode <= 0;   // (one side stream: the brick pass has it when it runs there)
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
    // bucket totals (and, when this call transposes, the per-level max |grad_output| right behind them) start at zero
    const bool need_words = whole.nbl > 0 || (!staged && use_fx);
    const uint32_t words = (uint32_t)kTotalShards * kMaxBuckets + ((!staged && use_fx) ? SHACIRA_MAX_LODS : 0);
    bool words_done = false;
    if (zero_table) {   // at::zeros_like of the reference
        if (!selective) {
            SHACIRA_CHECK(zero_fill_async(acc, (int64_t)lt.table_rows * lt.feature_dim, zs));
        } else {
            // (same stream: the control words ride along as one extra slice of this launch)
            words_done = need_words && zs == s;
            hipLaunchKernelGGL(zero_unowned_rows_kernel, dim3(256, (uint32_t)L + (words_done ? 1u : 0u)), dim3(256), 0, zs, acc,
                               first_idx, lt, whole, w.totals, words);
            SHACIRA_CHECK_LAUNCH();
        }
    }
    if (need_words && !words_done) {
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
            int rounds = (int)((tiles + 511) / 512);
            // padded runs: a workgroup's counting tiles must be whole scatter tiles (the pad units are per scatter tile)
            const int cps16 = TileOf<DIM>::value / ts16;
            if (whole.pad > 1u && front_counts) rounds = (rounds + cps16 - 1) / cps16 * cps16;
            const uint32_t blocks = (uint32_t)((tiles + rounds - 1) / rounds);
#define SHACIRA_FRONT(TT, GM, CN)                                                                                         \
            hipLaunchKernelGGL((front16_kernel<DIM, TT, F, GM, CN>), dim3(blocks), dim3(kFrontThreads), front_shmem, s, lt, \
                               whole, static_cast<const TT *>(grad_out), w.gT, coords, w.totals, w.cnt, n, NP, t_lb, t_le, ts_log2, \
                               rounds, GM ? w.gmax : nullptr, (uint32_t)cps16)
#define SHACIRA_FRONT_SORTED(TT)                                                                                          \
            hipLaunchKernelGGL((front16_kernel<DIM, TT, F, true, true, true>), dim3(blocks), dim3(kFrontThreads), front_shmem, s, lt, \
                               whole, static_cast<const TT *>(grad_out), w.gT, coords, w.totals, w.cnt, n, NP, t_lb, t_le, ts_log2, \
                               rounds, w.gmax, (uint32_t)cps16, sb->sorted4)
            if (sorted) {   // (implies use_fx and front_counts)
                if (dtype == SHACIRA_F32) SHACIRA_FRONT_SORTED(float);
                else SHACIRA_FRONT_SORTED(__half);
            } else if (dtype == SHACIRA_F32) {
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
#undef SHACIRA_FRONT_SORTED
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
    if (brick_mode == 1) SHACIRA_CHECK(fork_brick());
    // When nothing is transposed (every level is direct: the image configs) there is no gmax, and a pass of its own over
    // grad_output costs more than the faster atomics return (tried: a streaming abs-max kernel, 0.103 vs 0.082 ms on config
    // B): the direct kernel's workgroups take a pilot maximum over their own samples instead (direct_accumulate_kernel).
    // direct levels: one pass over the whole batch, no items (they add into the zeroed table)
    if (whole.ngroups > 0) {
        const BinPlan &plan = whole;
        // one level group (S1's level 0): ~512 workgroups measured best; several groups (the all-direct image tables,
        // 128 KiB images = one resident workgroup per CU): 256 in total = one wave of workgroups, no tail
        // (config B backward 65 vs 77 us, tools/attic/direct_blocks.py)
        uint32_t bpg = (plan.ngroups > 1 ? 256u : 512u) / plan.ngroups;
        const uint32_t need = (uint32_t)((n + 2047) / 2048);      // at least ~2 samples per thread each
        if (bpg > need) bpg = need;
        if (bpg < 1) bpg = 1;
        const size_t acc_bytes = (size_t)plan.BR * F * sizeof(double);
        const dim3 grid(bpg, plan.ngroups);
        // a row receives at most (samples walked by one workgroup) x (corners) contributions. Fixed-point images always:
        // scaled by the transposing pass's max |gradient| per level when there is one (gmax), else by the workgroup's own
        // pilot maximum (all-direct tables, small batches: direct_accumulate_kernel; config C backward 0.84 -> 0.63 ms,
        // B 0.056 -> 0.047, kodak.yaml-shaped 0.104 -> 0.082 against the fp64 images)
        const int headroom = fx_headroom(((uint64_t)n / bpg + kConsumeThreads) * (1u << DIM));
        const uint32_t *gm = (need_T && use_fx) ? w.gmax : nullptr;
        if (need_T)
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, float, true, true>), grid, dim3(kConsumeThreads),
                               acc_bytes, zs, lt, plan, first_idx, cptr, w.gT, acc, n, NP, gm, headroom, cstride);
        else if (dtype == SHACIRA_F32)
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, float, false, true>), grid, dim3(kConsumeThreads),
                               acc_bytes, zs, lt, plan, first_idx, coords, static_cast<const float *>(grad_out), acc, n,
                               NP, gm, headroom);
        else
            hipLaunchKernelGGL((direct_accumulate_kernel<DIM, F, __half, false, true>), grid, dim3(kConsumeThreads),
                               acc_bytes, zs, lt, plan, first_idx, coords, static_cast<const __half *>(grad_out), acc, n,
                               NP, gm, headroom);
        SHACIRA_CHECK_LAUNCH();
    }
    if (fork) SHACIRA_CHECK(hipEventRecord(ss->join, ss->stream));
    if (whole.nbl == 0) return hipSuccess;   // (fork implies binned levels)
    constexpr int NPAIR = 1 << (DIM - 1);
    // half-precision item stream: fp16 tables with F = 2 (8-byte units)
    const int fmt = item_format(DIM, dtype, lt, n);
    const bool half = fmt == 1;
    // (one staging window: the tile's items + the pad units of its runs, plan.stage_cap; windows: exact runs, 1 / split of the tile)
    auto stage_bytes = [&](const BinPlan &pl) -> size_t {
        const size_t isz = fmt == 2 ? sizeof(Item12) : half ? sizeof(typename ItemSel<F, true>::type) : sizeof(Item<F>);
        const int split = half ? ScatterSplit<typename ItemSel<F, true>::type>::value : ScatterSplit<Item<F>>::value;
        const size_t units = split == 1 ? (size_t)pl.stage_cap : (size_t)TileOf<DIM>::value * NPAIR / split;
        return units * (isz + 1);
    };
#ifdef SCATTER_LEVEL_FAST
#define SCATTER_GRID(pl) dim3((pl).nbl, (pl).num_tiles)
#else
#define SCATTER_GRID(pl) dim3((pl).num_tiles, (pl).nbl)
#endif
    bool first_batch = true;
    for (int64_t s0 = 0; s0 < n; s0 += nb) {
        const int64_t hi = (s0 + nb < n) ? (s0 + nb) : n;
        BinPlan plan;
        make_plan(DIM, dtype, lt, hi - s0, plan, acc_kib, oic, skip_mask, n);
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
                           plan.total_buckets, plan, w.work_counter, w.cursor, use_fx ? w.gmax : nullptr);
        SHACIRA_CHECK_LAUNCH();
        // (selective zeroing: hashed buckets with 0 or several units are zeroed by the scatter pass's tail)
        float *zacc = selective ? acc : nullptr;
        const uint32_t cps = fused_now ? (uint32_t)(TileOf<DIM>::value / ts16) : 1u;
        const uint32_t cnt_rows = fused_now ? (uint32_t)((n + ts16 - 1) / ts16) : plan.num_tiles;
        bool fork_signalled = false;   // the scatter launch itself signals the brick pass's fork event (below)
        if (fmt == 2) {
            if constexpr (DIM == 3 && F == 2) {
                // Brick pass beside the consume pass: the event the side stream waits for rides on the scatter kernel's own
                // completion signal (hipExtLaunchKernelGGL's stop event) instead of a marker of its own behind it -- that marker
                // kept the consume pass from being staged behind the scatter pass (7.6 us between the two under the profiler).
                // Not while the stream is being captured into a graph (plain launch + event record there).
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                if (brick_mode == 2 && opt().bwd_ext_fork != 0 && hipStreamIsCapturing(s, &cap) == hipSuccess &&
                    cap == hipStreamCaptureStatusNone) {
                    hipExtLaunchKernelGGL((bin_scatter_kernel<DIM, F, 2>), SCATTER_GRID(plan),
                                          dim3(ScatterThreads<DIM, F, 2>::value), (uint32_t)stage_bytes(plan), s, nullptr,
                                          bss->bfork, 0u, lt, plan, cptr, (const float *)w.gT, w.cursor, (const uint32_t *)w.cnt, cps, cnt_rows,
                                          reinterpret_cast<Item12 *>(w.items), s0, hi, NP, zacc, first_idx,
                                          (const uint32_t *)w.unit_first, cstride);
                    fork_signalled = true;
                } else {
                    hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, 2>), SCATTER_GRID(plan), dim3(ScatterThreads<DIM, F, 2>::value),
                                       stage_bytes(plan), s, lt, plan, cptr, w.gT, w.cursor, w.cnt, cps, cnt_rows,
                                       reinterpret_cast<Item12 *>(w.items), s0, hi, NP, zacc, first_idx, w.unit_first, cstride);
                }
            }
        } else if (half)
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, true>), SCATTER_GRID(plan), dim3(ScatterThreads<DIM, F, 1>::value),
                               stage_bytes(plan), s, lt, plan, cptr, w.gT, w.cursor, w.cnt, cps, cnt_rows,
                               reinterpret_cast<typename ItemSel<F, true>::type *>(w.items), s0, hi, NP, zacc, first_idx,
                               w.unit_first, cstride);
        else if (use_fx)
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, false>), SCATTER_GRID(plan), dim3(ScatterThreads<DIM, F, 0>::value),
                               stage_bytes(plan), s, lt, plan, cptr, w.gT, w.cursor, w.cnt, cps, cnt_rows,
                               reinterpret_cast<Item<F> *>(w.items), s0, hi, NP, zacc, first_idx, w.unit_first, cstride);
        else   // batches below 2^17 samples: plain item stores, the consume pass reads them back from the caches
            hipLaunchKernelGGL((bin_scatter_kernel<DIM, F, false, false>), SCATTER_GRID(plan), dim3(ScatterThreads<DIM, F, 0>::value),
                               stage_bytes(plan), s, lt, plan, cptr, w.gT, w.cursor, w.cnt, cps, cnt_rows,
                               reinterpret_cast<Item<F> *>(w.items), s0, hi, NP, zacc, first_idx, w.unit_first, cstride);
        SHACIRA_CHECK_LAUNCH();
        if (fork) SHACIRA_CHECK(hipStreamWaitEvent(s, ss->join, 0));   // table zeroed, direct levels in
        if (brick_mode == 2) {
            if (fork_signalled) {
                SHACIRA_CHECK(hipStreamWaitEvent(bss->stream, bss->bfork, 0));
                SHACIRA_CHECK(launch_brick(bss->stream, bss->bjoin));   // (and the join event on the brick launch)
            } else {
                SHACIRA_CHECK(fork_brick());
            }
        }
        const uint64_t max_items = (uint64_t)(hi - s0) * plan.nbl * NPAIR +
                                   (uint64_t)plan.num_tiles * plan.total_buckets * (plan.pad - 1u);
        uint32_t grid_units = (uint32_t)(max_items / plan.chunk_min) + plan.total_buckets + 1;
        const size_t acc_bytes = (size_t)plan.BR * F * sizeof(double);
        // persistent: as many workgroups as the chip holds fetch units from the work counter (measured: S1 backward
        // 0.611 -> 0.595 ms, 2-D 0.375 -> 0.369; with the fetch pipelined one unit ahead also at 65 536 samples)
        uint32_t *wc = (opt().bwd_persistent != 0 && n >= SHACIRA_PERSIST_MIN) ? w.work_counter : nullptr;
        if (wc != nullptr && grid_units > 512u) grid_units = 512u;
        const int headroom = use_fx ? fx_headroom((uint64_t)plan.chunk + 1) : -1;   // a unit streams <= chunk items
        const int fa = multi ? 1 : 0;
        __half *hout = direct_half ? half_table : nullptr;
        if (fmt == 2) {
            if constexpr (DIM == 3 && F == 2)
                hipLaunchKernelGGL((bin_consume_kernel<F, true, 2>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s, lt,
                                   plan, first_idx, w.unit_first, w.unit_desc, reinterpret_cast<const Item12 *>(w.items), acc,
                                   fa, headroom, wc);
        } else if (half && use_fx)
            hipLaunchKernelGGL((bin_consume_kernel<F, true, true>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s,
                               lt, plan, first_idx, w.unit_first, w.unit_desc,
                               reinterpret_cast<const typename ItemSel<F, true>::type *>(w.items), acc, fa, headroom, wc, hout);
        else if (half)
            hipLaunchKernelGGL((bin_consume_kernel<F, false, true>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s,
                               lt, plan, first_idx, w.unit_first, w.unit_desc,
                               reinterpret_cast<const typename ItemSel<F, true>::type *>(w.items), acc, fa, headroom, wc, hout);
        else if (use_fx)
            hipLaunchKernelGGL((bin_consume_kernel<F, true, false>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s, lt,
                               plan, first_idx, w.unit_first, w.unit_desc,
                               reinterpret_cast<const Item<F> *>(w.items), acc, fa, headroom, wc);
        else if (!half)
            hipLaunchKernelGGL((bin_consume_kernel<F, false, false>), dim3(grid_units), dim3(kConsumeThreads), acc_bytes, s, lt,
                               plan, first_idx, w.unit_first, w.unit_desc,
                               reinterpret_cast<const Item<F> *>(w.items), acc, fa, headroom, wc);
        SHACIRA_CHECK_LAUNCH();
    }
    if (brick_mode == 0) SHACIRA_CHECK(launch_brick(s));   // LAST on the caller's stream
    if (brick_mode > 0) SHACIRA_CHECK(hipStreamWaitEvent(s, bss->bjoin, 0));
    if (direct_half) {
        hipLaunchKernelGGL(f32_to_f16_skip_kernel, dim3(128, (uint32_t)L), dim3(256), 0, s, acc, half_table, first_idx, lt,
                           whole, w.unit_first, F);
        SHACIRA_CHECK_LAUNCH();
        if (converted != nullptr) *converted = true;
    }
    return hipSuccess;
}

// largest dynamic LDS a scatter launch can ask for: one staging window of the tile's item units (+ the pad units of
// line-aligned runs when the tile goes through in one window) and one bucket byte per unit
template <class ItemT> static constexpr size_t stage_max(size_t tile_units) {
    constexpr size_t split = ScatterSplit<ItemT>::value;
    constexpr size_t padu = split != 1 ? 0 : sizeof(ItemT) == 16 ? SHACIRA_RUN_ALIGN / 16 - 1 : sizeof(ItemT) == 12 ? SHACIRA_RUN_ALIGN / 4 - 1
                            : (SHACIRA_PAD_HALF && sizeof(ItemT) == 8) ? SHACIRA_RUN_ALIGN / 8 - 1 : 0;
    return (tile_units / split + (size_t)kMaxLevelBuckets * padu) * (sizeof(ItemT) + 1);
}

hipError_t bin_backward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                        const void *grad_out, float *acc, void *workspace, int64_t n, hipStream_t s, bool zero_table,
                        void *half_table, bool *converted, const SortedBatch *sb) {
    if (converted != nullptr) *converted = false;
    __half *ht = (dtype == SHACIRA_F16) ? static_cast<__half *>(half_table) : nullptr;
    BrickPlan brick;
    uint32_t skip_mask = 0;
    resolve_call(dim, dtype, lt, n, sb, zero_table, (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0, brick, skip_mask);
    const BinWorkspace w = carve(dim, dtype, lt, n, workspace, carve_skip(dtype, skip_mask));
    static PerDeviceOnce once;  // kernels that use more than 64 KiB of dynamic LDS must opt in once per device
    const hipError_t attr_err = once.run([]() -> hipError_t {
        hipError_t attr_err = hipSuccess;
        // dynamic LDS actually requested (static LDS of the kernels comes on top and must fit in 160 KiB too)
        auto set = [&attr_err](const void *fn, size_t bytes) {
            if (bytes <= 64 * 1024 || bytes > 160 * 1024) return;   // (beyond the CU's LDS: such a launch fails by itself)
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
        set(reinterpret_cast<const void *>(&front16_kernel<3, TT, FF, false, false>), 156 * 1024);      \
        set(reinterpret_cast<const void *>(&front16_kernel<3, TT, FF, true, true, true>), 156 * 1024);
        SHACIRA_T_ATTR(float, 2) SHACIRA_T_ATTR(float, 4) SHACIRA_T_ATTR(__half, 2) SHACIRA_T_ATTR(__half, 4)
#undef SHACIRA_T_ATTR
#define SHACIRA_DIRECT_ATTR(D, FF)                                                                              \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, true, true>), 16384 * sizeof(double));   \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, float, false, true>), 16384 * sizeof(double));  \
        set(reinterpret_cast<const void *>(&direct_accumulate_kernel<D, FF, __half, false, true>), 16384 * sizeof(double));
        SHACIRA_DIRECT_ATTR(2, 2) SHACIRA_DIRECT_ATTR(2, 4) SHACIRA_DIRECT_ATTR(3, 2) SHACIRA_DIRECT_ATTR(3, 4)
#undef SHACIRA_DIRECT_ATTR
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, true, 2>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 2, 2>), stage_max<Item12>(TileOf<3>::value << 2));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, true, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, true, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, false, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, false, false>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, true, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<2, false, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, true, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_consume_kernel<4, false, true>), 16384 * sizeof(double));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 2, true>), stage_max<ItemH>(TileOf<2>::value << 1));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 2, true>), stage_max<ItemH>(TileOf<3>::value << 2));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 4, true>), stage_max<typename ItemSel<4, true>::type>(TileOf<2>::value << (2 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 4, true>), stage_max<typename ItemSel<4, true>::type>(TileOf<3>::value << (3 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 2, false>), stage_max<Item<2>>(TileOf<2>::value << (2 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 4, false>), stage_max<Item<4>>(TileOf<2>::value << (2 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 2, false>), stage_max<Item<2>>(TileOf<3>::value << (3 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 4, false>), stage_max<Item<4>>(TileOf<3>::value << (3 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 2, false, false>), stage_max<Item<2>>(TileOf<2>::value << (2 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<2, 4, false, false>), stage_max<Item<4>>(TileOf<2>::value << (2 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 2, false, false>), stage_max<Item<2>>(TileOf<3>::value << (3 - 1)));
        set(reinterpret_cast<const void *>(&bin_scatter_kernel<3, 4, false, false>), stage_max<Item<4>>(TileOf<3>::value << (3 - 1)));
        return attr_err;
    });
    if (attr_err != hipSuccess) return attr_err;
    if (dim == 3) {
        return lt.feature_dim == 2 ? run_bin<3, 2>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table, ht, converted, sb)
                                   : run_bin<3, 4>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table, ht, converted, sb);
    }
    return lt.feature_dim == 2 ? run_bin<2, 2>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table, ht, converted, nullptr)
                               : run_bin<2, 4>(dtype, lt, first_idx, coords, grad_out, acc, w, n, s, zero_table, ht, converted, nullptr);
}

}  // namespace shacira

#ifdef CONSUME_TRACE
// instrumented build only: copies the trace out and clears it (tools/attic/consume_trace.py)
extern "C" __attribute__((visibility("default"))) int shacira_debug_consume_trace(unsigned long long *host, unsigned int cap,
                                                                                   unsigned int *count) {
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(shacira::g_consume_trace_n), sizeof(n)) != hipSuccess) return 1;
    *count = n;
    if (n > cap) n = cap;
    if (n > 16384u) n = 16384u;
    if (n && hipMemcpyFromSymbol(host, HIP_SYMBOL(shacira::g_consume_trace), (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess)
        return 2;
    unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(shacira::g_consume_trace_n), &zero, sizeof(zero)) != hipSuccess) return 3;
    return 0;
}
#endif
