// internal.h -- declarations shared between the translation units of libshacira_hip.so (not part of the ABI).
#pragma once

#include <atomic>
#include <mutex>

#include "hashgrid_device.h"

namespace shacira {

// hipFuncSetAttribute (the opt-in for > 64 KiB of dynamic LDS) is a per-DEVICE setting: the opt-ins run once for every
// device this process launches on (the current device must be the one the caller's stream belongs to; the PyTorch
// wrappers enter `torch.cuda.device(tensor.device)` around every call).
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
    std::once_flag flag[kMaxDevices];
    hipError_t err[kMaxDevices] = {};
    template <typename Fn> hipError_t run(Fn &&fn) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
        std::call_once(flag[dev], [&] { err[dev] = fn(); });
        return err[dev];
    }
};

// hashgrid_fwd.hip
size_t hashgrid_forward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n);
hipError_t hashgrid_forward_dispatch(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx,
                                     const float *coords, const void *table, void *feats, void *workspace, int64_t n,
                                     hipStream_t s, void *plan = nullptr, bool plan_ready = false);
// levels [lt.level_begin, lt.level_end) of the level-per-XCD pair kernel into a level-major staging buffer [L][N][F]
hipError_t hashgrid_forward_levels_staged(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx,
                                          const float *coords, const void *table, void *staged, int64_t n,
                                          hipStream_t s);
// coarse levels [0, lc) over cell-sorted coordinates + assembly of whole feature rows through perm
hipError_t hashgrid_forward_rows(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *sorted4,
                                 const void *table, const void *staged, void *feats, int64_t n, int lc, hipStream_t s);
// hashgrid_tiled.hip: cell-sorted ("tiled") forward for large batches
bool tiled_supported(int dim, int dtype, const LevelTable &lt, int64_t n);
size_t tiled_forward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n);
// `plan`: caller-owned plan buffer (sample_plan_bytes) the sort writes into, NULL = inside the workspace; `plan_ready`: it
// already holds this batch's plan (the sort is skipped)
hipError_t tiled_forward(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx, const float *coords,
                         const void *table, void *feats, void *workspace, int64_t n, hipStream_t s, void *plan = nullptr,
                         bool plan_ready = false);
// The PLAN of a coordinate batch: its samples counting-sorted by spatial block (16-byte records {x, y, z, sample index}) and
// the blocks' offsets -- what the cell-sorted forward computes first and the backward's brick pass walks again. A function
// of (dim, n, coords) only; (dim, n) fix its layout and block grid.
struct SortedBatch {
    const float4 *sorted4;        // [n]
    const uint32_t *block_start;  // [num_blocks + 1]
    uint32_t num_blocks;
    int32_t nb[3];                // blocks per axis; block id = qx + nb[0] * (qy + nb[1] * qz)
};
size_t sample_plan_bytes(int dim, int64_t n);
size_t sample_plan_scratch_bytes(int dim, int64_t n);
hipError_t sample_plan_build(int dim, const float *coords, int64_t n, void *plan, void *scratch, hipStream_t s);
void sample_plan_view(int dim, int64_t n, const void *plan, SortedBatch &out);
void sample_plan_grid(int dim, int64_t n, SortedBatch &out);   // block grid only (no buffer)
hipError_t hashgrid_debug_corners(int dim, const LevelTable &lt, const float *coords, int64_t n, int32_t *idx, float *w,
                                  hipStream_t s);
// hashgrid_bwd.hip
hipError_t zero_fill_async(float *p, int64_t n, hipStream_t s);   // zero fill as a kernel (graph-capture safe)
size_t hashgrid_backward_workspace(int dim, int dtype, const LevelTable &lt, int64_t n);
// ... of a call that brings the batch's plan (whole level range; 16-byte aligned grad_output or not)
size_t hashgrid_backward_workspace_planned(int dim, int dtype, const LevelTable &lt, int64_t n, bool grad_aligned);
hipError_t hashgrid_backward_dispatch(int dim, int dtype, const LevelTable &lt, const int32_t *first_idx,
                                      const float *coords, const void *grad_out, void *grad_table, void *workspace,
                                      size_t workspace_bytes, int64_t n, hipStream_t s, const void *plan = nullptr);

// latent.hip
struct DecodeArgs {
    const float *latent, *div, *matrix, *colscale, *shift;
    float clampw;
    float *decoded;
    const float *grad_decoded;
    float *grad_latent, *grad_matrix, *grad_colscale, *grad_shift;
    double *partials;
    int64_t rows;
    const float *uniforms;   // non-NULL: stochastic Gumbel annealing instead of rounding ([rows, ld, 2] uniforms)
    float temperature;
    int diff_sampling;
    const float *temperature_dev;   // non-NULL: SGA temperature read on the device (single / per-level decoders)
};
struct EntropyArgs {
    const float *latent, *noise, *params, *grad_total;
    int num_layers;
    float *total_bits, *grad_latent, *grad_params;
    double *partials;
    int64_t rows;
};
// latent_mlp.hip
struct LatentMlpArgs {
    int num_layers;
    const int32_t *widths;   // host
    const float *latent, *uniforms, *div, *params;
    float temperature;
    int diff_sampling, act, final_act;
    float clampw;
    float *decoded;
    const float *grad_decoded;
    float *grad_latent, *grad_params;
    double *partials;
    int64_t rows;
};
bool latent_mlp_supported(int num_layers, const int32_t *widths);
size_t latent_mlp_workspace_bytes(int num_layers, const int32_t *widths);
hipError_t latent_mlp_dispatch(bool bwd, const LatentMlpArgs &a, hipStream_t s);

size_t latent_workspace_bytes();
bool latent_decode_supported(int ld, int f);
hipError_t latent_decode_dispatch(bool bwd, int ld, int f, const DecodeArgs &a, hipStream_t s);
hipError_t latent_decode_levels_dispatch(bool bwd, int ld, int f, int levels, const int64_t *offsets,
                                         const DecodeArgs &a, hipStream_t s);
bool latent_multi_supported(int ld, int f, int K);
hipError_t latent_multi_dispatch(bool bwd, int ld, int f, bool dft, const float *latent, const float *alpha, int K,
                                 const float *uniforms, float temperature, int straight_through, int diff_sampling,
                                 const float *div, const float *scale, const float *dftm, const float *shift,
                                 float clampw, int64_t rows, float *decoded, const float *grad_decoded,
                                 float *grad_latent, float *grad_alpha, float *grad_scale, float *grad_shift,
                                 double *partials, hipStream_t s);
bool entropy_supported(int ld);
hipError_t entropy_dispatch(bool bwd, int ld, const EntropyArgs &a, hipStream_t s);

// symbols.hip
bool symbols_supported(int ld);
hipError_t symbol_range_launch(const float *latent, int64_t rows, int ld, int32_t *minmax, hipStream_t s);
hipError_t symbol_histogram_launch(const float *latent, int64_t rows, int ld, const int32_t *minmax, int nbins,
                                   uint64_t *counts, hipStream_t s);
size_t rc_encode_bound(int64_t n);
int rc_encode(const int32_t *sym, int64_t n, const uint32_t *freq, int nsym, uint8_t *out, size_t cap, size_t *len);
int rc_decode(const uint8_t *in, size_t len, const uint32_t *freq, int nsym, int64_t n, int32_t *sym);

// render.hip
hipError_t pack_integrate_launch(bool bwd, int64_t R, int C, const float *feats, const float *tau,
                                 const int64_t *pack_start, float *ray_feats, float *weights, const float *g_ray,
                                 const float *g_w, float *g_feats, float *g_tau, hipStream_t s);
hipError_t pack_sum_launch(bool broadcast, int64_t R, int C, const float *in, const int64_t *pack_start, float *out,
                           hipStream_t s);
hipError_t raymarch_ray_launch(bool emit, int64_t num_rays, int ns, const float *origins, const float *dirs,
                               float dist_min, float dist_max, const float *lin, const float *jitter,
                               const uint8_t *occ, int level, int32_t *counts, const int64_t *offsets, int64_t *ridx,
                               float *samples, float *depth, float *deltas, uint8_t *boundary, int64_t capacity,
                               hipStream_t s);
hipError_t raytrace_dense_launch(bool emit, int64_t num_rays, const float *origins, const float *dirs,
                                 const uint8_t *occ, int level, int32_t *counts, const int64_t *offsets, int32_t *ridx,
                                 int32_t *pidx, float *depth, hipStream_t s);

// probe.hip
hipError_t stream_probe_launch(int kind, const void *a, void *b, size_t bytes, uint32_t *sink, hipStream_t s);

// adam.hip
hipError_t adam_step_launch(float *p, float *g, float *m, float *v, int64_t n, float lr, float b1, float b2, float eps,
                            float wd, int step, const int32_t *step_dev, int zero_grad, hipStream_t s);

hipError_t adam_multi_launch(int count, float *const *p, float *const *g, float *const *m, float *const *v,
                             const int64_t *n, const float *lr, const float *wd, float b1, float b2, float eps,
                             int step, const int32_t *step_dev, int zero_grad, hipStream_t s);

// mlp.hip / mlp_mfma.hip
constexpr int kMlpMaxBlocks = 512;        // backward grid (= rows of the fp64 partial-gradient workspace), VALU and wide MFMA
constexpr int kMlpNarrowMaxBlocks = 512; // same for the narrow (16x16x4) MFMA decoders (measured: 256 / 1024 / 2048 slower)
bool mlp_supported(int in, int h, int nh, int out);
int mlp_num_params(int in, int h, int nh, int out);
size_t mlp_workspace_bytes(int in, int h, int nh, int out);
hipError_t mlp_dispatch(bool bwd, int in, int h, int nh, int out, int64_t N, const float *x, const float *params,
                        float *y, const float *gy, float *gx, float *gparams, double *partials, hipStream_t s);

// api.hip: tunables. shacira_set_option() stores process-wide atomics; every API entry point takes ONE snapshot of them
// into a thread-local Options and everything below reads that snapshot (opt()), so a call -- its workspace-size check
// included -- sees one consistent set of values even while another thread changes an option.
struct Options {
    int fwd_variant = -1;         // -1 measured rule, 0 reference-shaped kernel, 3 lane pairs, 6 level-per-XCD staged, 8 cell-sorted
    int bwd_variant = -1;         // -1 by batch size, 0 scattered atomics (the reference's design), 1 binned
    int mlp_variant = -1;         // -1 MFMA decoders wherever instantiated, 0 VALU kernels
    int bwd_compact = 1;          // dense 3-D levels: one two-slot item per sample, z-slab buckets with a halo plane
    int bwd_selective_zero = 1;   // zero only the rows the consume pass does not overwrite (0: the whole table)
    int bwd_persistent = 1;       // consume pass: persistent workgroups fetching units from a counter
    int bwd_fork = 1;             // table zeroing + direct levels on a side stream when the batch is large
    int bwd_item12 = -1;          // 12-byte item units for fp32 tables (3-D, F = 2, batches >= 2^17): 1 always, 0 never (the 16-byte
                                  // stream), -1 = with the brick pass (planned calls in sorted order: the one place they pay)
    int bwd_run_pad = 1;          // scatter pass: (tile, bucket) runs reserved in whole 64-byte pieces (SHACIRA_RUN_ALIGN; large batches:
                                  // 4 units of 16 bytes, 16 units of 12 bytes = 192 bytes; the 8-byte half-precision units are not padded)
    int bin_acc_kib = 0;          // LDS accumulator image per consumer workgroup: 64, 128, 0 = by batch size
    int bin_batch_mib = 1536;     // cap of the backward's item array per sub-batch
    int tiled = -1;               // cell-sorted forward: -1 by batch size, 0 never, 1 whenever the shape allows
    int tiled_lc_fwd = -1;        // its number of coarse levels (rows kernel), -1 = planner
    int bwd_brick = -1;           // brick pass of the backward (needs the batch's plan): -1 / 1 whenever the shape allows, 0 never
    int bwd_brick_lo = -1;        // explicit brick level range [lo, hi) instead of the planner's rule (-1 = planner)
    int bwd_brick_hi = -1;
    int bwd_brick_fork = 2;       // where the brick pass runs: 0 last on the caller's stream, 1 / 2 side stream from behind the front / scatter pass
    int bwd_brick_span = 0;       // blocks per brick unit along x (0 = planner)
    int bwd_ext_fork = 1;         // planned calls: the brick pass's fork event rides on the scatter launch (hipExtLaunchKernelGGL)
    int fwd_direct = -1;          // level-per-XCD forward writes the output rows itself (no staging): -1 = small batches, 0 / 1
};
const Options &opt();             // the calling thread's snapshot
void options_snapshot();          // taken at every extern "C" entry point that reads options
void options_resolve_item12(int value);   // a call's decision for the automatic setting (-1), into the calling thread's snapshot

}  // namespace shacira
