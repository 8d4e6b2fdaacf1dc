// api.hip -- the extern "C" entry points declared in include/shacira_hip.h: argument validation, level-table
// construction and dispatch. No allocation, no synchronisation, no global mutable state besides the tunables (atomics,
// snapshotted once per call).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "internal.h"

namespace shacira {

// ---- tunables: process-wide atomics, one snapshot per API call (internal.h) ------------------------------------------
struct OptionSlot {
    const char *name;
    int Options::*field;
    int lo, hi;           // accepted range (inclusive)
    std::atomic<int> value;
};
static OptionSlot g_slots[] = {
    {"fwd_variant", &Options::fwd_variant, -1, 9, {-1}},
    {"bwd_variant", &Options::bwd_variant, -1, 1, {-1}},
    {"mlp_variant", &Options::mlp_variant, -1, 1, {-1}},
    {"bwd_compact", &Options::bwd_compact, 0, 1, {1}},
    {"bwd_selective_zero", &Options::bwd_selective_zero, 0, 1, {1}},
    {"bwd_persistent", &Options::bwd_persistent, 0, 1, {1}},
    {"bwd_fork", &Options::bwd_fork, 0, 1, {1}},
    {"bwd_run_pad", &Options::bwd_run_pad, 0, 1, {1}},
    {"bwd_item12", &Options::bwd_item12, -1, 1, {-1}},
    {"bin_acc_kib", &Options::bin_acc_kib, 0, 128, {0}},
    {"bin_batch_mib", &Options::bin_batch_mib, 1, 1 << 20, {1536}},
    {"tiled", &Options::tiled, -1, 1, {-1}},
    {"tiled_lc_fwd", &Options::tiled_lc_fwd, -1, SHACIRA_MAX_LODS, {-1}},
    {"bwd_brick", &Options::bwd_brick, -1, 1, {-1}},
    {"bwd_brick_lo", &Options::bwd_brick_lo, -1, SHACIRA_MAX_LODS, {-1}},
    {"bwd_brick_hi", &Options::bwd_brick_hi, -1, SHACIRA_MAX_LODS, {-1}},
    {"bwd_brick_fork", &Options::bwd_brick_fork, 0, 2, {2}},
    {"bwd_brick_span", &Options::bwd_brick_span, 0, 64, {0}},
    {"fwd_direct", &Options::fwd_direct, -1, 1, {-1}},
    {"bwd_ext_fork", &Options::bwd_ext_fork, 0, 1, {1}},
};
static thread_local Options tl_options;
const Options &opt() { return tl_options; }
void options_snapshot() {
    for (OptionSlot &sl : g_slots) tl_options.*(sl.field) = sl.value.load(std::memory_order_relaxed);
}
void options_resolve_item12(int value) { tl_options.bwd_item12 = value; }
static bool option_value_ok(const OptionSlot &sl, int v) {
    if (v < sl.lo || v > sl.hi) return false;
    if (!std::strcmp(sl.name, "fwd_variant")) return v == -1 || v == 0 || v == 3 || v == 6 || v == 8 || v == 9;
    if (!std::strcmp(sl.name, "bin_acc_kib")) return v == 0 || v == 64 || v == 128;
    return true;
}

static int build_level_table(int dim, int num_lods, int feature_dim, int bw, const int32_t *res_host,
                             int64_t table_rows, LevelTable &lt) {
    if (dim != 2 && dim != 3) return SHACIRA_EINVAL;
    if (num_lods < 1 || num_lods > SHACIRA_MAX_LODS) return SHACIRA_EINVAL;
    if (feature_dim < 1) return SHACIRA_EINVAL;
    if (feature_dim % 2 == 1) return SHACIRA_EODD;
    if (bw < 1 || bw > 30) return SHACIRA_EINVAL;  // int32_t codebook_size = pow(2, bw), .cpp:56
    if (!res_host || table_rows < 0) return SHACIRA_EINVAL;
    std::memset(&lt, 0, sizeof(lt));
    const int32_t cs = (int32_t)std::pow(2.0, (double)bw);
    for (int l = 0; l < num_lods; ++l) {
        const int32_t r = res_host[l];
        if (r < 1) return SHACIRA_EINVAL;
        lt.res[l] = r;
        lt.hi[l] = (float)((double)r - 1.0 - 1e-5);  // `resolution-1-1e-5` narrowed at the clamp() call
        lt.dense[l] = level_is_dense(dim, r, cs) ? 1 : 0;
    }
    lt.mask = (uint32_t)cs - 1u;
    lt.num_lods = num_lods;
    lt.feature_dim = feature_dim;
    lt.table_rows = table_rows;
    lt.level_begin = 0;
    lt.level_end = num_lods;
    lt.stage_flags = 0;
    return 0;
}

}  // namespace shacira

using namespace shacira;

extern "C" {

int shacira_abi_version(void) { return SHACIRA_ABI_VERSION; }

const char *shacira_strerror(int code) {
    switch (code) {
        case 0: return "success";
        case SHACIRA_EINVAL: return "shacira: invalid argument (dim/sizes/bitwidth/null pointer)";
        case SHACIRA_EDTYPE: return "shacira: unsupported scalar type or operator shape";
        case SHACIRA_EODD: return "The codebook feature dimension needs to be a multiple of 2.";
        case SHACIRA_EWORKSPACE: return "shacira: workspace too small";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "shacira: unknown error";
}

int shacira_set_option(const char *name, int value) {
    if (!name) return SHACIRA_EINVAL;
    for (OptionSlot &sl : g_slots) {
        if (std::strcmp(name, sl.name)) continue;
        if (!option_value_ok(sl, value)) return SHACIRA_EINVAL;
        sl.value.store(value, std::memory_order_relaxed);
        return 0;
    }
    return SHACIRA_EINVAL;
}

int shacira_get_option(const char *name) {
    if (!name) return SHACIRA_EINVAL;
    for (OptionSlot &sl : g_slots)
        if (!std::strcmp(name, sl.name)) return sl.value.load(std::memory_order_relaxed);
    return SHACIRA_EINVAL;
}

size_t shacira_hashgrid_forward_workspace_bytes(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                                int codebook_bitwidth, const int32_t *resolutions_host,
                                                int64_t table_rows, int dtype) {
    options_snapshot();
    LevelTable lt;
    if (build_level_table(dim, num_lods, feature_dim, codebook_bitwidth, resolutions_host, table_rows, lt)) return 0;
    return hashgrid_forward_workspace(dim, dtype, lt, num_coords);
}

int shacira_hashgrid_forward(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                             const int32_t *resolutions_host, const int32_t *codebook_first_idx, int64_t table_rows,
                             const float *coords, const void *codebook, int dtype, void *feats, void *workspace,
                             size_t workspace_bytes, void *stream) {
    return shacira_hashgrid_forward_planned(dim, num_coords, num_lods, feature_dim, codebook_bitwidth, resolutions_host,
                                            codebook_first_idx, table_rows, coords, codebook, dtype, feats, nullptr, 0, 0,
                                            workspace, workspace_bytes, stream);
}

size_t shacira_hashgrid_plan_bytes(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                                   const int32_t *resolutions_host, int64_t table_rows, int dtype) {
    options_snapshot();
    LevelTable lt;
    if (build_level_table(dim, num_lods, feature_dim, codebook_bitwidth, resolutions_host, table_rows, lt)) return 0;
    if (dtype == SHACIRA_F64 || table_rows == 0 || !tiled_supported(dim, dtype, lt, num_coords)) return 0;
    return sample_plan_bytes(dim, num_coords);
}

int shacira_hashgrid_forward_planned(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                                     const int32_t *resolutions_host, const int32_t *codebook_first_idx, int64_t table_rows,
                                     const float *coords, const void *codebook, int dtype, void *feats, void *plan,
                                     size_t plan_bytes, int plan_flags, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    options_snapshot();
    LevelTable lt;
    int rc = build_level_table(dim, num_lods, feature_dim, codebook_bitwidth, resolutions_host, table_rows, lt);
    if (rc) return rc;
    if (dtype != SHACIRA_F32 && dtype != SHACIRA_F16 && dtype != SHACIRA_F64) return SHACIRA_EDTYPE;
    if (num_coords < 0) return SHACIRA_EINVAL;
    if (num_coords == 0) return 0;
    if (!codebook_first_idx || !coords || !codebook || !feats) return SHACIRA_EINVAL;
    const size_t need = hashgrid_forward_workspace(dim, dtype, lt, num_coords);
    if (need > 0 && (!workspace || workspace_bytes < need)) return SHACIRA_EWORKSPACE;
    // a plan buffer is used only by the shapes that sort (shacira_hashgrid_plan_bytes > 0); others ignore it
    const bool sorts = dtype != SHACIRA_F64 && table_rows > 0 && tiled_supported(dim, dtype, lt, num_coords);
    if (plan != nullptr && sorts && plan_bytes < sample_plan_bytes(dim, num_coords)) return SHACIRA_EWORKSPACE;
    if ((plan_flags & ~SHACIRA_PLAN_READY) != 0 || ((plan_flags & SHACIRA_PLAN_READY) != 0 && plan == nullptr)) return SHACIRA_EINVAL;
    return (int)hashgrid_forward_dispatch(dim, dtype, lt, codebook_first_idx, coords, codebook, feats, workspace,
                                          num_coords, (hipStream_t)stream, sorts ? plan : nullptr,
                                          (plan_flags & SHACIRA_PLAN_READY) != 0);
}

int shacira_hashgrid_debug_corners(int dim, int64_t num_coords, int num_lods, int codebook_bitwidth,
                                   const int32_t *resolutions_host, const float *coords, int32_t *corner_rows,
                                   float *corner_weights, void *stream) {
    LevelTable lt;
    int rc = build_level_table(dim, num_lods, 2, codebook_bitwidth, resolutions_host, 0, lt);
    if (rc) return rc;
    if (num_coords < 0 || num_coords * (int64_t)num_lods >= ((int64_t)1 << 31) * 256) return SHACIRA_EINVAL;
    if (num_coords > 0 && (!coords || (!corner_rows && !corner_weights))) return SHACIRA_EINVAL;
    return (int)hashgrid_debug_corners(dim, lt, coords, num_coords, corner_rows, corner_weights, (hipStream_t)stream);
}

size_t shacira_hashgrid_backward_workspace_bytes(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                                 int codebook_bitwidth, const int32_t *resolutions_host,
                                                 int64_t table_rows, int dtype) {
    options_snapshot();
    LevelTable lt;
    if (build_level_table(dim, num_lods, feature_dim, codebook_bitwidth, resolutions_host, table_rows, lt)) return 0;
    return hashgrid_backward_workspace(dim, dtype, lt, num_coords);
}

int shacira_hashgrid_backward(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                              const int32_t *resolutions_host, const int32_t *codebook_first_idx, int64_t table_rows,
                              const float *coords, const void *grad_output, int dtype, void *grad_codebook,
                              void *workspace, size_t workspace_bytes, void *stream) {
    return shacira_hashgrid_backward_levels(dim, num_coords, num_lods, feature_dim, codebook_bitwidth, resolutions_host,
                                            codebook_first_idx, table_rows, coords, grad_output, dtype, grad_codebook,
                                            0, num_lods, 0, workspace, workspace_bytes, stream);
}

static int backward_call(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                         const int32_t *resolutions_host, const int32_t *codebook_first_idx, int64_t table_rows,
                         const float *coords, const void *grad_output, int dtype, void *grad_codebook, int level_begin,
                         int level_end, int flags, const void *plan, size_t plan_bytes, void *workspace,
                         size_t workspace_bytes, void *stream);

size_t shacira_hashgrid_backward_planned_workspace_bytes(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                                         int codebook_bitwidth, const int32_t *resolutions_host,
                                                         int64_t table_rows, int dtype) {
    options_snapshot();
    LevelTable lt;
    if (build_level_table(dim, num_lods, feature_dim, codebook_bitwidth, resolutions_host, table_rows, lt)) return 0;
    LevelTable full = lt;
    const bool sorts = dtype != SHACIRA_F64 && table_rows > 0 && tiled_supported(dim, dtype, full, num_coords);
    if (!sorts) return hashgrid_backward_workspace(dim, dtype, lt, num_coords);   // no plan for this shape: the plain call
    return hashgrid_backward_workspace_planned(dim, dtype, lt, num_coords, true);
}

int shacira_hashgrid_backward_planned(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                                      const int32_t *resolutions_host, const int32_t *codebook_first_idx, int64_t table_rows,
                                      const float *coords, const void *grad_output, int dtype, void *grad_codebook,
                                      const void *plan, size_t plan_bytes, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    return backward_call(dim, num_coords, num_lods, feature_dim, codebook_bitwidth, resolutions_host, codebook_first_idx,
                         table_rows, coords, grad_output, dtype, grad_codebook, 0, num_lods, 0, plan, plan_bytes, workspace,
                         workspace_bytes, stream);
}

int shacira_hashgrid_backward_levels(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                                     const int32_t *resolutions_host, const int32_t *codebook_first_idx,
                                     int64_t table_rows, const float *coords, const void *grad_output, int dtype,
                                     void *grad_codebook, int level_begin, int level_end, int flags, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    return backward_call(dim, num_coords, num_lods, feature_dim, codebook_bitwidth, resolutions_host, codebook_first_idx,
                         table_rows, coords, grad_output, dtype, grad_codebook, level_begin, level_end, flags, nullptr, 0,
                         workspace, workspace_bytes, stream);
}

static int backward_call(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                         const int32_t *resolutions_host, const int32_t *codebook_first_idx, int64_t table_rows,
                         const float *coords, const void *grad_output, int dtype, void *grad_codebook, int level_begin,
                         int level_end, int flags, const void *plan, size_t plan_bytes, void *workspace,
                         size_t workspace_bytes, void *stream) {
    options_snapshot();
    LevelTable lt;
    int rc = build_level_table(dim, num_lods, feature_dim, codebook_bitwidth, resolutions_host, table_rows, lt);
    if (rc) return rc;
    if (level_begin < 0 || level_end > num_lods || level_begin >= level_end) return SHACIRA_EINVAL;
    const bool partial = !(level_begin == 0 && level_end == num_lods);
    // level ranges: fp32 tables, and (round 4) fp16 tables without the staged-gradient flags -- each call then converts its own
    // rows of the fp32 accumulation image only; double tables take whole calls
    if (partial && dtype == SHACIRA_F64) return SHACIRA_EDTYPE;
    if (dtype == SHACIRA_F16 && (flags & (SHACIRA_BWD_STAGE_ALL_LEVELS | SHACIRA_BWD_REUSE_STAGED)) != 0) return SHACIRA_EDTYPE;
    lt.level_begin = level_begin;
    lt.level_end = level_end;
    lt.stage_flags = flags & (SHACIRA_BWD_STAGE_ALL_LEVELS | SHACIRA_BWD_REUSE_STAGED);
    if (dtype != SHACIRA_F32 && dtype != SHACIRA_F16 && dtype != SHACIRA_F64) return SHACIRA_EDTYPE;
    if (num_coords < 0 || !grad_codebook) return SHACIRA_EINVAL;
    if (num_coords > 0 && (!codebook_first_idx || !coords || !grad_output)) return SHACIRA_EINVAL;
    // the plan of a shape whose forward sorts nothing does not exist: such a buffer is ignored
    if (plan != nullptr) {
        LevelTable full = lt;
        full.level_begin = 0;
        full.level_end = num_lods;
        const bool sorts = dtype != SHACIRA_F64 && table_rows > 0 && tiled_supported(dim, dtype, full, num_coords);
        if (!sorts) plan = nullptr;
        else if (plan_bytes < sample_plan_bytes(dim, num_coords)) return SHACIRA_EWORKSPACE;
    }
    // a planned call is sized by what it runs: shacira_hashgrid_backward_planned_workspace_bytes when its brick pass takes the
    // coarse levels (16-byte aligned grad_output), the plain size otherwise
    const size_t need = plan != nullptr
                            ? hashgrid_backward_workspace_planned(dim, dtype, lt, num_coords,
                                                                  (reinterpret_cast<uintptr_t>(grad_output) & 15u) == 0)
                            : hashgrid_backward_workspace(dim, dtype, lt, num_coords);
    if (need > 0 && (!workspace || workspace_bytes < need)) return SHACIRA_EWORKSPACE;
    return (int)hashgrid_backward_dispatch(dim, dtype, lt, codebook_first_idx, coords, grad_output, grad_codebook,
                                           workspace, workspace_bytes, num_coords, (hipStream_t)stream, plan);
}

int shacira_latent_decode_forward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                  const float *div, const float *matrix, const float *colscale, const float *shift,
                                  float clamp_weights, float *decoded, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1) return SHACIRA_EINVAL;
    if (!latent_decode_supported(latent_dim, feature_dim)) return SHACIRA_EDTYPE;
    if (num_rows == 0) return 0;
    if (!latent || !div || !matrix || !decoded) return SHACIRA_EINVAL;
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.decoded = decoded; a.rows = num_rows;
    return (int)latent_decode_dispatch(false, latent_dim, feature_dim, a, (hipStream_t)stream);
}

size_t shacira_latent_decode_backward_workspace_bytes(int64_t, int, int) { return latent_workspace_bytes(); }

int shacira_latent_decode_backward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                   const float *div, const float *matrix, const float *colscale, const float *shift,
                                   float clamp_weights, const float *grad_decoded, float *grad_latent,
                                   float *grad_matrix, float *grad_colscale, float *grad_shift, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1) return SHACIRA_EINVAL;
    if (!latent_decode_supported(latent_dim, feature_dim)) return SHACIRA_EDTYPE;
    if (!workspace || workspace_bytes < latent_workspace_bytes()) return SHACIRA_EWORKSPACE;
    if (num_rows > 0 && (!latent || !div || !matrix || !grad_decoded)) return SHACIRA_EINVAL;
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.grad_decoded = grad_decoded; a.grad_latent = grad_latent;
    a.grad_matrix = grad_matrix; a.grad_colscale = grad_colscale; a.grad_shift = grad_shift;
    a.partials = static_cast<double *>(workspace); a.rows = num_rows;
    return (int)latent_decode_dispatch(true, latent_dim, feature_dim, a, (hipStream_t)stream);
}

int shacira_latent_mlp_supported(int num_layers, const int32_t *widths_host) {
    return latent_mlp_supported(num_layers, widths_host) ? 1 : 0;
}

size_t shacira_latent_mlp_backward_workspace_bytes(int num_layers, const int32_t *widths_host) {
    return latent_mlp_workspace_bytes(num_layers, widths_host);
}

static int latent_mlp_call(bool bwd, int64_t num_rows, int num_layers, const int32_t *widths_host, const float *latent,
                           const float *uniforms, float temperature, int diff_sampling, const float *div,
                           const float *params, int activation, int final_activation, float clamp_weights, float *decoded,
                           const float *grad_decoded, float *grad_latent, float *grad_params, void *workspace,
                           size_t workspace_bytes, void *stream) {
    options_snapshot();
    if (num_rows < 0 || !latent_mlp_supported(num_layers, widths_host)) return SHACIRA_EINVAL;
    if (activation < SHACIRA_ACT_NONE || activation > SHACIRA_ACT_SINE30 || final_activation < SHACIRA_ACT_NONE ||
        final_activation > SHACIRA_ACT_SINE30)
        return SHACIRA_EINVAL;
    if (!div || !params) return SHACIRA_EINVAL;
    if (num_rows > 0 && (!latent || (bwd ? !grad_decoded : !decoded))) return SHACIRA_EINVAL;
    if (uniforms && !(temperature > 0.0f)) return SHACIRA_EINVAL;
    if (bwd) {
        if (!grad_params) return SHACIRA_EINVAL;
        if (!workspace || workspace_bytes < latent_mlp_workspace_bytes(num_layers, widths_host)) return SHACIRA_EWORKSPACE;
    }
    LatentMlpArgs a{};
    a.num_layers = num_layers; a.widths = widths_host; a.latent = latent; a.uniforms = uniforms; a.div = div;
    a.params = params; a.temperature = temperature; a.diff_sampling = diff_sampling; a.act = activation;
    a.final_act = final_activation; a.clampw = clamp_weights; a.decoded = decoded; a.grad_decoded = grad_decoded;
    a.grad_latent = grad_latent; a.grad_params = grad_params; a.partials = static_cast<double *>(workspace);
    a.rows = num_rows;
    return (int)latent_mlp_dispatch(bwd, a, (hipStream_t)stream);
}

int shacira_latent_mlp_forward(int64_t num_rows, int num_layers, const int32_t *widths_host, const float *latent,
                               const float *uniforms, float temperature, int diff_sampling, const float *div,
                               const float *params, int activation, int final_activation, float clamp_weights,
                               float *decoded, void *stream) {
    return latent_mlp_call(false, num_rows, num_layers, widths_host, latent, uniforms, temperature, diff_sampling, div, params,
                           activation, final_activation, clamp_weights, decoded, nullptr, nullptr, nullptr, nullptr, 0,
                           stream);
}

int shacira_latent_mlp_backward(int64_t num_rows, int num_layers, const int32_t *widths_host, const float *latent,
                                const float *uniforms, float temperature, int diff_sampling, const float *div,
                                const float *params, int activation, int final_activation, float clamp_weights,
                                const float *grad_decoded, float *grad_latent, float *grad_params, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return latent_mlp_call(true, num_rows, num_layers, widths_host, latent, uniforms, temperature, diff_sampling, div, params,
                           activation, final_activation, clamp_weights, nullptr, grad_decoded, grad_latent, grad_params,
                           workspace, workspace_bytes, stream);
}

int shacira_latent_decode_sga_forward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                      const float *uniforms, float temperature, int diff_sampling, const float *div,
                                      const float *matrix, const float *colscale, const float *shift,
                                      float clamp_weights, float *decoded, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1 || !(temperature > 0.0f)) return SHACIRA_EINVAL;
    if (!latent_decode_supported(latent_dim, feature_dim)) return SHACIRA_EDTYPE;
    if (num_rows == 0) return 0;
    if (!latent || !uniforms || !div || !matrix || !decoded) return SHACIRA_EINVAL;
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.decoded = decoded; a.rows = num_rows;
    a.uniforms = uniforms; a.temperature = temperature; a.diff_sampling = diff_sampling;
    return (int)latent_decode_dispatch(false, latent_dim, feature_dim, a, (hipStream_t)stream);
}

int shacira_latent_decode_sga_backward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                       const float *uniforms, float temperature, int diff_sampling, const float *div,
                                       const float *matrix, const float *colscale, const float *shift,
                                       float clamp_weights, const float *grad_decoded, float *grad_latent,
                                       float *grad_matrix, float *grad_colscale, float *grad_shift, void *workspace,
                                       size_t workspace_bytes, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1 || !(temperature > 0.0f)) return SHACIRA_EINVAL;
    if (!latent_decode_supported(latent_dim, feature_dim)) return SHACIRA_EDTYPE;
    if (!workspace || workspace_bytes < latent_workspace_bytes()) return SHACIRA_EWORKSPACE;
    if (num_rows > 0 && (!latent || !uniforms || !div || !matrix || !grad_decoded)) return SHACIRA_EINVAL;
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.grad_decoded = grad_decoded; a.grad_latent = grad_latent;
    a.grad_matrix = grad_matrix; a.grad_colscale = grad_colscale; a.grad_shift = grad_shift;
    a.partials = static_cast<double *>(workspace); a.rows = num_rows;
    a.uniforms = uniforms; a.temperature = temperature; a.diff_sampling = diff_sampling;
    return (int)latent_decode_dispatch(true, latent_dim, feature_dim, a, (hipStream_t)stream);
}

// The same pair with the temperature in DEVICE memory (one float, read by the kernels): a training step captured into a HIP
// graph anneals the temperature between replays (base_trainer.py:155-157 decays it every iteration) without re-capturing.
int shacira_latent_decode_sga_forward_tdev(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                           const float *uniforms, const float *temperature_dev, int diff_sampling,
                                           const float *div, const float *matrix, const float *colscale, const float *shift,
                                           float clamp_weights, float *decoded, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1 || !temperature_dev) return SHACIRA_EINVAL;
    if (!latent_decode_supported(latent_dim, feature_dim)) return SHACIRA_EDTYPE;
    if (num_rows == 0) return 0;
    if (!latent || !uniforms || !div || !matrix || !decoded) return SHACIRA_EINVAL;
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.decoded = decoded; a.rows = num_rows;
    a.uniforms = uniforms; a.temperature = 1.0f; a.temperature_dev = temperature_dev; a.diff_sampling = diff_sampling;
    return (int)latent_decode_dispatch(false, latent_dim, feature_dim, a, (hipStream_t)stream);
}

int shacira_latent_decode_sga_backward_tdev(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                            const float *uniforms, const float *temperature_dev, int diff_sampling,
                                            const float *div, const float *matrix, const float *colscale, const float *shift,
                                            float clamp_weights, const float *grad_decoded, float *grad_latent,
                                            float *grad_matrix, float *grad_colscale, float *grad_shift, void *workspace,
                                            size_t workspace_bytes, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1 || !temperature_dev) return SHACIRA_EINVAL;
    if (!latent_decode_supported(latent_dim, feature_dim)) return SHACIRA_EDTYPE;
    if (!workspace || workspace_bytes < latent_workspace_bytes()) return SHACIRA_EWORKSPACE;
    if (num_rows > 0 && (!latent || !uniforms || !div || !matrix || !grad_decoded)) return SHACIRA_EINVAL;
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.grad_decoded = grad_decoded; a.grad_latent = grad_latent;
    a.grad_matrix = grad_matrix; a.grad_colscale = grad_colscale; a.grad_shift = grad_shift;
    a.partials = static_cast<double *>(workspace); a.rows = num_rows;
    a.uniforms = uniforms; a.temperature = 1.0f; a.temperature_dev = temperature_dev; a.diff_sampling = diff_sampling;
    return (int)latent_decode_dispatch(true, latent_dim, feature_dim, a, (hipStream_t)stream);
}

static int levels_args_ok(int num_levels, const int64_t *row_offsets_host, int64_t num_rows, int latent_dim,
                          int feature_dim) {
    if (num_levels < 1 || num_levels > SHACIRA_MAX_LODS || !row_offsets_host) return SHACIRA_EINVAL;
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1) return SHACIRA_EINVAL;
    if (!latent_decode_supported(latent_dim, feature_dim)) return SHACIRA_EDTYPE;
    int64_t prev = 0;
    for (int l = 0; l <= num_levels; ++l) {
        const int64_t o = row_offsets_host[l];
        if (o < 0 || o > num_rows) return SHACIRA_EINVAL;
        if (l < num_levels && o < prev) return SHACIRA_EINVAL;   // level starts ascend; the LAST boundary may fall short
        prev = o;
    }
    return 0;
}

int shacira_latent_decode_levels_forward(int num_levels, const int64_t *row_offsets_host, int64_t num_rows,
                                         int latent_dim, int feature_dim, const float *latent, const float *uniforms,
                                         float temperature, int diff_sampling, const float *div, const float *matrix,
                                         const float *colscale, const float *shift, float clamp_weights, float *decoded,
                                         void *stream) {
    if (int rc = levels_args_ok(num_levels, row_offsets_host, num_rows, latent_dim, feature_dim)) return rc;
    if (uniforms && !(temperature > 0.0f)) return SHACIRA_EINVAL;
    if (num_rows == 0) return 0;
    if (!latent || !div || !matrix || !decoded) return SHACIRA_EINVAL;
    // rows no level owns (before the first level, after the last boundary) decode to 0
    hipError_t e = hipMemsetAsync(decoded, 0, (size_t)num_rows * feature_dim * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.decoded = decoded; a.rows = num_rows;
    a.uniforms = uniforms; a.temperature = temperature; a.diff_sampling = diff_sampling;
    return (int)latent_decode_levels_dispatch(false, latent_dim, feature_dim, num_levels, row_offsets_host, a,
                                              (hipStream_t)stream);
}

int shacira_latent_decode_levels_backward(int num_levels, const int64_t *row_offsets_host, int64_t num_rows,
                                          int latent_dim, int feature_dim, const float *latent, const float *uniforms,
                                          float temperature, int diff_sampling, const float *div, const float *matrix,
                                          const float *colscale, const float *shift, float clamp_weights,
                                          const float *grad_decoded, float *grad_latent, float *grad_matrix,
                                          float *grad_colscale, float *grad_shift, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    if (int rc = levels_args_ok(num_levels, row_offsets_host, num_rows, latent_dim, feature_dim)) return rc;
    if (uniforms && !(temperature > 0.0f)) return SHACIRA_EINVAL;
    if (!workspace || workspace_bytes < latent_workspace_bytes()) return SHACIRA_EWORKSPACE;
    if (num_rows > 0 && (!latent || !div || !matrix || !grad_decoded)) return SHACIRA_EINVAL;
    if (grad_latent && num_rows > 0) {
        hipError_t e = hipMemsetAsync(grad_latent, 0, (size_t)num_rows * latent_dim * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    DecodeArgs a{};
    a.latent = latent; a.div = div; a.matrix = matrix; a.colscale = colscale; a.shift = shift;
    a.clampw = clamp_weights; a.grad_decoded = grad_decoded; a.grad_latent = grad_latent;
    a.grad_matrix = grad_matrix; a.grad_colscale = grad_colscale; a.grad_shift = grad_shift;
    a.partials = static_cast<double *>(workspace); a.rows = num_rows;
    a.uniforms = uniforms; a.temperature = temperature; a.diff_sampling = diff_sampling;
    return (int)latent_decode_levels_dispatch(true, latent_dim, feature_dim, num_levels, row_offsets_host, a,
                                              (hipStream_t)stream);
}

int shacira_latent_multi_supported(int latent_dim, int feature_dim, int num_decoders) {
    return latent_multi_supported(latent_dim, feature_dim, num_decoders) ? 1 : 0;
}

static int multi_args_ok(int64_t num_rows, int latent_dim, int feature_dim, int num_decoders, float temperature) {
    if (num_rows < 0 || latent_dim < 1 || feature_dim < 1 || num_decoders < 1 || !(temperature > 0.0f)) return SHACIRA_EINVAL;
    if (!latent_multi_supported(latent_dim, feature_dim, num_decoders)) return SHACIRA_EDTYPE;
    return 0;
}

int shacira_latent_multi_decode_forward(int64_t num_rows, int latent_dim, int feature_dim, int num_decoders,
                                        const float *latent, const float *alpha, const float *uniforms,
                                        float temperature, int straight_through, int diff_sampling, const float *div,
                                        const float *scale, const float *dft, const float *shift, float clamp_weights,
                                        float *decoded, void *stream) {
    if (int rc = multi_args_ok(num_rows, latent_dim, feature_dim, num_decoders, temperature)) return rc;
    if (num_rows == 0) return 0;
    if (!latent || !alpha || !div || !scale || !decoded) return SHACIRA_EINVAL;
    return (int)latent_multi_dispatch(false, latent_dim, feature_dim, dft != nullptr, latent, alpha, num_decoders, uniforms,
                                      temperature, straight_through, diff_sampling, div, scale, dft, shift, clamp_weights,
                                      num_rows, decoded, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                      (hipStream_t)stream);
}

int shacira_latent_multi_decode_backward(int64_t num_rows, int latent_dim, int feature_dim, int num_decoders,
                                         const float *latent, const float *alpha, const float *uniforms,
                                         float temperature, int straight_through, int diff_sampling, const float *div,
                                         const float *scale, const float *dft, const float *shift, float clamp_weights,
                                         const float *grad_decoded, float *grad_latent, float *grad_alpha,
                                         float *grad_scale, float *grad_shift, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    if (int rc = multi_args_ok(num_rows, latent_dim, feature_dim, num_decoders, temperature)) return rc;
    if (!workspace || workspace_bytes < latent_workspace_bytes()) return SHACIRA_EWORKSPACE;
    if (!grad_scale || (num_rows > 0 && (!latent || !alpha || !div || !scale || !grad_decoded))) return SHACIRA_EINVAL;
    return (int)latent_multi_dispatch(true, latent_dim, feature_dim, dft != nullptr, latent, alpha, num_decoders, uniforms,
                                      temperature, straight_through, diff_sampling, div, scale, dft, shift, clamp_weights,
                                      num_rows, nullptr, grad_decoded, grad_latent, grad_alpha, grad_scale, grad_shift,
                                      static_cast<double *>(workspace), (hipStream_t)stream);
}

int shacira_latent_symbol_range(int64_t num_rows, int latent_dim, const float *latent, int32_t *minmax, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || !minmax || (num_rows > 0 && !latent)) return SHACIRA_EINVAL;
    if (!symbols_supported(latent_dim)) return SHACIRA_EDTYPE;
    return (int)symbol_range_launch(latent, num_rows, latent_dim, minmax, (hipStream_t)stream);
}

int shacira_latent_symbol_histogram(int64_t num_rows, int latent_dim, const float *latent, const int32_t *minmax,
                                    int nbins, uint64_t *counts, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || nbins < 1 || !minmax || !counts || (num_rows > 0 && !latent))
        return SHACIRA_EINVAL;
    if (!symbols_supported(latent_dim)) return SHACIRA_EDTYPE;
    return (int)symbol_histogram_launch(latent, num_rows, latent_dim, minmax, nbins, counts, (hipStream_t)stream);
}

size_t shacira_rc_encode_bound(int64_t n) { return rc_encode_bound(n < 0 ? 0 : n); }

int shacira_rc_encode(const int32_t *symbols_host, int64_t n, const uint32_t *freq_host, int num_symbols,
                      uint8_t *out_host, size_t capacity, size_t *out_len) {
    return rc_encode(symbols_host, n, freq_host, num_symbols, out_host, capacity, out_len);
}

int shacira_rc_decode(const uint8_t *in_host, size_t len, const uint32_t *freq_host, int num_symbols, int64_t n,
                      int32_t *symbols_host) {
    return rc_decode(in_host, len, freq_host, num_symbols, n, symbols_host);
}

static int pack_args_ok(int64_t S, int64_t R, int C) {
    if (S < 0 || R < 0 || C < 1) return SHACIRA_EINVAL;
    if (C > 16) return SHACIRA_EDTYPE;
    return 0;
}

int shacira_pack_integrate_forward(int64_t num_samples, int64_t num_packs, int channels, const float *feats,
                                   const float *tau, const int64_t *pack_start, float *ray_feats, float *weights,
                                   void *stream) {
    if (int rc = pack_args_ok(num_samples, num_packs, channels)) return rc;
    if (num_packs == 0) return 0;
    if (!pack_start || !ray_feats || (num_samples > 0 && (!feats || !tau || !weights))) return SHACIRA_EINVAL;
    return (int)pack_integrate_launch(false, num_packs, channels, feats, tau, pack_start, ray_feats, weights, nullptr,
                                      nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int shacira_pack_integrate_backward(int64_t num_samples, int64_t num_packs, int channels, const float *feats,
                                    const float *tau, const int64_t *pack_start, const float *grad_ray_feats,
                                    const float *grad_weights, float *grad_feats, float *grad_tau, void *stream) {
    if (int rc = pack_args_ok(num_samples, num_packs, channels)) return rc;
    if (num_packs == 0 || num_samples == 0) return 0;
    if (!pack_start || !feats || !tau || !grad_ray_feats || !grad_feats || !grad_tau) return SHACIRA_EINVAL;
    return (int)pack_integrate_launch(true, num_packs, channels, feats, tau, pack_start, nullptr, nullptr,
                                      grad_ray_feats, grad_weights, grad_feats, grad_tau, (hipStream_t)stream);
}

int shacira_pack_sum(int64_t num_samples, int64_t num_packs, int channels, const float *x, const int64_t *pack_start,
                     float *out, void *stream) {
    if (int rc = pack_args_ok(num_samples, num_packs, channels)) return rc;
    if (num_packs == 0) return 0;
    if (!pack_start || !out || (num_samples > 0 && !x)) return SHACIRA_EINVAL;
    return (int)pack_sum_launch(false, num_packs, channels, x, pack_start, out, (hipStream_t)stream);
}

int shacira_pack_broadcast(int64_t num_samples, int64_t num_packs, int channels, const float *per_pack,
                           const int64_t *pack_start, float *out, void *stream) {
    if (int rc = pack_args_ok(num_samples, num_packs, channels)) return rc;
    if (num_packs == 0 || num_samples == 0) return 0;
    if (!pack_start || !out || !per_pack) return SHACIRA_EINVAL;
    return (int)pack_sum_launch(true, num_packs, channels, per_pack, pack_start, out, (hipStream_t)stream);
}

static int march_args_ok(int64_t num_rays, const float *origins, const float *dirs, const uint8_t *occ, int level) {
    if (num_rays < 0 || level < 0 || level > 10) return SHACIRA_EINVAL;
    if (num_rays > 0 && (!origins || !dirs || !occ)) return SHACIRA_EINVAL;
    return 0;
}

int shacira_raymarch_ray_count(int64_t num_rays, int num_samples, const float *origins, const float *dirs,
                               float dist_min, float dist_max, const float *lin, const float *jitter,
                               const uint8_t *occupancy, int level, int32_t *counts, void *stream) {
    if (int rc = march_args_ok(num_rays, origins, dirs, occupancy, level)) return rc;
    if (num_samples < 1 || (num_rays > 0 && (!lin || !jitter || !counts))) return SHACIRA_EINVAL;
    return (int)raymarch_ray_launch(false, num_rays, num_samples, origins, dirs, dist_min, dist_max, lin, jitter,
                                    occupancy, level, counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    INT64_MAX, (hipStream_t)stream);
}

int shacira_raymarch_ray_emit(int64_t num_rays, int num_samples, const float *origins, const float *dirs,
                              float dist_min, float dist_max, const float *lin, const float *jitter,
                              const uint8_t *occupancy, int level, const int64_t *offsets, int64_t *ridx,
                              float *samples, float *depth, float *deltas, uint8_t *boundary, void *stream) {
    if (int rc = march_args_ok(num_rays, origins, dirs, occupancy, level)) return rc;
    if (num_samples < 1 || (num_rays > 0 && (!lin || !jitter || !offsets))) return SHACIRA_EINVAL;
    return (int)raymarch_ray_launch(true, num_rays, num_samples, origins, dirs, dist_min, dist_max, lin, jitter,
                                    occupancy, level, nullptr, offsets, ridx, samples, depth, deltas, boundary,
                                    INT64_MAX, (hipStream_t)stream);
}

int shacira_raymarch_ray_emit_capped(int64_t num_rays, int num_samples, const float *origins, const float *dirs,
                                     float dist_min, float dist_max, const float *lin, const float *jitter,
                                     const uint8_t *occupancy, int level, const int64_t *offsets, int64_t capacity,
                                     int64_t *ridx, float *samples, float *depth, float *deltas, uint8_t *boundary,
                                     void *stream) {
    if (int rc = march_args_ok(num_rays, origins, dirs, occupancy, level)) return rc;
    if (num_samples < 1 || capacity < 0 || (num_rays > 0 && (!lin || !jitter || !offsets))) return SHACIRA_EINVAL;
    return (int)raymarch_ray_launch(true, num_rays, num_samples, origins, dirs, dist_min, dist_max, lin, jitter,
                                    occupancy, level, nullptr, offsets, ridx, samples, depth, deltas, boundary,
                                    capacity, (hipStream_t)stream);
}

int shacira_raytrace_dense_count(int64_t num_rays, const float *origins, const float *dirs, const uint8_t *occupancy,
                                 int level, int32_t *counts, void *stream) {
    if (int rc = march_args_ok(num_rays, origins, dirs, occupancy, level)) return rc;
    if (num_rays > 0 && !counts) return SHACIRA_EINVAL;
    return (int)raytrace_dense_launch(false, num_rays, origins, dirs, occupancy, level, counts, nullptr, nullptr,
                                      nullptr, nullptr, (hipStream_t)stream);
}

int shacira_raytrace_dense_emit(int64_t num_rays, const float *origins, const float *dirs, const uint8_t *occupancy,
                                int level, const int64_t *offsets, int32_t *ridx, int32_t *pidx, float *depth,
                                void *stream) {
    if (int rc = march_args_ok(num_rays, origins, dirs, occupancy, level)) return rc;
    if (num_rays > 0 && !offsets) return SHACIRA_EINVAL;
    return (int)raytrace_dense_launch(true, num_rays, origins, dirs, occupancy, level, nullptr, offsets, ridx, pidx,
                                      depth, (hipStream_t)stream);
}

size_t shacira_entropy_bits_workspace_bytes(int64_t, int) { return latent_workspace_bytes(); }

int shacira_entropy_bits_forward(int64_t num_rows, int latent_dim, int num_layers, const float *latent,
                                 const float *noise, const float *params, float *total_bits, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    if (num_rows < 0 || latent_dim < 1 || num_layers < 1 || num_layers > 4) return SHACIRA_EINVAL;
    if (!entropy_supported(latent_dim)) return SHACIRA_EDTYPE;
    if (!workspace || workspace_bytes < latent_workspace_bytes()) return SHACIRA_EWORKSPACE;
    if (!params || !total_bits || (num_rows > 0 && !latent)) return SHACIRA_EINVAL;
    EntropyArgs a{};
    a.latent = latent; a.noise = noise; a.params = params; a.num_layers = num_layers; a.total_bits = total_bits;
    a.partials = static_cast<double *>(workspace); a.rows = num_rows;
    return (int)entropy_dispatch(false, latent_dim, a, (hipStream_t)stream);
}

int shacira_entropy_bits_backward(int64_t num_rows, int latent_dim, int num_layers, const float *latent,
                                  const float *noise, const float *params, const float *grad_total_bits,
                                  float *grad_latent, float *grad_params, void *workspace, size_t workspace_bytes,
                                  void *stream) {
    if (num_rows < 0 || latent_dim < 1 || num_layers < 1 || num_layers > 4) return SHACIRA_EINVAL;
    if (!entropy_supported(latent_dim)) return SHACIRA_EDTYPE;
    if (!workspace || workspace_bytes < latent_workspace_bytes()) return SHACIRA_EWORKSPACE;
    if (!params || !grad_total_bits || (num_rows > 0 && !latent)) return SHACIRA_EINVAL;
    EntropyArgs a{};
    a.latent = latent; a.noise = noise; a.params = params; a.num_layers = num_layers; a.grad_total = grad_total_bits;
    a.grad_latent = grad_latent; a.grad_params = grad_params;
    a.partials = static_cast<double *>(workspace); a.rows = num_rows;
    return (int)entropy_dispatch(true, latent_dim, a, (hipStream_t)stream);
}

int shacira_adam_step(int64_t numel, float *param, float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, int zero_grad, void *stream) {
    if (numel < 0 || step < 1) return SHACIRA_EINVAL;
    if (numel > 0 && (!param || !grad || !exp_avg || !exp_avg_sq)) return SHACIRA_EINVAL;
    if (!(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f)) return SHACIRA_EINVAL;
    return (int)adam_step_launch(param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, step,
                                 nullptr, zero_grad, (hipStream_t)stream);
}

int shacira_adam_step_capturable(int64_t numel, float *param, float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, const int32_t *step_dev,
                                 int zero_grad, void *stream) {
    if (numel < 0 || !step_dev) return SHACIRA_EINVAL;
    if (numel > 0 && (!param || !grad || !exp_avg || !exp_avg_sq)) return SHACIRA_EINVAL;
    if (!(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f)) return SHACIRA_EINVAL;
    return (int)adam_step_launch(param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, 1,
                                 step_dev, zero_grad, (hipStream_t)stream);
}

int shacira_adam_step_multi(int num_tensors, const int64_t *numel_host, float *const *param, float *const *grad,
                            float *const *exp_avg, float *const *exp_avg_sq, const float *lr_host,
                            const float *weight_decay_host, float beta1, float beta2, float eps, int step,
                            const int32_t *step_dev, int zero_grad, void *stream) {
    if (num_tensors < 0 || num_tensors > 32) return SHACIRA_EINVAL;
    if (num_tensors == 0) return 0;
    if (!numel_host || !param || !grad || !exp_avg || !exp_avg_sq || !lr_host || !weight_decay_host)
        return SHACIRA_EINVAL;
    if (!step_dev && step < 1) return SHACIRA_EINVAL;
    if (!(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f)) return SHACIRA_EINVAL;
    for (int t = 0; t < num_tensors; ++t)
        if (numel_host[t] < 0 || (numel_host[t] > 0 && (!param[t] || !grad[t] || !exp_avg[t] || !exp_avg_sq[t])))
            return SHACIRA_EINVAL;
    return (int)adam_multi_launch(num_tensors, param, grad, exp_avg, exp_avg_sq, numel_host, lr_host,
                                  weight_decay_host, beta1, beta2, eps, step, step_dev, zero_grad,
                                  (hipStream_t)stream);
}

int shacira_stream_probe(int kind, const void *src, void *dst, size_t bytes, void *stream) {
    if (kind < 0 || kind > 2 || (bytes & 15u)) return SHACIRA_EINVAL;
    if (bytes == 0) return 0;
    if ((kind != 1 && !src) || (kind != 0 && !dst)) return SHACIRA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) return SHACIRA_EINVAL;
    return (int)stream_probe_launch(kind, src, dst, bytes, kind == 0 ? static_cast<uint32_t *>(dst) : nullptr,
                                    (hipStream_t)stream);
}

int shacira_mlp_supported(int in_dim, int hidden_dim, int num_hidden, int out_dim) {
    options_snapshot();
    return mlp_supported(in_dim, hidden_dim, num_hidden, out_dim) ? 1 : 0;
}

size_t shacira_mlp_backward_workspace_bytes(int in_dim, int hidden_dim, int num_hidden, int out_dim) {
    options_snapshot();
    return mlp_supported(in_dim, hidden_dim, num_hidden, out_dim)
               ? mlp_workspace_bytes(in_dim, hidden_dim, num_hidden, out_dim) : 0;
}

int shacira_mlp_forward(int64_t num_rows, int in_dim, int hidden_dim, int num_hidden, int out_dim, const float *x,
                        const float *params, float *y, void *stream) {
    options_snapshot();
    if (num_rows < 0) return SHACIRA_EINVAL;
    if (!mlp_supported(in_dim, hidden_dim, num_hidden, out_dim)) return SHACIRA_EDTYPE;
    if (num_rows == 0) return 0;
    if (!x || !params || !y) return SHACIRA_EINVAL;
    return (int)mlp_dispatch(false, in_dim, hidden_dim, num_hidden, out_dim, num_rows, x, params, y, nullptr, nullptr,
                             nullptr, nullptr, (hipStream_t)stream);
}

int shacira_mlp_backward(int64_t num_rows, int in_dim, int hidden_dim, int num_hidden, int out_dim, const float *x,
                         const float *params, const float *grad_y, float *grad_x, float *grad_params, void *workspace,
                         size_t workspace_bytes, void *stream) {
    options_snapshot();
    if (num_rows < 0) return SHACIRA_EINVAL;
    if (!mlp_supported(in_dim, hidden_dim, num_hidden, out_dim)) return SHACIRA_EDTYPE;
    if (!workspace || workspace_bytes < mlp_workspace_bytes(in_dim, hidden_dim, num_hidden, out_dim))
        return SHACIRA_EWORKSPACE;
    if (!params || !grad_params || (num_rows > 0 && (!x || !grad_y))) return SHACIRA_EINVAL;
    return (int)mlp_dispatch(true, in_dim, hidden_dim, num_hidden, out_dim, num_rows, x, params, nullptr, grad_y,
                             grad_x, grad_params, static_cast<double *>(workspace), (hipStream_t)stream);
}

}  // extern "C"
