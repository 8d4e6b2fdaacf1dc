/*
 * shacira_hip.h -- C-ABI of libshacira_hip.so, the MI355X (gfx950) implementation of SHACIRA's hash-grid
 * interpolation and latent quantisation / entropy path.
 *
 * This is the drop-in boundary: exactly the operators the reference binds through pybind11 in
 * wisp/csrc/bindings.cpp:24-28 (declared in wisp/csrc/ops/hashgrid_interpolate.h:18-50), restated with plain
 * pointers and sizes (no ATen types), plus the per-entry latent decode / entropy-bit operators that the
 * reference evaluates as chains of ATen elementwise kernels
 * (wisp/models/latent_decoders/basic_latent_decoder.py:182-198, wisp/models/grids/latent_grid.py:122-136,
 * wisp/models/prob_models/bit_estimator.py:27-65).
 *
 * Conventions (all entry points):
 *   - every buffer is CALLER-OWNED device memory (hipMalloc / the PyTorch caching allocator) unless the name
 *     ends in _host; the library never allocates or frees device memory and never synchronises the host (the backward
 *     of large batches with LDS-resident levels (3-D: n * L * F >= 7 * 2^21, 2-D: >= 2^23) creates, once per host thread and
 *     device, one non-blocking side stream and three events that it forks from / joins back into `stream`: stream semantics are unchanged, HIP-graph capture works
 *     after one eager call). Those objects belong to the CURRENT device (hipGetDevice): like every HIP launch, a call
 *     must be made with the device of `stream` and of its buffers current;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and the call
 *     returns immediately; calls are reentrant and thread-safe (the backward runs on autograd worker threads);
 *   - return value: 0 on success, a negative SHACIRA_E* code for invalid arguments (nothing enqueued), or a
 *     positive hipError_t if the launch failed. shacira_strerror() describes either;
 *   - tensors are dense row-major.
 */
#ifndef SHACIRA_HIP_H
#define SHACIRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHACIRA_ABI_VERSION 11

#if defined(__GNUC__)
#define SHACIRA_API __attribute__((visibility("default")))
#else
#define SHACIRA_API
#endif

/* table / feature scalar types (AT_DISPATCH_FLOATING_TYPES_AND_HALF in hashgrid_interpolate_cuda.cu:125) */
#define SHACIRA_F32 0
#define SHACIRA_F16 1
/* double tables (the third type of the reference's dispatch macro, .cu:125,290). Forward: every table value narrowed to
 * float, fp32 interpolation, result widened (.cu:96-107) -- the reference-shaped kernel only. Backward: (float)(grad * weight)
 * accumulated with atomicAdd(double). NOTE the reference's own double backward is broken: .cu:212-221 adds a float through
 * `(float*)(grad_codebook + ...)`, i.e. into the LOW WORD of each double; this library computes the intended gradient. */
#define SHACIRA_F64 2

#define SHACIRA_MAX_LODS 32

#define SHACIRA_EINVAL   (-1) /* bad dim / sizes / bitwidth / null pointer              */
#define SHACIRA_EDTYPE   (-2) /* unsupported scalar type                                */
#define SHACIRA_EODD     (-3) /* feature_dim is odd (wisp/ops/grid.py:75-76 raises too) */
#define SHACIRA_EWORKSPACE (-4) /* workspace too small for the requested algorithm      */

SHACIRA_API int shacira_abi_version(void);
SHACIRA_API const char *shacira_strerror(int code);

/*
 * Forward: replaces hashgrid_interpolate_cuda / hashgrid_interpolate2d_cuda
 * (wisp/csrc/ops/hashgrid_interpolate.cpp:44-66 and :130-152; kernels hashgrid_interpolate_cuda.cu:47-109,
 * hashgrid_interpolate2d_cuda.cu:44-99). All levels are evaluated by one launch.
 *
 *   dim                2 or 3 (coords are [num_coords, dim] fp32 in [-1, 1])
 *   codebook           [table_rows, feature_dim] of `dtype`; levels concatenated, level l starts at row
 *                      codebook_first_idx[l]
 *   codebook_first_idx DEVICE int32 [num_lods] (the module's `codebook_lod_first_idx` buffer)
 *   resolutions_host   HOST int32 [num_lods] (the Python list the reference converts to std::vector<int32_t>)
 *   feats              out, [num_coords, num_lods*feature_dim] of `dtype` (level-major, feature-minor)
 *   table_rows         total rows; used only to keep the reference's out-of-table corner (coord == +1 on a
 *                      dense level with res >= 258, weight 0) memory-safe
 *   workspace          scratch of at least shacira_hashgrid_forward_workspace_bytes(...) bytes (level-major staging
 *                      of the features, and the sample sort of large batches; may be NULL when that returns 0)
 */
SHACIRA_API size_t shacira_hashgrid_forward_workspace_bytes(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                                int codebook_bitwidth, const int32_t *resolutions_host,
                                                int64_t table_rows, int dtype);

SHACIRA_API int shacira_hashgrid_forward(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                             const int32_t *resolutions_host, const int32_t *codebook_first_idx,
                             int64_t table_rows, const float *coords, const void *codebook, int dtype,
                             void *feats, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Test hook (not used by any caller of the operators): the integers and weights behind one lookup, so that "hash
 * indices bit-exact" can be checked directly instead of through the features. For every (sample, level):
 *   corner_rows    int32 [num_coords, num_lods, 2^dim]  level-local row of corner k, i.e. hash_index / hash_index2d of
 *                  the reference (hashgrid_interpolate_cuda.cu:17-39, hashgrid_interpolate2d_cuda.cu:17-36) applied to
 *                  the corner positions of .cu:86-94 (k: bit2 -> x, bit1 -> y, bit0 -> z; 2-D: bit1 -> x, bit0 -> y)
 *   corner_weights fp32  [num_coords, num_lods, 2^dim]  the weight products of .cu:77-84 / 2d.cu:72-75
 * computed by the same device function every forward / backward kernel of this library uses. Either output may be NULL.
 */
SHACIRA_API int shacira_hashgrid_debug_corners(int dim, int64_t num_coords, int num_lods, int codebook_bitwidth,
                                   const int32_t *resolutions_host, const float *coords, int32_t *corner_rows,
                                   float *corner_weights, void *stream);

/*
 * Backward: replaces hashgrid_interpolate_backward_cuda / hashgrid_interpolate2d_backward_cuda
 * (hashgrid_interpolate.cpp:68-100 and :154-186; kernels .cu:143-221, 2d.cu:133-208).
 *
 *   grad_output    [num_coords, num_lods*feature_dim] of `dtype`
 *   grad_codebook  out, [table_rows, feature_dim] of `dtype`; the call overwrites it completely
 *                  (the reference's at::zeros_like + atomicAdd), no pre-zeroing needed
 *   workspace      scratch of at least shacira_hashgrid_backward_workspace_bytes(...) bytes (may be NULL if 0)
 *
 * The reference's `require_grad_coords` output is dead code there (computed into a tensor that is never
 * returned, hashgrid_interpolate.cpp:96-97; wisp/ops/grid.py:111 returns None for coords) and is not provided.
 */
SHACIRA_API size_t shacira_hashgrid_backward_workspace_bytes(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                                 int codebook_bitwidth, const int32_t *resolutions_host,
                                                 int64_t table_rows, int dtype);

SHACIRA_API int shacira_hashgrid_backward(int dim, int64_t num_coords, int num_lods, int feature_dim, int codebook_bitwidth,
                              const int32_t *resolutions_host, const int32_t *codebook_first_idx,
                              int64_t table_rows, const float *coords, const void *grad_output, int dtype,
                              void *grad_codebook, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Same, restricted to levels [level_begin, level_end): writes (completely) only the rows of those levels,
 * [codebook_first_idx[level_begin], codebook_first_idx[level_end]) (to table_rows for level_end == num_lods), and leaves
 * the rest of grad_codebook untouched. Lets a data-parallel caller start the all-reduce of the finished rows while the
 * remaining levels are still being computed. fp32 tables, and fp16 tables without the two flags below (a half table converts
 * only the call's own rows of its fp32 accumulation image); double tables take whole calls. Same workspace size as the full call.
 *   flags: SHACIRA_BWD_STAGE_ALL_LEVELS  stage (transpose) the gradients of ALL levels into the workspace, not only
 *                                        this call's, so that later calls on the SAME workspace can skip that pass;
 *          SHACIRA_BWD_REUSE_STAGED      the workspace already holds them (set by an earlier call with the flag above,
 *                                        same arguments, same workspace, same stream order).
 */
#define SHACIRA_BWD_STAGE_ALL_LEVELS 1
#define SHACIRA_BWD_REUSE_STAGED 2
SHACIRA_API int shacira_hashgrid_backward_levels(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                     int codebook_bitwidth, const int32_t *resolutions_host,
                                     const int32_t *codebook_first_idx, int64_t table_rows, const float *coords,
                                     const void *grad_output, int dtype, void *grad_codebook, int level_begin,
                                     int level_end, int flags, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The PLAN of a coordinate batch (round 6, ABI 10): forward and backward of one training step see the same coordinates
 * (the reference saves `coords` for the backward, wisp/ops/grid.py:86,106) -- what the forward of a large batch computes
 * first, the batch counting-sorted by spatial block (16-byte records {x, y, z, sample index} + block offsets), is exactly
 * what lets the backward accumulate its coarse levels block by block on chip instead of through the item stream. The two
 * calls below are shacira_hashgrid_forward / shacira_hashgrid_backward with one more CALLER-OWNED buffer: the forward
 * writes the batch's plan into it, the backward of the same (dim, num_coords, coords) reads it. Results are those of the
 * plain calls (forward: bit-identical; backward: the same products, summed in a different order).
 *
 *   shacira_hashgrid_plan_bytes      size of the plan buffer for this shape; 0 = the forward of this shape sorts nothing
 *                                    (small batches, cache-resident tables): pass plan = NULL, the calls are the plain ones
 *   plan                             device buffer of at least that size (256-byte aligned), or NULL. The forward overwrites
 *                                    it; the backward only reads it. A plan is a function of (dim, num_coords, coords): a
 *                                    caller that trains on a FIXED batch (the reference's image trainer revisits the same
 *                                    pixel lattice every step, wisp/trainers/image_trainer.py:234-266) may keep it across
 *                                    steps and pass plan_flags = SHACIRA_PLAN_READY to the forward, which then skips its sort
 *   plan_bytes                       size of the buffer; SHACIRA_EWORKSPACE when it is non-NULL and too small
 *   shacira_hashgrid_backward_planned_workspace_bytes (ABI 11)
 *                                    scratch the planned backward of this shape needs: the levels its brick pass takes write no
 *                                    items and the others 12-byte units, S1 at 2^20 samples: 0.70 GB against the 1.10 GB of
 *                                    shacira_hashgrid_backward_workspace_bytes (which such a call accepts, too). Holds for a
 *                                    16-byte aligned grad_output; a planned call on an unaligned one runs the plain passes
 *                                    and asks for the plain size (SHACIRA_EWORKSPACE below it). Equal to the plain size for
 *                                    shapes without a plan or outside the brick pass's rule.
 */
#define SHACIRA_PLAN_READY 1
SHACIRA_API size_t shacira_hashgrid_plan_bytes(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                               int codebook_bitwidth, const int32_t *resolutions_host, int64_t table_rows,
                                               int dtype);
SHACIRA_API size_t shacira_hashgrid_backward_planned_workspace_bytes(int dim, int64_t num_coords, int num_lods,
                                                                     int feature_dim, int codebook_bitwidth,
                                                                     const int32_t *resolutions_host, int64_t table_rows,
                                                                     int dtype);
SHACIRA_API int shacira_hashgrid_forward_planned(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                                 int codebook_bitwidth, const int32_t *resolutions_host,
                                                 const int32_t *codebook_first_idx, int64_t table_rows, const float *coords,
                                                 const void *codebook, int dtype, void *feats, void *plan, size_t plan_bytes,
                                                 int plan_flags, void *workspace, size_t workspace_bytes, void *stream);
SHACIRA_API int shacira_hashgrid_backward_planned(int dim, int64_t num_coords, int num_lods, int feature_dim,
                                                  int codebook_bitwidth, const int32_t *resolutions_host,
                                                  const int32_t *codebook_first_idx, int64_t table_rows, const float *coords,
                                                  const void *grad_output, int dtype, void *grad_codebook, const void *plan,
                                                  size_t plan_bytes, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Latent decode, deterministic (non-SGA) path of LatentDecoder.forward with num_layers_dec == 0
 * (basic_latent_decoder.py:192-198 with DecoderLayer.forward :86-91):
 *     q        = rint(latent)                       round-half-to-even, torch.round (StraightThrough :28-36)
 *     z[c]     = q[c] / div[c]
 *     y[j]     = (sum_c z[c] * matrix[c, j]) * colscale[j] + shift[j]
 *     decoded  = clamp(y, -clamp_weights, +clamp_weights) if clamp_weights > 0
 *   'sq'  decoders pass matrix = scale [latent_dim, feature_dim], colscale = NULL (treated as 1);
 *   'dft' decoders pass matrix = dft   [latent_dim, feature_dim], colscale = scale [feature_dim].
 *   shift may be NULL (use_shift False). All fp32.
 *
 * Backward (straight-through for the rounding): given grad_decoded [T, feature_dim]
 *     grad_latent[c]   = (sum_j gy[j] * colscale[j] * matrix[c, j]) / div[c]            (gy zeroed where clamped)
 *     grad_matrix[c,j] = sum_T z[c] * gy[j] * colscale[j]          ('sq': this is grad(scale))
 *     grad_colscale[j] = sum_T gy[j] * (sum_c z[c] * matrix[c, j]) ('dft': this is grad(scale))
 *     grad_shift[j]    = sum_T gy[j]
 *   Any of grad_latent / grad_matrix / grad_colscale / grad_shift may be NULL (not computed). Reductions over T
 *   are accumulated in fp64 partials held in `workspace` and written (not accumulated) to the outputs.
 */
SHACIRA_API int shacira_latent_decode_forward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                  const float *div, const float *matrix, const float *colscale,
                                  const float *shift, float clamp_weights, float *decoded, void *stream);

SHACIRA_API size_t shacira_latent_decode_backward_workspace_bytes(int64_t num_rows, int latent_dim, int feature_dim);

SHACIRA_API int shacira_latent_decode_backward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                   const float *div, const float *matrix, const float *colscale,
                                   const float *shift, float clamp_weights, const float *grad_decoded,
                                   float *grad_latent, float *grad_matrix, float *grad_colscale,
                                   float *grad_shift, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Entropy bits of the latents under the factorised BitEstimator (latent_grid.py:122-136,
 * bit_estimator.py:27-65):
 *     w     = latent + noise           (noise != NULL: training)   or   rint(latent)   (noise == NULL: is_val)
 *     p     = CDF(w + 0.5) - CDF(w - 0.5)
 *     bits  = clamp(-log(p + 1e-10) / ln 2, 0, 50)
 *     *total_bits = sum over all [num_rows, latent_dim] entries          (fp32 scalar on the device)
 *   CDF: for each of the first (num_layers-1) of {f1,f2,f3}: x = x*softplus(h)+b ; x = x + tanh(x)*tanh(a);
 *        then f4: sigmoid(x*softplus(h)+b).   params = fp32 [4][3][latent_dim] = (f1.h,f1.b,f1.a, f2..., f4.h,f4.b,unused)
 *
 * Backward, given the upstream scalar gradient d(loss)/d(total_bits) (device fp32 scalar):
 *     grad_latent [num_rows, latent_dim]   (zero where noise == NULL: round() has zero gradient; and where clamped)
 *     grad_params fp32 [4][3][latent_dim]  (unused slots written as 0)
 *   Either output may be NULL.
 */
SHACIRA_API size_t shacira_entropy_bits_workspace_bytes(int64_t num_rows, int latent_dim);

SHACIRA_API int shacira_entropy_bits_forward(int64_t num_rows, int latent_dim, int num_layers, const float *latent,
                                 const float *noise, const float *params, float *total_bits, void *workspace,
                                 size_t workspace_bytes, void *stream);

SHACIRA_API int shacira_entropy_bits_backward(int64_t num_rows, int latent_dim, int num_layers, const float *latent,
                                  const float *noise, const float *params, const float *grad_total_bits,
                                  float *grad_latent, float *grad_params, void *workspace, size_t workspace_bytes,
                                  void *stream);

/*
 * Fused decoder MLP (row a14): BasicDecoder.forward of the reference (wisp/models/decoders/basic_decoders.py:74-101)
 * as NeuralImage uses it (wisp/models/nefs/image.py:107-116, :152): `num_hidden` Linear(+bias)+ReLU layers of width
 * `hidden_dim`, then the linear `lout`; fp32.
 *   params / grad_params: one flat buffer  W1 [H, IN] (nn.Linear.weight layout), b1 [H], W2 [H, H], b2 [H], ...,
 *                         Wout [OUT, H], bout [OUT]
 *   x [num_rows, in_dim], y / grad_y [num_rows, out_dim], grad_x [num_rows, in_dim] (may be NULL)
 * Only a fixed set of shapes is compiled (shacira_mlp_supported); others return SHACIRA_EDTYPE and the caller keeps
 * using its own Linear layers. The backward recomputes the hidden activations (nothing is saved by the forward).
 */
SHACIRA_API int shacira_mlp_supported(int in_dim, int hidden_dim, int num_hidden, int out_dim);
SHACIRA_API size_t shacira_mlp_backward_workspace_bytes(int in_dim, int hidden_dim, int num_hidden, int out_dim);
SHACIRA_API int shacira_mlp_forward(int64_t num_rows, int in_dim, int hidden_dim, int num_hidden, int out_dim, const float *x,
                        const float *params, float *y, void *stream);
SHACIRA_API int shacira_mlp_backward(int64_t num_rows, int in_dim, int hidden_dim, int num_hidden, int out_dim, const float *x,
                         const float *params, const float *grad_y, float *grad_x, float *grad_params, void *workspace,
                         size_t workspace_bytes, void *stream);

/*
 * Fused Adam step over a flat fp32 buffer ("next" row f1: the optimizer step that follows the backward in the
 * reference's trainers, torch.optim.Adam built by wisp/trainers/base_trainer.py:206-266 and stepped at
 * wisp/trainers/image_trainer.py:355-359). torch.optim.Adam semantics with amsgrad=False, maximize=False:
 *     g' = grad + weight_decay*param ; m = beta1*m + (1-beta1)*g' ; v = beta2*v + (1-beta2)*g'^2
 *     param -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps)
 *   step counts from 1. zero_grad != 0 also clears `grad` in the same pass (saves the next step's memset).
 */
SHACIRA_API int shacira_adam_step(int64_t numel, float *param, float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, int zero_grad, void *stream);

/* Same, with the step count read from DEVICE memory (int32, >= 1) so that the launch can be captured into a
 * hipGraph and replayed while the count advances (the caller increments it on the stream). */
SHACIRA_API int shacira_adam_step_capturable(int64_t numel, float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                                 float lr, float beta1, float beta2, float eps, float weight_decay,
                                 const int32_t *step_dev, int zero_grad, void *stream);

/* Multi-tensor form: up to 32 parameters (HOST arrays of device pointers, sizes, per-tensor lr / weight decay) in
 * one launch. `step_dev` non-NULL: step count read from device memory (graph-capturable), else `step` is used. */
SHACIRA_API int shacira_adam_step_multi(int num_tensors, const int64_t *numel_host, float *const *param, float *const *grad,
                            float *const *exp_avg, float *const *exp_avg_sq, const float *lr_host,
                            const float *weight_decay_host, float beta1, float beta2, float eps, int step,
                            const int32_t *step_dev, int zero_grad, void *stream);

/*
 * Latent decode with stochastic Gumbel annealing (SGA) instead of rounding -- the `use_sga` branch of
 * LatentDecoder.forward (wisp/models/latent_decoders/basic_latent_decoder.py:183-191), the training mode of the
 * reference's shipped configs (kodak.yaml / nerf_lego.yaml: use_sga, diff_sampling) until `decay_period`:
 *     wf = floor(w), wc = wf + 1
 *     logits = -tanh(clamp(w - wf, +-(1 - 1e-6))) / T ,  -tanh(clamp(wc - w, +-(1 - 1e-6))) / T
 *     (s0, s1) = RelaxedOneHotCategorical(T, logits).rsample()      [diff_sampling]  /  .sample()  [otherwise]
 *              = softmax((logits + g) / T),  g = -log(-log(clamp(u, eps, 1 - eps)))
 *     q = wf * s0 + wc * s1 ;  then the decode of shacira_latent_decode_forward on q instead of round(w).
 * `uniforms` [num_rows, latent_dim, 2] fp32 in [0, 1) are supplied by the caller (torch.rand on the device: the one
 * draw the reference's sampler makes), so the operator is deterministic. Backward: grad_latent through rsample() when
 * diff_sampling (floor carries no gradient), through a straight-through floor otherwise; other outputs as in
 * shacira_latent_decode_backward.
 */
SHACIRA_API int shacira_latent_decode_sga_forward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                      const float *uniforms, float temperature, int diff_sampling, const float *div,
                                      const float *matrix, const float *colscale, const float *shift,
                                      float clamp_weights, float *decoded, void *stream);
SHACIRA_API int shacira_latent_decode_sga_backward(int64_t num_rows, int latent_dim, int feature_dim, const float *latent,
                                       const float *uniforms, float temperature, int diff_sampling, const float *div,
                                       const float *matrix, const float *colscale, const float *shift,
                                       float clamp_weights, const float *grad_decoded, float *grad_latent,
                                       float *grad_matrix, float *grad_colscale, float *grad_shift, void *workspace,
                                       size_t workspace_bytes, void *stream);
/* ABI 9: the same pair with the temperature in DEVICE memory (one float the kernels read): a training step captured into a
 * HIP graph anneals it between replays (base_trainer.py:155-157 decays it every iteration) without re-capturing. */
SHACIRA_API int shacira_latent_decode_sga_forward_tdev(int64_t num_rows, int latent_dim, int feature_dim,
                                       const float *latent, const float *uniforms, const float *temperature_dev,
                                       int diff_sampling, const float *div, const float *matrix, const float *colscale,
                                       const float *shift, float clamp_weights, float *decoded, void *stream);
SHACIRA_API int shacira_latent_decode_sga_backward_tdev(int64_t num_rows, int latent_dim, int feature_dim,
                                       const float *latent, const float *uniforms, const float *temperature_dev,
                                       int diff_sampling, const float *div, const float *matrix, const float *colscale,
                                       const float *shift, float clamp_weights, const float *grad_decoded,
                                       float *grad_latent, float *grad_matrix, float *grad_colscale, float *grad_shift,
                                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * Latent decoder WITH hidden layers / activations -- LatentDecoder with num_layers_dec > 0 and / or activation,
 * final_activation != 'none' (wisp/models/latent_decoders/basic_latent_decoder.py:97-198: the layer stack :139-147, forward
 * :182-198, DecoderLayer.forward :86-91, SineScaled(30.0) wisp/models/activations): a per-row MLP over the table,
 *     decoded = clamp(final_act(L_n(act(... act(L_1(q(latent) / div)) ...)))),     L_k(x) = x @ W_k + b_k,
 * q = round (straight-through) or, with uniforms != NULL, the SGA sample of the operators above. One pass each way.
 *   num_layers   hidden layers + 1, 1 .. SHACIRA_LATENT_MLP_MAX_LAYERS
 *   widths_host  HOST int32 [num_layers + 1]: latent_dim, hidden widths ..., feature_dim; each 1 .. SHACIRA_LATENT_MLP_MAX_WIDTH
 *   params       device fp32, packed per layer: W_k [widths[k], widths[k+1]] row-major (the layer's effective matrix: `scale`
 *                for 'sq', `dft * scale` for 'dft*'), then b_k [widths[k+1]] (`shift`; zeros when the layer has none)
 *   activation / final_activation   SHACIRA_ACT_* (the reference's act_dict keys)
 *   backward: grad_latent [num_rows, latent_dim] (NULL = skip), grad_params packed like params (the caller chains it to
 *   scale / shift); workspace of shacira_latent_mlp_backward_workspace_bytes() bytes (fp64 block partials: the table
 *   reductions are bitwise reproducible).
 * Returns SHACIRA_EINVAL for shapes outside the limits (the caller keeps its own per-layer evaluation for those).
 */
#define SHACIRA_LATENT_MLP_MAX_LAYERS 4
#define SHACIRA_LATENT_MLP_MAX_WIDTH 16
#define SHACIRA_ACT_NONE 0
#define SHACIRA_ACT_SIGMOID 1
#define SHACIRA_ACT_TANH 2
#define SHACIRA_ACT_RELU 3
#define SHACIRA_ACT_SINE30 4
SHACIRA_API int shacira_latent_mlp_supported(int num_layers, const int32_t *widths_host);
SHACIRA_API size_t shacira_latent_mlp_backward_workspace_bytes(int num_layers, const int32_t *widths_host);
SHACIRA_API int shacira_latent_mlp_forward(int64_t num_rows, int num_layers, const int32_t *widths_host, const float *latent,
                               const float *uniforms, float temperature, int diff_sampling, const float *div,
                               const float *params, int activation, int final_activation, float clamp_weights,
                               float *decoded, void *stream);
SHACIRA_API int shacira_latent_mlp_backward(int64_t num_rows, int num_layers, const int32_t *widths_host, const float *latent,
                                const float *uniforms, float temperature, int diff_sampling, const float *div,
                                const float *params, int activation, int final_activation, float clamp_weights,
                                const float *grad_decoded, float *grad_latent, float *grad_params, void *workspace,
                                size_t workspace_bytes, void *stream);

/*
 * Per-level latent decoders -- HierarchicalLatentDecoder (wisp/models/latent_decoders/hierarchical_latent_decoder.py:3-36,
 * built by LatentGrid.setup_decoders, wisp/models/grids/latent_grid.py:176-190): level l decodes the rows
 * [row_offsets[l], row_offsets[l+1]) of the table with ITS OWN div / matrix / colscale / shift; rounding or SGA
 * (uniforms != NULL) as in the single-decoder operators above. One launch for all levels.
 *   row_offsets_host  HOST int64 [num_levels + 1]; level starts ascend. The reference builds the last entry as the LAST
 *                     LEVEL'S SIZE, not the table's end (latent_grid.py:182): a boundary that falls before a level's
 *                     start makes that level empty. Rows no level owns decode to 0 (the reference leaves them
 *                     uninitialised) and receive a zero latent gradient.
 *   div [num_levels, latent_dim], matrix [num_levels, latent_dim, feature_dim], colscale / shift [num_levels,
 *   feature_dim] (NULL as above); the gradients of the backward have the same stacked shapes, written per level.
 */
SHACIRA_API int shacira_latent_decode_levels_forward(int num_levels, const int64_t *row_offsets_host, int64_t num_rows,
                                         int latent_dim, int feature_dim, const float *latent, const float *uniforms,
                                         float temperature, int diff_sampling, const float *div, const float *matrix,
                                         const float *colscale, const float *shift, float clamp_weights,
                                         float *decoded, void *stream);
SHACIRA_API int shacira_latent_decode_levels_backward(int num_levels, const int64_t *row_offsets_host, int64_t num_rows,
                                          int latent_dim, int feature_dim, const float *latent, const float *uniforms,
                                          float temperature, int diff_sampling, const float *div, const float *matrix,
                                          const float *colscale, const float *shift, float clamp_weights,
                                          const float *grad_decoded, float *grad_latent, float *grad_matrix,
                                          float *grad_colscale, float *grad_shift, void *workspace,
                                          size_t workspace_bytes, void *stream);

/*
 * MultiLatentDecoder (wisp/models/latent_decoders/multi_latent_decoder.py:27-210, built by LatentGrid.setup_decoders for
 * `ldecode_type: multi`), no hidden layers: num_decoders affine decoders mixed per table entry by the selector alpha:
 *     a_soft = softmax_k(alpha[k, r] / temperature) ;  a = one-hot(arg-max) if straight_through (gradient: identity) else a_soft
 *     x      = rint(latent[r]) / div            (uniforms == NULL)   or the SGA sample with the same temperature
 *     'dft' (dft != NULL): per_k = (x @ dft) * scale[k, 0, :] + shift[k, :] ;  decoded = sum_k per_k * a_k
 *     'sq'  (dft == NULL): mixed = sum_k (x @ scale[k]) * a_k ;  decoded = sum_k (mixed + shift[k, :]) * a_k
 *       (the reference's forward mixes twice, multi_latent_decoder.py:74-81; kept)
 *   alpha [num_decoders, num_rows]; scale [num_decoders, S, feature_dim] with S = latent_dim ('sq') or 1 ('dft');
 *   dft [latent_dim, feature_dim]; shift [num_decoders, feature_dim] or NULL; clamp as above.
 * Backward writes grad_latent [num_rows, latent_dim] (straight-through rounding / SGA), grad_alpha [num_decoders, num_rows]
 * (through the softmax), grad_scale, grad_shift (NULL = skip, except grad_scale). num_decoders <= 8 and the
 * (latent_dim, feature_dim) pairs of shacira_latent_multi_supported; else SHACIRA_EDTYPE.
 */
SHACIRA_API int shacira_latent_multi_supported(int latent_dim, int feature_dim, int num_decoders);
SHACIRA_API int shacira_latent_multi_decode_forward(int64_t num_rows, int latent_dim, int feature_dim, int num_decoders,
                                        const float *latent, const float *alpha, const float *uniforms,
                                        float temperature, int straight_through, int diff_sampling, const float *div,
                                        const float *scale, const float *dft, const float *shift, float clamp_weights,
                                        float *decoded, void *stream);
SHACIRA_API int shacira_latent_multi_decode_backward(int64_t num_rows, int latent_dim, int feature_dim, int num_decoders,
                                         const float *latent, const float *alpha, const float *uniforms,
                                         float temperature, int straight_through, int diff_sampling, const float *div,
                                         const float *scale, const float *dft, const float *shift, float clamp_weights,
                                         const float *grad_decoded, float *grad_latent, float *grad_alpha,
                                         float *grad_scale, float *grad_shift, void *workspace, size_t workspace_bytes,
                                         void *stream);

/*
 * Symbol statistics and entropy coding of the rounded latents -- replaces the per-channel
 * `torch.round(...).long()` + `torch.unique(return_counts=True)` of LatentGrid.size
 * (wisp/models/grids/latent_grid.py:141-143) and the torchac.encode_float_cdf call (:155-172).
 *
 *   shacira_latent_symbol_range      minmax[c] = {min, max} over rows of (int)rint(latent[r, c])   (device int32 [ld][2];
 *                                    {INT32_MAX, INT32_MIN} when num_rows == 0). rint = round-half-even = torch.round.
 *   shacira_latent_symbol_histogram  counts[c][k] = #rows with rint(latent[r, c]) == minmax[c][0] + k, 0 <= k < nbins
 *                                    (device uint64 [ld][nbins], overwritten). `minmax` is the device array written by
 *                                    shacira_latent_symbol_range; nbins >= max_c (max - min + 1).
 *   latent_dim <= 16, else SHACIRA_EDTYPE.
 *
 * Range coder (HOST buffers, runs on the calling thread; the reference codes on the CPU as well):
 *   static model = freq_host[num_symbols] with sum exactly 65536 and freq >= 1 for every symbol that occurs.
 *   shacira_rc_encode  symbols (indices into the model) -> bytes; *out_len = bytes produced; SHACIRA_EWORKSPACE if
 *                      `capacity` < *out_len (nothing valid written); shacira_rc_encode_bound(n) is always enough.
 *   shacira_rc_decode  exact inverse: reproduces the n symbols.
 */
SHACIRA_API int shacira_latent_symbol_range(int64_t num_rows, int latent_dim, const float *latent, int32_t *minmax,
                                void *stream);
SHACIRA_API int shacira_latent_symbol_histogram(int64_t num_rows, int latent_dim, const float *latent, const int32_t *minmax,
                                    int nbins, uint64_t *counts, void *stream);
SHACIRA_API size_t shacira_rc_encode_bound(int64_t num_symbols_to_code);
SHACIRA_API int shacira_rc_encode(const int32_t *symbols_host, int64_t n, const uint32_t *freq_host, int num_symbols,
                      uint8_t *out_host, size_t capacity, size_t *out_len);
SHACIRA_API int shacira_rc_decode(const uint8_t *in_host, size_t len, const uint32_t *freq_host, int num_symbols, int64_t n,
                      int32_t *symbols_host);

/*
 * Volume integration over variable-length sample packs and sample generation on a dense occupancy grid -- the steps
 * either side of the hash-grid lookup in the NeRF pipeline. The reference runs them through kaolin 0.13 (un-vendored):
 * `spc_render.exponential_integration` / `sum_reduce` (call sites wisp/tracers/packed_rf_tracer.py:131-151),
 * `OctreeAS._raymarch_ray` / `_raymarch_voxel` (wisp/accelstructs/octree_as.py:171-290) on `unbatched_query` /
 * `unbatched_raytrace`.
 *
 * Packs: the samples of ray r are rows [pack_start[r], pack_start[r+1]) (device int64 [num_packs + 1], ascending).
 *   shacira_pack_integrate_forward   weights[i] = exp(-sum_{j<i in pack} tau[j]) * (1 - exp(-tau[i]))
 *                                    ray_feats[r, c] = sum_i weights[i] * feats[i, c]
 *                                    (= exponential_integration(feats, tau, boundary, exclusive=True))
 *   shacira_pack_integrate_backward  gradients of both outputs w.r.t. feats and tau; grad_weights may be NULL (zero)
 *   shacira_pack_sum                 out[r, c] = sum_i x[i, c]                  (= sum_reduce)
 *   shacira_pack_broadcast           out[i, c] = per_pack[r(i), c]              (its gradient)
 *   channels <= 16, else SHACIRA_EDTYPE.
 *
 * Occupancy: uint8 [G][G][G], G = 2^level, indexed [x][y][z]; a point p in [-1,1]^3 lies in cell
 * floor(clamp(G*(p+1)/2, 0, G-1)) (kaolin quantize_points). Both generators are two-pass (count, host/device scan of
 * the counts by the caller, emit at offsets [num_rays + 1]).
 *   shacira_raymarch_ray_{count,emit}    `num_samples` stratified depths per ray:
 *                                        depth = (lin[j] + jitter[r, j] / num_samples) * (dist_max - dist_min) + dist_min
 *                                        (lin = torch.linspace(0, 1, num_samples), jitter ~ U[0,1) supplied by the caller),
 *                                        kept when the cell of origin + dir * depth is occupied. emit writes, in order,
 *                                        ridx int64, samples [.,3], depth, deltas (depth - previous depth of the ray,
 *                                        dist_min before the first), boundary uint8 (1 at a ray's first kept sample).
 *   shacira_raytrace_dense_{count,emit}  every (ray, occupied cell) crossing in depth order: ridx int32, pidx int32
 *                                        (Morton index of the cell, x most significant), depth [., 2] = entry (>= 0), exit.
 */
SHACIRA_API int shacira_pack_integrate_forward(int64_t num_samples, int64_t num_packs, int channels, const float *feats,
                                   const float *tau, const int64_t *pack_start, float *ray_feats, float *weights,
                                   void *stream);
SHACIRA_API int shacira_pack_integrate_backward(int64_t num_samples, int64_t num_packs, int channels, const float *feats,
                                    const float *tau, const int64_t *pack_start, const float *grad_ray_feats,
                                    const float *grad_weights, float *grad_feats, float *grad_tau, void *stream);
SHACIRA_API int shacira_pack_sum(int64_t num_samples, int64_t num_packs, int channels, const float *x,
                     const int64_t *pack_start, float *out, void *stream);
SHACIRA_API int shacira_pack_broadcast(int64_t num_samples, int64_t num_packs, int channels, const float *per_pack,
                           const int64_t *pack_start, float *out, void *stream);
SHACIRA_API int shacira_raymarch_ray_count(int64_t num_rays, int num_samples, const float *origins, const float *dirs,
                               float dist_min, float dist_max, const float *lin, const float *jitter,
                               const uint8_t *occupancy, int level, int32_t *counts, void *stream);
SHACIRA_API int shacira_raymarch_ray_emit(int64_t num_rays, int num_samples, const float *origins, const float *dirs,
                              float dist_min, float dist_max, const float *lin, const float *jitter,
                              const uint8_t *occupancy, int level, const int64_t *offsets, int64_t *ridx,
                              float *samples, float *depth, float *deltas, uint8_t *boundary, void *stream);
/* ABI 9: the same into caller buffers of `capacity` rows -- survivors whose row would be >= capacity are dropped (the
 * caller clamps `offsets` to capacity for its pack reductions). For steps captured into a HIP graph: the buffers cannot
 * be sized from a count read back to the host there. Rows behind the last survivor are left untouched. */
SHACIRA_API int shacira_raymarch_ray_emit_capped(int64_t num_rays, int num_samples, const float *origins,
                              const float *dirs, float dist_min, float dist_max, const float *lin, const float *jitter,
                              const uint8_t *occupancy, int level, const int64_t *offsets, int64_t capacity,
                              int64_t *ridx, float *samples, float *depth, float *deltas, uint8_t *boundary,
                              void *stream);
SHACIRA_API int shacira_raytrace_dense_count(int64_t num_rays, const float *origins, const float *dirs,
                                 const uint8_t *occupancy, int level, int32_t *counts, void *stream);
SHACIRA_API int shacira_raytrace_dense_emit(int64_t num_rays, const float *origins, const float *dirs,
                                const uint8_t *occupancy, int level, const int64_t *offsets, int32_t *ridx,
                                int32_t *pidx, float *depth, void *stream);

/*
 * Diagnostics: one streaming pass over `bytes` (a multiple of 16; 16-byte aligned buffers) with the access shape of the
 * hash-grid kernels -- 16 bytes per lane, 8 in flight, non-temporal. kind 0 = read `src` (`dst` may be NULL or a 4-byte
 * scratch word), 1 = write `dst`, 2 = copy `src` -> `dst`. bench.py times it for `roofline.measured_stream_rates`.
 */
SHACIRA_API int shacira_stream_probe(int kind, const void *src, void *dst, size_t bytes, void *stream);

/*
 * Tunables (for benchmarking and A/B only). Process-wide atomics; every entry point takes ONE snapshot of them when it is
 * entered and uses that for the whole call (its workspace-size check included), so changing an option from another thread
 * never makes a running call inconsistent -- it applies to calls entered later. Unknown names / values -> SHACIRA_EINVAL.
 *   "fwd_variant": -1 (default) = measured rule; 0 = one gather per corner, the reference's kernel shape (any even F);
 *               3 = lane pairs, sample-major; 6 = one level per XCD + level-major staging; 8 = cell-sorted forward;
 *               9 = tables whose levels fit LDS images (groups of levels resident in LDS; refused silently, i.e. the rule
 *               applies, when a level does not fit).
 *   "bwd_variant": -1 (default) = by batch size; 0 = scattered atomics (the reference's design); 1 = binned.
 *   "bin_batch_mib": cap (MiB) of the backward's item array; larger batches are processed in sub-batches.
 *   "bin_acc_kib": LDS accumulator image per consumer workgroup: 64, 128, or 0 = chosen from the batch size (default).
 *   "bwd_compact": 1 (default) = dense levels travel as one item per sample (3-D: z-slab buckets, two slots; 2-D: line-slab
 *               buckets, one slot), 0 = pair items.
 *   "mlp_variant": -1 (default) = decoder MLPs on the fp32 matrix cores wherever instantiated, 0 = VALU kernels.
 *   "tiled": -1 (default) = the cell-sorted forward (hashgrid_tiled.hip) for batches where it measured faster (tables
 *            > 8 MB; 3-D: from 2^18 samples for F = 2, 80 K for F = 4; 2-D: from 192 K), 0 = never, 1 = whenever the shape
 *            allows it. "fwd_variant" 8 also forces it; other explicit variants exclude it.
 *   "tiled_lc_fwd": its number of coarse levels, -1 (default) = planner's choice.
 *   "bwd_fork": 1 (default) = large batches with LDS-resident ("direct") levels zero the table and accumulate those levels
 *               on a library-owned side stream, forked from and joined back into the caller's stream with events (stream
 *               semantics unchanged, capturable); 0 = one stream.
 *   "bwd_persistent": 1 (default) = the backward's last pass runs as persistent workgroups that fetch their work units from a
 *               counter (batches >= 2^17 samples); 0 = one workgroup per unit.
 *   "bwd_selective_zero": 1 (default) = the backward zeroes only the gradient rows its last pass does not overwrite with
 *               plain stores (all rows outside the hashed levels, plus hashed buckets that received 0 or several work
 *               units); 0 = the whole table first. Same result either way.
 * (ABI 7 removed "bwd_fuse", "bwd_groups", "bwd_rows", "bwd_direct_side" and forward variants 1, 2, 4, 5, 7 -- code paths
 * that measured slower in rounds 1-2; they are recorded by git hash in profiles/.)
 */
SHACIRA_API int shacira_set_option(const char *name, int value);
SHACIRA_API int shacira_get_option(const char *name);

#ifdef __cplusplus
}
#endif
#endif /* SHACIRA_HIP_H */
