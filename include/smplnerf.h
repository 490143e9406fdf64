/* smplnerf.h - C-ABI of libsmplnerf_hip.so: the MI355X (gfx950) NeRF ray-march path.
 *
 * Drop-in boundary for the hot path of HannesStark/SMPL-NeRF (citations are reference file:line):
 *
 *   snerf_searchsorted_f32   replaces  torchsearchsorted.searchsorted
 *                            (torchsearchsorted/src/torchsearchsorted/searchsorted.py:20-53; native
 *                            searchsorted_cpu_wrapper.cpp:82-122, searchsorted_cuda_kernel.cu:84-141)
 *   snerf_posenc_f32         replaces  PositionalEncoder.encode            (utils.py:114-131)
 *   snerf_composite_fwd_f32  replaces  raw2outputs                         (utils.py:134-191)
 *   snerf_sample_pdf_f32     replaces  sample_pdf + fine_sampling          (utils.py:194-264)
 *   snerf_mlp_*              replaces  RenderRayNet.forward                (models/render_ray_net.py:42-61)
 *                            fused with the positional encoding of its inputs as used by
 *                            NerfPipeline.forward                          (models/nerf_pipeline.py:29-41, :49-60)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; buffers are borrowed
 *     for the duration of the call only; outputs are caller-allocated; nothing is retained.
 *   - `stream` is a hipStream_t (pass PyTorch's current stream); all work is enqueued
 *     asynchronously on it, no call synchronises the device.
 *   - return value: 0 on success, <0 = SNERF_E_*; nothing throws across the ABI.
 *     snerf_last_error_string() describes the last failure on the calling thread.
 *   - row-major, densely packed fp32 tensors; indices are int64 (torch.long) like the reference.
 *   - re-entrant and thread-safe; calls act on the calling thread's current HIP device.
 *
 * State.  The library keeps no caller-visible state between calls.  What it does keep, process-wide:
 *   - the error text of the last failure, per thread (snerf_last_error_string);
 *   - two HIP events per host thread and device, created at the first training call that is given an auxiliary stream
 *     (fork / join of the concurrent backward; snerf_shutdown() destroys them);
 *   - per HIP device ordinal (up to 64 devices): the CU count and, per kernel, whether its dynamic-LDS limit was
 *     raised (hipFuncSetAttribute is per device) - so one process may drive several GPUs through the library;
 *   - environment variables, read ONCE at the first call that consults them: SNERF_LAT, SNERF_MLP_FOLD, SNERF_WARP_FOLD,
 *     SNERF_RCCL_LIB and the debugging aid SNERF_DEBUG_POISON_LDS.  INTEGRATION.md ("Environment variables") describes them and
 *     the Python host's; a release build can ignore all of them (results are identical under every setting, up to the summation
 *     order of the folded per-ray columns).
 */
#ifndef SMPLNERF_H
#define SMPLNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNERF_VERSION 109 /* 0.1.2: + snerf_searchsorted (all scalar types), snerf_posenc_bwd_f32,
                             snerf_composite_bwd_all_f32; composite forward accepts any N
                             0.1.3: same entry points; descriptors accept any width <= 256 and n_layers >= 1; fp32 inference
                             folds per-ray inputs (dirs_per_sample bit 1 = SNERF_FWD_NO_RAY_FOLD keeps the per-sample form)
                             0.1.4: + the training step as one call (snerf_nerf_train_step_f32 / _grads_f32, snerf_adam_step_f32,
                             snerf_mlp_stream_slots), snerf_dy_contract_f32; the per-ray fold tables moved from stream-ordered allocations inside
                             the library to caller workspaces (snerf_mlp_fwd_ws_f32, snerf_warp_fwd_ws_f32): the library allocates nothing
                             0.1.5: + snerf_render_rays_add_f32 (the single-call render for nets with per-ray additional inputs)
                             0.1.6: + snerf_shutdown; hidden visibility: the entry points of this header are the only dynamic symbols
                             0.1.7: + snerf_comm_*, snerf_nerf_train_step_dp_f32, snerf_smpl_nerf_train_step_dp_f32 (RCCL inside the
                             boundary); latency-class kernels behind the same entry points for small calls
                             0.1.8: + snerf_smpl_nerf_train_grads_aux_f32, snerf_smpl_nerf_train_step_aux_f32 (the smpl_nerf step with an
                             auxiliary stream: small chunks run the coarse chain beside the fine chain); snerf_mlp_desc.width
                             up to 512
                             0.1.9: + snerf_nerf_train_step_dp_ig_f32; the data-parallel steps issue the same collectives on every
                             rank whatever its batch size (B == 0 included); SNERF_RCCL_LIB; snerf_linear_* / snerf_relu_bwd_f32 (any --netwidth) */

#define SNERF_OK 0
#define SNERF_E_BADARG (-1)   /* null pointer, negative size, unsupported shape */
#define SNERF_E_ALIGN (-2)    /* pointer not aligned as documented */
#define SNERF_E_LAUNCH (-3)   /* HIP launch / runtime error */
#define SNERF_E_NODEVICE (-4) /* no gfx950 device visible */

typedef void *snerf_stream_t; /* hipStream_t */

/* The library is built with -fvisibility=hidden: the entry points declared here are its only dynamic symbols
 * (tests/test_abi.py: `nm -D --defined-only` lists nothing but snerf_*). */
#if defined(__GNUC__) || defined(__clang__)
#define SNERF_API __attribute__((visibility("default")))
#else
#define SNERF_API
#endif

SNERF_API int snerf_version(void);
SNERF_API const char *snerf_last_error_string(void);
/* Number of visible HIP devices (0 when none); never fails. */
SNERF_API int snerf_device_count(void);
/* Releases what the library holds process-wide (section "State"): the fork / join events of every host thread and device.
 * Optional - a process may simply exit; for hosts that unload the library or count HIP objects.  Must not run concurrently
 * with another call into the library; the next training call creates its events again.  Returns SNERF_OK. */
SNERF_API int snerf_shutdown(void);

/* ---- a6: batched searchsorted --------------------------------------------------------------
 * out[r, c] = #{ k : a[ra, k] <  v[rv, c] }   (side_left != 0, numpy side='left')
 *           = #{ k : a[ra, k] <= v[rv, c] }   (side_left == 0, side='right')
 * rows of `a` must be sorted ascending.  ra = 0 if nrow_a == 1 else r (same for rv): the
 * reference's row broadcast (searchsorted_cpu_wrapper.cpp:109-110).  nrow_a and nrow_v must be
 * equal or one of them 1 (searchsorted.py:23-28).  out: int64 [max(nrow_a,nrow_v), ncol_v]. */
SNERF_API int snerf_searchsorted_f32(const float *a, int64_t nrow_a, int64_t ncol_a,
                           const float *v, int64_t nrow_v, int64_t ncol_v,
                           int64_t *out, int side_left, snerf_stream_t stream);

/* The same for every scalar type the reference dispatches (AT_DISPATCH_ALL_TYPES, searchsorted_cpu_wrapper.cpp:100,
 * searchsorted_cuda_kernel.cu:132); `a` and `v` have the same element type `dtype`. */
#define SNERF_DTYPE_F32 0
#define SNERF_DTYPE_F64 1
#define SNERF_DTYPE_I32 2
#define SNERF_DTYPE_I64 3
#define SNERF_DTYPE_I16 4
#define SNERF_DTYPE_I8 5
#define SNERF_DTYPE_U8 6
SNERF_API int snerf_searchsorted(int dtype, const void *a, int64_t nrow_a, int64_t ncol_a, const void *v, int64_t nrow_v,
                       int64_t ncol_v, int64_t *out, int side_left, snerf_stream_t stream);

/* ---- a1: positional encoding ----------------------------------------------------------------
 * x [n, c] -> out [n, c*(identity + 2*L)], frequency-major: [x] [sin(2^0 x) cos(2^0 x)] ...
 * (utils.py:116-131; no pi factor). */
SNERF_API int snerf_posenc_f32(const float *x, int64_t n, int c, int L, int identity, float *out,
                     snerf_stream_t stream);

/* Backward of the encoding: d_out [n, c*(identity + 2*L)] -> d_x [n, c] (autograd through utils.py:123-131). */
SNERF_API int snerf_posenc_bwd_f32(const float *x, const float *d_out, int64_t n, int c, int L, int identity, float *d_x,
                         snerf_stream_t stream);

/* ---- a4: alpha compositing --------------------------------------------------------------------
 * raw [B, N, 4] (r, g, b, sigma), z [B, N], dirs: [B, 3] when dirs_per_sample == 0 (the ray
 * direction broadcast over samples) or [B, N, 3]; dists are scaled by ||dirs|| (utils.py:165).
 * noise: nullable [B, N], added to sigma before relu (utils.py:171-173; the caller draws it).
 * rgb [B, 3], weights [B, N], alpha [B, N] (any of the three may be NULL to skip the store).
 * N == 1 reproduces the reference's early return (utils.py:168-169): weights = alpha = 1.  Any N >= 1 (the backward
 * entry points take N <= 4096). */
SNERF_API int snerf_composite_fwd_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                            const float *noise, int64_t B, int N, int white_background,
                            float *rgb, float *weights, float *alpha, snerf_stream_t stream);

/* Backward of the compositing w.r.t. raw: d_rgb [B,3] -> d_raw [B,N,4] (same raw/z/dirs/noise as the
 * forward call; autograd through utils.py:161-191).  weights/alpha are treated as outputs without
 * gradient (the pipeline detaches what it derives from them, utils.py:260).  N <= 4096.  A per-sample direction of
 * norm 0 gets d_dirs = 0 (torch.norm's subgradient). */
SNERF_API int snerf_composite_bwd_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                            const float *noise, int64_t B, int N, int white_background,
                            const float *d_rgb, float *d_raw, float *d_dirs /* nullable, [B,N,3]: only with per-sample
                            directions (dists are scaled by their norm, utils.py:165) */, snerf_stream_t stream);

/* The complete backward of raw2outputs (utils.py:134-191 under autograd): gradients arriving at ALL three outputs -
 * d_rgb [B,3], d_weights [B,N], d_alpha [B,N], each nullable (= zeros) - are propagated to d_raw [B,N,4] and, on request,
 * to the directions (d_dirs: [B,N,3] with per-sample directions, [B,3] otherwise; dists are scaled by their norm,
 * utils.py:165) and to the depths (d_z [B,N]; the last interval is the constant 1e10, utils.py:164).  N <= 4096.  N == 1:
 * weights and alpha are constants, d_dirs = d_z = 0. */
SNERF_API int snerf_composite_bwd_all_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                                const float *noise, int64_t B, int N, int white_background, const float *d_rgb,
                                const float *d_weights, const float *d_alpha, float *d_raw, float *d_dirs /* nullable */,
                                float *d_z /* nullable */, snerf_stream_t stream);

/* ---- a5: inverse-CDF hierarchical sampling + merge + point generation ------------------------------
 * z [B, Nc] coarse depths (ascending), weights [B, Nc] from compositing, u [Nf] = linspace(0,1,Nf)
 * (passed in so that the caller controls its bits, see oracle/nerf_oracle.py:linspace01),
 * o, d [B, 3].  Outputs (each nullable): inds int64 [B, Nf] (searchsorted(cdf,u,'right')),
 * z_samples [B, Nf], z_fine [B, Nc+Nf] ascending, pts [B, Nc+Nf, 3] = o + d * z_fine.
 * 3 <= Nc <= 1024, 1 <= Nf <= 1024 (with fewer than three coarse samples there is no interior weight: the reference's own
 * sample_pdf builds an empty cdf and its gather raises, utils.py:200-221). */
SNERF_API int snerf_sample_pdf_f32(const float *z, const float *weights, const float *u,
                         const float *o, const float *d, int64_t B, int Nc, int Nf,
                         int64_t *inds, float *z_samples, float *z_fine, float *pts,
                         snerf_stream_t stream);

/* Strict variants (SURVEY 8b `strict_cumsum`): index parity with the reference FROM THE SAME WEIGHTS.  The reference's
 * pdf is weights / torch.sum(weights) (utils.py:201); torch's CPU sum is a vectorised fp32 cascade whose bits depend on the
 * host's SIMD width, and a 1-ulp difference in it moves ~0.15 % of the indices.  `tot` [B] = that sum as the reference's
 * host evaluated it (e.g. (weights[:, 1:-1] + 1e-5).sum(-1) with torch on the CPU); every other step is independent of
 * the evaluation order, so cdf, inds and the samples are then bit-identical to the reference's. */
SNERF_API int snerf_sample_pdf_strict_f32(const float *z, const float *weights, const float *u, const float *o, const float *d,
                                const float *tot, int64_t B, int Nc, int Nf, int64_t *inds, float *z_samples,
                                float *z_fine, float *pts, snerf_stream_t stream);
SNERF_API int snerf_sample_pdf_bins_strict_f32(const float *bins, const float *weights, const float *u, const float *tot, int64_t B,
                                     int Nb, int Nf, int64_t *inds, float *z_samples, snerf_stream_t stream);

/* The literal sample_pdf(bins, weights, args) convention (utils.py:194-228): bins [B, Nb] (coarse
 * midpoints), weights [B, Nb-1] (interior coarse weights) -> inds int64 [B, Nf] (nullable),
 * z_samples [B, Nf].  2 <= Nb <= 1023. */
SNERF_API int snerf_sample_pdf_bins_f32(const float *bins, const float *weights, const float *u, int64_t B,
                              int Nb, int Nf, int64_t *inds, float *z_samples, snerf_stream_t stream);

/* Backward of sample_pdf(bins, weights, args) (utils.py:194-228 under autograd; fine_sampling detaches its result,
 * utils.py:260, so no pipeline needs it): d_z_samples [B, Nf] -> d_bins [B, Nb], d_weights [B, Nb-1], with the forward's
 * searchsorted indices `inds` [B, Nf] held fixed (integers carry no gradient in the reference either). */
SNERF_API int snerf_sample_pdf_bins_bwd_f32(const float *bins, const float *weights, const float *u, const int64_t *inds,
                                  const float *tot /* nullable [B]: the strict forward's normalising sums */,
                                  const float *d_z_samples, int64_t B, int Nb, int Nf, float *d_bins, float *d_weights,
                                  snerf_stream_t stream);

/* ---- a2: RenderRayNet -----------------------------------------------------------------------------
 * Mirrors RenderRayNet.__init__ (models/render_ray_net.py:8): n_layers, width, skips as a bit mask
 * over positional_net indices, use_directional_input; positions_dim / directions_dim are expressed
 * through the encoders that feed them (pos_freqs/pos_identity, dir_freqs/dir_identity; 3 input
 * channels each), additional_input_dim = add_dim per-ray constants appended to the position
 * encoding (train.py:154-159). */
typedef struct snerf_mlp_desc {
    int32_t n_layers;     /* 8 */
    int32_t width;        /* 2 .. 512 (--netwidth, config_parser.py:20).  The kernels are built for trunks of 64, 128, 256 and
                             (0.1.8; one wave per SIMD) 320, 384, 448, 512 features; any other width runs zero-padded inside
                             the next larger one (same results, the cost of that kernel).  The split-precision entry points
                             take 256 only; a net above 256 needs at least 17 input columns in its first layer. */
    int32_t pos_freqs;    /* 10 */
    int32_t pos_identity; /* 0 */
    int32_t dir_freqs;    /* 4 */
    int32_t dir_identity; /* 0 */
    int32_t add_dim;      /* 0 */
    uint32_t skip_mask;   /* bit i set <=> i in skips */
    int32_t use_dir;      /* 1 */
    int32_t add_first;    /* 0: input columns [PE(x) | add] (RenderRayNet(additional_input_dim) fed by train.py:154-159
                             style rows [samples_encoding | extra]); 1: [add | PE(x)], the order the append_smpl_params /
                             append_to_nerf pipelines build (models/append_smpl_params_pipeline.py:49-51) */
} snerf_mlp_desc;

/* Number of floats in the flat parameter vector: weights and biases in state_dict order
 * (positions_pose_input.weight, .bias, positional_net.0.weight, ... rgb_out_layer.bias). */
SNERF_API int64_t snerf_mlp_param_floats(const snerf_mlp_desc *desc);
/* Number of floats of the MFMA-ordered weight stream produced by snerf_mlp_pack_f32. */
SNERF_API int64_t snerf_mlp_packed_floats(const snerf_mlp_desc *desc);
/* params_flat -> packed (both device).  Run once per weight update. */
SNERF_API int snerf_mlp_pack_f32(const snerf_mlp_desc *desc, const float *params_flat, float *packed,
                       snerf_stream_t stream);

/* Fused positional encoding + MLP.  x [n, 3] sample positions; directions dirs: [n/samples_per_ray, 3]
 * when dirs_per_sample == 0 (one per ray, samples of a ray contiguous) or [n, 3]; directions are
 * normalised inside (models/nerf_pipeline.py:33-34).  add: nullable [n/samples_per_ray, add_dim].
 * raw [n, 4] = [rgb | sigma] (models/render_ray_net.py:61).
 * dirs_per_sample is a bit set (other bits: SNERF_E_BADARG): bit 0 = directions per sample; bit 1 (SNERF_FWD_NO_RAY_FOLD,
 * snerf_mlp_fwd_ws_f32 only) = multiply the additional-input columns per sample even when a fold workspace is given, in the
 * order of snerf_mlp_fwd_train_f32 (whose `raw` the call then reproduces bit for bit). */
#define SNERF_FWD_DIRS_PER_SAMPLE 1
#define SNERF_FWD_NO_RAY_FOLD 2
SNERF_API int snerf_mlp_fwd_f32(const snerf_mlp_desc *desc, const float *packed, const float *x,
                      const float *dirs, int dirs_per_sample, const float *add,
                      int64_t n, int samples_per_ray, float *raw, snerf_stream_t stream);

/* The same call with a caller-allocated workspace for the per-ray fold: nets with additional inputs (add_dim > 0) read them
 * as per-RAY constants, so W_add . add is evaluated once per ray into a table of snerf_mlp_fold_workspace_bytes(desc, n,
 * samples_per_ray) bytes (0 when the fold does not apply: no additional inputs, fewer than 8 samples per ray, ...) and added to
 * the accumulators; their k-blocks are skipped (same values up to the summation order of those columns).  workspace NULL
 * (= snerf_mlp_fwd_f32): the per-sample form.  A workspace smaller than needed is SNERF_E_BADARG - the library allocates
 * nothing and never switches form silently.  16-byte aligned. */
SNERF_API int64_t snerf_mlp_fold_workspace_bytes(const snerf_mlp_desc *desc, int64_t n, int samples_per_ray);
SNERF_API int snerf_mlp_fwd_ws_f32(const snerf_mlp_desc *desc, const float *packed, const float *x, const float *dirs,
                         int dirs_per_sample, const float *add, int64_t n, int samples_per_ray, float *raw, void *workspace,
                         int64_t workspace_bytes, snerf_stream_t stream);

/* ---- a2 on the bf16 matrix cores with fp32-class accuracy (split-bf16, inference) -------------------------
 * Every fp32 operand is split into nsplit bf16 parts and the cross terms are accumulated in fp32:
 * nsplit = 3 (6 products, ~2^-24 relative: same parity class as the fp32 kernel, 2.7x less matrix-pipe time),
 * nsplit = 2 (3 products, ~2^-16 relative: RGB within ~8e-5, 5.3x less).  Same inputs/outputs as
 * snerf_mlp_fwd_f32; `packed` comes from snerf_mlp_pack_bf16 with the same nsplit.  Width 256 only.
 * nsplit = SNERF_SPLIT_F16X3 selects two fp16 parts instead (3 products, ~2^-22 relative: raw outputs within the fp32
 * kernel's own tolerance, at the speed of nsplit = 2): operands are scaled by exact powers of two - weights per layer at
 * pack time, activations (and, in the backward, gradients) per sample in the kernel - so that fp16's range is never
 * left.  Accepted wherever an nsplit / precision is: forward, training forward, snerf_render_rays*, and the backward
 * entry points (dgrad with per-sample scales; the wide wgrad GEMMs, which contract over samples, with per-layer scales
 * taken from the exponents the f16x3 training forward leaves behind the rows of `act` and the f16x3 dgrad behind those
 * of `dy`: an f16x3 backward therefore needs the `act` of an f16x3 training forward). */
#define SNERF_SPLIT_F16X3 16
SNERF_API int64_t snerf_mlp_packed_bf16_bytes(const snerf_mlp_desc *desc, int nsplit);
SNERF_API int snerf_mlp_pack_bf16(const snerf_mlp_desc *desc, const float *params_flat, void *packed, int nsplit,
                        snerf_stream_t stream);
SNERF_API int snerf_mlp_fwd_bf16_f32(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x,
                           const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                           int samples_per_ray, float *raw, snerf_stream_t stream);
/* The same forward for training: additionally saves every layer input into `act`, in exactly the layout
 * snerf_mlp_fwd_train_f32 writes (act_floats of snerf_mlp_train_sizes), so that snerf_mlp_bwd_f32 /
 * snerf_mlp_bwd_inputs_f32 run on it unchanged (any mix of fp32 / split-bf16 forward and backward works: the buffers are
 * fp32 - except that the f16x3 backward needs the f16x3 forward, see SNERF_SPLIT_F16X3). */
SNERF_API int snerf_mlp_fwd_train_bf16_f32(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x,
                                 const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                                 int samples_per_ray, float *raw, float *act, snerf_stream_t stream);

/* The backward on the bf16 matrix cores: dgrad (the transposed network) with split-bf16 operands, fp32 accumulate
 * and fp32 stored d Y, and the wide wgrad GEMMs (the 256x256 and 128x256 layers) with both operands split the same
 * way; the narrow wgrad jobs and the reduction are exact fp32 (SNERF_WGRAD_BF16=0: all of wgrad in fp32).  Same buffers as snerf_mlp_bwd_f32 / _bwd_inputs_f32
 * (sizes from snerf_mlp_train_sizes) except the transposed weight stream, which comes from snerf_mlp_pack_t_bf16. */
SNERF_API int64_t snerf_mlp_packed_t_bf16_bytes(const snerf_mlp_desc *desc, int nsplit, int input_grad);
SNERF_API int snerf_mlp_pack_t_bf16(const snerf_mlp_desc *desc, const float *params_flat, void *packed_t, int nsplit,
                          int input_grad, snerf_stream_t stream);
SNERF_API int snerf_mlp_bwd_bf16_f32(const snerf_mlp_desc *desc, const void *packed_t, int nsplit, const float *act,
                           const float *d_raw, int64_t n, float *dy, float *gpart, float *flat_grad,
                           snerf_stream_t stream);
SNERF_API int snerf_mlp_bwd_inputs_bf16_f32(const snerf_mlp_desc *desc, const void *packed_t, int nsplit, const float *act,
                                  const float *d_raw, const float *x, const float *dirs, int dirs_per_sample,
                                  int samples_per_ray, int64_t n, float *dy, float *gpart, float *flat_grad,
                                  float *d_x, float *d_dirs, snerf_stream_t stream);

/* ---- a2 backward (training) ------------------------------------------------------------------------
 * Buffer sizes for n samples: activations saved by the forward, per-layer output gradients, the
 * transposed weight stream, the split-K partial gradients (gpart_count chunks). */
SNERF_API int snerf_mlp_train_sizes(const snerf_mlp_desc *desc, int64_t n, int64_t *act_floats, int64_t *dy_floats,
                          int64_t *packed_t_floats, int64_t *gpart_floats, int32_t *gpart_count);
/* Where the backward leaves the per-layer output gradients in `dy` (an output, not only scratch: gradients w.r.t.
 * per-ray additional inputs and w.r.t. already-encoded input rows are contractions of these with weight columns, see
 * smpl_nerf_amd/nets.py).  Layers in snerf_mlp_param_floats order (positions_pose_input, positional_net[*],
 * additional_linear_layer, sigma_out_layer, directional_input, directional_net[0], rgb_out_layer); arrays of
 * SNERF_MAX_MLP_LAYERS entries, each nullable.  d Y_l[s, f] (sample s of n, output feature f < n_out[l]) is the float at
 *     dy[((first_row[l] + f / 16) * n + s) * 16 + f % 16]                                   (tile-row-major) */
#define SNERF_MAX_MLP_LAYERS 21
SNERF_API int snerf_mlp_dy_layout(const snerf_mlp_desc *desc, int32_t *n_layers, int32_t *first_row, int32_t *n_out, int32_t *n_in);
/* Gradients w.r.t. inputs that enter a layer through plain weight columns - per-ray additional inputs, already-encoded input
 * rows, the pose rows of the warp net - as contractions of a stored d Y_l with those columns (what autograd leaves in the
 * .grad of x when y = x W^T; models/render_ray_net.py:43-50, models/append_vertices_pipeline.py:30-58):
 *     out[s, out_col0 + c] (+)= sum_{f < n_feat} d Y_l[s, f] * w[f, col0 + c],   c < ncols
 * dy / first_row: the backward's `dy` buffer and the layer's first tile-row (snerf_mlp_dy_layout; the warp net's layer 0 is
 * tile-row 0); w [n_feat, w_stride] row-major = the layer's weight matrix in params_flat (n_feat <= 256).
 * samples_per_ray == 0: one output row per sample, out [n, out_stride].  samples_per_ray > 0: the rows of a ray's samples
 * are summed (the pipelines expand one input row per ray over its samples), out [n / samples_per_ray, out_stride];
 * scratch: snerf_dy_contract_scratch_floats(n, ncols, samples_per_ray) floats.  accumulate != 0: out += (several layers
 * read the same inputs: layer 0 and every skip layer). */
SNERF_API int64_t snerf_dy_contract_scratch_floats(int64_t n, int ncols, int samples_per_ray);
SNERF_API int snerf_dy_contract_f32(const float *dy, int64_t n, int first_row, int n_feat, const float *w, int w_stride, int col0,
                          int ncols, int samples_per_ray, float *out, int64_t out_stride, int out_col0, int accumulate,
                          float *scratch, snerf_stream_t stream);

/* snerf_mlp_fwd_f32 that also saves every layer input into `act` (act_floats): fp32 tile-rows, followed by one
 * sign bit per ReLU output (the masks the split-bf16 dgrad reads instead of the activation rows). */
SNERF_API int snerf_mlp_fwd_train_f32(const snerf_mlp_desc *desc, const float *packed, const float *x,
                            const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                            int samples_per_ray, float *raw, float *act, snerf_stream_t stream);
/* The same for already-encoded rows (RenderRayNet.forward(x) under autograd). */
SNERF_API int snerf_mlp_fwd_encoded_train_f32(const snerf_mlp_desc *desc, const float *packed, const float *x_enc,
                                    int64_t n, int64_t row_floats, float *raw, float *act,
                                    snerf_stream_t stream);
/* params_flat -> transposed weight stream for the dgrad kernel (once per weight update).  input_grad != 0
 * adds the encoder-column transposes snerf_mlp_bwd_inputs_f32 consumes (a different stream: pack one per use). */
SNERF_API int snerf_mlp_pack_t_f32(const snerf_mlp_desc *desc, const float *params_flat, float *packed_t,
                         int input_grad, snerf_stream_t stream);
/* d_raw [n,4] -> flat_grad (snerf_mlp_param_floats floats, state_dict order, OVERWRITTEN): what
 * autograd leaves in .grad of the 26 parameter tensors after (raw * d_raw).sum().backward().
 * dy, gpart: scratch (snerf_mlp_train_sizes).  Three launches: dgrad, split-K wgrad, reduce. */
SNERF_API int snerf_mlp_bwd_f32(const snerf_mlp_desc *desc, const float *packed_t, const float *act,
                      const float *d_raw, int64_t n, float *dy, float *gpart, float *flat_grad,
                      snerf_stream_t stream);

/* snerf_mlp_bwd_f32 that also back-propagates into the inputs of snerf_mlp_fwd_train_f32: d_x [n,3] (through
 * the position encoding of layer 0 and the skip layers) and d_dirs [n,3] (through the direction encoding and
 * the normalisation d/|d|, models/smpl_nerf_pipeline.py:54-56).  Needs the input_grad=1 transposed stream.
 * Encoders of up to 8 position / 8 direction k-blocks of 16 slots (identity columns, up to 16 frequencies); the
 * split-precision variant (snerf_mlp_bwd_inputs_bf16_f32) takes the default-sized ones (<= 4 / 2 k-blocks: L = 10 / 4
 * without identity columns, or smaller). */
SNERF_API int snerf_mlp_bwd_inputs_f32(const snerf_mlp_desc *desc, const float *packed_t, const float *act,
                             const float *d_raw, const float *x, const float *dirs, int dirs_per_sample,
                             int samples_per_ray, int64_t n, float *dy, float *gpart, float *flat_grad,
                             float *d_x, float *d_dirs, snerf_stream_t stream);

/* Same network on already-encoded rows x_enc [n, row_floats] (the literal RenderRayNet.forward(x)
 * signature): positions_pose = x[:, :positions_dim+add_dim], directions = x[:, -directions_dim:]
 * (models/render_ray_net.py:42-43). */
SNERF_API int snerf_mlp_fwd_encoded_f32(const snerf_mlp_desc *desc, const float *packed, const float *x_enc,
                              int64_t n, int64_t row_floats, float *raw, snerf_stream_t stream);

/* ---- a7: WarpFieldNet fused with x' = x + warp and the per-sample view direction ---------------------
 * Mirrors WarpFieldNet.__init__ (models/warp_field_net.py:8-15): linear1 [width, positions_dim+pose_dim],
 * linear2 [3, width]; positions_dim is expressed through the position encoder. */
typedef struct snerf_warp_desc {
    int32_t width;        /* 1 .. 256 (--netwidth_warp); other widths than 256 / 128 run zero-padded; split precision: 256 */
    int32_t pos_freqs;    /* 10 */
    int32_t pos_identity; /* 0 */
    int32_t pose_dim;     /* 40 = encoded pose of the two joints (models/smpl_nerf_pipeline.py:28-30) */
} snerf_warp_desc;
SNERF_API int64_t snerf_warp_param_floats(const snerf_warp_desc *desc); /* linear1.weight, .bias, linear2.weight, .bias */
SNERF_API int64_t snerf_warp_packed_floats(const snerf_warp_desc *desc);
SNERF_API int snerf_warp_pack_f32(const snerf_warp_desc *desc, const float *params_flat, float *packed,
                        snerf_stream_t stream);
/* x [n,3], pose_enc [n/samples_per_ray, pose_dim], o [n/samples_per_ray, 3] ->
 * warp [n,3] = net([PE(x) | pose_enc]) (models/warp_field_net.py:17-21),
 * warped [n,3] = x + warp (models/smpl_nerf_pipeline.py:49), sdirs [n,3] = warped - o (:52-53).
 * warped / sdirs nullable.  With pos_freqs = pos_identity = 0 the net reads only pose_enc rows
 * (x may be NULL): the literal WarpFieldNet.forward(x_rows) with samples_per_ray = 1. */
SNERF_API int snerf_warp_fwd_f32(const snerf_warp_desc *desc, const float *packed, const float *x,
                       const float *pose_enc, const float *o, int64_t n, int samples_per_ray,
                       float *warp, float *warped, float *sdirs, snerf_stream_t stream);

/* With a caller-allocated workspace of snerf_warp_fold_workspace_bytes(desc, n, samples_per_ray) bytes (0: the fold does not
 * apply) the pose columns of linear1 - per-ray constants - are folded into one vector per ray; NULL (= snerf_warp_fwd_f32):
 * the per-sample form; too small: SNERF_E_BADARG.  Same contract as snerf_mlp_fwd_ws_f32. */
SNERF_API int64_t snerf_warp_fold_workspace_bytes(const snerf_warp_desc *desc, int64_t n, int samples_per_ray);
SNERF_API int snerf_warp_fwd_ws_f32(const snerf_warp_desc *desc, const float *packed, const float *x, const float *pose_enc,
                          const float *o, int64_t n, int samples_per_ray, float *warp, float *warped, float *sdirs,
                          void *workspace, int64_t workspace_bytes, snerf_stream_t stream);

/* The same forward on the bf16 matrix cores (split-bf16, always three parts / six products: fp32-class accuracy - the warp
 * moves the sample in front of the 2^9 band of the position encoding).  packed from snerf_warp_pack_bf16
 * (snerf_warp_packed_bf16_bytes bytes).  Fused mode only (x given); width 256. */
SNERF_API int64_t snerf_warp_packed_bf16_bytes(const snerf_warp_desc *desc);
SNERF_API int snerf_warp_pack_bf16(const snerf_warp_desc *desc, const float *params_flat, void *packed, snerf_stream_t stream);
SNERF_API int snerf_warp_fwd_bf16_f32(const snerf_warp_desc *desc, const void *packed, const float *x, const float *pose_enc,
                            const float *o, int64_t n, int samples_per_ray, float *warp, float *warped, float *sdirs,
                            snerf_stream_t stream);

/* Training of the warp net: forward that saves [PE(x) | pose | h] tile-rows, the transposed head, and the
 * backward d_warp [n,3] (= d loss / d warp, the sum of what arrives through warp, warped and sdirs) ->
 * flat_grad (snerf_warp_param_floats floats, state_dict order, overwritten).  Sizes as snerf_mlp_train_sizes. */
SNERF_API int snerf_warp_train_sizes(const snerf_warp_desc *desc, int64_t n, int64_t *act_floats, int64_t *dy_floats,
                           int64_t *packed_t_floats, int64_t *gpart_floats);
SNERF_API int snerf_warp_fwd_train_f32(const snerf_warp_desc *desc, const float *packed, const float *x,
                             const float *pose_enc, const float *o, int64_t n, int samples_per_ray,
                             float *warp, float *warped, float *sdirs, float *act, snerf_stream_t stream);
SNERF_API int snerf_warp_pack_t_f32(const snerf_warp_desc *desc, const float *params_flat, float *packed_t,
                          snerf_stream_t stream);
SNERF_API int snerf_warp_bwd_f32(const snerf_warp_desc *desc, const float *packed_t, const float *act,
                       const float *d_warp, int64_t n, float *dy, float *gpart, float *flat_grad,
                       snerf_stream_t stream);

/* ---- 8(f)-1: on-device ray generation + stratified coarse sampling --------------------------------------
 * Replaces get_rays (utils.py:50-54) + CoarseSampling (datasets/transforms.py:80-89) + ToTensor (:13-21) for a
 * batch of rays.  poses: fp64 [n_frames, 4, 4] camera-to-world; ray_index int64 [B] = frame*H*W + row*W + col;
 * jitter fp64 [B] = the per-ray np.random.rand() scalar; lower/span fp64 [Nc] = the bin tables of
 * CoarseSampling (lower, upper - lower).  fp64 arithmetic in the reference's order, one rounding to fp32:
 * bit-identical to the numpy path.  Outputs: samples [B,Nc,3], o [B,3], d [B,3], z [B,Nc] fp32. */
SNERF_API int snerf_raygen_f64(const double *poses, int64_t n_frames, int H, int W, double focal, const double *lower,
                     const double *span, int Nc, const int64_t *ray_index, const double *jitter, int64_t B,
                     float *samples, float *o, float *d, float *z, snerf_stream_t stream);

/* ---- a3: the whole NerfPipeline.forward for inference in one call (models/nerf_pipeline.py:14-67) -------------
 * Five launches on `stream`: fused encode+MLP (coarse) -> composite -> inverse-CDF sampler + merge + points ->
 * fused encode+MLP (fine) -> composite.  precision: 0 = fp32 kernel (packed_* from snerf_mlp_pack_f32), 2 / 3 / 16 =
 * split-bf16 / split-fp16 with that nsplit (packed_* from snerf_mlp_pack_bf16).  Inputs as the Solver hands them over
 * (solver/nerf_solver.py:77-81): ray_samples [B,Nc,3], rays_o [B,3], rays_d [B,3], z_vals [B,Nc]; u [Nf] =
 * linspace(0,1,Nf) (utils.py:204-205); noise_coarse [B,Nc] / noise_fine [B,Nc+Nf] nullable (sigma noise,
 * utils.py:171-173).  Nf == 0 is run_fine = 0: the fine outputs are copies of the coarse ones (:43-44).
 * workspace: snerf_render_rays_workspace_bytes(B, Nc, Nf) bytes, 16-byte aligned.  Outputs: rgb [B,3],
 * rgb_fine [B,3], samples_fine [B,Nc+Nf,3], densities_fine [B,Nc+Nf] (the `alpha` of utils.py:169). */
SNERF_API int64_t snerf_render_rays_workspace_bytes(int64_t B, int Nc, int Nf);
SNERF_API int snerf_render_rays_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse,
                          const snerf_mlp_desc *desc_fine, const void *packed_fine, int precision,
                          const float *ray_samples, const float *rays_o, const float *rays_d, const float *z_vals,
                          const float *u, const float *noise_coarse, const float *noise_fine, int64_t B, int Nc,
                          int Nf, int white_background, void *workspace, float *rgb, float *rgb_fine,
                          float *samples_fine, float *densities_fine, snerf_stream_t stream);

/* ---- 8(f)-4: the same for nets with per-ray additional inputs (models/append_smpl_params_pipeline.py:14-91,
 * append_to_nerf_pipeline.py:14-90 - the paper's headline model) ------------------------------------------------------
 * snerf_render_rays_f32 with additional [B, add_dim] = the pose row of each ray (raw or encoded, :29-37), read by both nets as
 * their `add` input (descriptors with add_dim > 0, add_first as the pipeline has it).  In fp32 the rows are folded into one
 * vector per ray and layer inside the call (snerf_mlp_fwd_ws_f32) - the fold table lives in the workspace:
 * snerf_render_rays_add_workspace_bytes(...) bytes, 16-byte aligned. */
SNERF_API int64_t snerf_render_rays_add_workspace_bytes(const snerf_mlp_desc *desc_coarse, const snerf_mlp_desc *desc_fine, int64_t B, int Nc,
                                              int Nf);
SNERF_API int snerf_render_rays_add_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const snerf_mlp_desc *desc_fine,
                              const void *packed_fine, int precision, const float *ray_samples, const float *rays_o,
                              const float *rays_d, const float *z_vals, const float *additional, const float *u,
                              const float *noise_coarse, const float *noise_fine, int64_t B, int Nc, int Nf, int white_background,
                              void *workspace, float *rgb, float *rgb_fine, float *samples_fine, float *densities_fine,
                              snerf_stream_t stream);

/* ---- a7: the whole SmplNerfPipeline.forward for inference in one call (models/smpl_nerf_pipeline.py:16-100) ------
 * snerf_render_rays_f32 with the warp stage in front of both nets: warp(samples) -> x' = x + warp, per-sample directions
 * x' - o -> net -> composite (coarse: distances scaled by |x' - o| per sample, :63; fine: by the ray direction, :95-98);
 * the hierarchical samples are drawn on the un-warped ray (:68).  pose_enc [B, pose_dim] = the encoded two joint angles
 * (:28-30).  precision as in snerf_render_rays_f32 (the warp net runs fp32 for 0 and three-part split-bf16 otherwise;
 * packed_warp from snerf_warp_pack_f32 / snerf_warp_pack_bf16 accordingly).  Nf >= 1.  workspace:
 * snerf_render_rays_smpl_workspace_bytes bytes.  Outputs: rgb [B,3], rgb_fine [B,3], warp_fine / samples_fine /
 * warped_fine [B,Nc+Nf,3], densities_fine [B,Nc+Nf]. */
SNERF_API int64_t snerf_render_rays_smpl_workspace_bytes(int64_t B, int Nc, int Nf);
SNERF_API int snerf_render_rays_smpl_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse,
                               const snerf_mlp_desc *desc_fine, const void *packed_fine,
                               const snerf_warp_desc *desc_warp, const void *packed_warp, int precision,
                               const float *ray_samples, const float *rays_o, const float *rays_d, const float *z_vals,
                               const float *pose_enc, const float *u, const float *noise_coarse,
                               const float *noise_fine, int64_t B, int Nc, int Nf, int white_background,
                               void *workspace, float *rgb, float *rgb_fine, float *warp_fine, float *samples_fine,
                               float *warped_fine, float *densities_fine, snerf_stream_t stream);

/* ---- a9 + 8(f)-2: the per-batch body of NerfSolver.train as one call (solver/nerf_solver.py:76-87) ---------------------
 *     rgb, rgb_fine, .. = pipeline(batch); loss = MSE(rgb, gt) + MSE(rgb_fine, gt); loss.backward(); Adam.step()
 * (models/nerf_pipeline.py:14-67, solver/nerf_solver.py:48-52, :31-33).  Everything is enqueued on `stream`; nothing is
 * allocated, nothing synchronises, and no argument changes from step to step except the batch pointers (the optimiser's
 * step counter and bias corrections live on the device): the call can be captured into a HIP graph and replayed.
 *
 * Weight streams.  A training step consumes two streams per net - `packed` (snerf_mlp_pack_f32 / snerf_mlp_pack_bf16) and
 * `packed_t` (snerf_mlp_pack_t_f32 / snerf_mlp_pack_t_bf16 with input_grad = 0) - which the caller packs ONCE; afterwards the
 * optimiser step keeps them current: fp32 streams are refreshed in place, element by element, through the slot tables of
 * snerf_mlp_stream_slots (no re-pack launches); split-precision streams (pre-split parts) are re-packed inside the call. */

/* slot_fwd[i] / slot_t[i] (int32, snerf_mlp_param_floats entries each, nullable): index of the float of the fp32 forward /
 * transposed stream that holds parameter i of params_flat, or -1 (biases do not appear in the transposed stream).
 * input_grad selects WHICH transposed stream (snerf_mlp_pack_t_f32's input_grad: 0 for snerf_nerf_train_*, 1 for
 * snerf_smpl_nerf_train_*).  Every parameter occupies at most one float of each stream. */
SNERF_API int snerf_mlp_stream_slots(const snerf_mlp_desc *desc, int32_t *slot_fwd, int32_t *slot_t, int input_grad, snerf_stream_t stream);

/* torch.optim.Adam(params, lr, betas, eps, weight_decay) (amsgrad = False) over ONE flat fp32 parameter buffer - the
 * statements of torch's single-tensor update in their order (solver/nerf_solver.py:11-14, 31-33, 87). */
typedef struct snerf_adam_state {
    float *params;       /* [n_params] */
    const float *grads;  /* [n_params] */
    float *exp_avg;      /* [n_params], zero before the first step */
    float *exp_avg_sq;   /* [n_params], zero before the first step */
    int64_t n_params;
    float *scratch;      /* device, 2 floats per range of a call (64 floats cover the maximum of 32 ranges) */
    double lr, beta1, beta2, eps, weight_decay;
} snerf_adam_state;
/* A run of adjacent parameter tensors that take part in a step.  torch keeps one step counter per parameter tensor and
 * skips tensors whose .grad is None (the fine net with run_fine = 0): leave those out of the ranges.  step: DEVICE int64
 * [n_steps] = the counters of the range's tensors (0 before the first step), all equal on entry - tensors with different
 * counts go into different ranges - and all incremented by the call; the bias corrections use step[0] + 1. */
typedef struct snerf_adam_range {
    int64_t begin, end;  /* parameters [begin, end) of the flat buffer */
    int64_t *step;
    int32_t n_steps;
} snerf_adam_range;
/* A RenderRayNet inside the flat buffer whose weight streams the step keeps current. */
typedef struct snerf_adam_net {
    const snerf_mlp_desc *desc;
    int64_t param_offset; /* its snerf_mlp_param_floats parameters (state_dict order) start at params + param_offset */
    int32_t precision;    /* 0: fp32 streams (refreshed in place); 2 / 3 / SNERF_SPLIT_F16X3: split streams (re-packed) */
    void *packed;         /* nullable: not kept current */
    void *packed_t;       /* nullable */
    const int32_t *slot_fwd, *slot_t; /* snerf_mlp_stream_slots (precision 0 only) */
} snerf_adam_net;
/* One optimiser step on the ranges (HOST array, at most 32) with the streams of at most 8 nets (HOST array) kept current. */
SNERF_API int snerf_adam_step_f32(const snerf_adam_state *state, const snerf_adam_range *ranges_host, int n_ranges,
                        const snerf_adam_net *nets_host, int n_nets, snerf_stream_t stream);

/* The batch as the Solver hands it to the pipeline (solver/nerf_solver.py:77-81) plus what utils.py draws inside. */
typedef struct snerf_nerf_batch {
    const float *ray_samples;  /* [B, Nc, 3] */
    const float *rays_o;       /* [B, 3] */
    const float *rays_d;       /* [B, 3] */
    const float *z_vals;       /* [B, Nc] */
    const float *rgb_truth;    /* [B, 3] */
    const float *u;            /* [Nf] = linspace(0, 1, Nf) (utils.py:204-206); unused with Nf == 0 */
    const float *noise_coarse; /* nullable [B, Nc]: sigma noise (utils.py:171-173) */
    const float *noise_fine;   /* nullable [B, Nc + Nf] */
    const float *additional;   /* [B, add_dim] per-ray additional inputs of nets with add_dim > 0 (the pose rows of
                                  models/append_smpl_params_pipeline.py:29-52, append_to_nerf_pipeline.py:26); else NULL */
    int64_t B;
    int32_t Nc, Nf;            /* Nf == 0 is run_fine = 0: loss = 2 MSE(rgb), no fine gradients (nerf_pipeline.py:43-44) */
    int32_t white_background;
} snerf_nerf_batch;

/* Forward with saved layer inputs, loss, backward: grad_coarse / grad_fine (snerf_mlp_param_floats floats each, state_dict
 * order, OVERWRITTEN; grad_fine untouched with Nf == 0) = what autograd leaves in .grad after loss.backward();
 * loss [3] = {MSE(rgb) + MSE(rgb_fine), MSE(rgb), MSE(rgb_fine)}; rgb / rgb_fine [B, 3] = the rendered colours.
 * precision as in snerf_render_rays_f32 (packed_* / packed_t_* from the matching pack calls).
 * The batch is walked in chunks of rays_per_chunk rays (<= 0 or > B: one chunk): d loss / d rgb of a ray does not depend on
 * the other rays, so forward and backward run chunk by chunk and the parameter gradients are summed in chunk order - the
 * saved activations (21 KB per ray-sample) are sized by the chunk, nothing is recomputed.
 * aux_stream (nullable): a second stream of the caller's.  Chunks of at most 262 144 samples (1024 rays of 64 + 192; the
 * README's 64-ray batches) leave CUs idle in some kernel; there the coarse net's backward - independent of the fine net's, the
 * hierarchical samples being detached (utils.py:260) - is enqueued on aux_stream beside it (forked from and joined back into
 * `stream` with events, also under graph capture).  NULL: everything on `stream`.  Same results either way.
 * workspace: snerf_nerf_train_workspace_bytes(...) bytes, 256-byte aligned. */
SNERF_API int64_t snerf_nerf_train_workspace_bytes(const snerf_mlp_desc *desc_coarse, const snerf_mlp_desc *desc_fine, int64_t B, int Nc,
                                         int Nf, int64_t rays_per_chunk);
SNERF_API int snerf_nerf_train_grads_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                               const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine, int precision,
                               const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace, float *grad_coarse,
                               float *grad_fine, float *loss, float *rgb, float *rgb_fine, snerf_stream_t stream,
                               snerf_stream_t aux_stream);
/* snerf_nerf_train_grads_f32 followed by snerf_adam_step_f32 (single-GPU step; a data-parallel trainer calls the two halves
 * with its gradient all-reduce in between). */
SNERF_API int snerf_nerf_train_step_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                              const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine, int precision,
                              const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace, float *grad_coarse,
                              float *grad_fine, float *loss, float *rgb, float *rgb_fine, const snerf_adam_state *adam,
                              const snerf_adam_range *ranges_host, int n_ranges, const snerf_adam_net *nets_host, int n_nets,
                              snerf_stream_t stream, snerf_stream_t aux_stream);

/* The same with d loss / d batch->additional as one more output (nets with add_dim > 0): what autograd returns for the per-ray pose
 * rows / vertex floats the pipelines expand over a ray's samples (models/append_smpl_params_pipeline.py:29-52,
 * append_to_nerf_pipeline.py:26, append_vertices_pipeline.py:37-58) - the gradient AppendVerticesSolver's second parameter
 * group (solver/append_vertices_solver.py: the pose estimator, lrate_pose) and a goal_pose that requires a gradient are trained
 * from.  d_additional [B, add_dim] is OVERWRITTEN: the stored d Y of layer 0 and of every skip layer of both nets contracted with
 * the weight columns that read the additional inputs (snerf_dy_contract_f32), summed over the samples of each ray.  params_*: the
 * nets' parameters (snerf_mlp_param_floats floats, state_dict order - the weight columns are read from there); the step variant
 * reads them before Adam updates them.  input_grads == NULL or d_additional == NULL: exactly snerf_nerf_train_grads_f32 / _step_f32.
 * The workspace of snerf_nerf_train_workspace_bytes covers the contraction's scratch. */
typedef struct snerf_input_grads {
    float *d_additional;
    const float *params_coarse;
    const float *params_fine; /* unused with Nf == 0 */
} snerf_input_grads;
SNERF_API int snerf_nerf_train_grads_ig_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                  const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine, int precision,
                                  const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace, float *grad_coarse,
                                  float *grad_fine, float *loss, float *rgb, float *rgb_fine, const snerf_input_grads *input_grads,
                                  snerf_stream_t stream, snerf_stream_t aux_stream);
SNERF_API int snerf_nerf_train_step_ig_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                 const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine, int precision,
                                 const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace, float *grad_coarse,
                                 float *grad_fine, float *loss, float *rgb, float *rgb_fine, const snerf_adam_state *adam,
                                 const snerf_adam_range *ranges_host, int n_ranges, const snerf_adam_net *nets_host, int n_nets,
                                 const snerf_input_grads *input_grads, snerf_stream_t stream, snerf_stream_t aux_stream);

/* ---- a7 + a9: SmplNerfSolver.train's per-batch body as one call (solver/smpl_nerf_solver.py:76-89 with the default loss,
 * models/smpl_nerf_pipeline.py:16-100, human_pose_encoding = 1) ------------------------------------------------------
 * snerf_nerf_train_* with the warp stage in front of both nets and its backward behind them: warp forward (saving its rows)
 * -> net on (x', x' - o) -> compositing (coarse: scaled per sample by |x' - o|; fine: by the ray direction) -> ... -> MSE ->
 * compositing backward (coarse: also into x' - o) -> net backward INTO its inputs (snerf_mlp_bwd_inputs_*: packed_t_* are the
 * input_grad = 1 streams) -> d warp = d x' + d (x' - o) -> warp backward; the warp net's gradient is the sum over its two
 * evaluations.  pose_enc [B, pose_dim] = the encoded two joint angles (:28-30).  The warp net trains in fp32 in every
 * precision mode.  In the split-precision modes the nets' input gradients exist for encoders of at most 4 position / 2
 * direction k-blocks of 16 slots (the defaults: 10 / 4 frequencies without identity columns); wider ones are SNERF_E_BADARG
 * here - run those nets with precision 0 (the Python trainer takes its autograd path, which routes that dgrad to fp32).  grad_warp: snerf_warp_param_floats floats.  Ray chunks, workspace, loss, rgb as in snerf_nerf_train_grads_f32.
 * The step variant runs snerf_adam_step_f32 and then re-packs the warp net's two streams (packed_warp from snerf_warp_pack_f32,
 * packed_t_warp from snerf_warp_pack_t_f32) from its parameters at params + warp_param_offset. */
SNERF_API int64_t snerf_smpl_nerf_train_workspace_bytes(const snerf_mlp_desc *desc_coarse, const snerf_mlp_desc *desc_fine,
                                              const snerf_warp_desc *desc_warp, int64_t B, int Nc, int Nf, int64_t rays_per_chunk);
SNERF_API int snerf_smpl_nerf_train_grads_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                    const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                    const snerf_warp_desc *desc_warp, const float *packed_warp, const float *packed_t_warp,
                                    int precision, const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk,
                                    void *workspace, float *grad_coarse, float *grad_fine, float *grad_warp, float *loss, float *rgb,
                                    float *rgb_fine, snerf_stream_t stream);
SNERF_API int snerf_smpl_nerf_train_step_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                   const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                   const snerf_warp_desc *desc_warp, float *packed_warp, float *packed_t_warp, int precision,
                                   const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk, void *workspace,
                                   float *grad_coarse, float *grad_fine, float *grad_warp, float *loss, float *rgb, float *rgb_fine,
                                   const snerf_adam_state *adam, const snerf_adam_range *ranges_host, int n_ranges,
                                   const snerf_adam_net *nets_host, int n_nets, int64_t warp_param_offset, snerf_stream_t stream);
/* packed_warp / packed_t_warp (each nullable) from the warp net's parameters at params + warp_param_offset: what a
 * data-parallel caller runs behind snerf_adam_step_f32. */
SNERF_API int snerf_warp_repack_f32(const snerf_warp_desc *desc_warp, const float *params, int64_t n_params, int64_t warp_param_offset,
                          float *packed_warp, float *packed_t_warp, snerf_stream_t stream);

/* ---- a2 / a7 for ANY width: nn.Linear as stand-alone fp32 MFMA GEMMs (0.1.9) -----------------------------------------------
 * config_parser.py:20,24,30 accept any --netwidth / --netwidth_fine / --netwidth_warp.  The fused kernels above cover
 * RenderRayNet up to 512 features and WarpFieldNet up to 256; wider nets run one nn.Linear at a time with the activations in HBM,
 * as models/render_ray_net.py:42-61 / models/warp_field_net.py:17-21 do, through these entries.  Weights are read in the
 * reference's own layout: w is [m, k] = [out, in] row-major with leading dimension ldw (a slice of the flat parameter vector; a
 * column block of a wider matrix - the two input blocks of a skip layer - is w + offset with the full matrix's ldw).  All matrices
 * row-major fp32 with explicit leading dimensions (in floats); exact fp32 (v_mfma_f32_16x16x4_f32).
 *   snerf_linear_fwd_f32         y[n, m]  (+)= x[n, k] w^T, then + bias[m] (nullable), then ReLU (relu != 0)
 *   snerf_linear_bwd_input_f32   dx[n, k] (+)= dy[n, m] w
 *   snerf_linear_bwd_weight_f32  dw[m, k] (+)= dy^T x and db[m] (+)= column sums of dy (db nullable); the sum over the samples is
 *                                split into slices whose partials go through `scratch`
 *                                (snerf_linear_bwd_weight_scratch_floats(n, m, k) floats) and are added in slice order
 *   snerf_relu_bwd_f32           dy[i, j] = y[i, j] > 0 ? dy[i, j] : 0, in place
 * (accumulate != 0: += instead of =). */
SNERF_API int snerf_linear_fwd_f32(const float *x, int64_t n, int k, int64_t ldx, const float *w, int64_t ldw, int m, const float *bias,
                         int accumulate, int relu, float *y, int64_t ldy, snerf_stream_t stream);
SNERF_API int snerf_linear_bwd_input_f32(const float *dy, int64_t n, int m, int64_t lddy, const float *w, int64_t ldw, int k, int accumulate,
                               float *dx, int64_t lddx, snerf_stream_t stream);
SNERF_API int64_t snerf_linear_bwd_weight_scratch_floats(int64_t n, int m, int k);
SNERF_API int snerf_linear_bwd_weight_f32(const float *dy, int64_t n, int m, int64_t lddy, const float *x, int64_t ldx, int k, int accumulate,
                                float *dw, int64_t lddw, float *db, float *scratch, snerf_stream_t stream);
SNERF_API int snerf_relu_bwd_f32(float *dy, const float *y, int64_t n, int m, int64_t lddy, int64_t ldy, snerf_stream_t stream);

/* ---- 8(e): the data-parallel step as one call ---------------------------------------------------------------------------
 * Rays of independent images shard over the GPUs of a node, one process per GPU; the only exchange of the path is the average of
 * the replicated nets' gradients.  snerf_comm_t is an ncclComm_t of RCCL (bound at run time: librccl.so.1 is loaded by the first
 * snerf_comm_* / *_dp_* call, the copy the process already holds if there is one).  A host with its own communicator passes it
 * as is; one without creates it here: rank 0 calls snerf_comm_unique_id, hands the SNERF_COMM_ID_BYTES bytes to the other ranks
 * by whatever channel it has (a file, MPI, torch.distributed's store), and every rank calls snerf_comm_init_rank - collective,
 * with its HIP device current.  snerf_comm_destroy(NULL) is a no-op. */
typedef void *snerf_comm_t; /* ncclComm_t */
#define SNERF_COMM_ID_BYTES 128
SNERF_API int snerf_comm_unique_id(void *id_host);
SNERF_API int snerf_comm_init_rank(const void *id_host, int world_size, int rank, snerf_comm_t *comm);
SNERF_API int snerf_comm_destroy(snerf_comm_t comm);
SNERF_API int snerf_comm_info(snerf_comm_t comm, int32_t *world_size, int32_t *rank);
/* buf[0 .. n) <- its average over the ranks, in place, enqueued on `stream` (ncclAllReduce, ncclAvg, fp32). */
SNERF_API int snerf_comm_allreduce_avg_f32(snerf_comm_t comm, float *buf, int64_t n, snerf_stream_t stream);
/* snerf_nerf_train_step_f32 / snerf_smpl_nerf_train_step_f32 with the gradient average between the backward and the optimiser:
 * ... -> backward -> ncclAllReduce(ncclAvg) of adam->grads[0 .. adam->n_params) - the trainer's flat gradient buffer, which holds
 * grad_coarse / grad_fine (/ grad_warp) as segments and is the same size on every rank -> Adam.  The collectives of a nerf step
 * are the same two on every rank whatever its batch size, chunking and streams (0.1.9): the coarse net's segment, then the rest
 * of the buffer (one grouped launch), both on `stream` behind the join of the concurrent backward - one communicator never has
 * collectives in flight on two streams; the smpl_nerf steps issue one all-reduce of the whole buffer on `stream`.  batch->B == 0 is valid here (a rank whose
 * shard ran out): zero gradient, zero loss, the same collectives, the optimiser step.  Nothing is synchronised; every rank must
 * make the same calls in the same order (RCCL's rule).  Graph-capturable.  SNERF_RCCL_LIB=<path> (read at the first call)
 * names the library to bind instead of librccl.so.1. */
SNERF_API int snerf_nerf_train_step_dp_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                 const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine, int precision,
                                 const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace, float *grad_coarse,
                                 float *grad_fine, float *loss, float *rgb, float *rgb_fine, const snerf_adam_state *adam,
                                 const snerf_adam_range *ranges_host, int n_ranges, const snerf_adam_net *nets_host, int n_nets,
                                 snerf_comm_t comm, snerf_stream_t stream, snerf_stream_t aux_stream);
/* snerf_nerf_train_step_ig_f32 with the gradient average: the data-parallel step of the pose-conditioned pipelines whose
 * estimator / goal_pose is trained too (BASELINE configs[4]; solver/append_vertices_solver.py:26-31, 77-82: the lrate_pose group).
 * input_grads->d_additional [B, add_dim] holds THIS rank's rows (per-ray data: nothing to average); the caller continues its
 * autograd from them into the estimator and averages that module's gradient segment with snerf_comm_allreduce_avg_f32 before
 * its optimiser step.  Same collectives, in the same order, as snerf_nerf_train_step_dp_f32. */
SNERF_API int snerf_nerf_train_step_dp_ig_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                    const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine, int precision,
                                    const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace, float *grad_coarse,
                                    float *grad_fine, float *loss, float *rgb, float *rgb_fine, const snerf_adam_state *adam,
                                    const snerf_adam_range *ranges_host, int n_ranges, const snerf_adam_net *nets_host, int n_nets,
                                    const snerf_input_grads *input_grads, snerf_comm_t comm, snerf_stream_t stream,
                                    snerf_stream_t aux_stream);
SNERF_API int snerf_smpl_nerf_train_step_dp_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                      const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                      const snerf_warp_desc *desc_warp, float *packed_warp, float *packed_t_warp, int precision,
                                      const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk, void *workspace,
                                      float *grad_coarse, float *grad_fine, float *grad_warp, float *loss, float *rgb, float *rgb_fine,
                                      const snerf_adam_state *adam, const snerf_adam_range *ranges_host, int n_ranges,
                                      const snerf_adam_net *nets_host, int n_nets, int64_t warp_param_offset, snerf_comm_t comm,
                                      snerf_stream_t stream);

/* The smpl_nerf step with an auxiliary stream (0.1.8).  As snerf_smpl_nerf_train_grads_f32 / _step_f32 / _step_dp_f32 with one more
 * argument: aux_stream (may be NULL or == stream: then exactly those calls).  For chunks of at most 262144 fine-pass samples
 * (snerf_smpl_nerf_train_workspace_bytes already counts the second scratch set under that rule) the coarse chain of the backward -
 * compositing, coarse net dgrad + wgrad, the warp net's backward on the coarse samples - runs on aux_stream beside the fine chain on
 * `stream`, forked and joined with events inside the call; the coarse chain's warp-net gradient is added behind the join in the order
 * of the sequential form, so the results are bit-identical with and without aux_stream.  comm (step form; may be NULL): the flat
 * gradient buffer adam->grads is averaged over the ranks (ncclAllReduce, ncclAvg) between the backward and Adam.
 * Replaces: solver/smpl_nerf_solver.py:76-89 per batch, as the entry points it extends. */
SNERF_API int snerf_smpl_nerf_train_grads_aux_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                        const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                        const snerf_warp_desc *desc_warp, const float *packed_warp, const float *packed_t_warp,
                                        int precision, const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk,
                                        void *workspace, float *grad_coarse, float *grad_fine, float *grad_warp, float *loss, float *rgb,
                                        float *rgb_fine, snerf_stream_t stream, snerf_stream_t aux_stream);
SNERF_API int snerf_smpl_nerf_train_step_aux_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                       const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                       const snerf_warp_desc *desc_warp, float *packed_warp, float *packed_t_warp, int precision,
                                       const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk, void *workspace,
                                       float *grad_coarse, float *grad_fine, float *grad_warp, float *loss, float *rgb, float *rgb_fine,
                                       const snerf_adam_state *adam, const snerf_adam_range *ranges_host, int n_ranges,
                                       const snerf_adam_net *nets_host, int n_nets, int64_t warp_param_offset, snerf_comm_t comm,
                                       snerf_stream_t stream, snerf_stream_t aux_stream);

#ifdef __cplusplus
}
#endif
#endif /* SMPLNERF_H */
