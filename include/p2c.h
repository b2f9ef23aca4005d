/*
 * p2c.h -- C ABI of libp2c_hip.so: the MI355X (gfx950) pose-sequence hot path of pedestrians-video-2-carla.
 *
 * The reference (wielgosz-info/pedestrians-video-2-carla) is pure Python: there is no FFI to match. These entry
 * points are what a ctypes / cffi binding on the reference side would bind to replace the chains of small PyTorch ops
 * listed below (INTEGRATION.md shows the binding). Paths are relative to src/pedestrians_video_2_carla/ of the reference.
 *
 *   p2c_pose_head_fwd / _bwd   replaces, fused in one launch each:
 *        modules/movements/movements.py:105-118          rotation_6d_to_matrix on the model output
 *        modules/layers/projection.py:52-71              per-clip reference skeleton (O(B) python) -> skel_type index
 *        modules/layers/projection.py:170-195 + walker_control/p3d_pose.py:98-213    cumulative rotations + FK
 *        modules/layers/projection.py:125-136 + transforms/pose/normalization/reference_skeletons_denormalizer.py:67-91
 *        utils/world.py:16-63                            world transform from changes
 *        walker_control/p3d_pose_projection.py:115-152   pinhole projection
 *        transforms/pose/normalization/normalizer.py:20-41 (+ extractors)   dm.transform_callable on the projection
 *        loss/loc_2d.py:69-89, loss/loc_3d.py:12-40, loss/loc_2d_3d.py:6-17
 *   p2c_normalize_fwd / _bwd   transforms/pose/normalization/normalizer.py:20-41 stand-alone (any skeleton)
 *   p2c_loss2d_fwd / _bwd      loss/base_pose_loss.py:36-66 + loss/loc_2d.py:69-89 (autoencoder flow)
 *   p2c_remap_nodes            data/base/base_dataset.py:156-167 (_get_common_tensor, zero-filled joint scatter)
 *   p2c_mlp_fwd / _bwd         modules/movements/linear_ae/linear_ae.py:25-59 (the six nn.Linear + ReLU of LinearAE)
 *
 * Conventions: every pointer is a DEVICE pointer unless named host_*; tensors are dense row-major fp32; `stream` is a
 * hipStream_t passed as void*; nothing is allocated, freed or synchronised inside; no exception crosses the ABI.
 * Return value: 0 = launched, negative = argument error (P2C_E_*), positive = hipError_t of the failed launch.
 */
#ifndef P2C_H
#define P2C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(P2C_BUILD)
#define P2C_API __attribute__((visibility("default")))
#else
#define P2C_API
#endif

#define P2C_JOINTS 26            /* CARLA_SKELETON bones; pose-head tensors are (B, T, 26, k) */
#define P2C_SKELETON_TYPES 4     /* (adult,female) (adult,male) (child,female) (child,male) */

/* kind of the movements-model output handed to the pose head (modules/flow/output_types.py:4-20) */
enum {
  P2C_KIND_POSE_CHANGES_6D = 0,  /* y (B,T,26,6)   6-D rotation *changes*, orthonormalised in-kernel          */
  P2C_KIND_POSE_CHANGES_MAT = 1, /* y (B,T,26,3,3) rotation-matrix changes (the reference's API tensor)        */
  P2C_KIND_RELATIVE_ROT_6D = 2,  /* y (B,T,26,6)   relative rotations (no cumulative product)                  */
  P2C_KIND_RELATIVE_ROT_MAT = 3, /* y (B,T,26,3,3)                                                             */
  P2C_KIND_ABSOLUTE_LOC = 4      /* y (B,T,26,3)   absolute locations, hips-neck re-normalised to the skeleton */
};

/* data-module transform applied to the projection (data/base/base_transforms.py) */
enum { P2C_TRANSFORM_NONE = 0, P2C_TRANSFORM_HIPS_NECK = 1, P2C_TRANSFORM_BBOX = 2, P2C_TRANSFORM_HIPS_NECK_BBOX = 3 };

enum { P2C_E_NULL = -1, P2C_E_SHAPE = -2, P2C_E_ENUM = -3, P2C_E_INDEX = -4 };

typedef struct p2c_pose_head_desc {
  /* ---- shapes / modes ---- */
  int32_t B, T;                 /* clips, frames per clip */
  int32_t kind;                 /* P2C_KIND_* */
  int32_t transform;            /* P2C_TRANSFORM_* */
  int32_t t0, t1;               /* eval_slice [t0, t1) over frames: losses / transformed outputs only there */
  int32_t mask_missing_joints;  /* loc_2d: ignore joints whose gt is exactly (0,0) */
  int32_t hips_lane;            /* predicted joint never masked (tensors.py:33-38), -1 = none */
  int32_t n_hips, n_neck;       /* 1 or 2 joints averaged for the shift / scale points */
  int32_t hips_idx[2], neck_idx[2];
  int32_t gt2d_joints, gt2d_channels;   /* gt2d is (B,T,gt2d_joints,gt2d_channels), channels >= 2 */
  int32_t gt3d_joints;                  /* gt3d is (B,T,gt3d_joints,3) */
  int32_t gmap2d[P2C_JOINTS];   /* per predicted joint: index of its gt joint, -1 = not a common joint */
  int32_t gmap3d[P2C_JOINTS];
  int32_t n_common2d, n_common3d;       /* number of gmap entries >= 0 (loss denominators) */
  int32_t world_absolute;               /* 0: dloc/drot are per-frame changes (utils/world.py:16-63); 1: they are the world
                                           location / rotation of each frame (projection.py:215-226) */
  float cam_f, cam_cx, cam_cy, cam_dist, cam_elev;
  float near_zero;              /* 1e-5 */
  /* ---- inputs ---- */
  const float *y;               /* model output, layout by kind */
  const int32_t *skel_type;     /* (B) in [0,4) */
  const float *ref_rel_loc;     /* (4,26,3)   reference skeleton tables (data/carla/reference.py) */
  const float *ref_rel_rot;     /* (4,26,3,3) */
  const float *ref_hn_shift;    /* (4,3) hips of the reference absolute pose  (absolute_loc kind) */
  const float *ref_hn_scale;    /* (4)   |neck-hips| of the reference absolute pose */
  const float *dloc;            /* (B,T,3)   world location changes (or locations) or NULL (= ZeroTrajectory) */
  const float *drot;            /* (B,T,3,3) world rotation changes or NULL */
  const float *gt2d;            /* targets['projection_2d_transformed'] (or 'projection_2d'); NULL = no loc_2d */
  const float *gt3d;            /* targets['absolute_pose_loc']; NULL = loc_3d unavailable */
  /* ---- outputs ---- */
  float *partials;              /* workspace, p2c_pose_head_workspace_floats(B) floats */
  float *loss_sums;             /* (4): sum_sq_2d, n_unmasked_2d, sum_sq_3d, n_elems_3d */
  float *losses;                /* (3): loc_2d, loc_3d, loc_2d_3d   (NaN when the gt is absent) */
  float *final_rel_rot;         /* (B,26,3,3) last relative rotation, consumed by the backward; pose_changes kinds */
  /* optional materialised tensors (eval / predict); NULL = not written */
  float *out_pose_changes;      /* (B,T,26,3,3) */
  float *out_projection_2d;     /* (B,T,26,3) */
  float *out_projection_2d_transformed; /* (B,T,26,3) frames outside [t0,t1) untouched */
  float *out_shift;             /* (B,T,2) */
  float *out_scale;             /* (B,T) */
  float *out_relative_pose_loc; /* (B,T,26,3) */
  float *out_relative_pose_rot; /* (B,T,26,3,3) */
  float *out_absolute_pose_loc; /* (B,T,26,3) */
  float *out_absolute_pose_rot; /* (B,T,26,3,3) */
  float *out_world_loc;         /* (B,T,3) */
  float *out_world_rot;         /* (B,T,3,3) */
  /* 1 = the caller guarantees that p2c_pose_head_bwd follows with the same desc before anyone reads `losses` / `loss_sums`
   * (a captured train step): for the time-parallel 6-D kernels the forward then skips the one-workgroup finalize launch
   * and the backward kernel finishes the loss reduction itself (same values to fp32 rounding). 2 = same guarantee, and
   * the forward call only counts the unmasked target pairs: the backward kernel, which recomputes the pose head anyway,
   * produces the losses as well as grad_y (+ the finalize launch). Ignored for every other kernel family. */
  int32_t defer_loss_finalize;
  /* ---- rot_3d (loss/rot_3d.py:9-37), 6-D kinds only (P2C_E_ENUM otherwise) ----
   * gt_rot: targets['absolute_pose_rot'] (B,T,gt3d_joints,3,3) or NULL. When set, the forward also sums the squared difference
   * between the absolute joint rotations of the kinematic chain and gt_rot over the joints gmap3d maps and the frames [t0,t1):
   * loss_sums must hold 6 floats ([4] = sum_sq_rot, [5] = n_elems_rot) and losses 4 ([3] = rot_3d = their quotient); nothing
   * else is written (no absolute_pose_rot tensor). The backward adds 2 (A - gt) grad_loss_rot / n as a torque in its
   * tangent-space pass. The joint-lane kernels run (time-parallel or clip-sequential, defer_loss_finalize ignored). */
  const float *gt_rot;
  const float *grad_loss_rot;   /* p2c_pose_head_bwd: device scalar, upstream gradient of rot_3d; NULL = none */
} p2c_pose_head_desc;

/* library / build identification: "p2c-hip <version> gfx950" */
P2C_API const char *p2c_version(void);

/* number of floats `partials` must hold for a batch of B clips */
P2C_API int64_t p2c_pose_head_workspace_floats(int32_t B);

/* Kernel selection for the 6-D kinds with lean outputs: batches of at most `max_b` clips (default 2048, or the
 * environment variable P2C_TP_MAX_B) run the time-parallel kernels (one workgroup per clip, one 32-lane group per frame,
 * T <= 32), larger ones the clip-sequential kernels. Both compute the same function (fp32 rounding order differs in the
 * cumulative rotation product). max_b < 0 only queries. Returns the previous value. */
P2C_API int p2c_pose_head_set_time_parallel_max_batch(int32_t max_b);
/* Experimental, off by default (min_b = 2^30; env P2C_PK_MIN_B): batches of at least `min_b` clips of the same
 * configuration without world motion run the packed-fp32 clip-sequential kernels (two clips per lane, v_pk_fma_f32) --
 * parity-tested, currently slower than the scalar kernels. min_b < 0 only queries. */
P2C_API int p2c_pose_head_set_packed_min_batch(int32_t min_b);
/* Batches of at least `min_b` clips (default 8192, env P2C_CHAIN_MIN_B) that the time-parallel kernels do not take run the
 * chain-lane kernels (csrc/p2c_pose_head_chain.hip: eight clips per wavefront, a lane owns up to four consecutive bones)
 * when the configuration is the training one: 6-D kind, lean outputs, no world motion, no external gradients, targets in
 * the CARLA joint layout with two channels. Otherwise, and below min_b, the joint-lane kernels run. Same function, fp32
 * rounding order differs in the kinematic chain. min_b < 0 only queries. Returns the previous value. */
P2C_API int p2c_pose_head_set_chain_min_batch(int32_t min_b);

/* Forward: fills loss_sums, losses, final_rel_rot and any non-NULL out_* tensor. Two launches on `stream`
 * (pose head + deterministic reduction of the per-wave partial sums); with desc->defer_loss_finalize = 1 / 2 one launch
 * (no reduction / only the per-clip count of unmasked target pairs) -- see the field's comment. */
P2C_API int p2c_pose_head_fwd(const p2c_pose_head_desc *desc, void *stream);
/* Measurement hook (bench.py, rocprofv3 cross-check): the launches of p2c_pose_head_fwd one at a time. which = 1: the
 * pose-head kernel alone, 2: the one-workgroup loss reduction alone, 3: both (= p2c_pose_head_fwd). */
P2C_API int p2c_pose_head_fwd_launch(const p2c_pose_head_desc *desc, int32_t which, void *stream);

/* Backward (recompute): grad_y has the layout of desc->y. `grad_losses` = host array of three device pointers (each
 * NULL = no gradient, or one float): the upstream gradients of (loc_2d, loc_3d, loc_2d_3d) -- for a 3-vector gradient g
 * pass {g, g+1, g+2}; NULL array = all zero. desc->loss_sums and desc->final_rel_rot must hold the forward's values.
 * Optional upstream gradients of materialised outputs (NULL = none): grad_absolute_pose_loc (B,T,26,3),
 * grad_projection_2d_transformed (B,T,26,3) [channel 2 ignored], grad_absolute_pose_rot (B,T,26,3,3) [6-D kinds only:
 * rot_3d-type losses, reference loss/rot_3d.py; P2C_E_ENUM for the matrix kinds]. One launch (two with
 * desc->defer_loss_finalize = 2: the kernel that also sums the losses + their final reduction). */
P2C_API int p2c_pose_head_bwd(const p2c_pose_head_desc *desc, const float *const grad_losses[3],
                      const float *grad_absolute_pose_loc, const float *grad_projection_2d_transformed,
                      const float *grad_absolute_pose_rot, float *grad_y, void *stream);

/* Stand-alone normaliser (Normalizer.__call__, dim = 2 or 3) over N frames of J joints with C channels (C >= dim).
 * out (N,J,C), shift (N,dim), scale (N); shift/scale may be NULL. */
P2C_API int p2c_normalize_fwd(const float *x, float *out, float *shift, float *scale, int64_t N, int32_t J, int32_t C,
                      int32_t dim, int32_t transform, int32_t n_hips, const int32_t *host_hips_idx, int32_t n_neck,
                      const int32_t *host_neck_idx, float near_zero, void *stream);
P2C_API int p2c_normalize_bwd(const float *x, const float *grad_out, float *grad_x, int64_t N, int32_t J, int32_t C,
                      int32_t dim, int32_t transform, int32_t n_hips, const int32_t *host_hips_idx, int32_t n_neck,
                      const int32_t *host_neck_idx, float near_zero, void *stream);

/* Masked 2-D MSE (Loc2DPoseLoss): pred (N,Jp,Cp), gt (N,Jg,Cg); K common joints, host index lists; hips_col = position
 * in the common list that is never masked or -1. loss_sums (2) = sum_sq, n_unmasked; loss (1). */
P2C_API int64_t p2c_loss2d_workspace_floats(int64_t N);
P2C_API int p2c_loss2d_fwd(const float *pred, const float *gt, int64_t N, int32_t Jp, int32_t Cp, int32_t Jg, int32_t Cg,
                   int32_t K, const int32_t *host_pred_idx, const int32_t *host_gt_idx, int32_t hips_col,
                   int32_t mask_missing_joints, float *partials, float *loss_sums, float *loss, void *stream);
P2C_API int p2c_loss2d_bwd(const float *pred, const float *gt, int64_t N, int32_t Jp, int32_t Cp, int32_t Jg, int32_t Cg,
                   int32_t K, const int32_t *host_pred_idx, const int32_t *host_gt_idx, int32_t hips_col,
                   int32_t mask_missing_joints, const float *loss_sums, const float *grad_loss, float *grad_pred,
                   void *stream);

/* Zero-filled joint remap: dst[n, dst_idx[k], :] = src[n, src_idx[k], :], every other dst joint = 0. */
P2C_API int p2c_remap_nodes(const float *src, float *dst, int64_t N, int32_t Jsrc, int32_t Jdst, int32_t C, int32_t K,
                    const int32_t *host_src_idx, const int32_t *host_dst_idx, void *stream);

/* ---- fused small MLP (LinearAE: modules/movements/linear_ae/linear_ae.py:25-59) on fp32 MFMA -------------------------
 * y = W_{L-1} relu( ... relu(W_0 x + b_0) ... ) + b_{L-1} over N rows; dims[l] -> dims[l+1], every width <= 159.
 * Forward: x (N,dims[0]) -> y (N,dims[L]). Backward (activations recomputed): gy (N,dims[L]) -> gW[l] (dims[l+1],dims[l]),
 * gb[l] (dims[l+1]) -- WRITTEN, not accumulated -- through `partials` (p2c_mlp_workspace_floats floats) and a fixed-order
 * reduction (bitwise reproducible). x receives no gradient (the flows feed data). Two or three launches: per-workgroup
 * partial gradient tiles + reduction, or -- around one 16-row tile per CU -- activation / gradient factors + a contraction
 * over all rows + reduction (environment P2C_MLP_WGRAD=fused|split overrides the choice); the workspace size covers both. */
#define P2C_MLP_MAX_LAYERS 8
/* operand precision of the MLP's matrix products (accumulation is always fp32):
 *   F32    v_mfma_f32_16x16x4_f32, bit-for-bit an fmaf chain (default; what every parity claim is made with)
 *   BF16   operands rounded to bf16, v_mfma_f32_16x16x16_bf16: 1/8 of the MFMA cycles, ~2^-9 relative error per product
 *   BF16X3 split-bf16 (hi + lo, three MFMAs): ~fp32-grade products at 3/8 of the cycles
 * The reduced-precision arms exist for the LinearAE shapes with compile-time geometry (BASELINE.json configs[1] names bf16);
 * any other shape with precision != F32 returns P2C_E_ENUM. */
enum { P2C_PREC_F32 = 0, P2C_PREC_BF16 = 1, P2C_PREC_BF16X3 = 2 };
struct p2c_adamw_desc;
typedef struct p2c_mlp_desc {
  int32_t n_layers;
  int32_t dims[P2C_MLP_MAX_LAYERS + 1];
  int64_t N;
  const float *x;
  const float *W[P2C_MLP_MAX_LAYERS];   /* row-major (out, in), as nn.Linear.weight */
  const float *b[P2C_MLP_MAX_LAYERS];
  float *y;                             /* forward output */
  const float *gy;                      /* backward input */
  float *gW[P2C_MLP_MAX_LAYERS];
  float *gb[P2C_MLP_MAX_LAYERS];
  float *partials;                      /* backward workspace, p2c_mlp_workspace_floats floats */
  float *w_image;                       /* p2c_mlp_image_floats floats: packed weights, WRITTEN by p2c_mlp_fwd and
                                           read by p2c_mlp_bwd (same weights: call bwd before the optimizer) */
  /* optional, p2c_mlp_bwd only: apply this optimizer step to the MLP's parameters inside the gradient reduction (single-GPU
   * training, no all-reduce in between): gW/gb must be views of fused_adamw->grad, the MLP must be all of its n parameters;
   * w_image (if set) is refreshed with the new weights; with fused_adamw->zero_grad set, gW / gb are left ZEROED (as
   * p2c_adamw_step leaves them) instead of holding the gradient. Host pointer, read during the call. */
  const struct p2c_adamw_desc *fused_adamw;
  int32_t skip_pack;                    /* 1 = w_image is already current (kept so by p2c_mlp_pack + the optimizer's
                                           scatter, see p2c_adamw_desc): p2c_mlp_fwd does not launch the pack kernel */
  float *saved;                         /* optional, p2c_mlp_saved_floats floats (0 = not worth it at this N: pass NULL): the
                                           forward leaves the hidden activations there and the backward of the SAME call pair
                                           loads them instead of recomputing (bit-identical results) */
  int32_t precision;                    /* P2C_PREC_*: same value for the forward and the backward of a step */
} p2c_mlp_desc;
P2C_API int64_t p2c_mlp_image_floats(const p2c_mlp_desc *desc);
/* writes the packed image from the current weights (what p2c_mlp_fwd does first unless skip_pack) */
P2C_API int p2c_mlp_pack(const p2c_mlp_desc *desc, void *stream);
/* HOST array out: index[i] = float offset inside the image of parameter i, parameters counted in the order
 * W_0 (row-major), b_0, W_1, b_1, ... (n = sum of n_out * (n_in + 1)); returns n, or a negative P2C_E_* code */
P2C_API int64_t p2c_mlp_image_index(const p2c_mlp_desc *desc, int32_t *index, int64_t capacity);
P2C_API int64_t p2c_mlp_workspace_floats(const p2c_mlp_desc *desc);
P2C_API int64_t p2c_mlp_saved_floats(const p2c_mlp_desc *desc);
P2C_API int p2c_mlp_fwd(const p2c_mlp_desc *desc, void *stream);
P2C_API int p2c_mlp_bwd(const p2c_mlp_desc *desc, void *stream);

/* ---- K13: the small-batch train step of LitPoseLiftingFlow(LinearAE) in two launches ------------------------------------
 * Replaces, for one optimisation step on a batch of B clips of T <= 16 frames (one clip = one 16-sample MFMA tile):
 *   modules/flow/base.py:397-410 (_step) -> modules/flow/pose_lifting.py:121-144 (_inner_step)
 *     modules/movements/linear_ae/linear_ae.py:50-59 (LinearAE.forward, 6-D rotation output: 52-26-13-6-39-78-156)
 *     everything p2c_pose_head_fwd / _bwd replace (projection layer, transform_callable, loc_2d / loc_3d / loc_2d_3d)
 *   the backward of all of it, and -- with mlp.fused_adamw -- torch.optim.AdamW.step (flow/base_model.py:156-158).
 * Launch 1: one workgroup per clip (LinearAE forward -> pose head forward + backward -> dgrad chain; the model output and
 * its gradient never leave LDS), leaving per-clip loss sums and weight-gradient factors in `mlp.partials`; launch 2: one
 * workgroup per 16x16 weight-gradient tile contracts the factors over all clips in a fixed order, writes gW/gb, applies
 * the optimizer step and refreshes mlp.w_image, and finishes the loss reduction. Same arithmetic, same summation orders
 * as the separate entry points: results are bitwise reproducible, and bit-identical to p2c_mlp_fwd -> p2c_pose_head_* ->
 * p2c_mlp_bwd where that path runs its split weight gradient.
 * head: y / final_rel_rot / out_* unused (out_* must be NULL); kind POSE_CHANGES_6D or RELATIVE_ROT_6D; partials = B*4
 * floats; losses / loss_sums are written by launch 2. mlp: x (B*T, 52), W / b (for p2c_mlp_pack unless skip_pack),
 * w_image, gW / gb (written), partials = p2c_train_step_workspace_floats floats, optional fused_adamw; y / gy / saved unused.
 * pair_counts: (B) floats from p2c_count_target_pairs on the SAME targets (a property of the batch: computed when the batch
 * is staged, not per step). grad_losses: as p2c_pose_head_bwd. */
typedef struct p2c_train_step_desc {
  p2c_pose_head_desc head;
  p2c_mlp_desc mlp;
  const float *pair_counts;
} p2c_train_step_desc;
P2C_API int p2c_train_step_supported(const p2c_train_step_desc *desc);          /* 1 = shapes / kind the kernels cover */
P2C_API int64_t p2c_train_step_workspace_floats(const p2c_train_step_desc *desc);
P2C_API int p2c_train_step(const p2c_train_step_desc *desc, const float *const grad_losses[3], void *stream);
/* Measurement hook: the same call with only some of its launches -- which = 1: the per-clip kernel, 2: the weight-gradient /
 * optimizer / loss kernel (on the factors a previous full call left in the workspace; each launch re-applies the optimizer
 * step), 3: both (= p2c_train_step). bench.py times the launches one by one with it. */
P2C_API int p2c_train_step_launch(const p2c_train_step_desc *desc, const float *const grad_losses[3], int32_t which,
                                  void *stream);
/* The first launch has two forms: one workgroup of eight wavefronts per clip (latency form, up to one clip per CU) and a
 * pair of wavefronts per clip, four pairs sharing a workgroup's weight image (throughput form, csrc/p2c_train_stream.hip), taken
 * from min_b clips on when the descriptor allows it (identity joint maps, CARLA targets, no world motion). Returns the previous threshold;
 * min_b < 0 only queries. Default 257 = more than one clip per CU (env P2C_STREAM_MIN_B). Both leave the same factor blocks for the second launch. */
P2C_API int p2c_train_step_set_stream_min_batch(int32_t min_b);
/* The second launch has two forms as well: a workgroup per dW tile and clip slice with the combine, the optimizer and the loss
 * reduction in the same launch (few hundred clips), and -- from min_b clips on (default 3072, env P2C_WGRAD_STREAM_MIN_B) -- a
 * persistent workgroup per clip slice that reads every factor block once and holds all 79 tiles, followed by a third launch
 * that adds the per-workgroup partials in a fixed order (+ optimizer + losses). Returns the previous threshold. */
P2C_API int p2c_train_step_set_wgrad_stream_min_batch(int32_t min_b);
/* counts[b] = number of (frame, joint) pairs of clip b inside [t0, t1) whose 2-D target the loss does not mask
 * (utils/tensors.py:29-40 via loss/base_pose_loss.py:36-66): reads only the target-side fields of desc (gt2d, gmap2d,
 * hips_lane, mask_missing_joints, t0, t1); y / partials / losses may be NULL. One launch. */
P2C_API int p2c_count_target_pairs(const p2c_pose_head_desc *desc, float *counts, void *stream);

/* ---- grouped per-joint embeddings (K7a) --------------------------------------------------------------------------------
 * Replaces the loop over 26 nn.Linear(2, 64) of Seq2SeqEmbeddings._format_input (modules/movements/seq2seq/
 * seq2seq_embeddings.py:53-78): y[(t', b), j, :] = W_j x[b, t, j, :] + b_j with t' = t (or T-1-t when `flip`, the
 * reference's invert_sequence), written sequence-first as the LSTM wants it. x (B,T,J,C), C <= 4; y (T,B,J,E), E % 4 == 0.
 * Joint j's weight (E,C) starts at W + j*w_stride, its bias at b + j*b_stride (floats): stacked tensors and views of a
 * flat parameter buffer are both addressed without a copy. Backward: gW / gb with the same strides (x gets no gradient:
 * it is data); partials = p2c_embed_workspace_floats floats; fixed-order two-stage reduction (bitwise reproducible). */
P2C_API int64_t p2c_embed_workspace_floats(int32_t B, int32_t T, int32_t J, int32_t C, int32_t E);
P2C_API int p2c_embed_fwd(const float *x, const float *W, const float *b, int64_t w_stride, int64_t b_stride, float *y,
                  int32_t B, int32_t T, int32_t J, int32_t C, int32_t E, int32_t flip, void *stream);
P2C_API int p2c_embed_bwd(const float *x, const float *gy, int64_t w_stride, int64_t b_stride, float *gW, float *gb,
                  float *partials, int32_t B, int32_t T, int32_t J, int32_t C, int32_t E, int32_t flip, void *stream);

/* ---- folded input map (K7a') ------------------------------------------------------------------------------------------
 * Seq2SeqEmbeddings feeds concat_j(W_j x_j + b_j) into the encoder LSTM's first input projection (seq2seq_embeddings.py:
 * 53-78 -> seq2seq.py:36-58) with nothing non-linear in between; the train step composes the two linear maps instead of
 * materialising the (T,B,J*E) embedding tensor:
 *   w_eff[g, j*C + c] = sum_e w_ih[g, j*E + e] W_j[e, c]        b_eff[g] = b_ih[g] + b_hh[g] + sum_{j,e} w_ih[g, j*E+e] b_j[e]
 * w_ih (G, J*E); W_j / b_j addressed with strides as in p2c_embed_*; b_ih / b_hh (G) optional. Backward: from g_eff
 * (G, J*C) and g_b (G) it WRITES (accumulate_w = 0) or ADDS TO g_w_ih (G, J*E), ADDS to gW / gb (the strides of W / b)
 * and, when given, ADDS g_b to g_b_ih / g_b_hh. One launch each, fixed summation order. C <= 4. */
P2C_API int p2c_fold_fwd(const float *w_ih, const float *W, const float *b, int64_t w_stride, int64_t b_stride,
                 const float *b_ih, const float *b_hh, float *w_eff, float *b_eff, int32_t G, int32_t J, int32_t E,
                 int32_t C, void *stream);
P2C_API int p2c_fold_bwd(const float *w_ih, const float *W, const float *b, int64_t w_stride, int64_t b_stride,
                 const float *g_eff, const float *g_b, float *g_w_ih, int32_t accumulate_w, float *gW, float *gb,
                 float *g_b_ih, float *g_b_hh, int32_t G, int32_t J, int32_t E, int32_t C, void *stream);

/* ---- LSTM recurrence (K7b) ---------------------------------------------------------------------------------------------
 * The time loop of one torch.nn.LSTM layer (gate order i, f, g, o; reference seq2seq.py:36-58 Encoder / Decoder):
 *   gates[t] = gx[t] + h[t-1] W_hh^T ;  c[t] = f c[t-1] + i g ;  h[t] = o tanh(c[t])
 * with gx[t] = x[t] W_ih^T + b_ih + b_hh computed by the caller (one dense library GEMM for all t). H in {16,32,48,64,96,128}
 * (above 64 the W_hh image is staged through LDS in two chunks),
 * all tensors fp32 row-major, 16-byte aligned. Forward fills out (T,B,H), optional hT/cT (B,H) and the saved activations
 * acts (T,B,4H) / cs (T,B,H). Backward takes g_out / g_hT / g_cT (each optional), acts, cs, c0, w_hh and writes
 * g_gx (T,B,4H) = d gates (from which the caller forms dW_hh = sum_t g_gx[t]^T h[t-1], dW_ih, db with library GEMMs)
 * and optional g_h0 / g_c0. One launch each. B <= 2^20 (per-step rows are addressed through 32-bit buffer offsets); up to
 * B = 4096 the launch tiles 4 sequences per workgroup (v_mfma_f32_4x4x1_16B), above that 16 (v_mfma_f32_16x16x4);
 * P2C_REC_TILE=wide|narrow in the environment forces one. */
typedef struct p2c_lstm_desc {
  int32_t T, B, H;
  const float *gx;              /* (T,B,4H) */
  const float *h0, *c0;         /* (B,H) or NULL = zeros */
  const float *w_hh;            /* (4H,H) */
  float *out;                   /* (T,B,H) */
  float *hT, *cT;               /* (B,H) or NULL */
  float *acts, *cs;             /* (T,B,4H), (T,B,H): written by fwd (may be NULL for inference), read by bwd */
  const float *g_out;           /* (T,B,H) or NULL */
  const float *g_hT, *g_cT;     /* (B,H) or NULL */
  float *g_gx;                  /* (T,B,4H) */
  float *g_h0, *g_c0;           /* (B,H) or NULL */
  /* optional (zero-initialise the struct): */
  const float *bias_a, *bias_b; /* (4H) or NULL: fwd adds them to gx as it reads it (b_ih, b_hh), so the projection GEMM runs bias-free */
  float *g_gx_bt;               /* (B,T,4H) or NULL: bwd writes a second, batch-first copy of g_gx (pairs with a batch-first layer input) */
  int32_t gx_bt;                /* fwd: gx is laid out (B,T,4H) -- the projection of a batch-first input, no permute copy */
  /* nn.LSTM's inter-layer dropout on this layer's output, drawn inside the kernels (drop_state != NULL; the protocol of the four
   * words: p2c_decoder_desc): fwd writes out_drop (T,B,H) = out * mask beside the raw `out` (fwd without out_drop draws nothing),
   * bwd takes g_out as the gradient of out_drop. */
  float *out_drop;
  int32_t *drop_state;
  float drop_p;
  int32_t drop_site;
} p2c_lstm_desc;
P2C_API int p2c_lstm_rec_fwd(const p2c_lstm_desc *desc, void *stream);
P2C_API int p2c_lstm_rec_bwd(const p2c_lstm_desc *desc, void *stream);

/* ---- Seq2Seq decoder loop (K7c) -----------------------------------------------------------------------------------------
 * for t in range(T): out_t = fc(LSTM_2layers(x_t; encoder state)); x_{t+1} = out_t   (reference seq2seq.py:245-349; the
 * decoder state is NOT carried between frames, 272-288). The caller provides the frame-invariant recurrent terms
 * k_l = b_ih_l + b_hh_l + W_hh_l hidden_l (B,4H) and the encoder cell states c_l (B,H). H = 64, O <= 64. Forward writes
 * out (T,B,O) and the saved activations; backward consumes g_out (T,B,O) and writes d gates0 / d gates1 (T,B,4H),
 * d out_total (T,B,O) [= g_out + the gradient that flows back through the fed-back input], and d c_l (B,H): the weight
 * gradients are dense reductions of those over all (t,b) (dW_ih0 = dgates0^T x_prev, dW_ih1 = dgates1^T h0d,
 * dW_fc = dout_total^T h1, dk_l = sum_t dgates_l, db_fc = sum dout_total), left to library GEMMs. One launch each. */
typedef struct p2c_decoder_desc {
  int32_t T, B, H, O;
  const float *k0, *c0, *k1, *c1;
  const float *w_ih0, *w_ih1, *w_fc, *b_fc;   /* (4H,O), (4H,H), (O,H), (O) */
  const float *x0;                /* (B,O) first input, NULL = zeros (<sos>) */
  const float *drop;              /* (T,B,H) multiplicative dropout mask applied to the layer-0 output, or NULL */
  float *out;                     /* (T,B,O) */
  float *acts0, *acts1;           /* (T,B,4H) activated gates, written by fwd, read by bwd */
  float *h0d, *h1;                /* (T,B,H) layer outputs (after dropout for layer 0), written by fwd */
  const float *g_out;             /* (T,B,O) */
  float *g_gates0, *g_gates1;     /* (T,B,4H) */
  float *g_outtot;                /* (T,B,O) */
  float *g_c0, *g_c1;             /* (B,H) */
  /* optional (zero-initialise the struct): the frame-invariant terms formed inside the launch. With hid0 != NULL,
   * k_l = b{l}a + b{l}b + hid_l w_hh{l}^T is computed by the forward itself from the encoder hidden states hid_l (B,H),
   * decoder.rnn.weight_hh_l{l} (4H,H) and the two bias vectors (4H; either may be NULL); k0 / k1 are then ignored and
   * kw0 / kw1 (B,4H) are scratch the 16-clip tiling (B > 4096) writes k_l to. out_bt: a second copy of out laid out
   * (B,T,O), the layout the model returns (saves the permute copy). Backward: g_out_bt != 0 says g_out is laid out (B,T,O);
   * g_k0 / g_k1 (B,4H) receive sum_t d gates_l (= d k_l), g_hid0 / g_hid1 (B,H) receive d hid_l = g_k_l w_hh{l} (both or
   * neither; they need w_hh{l} and, above B = 4096, g_k0 / g_k1). */
  const float *hid0, *hid1, *w_hh0, *w_hh1, *b0a, *b0b, *b1a, *b1b;
  float *kw0, *kw1;
  float *out_bt;
  float *g_k0, *g_k1, *g_hid0, *g_hid1;
  int32_t g_out_bt;
  /* teacher forcing (seq2seq.py:272-288,323-349), both or neither: force (T,B) 0 / 1 and target (T,B,O). Where force[t][b] != 0
   * frame t of clip b -- the stored output AND the next step's input -- is target[t][b]; its rows of d out_total are zero (the
   * reference writes the targets into the output tensor itself: those rows carry neither loss nor gradient). */
  const float *force, *target;
  /* the dropout mask drawn inside the kernels instead of read from `drop` (drop == NULL, drop_state != NULL): keep(e) =
   * hash(seed, step, site, element e of the (T,B,H) mask) >= drop_p 2^32 (csrc/p2c_rec_dev.h: drop_value), mask = keep / (1 - drop_p). drop_state: 4 int32 words
   * on the device {seed_lo, seed_hi, step, next}; fwd reads step and leaves next = step + 1, bwd reads next - 1 and leaves
   * step = next, so the two launches of a step draw the same mask and a replayed graph advances by itself. */
  int32_t *drop_state;
  float drop_p;
  int32_t drop_site;
} p2c_decoder_desc;
P2C_API int p2c_decoder_fwd(const p2c_decoder_desc *desc, void *stream);
P2C_API int p2c_decoder_bwd(const p2c_decoder_desc *desc, void *stream);

/* ---- validation metrics on device -------------------------------------------------------------------------
 * p2c_eval_pose3d: MPJPE (metrics/mpjpe.py:29-45) and MRPE (metrics/mrpe.py:38-76) of one batch, ADDED to `state`
 * (4 doubles, device, zeroed by the caller at reset): state[0] += sum over clips of mean_{t,j} |pred - gt| over the common
 * joints, state[1] += B, state[2] += sum over clips of mean_t |(world_pred + hips_pred) - (world_gt + hips_gt)|,
 * state[3] += B (only when the world locations are given). pred (B,T,26,3); gt (B,T,Jg,3); gmap[26] = gt joint of
 * prediction joint j or -1; world_* = ABSOLUTE world locations (B,T,3) or both NULL. Metres in, metres out (compute() =
 * 1000 * sum / count). partials: p2c_eval_workspace_floats(B) floats.
 * p2c_eval_pck: PCK (metrics/pck.py:66-98): state[0] += correct, state[2] += total. pred (N,Jp,Cp), gt (N,Jg,Cg), N = B*T,
 * Jg <= 64; pmap[Jg] = prediction joint paired with gt joint i or -1; mask_src = targets['projection_2d'] (N,Jg,Cg) or NULL
 * (= gt); hips_joint = gt joint that is never masked or -1; norm_mode 0 = bounding-box diagonal, 1 = |neck - hips| of the
 * gt frame. partials: p2c_eval_workspace_floats(N) floats. */
P2C_API int64_t p2c_eval_workspace_floats(int64_t units);
P2C_API int p2c_eval_pose3d(const float *pred, const float *gt, int32_t B, int32_t T, int32_t Jg, const int32_t *gmap,
                    const int32_t *pred_hips, int32_t n_pred_hips, const int32_t *gt_hips, int32_t n_gt_hips,
                    const float *world_pred, const float *world_gt, float *partials, double *state, void *stream);
P2C_API int p2c_eval_pck(const float *pred, const float *gt, const float *mask_src, int64_t N, int32_t Jp, int32_t Cp,
                 int32_t Jg, int32_t Cg, const int32_t *pmap, int32_t mask_missing, int32_t hips_joint, int32_t norm_mode,
                 const int32_t *hips_idx, int32_t n_hips, const int32_t *neck_idx, int32_t n_neck, float threshold,
                 float near_zero, float *partials, double *state, void *stream);

/* ---- fused AdamW / Adam over one flat fp32 buffer ------------------------------------------------------------------
 * Replaces torch.optim.AdamW.step() as configured by the reference (modules/flow/base_model.py:156-158) when all
 * trainable parameters live in one flat buffer. Update rule = torch/optim/adamw.py (amsgrad=False, maximize=False):
 *   g *= grad_scale;  p -= lr*wd*p (adamw) | g += wd*p (adam);  m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2;
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = *step + 1;  *step = t. One launch, graph-capturable. */
typedef struct p2c_adamw_desc {
  int64_t n;                 /* parameters */
  float *param, *grad, *exp_avg, *exp_avg_sq;   /* device, n floats each, 16-byte aligned */
  float *step;               /* device scalar: steps taken so far; incremented by the call */
  int32_t *ticket;           /* device int, zero-initialised once by the caller (workgroup completion counter) */
  const float *hyper;        /* device, 6 floats: lr, beta1, beta2, eps, weight_decay, grad_scale */
  int32_t adamw;             /* 1 = decoupled weight decay (AdamW), 0 = L2 penalty (Adam) */
  int32_t zero_grad;         /* 1 = leave grad zeroed (replaces the next step's zero_grad memset) */
  /* optional (both or neither): after the update, param[i] is also written to scatter_dst[scatter_idx[i]] where the index
   * is >= 0 -- keeps a consumer's re-laid-out copy of the weights (the fused MLP's LDS image) current without a pack launch */
  const int32_t *scatter_idx;
  float *scatter_dst;
} p2c_adamw_desc;
P2C_API int p2c_adamw_step(const p2c_adamw_desc *desc, void *stream);

/* ---- K11: dataset-side input pipeline for a batch of clips, one launch -------------------
 * Replaces, per clip and on the CPU in the reference, BaseDataset.__getitem__ (data/base/base_dataset.py:206-234):
 *   Projection2DMixin.process_projection_2d   data/base/mixins/dataset/projection_2d_mixin.py:209-232
 *     AugmentPose.__call__                     transforms/pose/augmentation/augment_pose.py:43-76
 *       RandomFlip / RandomRotation            .../random_flip.py:39-76, .../random_rotation.py:34-68
 *     apply_deform                             projection_2d_mixin.py:137-171 (noise, per-joint missing mask)
 *     apply_transform x2                       projection_2d_mixin.py:177-189 -> normalizer.py:20-41 + extractors
 *   ConfidenceMixin.process_confidence         data/base/mixins/dataset/confidence_mixin.py:13-20
 *   BaseDataset._map_nodes                     data/base/base_dataset.py:156-190
 * The reference draws its random numbers inside these calls; here every draw is an input tensor so that the kernel is
 * a pure function of its arguments. Device pointers unless marked host; NULL = that step is off / that output is not
 * wanted. Errors mirror the reference's: return_confidence with a 2-channel pose and rotation with boxes derived from
 * a 3-channel pose both raise there (P2C_E_SHAPE here). */
typedef struct p2c_collate_desc {
  int64_t N;                   /* clips */
  int32_t T, Jd, C;            /* frames per clip, joints of the data skeleton (<= 64), channels: 2 (x,y) or 3 (+confidence) */
  const float *raw;            /* (N,T,Jd,C) */
  const uint8_t *is_flipped;   /* (N) 0/1: RandomFlip decision per clip; NULL = no flip augmentation */
  const int32_t *flip_perm;    /* host, Jd: Skeleton.get_flip_mask() of the data skeleton */
  const float *rotation_deg;   /* (N) RandomRotation angle per clip; NULL = no rotation augmentation */
  const float *bboxes;         /* (N,T,2,2) targets['bboxes'] or NULL (boxes of the raw pose, utils/tensors.py:12-26) */
  const float *clip_size;      /* (N,2) meta clip_width / clip_height (0 = unknown) or NULL */
  const float *noise;          /* (N,T,Jd,2) additive noise or NULL */
  const float *miss_u;         /* (N,T,Jd) uniform draws or NULL; joint j is dropped where miss_u < miss_prob[j] */
  const float *miss_prob;      /* host, Jd */
  int32_t transform;           /* P2C_TRANSFORM_* of the data module (NONE: frames = deformed pose) */
  int32_t n_hips, hips_idx[2], n_neck, neck_idx[2];   /* hips / neck points of the DATA skeleton (1 or 2 joints each) */
  float near_zero;             /* 1e-5 in the reference */
  int32_t return_confidence;   /* model needs_confidence: frames keep the confidence channel */
  int32_t Ji, K;               /* joints of the model-input skeleton; common joints (0 = same skeleton, Ji == Jd) */
  const int32_t *src_idx;      /* host, K: data joint ...                    (get_common_indices, skeleton.py:26-56) */
  const int32_t *dst_idx;      /* host, K: ... lands on this model-input joint */
  float *frames;               /* (N,T,Ji,return_confidence ? 3 : 2) model input */
  float *t_projection_2d;      /* (N,T,Ji,2) augmented pose, or NULL */
  float *t_deformed;           /* (N,T,Ji,2) projection_2d_deformed, or NULL */
  float *t_transformed;        /* (N,T,Ji,2) projection_2d_transformed, or NULL */
  float *shift, *scale;        /* (N,T,2), (N,T): projection_2d_shift / _scale, or NULL */
  float *bboxes_out;           /* (N,T,2,2) augmented boxes (needs bboxes), or NULL */
} p2c_collate_desc;
P2C_API int p2c_collate_fwd(const p2c_collate_desc *desc, void *stream);

/* ---- C (+)= A^T [B | 1] over K rows (weight + bias gradient of a dense layer) ------------------------------------------------
 * Replaces, in the backward of the Seq2Seq models (modules/movements/seq2seq/seq2seq.py:36-94: the nn.LSTM projections and
 * Decoder.fc_out under autograd), the split-K library GEMM dW = dY^T X and the column reduction db = sum_rows dY.
 * A (K, M) row-major with row pitch lda (= dY), B (K, N) with pitch ldb (= X), C (M, N) with pitch ldc; bias_out (M) or
 * NULL; accumulate: bit 0 = add to C instead of overwriting, bit 1 = the same for bias_out; workspace = p2c_atb_workspace_floats floats (partial
 * tiles of the K slices). Two launches, fixed summation order (bitwise reproducible). */
P2C_API int64_t p2c_atb_workspace_floats(int64_t K, int32_t M, int32_t N, int32_t with_bias);
P2C_API int p2c_atb(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t K, int32_t M, int32_t N, float *C,
                    int64_t ldc, float *bias_out, int32_t accumulate, float *workspace, void *stream);
/* The same with A's row k multiplied by a_scale[k / rows_per_scale] as it is loaded (NULL = p2c_atb): dY of a transformer
 * sub-layer whose output carried a per-sample stochastic-depth factor (PoseTransformer blocks behind
 * modules/movements/pose_former/pose_former.py:62-76), so that no scaled copy of dY is written first. */
P2C_API int p2c_atb_scaled(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t K, int32_t M, int32_t N, float *C,
                    int64_t ldc, float *bias_out, int32_t accumulate, const float *a_scale, int64_t rows_per_scale,
                    float *workspace, void *stream);

/* Grouped form: up to 8 independent problems behind ONE launch pair (the backward of a Seq2Seq layer stack is a row of
 * these contractions). Per problem the fields of p2c_atb; bias_out2 = a second vector that receives the same column sums
 * (bias_ih / bias_hh of an LSTM layer), needs bias_out. flags = the `accumulate` bits of p2c_atb. workspace =
 * p2c_atb_group_workspace_floats floats. Same kernels, same per-problem summation order as p2c_atb (bitwise equal results). */
typedef struct p2c_atb_problem {
  const float *a; int64_t a_stride;   /* (K, M) */
  const float *b; int64_t b_stride;   /* (K, N) */
  int64_t K;
  int32_t M, N;
  float *out; int64_t out_stride;     /* (M, N) */
  float *bias_out, *bias_out2;        /* (M) or NULL */
  int32_t flags;
} p2c_atb_problem;
P2C_API int64_t p2c_atb_group_workspace_floats(const p2c_atb_problem *problems, int32_t n);
P2C_API int p2c_atb_group(const p2c_atb_problem *problems, int32_t n, float *workspace, void *stream);

/* ---- grouped copy (batch staging of a captured step) ------------------------------------------------------------------------
 * dst[i][0:bytes[i]] = src[i][0:bytes[i]] for up to 24 device buffers in ONE launch: a HIP-graph train step reads fixed
 * addresses, so the tensors of every new batch the reference's trainer would hand to `training_step`
 * (modules/flow/base.py:231-246 `_step(batch, ...)`) are copied into the static buffers first. src / dst / bytes are HOST
 * arrays of n entries (device pointers inside). */
P2C_API int p2c_copy_group(const void *const *src, void *const *dst, const int64_t *bytes, int32_t n, void *stream);

/* Inspection: number of nodes / kernel nodes of a hipGraph_t (the trainer replays a captured step by calling its recorded
 * entry point directly when the graph holds nothing but that entry point's launches). */
P2C_API int p2c_graph_node_counts(void *graph, int32_t *n_total, int32_t *n_kernel);

/* ---- multi-head self-attention over short token sequences (K14) --------------------------------------------------------------
 * The attention of the build's PoseTransformer (reference modules/movements/pose_former/pose_former.py:33-76 binds the
 * third-party PoseTransformer: 26 joint tokens x 8 heads of 4 channels in the spatial blocks, 9 frame tokens x 8 heads of
 * 104 in the temporal ones). qkv (S, N, 3, heads, head_dim) = the qkv projection's output viewed; out (S, N, heads*head_dim)
 * = softmax(scale * q k^T) v per head, heads concatenated. Backward: from qkv and g_out (S, N, heads*head_dim) to g_qkv (the
 * layout of qkv); the probabilities are recomputed. One launch each; one workgroup per sequence with everything in LDS:
 * N <= 64, heads*head_dim % 4 == 0 and (4 N heads head_dim + 2 heads N^2) floats <= 156 KB (p2c_attn_small_supported). No
 * attention dropout, no mask. */
P2C_API int p2c_attn_small_supported(int32_t N, int32_t heads, int32_t head_dim);
P2C_API int p2c_attn_small_fwd(const float *qkv, float *out, float scale, int32_t S, int32_t N, int32_t heads, int32_t head_dim,
                       void *stream);
P2C_API int p2c_attn_small_bwd(const float *qkv, const float *g_out, float *g_qkv, float scale, int32_t S, int32_t N,
                       int32_t heads, int32_t head_dim, void *stream);

/* ---- LayerNorm over the last dimension of many short rows (K15) ----------------------------------------------------------------
 * torch.nn.LayerNorm(D) (biased variance, eps inside the root) as the build's PoseTransformer applies it (546 624 rows of 32,
 * 21 024 rows of 832; reference binding: modules/movements/pose_former/pose_former.py:33-76). x, y, gy, gx (rows, D)
 * row-major and 16-byte aligned, D % 4 == 0, D <= 1024; gamma / beta (D), any 4-byte alignment. Forward also writes mean and
 * rstd (rows) for the backward. Backward: gx (+ gx_add (rows, D) when not NULL: the gradient arriving over the residual
 * connection that branches off in front of a pre-norm layer, added in the same pass), and g_gamma / g_beta written
 * (accumulate = 0) or added to; partials = p2c_layernorm_workspace_floats floats; two launches, fixed summation order. */
P2C_API int p2c_layernorm_supported(int32_t D);
P2C_API int64_t p2c_layernorm_workspace_floats(int64_t rows, int32_t D);
P2C_API int p2c_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                      int64_t rows, int32_t D, float eps, void *stream);
P2C_API int p2c_layernorm_bwd(const float *x, const float *gamma, const float *mean, const float *rstd, const float *gy,
                      const float *gx_add, float *gx, float *g_gamma, float *g_beta, int32_t accumulate, float *partials,
                      int64_t rows, int32_t D, void *stream);

/* Learned weighted mean over F frame tokens: out (B, C) = sum_f w[f] x[b, f, c] + bias[0] (bias may be NULL) -- the forward of
 * PoseTransformer's weighted_mean = Conv1d(F, 1, kernel 1), bound at modules/movements/pose_former/pose_former.py:62-76. x (B, F, C)
 * and out (B, C) dense and 16-byte aligned, C % 4 == 0. One launch. */
P2C_API int p2c_frame_mean_fwd(const float *x, const float *w, const float *bias, float *out, int64_t B, int32_t F, int32_t C,
                       void *stream);

/* Testing aid (no counterpart in the reference): fills the LDS of every CU with NaN bit patterns, so that a kernel reading LDS it
 * never wrote fails deterministically instead of by what the previous workgroup left behind. One launch on `stream`. */
P2C_API int p2c_debug_poison_lds(void *stream);

/* ---- dense layers on fp32 MFMA with a fused epilogue (K16, csrc/p2c_gemm.hip) -----------------------------------------------------
 * C (M, N) = epilogue(A (M, K) * op(B)): trans_b = 1: B is (N, K) row-major -- y = x W^T, the forward of torch.nn.Linear as the
 * reference's model plugins use it (PoseTransformer qkv / proj / fc1 / fc2: modules/movements/pose_former/pose_former.py:62-76;
 * the input projections of the Seq2Seq LSTMs: movements/seq2seq/seq2seq.py:36-38,72-73); trans_b = 0: B is (K, N) row-major --
 * dx = dy W, its input gradient (the weight gradient dy^T x is p2c_atb). All row-major with leading dimensions in floats.
 * Epilogue, in this order: v = acc + bias[n] (bias may be NULL); act = 1: aux_out[m][n] = v if aux_out != NULL, then
 * v = gelu(v) (erf form, torch.nn.GELU()); act = 2: v *= gelu'(aux[m][n]) (the backward through that activation, aux = the
 * stored pre-activation); v *= row_scale[m / rows_per_scale] if row_scale != NULL (per-sample stochastic-depth factor);
 * v += residual[m][n] if residual != NULL; C[m][n] = v. C may alias residual. fp32 in, fp32 accumulate (an fmaf chain in k
 * order). 16-byte loads when every base / leading dimension allows, dword loads otherwise. Returns 0 or P2C_E_*. */
typedef struct p2c_gemm_desc {
  int32_t M, N, K, trans_b;
  const float *a; int64_t lda;
  const float *b; int64_t ldb;
  float *c; int64_t ldc;
  const float *bias;
  int32_t act, rows_per_scale;
  const float *aux; float *aux_out; int64_t ldaux;
  const float *row_scale;
  const float *residual; int64_t ldr;
} p2c_gemm_desc;
P2C_API int p2c_gemm(const p2c_gemm_desc *desc, void *stream);
/* The weight gradient of a WIDE dense layer, C (M, N) (+)= A^T B with A (K, M) and B (K, N) row-major over K = rows >> M, N
 * (dW = dy^T x; p2c_atb covers layers of up to ~128 features): the same MFMA tiles, K split into slices that each write a slab
 * of `workspace` (p2c_gemm_tn_workspace_floats(M, N, K) floats), added in slice order by a second launch -- bitwise
 * reproducible. accumulate bit 0: add to C; bit 1: add to bias_out. row_scale (NULL = none): A's row k is multiplied by
 * row_scale[k / rows_per_scale] as it is loaded (dy of a layer whose output carried a per-sample stochastic-depth factor).
 * bias_out (M) or NULL: the column sums of the scaled A (= db) from the same pass. */
P2C_API int64_t p2c_gemm_tn_workspace_floats(int32_t M, int32_t N, int32_t K);
/* The tile / slice experiment variables (P2C_GEMM_BN, P2C_GEMM_TN_BN, P2C_GEMM_TN_SLICES) are read once; a tool that changes
 * them inside one process calls this to have them read again. */
P2C_API void p2c_gemm_reload_env(void);
P2C_API int p2c_gemm_tn(const float *a, int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, int32_t M, int32_t N,
                int32_t K, int32_t accumulate, const float *row_scale, int32_t rows_per_scale, float *bias_out,
                float *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* P2C_H */
