/*
 * idh.h — C ABI of the MI355X-native cost-volume hot path of nianticlabs/implicit-depth.
 *
 * The reference has no native/FFI layer: its hot path is a set of Python nn.Module
 * attributes of BDModel / DepthModel that are swapped by attribute replacement
 * (reference test_bd.py:80-81, `model.cost_volume = model.cost_volume.to_fast()`).
 * Each entry point below replaces the body of one such module's forward(); the
 * reference file:line it replaces is cited per function.  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (gfx950 HBM) unless the name ends in _host;
 *   - tensors are dense fp32; layouts are spelled in the parameter names
 *     (nchw / nhwc / bdhw ...), strides are in floats;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     asynchronous on that stream, never synchronise the device and never allocate:
 *     the caller owns outputs and the workspace (size from the *_workspace_bytes call);
 *   - return value: IDH_OK (0) or a negative IDH_E* code; nothing throws.
 */
#ifndef IDH_H_
#define IDH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations of this header are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define IDH_OK 0
#define IDH_EINVAL (-1)      /* bad shape / null pointer / unsupported parameter */
#define IDH_EUNSUPPORTED (-2) /* valid request the kernels do not cover (e.g. C != 16) */
#define IDH_ELAUNCH (-3)      /* hipLaunchKernel reported an error */
#define IDH_EWORKSPACE (-4)   /* workspace pointer null or too small */

#define IDH_MAX_SOURCE_VIEWS 16

/* activation codes for the conv / MLP epilogues */
#define IDH_ACT_NONE 0
#define IDH_ACT_LRELU 1 /* LeakyReLU(slope) */
#define IDH_ACT_ELU 2   /* ELU(alpha=1) */

int idh_version(void);
const char *idh_error_string(int code);

/* ---- layout helpers ------------------------------------------------------------ */
/* (n_img, C, HW) -> (n_img, HW, C).  Replaces the implicit .contiguous()/permute the
 * reference does around grid_sample (reference bd_model.py:170-171). */
int idh_nchw_to_nhwc_f32(const float *src_nchw, float *dst_nhwc, int n_img, int C, int HW, void *stream);
int idh_nhwc_to_nchw_f32(const float *src_nhwc, float *dst_nchw, int n_img, int C, int HW, void *stream);

/* ---- options shared by the two plane-sweep volumes ------------------------------ */
/* Optional extras of the two volume entry points (the *_ex_fwd forms; NULL = all defaults).
 *   cur/src_batch_stride  floats between consecutive batch elements of cur_nhwc / src_nhwc (0 = dense:
 *                         H*W*C and K*H*W*C).  Lets the matching-encoder head hand over ONE
 *                         (B, K+1, H, W, C) buffer — frame b's current image followed by its K source images, the
 *                         order reference bd_model.py:149-160 produces — without a regrouping copy.
 *   planes                caller-supplied depth planes = the reference's `depth_planes_bdhw` argument
 *                         (modules/cost_volume.py:324-347, used instead of generate_depth_planes): element
 *                         (b,d,pixel) at planes[b*planes_batch_stride + d*planes_plane_stride +
 *                         pixel*planes_pixel_stride]; pixel stride 0 (an expand()ed (B,D,1,1) view) or 1
 *                         (dense per-pixel planes).  NULL = log-spaced planes from dmin/dmax; when given,
 *                         dmin/dmax are ignored and planes_d is not written.
 * Host struct, read during the call. */
typedef struct idh_volume_opts {
    int64_t cur_batch_stride;
    int64_t src_batch_stride;
    const float *planes;
    int64_t planes_batch_stride;
    int64_t planes_plane_stride;
    int32_t planes_pixel_stride;
    int32_t kernel; /* dot-product volume only: 0 = automatic (the launcher's choice), or force one IDH_CV_KERNEL_* —
                       a test / profiling hook so that every kernel can be checked against the same goldens */
    float *scratch; /* dot-product volume only, optional: >= idh_cost_volume_dot_scratch_floats(...) floats of device memory.
                       When the launch splits the depth planes over workgroups, each group leaves its (best cost, plane) per
                       pixel here and the arg-max pass combines those instead of re-reading the whole volume (same result:
                       first maximum wins).  NULL: no scratch, the pass reads the volume. */
    int64_t scratch_floats;
    int64_t struct_size; /* = sizeof(idh_volume_opts) of the header the caller was built against.  The fields from `scratch` on exist since ABI
                            version 101; the library honours them only when struct_size >= offsetof(struct_size) + 8, i.e. when the caller says
                            its struct reaches at least this far (a LARGER value - a later header with more fields appended - is accepted).
                            This is a consistency check between a caller and the header it was built with, NOT protection for a caller built
                            against the shorter version-100 struct: for such a caller the field itself lies past the end of its struct.  A
                            binding must therefore compare idh_sizeof_volume_opts() / idh_version() with its own mirror when it loads the
                            library (implicit-depth_amd/_lib.py does, and refuses to bind on a mismatch). */
} idh_volume_opts;

/* sizeof(idh_volume_opts) as compiled into the library (bindings assert their mirror matches; idh_version() >= 101). */
size_t idh_sizeof_volume_opts(void);

#define IDH_CV_KERNEL_LANE 1   /* cv_dot_k: one lane per sample, taps through the vector L1 */
#define IDH_CV_KERNEL_QUAD 2   /* cv_dot_quad_k: four lanes per sample, quad-coalesced taps */
#define IDH_CV_KERNEL_WINDOW 3 /* cv_dot_win_k: source windows staged in LDS (needs a map of >= 48 x 12 texels) */

/* ---- plane-sweep dot-product cost volume --------------------------------------- */
/* Replaces CostVolumeManager.build_cost_volume + forward
 * (reference modules/cost_volume.py:221-358; geometry utils/geometry_utils.py:55-89):
 *   cost[b,d,y,x] = sum_k sum_c cur[b,y,x,c] * bilinear_zeros(src[b,k], proj_k(x,y,depth_d))[c]
 *   lowest[b,y,x] = depth_{argmax_d cost[b,d,y,x]}   (first maximum wins)
 * with depth_d log-spaced in [dmin,dmax] (cost_volume.py:98-132).
 *   cur_nhwc   (B,H,W,C)      src_nhwc (B,K,H,W,C)        C must be 16
 *   src_K_44   (B,K,4,4) source intrinsics at matching scale
 *   src_E_44   (B,K,4,4) src_cam_T_cur_cam
 *   cur_invK_44(B,4,4)
 *   cost       out: (B,D,H,W) when cost_nhwc_cs == 0 (the reference's layout), or NHWC
 *              (B,H,W,cost_nhwc_cs >= D) so the CVEncoder's first conv reads it directly
 *   lowest_bhw (B,H,W) out or NULL; planes_d (D) out or NULL
 * One volume kernel (plus a small arg-max pass when the planes are split over workgroups), no workspace needed.
 */
int idh_cost_volume_dot_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                            const float *src_E_44, const float *cur_invK_44, float dmin, float dmax,
                            int B, int K, int C, int H, int W, int D, float *cost, int cost_nhwc_cs,
                            float *lowest_bhw, float *planes_d, void *stream);
int idh_cost_volume_dot_ex_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                               const float *src_E_44, const float *cur_invK_44, float dmin, float dmax,
                               int B, int K, int C, int H, int W, int D, float *cost, int cost_nhwc_cs,
                               float *lowest_bhw, float *planes_d, const idh_volume_opts *opts, void *stream);

/* Floats of idh_volume_opts.scratch the window kernel can use for this shape (0: another kernel takes it): the arg-max partials of a launch that
 * splits the planes over workgroups (2 * groups * B * H * W), followed - since ABI 105 - by the run lists of every (frame, tile, plane group) task,
 * which a small kernel (cv_runs_k) then builds AHEAD of the volume kernel instead of every workgroup in its own prologue (~10 % of its life).  Results
 * are bit-identical with any scratch size: a scratch that only covers the first part just forgoes the second. */
long long idh_cost_volume_dot_scratch_floats(int B, int K, int C, int H, int W, int D);

/* Name of the kernel idh_cost_volume_dot*_fwd launches for this shape (what rocprofv3 --kernel-trace will show);
 * lets bench.py label its roofline without guessing the launcher's choice. */
const char *idh_cost_volume_dot_kernel_name(int B, int K, int H, int W, int D);

/* ---- plane-sweep MLP feature volume ---------------------------------------------------- */
/* Replaces FeatureVolumeManager.build_cost_volume + forward (reference
 * modules/cost_volume.py:437-706, 324-358) for K <= 8 source views, C = 16, hidden width 128:
 *   vol[b,d,y,x] = MLP([warped src feats (16K), cur feats (16), mask (K), depths (K), plane depth,
 *                       dot products (K), ray angles (K), rays (3(K+1)), pose distances (3K)])
 * with MLP = Linear -> LeakyReLU(.01) -> Linear -> LeakyReLU(.01) -> Linear (networks.py:218-233).
 * The first Linear's weight W1 (128 x 16(K+1)+10K+4) is passed in three pieces, in the K order
 * the kernel builds its operands in (implicit-depth_amd/cost_volume.py:pack_feature_mlp):
 *   w1_voxel_packed  per-voxel columns [warped K*16 | 4 metadata blocks], MFMA fragment order.  Metadata blocks of the fp32 kernel for
 *                    K <= 8, C = 16 (fv_mlp_k, ABI >= 103): lane quarter q carries [z, dot, ray angle, ray xyz] of views q and q + 4 at slots
 *                    0..5 / 6..11 of three 16-column blocks; the plane depth sits in quarter 3's slot 6 when K < 8, else alone in block 3
 *                    (quarter 0, k-step 0).  The K per-view "valid" columns are NOT in the blob: those inputs are identically 1 (z is
 *                    clamped to 1e-5 before the z > 0 test, geometry_utils.py:86), so the caller adds their weight columns to `b1`
 *                    (implicit-depth_amd/cost_volume.py: feature_mlp_column_maps(fold_mask=True), feature_mlp_mask_columns).  The generic
 *                    kernel (K > 8 / C = 32, fv_mlp_gen_k) takes EIGHT slots per view group j (views q + 4j): [valid, z, dot, ray angle |
 *                    ray xyz, plane depth (group 0, quarter 0) / 0] = 2 ceil(K/4) blocks (layout="gen8"); the f16x3 kernel keeps the
 *                    seven-slot packing with the mask column (the default of feature_mlp_column_maps).
 *   w1_pixel_packed  per-pixel columns [cur feats 16 | cur ray 3 + pad], MFMA fragment order
 *   w1_pose_rowmajor (128, 3K) columns of the pose-distance / R / t measures (folded into a
 *                    per-batch-element bias on the device)
 *   vecs_b2_w3_b3    3 x 128 floats: b2, W3[0,:], [b3, 0...]
 *   vol              (B,D,H,W) when vol_nhwc_cs == 0, else NHWC (B,H,W,vol_nhwc_cs >= D)
 *   lowest_bhw       (B,H,W) or NULL; mask_bhw (B,H,W) uint8 "overall mask" of the LAST plane or
 *                    NULL; planes_d (D) or NULL (written together with lowest_bhw)
 *   workspace        >= idh_feature_volume_workspace_bytes(B)
 * One frame's K source maps are addressed through a buffer descriptor: K*H*W*C*4 bytes must stay below 2 GiB (else IDH_EUNSUPPORTED).
 */
size_t idh_feature_volume_workspace_bytes(int B);
int idh_feature_volume_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                           const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                           float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                           const float *w1_voxel_packed, const float *w1_pixel_packed,
                           const float *w1_pose_rowmajor, const float *b1, const float *w2_packed,
                           const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                           unsigned char *mask_bhw, float *planes_d, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Same with idh_volume_opts (batch strides, caller-supplied planes); f16x3 != 0 selects the split-precision
 * variant below (then w1_voxel / w2 are the f16 packs). */
int idh_feature_volume_ex_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                              const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                              float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                              const void *w1_voxel, const float *w1_pixel_packed,
                              const float *w1_pose_rowmajor, const float *b1, const void *w2,
                              const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                              unsigned char *mask_bhw, float *planes_d, void *workspace,
                              size_t workspace_bytes, int f16x3, const idh_volume_opts *opts, void *stream);

/* Split-precision ("f16x3") variant of the same kernel: the two 128-wide layers run on
 * v_mfma_f32_16x16x32_f16 with every fp32 operand expanded into two round-to-nearest f16 pieces
 * (power-of-two scaled per hidden unit for the weights, per voxel for the activations) and the three
 * significant cross products accumulated in fp32 — fp32-equivalent results (tests/test_mlp_split_gpu.py)
 * at 3/16 of the fp32-MFMA cost.  Same arguments, except that w1_voxel_f16 / w2_f16 come from
 * idh_pack_mlp_weight_f16 and the voxel columns are ordered [4 metadata blocks | warped K*16]
 * (implicit-depth_amd/cost_volume.py).  idh_pack_mlp_weight_f16 packs columns [col0, col0+n_in) of a
 * row-major (128, ld) matrix as [ceil(n_in/32)][8][piece 2][lane 64][8 halves] followed by the 128
 * per-row scale floats. */
size_t idh_packed_mlp_weight_f16_bytes(int n_in);
int idh_pack_mlp_weight_f16(const float *w_row_major, void *dst, int ld, int col0, int n_in, void *stream);
int idh_feature_volume_f16x3_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                                 const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                                 float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                                 const void *w1_voxel_f16, const float *w1_pixel_packed,
                                 const float *w1_pose_rowmajor, const float *b1, const void *w2_f16,
                                 const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                                 unsigned char *mask_bhw, float *planes_d, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* ---- per-pixel occlusion MLP over all query planes ---------------------------------- */
/* Replaces the per-plane loop bd_model.py:293-304 -> run_mlp_val (:412-442) ->
 * BinaryMLPNetwork scale 0 (modules/networks.py:98-115): for every pixel m and plane p
 *   x = [depth[b,p,pix], feat[m, 0:Cf], (prior[b,p,pix])]
 *   out[b,p,pix] = W3 . ELU(W2 . ELU(W1 . x + b1) + b2) + b3          (hidden width 128)
 * W1's feature columns and W2 are passed in MFMA fragment order (idh_pack_mlp_weight);
 * vecs6x128 = rows {b1, W1[:,depth], W1[:,prior], b2, W3[0,:], [b3,0,...]}.
 *   feat_nhwc  rows of Cf floats, feat_cs floats apart (B*HW rows); any feat_cs >= Cf and any 4-byte-aligned base since ABI 105
 *              (rows that are not 16-byte aligned - the reference's 65- / 66-float [depth | feat | prior] rows - are read with dword loads)
 *   depth_bphw (B,P,HW); prior_bphw (B,P,HW) or NULL (then prior_const is used when has_prior)
 *   out_bphw   (B,P,HW) logits
 */
size_t idh_packed_mlp_weight_floats(int n_in);
int idh_pack_mlp_weight(const float *w_row_major, float *dst, int ld, int col0, int n_in, void *stream);
int idh_binary_mlp_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *depth_bphw,
                       const float *prior_bphw, int has_prior, float prior_const,
                       const float *w1f_packed, const float *w2_packed, const float *vecs6x128, int B,
                       int P, int HW, float *out_bphw, void *stream);
/* The same pass over features with arbitrary strides: element (b, pix, c) at feat[b * feat_batch_stride + pix * feat_pixel_stride +
 * c * feat_channel_stride] (floats).  (1, HW-contiguous planes) reads the channel planes of an NCHW tensor in place: BDModel.run_mlp_val
 * (bd_model.py:415-439) concatenates [rendered_depth | feature_s0 | prior] along dim 1 and hands the MLP a permute(0, 2, 3, 1) VIEW of it -
 * with this entry point the drop-in's BinaryMLPNetwork.forward reads that view without materialising (B,H,W,65) rows.  ABI 105. */
int idh_binary_mlp_strided_fwd(const float *feat, long long feat_batch_stride, int feat_pixel_stride, int feat_channel_stride, int Cf,
                               const float *depth_bphw, const float *prior_bphw, int has_prior, float prior_const,
                               const float *w1f_packed, const float *w2_packed, const float *vecs6x128, int B, int P, int HW,
                               float *out_bphw, void *stream);
/* Same, with the per-plane 128x128 layer in "f16x3" split precision (w2_f16 from
 * idh_pack_mlp_weight_f16, csrc/split_f16.h) and ELU's exp on v_exp_f32: fp32-equivalent results
 * (tests/test_mlp_split_gpu.py) at a fraction of the fp32-MFMA cost. */
int idh_binary_mlp_f16x3_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *depth_bphw,
                             const float *prior_bphw, int has_prior, float prior_const,
                             const float *w1f_packed, const void *w2_f16, const float *vecs6x128, int B,
                             int P, int HW, float *out_bphw, void *stream);

/* Fused per-pixel binary depth search (reference bd_model.py:273-292, infer_depth=True): `iters`
 * dependent evaluations of the same MLP at each pixel's current query depth, bounds [lo,hi], first
 * query (hi-lo)/2, a pixel is "visible" when sigmoid(logit) < threshold.  Outputs the final query
 * depths and the logits of the last evaluation (the reference's outputs["search_depths"] / ["pred_0"]).
 * Constant threshold; idh_binary_mlp_search_thr_fwd takes the per-depth Thresholder. */
int idh_binary_mlp_search_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *prior_b1hw,
                              int has_prior, float prior_const, const float *w1f_packed,
                              const float *w2_packed, const float *vecs6x128, int B, int HW, int iters,
                              float lo, float hi, float threshold, float *search_depths_b1hw,
                              float *last_logits_b1hw, void *stream);
/* The same search with the per-depth Thresholder (binary_metrics_utils.py:42-52; bd_model.py:282-283):
 * threshold = thresholds[bucketize(query depth, bins)]; pass thr_logits[i] = logit(thresholds[i]). */
int idh_binary_mlp_search_thr_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *prior_b1hw,
                                  int has_prior, float prior_const, const float *w1f_packed,
                                  const float *w2_packed, const float *vecs6x128, int B, int HW, int iters,
                                  float lo, float hi, const float *bins, const float *thr_logits, int n_bins,
                                  float *search_depths_b1hw, float *last_logits_b1hw, void *stream);
/* f16x3 variant of both searches (w2_f16 from idh_pack_mlp_weight_f16): n_bins == 0 -> constant
 * `threshold`, else the per-depth table. */
int idh_binary_mlp_search_f16x3_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *prior_b1hw,
                                    int has_prior, float prior_const, const float *w1f_packed,
                                    const void *w2_f16, const float *vecs6x128, int B, int HW, int iters,
                                    float lo, float hi, float threshold, const float *bins,
                                    const float *thr_logits, int n_bins, float *search_depths_b1hw,
                                    float *last_logits_b1hw, void *stream);

/* ---- temporal prior ------------------------------------------------------------------ */
/* Replaces BDModel.sample_prior (reference experiment_modules/bd_model.py:395-410): back-project
 * the rendered depth, project into the previous frame's camera, nearest-neighbour sample the
 * previous prediction (zeros outside), -1 where the rendered depth is not positive.
 *   rendered_depth_bphw (B,P,H,W); prior_pred_bqhw (B,Q,H,W) (plane p samples channel min(p,Q-1))
 *   cur_world_T_cam_44, prior_cam_T_world_44, K_44, invK_44: (B,4,4) at the prediction resolution
 */
int idh_sample_prior_fwd(const float *rendered_depth_bphw, const float *prior_pred_bqhw, int Q,
                         const float *cur_world_T_cam_44, const float *prior_cam_T_world_44,
                         const float *K_44, const float *invK_44, int B, int P, int H, int W,
                         float *out_bphw, void *stream);

/* ---- evaluation metrics (the step right after the path; rows of the metrics all-gather) ---- */
/* Plane IoU — PlaneEvaluator.compute_batch_scores / compute_batch_scores_test (reference
 * utils/binary_metrics_utils.py:59-192): out[b,d,t,{iou, iou_pos, iou_neg}] over pixels with gt > 0 and
 * query > 0; target = query < gt; prediction = pred > threshold.  Either T (<= 8) constant thresholds
 * (bins == NULL) or the per-depth Thresholder (:42-52): bins (n_bins sorted edges), thresholds has
 * n_bins entries, T must be 1.  Depth metrics — compute_depth_metrics_batched (utils/metrics_utils.py:52-120):
 * out[b, 12] = abs_diff, abs_rel, sq_rel, rmse, rmse_log, a5, a10, a25, a0, a1, a2, a3 over valid_bn != 0.
 * workspace >= idh_metrics_workspace_bytes(B, D, N, T) (8-byte aligned). */
size_t idh_metrics_workspace_bytes(int B, int D, int N, int T);
int idh_plane_iou_fwd(const float *query_depth_bdn, const float *gt_depth_b1n, const float *prediction_bdn,
                      const float *thresholds, int T, const float *bins, int n_bins, int B, int D, int N,
                      float *out_bdt3, void *workspace, size_t workspace_bytes, void *stream);
int idh_depth_metrics_fwd(const float *gt_bn, const float *pred_bn, const unsigned char *valid_bn, int B, int N,
                          int mult_a, float *out_b12, void *workspace, size_t workspace_bytes, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* IDH_H_ */
