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

#define IDH_OK 0
#define IDH_EINVAL (-1)      /* bad shape / null pointer / unsupported parameter */
#define IDH_EUNSUPPORTED (-2) /* valid request the kernels do not cover (e.g. C != 16) */
#define IDH_ELAUNCH (-3)      /* hipLaunchKernel reported an error */
#define IDH_EWORKSPACE (-4)   /* workspace pointer null or too small */

#define IDH_MAX_SOURCE_VIEWS 16

/* activation codes for the conv / MLP epilogues */
#define IDH_ACT_NONE 0
#define IDH_ACT_LRELU 1 /* LeakyReLU(slope) */

int idh_version(void);
const char *idh_error_string(int code);

/* ---- layout helpers ------------------------------------------------------------ */
/* (n_img, C, HW) -> (n_img, HW, C).  Replaces the implicit .contiguous()/permute the
 * reference does around grid_sample (reference bd_model.py:170-171). */
int idh_nchw_to_nhwc_f32(const float *src_nchw, float *dst_nhwc, int n_img, int C, int HW, void *stream);
int idh_nhwc_to_nchw_f32(const float *src_nhwc, float *dst_nchw, int n_img, int C, int HW, void *stream);

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
 *   cost_bdhw  (B,D,H,W) out; lowest_bhw (B,H,W) out or NULL; planes_d (D) out or NULL
 * One launch, no workspace.
 */
int idh_cost_volume_dot_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                            const float *src_E_44, const float *cur_invK_44, float dmin, float dmax,
                            int B, int K, int C, int H, int W, int D, float *cost_bdhw,
                            float *lowest_bhw, float *planes_d, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IDH_H_ */
