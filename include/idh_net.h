/*
 * idh_net.h — network-level entry points of the conv stage (ABI 105).
 *
 * include/idh_ops.h stops at op lists (idh_run_ops); these are the three entry points SURVEY.md 8(b) names above it, for hosts that do
 * not want to re-implement the plan builder of implicit-depth_amd/nhwc.py:
 *
 *   idh_basic_block_fwd   replaces BasicBlock.forward            modules/layers.py:78-95   (conv3x3 :8-26, conv1x1 :29-31)
 *   idh_cvencoder_fwd     replaces CVEncoder.forward             modules/networks.py:186-215
 *   idh_unetpp_fwd        replaces BDDecoderPP / DepthDecoderPP  modules/networks.py:20-84, 118-183 (+ upsample, utils/generic_utils.py:94-103)
 *
 * Each is a thin host-side builder (csrc/networks.hip) over idh_run_ops: it lays the network out as idh_op descriptors — the same kernel
 * selection (Winograd F(4x4) / F(2x2) / LDS-staged / direct by tile counts), concat elimination, split-K, level scheduling and activation-buffer
 * reuse as nhwc.py's Plan with its default thresholds, so results are bit-identical to the Python drop-ins — and submits them on the caller's
 * stream.  No device allocation, no synchronisation: the caller passes
 *   - a WEIGHT BLOB filled once per (parameters, shape) by the matching *_pack call (packed weights in the layout each layer's kernel reads +
 *     summed biases; the layouts depend on the kernel selection, hence on N / H / W), and
 *   - a WORKSPACE for activations and split-K partials,
 * both sized by the matching *_sizes query.  fp32 throughout (v_mfma_f32_16x16x4_f32).  Return 0 or a negative IDH_E* code; never throw.
 *
 * Tensors are described by idh_tensor.  NHWC tensors are read / written in place (a channel slice of a wider buffer is fine: cs = floats
 * between pixels); a channel count that is not a multiple of 16 must be a whole zero-padded buffer (cs == ceil16(C), padding channels zero)
 * because the conv kernels read whole 16-channel blocks.  NCHW tensors (the reference's layout) are imported / exported by an extra
 * layout op through the workspace.
 */
#ifndef IDH_NET_H_
#define IDH_NET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define IDH_LAYOUT_NHWC 0
#define IDH_LAYOUT_NCHW 1

typedef struct idh_tensor {
    float *ptr;      /* device pointer (ignored by the *_sizes queries and by *_pack) */
    int32_t layout;  /* IDH_LAYOUT_* */
    int32_t C, H, W; /* per image */
    int32_t cs;      /* NHWC: floats between consecutive pixels (>= C); NCHW: ignored (dense (N,C,H,W)) */
} idh_tensor;

/* One nn.Conv2d: OIHW weights (cout, cin, ks, ks), bias (cout) or NULL.  The pointers are read by *_pack only. */
typedef struct idh_conv_params {
    const float *weight;
    const float *bias;
    int32_t cout, cin, ks, stride;
} idh_conv_params;

/* One BasicBlock (layers.py:34-95): conv1 3x3 (stride 1 | 2) + LeakyReLU(0.2), conv2 3x3, shortcut = identity (downsample.ks == 0) or
 * downsample[0]: 1x1 stride 1 / 3x3 stride 2 (layers.py:68-75), LeakyReLU(0.2) after the sum. */
typedef struct idh_block_params {
    idh_conv_params conv1, conv2, downsample;
} idh_block_params;

typedef struct idh_net_sizes {
    size_t workspace_floats; /* activations + split-K partials (+ layout staging) */
    size_t weight_floats;    /* the weight blob */
    int32_t ops;             /* idh_op descriptors of a pass */
    int32_t launches;        /* kernel launches of a pass (idh_count_launches) */
    int32_t wino4, wino2;    /* conv ops on conv3x3_wino4_k / conv3x3_wino_k (the rest: LDS-staged / direct kernels) */
    int32_t recycled;        /* activation buffers that alias an earlier, dead one */
} idh_net_sizes;

/* ---- BasicBlock --------------------------------------------------------------------------------------------------------------------- */
int idh_basic_block_sizes(const idh_block_params *blk, int N, const idh_tensor *x, const idh_tensor *out, idh_net_sizes *sizes);
int idh_basic_block_pack(const idh_block_params *blk, int N, const idh_tensor *x, const idh_tensor *out, float *weight_blob, void *stream);
int idh_basic_block_fwd(const idh_block_params *blk, const float *weight_blob, int N, const idh_tensor *x, const idh_tensor *out,
                        float *workspace, size_t workspace_floats, void *stream);

/* ---- CVEncoder (networks.py:186-215) ------------------------------------------------------------------------------------------------
 * blocks[3 i + 0 / 1 / 2] = convs["ds_conv_i"], convs["conv_i"][0], convs["conv_i"][1], i = 0 .. num_blocks - 1 (4 in every shipped config).
 * cost: the (N, D, H, W) cost / feature volume (NHWC (N,H,W,D) is what the volume kernels write); img_feats[i]: the image-encoder map
 * concatenated at level i (bd_model.py:253-258), IDH_LAYOUT_NCHW as the reference's encoder hands it (the layout import writes it straight into
 * its channel slice of the level's concat buffer; an NHWC map: IDH_EUNSUPPORTED); outs[i]: level i's output, NHWC or NCHW. */
int idh_cvencoder_sizes(const idh_block_params *blocks, int num_blocks, int N, const idh_tensor *cost, const idh_tensor *img_feats,
                        const idh_tensor *outs, idh_net_sizes *sizes);
int idh_cvencoder_pack(const idh_block_params *blocks, int num_blocks, int N, const idh_tensor *cost, const idh_tensor *img_feats,
                       const idh_tensor *outs, float *weight_blob, void *stream);
int idh_cvencoder_fwd(const idh_block_params *blocks, int num_blocks, const float *weight_blob, int N, const idh_tensor *cost,
                      const idh_tensor *img_feats, const idh_tensor *outs, float *workspace, size_t workspace_floats, void *stream);

/* ---- UNet++ decoders (networks.py:20-84 BDDecoderPP, :118-183 DepthDecoderPP) -------------------------------------------------------
 * blocks: in the order the reference's forward visits them — for j = 1..4, for i = 4-j..0:
 *     right_conv_{i}{j-1}, diag_conv_{i+1}{j-1}, [up_conv_{i+1}{j} when i + j != 4], in_conv_{i}{j}[0], in_conv_{i}{j}.conv_0
 * (46 blocks) followed by output_1[0], output_2[0], output_3[0] (the surviving registrations, networks.py:60-62): 49 blocks.
 * heads: NULL (BDDecoderPP) or the four 1x1 convs output_i[1] (DepthDecoderPP, :158-161), i = 0..3.
 * feats[0..4]: the five input maps (image-encoder level 0 + the four CVEncoder outputs), each level half the size of the one before.
 * feature_outs[i] (i = 0..3; C == 0 skips a level — the same array, same C values, must be passed to _sizes, _pack and _fwd; NULL = none):
 *     "feature_s{i}_b1hw" of BDDecoderPP — for DepthDecoderPP the input of head i.
 * log_depth_outs[i] / depth_outs[i] (heads != NULL): dense (N,1,H_i,W_i) maps "log_depth_pred_s{i}_b1hw" and exp() of it
 *     (depth_model.py:425-433); depth_outs may be NULL. */
#define IDH_UNETPP_BLOCKS 49
int idh_unetpp_sizes(const idh_block_params *blocks, int n_blocks, const idh_conv_params *heads, int N, const idh_tensor *feats,
                     const idh_tensor *feature_outs, idh_net_sizes *sizes);
int idh_unetpp_pack(const idh_block_params *blocks, int n_blocks, const idh_conv_params *heads, int N, const idh_tensor *feats,
                    const idh_tensor *feature_outs, float *weight_blob, void *stream);
int idh_unetpp_fwd(const idh_block_params *blocks, int n_blocks, const idh_conv_params *heads, const float *weight_blob, int N,
                   const idh_tensor *feats, const idh_tensor *feature_outs, float *const *log_depth_outs, float *const *depth_outs,
                   float *workspace, size_t workspace_floats, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
