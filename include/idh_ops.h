/*
 * idh_ops.h — op descriptors for the conv / upsample / layout stage of the hot path.
 *
 * The reference runs CVEncoder / BDDecoderPP / DepthDecoderPP (modules/networks.py:20-215)
 * as ~150 nn.Conv2d + F.interpolate + torch.cat calls.  Here a network pass is a flat array
 * of idh_op descriptors, built once per (module, shape) by the host and submitted with ONE
 * call, idh_run_ops(); an op is one gfx950 kernel launch on the given stream, or shares a launch with the other
 * independent ops of its dependency level (idh_op.group).
 * All activations are NHWC fp32.  "cs" = channel stride = floats between consecutive pixels
 * (>= channel count), which is how torch.cat along channels is eliminated: producers write
 * straight into a channel slice of the consumer's input buffer.
 */
#ifndef IDH_OPS_H_
#define IDH_OPS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations of this header are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

enum {
    IDH_OP_CONV = 1,        /* implicit-GEMM conv on fp32 MFMA (layers.py:59-75 convs)       */
    IDH_OP_UPSAMPLE2 = 2,   /* bilinear x2, align_corners=False (generic_utils.py:94-103).  32-bit element indices since ABI 104:
                               N*H*W*C/4 (input channel quads) and the same count of the output must stay below 2^31 - i.e. inputs
                               under 32 GiB - else IDH_EUNSUPPORTED (no 64-bit-index variant is kept; split the batch) */
    IDH_OP_NCHW_TO_NHWC = 3,/* strided layout import: (N,C,H,W) -> NHWC slice                */
    IDH_OP_NHWC_TO_NCHW = 4,/* strided layout export: NHWC slice -> (N,C,H,W)                */
    IDH_OP_SPLITK_REDUCE = 5,/* sum split-K partials + bias + residual + activation          */
    IDH_OP_POINTWISE_HEAD = 6,/* 1x1 conv to 1 channel (DepthDecoderPP heads, networks.py:158-161); ws != NULL: a second
                                 (N,1,H,W) output = exp(out), the depth map of depth_model.py:425-433 */
    IDH_OP_COPY = 9,        /* channel-strided NHWC -> NHWC slice copy */
    IDH_OP_UPSAMPLE2_NEAREST = 8, /* nearest x2 (SkipDecoder, networks_fast.py:43) */
    IDH_OP_POINTWISE_NCHW = 10, /* 1x1 conv 64 -> 128 read straight from a dense (N,64,H,W) tensor into an NHWC slice: the first
                                   layer of the matching-encoder head (networks.py:279) without a layout-import pass;
                                   src[0].w = idh_pack_conv_weight(128, 64, 1); other widths: IDH_EUNSUPPORTED */
    IDH_OP_POINTWISE_UP = 11, /* out = W . x (+ bias) + up2(low): a 1x1 conv of an NHWC tensor plus the x2 bilinear upsampling (IDH_OP_UPSAMPLE2's
                                 expression) of a half-resolution NHWC tensor of Cout channels - the projection branch of the decoder blocks on
                                 cat(right, up(lo), up(lo2)) (networks.py:52-77, layers.py:86-92), whose upsampled two thirds are projected at low
                                 resolution because a 1x1 conv commutes with bilinear upsampling.  src[0] = x (N,H,W,Cin), w =
                                 idh_pack_conv_weight(Cout, Cin, 1); src[1].in / .cs / .H / .W = the (N,H/2,W/2,Cout) map; Cin == Cout in {64, 128},
                                 H and W even; other shapes: IDH_EUNSUPPORTED */
    IDH_OP_INSTNORM = 7     /* nn.InstanceNorm2d (no affine, eps 1e-5) [+ LeakyReLU] on NHWC; matching-encoder
                               head networks.py:279-283.  ws: N*(ceil(HW/1024)+1)*2*C floats (chunk partials + mean/rstd);
                               out == NULL: statistics only — float[N][2][C] at ws + N*ceil(HW/1024)*2*C, for a
                               consumer convolution that normalises on load (idh_conv_src.norm) */
};

#define IDH_PAD_ZEROS 0
#define IDH_PAD_REPLICATE 1

/* One conv source: input tensor + its packed weights.  A conv op may sum TWO sources into
 * the same accumulator: BasicBlock's conv2(h) + downsample(x) (layers.py:86-92) becomes one
 * launch whose second source is the 1x1 (or strided 3x3) projection of the block input. */
typedef struct idh_conv_src {
    const float *in; /* NHWC, N x H x W x (cs) */
    const float *w;  /* packed by idh_pack_conv_weight: [ks*ks][Cin_pad/4][Cout_pad][4] */
    int32_t cs, H, W, Cin;
    int32_t ks, stride, pad_mode, _r;
    /* Virtual concat with fused bilinear x2 (the UNet++ decoder's torch.cat([x, upsample(a), upsample(b)], 1),
     * networks.py:64-76): channels [0, up_c0) come from `in`; channels [up_c0 + i*up_C, up_c0 + (i+1)*up_C) are the
     * x2 bilinear upsampling (align_corners=False, generic_utils.py:94-103) of up_in[i], an NHWC map of (H/2, W/2)
     * with channel stride up_cs[i], interpolated while the consumer conv stages its halo — the upsampled tensor and
     * the concat buffer are never written.  up_in[0] == NULL: plain source.  LDS-staged 3x3 / fused 1x1 sources
     * only; up_c0 and up_C multiples of 16, H and W even, Cin == up_c0 + n_segments * up_C. */
    const float *up_in[2];
    int32_t up_cs[2];
    int32_t up_c0, up_C;
    /* Normalise-on-load: norm != NULL -> the kernel reads x' = act((x - mean[n][c]) * rstd[n][c]) instead of x, with
     * norm = the statistics an IDH_OP_INSTNORM left in its workspace (float[N][2][Cin]: means then 1/sqrt(var + eps)).
     * nn.InstanceNorm2d + LeakyReLU between two convolutions of the matching-encoder head (networks.py:280-281) then
     * cost one statistics pass and no normalised copy of the tensor.  Same expression as the materialising kernel
     * (bit-identical); padding is applied after the normalisation (a zero-padded tap stays 0).  src[0] of the
     * LDS-staged 3x3 kernel with 16-channel output tiles only (IDH_EUNSUPPORTED otherwise). */
    const float *norm;
    float norm_slope;
    int32_t norm_act;  /* IDH_ACT_* */
} idh_conv_src;

typedef struct idh_op {
    int32_t kind;
    int32_t N;            /* batch */
    idh_conv_src src[2];  /* src[1].in == NULL when unused */
    const float *bias;    /* Cout floats (already summed over sources) or NULL */
    const float *res;     /* residual NHWC (same Ho x Wo) or NULL */
    float *out;           /* NHWC Ho x Wo x out_cs (or NCHW for the export op) */
    float *ws;            /* split-K partials: split_k x M x Cout_pad floats */
    int32_t res_cs, out_cs;
    int32_t Ho, Wo, Cout;
    int32_t act;          /* IDH_ACT_* */
    float slope;
    int32_t split_k;      /* 1 = no split */
    int32_t tile_m, tile_n; /* direct kernel: wave tile in 16-wide MFMA sub-tiles (1,2,4), 0 = auto;
                               tile_m = 8 / 9 selects the LDS-staged kernel with 8- / 4-row tiles (3x3 stride 1 [+ a 1x1 or a
                               3x3 stride-2 second source], or a LONE 3x3 stride-2 zero-padded source with Cout % 32 == 0 and
                               Wo >= 16 - conv1 of a stride-2 BasicBlock - which runs on the kernel's stride-2 loader);
                               tile_m = IDH_SPLIT_F16X3 selects the split-precision
                               kernel (3x3 stride 1, one source, Cout % 64 == 0; src[0].w =
                               idh_pack_conv_weight_split output of the same mode); there tile_n = 8
                               selects 8-row instead of 16-row tiles */
    int32_t group;        /* != 0: consecutive CONV / UPSAMPLE2 ops with the same id are mutually independent (one
                               dependency level of the plan, see Plan.schedule in nhwc.py) and may be launched as
                               ONE grid: runs of 4-row LDS convs with equal channel tiles always are; a mixed run
                               (4-row LDS convs with 64- / 32-channel tiles, the 16x64-tile direct conv, bilinear
                               upsampling) is when each member has <= 512 workgroups, i.e. at small batch */
} idh_op;

/* Repack OIHW conv weights (reference nn.Conv2d layout) for the MFMA B-fragment loads:
 * dst[tap][ci/4][co][ci%4], zero padded to Cin_pad = ceil16(Cin), Cout_pad = ceil16(Cout).
 * dst must hold idh_packed_weight_floats(Cout,Cin,ks) floats. */
size_t idh_packed_weight_floats(int Cout, int Cin, int ks);
int idh_pack_conv_weight(const float *w_oihw, float *dst, int Cout, int Cin, int ks, void *stream);

/* Split-precision convolution (csrc/conv_split.hip), opt-in: fp32 operands expanded into 16-bit pieces,
 * cross products accumulated in fp32 on the f16 matrix cores — fp32-equivalent results
 * (error ~4e-7 of the output scale, like an fp32 FMA chain) at 3/16 of the fp32-MFMA cost.
 *   IDH_SPLIT_F16X3:  x/s = x0+x1 (2 f16 pieces, power-of-two scaling per output channel for the
 *                     weights and per halo chunk for the activations), 3 products.
 * Packed weights: [Cin_pad/16][tap row 3][Cout/64][tap in row 3][piece][ci half 2][co 64][8] 16-bit,
 * followed by Cout per-channel scale floats and Cout exponents.  3x3 kernels, Cout % 64 == 0
 * (IDH_EUNSUPPORTED otherwise).
 * A fused 1x1 second source (src[1]: BasicBlock's downsample(x)) is packed into the same blob
 * (w_1x1 = (Cout, Cin_1x1) row-major or NULL / 0): [3x3 panels][1x1 panels: Cin_1x1_pad/16 x Cout/64 x
 * piece x half x 64 x 8][scales]; src[1].w is then ignored by the kernel. */
#define IDH_SPLIT_F16X3 11
size_t idh_packed_split_weight_bytes(int Cout, int Cin, int Cin_1x1, int mode);
int idh_pack_conv_weight_split(const float *w_oihw, const float *w_1x1, void *dst, int Cout, int Cin, int Cin_1x1, int mode,
                               void *stream);

/* Winograd F(2x2,3x3) convolution (csrc/conv_wino.hip): the same fp32 operands and fp32 accumulation on
 * v_mfma_f32_16x16x4_f32, with 16 instead of 36 multiplications per 2x2 output tile and input channel (the 3x3
 * stride-1 convs of BasicBlock, layers.py:59-95).  IDH_OP_CONV with tile_m = IDH_TILE_WINO; src[0].w = the output of
 * idh_pack_conv_weight_wino (U = G g G^T per (co, ci), in MFMA A-fragment order:
 * [Cin_pad/8][Cout_pad/16][position 16][ci pair 4][co 16][2]); tile_n is ignored (32 x 8 pixel x 32 channel tiles).
 * Shape family: 3x3, stride 1, zero padding, Cout % 32 == 0, split_k == 1; src[1] may be a 1x1 stride-1 projection of a
 * tensor of the same size (weights packed by idh_pack_conv_weight as usual); anything else: IDH_EUNSUPPORTED.  Results differ from the direct kernel by fp32 rounding only (~1e-6 of the output scale). */
#define IDH_TILE_WINO 12
size_t idh_packed_wino_weight_floats(int Cout, int Cin);
int idh_pack_conv_weight_wino(const float *w_oihw, float *dst, int Cout, int Cin, void *stream);

/* Winograd F(4x4,3x3) convolution (csrc/conv_wino4.hip, conv3x3_wino4_k): 36 instead of 64 multiplications per 4x4 output pixels
 * and input channel (1.78x fewer MFMAs than IDH_TILE_WINO), fp32 operands and accumulation, interpolation points {0, +-1/2, +-2, inf};
 * for the plain 3x3 stride-1 convs of BasicBlock / the decoders (layers.py:59-95, networks.py:20-215).  IDH_OP_CONV with
 * tile_m = IDH_TILE_WINO4; src[0].w = the output of idh_pack_conv_weight_wino4 (U = G g G^T per (co, ci) in MFMA A-fragment order:
 * [Cin_pad/8][Cout_pad/16][k-step 2][position group 9][lane 64][4], ci = 8 stage + 2 (lane >> 4) + k-step; row g, element e = slot 4 g + e of the kernel's
 * V layout - since ABI 105 a quadrant's nine positions are slots 8 q .. 8 q + 7 and 32 + q (csrc/conv_wino4.hip w4_v2p), quadrant-major before: repack after upgrading);
 * 32 x 8 pixel x 64 channel tiles, the input transform shared through LDS by the four 16-channel waves of a tile, two persistent
 * workgroups per CU.  Shape family: 3x3, stride 1, zero padding, Cin > 16, Cout % 64 == 0, split_k == 1, act NONE / LRELU / ELU; src[1] may be
 * a 1x1 stride-1 projection of a tensor of the output's size (weights packed by idh_pack_conv_weight as usual; then res must be NULL);
 * anything else: IDH_EUNSUPPORTED.  Error against fp64 ~2-7e-6 of
 * the output scale (direct kernel ~1e-6, F(2x2) ~4e-7).  (ABI 101 had a register-transform kernel under this code with another
 * weight layout and code 14 for this one; 102 keeps one kernel, one code.) */
#define IDH_TILE_WINO4 13
size_t idh_packed_wino4_weight_floats(int Cout, int Cin);
int idh_pack_conv_weight_wino4(const float *w_oihw, float *dst, int Cout, int Cin, void *stream);

/* sizeof(idh_op) as compiled into the library (bindings assert their mirror matches). */
size_t idh_sizeof_op(void);

/* Launch n ops in order on `stream`. `ops_host` is HOST memory (pointers inside are device). */
int idh_run_ops(const idh_op *ops_host, int n, void *stream);

/* Number of kernel launches idh_run_ops() would issue for these ops (same validation and grouping decisions, nothing
 * is launched; usable without a GPU), or a negative IDH_E* code. */
int idh_count_launches(const idh_op *ops_host, int n);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
