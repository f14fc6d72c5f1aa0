// Kernel-argument structs shared by the conv kernels (conv.hip: fp32 MFMA; conv_split.hip:
// split-bf16 MFMA).  Internal to the library — the public descriptor is idh_op (include/idh_ops.h).
#pragma once
#include "idh_common.h"

namespace idh_conv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvSrc {
    const float *in;
    const float *w;
    int cs, H, W, Cin;
    int ks, stride, pad_mode, cblocks;  // cblocks = Cin_pad / 16
    // fused x2-upsampled segments of a virtual concat (idh_conv_src.up_*); up_in[0] == nullptr: none
    const float *up_in[2];
    int up_cs[2];
    int up_c0, up_C;
    // normalise-on-load (idh_conv_src.norm): stats[n][2][Cin], activation applied after (x - mean) * rstd
    const float *norm;
    float norm_slope;
    int norm_act;
};

struct ConvArgs {
    ConvSrc s[2];
    const float *bias;
    const float *res;
    float *out;
    float *ws;
    int res_cs, out_cs;
    int Ho, Wo, Cout, Cout_pad;
    int M;  // N*Ho*Wo
    int MT, NT, S;
    int steps_total;
    int act;
    float slope;
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == IDH_ACT_LRELU) return (slope >= 0.f && slope <= 1.f) ? fmaxf(v, v * slope) : (v < 0.f ? v * slope : v);  // max: one instruction less, same value
    if (act == IDH_ACT_ELU) return v > 0.f ? v : expm1f(v);  // nn.ELU(alpha=1), networks_fast.py:17
    return v;
}

// Epilogue activation with the (runtime, wave-uniform) selector tested ONCE: `body(fn)` is instantiated per activation
// with a straight-line element function.  Calling act_apply per element instead leaves ~3 scalar branches around an inlined
// expm1f for every output value — measured on the Winograd kernel: ~10k cycles of a 50k-cycle tile went to that.
template <typename Body>
__device__ __forceinline__ void act_dispatch(int act, float slope, Body &&body) {
    if (act == IDH_ACT_LRELU) {
        if (slope >= 0.f && slope <= 1.f) body([slope](float v) { return fmaxf(v, v * slope); });
        else body([slope](float v) { return v < 0.f ? v * slope : v; });
    } else if (act == IDH_ACT_ELU) {
        body([](float v) { return v > 0.f ? v : expm1f(v); });
    } else {
        body([](float v) { return v; });
    }
}

constexpr int kZeroFloats = 4096;

// conv_split.hip
constexpr int kSplitTile = 16;  // 16x16 output pixels x 64 channels per workgroup
int launch_conv_split(const ConvArgs &a, int N, int mode, int rows, hipStream_t st);  // mode = IDH_SPLIT_*, rows = 16 | 8

// conv_wino.hip: Winograd F(2x2,3x3) on the fp32 matrix cores (3x3 stride 1, zero padding, one source, Cout % 32 == 0)
bool wino_supported(const ConvArgs &a);
int launch_conv_wino(const ConvArgs &a, int N, int rows, hipStream_t st);  // rows = 16 | 8 (tile rows per workgroup)
int wino_max_group();
int launch_conv_wino_group(const ConvArgs *const *as, const int *Ns, int n, hipStream_t st);  // independent convs, one persistent grid

// conv_wino4.hip: Winograd F(4x4,3x3) on the fp32 matrix cores (3x3 stride 1, zero padding, ONE source, Cout % 64 == 0, Cin > 16)
bool wino4_supported(const ConvArgs &a);
int launch_conv_wino4(const ConvArgs &a, int N, hipStream_t st);
bool wino4_split_enabled();  // developer switch IDH_W4_SPLIT (conv_wino4.hip)

}  // namespace idh_conv
