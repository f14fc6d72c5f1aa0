// 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores of gfx950.
//
// Replaces the same nn.Conv2d calls as conv3x3_lds_k (BasicBlock convs, reference modules/layers.py:59-95, i.e. the
// 3x3 stride-1 layers of CVEncoder / BDDecoderPP / DepthDecoderPP, modules/networks.py:20-215) for the layers that
// carry the flops.  gfx950 runs v_mfma_f32_16x16x4_f32 at the fp32 VECTOR rate (157 TFLOP/s, no TF32), and the
// direct-convolution kernel already sits at 0.81-0.90 of that, so the only lever left for fp32 operands is fewer
// multiplications: F(2x2,3x3) needs 16 instead of 36 per output tile and input channel (2.25x), with fp32 operands
// and fp32 accumulation throughout (transform matrices hold 0, +-1, +-1/2 only).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// * Weights are transformed once at pack time (idh_pack_conv_weight_wino): U[pos][co][ci], pos = 4*xi + nu.
// * The 16 element-wise products over (ci) are 16 independent GEMMs  M[pos][co, tile] = sum_ci U[pos][co, ci] V[pos][ci, tile]
//   issued as D^T = U . V^T with the weights as the MFMA A operand: lane (n = lane & 15, h = lane >> 4) holds, as the B
//   operand of k-step s, channel 16c + 4h + s of TILE n — so the lane that loads the 4x4 patch of tile n for channel quad h
//   (16 ds_read_b128, 4 consecutive channels each) transforms it IN REGISTERS (32 adds per channel) and every transformed
//   value is directly the B operand of one MFMA.  No transformed input ever goes through LDS.
// * A wave owns 16 tiles in a row (32 x 2 output pixels) x 16*NCO output channels: 16 positions x NCO accumulators.
//   A workgroup = WAVES waves stacked vertically: (32 x 2*WAVES) pixels x 16*NCO channels.
// * Per 16-channel chunk the (2*WAVES+2) x 34 halo and the 16-position weight panel (NCO x 16 KiB) are copied into
//   one of two LDS stages by `buffer_load_dwordx4 ... lds` (LDS-DMA: no staging VGPRs, no ds_write; out-of-image
//   texels are out of the descriptor's range and arrive as zeros = zero padding) while the other stage is computed on:
//   one barrier per chunk.
//     halo   [q 4][row = 2*hy + column parity][17] float4, q-plane pitch a multiple of 64 slots: lanes (n, h) of a
//            ds_read_b128 lane group hit 16 different 16-byte bank groups;
//     panel  [co block][pos 16][h 4][m 16] float4 = A fragments in read order (packed in exactly this order).
// * Epilogue: output transform (24 adds per tile and channel quad) in registers, + bias + residual, LeakyReLU, 16-byte
//   NHWC stores (a lane holds 4 consecutive channels of the 2x2 pixels of its tile).
#include "conv_args.h"
#include "../../include/idh_ops.h"

using namespace idh_conv;

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kWinoTileW = 32;   // output pixels per workgroup row = 16 Winograd tiles
constexpr int kWinoHalf = 17;    // halo columns of one parity
constexpr int kOob = 0x7fffffff;

constexpr int wino_plane(int WAVES) { return (2 * WAVES + 2) * 2 * kWinoHalf; }             // float4 slots of one q plane
constexpr int wino_qs(int WAVES) { return (wino_plane(WAVES) + 63) / 64 * 64; }             // q-plane pitch
constexpr int wino_stage(int WAVES, int NCO) { return 4 * wino_qs(WAVES) + NCO * 1024; }    // float4 slots per stage

// OIHW 3x3 -> U = G g G^T in fragment order: dst[chunk c][co block][pos][h][m][e] = U[pos][co = 16 cb + m][ci = 16 c + 4 h + e]
__global__ __launch_bounds__(256) void pack_wino_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout, int Cin, int nC, int nCB) {
    const long long total = (long long)nC * nCB * 16 * 256;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int e = (int)(t & 3), m = (int)((t >> 2) & 15), h = (int)((t >> 6) & 3), pos = (int)((t >> 8) & 15);
        const long long r = t >> 12;
        const int cb = (int)(r % nCB), c = (int)(r / nCB);
        const int co = 16 * cb + m, ci = 16 * c + 4 * h + e;
        double u = 0.0;
        if (co < Cout && ci < Cin) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            const int xi = pos >> 2, nu = pos & 3;
            // rows of G: [1 0 0], [.5 .5 .5], [.5 -.5 .5], [0 0 1]
            const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) u += G[xi][a] * (double)g[a * 3 + b] * G[nu][b];
        }
        dst[t] = (float)u;
    }
}

// LDS-DMA of one 1 KiB piece: lane l's 16 bytes at (voff + soff) of `rs` land in dst[l]; out-of-range lanes write zeros.
// (A __device__ function, not a lambda: the builtin has no host-side declaration and clang then silently drops the
// kernel template's host stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, f32x4 *dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)dst, 16, voff, soff, 0, 0);
}

struct WinoArgs {
    ConvArgs c;
    int tiles_x, tiles_y;
};

template <int WAVES, int NCO>
__global__ __launch_bounds__(64 * WAVES) void conv3x3_wino_k(const WinoArgs wa) {
    constexpr int kRows = 2 * WAVES;
    constexpr int kPlane = wino_plane(WAVES);
    constexpr int kQS = wino_qs(WAVES);
    constexpr int kHaloInstr = 4 * (kQS / 64);   // 1 KiB DMA pieces of the halo
    constexpr int kPanelInstr = NCO * 16;        // ... of the weight panel
    constexpr int kTasks = kHaloInstr + kPanelInstr;
    constexpr int kTPW = (kTasks + WAVES - 1) / WAVES;  // DMA pieces per wave and chunk
    constexpr int kHalo = 4 * kQS;
    constexpr int kStage = wino_stage(WAVES, NCO);
    __shared__ f32x4 lds[2 * kStage];

    const ConvArgs &a = wa.c;
    const ConvSrc &s = a.s[0];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, h = lane >> 4;

    unsigned blk = idh_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = blk % a.NT; blk /= a.NT;
    const int tx = blk % wa.tiles_x; blk /= wa.tiles_x;
    const int ty = blk % wa.tiles_y;
    const int img = blk / wa.tiles_y;
    const int y0 = ty * kRows, x0 = tx * kWinoTileW;
    const int cb0 = nt * NCO;       // first 16-channel output block of this workgroup
    const int n0 = 16 * cb0;
    const int nC = s.cblocks;
    const int nCB = a.Cout_pad / 16;

    // ---- LDS-DMA descriptors: this wave's kTPW pieces of every chunk ----------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)img * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, nC * nCB * 16384, 0x00020000);
    int voff[kTPW];     // per-lane byte offset (halo: texel of this lane's slot, out of range = zero padding; panel: 16 * lane)
    int soff0[kTPW];    // wave-uniform byte offset at chunk 0
    int ldsoff[kTPW];   // wave-uniform float4 slot inside a stage
#pragma unroll
    for (int k = 0; k < kTPW; ++k) {
        const int t = wave * kTPW + k;
        if (t < kHaloInstr) {
            const int q = t & 3, j = t >> 2;
            const int L = 64 * j + lane;            // slot inside the q plane: [row][17]
            const int row = L / kWinoHalf, hxh = L - row * kWinoHalf;
            const int iy = y0 - 1 + (row >> 1), ix = x0 - 1 + 2 * hxh + (row & 1);
            const bool ok = (L < kPlane) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            voff[k] = ok ? (iy * s.W + ix) * s.cs * 4 : kOob;
            soff0[k] = 16 * q;
            ldsoff[k] = q * kQS + 64 * j;
        } else {
            const int i = t - kHaloInstr;
            voff[k] = t < kTasks ? 16 * lane : kOob;
            soff0[k] = (cb0 * 16 + i) * 1024;
            ldsoff[k] = kHalo + 64 * i;
        }
    }
    auto issue = [&](int c, int stage) {
#pragma unroll
        for (int k = 0; k < kTPW; ++k) {
            const int t = wave * kTPW + k;
            if (kTasks % WAVES != 0 && t >= kTasks) break;  // (uniform) the last wave may own fewer pieces
            const bool halo = t < kHaloInstr;
            dma16(halo ? rsA : rsW, lds + stage * kStage + ldsoff[k], voff[k], soff0[k] + (halo ? 64 : nCB * 16384) * c);
        }
    };

    f32x4 acc[16][NCO];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's patch origin inside a stage: q plane h, rows 4*wave .. (hy = 2*wave + y, parity x & 1), column n + (x >> 1)
    const int patch0 = h * kQS + (4 * wave) * kWinoHalf + n;
    const int frag0 = kHalo + h * 16 + n;

    auto compute = [&](int stage) {
        const f32x4 *sH = lds + stage * kStage + patch0;
        const f32x4 *sW = lds + stage * kStage + frag0;
        f32x4 v[4][4];
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) v[y][x] = sH[(2 * y + (x & 1)) * kWinoHalf + (x >> 1)];
        // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] (rows, then columns), in place
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const f32x4 d0 = v[0][x], d1 = v[1][x], d2 = v[2][x], d3 = v[3][x];
            v[0][x] = d0 - d2; v[1][x] = d1 + d2; v[2][x] = d2 - d1; v[3][x] = d1 - d3;
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const f32x4 d0 = v[y][0], d1 = v[y][1], d2 = v[y][2], d3 = v[y][3];
            v[y][0] = d0 - d2; v[y][1] = d1 + d2; v[y][2] = d2 - d1; v[y][3] = d1 - d3;
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            f32x4 A[NCO];
#pragma unroll
            for (int j = 0; j < NCO; ++j) A[j] = sW[(j * 16 + p) * 64];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < NCO; ++j)
                    acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j][k], v[p >> 2][p & 3][k], acc[p][j], 0, 0, 0);
        }
    };

    // ---- K loop: chunk c+1 lands in the other stage while chunk c is computed on; one barrier per chunk --------------
    issue(0, 0);
    __syncthreads();  // (the compiler drains vmcnt before the barrier: the DMA of every wave has landed)
#pragma unroll 1
    for (int c = 0; c < nC; ++c) {
        if (c + 1 < nC) issue(c + 1, (c + 1) & 1);
        compute(c & 1);
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; lane = 4 consecutive channels of the 2x2 pixels of tile n
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)img * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)img * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                          a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
    int voffO[2][2], voffR[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = y0 + 2 * wave + i, ox = x0 + 2 * n + j;
            const bool ok = (oy < a.Ho) & (ox < a.Wo);
            const int pixel = oy * a.Wo + ox;
            voffO[i][j] = ok ? (pixel * a.out_cs + n0 + 4 * h) * 4 : kOob;
            voffR[i][j] = ok ? (pixel * a.res_cs + n0 + 4 * h) * 4 : kOob;
        }
    f32x4 rv[2][2][NCO];
    if (a.res) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb) rv[i][j][cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, voffR[i][j] + 64 * cb, 0, 0));
    }
#pragma unroll
    for (int cb = 0; cb < NCO; ++cb) {
        const f32x4 b4 = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + n0 + 16 * cb + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 t[2][4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            t[0][nu] = acc[0 + nu][cb] + acc[4 + nu][cb] + acc[8 + nu][cb];
            t[1][nu] = acc[4 + nu][cb] - acc[8 + nu][cb] - acc[12 + nu][cb];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 y[2];
            y[0] = t[i][0] + t[i][1] + t[i][2];
            y[1] = t[i][1] - t[i][2] - t[i][3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 o = y[j] + b4;
                if (a.res) o += rv[i][j][cb];
                if (a.act != IDH_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = act_apply(o[r], a.act, a.slope);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, voffO[i][j] + 64 * cb, 0, 0);
            }
        }
    }
}

template <int WAVES, int NCO>
int launch_wino(const ConvArgs &a, int N, hipStream_t st) {
    WinoArgs wa{a, (a.Wo + kWinoTileW - 1) / kWinoTileW, (a.Ho + 2 * WAVES - 1) / (2 * WAVES)};
    wa.c.NT = a.Cout / (16 * NCO);
    const long long blocks = (long long)N * wa.tiles_x * wa.tiles_y * wa.c.NT;
    if (blocks >= (1ll << 31)) return IDH_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_wino_k<WAVES, NCO>), dim3((unsigned)blocks), dim3(64 * WAVES), 0, st, wa);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

}  // namespace

namespace idh_conv {

bool wino_supported(const ConvArgs &a) {
    const ConvSrc &s = a.s[0];
    return s.ks == 3 && s.stride == 1 && s.pad_mode == IDH_PAD_ZEROS && !s.up_in[0] && !s.norm && !a.s[1].in && a.S == 1 && (a.Cout % 32) == 0 &&
           (long long)s.H * s.W * s.cs * 4 < (1ll << 31) && (long long)a.Ho * a.Wo * a.out_cs * 4 < (1ll << 31) &&
           (!a.res || (long long)a.Ho * a.Wo * a.res_cs * 4 < (1ll << 31)) && (long long)s.cblocks * a.Cout_pad * 1024 < (1ll << 31);
}

int launch_conv_wino(const ConvArgs &a, int N, int rows, hipStream_t st) {
    if (!wino_supported(a)) return IDH_EUNSUPPORTED;
    if (rows == 8) return launch_wino<4, 2>(a, N, st);
    return launch_wino<8, 2>(a, N, st);
}

}  // namespace idh_conv

extern "C" size_t idh_packed_wino_weight_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)((Cin + 15) & ~15) * ((Cout + 15) & ~15) * 16;
}

extern "C" int idh_pack_conv_weight_wino(const float *w, float *dst, int Cout, int Cin, void *stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0) return IDH_EINVAL;
    const int nC = (Cin + 15) / 16, nCB = (Cout + 15) / 16;
    const long long total = (long long)nC * nCB * 4096;
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_wino_weight_k, dim3(grid), dim3(256), 0, idh_stream(stream), w, dst, Cout, Cin, nC, nCB);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}
