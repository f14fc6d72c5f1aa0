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

// OIHW 3x3 -> U = G g G^T in fragment order: dst[step c][co block][pos][h][m][e] = U[pos][co = 16 cb + m][ci = CH c + KS h + e],
// KS = CH / 4 consecutive channels per lane (CH = input channels per K step: 16 or 8)
__global__ __launch_bounds__(256) void pack_wino_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout, int Cin, int nS, int nCB, int KS) {
    const long long total = (long long)nS * nCB * 16 * 64 * KS;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int e = (int)(t % KS);
        long long r = t / KS;
        const int m = (int)(r & 15), h = (int)((r >> 4) & 3), pos = (int)((r >> 6) & 15);
        r >>= 10;
        const int cb = (int)(r % nCB), c = (int)(r / nCB);
        const int co = 16 * cb + m, ci = 4 * KS * c + KS * h + e;
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

template <int WAVES, int NCO, int CH>
__global__ __launch_bounds__(64 * WAVES, 2) void conv3x3_wino_k(const WinoArgs wa) {
    constexpr int KS = CH / 4;   // MFMA k-steps per K step = consecutive channels per lane
    constexpr int NP = CH / 4;   // 4-channel planes of the halo per K step
    typedef float vec __attribute__((ext_vector_type(KS)));
    constexpr int kRows = 2 * WAVES;
    constexpr int kPlane = wino_plane(WAVES);
    constexpr int kQS = wino_qs(WAVES);
    // LDS-DMA pieces (1 KiB each) of one K step, per wave: a plane of the halo is kJ pieces long; wave w copies piece
    // j = f*WAVES + w of all NP planes (kFull rounds), its share of the kRem remaining j's (kExtra pieces of one j) and
    // kPanel pieces of the weight panel -> one offset VGPR per distinct j plus one for the panel
    constexpr int kJ = kQS / 64;
    constexpr int kFull = kJ / WAVES, kRem = kJ % WAVES;
    constexpr int kExtra = NP * kRem / WAVES;
    static_assert(NP * kRem % WAVES == 0 && (kExtra == 0 || NP % kExtra == 0), "halo pieces must split evenly over the waves");
    constexpr int kPanelPieces = NCO * 4 * KS;
    static_assert(kPanelPieces % WAVES == 0, "panel pieces must split evenly over the waves");
    constexpr int kPanel = kPanelPieces / WAVES;
    constexpr int kTPW = NP * kFull + kExtra + kPanel;  // DMA pieces per wave and K step
    static_assert(kTPW <= 16, "one DMA piece per position group");
    constexpr int kHalo = NP * kQS;
    constexpr int kStage = kHalo + 64 * kPanelPieces;   // float4 slots per stage
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
    const int nS = s.cblocks * (16 / CH);  // K steps
    const int nCB = a.Cout_pad / 16;

    // ---- LDS-DMA descriptors: this wave's kTPW pieces of every K step ---------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)img * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, s.cblocks * nCB * 16384, 0x00020000);
    auto halo_voff = [&](int j) {  // byte offset of this lane's texel in piece j of a plane: slot L of [row][17]
        const int L = 64 * j + lane;
        const int row = L / kWinoHalf, hxh = L - row * kWinoHalf;
        const int iy = y0 - 1 + (row >> 1), ix = x0 - 1 + 2 * hxh + (row & 1);
        const bool ok = (L < kPlane) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
        return ok ? (iy * s.W + ix) * s.cs * 4 : kOob;  // out of the descriptor's range = zero padding
    };
    int voffF[kFull > 0 ? kFull : 1];
#pragma unroll
    for (int f = 0; f < kFull; ++f) voffF[f] = halo_voff(f * WAVES + wave);
    const int jx = kFull * WAVES + (wave * kExtra) / NP, qx = (wave * kExtra) % NP;  // this wave's share of the remaining j's
    const int voffX = kExtra ? halo_voff(jx) : 0;
    const int voffP = 16 * lane;
    // k-th DMA piece of this wave for K step c into `stage`; wave-uniform operands forced into SGPRs
    auto issue_one = [&](int k, int c, int stage) {
        int so, lo, vo;
        bool halo = true;
        if (k < NP * kFull) {
            const int f = k / NP, q = k % NP;
            vo = voffF[f]; so = 16 * q + 4 * CH * c; lo = q * kQS + 64 * (f * WAVES + wave);
        } else if (k < NP * kFull + kExtra) {
            const int q = qx + (k - NP * kFull);
            vo = voffX; so = 16 * q + 4 * CH * c; lo = q * kQS + 64 * jx;
        } else {
            const int i = wave * kPanel + (k - NP * kFull - kExtra);
            halo = false;
            vo = voffP; so = ((c * nCB + cb0) * (4 * KS) + i) * 1024; lo = kHalo + 64 * i;
        }
        dma16(halo ? rsA : rsW, lds + __builtin_amdgcn_readfirstlane(stage * kStage + lo), vo, __builtin_amdgcn_readfirstlane(so));
    };

    f32x4 acc[16][NCO];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's KS channels of a texel: plane (KS h) / 4, byte (KS h % 4) * 4 of the 16-byte slot; patch origin = rows
    // 4*wave .. of the plane (hy = 2*wave + y, column parity x & 1), column n + (x >> 1)
    const int patch0 = ((((KS * h) >> 2) * kQS + (4 * wave) * kWinoHalf + n) * 16 + ((KS * h) & 3) * 4);  // bytes
    const int frag0 = kHalo * 16 + (h * 16 + n) * (4 * KS);                                                   // bytes

    // One K step: 16 position groups of KS*NCO MFMAs, in the order xi = 1, 2, 0, 3 (rows 1 and 2 of the patch feed the
    // first eight groups, row 0 / row 3 are read while those run).  Every LDS read is issued kAhead groups ahead of its
    // first use and the DMA pieces of the NEXT step are spread over the first groups, so that neither the LDS latency nor
    // the ~100-cycle issue cost of an LDS-DMA instruction opens a gap in the matrix pipe; sched_barrier keeps the compiler
    // from sinking the reads back to their uses.
    constexpr int kAhead = CH == 16 ? 2 : 3;
    auto compute = [&](int stage, int cnext) {
        const char *sH = reinterpret_cast<const char *>(lds + stage * kStage) + patch0;
        const char *sW = reinterpret_cast<const char *>(lds + stage * kStage) + frag0;
        vec d[4][4], A[kAhead + 1][NCO], r[4];
        auto rd_frag = [&](int g) {
            const int p = ((g >> 2) == 0 ? 4 : (g >> 2) == 1 ? 8 : (g >> 2) == 2 ? 0 : 12) + (g & 3);
#pragma unroll
            for (int j = 0; j < NCO; ++j) A[g % (kAhead + 1)][j] = *reinterpret_cast<const vec *>(sW + (j * 16 + p) * 64 * (4 * KS));
        };
        auto rd_row = [&](int y) {
#pragma unroll
            for (int x = 0; x < 4; ++x) d[y][x] = *reinterpret_cast<const vec *>(sH + ((2 * y + (x & 1)) * kWinoHalf + (x >> 1)) * 16);
        };
        rd_frag(0);
        rd_row(1);
        rd_row(2);
#pragma unroll
        for (int g = 1; g < kAhead; ++g) rd_frag(g);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int xi = (g >> 2) == 0 ? 1 : (g >> 2) == 1 ? 2 : (g >> 2) == 2 ? 0 : 3, nu = g & 3;
            const int p = 4 * xi + nu;
            if (g + kAhead < 16) rd_frag(g + kAhead);
            if (g == 1) rd_row(0);
            if (g == 5) rd_row(3);
            if (g < kTPW && cnext >= 0) issue_one(g, cnext, cnext & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (nu == 0) {  // B^T d: rows combined for this xi
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    r[x] = xi == 0 ? d[0][x] - d[2][x] : xi == 1 ? d[1][x] + d[2][x] : xi == 2 ? d[2][x] - d[1][x] : d[1][x] - d[3][x];
            }
            const vec v = nu == 0 ? r[0] - r[2] : nu == 1 ? r[1] + r[2] : nu == 2 ? r[2] - r[1] : r[1] - r[3];
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int j = 0; j < NCO; ++j)
                    acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g % (kAhead + 1)][j][k], v[k], acc[p][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- K loop: step c+1 lands in the other stage while step c is computed on; one barrier per step ----------------
#pragma unroll
    for (int k = 0; k < kTPW; ++k) issue_one(k, 0, 0);
    __syncthreads();  // (the compiler drains vmcnt before the barrier: the DMA of every wave has landed)
#pragma unroll 1
    for (int c = 0; c < nS; ++c) {
        compute(c & 1, c + 1 < nS ? c + 1 : -1);
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

template <int WAVES, int NCO, int CH>
int launch_wino(const ConvArgs &a, int N, hipStream_t st) {
    WinoArgs wa{a, (a.Wo + kWinoTileW - 1) / kWinoTileW, (a.Ho + 2 * WAVES - 1) / (2 * WAVES)};
    wa.c.NT = a.Cout / (16 * NCO);
    const long long blocks = (long long)N * wa.tiles_x * wa.tiles_y * wa.c.NT;
    if (blocks >= (1ll << 31)) return IDH_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_wino_k<WAVES, NCO, CH>), dim3((unsigned)blocks), dim3(64 * WAVES), 0, st, wa);
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
    if (rows == 108) return launch_wino<4, 2, 8>(a, N, st);   // 8-row tiles, 8-channel K steps: 56 KiB of LDS -> 2 workgroups / CU
    if (rows == 8) return launch_wino<4, 2, 16>(a, N, st);
    return launch_wino<8, 2, 16>(a, N, st);
}

}  // namespace idh_conv

extern "C" size_t idh_packed_wino_weight_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)((Cin + 15) & ~15) * ((Cout + 15) & ~15) * 16;
}

extern "C" int idh_pack_conv_weight_wino(const float *w, float *dst, int Cout, int Cin, int ch, void *stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0 || (ch != 16 && ch != 8)) return IDH_EINVAL;
    const int nC = (Cin + 15) / 16, nCB = (Cout + 15) / 16;
    const long long total = (long long)nC * nCB * 4096;
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_wino_weight_k, dim3(grid), dim3(256), 0, idh_stream(stream), w, dst, Cout, Cin, nC * (16 / ch), nCB, ch / 4);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}
