// 3x3 stride-1 convolution as Winograd F(4x4, 3x3) on the fp32 matrix cores of gfx950.
//
// Replaces the same nn.Conv2d calls as conv3x3_wino_k / conv3x3_lds_k (BasicBlock convs, reference modules/layers.py:59-95,
// the 3x3 stride-1 layers of CVEncoder / BDDecoderPP / DepthDecoderPP, modules/networks.py:20-215) for the LARGE maps that
// carry most of the flops (64->64 and 192->64 at 192x256 and 96x128).  F(4x4,3x3) needs 36 multiplies per 4x4 output tile and
// channel pair instead of 64 for four F(2x2) tiles (144 direct): 1.78x fewer v_mfma_f32_16x16x4_f32 than conv3x3_wino_k,
// fp32 operands and fp32 accumulation throughout.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 6x6 input patch, g: 3x3 filter, Y: 4x4 outputs
//
// Interpolation points {0, +-1/2, +-2, inf} (not the textbook {0, +-1, +-2}): same operation count, about half the fp32
// error (2-4e-6 of the output scale against fp64 at 64-192 input channels; F(2x2): 3-5e-7, direct kernel: 1e-6):
//   B^T = [1 0 -17/4 0 1 0; 0 -2 -4 1/2 1 0; 0 2 -4 -1/2 1 0; 0 -1/2 -1/4 2 1 0; 0 1/2 -1/4 -2 1 0; 0 1 0 -17/4 0 1]
//   G   = [1 0 0; -8/15 -4/15 -2/15; -8/15 4/15 -2/15; 1/30 1/15 2/15; 1/30 -1/15 2/15; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1/2 -1/2 2 -2 0; 0 1/4 1/4 4 4 0; 0 1/8 -1/8 8 -8 1]
//
// Design (one wave per SIMD, 512 registers per lane):
// * The 36 element-wise products over the input channels are 36 independent GEMMs M[pos][co, tile] = sum_ci U[pos][co, ci]
//   V[pos][ci, tile], issued as D^T = U . V^T (weights = MFMA A operand).  Lane (n = lane & 15, h = lane >> 4) holds, as the B
//   operand of MFMA k-step ks, channel 8c + 4ks + h of TILE n — so the lane that reads the 6x6 patch of tile n for that channel
//   (36 ds_read_b32) transforms it IN REGISTERS (144 FMAs) and every transformed value is directly the B operand of two MFMAs
//   (two 16-channel output blocks).  No transformed input goes through LDS.
// * A wave owns 16 tiles in a row (64 x 4 output pixels) x 32 output channels = 36 positions x 2 accumulator quads = 288
//   accumulator registers; a workgroup = 4 waves stacked vertically (64 x 16 pixels x 32 channels), ONE workgroup per CU.
// * K loop in stages of 8 input channels (two passes of 4 = one MFMA k-step each).  Per stage the 18 x 66 halo (32 B per texel)
//   and the 36-position weight panel (36 KiB, packed in exactly the order the A fragments are read) are copied global ->
//   registers -> LDS (buffer_load_dwordx4 + ds_write_b128, three batches spread over the stage: with one wave per SIMD an
//   LDS-DMA instruction's ~100-cycle issue stall would come straight out of the matrix pipe).  Out-of-image texels are out of
//   the buffer descriptor's range and arrive as zeros = zero padding.  Two panel buffers + two halo buffers = 152 KiB of LDS.
// * The transform is software-pipelined across passes with two 6x6 register sets: while pass s multiplies row xi of W(s)
//   (horizontal transform of one row -> 6 B operands -> 12 MFMAs), column xi of the NEXT pass's patch is transformed
//   vertically in place, and the row of W(s) just consumed is refilled with the patch of the pass after that.  The halo is
//   therefore staged TWO stages ahead of the MFMAs, the panel one stage ahead; the stream of (tile, stage) pairs of a
//   persistent workgroup runs across tile boundaries without refilling the pipeline.
// * Epilogue: output transform (120 vector ops per channel quad) in registers, + bias + residual, activation, 16-byte NHWC
//   stores (a lane holds 4 consecutive channels of the 4x4 pixels of its tile).
#include <type_traits>

#include "conv_args.h"
#include "../../include/idh_ops.h"

using namespace idh_conv;

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(3))) float lds_cfloat;
typedef const __attribute__((address_space(3))) f32x4 lds_cf32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

constexpr int kOob = 0x7fffffff;
constexpr int kTW = 64, kTH = 16;          // output pixels of a workgroup tile (16 tiles x 4 waves of 4x4 pixels)
constexpr int kSlots = 17;                 // column quads of the halo (66 columns)
constexpr int kHaloBytes = 2560 * 16;      // 18 rows x 4 column phases x 17 quads x 32 B = 39168 B, rounded up to 10 granules per thread
constexpr int kPanelFloats = 36 * 32 * 8;  // one stage's weight panel: 36 positions x 32 output channels x 8 input channels
constexpr int kPanelBytes = kPanelFloats * 4;
constexpr int kH0 = 0, kH1 = kHaloBytes, kU0 = 2 * kHaloBytes, kU1 = 2 * kHaloBytes + kPanelBytes;
constexpr int kLdsBytes = 2 * kHaloBytes + 2 * kPanelBytes;  // 155648
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

// OIHW 3x3 -> U = G g G^T in A-fragment order:
//   dst[stage c][co tile nt (32)][ks 2][cb 2][group g 9][lane 64][e 4] = U[pos = 4g + e][co = 32 nt + 16 cb + (lane & 15)][ci = 8c + 4ks + (lane >> 4)]
__global__ __launch_bounds__(256) void pack_wino4_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout, int Cin, int nS, int NT) {
    const long long total = (long long)nS * NT * kPanelFloats;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int e = (int)(t & 3), lane = (int)((t >> 2) & 63);
        long long r = t >> 8;
        const int g = (int)(r % 9); r /= 9;
        const int cb = (int)(r & 1), ks = (int)((r >> 1) & 1);
        r >>= 2;
        const int nt = (int)(r % NT), c = (int)(r / NT);
        const int pos = 4 * g + e, co = 32 * nt + 16 * cb + (lane & 15), ci = 8 * c + 4 * ks + (lane >> 4);
        double u = 0.0;
        if (co < Cout && ci < Cin) {
            const float *gw = w + ((size_t)co * Cin + ci) * 9;
            const int xi = pos / 6, nu = pos % 6;
            const double G[6][3] = {{1.0, 0.0, 0.0},
                                    {-8.0 / 15, -4.0 / 15, -2.0 / 15},
                                    {-8.0 / 15, 4.0 / 15, -2.0 / 15},
                                    {1.0 / 30, 1.0 / 15, 2.0 / 15},
                                    {1.0 / 30, -1.0 / 15, 2.0 / 15},
                                    {0.0, 0.0, 1.0}};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) u += G[xi][a] * (double)gw[a * 3 + b] * G[nu][b];
        }
        dst[t] = (float)u;
    }
}

struct Wino4Args {
    ConvArgs c;
    int tiles_x, tiles_y;
    int tiles;  // N * tiles_y * tiles_x * NT
};

// 1-D input transform B^T (6 -> 6), in place: 12 FMA-class operations
__device__ __forceinline__ void bt6(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5) {
    const float a = __builtin_fmaf(-4.f, d2, d4), b = __builtin_fmaf(-4.f, d1, d3);
    const float c = __builtin_fmaf(-0.25f, d2, d4), e = __builtin_fmaf(-0.25f, d1, d3);
    const float t0 = __builtin_fmaf(-4.25f, d2, d0) + d4;
    const float t5 = __builtin_fmaf(-4.25f, d3, d1) + d5;
    d0 = t0;
    d1 = __builtin_fmaf(0.5f, b, a);
    d2 = __builtin_fmaf(-0.5f, b, a);
    d3 = __builtin_fmaf(2.f, e, c);
    d4 = __builtin_fmaf(-2.f, e, c);
    d5 = t5;
}

__device__ __forceinline__ f32x4 fma4(float s, f32x4 x, f32x4 y) {
    return (f32x4){__builtin_fmaf(s, x[0], y[0]), __builtin_fmaf(s, x[1], y[1]), __builtin_fmaf(s, x[2], y[2]), __builtin_fmaf(s, x[3], y[3])};
}
// 1-D output transform A^T (6 -> 4): 12 vector operations
__device__ __forceinline__ void at6(f32x4 m0, f32x4 m1, f32x4 m2, f32x4 m3, f32x4 m4, f32x4 m5, f32x4 &y0, f32x4 &y1, f32x4 &y2, f32x4 &y3) {
    const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = (m0 + s1) + s2;
    y1 = fma4(0.5f, d1, d2 * 2.f);
    y2 = fma4(0.25f, s1, s2 * 4.f);
    y3 = fma4(0.125f, d1, fma4(8.f, d2, m5));
}

// The 72 accumulator quads of a wave are 288 registers: more than the 256 AGPRs.  hipcc, left to itself, selects the AGPR form for
// every MFMA and shuttles the surplus through v_accvgpr copies (600 copies + scratch per 288 MFMAs).  So the MFMAs are issued
// through inline asm with the register file pinned per accumulator: positions 0..31 in AGPRs, positions 32..35 in VGPRs.
// (No software hazard applies: gfx950 needs no wait states between a VALU write and an MFMA SrcA/B read, and an accumulator is
// re-used 72 MFMAs later; the epilogue waits explicitly before it reads them.)
template <bool AGPR>
__device__ __forceinline__ void mfma_pinned(f32x4 &acc, float a, float b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int NCO>
__global__ __launch_bounds__(256, 1) void conv3x3_wino4_k(const Wino4Args wa) {
    static_assert(NCO == 2, "two 16-channel output blocks per wave");
    __shared__ __attribute__((aligned(16))) char lds_raw[kLdsBytes];
    lds_char *lds = (lds_char *)lds_raw;

    const ConvArgs &a = wa.c;
    const ConvSrc &s = a.s[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, h = lane >> 4;
    const int nS = s.cblocks * 2;  // stages of 8 input channels (always even)
    const int NT = a.NT;

    // ---- persistent workgroup: every XCD walks one contiguous range of tiles (channel tile fastest: the two channel tiles of
    // an input tile run side by side on one L2)
    const int T = wa.tiles;
    int t_cur, t_end, t_stride;
    {
        const unsigned vblock = blockIdx.x, vgrid = gridDim.x;
        if ((vgrid & 7) == 0) {
            const int xcd = vblock & 7;
            t_stride = vgrid >> 3;
            t_cur = (int)((long long)T * xcd / 8) + (int)(vblock >> 3);
            t_end = (int)((long long)T * (xcd + 1) / 8);
        } else {
            t_cur = vblock; t_end = T; t_stride = vgrid;
        }
    }
    if (t_cur >= t_end) return;

    struct Tile { int img, y0, x0, nt; };
    auto decode = [&](int t) {
        unsigned blk = (unsigned)t;
        Tile r;
        r.nt = blk % NT; blk /= NT;
        const int tx = blk % wa.tiles_x; blk /= wa.tiles_x;
        const int ty = blk % wa.tiles_y;
        r.img = blk / wa.tiles_y;
        r.y0 = ty * kTH; r.x0 = tx * kTW;
        return r;
    };

    // ---- global -> LDS staging ----------------------------------------------------------------------------------------------
    // Halo texel (row r 0..17, column col 0..65) lives at texel index p = (4 r + (col & 3)) * 17 + (col >> 2): the 16 tiles of a wave
    // (4 columns apart) read consecutive texels.  A texel = 32 B = two 16-byte granules (channel quads 0 / 1 of the stage), quad q
    // in granule q ^ swz, swz = (col >> 4) & 1: a ds_read_b32 of 32 lanes (16 tiles x 2 channels) is then 2-way bank conflicted
    // (the minimum for 8 bytes out of every 32).  Thread t copies granules t, t + 256, ..: 10 halo granules and 9 panel granules.
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, nS * NT * kPanelBytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsH;
    int voffH[10];
    auto set_halo_cursor = [&](const Tile &t) {
        rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)t.img * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int G = tid + 256 * k;
            const int p = G >> 1, half = G & 1;
            const int cq = p % kSlots, rc = p / kSlots;
            const int cm = rc & 3, r = rc >> 2;
            const int col = 4 * cq + cm;
            const int q = half ^ ((cq >> 2) & 1);
            const int iy = t.y0 - 1 + r, ix = t.x0 - 1 + col;
            const bool ok = (r < 18) & (col < 66) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            voffH[k] = ok ? (iy * s.W + ix) * s.cs * 4 + 16 * q : kOob;
        }
    };
    const int voffU = tid * 16;
    // load j of a stage: j < 9: panel granule row j of (stage cu, channel tile ntu); j >= 9: halo granule row j - 9 of stage ch
    auto ld = [&](int j, int cu, int ntu, int ch) -> f32x4 {
        if (j < 9) {
            const int so = __builtin_amdgcn_readfirstlane((cu * NT + ntu) * kPanelBytes + 4096 * j);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffU, so, 0));
        }
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, voffH[j - 9], __builtin_amdgcn_readfirstlane(32 * ch), 0));
    };
    auto st = [&](int j, int ubuf, int hbuf, f32x4 v) {
        const int off = j < 9 ? ubuf + 4096 * j : hbuf + 4096 * (j - 9);
        *(lds_f32x4 *)(lds + off + tid * 16) = v;
    };

    // ---- LDS read addresses of this lane ---------------------------------------------------------------------------------------
    // patch element (i, c) of tile n, wave row block `wave`: texel p = (4 (4 wave + i) + (c & 3)) * 17 + n + (c >> 2)
    int rbase[2][2];  // [ks][c >> 2]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) rbase[ks][dc] = 32 * (16 * wave * kSlots + n) + 4 * h + 16 * (ks ^ (((n + dc) >> 2) & 1));
    const int ubase = lane * 16;

    auto rd_row = [&](float (&X)[6][6], int i, int ks, int hbuf) {  // patch row i of pass ks
#pragma unroll
        for (int c = 0; c < 6; ++c) X[i][c] = *(lds_cfloat *)(lds + hbuf + rbase[ks][c >> 2] + 32 * ((4 * i + (c & 3)) * kSlots + (c >> 2)));
    };

    f32x4 acc[36][NCO];
    float A_[6][6], B_[6][6];  // the two patch / W register sets of the transform pipeline
    float v[6];                // B operands of the row about to be multiplied

    // ---- prologue: halo(0), halo(1), panel(0) of the first tile; W(0, ks 0) and the raw patch of (0, ks 1) -------------------------
    Tile cur = decode(t_cur);
    set_halo_cursor(cur);
    {
        f32x4 tmp[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) tmp[j] = ld(9 + j, 0, cur.nt, 0);
#pragma unroll
        for (int j = 0; j < 10; ++j) st(9 + j, kU0, kH0, tmp[j]);
#pragma unroll
        for (int j = 0; j < 10; ++j) tmp[j] = ld(9 + j, 0, cur.nt, 1);
#pragma unroll
        for (int j = 0; j < 10; ++j) st(9 + j, kU0, kH1, tmp[j]);
#pragma unroll
        for (int j = 0; j < 9; ++j) tmp[j] = ld(j, 0, cur.nt, 0);
#pragma unroll
        for (int j = 0; j < 9; ++j) st(j, kU0, kH0, tmp[j]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) rd_row(A_, i, 0, kH0);
#pragma unroll
    for (int i = 0; i < 6; ++i) rd_row(B_, i, 1, kH0);
#pragma unroll
    for (int j = 0; j < 6; ++j) bt6(A_[0][j], A_[1][j], A_[2][j], A_[3][j], A_[4][j], A_[5][j]);
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = A_[0][j];
    bt6(v[0], v[1], v[2], v[3], v[4], v[5]);

    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);

#pragma unroll 1
    for (;;) {
        const int t_next = t_cur + t_stride;
        const bool has_next = t_next < t_end;
        const Tile nxt = has_next ? decode(t_next) : cur;  // (past the end: re-read this tile's first stages, never used)
#pragma unroll
        for (int p = 0; p < 36; ++p)
#pragma unroll
            for (int j = 0; j < NCO; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // One stage S = (tile, c): multiplies W(S, 0) and W(S, 1) with panel(S) [U buffer PAR]; transforms the patches of (S, 1)
        // and (S + 1, 0); reads the patches of (S + 1, 0 / 1) from halo(S + 1) [H buffer PAR ^ 1]; copies panel(S + 1) into U
        // buffer PAR ^ 1 and halo(S + 2) into H buffer PAR.  nS is even, so a tile always starts at parity 0.
        auto stage = [&](auto parc, const int c) {
            constexpr int PAR = decltype(parc)::value;
            constexpr int kUr = PAR ? kU1 : kU0, kUw = PAR ? kU0 : kU1;
            constexpr int kHr = PAR ? kH0 : kH1, kHw = PAR ? kH1 : kH0;
            if (PAR == 0 && c + 2 == nS) set_halo_cursor(nxt);  // from here on the halo copies belong to the next tile
            const bool un = c + 1 >= nS;
            const int cu = un ? 0 : c + 1, ntu = un ? nxt.nt : cur.nt;
            const int ch = c + 2 >= nS ? c + 2 - nS : c + 2;
            f32x4 stg[7];
            f32x4 Af[NCO][9];
            auto rd_frag = [&](int ks, int g) {
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb) Af[cb][g] = *(lds_cf32x4 *)(lds + kUr + ubase + ((ks * 2 + cb) * 9 + g) * 1024);
            };
            rd_frag(0, 0);
            rd_frag(0, 1);
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int pass = it / 6, xi = it % 6;
                float(&X)[6][6] = pass == 0 ? A_ : B_;
                float(&Y)[6][6] = pass == 0 ? B_ : A_;
                // copies: three batches of 7 / 6 / 6 granules, issued at it = 0 / 4 / 8, written to LDS three iterations later
                if (it == 0) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) stg[j] = ld(j, cu, ntu, ch);
                }
                if (it == 4) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) stg[j] = ld(7 + j, cu, ntu, ch);
                }
                if (it == 8) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) stg[j] = ld(13 + j, cu, ntu, ch);
                }
                // A fragments first needed by the next iteration (those of the next stage wait for its barrier)
                if (xi == 0) rd_frag(pass, 2);
                if (xi == 1) { rd_frag(pass, 3); rd_frag(pass, 4); }
                if (xi == 2) rd_frag(pass, 5);
                if (xi == 3) { rd_frag(pass, 6); rd_frag(pass, 7); }
                if (xi == 4) rd_frag(pass, 8);
                if (it == 5) { rd_frag(1, 0); rd_frag(1, 1); }
                // row xi of X has been consumed (its B operands are in v): refill it with the patch of the pass after next
                rd_row(X, xi, pass, kHr);
                // vertical transform of column xi of the next pass's patch
                bt6(Y[0][xi], Y[1][xi], Y[2][xi], Y[3][xi], Y[4][xi], Y[5][xi]);
                // 12 MFMAs of row xi
                float vc[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) vc[j] = v[j];
                // horizontal transform of the NEXT row (row xi + 1 of X, or row 0 of Y which is complete after this iteration's column)
                if (xi < 5) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) v[j] = X[xi + 1][j];
                } else {
#pragma unroll
                    for (int j = 0; j < 6; ++j) v[j] = Y[0][j];
                }
                bt6(v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
                for (int nu = 0; nu < 6; ++nu)
#pragma unroll
                    for (int cb = 0; cb < NCO; ++cb) {
                        const int p = 6 * xi + nu;
                        if (p < 32) mfma_pinned<true>(acc[p][cb], Af[cb][p >> 2][p & 3], vc[nu]);
                        else mfma_pinned<false>(acc[p][cb], Af[cb][p >> 2][p & 3], vc[nu]);
                    }
                if (it == 3) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) st(j, kUw, kHw, stg[j]);
                }
                if (it == 7) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) st(7 + j, kUw, kHw, stg[j]);
                }
                if (it == 11) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) st(13 + j, kUw, kHw, stg[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };

#pragma unroll 1
        for (int c = 0; c < nS; c += 2) {
            stage(std::integral_constant<int, 0>{}, c);
            stage(std::integral_constant<int, 1>{}, c + 1);
        }

        // ---- epilogue: Y = A^T M A per 16-channel block; lane = 4 consecutive channels of the 4x4 pixels of tile n --------------
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results (inline asm: no compiler-inserted wait states)
        {
            const int n0 = 32 * cur.nt;
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.img * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)cur.img * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                                  a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
            const int oy0 = cur.y0 + 4 * wave, ox0 = cur.x0 + 4 * n;
            // LeakyReLU / identity only (wino4_supported): v < 0 ? v * slope : v with slope = 1 for "no activation" — branch-free,
            // and no inlined expm1f per output element (ELU layers stay on the other kernels)
            const float slope_eff = a.act == IDH_ACT_LRELU ? a.slope : 1.f;
            auto act4 = [&](f32x4 o) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? o[e] * slope_eff : o[e];
                return o;
            };
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb) {
                const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (n0 + 16 * cb + 4 * h) * 4, 0, 0));
                // horizontal pass (over nu) row by row; the 24 intermediate quads are parked in the AGPRs the row's accumulators leave
                f32x4 u[6][4];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) {
                    at6(acc[6 * xi][cb], acc[6 * xi + 1][cb], acc[6 * xi + 2][cb], acc[6 * xi + 3][cb], acc[6 * xi + 4][cb], acc[6 * xi + 5][cb], u[xi][0], u[xi][1], u[xi][2], u[xi][3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(u[xi][j]));
                }
                // vertical pass (over xi) per output column j: 4 pixels, finished and stored at once
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 r[4];
                    int voff[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool ok = (oy0 + i < a.Ho) & (ox0 + j < a.Wo);
                        const int pix = ok ? (oy0 + i) * a.Wo + ox0 + j : -1;
                        voff[i] = pix;
                        r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, ok ? (pix * a.res_cs + n0 + 16 * cb + 4 * h) * 4 : kOob, 0, 0));
                    }
                    f32x4 y[4];
                    at6(u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j], y[0], y[1], y[2], y[3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 o = act4(y[i] + b4 + r[i]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, voff[i] >= 0 ? (voff[i] * a.out_cs + n0 + 16 * cb + 4 * h) * 4 : kOob, 0, 0);
                    }
                }
            }
        }
        if (!has_next) break;
        t_cur = t_next;
        cur = nxt;
    }
}

template <int NCO>
int wino4_args(const ConvArgs &a, int N, Wino4Args &wa) {
    wa = Wino4Args{a, (a.Wo + kTW - 1) / kTW, (a.Ho + kTH - 1) / kTH, 0};
    wa.c.NT = a.Cout / (16 * NCO);
    const long long tiles = (long long)N * wa.tiles_x * wa.tiles_y * wa.c.NT;
    if (tiles >= (1ll << 31)) return IDH_EUNSUPPORTED;
    wa.tiles = (int)tiles;
    return IDH_OK;
}

int wino4_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        cus = c > 0 ? c : 256;
    }
    return cus;
}

}  // namespace

namespace idh_conv {

bool wino4_supported(const ConvArgs &a) {
    const ConvSrc &s = a.s[0];
    return !a.s[1].in && (a.act == IDH_ACT_NONE || a.act == IDH_ACT_LRELU) && s.ks == 3 && s.stride == 1 && s.pad_mode == IDH_PAD_ZEROS && !s.up_in[0] && !s.norm && a.S == 1 && (a.Cout % 32) == 0 &&
           (long long)s.H * s.W * s.cs * 4 < (1ll << 31) && (long long)a.Ho * a.Wo * a.out_cs * 4 < (1ll << 31) &&
           (!a.res || (long long)a.Ho * a.Wo * a.res_cs * 4 < (1ll << 31)) && (long long)s.cblocks * a.Cout_pad * 36 * 16 * 4 < (1ll << 31);
}

int launch_conv_wino4(const ConvArgs &a, int N, hipStream_t st) {
    if (!wino4_supported(a)) return IDH_EUNSUPPORTED;
    Wino4Args wa;
    if (int rc = wino4_args<2>(a, N, wa)) return rc;
    long long grid = wino4_cus();  // one persistent workgroup per CU (152 KiB of LDS, 1 wave per SIMD)
    if (grid > wa.tiles) grid = wa.tiles >= 8 ? wa.tiles / 8 * 8 : wa.tiles;
    hipLaunchKernelGGL((conv3x3_wino4_k<2>), dim3((unsigned)grid), dim3(256), 0, st, wa);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

}  // namespace idh_conv

extern "C" size_t idh_packed_wino4_weight_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)((Cin + 15) & ~15) * ((Cout + 31) & ~31) * 36;
}

extern "C" int idh_pack_conv_weight_wino4(const float *w, float *dst, int Cout, int Cin, void *stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0) return IDH_EINVAL;
    const int nS = ((Cin + 15) / 16) * 2, NT = (Cout + 31) / 32;
    const long long total = (long long)nS * NT * kPanelFloats;
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_wino4_weight_k, dim3(grid), dim3(256), 0, idh_stream(stream), w, dst, Cout, Cin, nS, NT);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}
