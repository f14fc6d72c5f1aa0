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
//   operand of MFMA k-step ks, channel 8c + 2h + ks of TILE n — so the lane that reads the 6x6 patch of tile n for that channel
//   PAIR (36 ds_read_b64 per 8-channel stage, conflict free) transforms it IN REGISTERS (144 FMAs per channel) and every
//   transformed value is directly the B operand of two MFMAs (two 16-channel output blocks).  No transformed input goes through LDS.
// * A wave owns 16 tiles in a row (64 x 4 output pixels) x 32 output channels = 36 positions x 2 accumulator quads = 288
//   accumulator registers; a workgroup = 4 waves stacked vertically (64 x 16 pixels x 32 channels), ONE workgroup per CU.
// * K loop in stages of 8 input channels (two passes of 4 = one MFMA k-step each).  Per stage the 18 x 66 halo (32 B per texel)
//   and the 36-position weight panel (36 KiB, packed in exactly the order the A fragments are read) are copied global ->
//   registers -> LDS (buffer_load_dwordx4 + ds_write_b128, three batches spread over the stage: with one wave per SIMD an
//   LDS-DMA instruction's ~100-cycle issue stall would come straight out of the matrix pipe).  Out-of-image texels are out of
//   the buffer descriptor's range and arrive as zeros = zero padding.  Two panel buffers + two halo buffers = 152 KiB of LDS.
// * Two 6x6 register sets hold the stage's patch of the lane's two channels, and the two 1-D transforms are applied in OPPOSITE
//   order to them: set 0 (k-step 0) rows first (as a row arrives from LDS), then column by column in pass 0, each column giving
//   the 6 B operands of positions (0..5, nu); set 1 (k-step 1) columns first (during pass 0), then row by row in pass 1, each row
//   giving positions (xi, 0..5) — pass 1 refills the rows of BOTH sets, as they are consumed, with the next stage's patch (one
//   ds_read_b64 per texel).  So every iteration is 12 MFMAs + 24 transform operations + <= 6 LDS reads, written as 12 slots of
//   "one MFMA, two vector operations, at most one LDS / global access" separated by sched_barrier: fp32 MFMA and fp32 VALU share
//   the SIMD's datapath, but everything else (LDS, global, scalar, waits) issues in an MFMA's 32-cycle shadow only if it sits
//   BETWEEN two MFMAs (hipcc would otherwise put the 12 MFMAs back to back and all other work between the bursts: 8.8k instead
//   of ~6k cycles per stage).  The halo is staged TWO stages ahead of the MFMAs, the panel one stage ahead; the stream of
//   (tile, stage) pairs of a persistent workgroup runs across tile boundaries without refilling the pipeline, and the first copies
//   of a tile's first stage are issued BEFORE the previous tile's output stores (stores and loads retire through one in-order
//   counter: a wait for loads issued after a store burst would wait for the burst).
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
//   dst[stage c][co tile nt (32)][ks 2][cb 2][group g 9][lane 64][e 4] = U[pos = 4g + e][co = 32 nt + 16 cb + (lane & 15)][ci = 8c + 2 (lane >> 4) + ks],
//   pos = 6 nu + xi for ks = 0 and 6 xi + nu for ks = 1
__global__ __launch_bounds__(256) void pack_wino4_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout, int Cin, int nS, int NT) {
    const long long total = (long long)nS * NT * kPanelFloats;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int e = (int)(t & 3), lane = (int)((t >> 2) & 63);
        long long r = t >> 8;
        const int g = (int)(r % 9); r /= 9;
        const int cb = (int)(r & 1), ks = (int)((r >> 1) & 1);
        r >>= 2;
        const int nt = (int)(r % NT), c = (int)(r / NT);
        const int pos = 4 * g + e, co = 32 * nt + 16 * cb + (lane & 15), ci = 8 * c + 2 * (lane >> 4) + ks;
        double u = 0.0;
        if (co < Cout && ci < Cin) {
            const float *gw = w + ((size_t)co * Cin + ci) * 9;
            // k-step 0 is multiplied column by column (pos = 6 nu + xi), k-step 1 row by row (pos = 6 xi + nu): see the kernel's stage body
            const int xi = ks == 0 ? pos % 6 : pos / 6, nu = ks == 0 ? pos / 6 : pos % 6;
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

// The same transform in six steps of two operations each (in place), so that a step can sit in the shadow of one MFMA
struct Bt6Steps {
    float a, b, c, e, t0, t5;
    template <int K>
    __device__ __forceinline__ void step(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5) {
        if constexpr (K == 0) { a = __builtin_fmaf(-4.f, d2, d4); b = __builtin_fmaf(-4.f, d1, d3); }
        if constexpr (K == 1) { c = __builtin_fmaf(-0.25f, d2, d4); e = __builtin_fmaf(-0.25f, d1, d3); }
        if constexpr (K == 2) { t0 = __builtin_fmaf(-4.25f, d2, d0); t5 = __builtin_fmaf(-4.25f, d3, d1); }
        if constexpr (K == 3) { d0 = t0 + d4; d5 = t5 + d5; }
        if constexpr (K == 4) { d1 = __builtin_fmaf(0.5f, b, a); d2 = __builtin_fmaf(-0.5f, b, a); }
        if constexpr (K == 5) { d3 = __builtin_fmaf(2.f, e, c); d4 = __builtin_fmaf(-2.f, e, c); }
    }
};

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
template <bool AGPR, bool ZERO = false>
__device__ __forceinline__ void mfma_pinned(f32x4 &acc, float a, float b) {
    if constexpr (ZERO) {  // first product of a tile: C = 0 (no 288-register clear per tile)
        if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
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
    // Every workgroup runs the same instruction stream on the same amount of work, so all 256 of them would hit the stages whose halo
    // copies miss the L2 (a pixel's 64 channels are two 128-byte lines: every fourth 8-channel stage opens a new one) at the same
    // moment — a 39 MB burst at HBM every fourth stage and nothing in between.  So the K loop is ROTATED per workgroup: stage c of a
    // tile works on channel stage (c + rot) mod nS (a sum over channels in another order), rot = workgroup pair index, pairs
    // = the two channel tiles of one input tile, which share their L2 lines.
    const int rot = (int)(((blockIdx.x >> 3) >> 1) % (unsigned)nS);
    auto rot1 = [&](int c) { const int r = c + rot; return r >= nS ? r - nS : r; };  // c in [0, nS)

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
    // in granule q ^ swz, swz = (col >> 5) & 1: the 32 lanes of a ds_read_b64 (16 tiles x 2 channel pairs of one quad) cover all 64
    // banks.  Thread t copies granules t, t + 256, .. (10 of the halo: lanes 2i, 2i + 1 = the 32 contiguous bytes of one texel, i.e.
    // one cache-line request per texel — the number of lines in flight, not bytes, is what the memory pipeline limits) and 9 of the panel.
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
            const int q = half ^ ((cq >> 3) & 1);
            const int iy = t.y0 - 1 + r, ix = t.x0 - 1 + col;
            const bool ok = (r < 18) & (col < 66) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            voffH[k] = ok ? (iy * s.W + ix) * s.cs * 4 + 16 * q : kOob;
#ifdef IDH_ABL_W4_HALFHALO
            if (k & 1) voffH[k] = kOob;  // half the cache-line requests (timing experiment)
#endif
        }
    };
    const int voffU = tid * 16;
    // load j of a stage: j < 9: panel granule row j of (stage cu, channel tile ntu); j >= 9: halo granule row j - 9 of stage ch
    auto ld = [&](int j, int cu, int ntu, int ch) -> f32x4 {
#ifdef IDH_ABL_W4_NOLOAD
        return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
#ifdef IDH_ABL_W4_NOPANEL
        if (j < 9) return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
#ifdef IDH_ABL_W4_NOHALO
        if (j >= 9) return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
        if (j < 9) {
            const int so = __builtin_amdgcn_readfirstlane((cu * NT + ntu) * kPanelBytes + 4096 * j);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffU, so, 0));
        }
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, voffH[j - 9], __builtin_amdgcn_readfirstlane(32 * ch), 0));
    };
    auto st = [&](int j, int ubuf, int hbuf, f32x4 v) {
#ifdef IDH_ABL_W4_NOLDSW
        asm volatile("" ::"v"(v));
        return;
#endif
        const int off = j < 9 ? ubuf + 4096 * j : hbuf + 4096 * (j - 9);
        *(lds_f32x4 *)(lds + off + tid * 16) = v;
    };

    // ---- LDS read addresses of this lane ---------------------------------------------------------------------------------------
    // patch element (i, c) of tile n, wave row block `wave`: texel p = (4 (4 wave + i) + (c & 3)) * 17 + n + (c >> 2); the lane's
    // channel pair 2h, 2h + 1 = 8 bytes at offset 8 (h & 1) of granule (h >> 1) ^ swz
    int rbase[2];  // [c >> 2]
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) rbase[dc] = 32 * (16 * wave * kSlots + n) + 16 * ((h >> 1) ^ (((n + dc) >> 3) & 1)) + 8 * (h & 1);
    const int ubase = lane * 16;

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(3))) volatile f32x2 lds_cf32x2;  // volatile: hipcc otherwise pairs the reads into ds_read2_b64 (half rate, 8-bit offsets -> an address register per 2 KiB)
    auto rd_elem = [&](int i, int c, int hbuf, float &x0, float &x1) {  // patch element (i, c) of both channels
#ifdef IDH_ABL_W4_NORAW
        return;
#endif
        const f32x2 t = *(lds_cf32x2 *)(lds + hbuf + rbase[c >> 2] + 32 * ((4 * i + (c & 3)) * kSlots + (c >> 2)));
        x0 = t[0];
        x1 = t[1];
    };

    // Developer build (-DIDH_ABL_W4_TRACE, tools/abl_wino4.sh trace): every wave logs s_memtime along its SECOND tile into ConvArgs.ws
    // (160 x 8 bytes per wave: [0] tile start, [1 + 15 c + k] stage c < 8: k = 0 entry, 1 first operands ready, 2..13 iteration done,
    // 14 barrier passed; [125..129] epilogue: K loop done, first copies of the next tile issued, row pass of channel block 0 / 1 done, stored)
#ifdef IDH_ABL_W4_TRACE
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(a.ws) + ((size_t)blockIdx.x * 4 + wave) * 160;
    int tile_i = 0;
#define W4T(idx) do { if (tile_i == 1 && lane == 0) trace[(idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4T(idx) do { } while (0)
#endif
    f32x4 acc[36][NCO];
    float A_[6][6], B_[6][6];  // the stage's patch (then W) of the lane's even / odd channel
    float v[6];                // B operands of the row about to be multiplied
    f32x4 stg[2][5];           // copies in flight: two batches of 5 granules

    // ---- prologue: halo(0), halo(1), panel(0) of the first tile; the raw patch of stage 0 ---------------------------------------
    Tile cur = decode(t_cur);
    set_halo_cursor(cur);
    {
        f32x4 tmp[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) tmp[j] = ld(9 + j, 0, cur.nt, rot1(0));
#pragma unroll
        for (int j = 0; j < 10; ++j) st(9 + j, kU0, kH0, tmp[j]);
#pragma unroll
        for (int j = 0; j < 10; ++j) tmp[j] = ld(9 + j, 0, cur.nt, rot1(1));
#pragma unroll
        for (int j = 0; j < 10; ++j) st(9 + j, kU0, kH1, tmp[j]);
#pragma unroll
        for (int j = 0; j < 9; ++j) tmp[j] = ld(j, rot1(0), cur.nt, 0);
#pragma unroll
        for (int j = 0; j < 9; ++j) st(j, kU0, kH0, tmp[j]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c < 6; ++c) rd_elem(i, c, kH0, A_[i][c], B_[i][c]);
#ifndef IDH_ABL_W4_NOXFORM
#pragma unroll
    for (int i = 0; i < 6; ++i) bt6(A_[i][0], A_[i][1], A_[i][2], A_[i][3], A_[i][4], A_[i][5]);  // set 0: rows first
#endif
    // batch 0 of the first stage's copies (panel(1), halo(2)): issued here for the first tile, before the epilogue stores for the others
    // Copy batches of a stage: 5 / 5 / 5 / 4 granules in load order (panel rows 0..8, then halo rows 0..9)
    auto batch_j = [](int b, int k) { return 5 * b + k; };
    auto batch_n = [](int b) { return b == 3 ? 4 : 5; };
    auto issue_first = [&](const Tile &t) {
#pragma unroll
        for (int k = 0; k < 5; ++k) stg[0][k] = ld(batch_j(0, k), rot1(1), t.nt, rot1(2));
    };
    issue_first(cur);

    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);

#pragma unroll 1
    for (;;) {
        const int t_next = t_cur + t_stride;
        const bool has_next = t_next < t_end;
        const Tile nxt = has_next ? decode(t_next) : cur;  // (past the end: re-read this tile's first stages, never used)
        W4T(0);

        // One stage S = (tile, c): multiplies W(S) (sets A_, B_) with panel(S) [U buffer PAR]; reads the patch of S + 1 from halo(S + 1)
        // [H buffer PAR ^ 1]; copies panel(S + 1) into U buffer PAR ^ 1 and halo(S + 2) into H buffer PAR.  nS is even (and >= 4), so a
        // tile always starts at parity 0.  On entry: A_ = patch of S with the row transform applied, B_ = raw patch of S.
        auto stage = [&](auto parc, auto firstc, const int c) {
            constexpr int PAR = decltype(parc)::value;
            constexpr bool FIRST = decltype(firstc)::value;  // the tile's first stage: pass 0 starts the accumulators (C = 0)
            constexpr int kUr = PAR ? kU1 : kU0, kUw = PAR ? kU0 : kU1;
            constexpr int kHr = PAR ? kH0 : kH1, kHw = PAR ? kH1 : kH0;
            if (PAR == 0 && c + 2 == nS) set_halo_cursor(nxt);  // from here on the halo copies belong to the next tile
            const bool un = c + 1 >= nS;
            const int cu = rot1(un ? 0 : c + 1), ntu = un ? nxt.nt : cur.nt;
            const int ch = rot1(c + 2 >= nS ? c + 2 - nS : c + 2);
            [[maybe_unused]] const int tr0 = 1 + 15 * (c < 8 ? c : 8);  // (stages >= 8 overwrite a scratch slot range that the tool ignores)
            W4T(tr0);
            f32x4 Af[NCO][9];
            auto rd_frag = [&](int ks, int g, int cb) { Af[cb][g] = *(lds_cf32x4 *)(lds + kUr + ubase + ((ks * 2 + cb) * 9 + g) * 1024); };
            rd_frag(0, 0, 0); rd_frag(0, 0, 1); rd_frag(0, 1, 0); rd_frag(0, 1, 1);
            if (!FIRST) {  // (a tile's first stage: issued before the previous tile's epilogue)
#pragma unroll
                for (int k = 0; k < 5; ++k) stg[0][k] = ld(batch_j(0, k), cu, ntu, ch);
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) stg[1][k] = ld(batch_j(1, k), cu, ntu, ch);
            // column 0 of set 0 -> the first B operands
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = A_[i][0];
#ifndef IDH_ABL_W4_NOXFORM
            bt6(v[0], v[1], v[2], v[3], v[4], v[5]);
#endif
            __builtin_amdgcn_sched_barrier(0);
            W4T(tr0 + 1);
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int pass = it / 6, xi = it % 6;  // pass 0: xi = column of set 0; pass 1: xi = row of set 1
                float vc[6], vn[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) vc[j] = v[j];
                // the transform in progress in slots 0..5 (t1) and 6..11 (t2)
                Bt6Steps t1, t2;
                // t2 = final transform of the next line: pass 0: column xi + 1 of set 0 (after column 5: row 0 of set 1); pass 1: row xi + 1 of
                // set 1 (after row 5: nothing — set 0's last row gets its row transform instead, and the next stage starts with column 0)
                if (pass == 0) {
                    if (xi < 5) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) vn[i] = A_[i][xi + 1];
                    }
                } else if (xi < 5) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) vn[j] = B_[xi + 1][j];
                }
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    // ---- the MFMA of the slot
#ifndef IDH_ABL_W4_NOMFMA
                    {
                        const int cb = k & 1, l = k >> 1;             // l: index along the line being multiplied
                        const int q = 6 * xi + l;                       // packed position (column-major in pass 0, row-major in pass 1)
                        const int p = pass == 0 ? 6 * l + xi : q;       // accumulator = position (xi_w, nu_w) = 6 xi_w + nu_w
                        if (FIRST && pass == 0) {
                            if (p < 32) mfma_pinned<true, true>(acc[p][cb], Af[cb][q >> 2][q & 3], vc[l]);
                            else mfma_pinned<false, true>(acc[p][cb], Af[cb][q >> 2][q & 3], vc[l]);
                        } else {
                            if (p < 32) mfma_pinned<true>(acc[p][cb], Af[cb][q >> 2][q & 3], vc[l]);
                            else mfma_pinned<false>(acc[p][cb], Af[cb][q >> 2][q & 3], vc[l]);
                        }
                    }
#else
                    asm volatile("" ::"v"(vc[k >> 1]));
#endif
                    // ---- two transform operations
#ifndef IDH_ABL_W4_NOXFORM
                    if (k < 6) {
                        auto s1 = [&](auto kc) {
                            constexpr int K = decltype(kc)::value;
                            if (pass == 0) t1.template step<K>(B_[0][xi], B_[1][xi], B_[2][xi], B_[3][xi], B_[4][xi], B_[5][xi]);        // set 1: columns first
                            else if (xi >= 1) t1.template step<K>(A_[xi - 1][0], A_[xi - 1][1], A_[xi - 1][2], A_[xi - 1][3], A_[xi - 1][4], A_[xi - 1][5]);  // set 0 of S + 1: rows first
                        };
                        if (k == 0) s1(std::integral_constant<int, 0>{});
                        if (k == 1) s1(std::integral_constant<int, 1>{});
                        if (k == 2) s1(std::integral_constant<int, 2>{});
                        if (k == 3) s1(std::integral_constant<int, 3>{});
                        if (k == 4) s1(std::integral_constant<int, 4>{});
                        if (k == 5) s1(std::integral_constant<int, 5>{});
                    } else {
                        auto s2 = [&](auto kc) {
                            constexpr int K = decltype(kc)::value;
                            if (it == 5) {  // row 0 of set 1 (its column transform completed in slot 5)
                                if (K == 0) {
#pragma unroll
                                    for (int j = 0; j < 6; ++j) vn[j] = B_[0][j];
                                }
                                t2.template step<K>(vn[0], vn[1], vn[2], vn[3], vn[4], vn[5]);
                            } else if (it == 11) {  // row 5 of set 0 of S + 1 (read in slots 0..5)
                                t2.template step<K>(A_[5][0], A_[5][1], A_[5][2], A_[5][3], A_[5][4], A_[5][5]);
                            } else {
                                t2.template step<K>(vn[0], vn[1], vn[2], vn[3], vn[4], vn[5]);
                            }
                        };
                        if (k == 6) s2(std::integral_constant<int, 0>{});
                        if (k == 7) s2(std::integral_constant<int, 1>{});
                        if (k == 8) s2(std::integral_constant<int, 2>{});
                        if (k == 9) s2(std::integral_constant<int, 3>{});
                        if (k == 10) s2(std::integral_constant<int, 4>{});
                        if (k == 11) s2(std::integral_constant<int, 5>{});
                    }
#endif
                    // ---- at most one LDS / global access pair
                    // pass 1, slots 0..5: row xi of both sets has been consumed (set 0 in pass 0; set 1's B operands are in vc): next stage's patch
                    if (pass == 1 && k < 6) rd_elem(xi, k, kHr, A_[xi][k], B_[xi][k]);
                    // A fragments first needed by the next iteration (those of the next stage wait for its barrier)
                    {
                        const int gi = k - 6;  // slots 6..9
                        int g = -1;
                        if (xi == 0 && gi < 2) g = 2;
                        if (xi == 1) g = gi < 2 ? 3 : 4;
                        if (xi == 2 && gi < 2) g = 5;
                        if (xi == 3) g = gi < 2 ? 6 : 7;
                        if (xi == 4 && gi < 2) g = 8;
                        if (it == 5) g = gi < 2 ? 0 : 1;
                        if (gi >= 0 && gi < 4 && g >= 0 && it != 11) rd_frag(it == 5 ? 1 : pass, g, gi & 1);
                    }
                    // copies: four batches of 5 / 5 / 5 / 4 granules, two in flight: batches 0 and 1 are issued at the stage start, batch b is
                    // written to LDS in slots 0..4 of iteration 3 b + 2 and batch b + 2 issued into its registers.  (Measured and rejected,
                    // profiles/r04/experiments.md: halo batches first with a 6-8 iteration lead; the reload one iteration after the write.)
                    if ((it == 2 || it == 5 || it == 8 || it == 11) && k < batch_n(it / 3)) {
                        const int b = it / 3;
                        st(batch_j(b, k), kUw, kHw, stg[b & 1][k]);
                        if (b < 2 && k < batch_n(b + 2)) stg[b & 1][k] = ld(batch_j(b + 2, k), cu, ntu, ch);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#ifdef IDH_ABL_W4_TRACE
                    if ((it == 6 || it == 7) && c == 2) W4T(130 + 12 * (it - 6) + k);  // slot stamps of two iterations of stage 2
#endif
                }
                if (it != 11) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) v[j] = vn[j];
                }
                W4T(tr0 + 2 + it);
            }
            __syncthreads();
            W4T(tr0 + 14);
        };

        stage(std::integral_constant<int, 0>{}, std::true_type{}, 0);
        stage(std::integral_constant<int, 1>{}, std::false_type{}, 1);
#pragma unroll 1
        for (int c = 2; c < nS; c += 2) {
            stage(std::integral_constant<int, 0>{}, std::false_type{}, c);
            stage(std::integral_constant<int, 1>{}, std::false_type{}, c + 1);
        }
        W4T(125);
        issue_first(nxt);  // batch 0 of the next tile's first stage, ahead of this tile's output stores

        // ---- epilogue: Y = A^T M A per 16-channel block; lane = 4 consecutive channels of the 4x4 pixels of tile n --------------
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results (inline asm: no compiler-inserted wait states)
        W4T(126);
#ifdef IDH_ABL_W4_NOEPI
#pragma unroll
        for (int p = 0; p < 36; ++p)
#pragma unroll
            for (int j = 0; j < NCO; ++j) {
                if (p < 32) asm volatile("" ::"a"(acc[p][j]));
                else asm volatile("" ::"v"(acc[p][j]));
            }
#else
        {
            const int n0 = 32 * cur.nt;
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.img * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)cur.img * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                                  a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
            const int oy0 = cur.y0 + 4 * wave, ox0 = cur.x0 + 4 * n;
            const bool has_res = a.res != nullptr;
            // LeakyReLU / identity only (wino4_supported): v < 0 ? v * slope : v with slope = 1 for "no activation" — branch-free,
            // and no inlined expm1f per output element (ELU layers stay on the other kernels)
            const float slope_eff = a.act == IDH_ACT_LRELU ? a.slope : 1.f;
            auto act4 = [&](f32x4 o) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? o[e] * slope_eff : o[e];
                return o;
            };
            // horizontal pass (over nu) row by row for both channel blocks; the 2 x 24 intermediate quads are parked in the AGPRs the
            // accumulators leave
            f32x4 b4[NCO], u[NCO][6][4];
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb) {
                b4[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (n0 + 16 * cb + 4 * h) * 4, 0, 0));
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) {
                    at6(acc[6 * xi][cb], acc[6 * xi + 1][cb], acc[6 * xi + 2][cb], acc[6 * xi + 3][cb], acc[6 * xi + 4][cb], acc[6 * xi + 5][cb], u[cb][xi][0], u[cb][xi][1], u[cb][xi][2],
                        u[cb][xi][3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(u[cb][xi][j]));
                }
                W4T(127 + cb);
            }
            // vertical pass (over xi) per output column j: 4 pixels x 2 channel blocks, finished and stored at once — the two 64-byte halves
            // of a pixel's 128-byte line leave back to back (stored thousands of cycles apart they reach HBM as two partial-line writes:
            // WRITE_SIZE 1.46x the output)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 r[NCO][4], y[NCO][4];
                int pix[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (oy0 + i < a.Ho) & (ox0 + j < a.Wo);
                    pix[i] = ok ? (oy0 + i) * a.Wo + ox0 + j : -1;
#pragma unroll
                    for (int cb = 0; cb < NCO; ++cb) r[cb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (has_res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int cb = 0; cb < NCO; ++cb)
                            r[cb][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, pix[i] >= 0 ? (pix[i] * a.res_cs + n0 + 16 * cb + 4 * h) * 4 : kOob, 0, 0));
                }
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb) at6(u[cb][0][j], u[cb][1][j], u[cb][2][j], u[cb][3][j], u[cb][4][j], u[cb][5][j], y[cb][0], y[cb][1], y[cb][2], y[cb][3]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int cb = 0; cb < NCO; ++cb) {
                        const f32x4 o = act4(y[cb][i] + b4[cb] + r[cb][i]);
#ifdef IDH_ABL_W4_NOSTORE
                        asm volatile("" ::"v"(o));
                        continue;
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, pix[i] >= 0 ? (pix[i] * a.out_cs + n0 + 16 * cb + 4 * h) * 4 : kOob, 0, 0);
                    }
            }
            W4T(129);
        }
#endif
        if (!has_next) break;
        t_cur = t_next;
        cur = nxt;
#ifdef IDH_ABL_W4_TRACE
        ++tile_i;
#endif
    }
}

// =====================================================================================================================================
// Second F(4x4) kernel: the input transform SHARED through LDS, two workgroups per CU (conv3x3_wino4s_k).
//
// conv3x3_wino4_k above needs 288 accumulators per wave = one wave per SIMD, and everything a second wave would cover — the epilogue
// (20 % of a 64-channel tile), copy waits, barriers — is exposed.  Here a wave owns ONE 16-channel block (36 positions x 4 = 144
// accumulator registers, all AGPRs; <= 112 VGPRs -> two waves per SIMD from two independent workgroups), and the 4 waves of a
// workgroup = the 4 output-channel blocks of a 64-channel tile share the transformed input of 16 tiles (64 x 4 pixels) through LDS:
// * per 8-channel stage every wave computes ONE 3x3 quadrant of the 6x6 transformed patch for all 16 tiles x 8 channels (lane (n, h):
//   tile n, channels 2h, 2h+1; 36 conflict-free ds_read_b64 of the raw halo; the partial transforms need 6 + 6 operations per column /
//   row instead of 12: 108 vector operations per wave and stage for 72 MFMAs = 1.5 per MFMA against 2 in conv3x3_wino4_k) and writes it
//   to V[k-step][lane][position] (positions quadrant-major: a reader's 36 values are 9 conflict-free ds_read_b128);
// * the MFMA B operands are read back from V by all four waves; the A operands (this wave's 16 output channels, 36 KiB per 64 channels
//   and stage in all) come straight from global memory / L2 as fragments, one 1 KiB row per 4 MFMAs, packed in read order;
// * the halo (10 x 34 texels) is copied through registers in PAIRS of stages (64 contiguous bytes per texel and load: 16 cache lines per wave
//   instruction instead of 32), all of it during the even stage, into three rotating 32-B-per-texel planes: halo(S) lives in plane S mod 3;
//   LDS per workgroup: 3 halo planes + 2 V buffers = 76.9 KiB (two workgroups per CU).
// Even stage S: loads halo(S + 2), halo(S + 3); MFMA(S): V(S) x panel(S); transform(S + 1): halo(S + 1) -> V(S + 1); one barrier.
constexpr int kSPlane = 432 * 32 + 128;            // 16 (row, column) phases x 3 x 9 texel slots x 32 B (+ 128: consecutive planes land on the other half of the banks)
constexpr int kSVBytes = 2 * 64 * 36 * 4;          // V of one stage: 2 k-steps x 64 lanes x 36 positions
constexpr int kSV0 = 3 * kSPlane, kSV1 = 3 * kSPlane + kSVBytes;
constexpr int kSLdsBytes = 3 * kSPlane + 2 * kSVBytes;  // 78720
constexpr int kSPanelFloats = 36 * 16 * 8;        // one stage's weights of one 16-channel block

// position order of V / the packed weights / the accumulators: quadrant-major, p' = 9 (2 a + b) + 3 (xi % 3) + (nu % 3) with xi = 3 a + .., nu = 3 b + ..
__host__ __device__ constexpr int w4s_xi(int pp) { return 3 * ((pp / 9) >> 1) + (pp % 9) / 3; }
__host__ __device__ constexpr int w4s_nu(int pp) { return 3 * ((pp / 9) & 1) + (pp % 9) % 3; }

// OIHW 3x3 -> U = G g G^T as A fragments: dst[stage c][co block cb (16)][ks 2][g 9][lane 64][e 4] = U[p' = 4g + e][co = 16 cb + (lane & 15)][ci = 8c + 2 (lane >> 4) + ks]
__global__ __launch_bounds__(256) void pack_wino4s_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout, int Cin, int nS, int nCB) {
    const long long total = (long long)nS * nCB * kSPanelFloats;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int e = (int)(t & 3), lane = (int)((t >> 2) & 63);
        long long r = t >> 8;
        const int g = (int)(r % 9); r /= 9;
        const int ks = (int)(r & 1); r >>= 1;
        const int cb = (int)(r % nCB), c = (int)(r / nCB);
        const int pp = 4 * g + e, co = 16 * cb + (lane & 15), ci = 8 * c + 2 * (lane >> 4) + ks;
        double u = 0.0;
        if (co < Cout && ci < Cin) {
            const float *gw = w + ((size_t)co * Cin + ci) * 9;
            const int xi = w4s_xi(pp), nu = w4s_nu(pp);
            const double G[6][3] = {{1.0, 0.0, 0.0}, {-8.0 / 15, -4.0 / 15, -2.0 / 15}, {-8.0 / 15, 4.0 / 15, -2.0 / 15},
                                    {1.0 / 30, 1.0 / 15, 2.0 / 15}, {1.0 / 30, -1.0 / 15, 2.0 / 15}, {0.0, 0.0, 1.0}};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) u += G[xi][a] * (double)gw[a * 3 + b] * G[nu][b];
        }
        dst[t] = (float)u;
    }
}

// half of the 1-D input transform B^T: outputs 0..2 (HI = false; they do not involve d5) or 3..5 (HI = true; no d0) of the 5 inputs
// x0..x4 = d0..d4 (d1..d5): 6 operations
template <bool HI>
__device__ __forceinline__ void bt3(float x0, float x1, float x2, float x3, float x4, float &o0, float &o1, float &o2) {
    if constexpr (!HI) {
        const float a = __builtin_fmaf(-4.f, x2, x4), b = __builtin_fmaf(-4.f, x1, x3);
        o0 = __builtin_fmaf(-4.25f, x2, x0) + x4;
        o1 = __builtin_fmaf(0.5f, b, a);
        o2 = __builtin_fmaf(-0.5f, b, a);
    } else {
        const float c = __builtin_fmaf(-0.25f, x1, x3), e = __builtin_fmaf(-0.25f, x0, x2);
        o0 = __builtin_fmaf(2.f, e, c);
        o1 = __builtin_fmaf(-2.f, e, c);
        o2 = __builtin_fmaf(-4.25f, x2, x0) + x4;
    }
}
// the same half transform fed ONE input at a time (k = 0..4), state (s0, s1, s2): the column pass of the streamed transform
template <bool HI, int K>
__device__ __forceinline__ void bt3_step(float w, float &s0, float &s1, float &s2) {
    if constexpr (!HI) {
        if constexpr (K == 0) s0 = w;
        if constexpr (K == 1) s1 = w;
        if constexpr (K == 2) { s0 = __builtin_fmaf(-4.25f, w, s0); s2 = w; }
        if constexpr (K == 3) s1 = __builtin_fmaf(-4.f, s1, w);                                // b = d3 - 4 d1
        if constexpr (K == 4) { s0 = s0 + w; s2 = __builtin_fmaf(-4.f, s2, w); }               // a = d4 - 4 d2
    } else {
        if constexpr (K == 0) { s0 = w; s2 = w; }
        if constexpr (K == 1) s1 = w;
        if constexpr (K == 2) { s2 = __builtin_fmaf(-4.25f, w, s2); s0 = __builtin_fmaf(-0.25f, s0, w); }  // e = d3 - d1 / 4
        if constexpr (K == 3) s1 = __builtin_fmaf(-0.25f, s1, w);                              // c = d4 - d2 / 4
        if constexpr (K == 4) s2 = s2 + w;
    }
}
template <bool HI>
__device__ __forceinline__ void bt3_finish(float s0, float s1, float s2, float &o0, float &o1, float &o2) {
    if constexpr (!HI) {
        o0 = s0; o1 = __builtin_fmaf(0.5f, s1, s2); o2 = __builtin_fmaf(-0.5f, s1, s2);
    } else {
        o0 = __builtin_fmaf(2.f, s0, s1); o1 = __builtin_fmaf(-2.f, s0, s1); o2 = s2;
    }
}

__global__ __launch_bounds__(256, 2) void conv3x3_wino4s_k(const Wino4Args wa) {
    __shared__ __attribute__((aligned(16))) char lds_raw[kSLdsBytes];
    lds_char *lds = (lds_char *)lds_raw;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(3))) volatile f32x2 lds_cf32x2;
    typedef __attribute__((address_space(3))) float lds_float;

    const ConvArgs &a = wa.c;
    const ConvSrc &s = a.s[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, h = lane >> 4;
    const int ty = n >> 3, tx = n & 7;  // the 16 tiles of a workgroup: 2 rows x 8 columns of 4x4 pixels = 32 x 8 pixels
    const int nS = s.cblocks * 2;       // stages of 8 input channels
    const int NT = a.NT;                // 64-channel tiles
    const int nCB = 4 * NT;             // 16-channel blocks of the packed weights

    // persistent workgroup over an XCD-contiguous range of tiles (channel tile fastest)
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
        const int txi = blk % wa.tiles_x; blk /= wa.tiles_x;
        const int tyi = blk % wa.tiles_y;
        r.img = blk / wa.tiles_y;
        r.y0 = tyi * 8; r.x0 = txi * 32;
        return r;
    };

    // ---- halo copies (global -> registers -> LDS), a pair of stages (16 channels = 64 B per texel) at a time.  Texel (row r 0..9, column col 0..33),
    // r = 4 R + rm, col = 4 cq + cm, has LDS index p = ((4 rm + cm) * 3 + R) * 9 + cq.  Thread t copies granule gr = t & 3 (4 lanes per texel: 16 cache
    // lines per wave load) of texel (row 2 k + (t >> 7), column (t >> 2) & 31) in copy k = 0..4 (the address advances by two image rows per copy), and
    // threads 0..79 copy columns 32, 33 of the ten rows in copy 5.  Rows above / below the image fall outside the buffer (zeros); columns outside it
    // are masked per lane.  Granules 0, 1 (the even stage's 8 channels) go to one plane, 2, 3 to the next; within a plane a texel has 32 B and its
    // channel quad q sits in granule q ^ (R & 1): the 16 tiles of a wave read, for a given patch element, a 2 x 8 block of (R, cq) whose 16-byte
    // granules fall into 16 different bank groups (conflict-free ds_read_b64).
    __amdgpu_buffer_rsrc_t rsH;
    int voffH0, voffH1;
    const int row_pair = s.W * s.cs * 8;  // bytes of two image rows
    auto set_halo_cursor = [&](const Tile &t) {
        rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)t.img * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));  // (re-derived per tile rather than spilled: see conv_wino.hip)
        const int gr = tid_o & 3;
        {
            const int iy = t.y0 - 1 + (tid_o >> 7), ix = t.x0 - 1 + ((tid_o >> 2) & 31);
            voffH0 = (unsigned)ix < (unsigned)s.W ? (iy * s.W + ix) * s.cs * 4 + 16 * gr : kOob;
        }
        {
            const int e = tid_o >> 2;
            const int iy = t.y0 - 1 + (e >> 1), ix = t.x0 + 31 + (e & 1);
            voffH1 = (e < 20) & (ix < s.W) & ((unsigned)iy < (unsigned)s.H) ? (iy * s.W + ix) * s.cs * 4 + 16 * gr : kOob;
        }
    };
    auto ld_halo = [&](int k, int ch) -> f32x4 {  // ch: the pair's first stage
#ifdef IDH_ABL_W4S_NODMA
        return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
        if (k < 5)  // (the row advance goes into the VECTOR offset: the scalar offset takes no part in the buffer range check)
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, (int)((unsigned)voffH0 + (unsigned)(k * row_pair)), __builtin_amdgcn_readfirstlane(32 * ch), 0));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, voffH1, __builtin_amdgcn_readfirstlane(32 * ch), 0));
    };
    // LDS address of copy k of this thread in the plane pair (offA: granules 0, 1; offB: granules 2, 3)
    int wdst0, wdst1;
    {
        const int wq = tid & 1, rr = tid >> 7, col = (tid >> 2) & 31;
        wdst0 = 32 * (108 * rr + 27 * (col & 3) + (col >> 2)) + 16 * wq;
        const int e = tid >> 2, r = e >> 1;
        wdst1 = 32 * (((4 * (r & 3) + (e & 1)) * 3 + (r >> 2)) * 9 + 8) + 16 * (wq ^ ((r >> 2) & 1));
    }
    const bool wselB = (tid >> 1) & 1;
    auto st_halo = [&](int k, f32x4 v, int offA, int offB) {
        const int off = wselB ? offB : offA;
        if (k < 5) *(lds_f32x4 *)(lds + ((off + wdst0) ^ (((k >> 1) & 1) << 4)) + 32 * (216 * (k & 1) + 9 * (k >> 1))) = v;  // row 2k + rr: rm = 2 (k & 1) + rr, R = k >> 1
        else if (tid < 80) *(lds_f32x4 *)(lds + off + wdst1) = v;
    };

    // ---- this lane's raw-patch read bases: element (i, c) of tile (ty, tx): texel p = ((4 (i & 3) + (c & 3)) * 3 + ty + (i >> 2)) * 9 + tx + (c >> 2); channels 2h, 2h+1
    int rbase[2];  // [i >> 2]
#pragma unroll
    for (int di = 0; di < 2; ++di) rbase[di] = 32 * (9 * ty + tx) + 16 * ((h >> 1) ^ ((ty + di) & 1)) + 8 * (h & 1);
    const int vbase = lane * 144;  // V[ks][lane][36]: 144 B per lane (36-dword stride: conflict-free ds_read_b128)
    const int qa = wave >> 1, qb = wave & 1;  // this wave's quadrant of positions: xi = 3 qa .., nu = 3 qb ..

    // transform of the stage whose halo is in the plane at `hoff` -> this wave's quadrant of V in `vbuf`.  The quadrant needs a 5 x 5 part of the
    // 6 x 6 patch only; it is streamed column by column (the next column's 5 reads in flight under the row pass of the current one).
    auto transform = [&](auto hic, auto hjc, int hoff, int vbuf) {
        constexpr bool HI_I = decltype(hic)::value, HI_J = decltype(hjc)::value;
        constexpr int I0 = HI_I ? 1 : 0, J0 = HI_J ? 1 : 0;
        const int rb[2] = {rbase[0] + hoff, rbase[1] + hoff};
        auto rd = [&](int i, int c) -> f32x2 { return *(lds_cf32x2 *)(lds + rb[i >> 2] + 32 * (((4 * (i & 3) + (c & 3)) * 3 + (i >> 2)) * 9 + (c >> 2))); };
        f32x2 d[2][5];
#pragma unroll
        for (int k = 0; k < 5; ++k) d[0][k] = rd(I0 + k, J0);
        float S[3][2][3];  // [row of the quadrant][channel][state]
        auto column = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if (K + 1 < 5) {
#pragma unroll
                for (int k = 0; k < 5; ++k) d[(K + 1) & 1][k] = rd(I0 + k, J0 + K + 1);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                float w[3];
                bt3<HI_I>(d[K & 1][0][ch], d[K & 1][1][ch], d[K & 1][2][ch], d[K & 1][3][ch], d[K & 1][4][ch], w[0], w[1], w[2]);
#pragma unroll
                for (int r = 0; r < 3; ++r) bt3_step<HI_J, K>(w[r], S[r][ch][0], S[r][ch][1], S[r][ch][2]);
            }
        };
        column(std::integral_constant<int, 0>{});
        column(std::integral_constant<int, 1>{});
        column(std::integral_constant<int, 2>{});
        column(std::integral_constant<int, 3>{});
        column(std::integral_constant<int, 4>{});
        const int q = 2 * (HI_I ? 1 : 0) + (HI_J ? 1 : 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float o0[3], o1[3];
            bt3_finish<HI_J>(S[r][0][0], S[r][0][1], S[r][0][2], o0[0], o0[1], o0[2]);
            bt3_finish<HI_J>(S[r][1][0], S[r][1][1], S[r][1][2], o1[0], o1[1], o1[2]);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                *(lds_float *)(lds + vbuf + vbase + 4 * (9 * q + 3 * r + cc)) = o0[cc];                  // k-step 0: channel 2h
                *(lds_float *)(lds + vbuf + 64 * 144 + vbase + 4 * (9 * q + 3 * r + cc)) = o1[cc];      // k-step 1: channel 2h + 1
            }
        }
    };
    auto transform_q = [&](int hoff, int vbuf) {  // (wave-uniform 4-way dispatch, once per stage)
#ifdef IDH_ABL_W4S_NOXFORM
        return;
#endif
        if (qa == 0 && qb == 0) transform(std::false_type{}, std::false_type{}, hoff, vbuf);
        else if (qa == 0) transform(std::false_type{}, std::true_type{}, hoff, vbuf);
        else if (qb == 0) transform(std::true_type{}, std::false_type{}, hoff, vbuf);
        else transform(std::true_type{}, std::true_type{}, hoff, vbuf);
    };

    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, nS * nCB * kSPanelFloats * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    const int voffA = lane * 16;
    // A fragments: a ring of kRing rows that runs across stage and tile boundaries: row j of a stage is consumed from Af[j % kRing] and the
    // register reloaded at once with the row kRing further on (24 MFMAs = ~0.8k cycles of lookahead)
    constexpr int kRing = 3;  // (divides 18)
    f32x4 Af[kRing];
    auto ldA = [&](int slot, int so) {
#ifdef IDH_ABL_W4S_NOA
        Af[slot] = (f32x4){1.f, 2.f, 3.f, 4.f};
        return;
#endif
        Af[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffA, so, 0));
    };

    // ---- prologue: halo(0), halo(1) of the first tile into planes 0, 1; V(0); the first kRing A rows
    Tile cur = decode(t_cur);
    set_halo_cursor(cur);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f32x4 t0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) t0[k] = ld_halo(3 * b + k, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) st_halo(3 * b + k, t0[k], 0, kSPlane);
    }
    {
        const int so0 = (4 * cur.nt + wave) * (kSPanelFloats * 4);
#pragma unroll
        for (int j = 0; j < kRing; ++j) ldA(j, __builtin_amdgcn_readfirstlane(so0 + 1024 * j));
    }
    __syncthreads();
    transform_q(0, kSV0);
    // even stage S: halo(S + 2) -> plane pl2, halo(S + 3) -> plane pl0 (halo(S) left it one barrier ago), in a batch of 2 and one of 4 copies; at its entry
    // stg[] holds the first batch in flight
    f32x4 stg[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) stg[k] = ld_halo(k, 2);
    __syncthreads();
    int pl0 = 0, pl1 = kSPlane, pl2 = 2 * kSPlane;  // LDS offsets of the planes of halo(S), halo(S + 1), halo(S + 2) (rotated every stage)

    // Developer build (-DIDH_ABL_W4S_TRACE, tools/abl_wino4.sh traces): every wave logs s_memtime along its SECOND tile into ConvArgs.ws (80 x 8 bytes
    // per wave: [0] tile start, [1 + 8 c + k] stage c < 8: k = 0 entry, 1 halo loads issued, 2/3/4 MFMA rows 0-5 / 6-11 / 12-17 issued, 5 halo written,
    // 6 transformed, 7 barrier passed; [70] epilogue start, [71] stored)
#ifdef IDH_ABL_W4S_TRACE
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(a.ws) + ((size_t)blockIdx.x * 4 + wave) * 80;
    int tile_i = 0;
#define W4ST(idx) do { if (tile_i == 1 && lane == 0 && (idx) >= 0) trace[(idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4ST(idx) do { } while (0)
#endif
    f32x4 acc[36];
#pragma unroll 1
    for (;;) {
        W4ST(0);
        const int t_next = t_cur + t_stride;
        const bool has_next = t_next < t_end;
        const Tile nxt = has_next ? decode(t_next) : cur;
#pragma unroll
        for (int p = 0; p < 36; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int cbw = 4 * cur.nt + wave;  // this wave's 16-channel block

        auto stage = [&](auto parc, const int c) {
            constexpr int PAR = decltype(parc)::value;
            constexpr int kVr = PAR ? kSV1 : kSV0, kVw = PAR ? kSV0 : kSV1;
            const int tr0 = c < 8 ? 1 + 8 * c : -100;
            W4ST(tr0);
            const int ch = c + 2 >= nS ? c + 2 - nS : c + 2;  // (even stages)
            W4ST(tr0 + 1);
            // MFMA(S): A rows from the ring, B fragments from V(S)
            const int aso = __builtin_amdgcn_readfirstlane((c * nCB + cbw) * (kSPanelFloats * 4));
            const bool last = c + 1 >= nS;
            const int aso_n = __builtin_amdgcn_readfirstlane(last ? (4 * nxt.nt + wave) * (kSPanelFloats * 4) : aso + nCB * (kSPanelFloats * 4));  // next stage (next tile: its stage 0)
            f32x4 Bf[18];
#ifdef IDH_ABL_W4S_NOB
            auto ldB = [&](int j) { Bf[j] = (f32x4){1.f, 2.f, 3.f, 4.f}; };
#else
            auto ldB = [&](int j) { Bf[j] = *(lds_cf32x4 *)(lds + kVr + (j / 9) * (64 * 144) + vbase + 16 * (j % 9)); };
#endif
            constexpr int kAheadB = 2;
#pragma unroll
            for (int j = 0; j < kAheadB; ++j) ldB(j);
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                if (j + kAheadB < 18) ldB(j + kAheadB);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * (j % 9) + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[j % kRing][e], Bf[j][e], acc[4 * (j % 9) + e], 0, 0, 0);
                ldA(j % kRing, j + kRing < 18 ? aso + 1024 * (j + kRing) : aso_n + 1024 * (j + kRing - 18));
                if (PAR == 0 && j == 8) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) st_halo(k, stg[k], pl2, pl0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) stg[k] = ld_halo(2 + k, ch);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef IDH_ABL_W4S_TRACE
                if (j % 6 == 5) W4ST(tr0 + 2 + j / 6);
#endif
            }
            if (PAR == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) st_halo(2 + k, stg[k], pl2, pl0);
            } else {
                // first half of the next even stage's copies (halo(E + 2), halo(E + 3), E = c + 1 or stage 0 of the next tile), issued HERE: loads complete in
                // order, so an A row issued after a copy cannot be used before the copy is back from HBM (~3k cycles); rows 0..5 of a stage use A rows
                // issued before this point
                if (c + 3 == nS) set_halo_cursor(nxt);
                const int chn = c + 3 >= nS ? c + 3 - nS : c + 3;
#pragma unroll
                for (int k = 0; k < 2; ++k) stg[k] = ld_halo(k, chn);
            }
            W4ST(tr0 + 5);
            // transform(S + 1): halo(S + 1) -> V(S + 1)
            transform_q(pl1, kVw);
            W4ST(tr0 + 6);
            __syncthreads();
            W4ST(tr0 + 7);
            const int t0 = pl0; pl0 = pl1; pl1 = pl2; pl2 = t0;
        };
#pragma unroll 1
        for (int c = 0; c < nS; c += 2) {
            stage(std::integral_constant<int, 0>{}, c);
            stage(std::integral_constant<int, 1>{}, c + 1);
        }

        // ---- epilogue: Y = A^T M A; lane = 4 consecutive channels of the 4x4 pixels of tile (ty, tx)
        W4ST(70);
#ifdef IDH_ABL_W4S_NOEPI
#pragma unroll
        for (int p = 0; p < 36; ++p) asm volatile("" ::"v"(acc[p]));
#else
        {
            const int n0 = 16 * cbw;
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.img * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)cur.img * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                                  a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
            const int oy0 = cur.y0 + 4 * ty, ox0 = cur.x0 + 4 * tx;
            const bool has_res = a.res != nullptr;
            const float slope_eff = a.act == IDH_ACT_LRELU ? a.slope : 1.f;
            const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (n0 + 4 * h) * 4, 0, 0));
            // M[xi][nu] = acc[p'(xi, nu)]
            auto M = [&](int xi, int nu) -> f32x4 & { return acc[9 * (2 * (xi / 3) + nu / 3) + 3 * (xi % 3) + nu % 3]; };
            f32x4 u[6][4];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) at6(M(xi, 0), M(xi, 1), M(xi, 2), M(xi, 3), M(xi, 4), M(xi, 5), u[xi][0], u[xi][1], u[xi][2], u[xi][3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 r[4], y[4];
                int pix[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (oy0 + i < a.Ho) & (ox0 + j < a.Wo);
                    pix[i] = ok ? (oy0 + i) * a.Wo + ox0 + j : -1;
                    r[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (has_res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, pix[i] >= 0 ? (pix[i] * a.res_cs + n0 + 4 * h) * 4 : kOob, 0, 0));
                }
                at6(u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j], y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 o = y[i] + b4 + r[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? o[e] * slope_eff : o[e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, pix[i] >= 0 ? (pix[i] * a.out_cs + n0 + 4 * h) * 4 : kOob, 0, 0);
                }
            }
        }
#endif
        W4ST(71);
        if (!has_next) break;
        t_cur = t_next;
        cur = nxt;
#ifdef IDH_ABL_W4S_TRACE
        ++tile_i;
#endif
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
    return !a.s[1].in && (a.act == IDH_ACT_NONE || a.act == IDH_ACT_LRELU) && s.cblocks >= 2 && s.ks == 3 && s.stride == 1 && s.pad_mode == IDH_PAD_ZEROS && !s.up_in[0] && !s.norm && a.S == 1 && (a.Cout % 32) == 0 &&
           (long long)s.H * s.W * s.cs * 4 < (1ll << 31) && (long long)a.Ho * a.Wo * a.out_cs * 4 < (1ll << 31) &&
           (!a.res || (long long)a.Ho * a.Wo * a.res_cs * 4 < (1ll << 31)) && (long long)s.cblocks * a.Cout_pad * 36 * 16 * 4 < (1ll << 31);
}

bool wino4s_supported(const ConvArgs &a) {
    return wino4_supported(a) && (a.Cout % 64) == 0 && (long long)a.s[0].cblocks * a.Cout_pad * 36 * 16 * 4 < (1ll << 31);
}

// shared-transform variant: 32 x 8 pixel x 64 channel tiles, two persistent workgroups per CU
int launch_conv_wino4s(const ConvArgs &a, int N, hipStream_t st) {
    if (!wino4s_supported(a)) return IDH_EUNSUPPORTED;
    Wino4Args wa{a, (a.Wo + 31) / 32, (a.Ho + 7) / 8, 0};
    wa.c.NT = a.Cout / 64;
    const long long tiles = (long long)N * wa.tiles_x * wa.tiles_y * wa.c.NT;
    if (tiles >= (1ll << 31)) return IDH_EUNSUPPORTED;
    wa.tiles = (int)tiles;
    long long grid = 2ll * wino4_cus();
    if (grid > wa.tiles) grid = wa.tiles >= 8 ? wa.tiles / 8 * 8 : wa.tiles;
    hipLaunchKernelGGL(conv3x3_wino4s_k, dim3((unsigned)grid), dim3(256), 0, st, wa);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
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

extern "C" size_t idh_packed_wino4s_weight_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)((Cin + 15) & ~15) * ((Cout + 15) & ~15) * 36;
}

extern "C" int idh_pack_conv_weight_wino4s(const float *w, float *dst, int Cout, int Cin, void *stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0) return IDH_EINVAL;
    const int nS = ((Cin + 15) / 16) * 2, nCB = (Cout + 15) / 16;
    const long long total = (long long)nS * nCB * kSPanelFloats;
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_wino4s_weight_k, dim3(grid), dim3(256), 0, idh_stream(stream), w, dst, Cout, Cin, nS, nCB);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

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
