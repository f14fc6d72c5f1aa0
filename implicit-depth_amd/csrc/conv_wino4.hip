// 3x3 stride-1 convolution as Winograd F(4x4, 3x3) on the fp32 matrix cores of gfx950 (conv3x3_wino4_k).
//
// Replaces the same nn.Conv2d calls as conv3x3_wino_k / conv3x3_lds_k (BasicBlock convs, reference modules/layers.py:59-95,
// the 3x3 stride-1 layers of CVEncoder / BDDecoderPP / DepthDecoderPP, modules/networks.py:20-215) wherever the layer is a plain
// conv (+ bias, LeakyReLU, and either a residual or BasicBlock's fused 1x1 projection of a second tensor, layers.py:86-92) with a multiple of
// 64 output channels and enough tiles to fill the chip.  F(4x4,3x3) needs
// 36 multiplies per 4x4 output tile and channel pair instead of 64 for four F(2x2) tiles (144 direct): 1.78x fewer
// v_mfma_f32_16x16x4_f32 than conv3x3_wino_k, fp32 operands and fp32 accumulation throughout.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 6x6 input patch, g: 3x3 filter, Y: 4x4 outputs
//
// Interpolation points {0, +-1/2, +-2, inf} (not the textbook {0, +-1, +-2}): same operation count, about half the fp32
// error (2-5e-6 of the output scale against fp64 at 64-384 input channels; F(2x2): 3-5e-7, direct kernel: 1e-6):
//   B^T = [1 0 -17/4 0 1 0; 0 -2 -4 1/2 1 0; 0 2 -4 -1/2 1 0; 0 -1/2 -1/4 2 1 0; 0 1/2 -1/4 -2 1 0; 0 1 0 -17/4 0 1]
//   G   = [1 0 0; -8/15 -4/15 -2/15; -8/15 4/15 -2/15; 1/30 1/15 2/15; 1/30 -1/15 2/15; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1/2 -1/2 2 -2 0; 0 1/4 1/4 4 4 0; 0 1/8 -1/8 8 -8 1]
//
// The 36 element-wise products over the input channels are 36 independent GEMMs M[pos][co, tile] = sum_ci U[pos][co, ci] V[pos][ci, tile],
// issued as D^T = U . V^T (weights = MFMA A operand, transformed input = B operand).
//
// (The round's first design kept the transformed input in registers - one wave per SIMD with 288 accumulators, every wave transforming its
// own tiles for two 16-channel blocks - and reached 1.03-1.19x over F(2x2); its timelines and ablations are in profiles/r04/experiments.md.
// This kernel superseded it on every layer and that code is gone.)
#include <cstdlib>
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

struct Wino4Args {
    ConvArgs c;
    int tiles_x, tiles_y;
    int tiles;  // N * tiles_y * tiles_x * NT
};

__device__ __forceinline__ f32x4 fma4(float s, f32x4 x, f32x4 y) {
    return (f32x4){__builtin_fmaf(s, x[0], y[0]), __builtin_fmaf(s, x[1], y[1]), __builtin_fmaf(s, x[2], y[2]), __builtin_fmaf(s, x[3], y[3])};
}
// the lane id read afresh (two v_mbcnt): per-tile lane constants (halo offsets, the epilogue's pixel / channel offsets) are derived from THIS instead of
// from values computed once at kernel entry, which hipcc keeps live across the stage loop - at 256 VGPRs that means scratch, and a scratch reload in
// front of a batch of loads waits, through the in-order vmcnt, for every global load in flight (profiles/r04/experiments.md section 4)
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// 1-D output transform A^T (6 -> 4): 12 operations per component, written component by component: vector-typed arithmetic becomes v_pk_*_f32 on 64-bit
// register pairs, whose alignment constraints fragment the register file at the epilogue's peak (scratch) and which cost ~3x a scalar operation
// beside the other wave's fp32 MFMAs (profiles/r04/experiments.md section 4)
__device__ __forceinline__ void at6(f32x4 m0, f32x4 m1, f32x4 m2, f32x4 m3, f32x4 m4, f32x4 m5, f32x4 &y0, f32x4 &y1, f32x4 &y2, f32x4 &y3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float s1 = m1[e] + m2[e], d1 = m1[e] - m2[e], s2 = m3[e] + m4[e], d2 = m3[e] - m4[e];
        y0[e] = (m0[e] + s1) + s2;
        y1[e] = __builtin_fmaf(0.5f, d1, d2 * 2.f);
        y2[e] = __builtin_fmaf(0.25f, s1, s2 * 4.f);
        y3[e] = __builtin_fmaf(0.125f, d1, __builtin_fmaf(8.f, d2, m5[e]));
    }
}

// Design (two workgroups per CU, two waves per SIMD):
// * A wave owns ONE 16-channel block of a tile group of 16 tiles (2 rows x 8 columns of 4x4 pixels = 32 x 8 pixels): 36 positions x 4 =
//   144 accumulator registers.  The 4 waves of a workgroup are the 4 channel blocks of a 64-channel tile and SHARE the transformed input of
//   the tile group through LDS: per 8-channel stage every wave computes ONE 3x3 quadrant of the 6x6 transformed patch for all 16 tiles x 8
//   channels (lane (n, h): tile n, channels 2h, 2h+1).  A quadrant needs a 5x5 part of the patch only: 25 conflict-free ds_read_b64 of the
//   raw halo, streamed column by column (the next column's reads in flight under the current one's row pass), 6 + 6 operations per column /
//   row = 108 vector operations per wave and stage for 72 MFMAs.  The quadrant goes to V[k-step][lane][position] (positions quadrant-major:
//   a reader's 36 values are 9 conflict-free ds_read_b128), which all four waves read back as MFMA B operands.
// * The A operands (this wave's 16 output channels: 18 KiB per stage) come straight from global memory / L2 as fragments, one 1 KiB row per
//   4 MFMAs, packed in read order, through a ring of registers that runs across stage and tile boundaries.
// * The halo (10 x 34 texels) is copied through registers in PAIRS of stages (64 contiguous bytes per texel and load: 16 cache lines per wave
//   instruction instead of 32) into three rotating 32-B-per-texel planes: halo(S) lives in plane S mod 3.  Out-of-image texels are out of the
//   buffer descriptor's range (rows) or masked (columns) and arrive as zeros = zero padding.  LDS per workgroup: 3 halo planes + 2 V
//   buffers = 76.9 KiB.
// * Even stage S: copies of halo(S + 2), halo(S + 3) land; MFMA(S): V(S) x panel(S); transform(S + 1): halo(S + 1) -> V(S + 1); one barrier
//   per stage.  While one wave of a SIMD transforms, waits or runs its epilogue, the other workgroup's wave has the matrix pipe.
// * Vector loads complete in order, and the halo comes from HBM (~3k cycles under load) while the A rows come from L2: an A row issued
//   after a halo copy cannot be used before that copy is back.  So the first copies of a pair are issued at the END of the odd stage's MFMA
//   loop (rows 0.. of the next stage use A rows issued before that point).
// * Epilogue: output transform (120 vector ops per channel quad) in registers, + bias + residual, LeakyReLU / ELU / identity, 16-byte NHWC stores
//   (a lane holds 4 consecutive channels of the 4x4 pixels of its tile).  conv3x3_wino4_k<true>: between the output transform and the stores the 1x1
//   projection of a second tensor is accumulated by pixel-domain MFMAs (the "P phase", described where it is written).
// Measured (B = 32, profiles/r04/perf_wino4_final.txt): 1.11-1.39x over conv3x3_wino_k on the network's plain layers, 1.07-1.25x on the blocks with a
// fused projection.
constexpr int kPlane = 432 * 32 + 128;            // 16 (row, column) phases x 3 x 9 texel slots x 32 B (+ 128: consecutive planes land on the other half of the banks)
constexpr int kVBytes = 2 * 64 * 36 * 4;          // V of one stage: 2 k-steps x 64 lanes x 36 positions
constexpr int kV0 = 3 * kPlane, kV1 = 3 * kPlane + kVBytes;
constexpr int kLdsBytes = 3 * kPlane + 2 * kVBytes;  // 78720
constexpr int kPanelFloats = 36 * 16 * 8;        // one stage's weights of one 16-channel block

// position order of V / the packed weights / the accumulators: quadrant-major, p' = 9 (2 a + b) + 3 (xi % 3) + (nu % 3) with xi = 3 a + .., nu = 3 b + ..
__host__ __device__ constexpr int w4_xi(int pp) { return 3 * ((pp / 9) >> 1) + (pp % 9) / 3; }
__host__ __device__ constexpr int w4_nu(int pp) { return 3 * ((pp / 9) & 1) + (pp % 9) % 3; }
// Slot order of a lane's 36 values in V (= the order of the packed weight rows and of the accumulators).  Round 6: a quadrant's nine values are the
// aligned 32 bytes [8 q, 8 q + 8) plus slot 32 + q, so the wave that computed the quadrant stores it with two ds_write_b128 + one ds_write_b32 per k-step
// (conflict-free: eight lanes x 16 bytes at a 144-byte pitch cover the 32 banks once) instead of nine ds_write_b32 that hit each bank four times - the V
// writes were ~10 % of the K loop (tools/micro/wino4_mfma_shape.hip).  -DIDH_W4_NO_VPACK: the quadrant-major order of rounds 4-5 (slot = position).
#ifdef IDH_W4_NO_VPACK
__host__ __device__ constexpr int w4_v2p(int v) { return v; }
__host__ __device__ constexpr int w4_p2v(int pp) { return pp; }
#else
__host__ __device__ constexpr int w4_v2p(int v) { return v < 32 ? 9 * (v / 8) + v % 8 : 9 * (v - 32) + 8; }
__host__ __device__ constexpr int w4_p2v(int pp) { return pp % 9 < 8 ? 8 * (pp / 9) + pp % 9 : 32 + pp / 9; }
#endif

// OIHW 3x3 -> U = G g G^T as A fragments: dst[stage c][co block cb (16)][ks 2][g 9][lane 64][e 4] = U[p' = 4g + e][co = 16 cb + (lane & 15)][ci = 8c + 2 (lane >> 4) + ks]
__global__ __launch_bounds__(256) void pack_wino4_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout, int Cin, int nS, int nCB) {
    const long long total = (long long)nS * nCB * kPanelFloats;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int e = (int)(t & 3), lane = (int)((t >> 2) & 63);
        long long r = t >> 8;
        const int g = (int)(r % 9); r /= 9;
        const int ks = (int)(r & 1); r >>= 1;
        const int cb = (int)(r % nCB), c = (int)(r / nCB);
        const int pp = w4_v2p(4 * g + e), co = 16 * cb + (lane & 15), ci = 8 * c + 2 * (lane >> 4) + ks;  // (row g, element e = V slot 4 g + e)
        double u = 0.0;
        if (co < Cout && ci < Cin) {
            const float *gw = w + ((size_t)co * Cin + ci) * 9;
            const int xi = w4_xi(pp), nu = w4_nu(pp);
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

// SRC2: BasicBlock's 1x1 projection of a second tensor (layers.py:86-92), accumulated in the PIXEL domain after the output transform (the epilogue's
// "P phase" below).  RES: a residual tensor is added in the epilogue (BasicBlock's identity shortcut; never together with SRC2).
template <bool SRC2, bool RES>
__global__ __launch_bounds__(256, 2) void conv3x3_wino4_k(const Wino4Args wa) {
    __shared__ __attribute__((aligned(16))) char lds_raw[kLdsBytes];
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
    // The cursor (image descriptor + tile origin) lives in scalar registers; the per-lane source offsets are derived from the lane id read afresh
    // at every batch of copies (halo_src0 / halo_src1, ~10 integer operations) instead of being held in vector registers from tile to tile: whatever
    // lane-invariant value is live across the epilogue's register peak hipcc parks in scratch, and its reload then sits in the stage loop right
    // behind the copies it addresses - where the in-order vmcnt makes it wait for them (see fresh_lane).
    __amdgpu_buffer_rsrc_t rsH;
    int cy0 = 0, cx0 = 0;
    const int row_pair = s.W * s.cs * 8;  // bytes of two image rows
    auto set_halo_cursor = [&](const Tile &t) {
        rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)t.img * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
        cy0 = t.y0; cx0 = t.x0;
    };
    auto halo_src0 = [&]() -> int {  // copies 0..4: texel (row 2 k + (t >> 7), column (t >> 2) & 31), granule t & 3
        const int t = 64 * wave + fresh_lane();
        const int iy = cy0 - 1 + (t >> 7), ix = cx0 - 1 + ((t >> 2) & 31);
        return (unsigned)ix < (unsigned)s.W ? (iy * s.W + ix) * s.cs * 4 + 16 * (t & 3) : kOob;
    };
    auto halo_src1 = [&]() -> int {  // copy 5: columns 32, 33 of the ten rows (threads 0..79)
        const int t = 64 * wave + fresh_lane();
        const int e = t >> 2;
        const int iy = cy0 - 1 + (e >> 1), ix = cx0 + 31 + (e & 1);
        return ((e < 20) & (ix < s.W) & ((unsigned)iy < (unsigned)s.H)) ? (iy * s.W + ix) * s.cs * 4 + 16 * (t & 3) : kOob;
    };
    auto ld_halo = [&](int k, int ch, int src0) -> f32x4 {  // ch: the pair's first stage; src0 = halo_src0() of this batch
#ifdef IDH_ABL_W4_NOHALO
        return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
        if (k < 5)  // (the row advance goes into the VECTOR offset: the scalar offset takes no part in the buffer range check)
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, (int)((unsigned)src0 + (unsigned)(k * row_pair)), __builtin_amdgcn_readfirstlane(32 * ch), 0));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, halo_src1(), __builtin_amdgcn_readfirstlane(32 * ch), 0));
    };
    // LDS address of copy k of this thread in the plane pair (offA: granules 0, 1; offB: granules 2, 3)
    // Per-lane constants of the stage loop.  They are (re)derived at the start of EVERY tile from the lane id read afresh (derive_lane_constants, ~25
    // integer operations per tile): nothing lane-invariant is then live across the epilogue, whose register peak would otherwise push them into
    // scratch - with the reload inside the stage loop, in front of the halo loads (see fresh_lane).
    int vbase, voffA;
    auto derive_lane_constants = [&]() {
        const int ln = fresh_lane();
        vbase = ln * 144;  // V[ks][lane][36]: 144 B per lane (36-dword stride: conflict-free ds_read_b128)
        voffA = ln * 16;
    };
    derive_lane_constants();
    // LDS address of this thread's copies in the plane pair (offA: granules 0, 1; offB: granules 2, 3), derived per batch of stores like the sources
    auto halo_dst = [&](int offA, int offB) -> int {
        const int t = 64 * wave + fresh_lane();
        const int wq = t & 1, rr = t >> 7, col = (t >> 2) & 31;
        return ((t >> 1) & 1 ? offB : offA) + 32 * (108 * rr + 27 * (col & 3) + (col >> 2)) + 16 * wq;
    };
    auto st_halo = [&](int k, f32x4 v, int dst0, int offA, int offB) {
        if (k < 5) *(lds_f32x4 *)(lds + (dst0 ^ (((k >> 1) & 1) << 4)) + 32 * (216 * (k & 1) + 9 * (k >> 1))) = v;  // row 2k + rr: rm = 2 (k & 1) + rr, R = k >> 1
        else if (wave < 2) {  // (threads 0..79)
            const int t = 64 * wave + fresh_lane();
            if (t < 80) {
                const int e = t >> 2, r = e >> 1;
                *(lds_f32x4 *)(lds + ((t >> 1) & 1 ? offB : offA) + 32 * (((4 * (r & 3) + (e & 1)) * 3 + (r >> 2)) * 9 + 8) + 16 * ((t & 1) ^ ((r >> 2) & 1))) = v;
            }
        }
    };

    // ---- this lane's raw-patch read bases: element (i, c) of tile (ty, tx): texel p = ((4 (i & 3) + (c & 3)) * 3 + ty + (i >> 2)) * 9 + tx + (c >> 2); channels 2h, 2h+1
    const int qa = wave >> 1, qb = wave & 1;  // this wave's quadrant of positions: xi = 3 qa .., nu = 3 qb ..

    // transform of the stage whose halo is in the plane at `hoff` -> this wave's quadrant of V in `vbuf`.  The quadrant needs a 5 x 5 part of the
    // 6 x 6 patch only; it is streamed column by column (the next column's 5 reads in flight under the row pass of the current one).
    auto transform = [&](auto hic, auto hjc, int hoff, int vbuf) {
        constexpr bool HI_I = decltype(hic)::value, HI_J = decltype(hjc)::value;
        constexpr int I0 = HI_I ? 1 : 0, J0 = HI_J ? 1 : 0;
        // (patch read base derived on the spot, once per stage: one vector register less held through the MFMA loop)
        const int ln = fresh_lane(), n_ = ln & 15, h_ = ln >> 4, ty_ = n_ >> 3, tx_ = n_ & 7;
        const int rbase = 32 * (9 * ty_ + tx_) + 16 * ((h_ >> 1) ^ (ty_ & 1)) + 8 * (h_ & 1);  // rows i < 4; rows 4, 5 (R + 1): the granule bit flips (^ 16)
        const int rb[2] = {rbase + hoff, (rbase ^ 16) + hoff};
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
#ifdef IDH_W4_NO_VPACK
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
#else
        // the quadrant's nine values per channel (index 3 r + cc) -> slots 8 q .. 8 q + 7 and 32 + q (w4_p2v): 2 x ds_write_b128 + ds_write_b32 per k-step
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {  // k-step ch: channel 2h + ch
            float o[9];
#pragma unroll
            for (int r = 0; r < 3; ++r) bt3_finish<HI_J>(S[r][ch][0], S[r][ch][1], S[r][ch][2], o[3 * r], o[3 * r + 1], o[3 * r + 2]);
            lds_char *dst = lds + vbuf + ch * (64 * 144) + vbase;
            *(lds_f32x4 *)(dst + 32 * q) = (f32x4){o[0], o[1], o[2], o[3]};
            *(lds_f32x4 *)(dst + 32 * q + 16) = (f32x4){o[4], o[5], o[6], o[7]};
            *(lds_float *)(dst + 128 + 4 * q) = o[8];
        }
#endif
    };
    auto transform_q = [&](int hoff, int vbuf) {  // (wave-uniform 4-way dispatch, once per stage)
#ifdef IDH_ABL_W4_NOXFORM
        return;
#endif
        if (qa == 0 && qb == 0) transform(std::false_type{}, std::false_type{}, hoff, vbuf);
        else if (qa == 0) transform(std::false_type{}, std::true_type{}, hoff, vbuf);
        else if (qb == 0) transform(std::true_type{}, std::false_type{}, hoff, vbuf);
        else transform(std::true_type{}, std::true_type{}, hoff, vbuf);
    };

    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, nS * nCB * kPanelFloats * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    // A fragments: a ring of kRing rows that runs across stage and tile boundaries: row j of a stage is consumed from Af[j % kRing] and the
    // register reloaded at once with the row kRing further on (12 MFMAs of lookahead)
#ifndef IDH_W4_RING
#define IDH_W4_RING 3
#endif
#ifndef IDH_W4_PRIO  // wave priority by phase (bit 0: input transform, bit 1: epilogue, bit 2: MFMA loop run at priority 1; else 0)
#define IDH_W4_PRIO 0
#endif
#define W4_PRIO(bit) __builtin_amdgcn_s_setprio((IDH_W4_PRIO >> (bit)) & 1)
    constexpr int kRing = IDH_W4_RING;  // (divides 18; measured 1: -7 ... -39 %, 2: 0 ... -12 %, 6: -6 ... -9 %: more rows in flight cost more at the issue of the other loads than they hide)
    f32x4 Af[kRing];
    auto ldA = [&](int slot, int so) {
#ifdef IDH_ABL_W4_NOA
        Af[slot] = (f32x4){1.f, 2.f, 3.f, 4.f};
        return;
#endif
#ifdef IDH_ABL_W4_HALFA  // (timing experiment: every second A row is not fetched - what would sharing each row between two waves be worth at most?)
        if ((so >> 10) & 1) return;
#endif
        Af[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffA, so, 0));
    };

    // ---- prologue: halo(0), halo(1) of the first tile into planes 0, 1; V(0); the first kRing A rows
    Tile cur = decode(t_cur);
    set_halo_cursor(cur);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f32x4 t0[3];
        const int hs = halo_src0(), hd = halo_dst(0, kPlane);
#pragma unroll
        for (int k = 0; k < 3; ++k) t0[k] = ld_halo(3 * b + k, 0, hs);
#pragma unroll
        for (int k = 0; k < 3; ++k) st_halo(3 * b + k, t0[k], hd, 0, kPlane);
    }
    {
#ifdef IDH_ABL_W4_SAMEA  // (timing experiment: every wave reads channel block 0's fragments: what would sharing the A rows in L1 be worth?)
#define W4_CB(x) 0
#else
#define W4_CB(x) (x)
#endif
        const int so0 = W4_CB(4 * cur.nt + wave) * (kPanelFloats * 4);
#pragma unroll
        for (int j = 0; j < kRing; ++j) ldA(j, __builtin_amdgcn_readfirstlane(so0 + 1024 * j));
    }
    __syncthreads();
    transform_q(0, kV0);
    // even stage S: halo(S + 2) -> plane pl2, halo(S + 3) -> plane pl0 (halo(S) left it one barrier ago), in a batch of 2 and one of 4 copies; at its entry
    // stg[] holds the first batch in flight
    f32x4 stg[4];
    {
        const int hs = halo_src0();
#pragma unroll
        for (int k = 0; k < 2; ++k) stg[k] = ld_halo(k, 2, hs);
    }
    __syncthreads();
    int pl0 = 0, pl1 = kPlane, pl2 = 2 * kPlane;  // LDS offsets of the planes of halo(S), halo(S + 1), halo(S + 2) (rotated every stage)

    // Developer build (-DIDH_ABL_W4_TRACE, tools/abl_wino4.sh trace): every wave logs s_memtime along its SECOND tile into ConvArgs.ws (80 x 8 bytes
    // per wave: [0] tile start, [1 + 8 c + k] stage c < 8: k = 0 entry, 1 halo loads issued, 2/3/4 MFMA rows 0-5 / 6-11 / 12-17 issued, 5 halo written,
    // 6 transformed, 7 barrier passed; [70] epilogue start, [71] stored)
#ifdef IDH_ABL_W4_TRACE
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(a.ws) + ((size_t)blockIdx.x * 4 + wave) * 80;
    int tile_i = 0;
#define W4T(idx) do { if (tile_i == 1 && lane == 0 && (idx) >= 0) trace[(idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4T(idx) do { } while (0)
#endif
    f32x4 acc[36];
#pragma unroll 1
    for (;;) {
        W4T(0);
        const int t_next = t_cur + t_stride;
        const bool has_next = t_next < t_end;
        const Tile nxt = has_next ? decode(t_next) : cur;
#pragma unroll
        for (int p = 0; p < 36; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int cbw = 4 * cur.nt + wave;  // this wave's 16-channel block

        auto stage = [&](auto parc, const int c) {
            constexpr int PAR = decltype(parc)::value;
            constexpr int kVr = PAR ? kV1 : kV0, kVw = PAR ? kV0 : kV1;
            [[maybe_unused]] const int tr0 = c < 8 ? 1 + 8 * c : -100;
            W4T(tr0);
            const int ch = c + 2 >= nS ? c + 2 - nS : c + 2;  // (even stages)
            W4T(tr0 + 1);
            // MFMA(S): A rows from the ring, B fragments from V(S)
            const int aso = __builtin_amdgcn_readfirstlane((c * nCB + W4_CB(cbw)) * (kPanelFloats * 4));
            const bool last = c + 1 >= nS;
            const int aso_n = __builtin_amdgcn_readfirstlane(last ? W4_CB(4 * nxt.nt + wave) * (kPanelFloats * 4) : aso + nCB * (kPanelFloats * 4));  // next stage (next tile: its stage 0)
            f32x4 Bf[18];
#ifdef IDH_ABL_W4_NOB
            auto ldB = [&](int j) { Bf[j] = (f32x4){1.f, 2.f, 3.f, 4.f}; };
#else
            auto ldB = [&](int j) { Bf[j] = *(lds_cf32x4 *)(lds + kVr + (j / 9) * (64 * 144) + vbase + 16 * (j % 9)); };
#endif
            constexpr int kAheadB = 2;
            if (IDH_W4_PRIO) W4_PRIO(2);
#pragma unroll
            for (int j = 0; j < kAheadB; ++j) ldB(j);
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                if (j + kAheadB < 18) ldB(j + kAheadB);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * (j % 9) + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[j % kRing][e], Bf[j][e], acc[4 * (j % 9) + e], 0, 0, 0);
                ldA(j % kRing, j + kRing < 18 ? aso + 1024 * (j + kRing) : aso_n + 1024 * (j + kRing - 18));
                if (PAR == 0 && j == 8) {
                    const int hd = halo_dst(pl2, pl0), hs = halo_src0();
#pragma unroll
                    for (int k = 0; k < 2; ++k) st_halo(k, stg[k], hd, pl2, pl0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) stg[k] = ld_halo(2 + k, ch, hs);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef IDH_ABL_W4_TRACE
                if (j % 6 == 5) W4T(tr0 + 2 + j / 6);
#endif
            }
            if (PAR == 0) {
                const int hd = halo_dst(pl2, pl0);
#pragma unroll
                for (int k = 0; k < 4; ++k) st_halo(2 + k, stg[k], hd, pl2, pl0);
            } else {
                // first half of the next even stage's copies (halo(E + 2), halo(E + 3), E = c + 1 or stage 0 of the next tile), issued HERE: loads complete in
                // order, so an A row issued after a copy cannot be used before the copy is back from HBM (~3k cycles); rows 0..5 of a stage use A rows
                // issued before this point
                if (c + 3 == nS) set_halo_cursor(nxt);
                const int chn = c + 3 >= nS ? c + 3 - nS : c + 3;
                const int hs = halo_src0();
#pragma unroll
                for (int k = 0; k < 2; ++k) stg[k] = ld_halo(k, chn, hs);
            }
            W4T(tr0 + 5);
            // transform(S + 1): halo(S + 1) -> V(S + 1)
            if (IDH_W4_PRIO) W4_PRIO(0);
            transform_q(pl1, kVw);
            W4T(tr0 + 6);
            __syncthreads();
            W4T(tr0 + 7);
            const int t0 = pl0; pl0 = pl1; pl1 = pl2; pl2 = t0;
        };
#pragma unroll 1
        for (int c = 0; c < nS; c += 2) {
            stage(std::integral_constant<int, 0>{}, c);
            stage(std::integral_constant<int, 1>{}, c + 1);
        }

        // ---- epilogue: Y = A^T M A; lane = 4 consecutive channels of the 4x4 pixels of tile (ty, tx)
        W4T(70);
        if (IDH_W4_PRIO) W4_PRIO(1);
#ifdef IDH_ABL_W4_NOEPI
#pragma unroll
        for (int p = 0; p < 36; ++p) asm volatile("" ::"v"(acc[p]));
#else
        {
            const int lane_e = fresh_lane(), tid_e = 64 * wave + lane_e;
            const int n = lane_e & 15, h = lane_e >> 4, ty = n >> 3, tx = n & 7;  // (shadow the kernel-entry values: see fresh_lane)
            (void)tid_e;
            const int n0 = 16 * cbw;
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.img * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)cur.img * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                                  a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
            const int oy0 = cur.y0 + 4 * ty, ox0 = cur.x0 + 4 * tx;
            const float slope_eff = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(a.act == IDH_ACT_LRELU ? __builtin_bit_cast(int, a.slope) : 0x3f800000));  // (scalar register)
            // nn.ELU(alpha = 1) as torch's kernel forms it, exp(x) - 1, with the exponential through v_exp_f32 (as csrc/mlp.hip: |err| ~1e-7 absolute)
            const bool elu = a.act == IDH_ACT_ELU;
            const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (n0 + 4 * h) * 4, 0, 0));
            // M[xi][nu] = acc[p'(xi, nu)]
            auto M = [&](int xi, int nu) -> f32x4 & { return acc[w4_p2v(9 * (2 * (xi / 3) + nu / 3) + 3 * (xi % 3) + nu % 3)]; };
            // pixel (i, j) of this lane's tile: element offset of its channel-0 value in an image of channel stride 1, or -1 outside the map
            auto pixel = [&](int i, int j) -> int { return ((oy0 + i < a.Ho) & (ox0 + j < a.Wo)) ? (oy0 + i) * a.Wo + ox0 + j : -1; };
            // residual loads run ahead of their use: column 0's are issued before the row pass (while the 144 accumulators are still live there is room
            // for one column), columns 1, 2's right after it and column 3's after column 0's pass (a ring of three) (issued one column at a time just before use, every column exposed a full memory round
            // trip: the NOEPI ablation put 27-33 % of a 64-channel tile with residual into the epilogue)
            f32x4 r[3][4];  // (ring: column j in slot j % 3)
            auto ld_res = [&](int j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#ifdef IDH_ABL_W4_EPILIN  // (timing experiment: residual loads / stores of a wave cover 8 FULL cache lines per instruction instead of 16 half lines: bound for an LDS-transposed epilogue)
                    r[j % 3][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, ((cur.y0 / 8 * wa.tiles_x + cur.x0 / 32) * 64 + (4 * i + j) * 4 + wave) * 1024 + lane_e * 16, 0, 0));
#else
                    const int px = pixel(i, j);
                    r[j % 3][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, px >= 0 ? (px * a.res_cs + n0 + 4 * h) * 4 : kOob, 0, 0));
#endif
                }
            };
            if constexpr (RES) ld_res(0);
            f32x4 u[6][4];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                at6(M(xi, 0), M(xi, 1), M(xi, 2), M(xi, 3), M(xi, 4), M(xi, 5), u[xi][0], u[xi][1], u[xi][2], u[xi][3]);
                // the row's 16 results are pinned HERE: left alone, LLVM sinks the outputs 1..3 of every row into the column passes that use them and keeps the
                // rows' sums / differences (20 values per row instead of 12) alive across the whole pass - the epilogue's register peak, i.e. scratch
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = u[xi][j][e];
                        asm volatile("" : "+v"(t));
                        u[xi][j][e] = t;
                    }
            }
            if constexpr (!SRC2) {
                if constexpr (RES) {
                    __builtin_amdgcn_sched_barrier(0);  // (the loads stay BELOW the row pass: hoisted above it their 48 registers do not fit beside the accumulators)
                    ld_res(1); ld_res(2);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 y[4];
                    at6(u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j], y[0], y[1], y[2], y[3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = y[i][e] + b4[e];
                            if constexpr (RES) o[e] += r[j % 3][i][e];
                            o[e] = fmaxf(o[e], o[e] * slope_eff);  // LeakyReLU / identity (slope_eff in [0, 1], wino4_supported): the sum above is canonical, so this is mul + v_max
                        }
                        if (elu) {  // (wave-uniform; the LeakyReLU / identity path above costs ELU layers 8 idle operations, the others nothing)
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? __expf(o[e]) - 1.0f : o[e];
                        }
#ifdef IDH_ABL_W4_EPILIN
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, ((cur.y0 / 8 * wa.tiles_x + cur.x0 / 32) * 64 + (4 * i + j) * 4 + wave) * 1024 + lane_e * 16, 0, 0);
#else
                        const int px = pixel(i, j);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, px >= 0 ? (px * a.out_cs + n0 + 4 * h) * 4 : kOob, 0, 0);
#endif
                    }
                    if constexpr (RES) {
                        if (j == 0) {  // column 3 into the slot column 0 just left
                            __builtin_amdgcn_sched_barrier(0);
                            ld_res(3);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else {
                // ---- P phase: Y[pixel] += W1 . x2[pixel] as MFMAs in the pixel domain.  After the output transform a lane holds, for each of the 16 pixels
                // of its tile, channels 4h..4h+3: exactly the C layout of a 16 (channels) x 16 (tiles) MFMA block per pixel.  x2 goes through LDS in chunks of
                // 16 channels x 256 pixels (16 KiB), double-buffered in what is free at this point: buffer 0 = the V buffer the last stage left (the other one
                // holds the next tile's V(0)), buffer 1 = the two free halo planes (pl0: the next tile's halo(0), already transformed; pl2), 8 KiB = two channel
                // quads in each.  Layout [channel-quad half][pixel of the tile][quad & 1][tile n] x 16 B: a wave's B operand of a pixel is one conflict-free
                // ds_read_b128.  One barrier per chunk: chunk m + 1 is written while chunk m is multiplied.  The 1x1 weights are read from the direct
                // kernels' layout [ci / 4][Cout][4] (idh_pack_conv_weight), one 16-byte fragment per lane and chunk.
                f32x4 Y[16];  // [4 i + j]
#pragma unroll
                for (int j = 0; j < 4; ++j) at6(u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j], Y[j], Y[4 + j], Y[8 + j], Y[12 + j]);
                const ConvSrc &s2 = a.s[1];
                const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s2.in + (size_t)cur.img * s2.H * s2.W * s2.cs), 0, s2.H * s2.W * s2.cs * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s2.w), 0, s2.cblocks * 4 * a.Cout_pad * 16, 0x00020000);
                // copies: thread t, round r: pixel (row 2 r + (t >> 7), column (t >> 2) & 31) of the 32 x 8 tile group, channel quad t & 3 (64 contiguous bytes per
                // pixel); its tile is 8 (r >> 1) + (column >> 2), its pixel of the tile 4 (2 (r & 1) + (t >> 7)) + (column & 3)
                int voffX[4];
                const int hq = tid_e & 3, xx = (tid_e >> 2) & 31, t7 = tid_e >> 7;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int iy = cur.y0 + 2 * r + t7, ix = cur.x0 + xx;
                    voffX[r] = ((iy < s2.H) & (ix < s2.W)) ? (iy * s2.W + ix) * s2.cs * 4 + 16 * hq : kOob;
                }
                const int wrel = 512 * (4 * t7 + (xx & 3)) + ((hq & 1) * 16 + (xx >> 2)) * 16;  // + 4096 (r & 1) + 128 (r >> 1)
                const int rrel = ((h & 1) * 16 + n) * 16;                                         // + 512 pixel
                int wb_cur = kV1 + 8192 * (hq >> 1) + wrel, wb_nxt = ((hq >> 1) ? pl2 : pl0) + wrel;
                int rb_cur = kV1 + 8192 * (h >> 1) + rrel, rb_nxt = ((h >> 1) ? pl2 : pl0) + rrel;
                const int nM = s2.cblocks;
                const int voffW = ((h * a.Cout_pad) + n0 + n) * 16;
                auto ld_x = [&](f32x4 (&xs)[4], int m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voffX[r], __builtin_amdgcn_readfirstlane(64 * m), 0));
                };
                auto st_x = [&](const f32x4 (&xs)[4], int wb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) *(lds_f32x4 *)(lds + wb + 4096 * (r & 1) + 128 * (r >> 1)) = xs[r];
                };
                auto ld_w = [&](int m) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW1, voffW, __builtin_amdgcn_readfirstlane(m * 4 * a.Cout_pad * 16), 0)); };
                f32x4 xs[4];
                ld_x(xs, 0);
                f32x4 Aw = ld_w(0);
                st_x(xs, wb_cur);
                ld_x(xs, nM > 1 ? 1 : 0);
                __syncthreads();
#pragma unroll 1
                for (int m = 0; m < nM; ++m) {
                    const f32x4 Acur = Aw;
                    Aw = ld_w(m + 1 < nM ? m + 1 : m);
                    st_x(xs, wb_nxt);                    // chunk m + 1 (its buffer was last read in iteration m - 1, one barrier ago)
                    ld_x(xs, m + 2 < nM ? m + 2 : m);   // (tail iterations reload a chunk they do not use: harmless)
#pragma unroll
                    for (int pp = 0; pp < 16; ++pp) {
                        const f32x4 Bx = *(lds_cf32x4 *)(lds + rb_cur + 512 * pp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) Y[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(Acur[e], Bx[e], Y[pp], 0, 0, 0);
                    }
                    __syncthreads();
                    int t = wb_cur; wb_cur = wb_nxt; wb_nxt = t;
                    t = rb_cur; rb_cur = rb_nxt; rb_nxt = t;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool ok = (oy0 + i < a.Ho) & (ox0 + j < a.Wo);
                        const int pix = ok ? (oy0 + i) * a.Wo + ox0 + j : -1;
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = Y[4 * i + j][e] + b4[e];
                            o[e] = fmaxf(o[e], o[e] * slope_eff);  // LeakyReLU / identity (slope_eff in [0, 1], wino4_supported): the sum above is canonical, so this is mul + v_max
                        }
                        if (elu) {  // (wave-uniform; the LeakyReLU / identity path above costs ELU layers 8 idle operations, the others nothing)
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? __expf(o[e]) - 1.0f : o[e];
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, pix >= 0 ? (pix * a.out_cs + n0 + 4 * h) * 4 : kOob, 0, 0);
                    }
                }
            }
        }
#endif
        W4T(71);
        if (!has_next) break;
        t_cur = t_next;
        cur = nxt;
        derive_lane_constants();
#ifdef IDH_ABL_W4_TRACE
        ++tile_i;
#endif
    }
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
    const ConvSrc &s1 = a.s[1];
    // second source: a 1x1 stride-1 projection of a tensor of the output's size (weights in idh_pack_conv_weight's layout), no residual beside it
    const bool src2_ok = !s1.in || (s1.ks == 1 && s1.stride == 1 && !s1.up_in[0] && !s1.norm && s1.H == a.Ho && s1.W == a.Wo && !a.res &&
                                    (long long)s1.H * s1.W * s1.cs * 4 < (1ll << 31) && (long long)s1.cblocks * a.Cout_pad * 64 < (1ll << 31));
    // (LeakyReLU is evaluated as max(x, slope x): slopes in [0, 1] - every activation of the reference's conv stacks, ReLU = 0 included)
    return src2_ok && (a.act == IDH_ACT_NONE || (a.act == IDH_ACT_LRELU && a.slope >= 0.f && a.slope <= 1.f) || a.act == IDH_ACT_ELU) && s.cblocks >= 2 && s.ks == 3 && s.stride == 1 && s.pad_mode == IDH_PAD_ZEROS && !s.up_in[0] && !s.norm && a.S == 1 && (a.Cout % 64) == 0 &&
           (long long)s.H * s.W * s.cs * 4 < (1ll << 30) && (long long)a.Ho * a.Wo * a.out_cs * 4 < (1ll << 31) &&  // (input: the halo offsets advance by up to 8 rows past an out-of-range marker)
           (!a.res || (long long)a.Ho * a.Wo * a.res_cs * 4 < (1ll << 31)) && (long long)s.cblocks * a.Cout_pad * 36 * 16 * 4 < (1ll << 31);
}

// 32 x 8 pixel x 64 channel tiles, two persistent workgroups per CU
int launch_conv_wino4p(const ConvArgs &a, int N, hipStream_t st);                                // conv_wino4p.hip
void launch_pack_wino4p(const float *w, float *dst, int Cout, int Cin, hipStream_t st);

// Developer switch IDH_W4_SPLIT=1 (read once per process): the layers without a fused 1x1 projection run on the position-split kernel
// conv3x3_wino4p_k (conv_wino4p.hip: 8 waves per workgroup, 18 positions per wave, four waves per SIMD) and the packed blob then holds BOTH
// fragment orders, conv3x3_wino4_k's first.  Measured 6-8 % SLOWER than this kernel on the big layers (profiles/r05/experiments.md), so it is
// off by default; kept buildable and parity-tested (tests/test_conv_wino4_gpu.py) as the record of that experiment.
static bool wino4_use_split() {
    static const bool on = getenv("IDH_W4_SPLIT") && atoi(getenv("IDH_W4_SPLIT")) != 0;
    return on;
}

bool wino4_split_enabled() { return wino4_use_split(); }

int launch_conv_wino4(const ConvArgs &a, int N, hipStream_t st) {
    if (!wino4_supported(a)) return IDH_EUNSUPPORTED;
    if (!a.s[1].in && wino4_use_split()) {
        ConvArgs b = a;
        b.s[0].w = a.s[0].w + (size_t)a.s[0].cblocks * 16 * a.Cout_pad * 36;
        return launch_conv_wino4p(b, N, st);
    }
    Wino4Args wa{a, (a.Wo + 31) / 32, (a.Ho + 7) / 8, 0};
    wa.c.NT = a.Cout / 64;
    const long long tiles = (long long)N * wa.tiles_x * wa.tiles_y * wa.c.NT;
    if (tiles >= (1ll << 31)) return IDH_EUNSUPPORTED;
    wa.tiles = (int)tiles;
    long long grid = 2ll * wino4_cus();
#ifdef IDH_ABL_W4_GRIDENV  // (experiment: IDH_W4_GRID = persistent workgroups)
    if (getenv("IDH_W4_GRID")) grid = atoi(getenv("IDH_W4_GRID"));
#endif
    if (grid > wa.tiles) grid = wa.tiles >= 8 ? wa.tiles / 8 * 8 : wa.tiles;
    if (a.s[1].in) hipLaunchKernelGGL((conv3x3_wino4_k<true, false>), dim3((unsigned)grid), dim3(256), 0, st, wa);
    else if (a.res) hipLaunchKernelGGL((conv3x3_wino4_k<false, true>), dim3((unsigned)grid), dim3(256), 0, st, wa);
    else hipLaunchKernelGGL((conv3x3_wino4_k<false, false>), dim3((unsigned)grid), dim3(256), 0, st, wa);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

}  // namespace idh_conv

extern "C" size_t idh_packed_wino4_weight_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (idh_conv::wino4_split_enabled() ? 2 : 1) * (size_t)((Cin + 15) & ~15) * ((Cout + 15) & ~15) * 36;  // (both fragment orders: launch_conv_wino4)
}

extern "C" int idh_pack_conv_weight_wino4(const float *w, float *dst, int Cout, int Cin, void *stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0) return IDH_EINVAL;
    const int nS = ((Cin + 15) / 16) * 2, nCB = (Cout + 15) / 16;
    const long long total = (long long)nS * nCB * kPanelFloats;
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_wino4_weight_k, dim3(grid), dim3(256), 0, idh_stream(stream), w, dst, Cout, Cin, nS, nCB);
    IDH_CHECK_LAUNCH();
    if (idh_conv::wino4_split_enabled()) {
        idh_conv::launch_pack_wino4p(w, dst + (size_t)((Cin + 15) & ~15) * ((Cout + 15) & ~15) * 36, Cout, Cin, idh_stream(stream));
        IDH_CHECK_LAUNCH();
    }
    return IDH_OK;
}
