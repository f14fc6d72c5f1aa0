// Winograd F(4x4, 3x3) on the fp32 matrix cores of gfx950, POSITION-SPLIT variant (conv3x3_wino4p_k): the same arithmetic, LDS layout and copy
// pipeline as conv3x3_wino4_k (conv_wino4.hip: read that header first), with the 36 transform-domain positions of a 16-channel block split
// between TWO waves - 18 positions x 4 = 72 accumulator registers per wave instead of 144 - so that a workgroup is 8 waves (4 channel blocks x 2
// position halves), a wave fits 128 registers and FOUR waves share a SIMD (two workgroups per CU as before).
//
// Why (profiles/r05/experiments.md): at two waves per SIMD the kernel is bound by the serial chain INSIDE a wave - 72 MFMAs (2.3k cycles of the
// pipe), then the transform (2.4-3.3k cycles of LDS round trips), copies, barrier: 8.5k cycles per stage against 4.6k of matrix time for the
// two waves - and every ablation (A rows, halo, transform, epilogue) removes 5-20 % because each shortens that chain.  With the positions split,
// a wave's chain per stage is 36 MFMAs + half a quadrant transform, four waves interleave on the SIMD, and the matrix pipe, not the chain, is
// the bound.  Replaces the same nn.Conv2d calls as conv3x3_wino4_k (reference modules/layers.py:59-95, modules/networks.py:20-215) for the
// layers WITHOUT a fused 1x1 projection; those keep conv3x3_wino4_k<true, false>.
//
// Differences from conv3x3_wino4_k:
// * wave w: channel block cb = w & 3, position half ph = w >> 2 (positions p' = 18 ph .. 18 ph + 17 of the quadrant-major order, i.e. the two
//   quadrants with xi in 3 ph .. 3 ph + 2).  Per 8-channel stage: 36 MFMAs, 9 A rows (1 KiB each; packed [stage][cb][ph][row][lane][4], element
//   m = 4 row + e: position m % 18, k-step m / 18), 18 ds_read_b64 of V (layout [k-step][ph][lane][18]: 72-byte lane stride, conflict-free).
// * transform: wave w takes quadrant w & 3 for the channels 4 (w >> 2) .. + 3 of the stage, ONE channel per lane (lane = (tile n, channel c)):
//   25 ds_read_b32, 54 vector operations, 9 V writes - half the registers of the two-channel form.
// * halo copies: 512 threads, three 64-byte-per-texel copies per pair of stages (rows 4 k + (t >> 7); the third also carries columns 32, 33).
// * epilogue: Y = A^T M A is linear in the rows of M, so each wave transforms ITS 18 positions (three rows xi of M) into a partial 4 x 4 output,
//   the two waves of a channel block exchange halves through LDS (two rounds of four pixel quads in what the stage loop leaves free) and each
//   finishes 8 of the 16 pixels of every tile: + bias (+ residual), activation, 16-byte stores.
#include <cstdlib>
#include <type_traits>

#include "conv_args.h"
#include "../../include/idh_ops.h"

using namespace idh_conv;

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(3))) volatile float lds_cfloat;
typedef const __attribute__((address_space(3))) f32x2 lds_cf32x2;
typedef const __attribute__((address_space(3))) f32x4 lds_cf32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) float lds_float;

constexpr int kOob = 0x7fffffff;

struct Wino4pArgs {
    ConvArgs c;
    int tiles_x, tiles_y;
    int tiles;  // N * tiles_y * tiles_x * NT
};

__device__ __forceinline__ int fresh_lane() {  // (see conv_wino4.hip)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

constexpr int kPlane = 432 * 32 + 128;  // as conv_wino4.hip
constexpr int kVBytes = 2 * 2 * 64 * 72;  // V of one stage: [k-step 2][position half 2][lane 64][18 floats]
constexpr int kV0 = 3 * kPlane, kV1 = 3 * kPlane + kVBytes;
constexpr int kLdsBytes = 3 * kPlane + 2 * kVBytes;  // 78720
constexpr int kPanelFloats = 36 * 16 * 8;           // one stage's weights of one 16-channel block (both position halves)

__host__ __device__ constexpr int w4_xi(int pp) { return 3 * ((pp / 9) >> 1) + (pp % 9) / 3; }
__host__ __device__ constexpr int w4_nu(int pp) { return 3 * ((pp / 9) & 1) + (pp % 9) % 3; }

// OIHW 3x3 -> U = G g G^T as A fragments of the position-split kernel:
// dst[stage c][co block cb][half ph][row 9][lane 64][e 4] = U[p' = 18 ph + m % 18][co = 16 cb + (lane & 15)][ci = 8 c + 2 (lane >> 4) + m / 18], m = 4 row + e
__global__ __launch_bounds__(256) void pack_wino4p_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout, int Cin, int nS, int nCB) {
    const long long total = (long long)nS * nCB * kPanelFloats;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int e = (int)(t & 3), lane = (int)((t >> 2) & 63);
        long long r = t >> 8;
        const int row = (int)(r % 9); r /= 9;
        const int ph = (int)(r & 1); r >>= 1;
        const int cb = (int)(r % nCB), c = (int)(r / nCB);
        const int m = 4 * row + e;
        const int pp = 18 * ph + m % 18, ks = m / 18, co = 16 * cb + (lane & 15), ci = 8 * c + 2 * (lane >> 4) + ks;
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

// half of the 1-D input transform B^T (see conv_wino4.hip)
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
template <bool HI, int K>
__device__ __forceinline__ void bt3_step(float w, float &s0, float &s1, float &s2) {
    if constexpr (!HI) {
        if constexpr (K == 0) s0 = w;
        if constexpr (K == 1) s1 = w;
        if constexpr (K == 2) { s0 = __builtin_fmaf(-4.25f, w, s0); s2 = w; }
        if constexpr (K == 3) s1 = __builtin_fmaf(-4.f, s1, w);
        if constexpr (K == 4) { s0 = s0 + w; s2 = __builtin_fmaf(-4.f, s2, w); }
    } else {
        if constexpr (K == 0) { s0 = w; s2 = w; }
        if constexpr (K == 1) s1 = w;
        if constexpr (K == 2) { s2 = __builtin_fmaf(-4.25f, w, s2); s0 = __builtin_fmaf(-0.25f, s0, w); }
        if constexpr (K == 3) s1 = __builtin_fmaf(-0.25f, s1, w);
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
// 1-D output transform A^T (6 -> 4) over the FULL row (component-wise, see conv_wino4.hip)
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
// ... and over ONE HALF of the rows (the column pass of a position half): rows xi = 0..2 (HI = false) or 3..5 (HI = true) of A^T
template <bool HI>
__device__ __forceinline__ void at3(f32x4 u0, f32x4 u1, f32x4 u2, f32x4 &y0, f32x4 &y1, f32x4 &y2, f32x4 &y3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (!HI) {  // columns [1 0 0 0], [1 1/2 1/4 1/8], [1 -1/2 1/4 -1/8]
            const float s = u1[e] + u2[e], d = u1[e] - u2[e];
            y0[e] = u0[e] + s; y1[e] = 0.5f * d; y2[e] = 0.25f * s; y3[e] = 0.125f * d;
        } else {  // columns [1 2 4 8], [1 -2 4 -8], [0 0 0 1]
            const float s = u0[e] + u1[e], d = u0[e] - u1[e];
            y0[e] = s; y1[e] = 2.f * d; y2[e] = 4.f * s; y3[e] = __builtin_fmaf(8.f, d, u2[e]);
        }
    }
}

template <bool RES>
__global__ __launch_bounds__(512, 4) void conv3x3_wino4p_k(const Wino4pArgs wa) {
    __shared__ __attribute__((aligned(16))) char lds_raw[kLdsBytes];
    lds_char *lds = (lds_char *)lds_raw;

    const ConvArgs &a = wa.c;
    const ConvSrc &s = a.s[0];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cbw = wave & 3, ph = wave >> 2;  // this wave's 16-channel block of the 64-channel tile / position half
    const int nS = s.cblocks * 2;  // stages of 8 input channels
    const int NT = a.NT;           // 64-channel tiles
    const int nCB = 4 * NT;        // 16-channel blocks of the packed weights

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

    // ---- halo copies (global -> registers -> LDS), a pair of stages (16 channels = 64 B per texel) at a time, 512 threads: copy k = 0, 1, 2 of thread t
    // is texel (row 4 k + (t >> 7), column (t >> 2) & 31), granule t & 3 (rows 10, 11 of copy 2 do not exist: its threads 256..335 copy columns 32, 33
    // of the ten rows instead).  LDS slot of texel (r, col): p = ((4 (r & 3) + (col & 3)) * 3 + (r >> 2)) * 9 + (col >> 2); granules 0, 1 go to the even
    // stage's plane, 2, 3 to the odd stage's, channel quad q in 16-byte half q ^ ((r >> 2) & 1).  All per-lane addresses are derived from the lane id
    // read afresh at the point of use (conv_wino4.hip: fresh_lane).
    __amdgpu_buffer_rsrc_t rsH;
    int cy0 = 0, cx0 = 0;
    auto set_halo_cursor = [&](const Tile &t) {
        rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)t.img * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
        cy0 = t.y0; cx0 = t.x0;
    };
    auto ld_halo = [&](int k, int chs) -> f32x4 {  // chs: the pair's first stage
#ifdef IDH_ABL_W4_NOHALO  // (timing experiments, tools/abl_wino4.sh: results are meaningless)
        return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
        const int t = 64 * wave + fresh_lane();
        int voff;
        if (k < 2 || t < 256) {
            const int iy = cy0 - 1 + 4 * k + (t >> 7), ix = cx0 - 1 + ((t >> 2) & 31);
            voff = ((unsigned)ix < (unsigned)s.W) & (4 * k + (t >> 7) < 10) ? (iy * s.W + ix) * s.cs * 4 + 16 * (t & 3) : kOob;
        } else {
            const int e = (t - 256) >> 2;
            const int iy = cy0 - 1 + (e >> 1), ix = cx0 + 31 + (e & 1);
            voff = ((e < 20) & (ix < s.W) & ((unsigned)iy < (unsigned)s.H)) ? (iy * s.W + ix) * s.cs * 4 + 16 * (t & 3) : kOob;
        }
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, voff, __builtin_amdgcn_readfirstlane(32 * chs), 0));
    };
    auto st_halo = [&](int k, f32x4 v, int offA, int offB) {
        const int t = 64 * wave + fresh_lane();
        const int off = (t >> 1) & 1 ? offB : offA, wq = t & 1;
        if (k < 2 || t < 256) {
            if (4 * k + (t >> 7) < 10) {
                const int rr = t >> 7, col = (t >> 2) & 31;
                *(lds_f32x4 *)(lds + off + 32 * ((4 * rr + (col & 3)) * 27 + 9 * k + (col >> 2)) + 16 * (wq ^ (k & 1))) = v;
            }
        } else {
            const int e = (t - 256) >> 2, r = e >> 1;
            if (e < 20) *(lds_f32x4 *)(lds + off + 32 * (((4 * (r & 3) + (e & 1)) * 3 + (r >> 2)) * 9 + 8) + 16 * (wq ^ ((r >> 2) & 1))) = v;
        }
    };

    // ---- transform of the stage whose halo is in the plane at `hoff`: this wave's quadrant (qa, qb) = (wave & 3) >> 1, wave & 1 for the channels
    // 4 (wave >> 2) + c, c = lane >> 4, of all 16 tiles (n = lane & 15) -> V.  The B operand lane of channel ch is (n, h = ch >> 1) at k-step ch & 1.
    auto transform = [&](auto hic, auto hjc, int hoff, int vbuf) {
        constexpr bool HI_I = decltype(hic)::value, HI_J = decltype(hjc)::value;
        constexpr int I0 = HI_I ? 1 : 0, J0 = HI_J ? 1 : 0;
        const int ln = fresh_lane(), n_ = ln & 15, c_ = ln >> 4, ty_ = n_ >> 3, tx_ = n_ & 7, half = wave >> 2;
        const int rbase = 32 * (9 * ty_ + tx_) + 16 * (half ^ (ty_ & 1)) + 4 * c_ + hoff;  // rows i < 4; rows 4, 5 (R + 1): the granule bit flips (^ 16)
        const int rb[2] = {rbase, ((rbase - hoff) ^ 16) + hoff};
        auto rd = [&](int i, int c) -> float { return *(lds_cfloat *)(lds + rb[i >> 2] + 32 * (((4 * (i & 3) + (c & 3)) * 3 + (i >> 2)) * 9 + (c >> 2))); };
        float d[2][5];
#pragma unroll
        for (int k = 0; k < 5; ++k) d[0][k] = rd(I0 + k, J0);
        float S[3][3];
        auto column = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if (K + 1 < 5) {
#pragma unroll
                for (int k = 0; k < 5; ++k) d[(K + 1) & 1][k] = rd(I0 + k, J0 + K + 1);
            }
            float w[3];
            bt3<HI_I>(d[K & 1][0], d[K & 1][1], d[K & 1][2], d[K & 1][3], d[K & 1][4], w[0], w[1], w[2]);
#pragma unroll
            for (int r = 0; r < 3; ++r) bt3_step<HI_J, K>(w[r], S[r][0], S[r][1], S[r][2]);
        };
        column(std::integral_constant<int, 0>{});
        column(std::integral_constant<int, 1>{});
        column(std::integral_constant<int, 2>{});
        column(std::integral_constant<int, 3>{});
        column(std::integral_constant<int, 4>{});
        const int ch = 4 * half + c_;
        const int vdst = vbuf + ((((ch & 1) * 2 + (HI_I ? 1 : 0)) * 64 + 16 * (ch >> 1) + n_) * 72) + 36 * (HI_J ? 1 : 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float o[3];
            bt3_finish<HI_J>(S[r][0], S[r][1], S[r][2], o[0], o[1], o[2]);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) *(lds_float *)(lds + vdst + 4 * (3 * r + cc)) = o[cc];
        }
    };
    auto transform_q = [&](int hoff, int vbuf) {  // (wave-uniform 4-way dispatch, once per stage)
#ifdef IDH_ABL_W4_NOXFORM
        return;
#endif
        const int qa = (wave & 3) >> 1, qb = wave & 1;
        if (qa == 0 && qb == 0) transform(std::false_type{}, std::false_type{}, hoff, vbuf);
        else if (qa == 0) transform(std::false_type{}, std::true_type{}, hoff, vbuf);
        else if (qb == 0) transform(std::true_type{}, std::false_type{}, hoff, vbuf);
        else transform(std::true_type{}, std::true_type{}, hoff, vbuf);
    };

    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, nS * nCB * kPanelFloats * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    // A fragments: a ring of 3 rows across stage and tile boundaries (row j of a stage from Af[j % 3], reloaded at once with the row 3 further on)
    constexpr int kRing = 3, kRows = 9, kHalfPanel = kPanelFloats * 2;  // bytes of one (stage, block, half): 9 KiB
    f32x4 Af[kRing];
    int voffA = 16 * fresh_lane();
    auto ldA = [&](int slot, int so) {
#ifdef IDH_ABL_W4_NOA
        Af[slot] = (f32x4){1.f, 2.f, 3.f, 4.f};
        return;
#endif
        Af[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffA, so, 0));
    };
    auto panel = [&](int c, int nt) { return ((c * nCB + 4 * nt + cbw) * 2 + ph) * kHalfPanel; };

    // ---- prologue: halo(0), halo(1) of the first tile into planes 0, 1; V(0); the first A rows
    Tile cur = decode(t_cur);
    set_halo_cursor(cur);
    {
        f32x4 t0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) t0[k] = ld_halo(k, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) st_halo(k, t0[k], 0, kPlane);
    }
#pragma unroll
    for (int j = 0; j < kRing; ++j) ldA(j, __builtin_amdgcn_readfirstlane(panel(0, cur.nt) + 1024 * j));
    __syncthreads();
    transform_q(0, kV0);
    // even stage S: halo(S + 2) -> plane pl2, halo(S + 3) -> plane pl0, in a batch of 1 and one of 2 copies; at its entry stg[0] holds the first in flight
    f32x4 stg[2];
    stg[0] = ld_halo(0, 2);
    __syncthreads();
    int pl0 = 0, pl1 = kPlane, pl2 = 2 * kPlane;  // LDS offsets of the planes of halo(S), halo(S + 1), halo(S + 2) (rotated every stage)

    f32x4 acc[18];
#pragma unroll 1
    for (;;) {
        const int t_next = t_cur + t_stride;
        const bool has_next = t_next < t_end;
        const Tile nxt = has_next ? decode(t_next) : cur;
#pragma unroll
        for (int p = 0; p < 18; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

        auto stage = [&](auto parc, const int c) {
            constexpr int PAR = decltype(parc)::value;
            constexpr int kVr = PAR ? kV1 : kV0, kVw = PAR ? kV0 : kV1;
            const int chs = c + 2 >= nS ? c + 2 - nS : c + 2;  // (even stages)
            const int aso = __builtin_amdgcn_readfirstlane(panel(c, cur.nt));
            const bool last = c + 1 >= nS;
            const int aso_n = __builtin_amdgcn_readfirstlane(last ? panel(0, nxt.nt) : panel(c + 1, cur.nt));
            // B operands: V[k-step][ph][lane][18] as 9 ds_read_b64 per k-step (positions 2 i, 2 i + 1)
            const int vb = kVr + (ph * 64 + fresh_lane()) * 72;
            f32x2 Bq[18];
#ifdef IDH_ABL_W4_NOB
            auto ldB = [&](int i) { Bq[i] = (f32x2){1.f, 2.f}; };
#else
            auto ldB = [&](int i) { Bq[i] = *(lds_cf32x2 *)(lds + vb + (i / 9) * (2 * 64 * 72) + 8 * (i % 9)); };
#endif
            constexpr int kAheadB = 4;
#pragma unroll
            for (int i = 0; i < kAheadB; ++i) ldB(i);
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 4 * j + e, pos = m % 18, ks = m / 18, bi = 9 * ks + pos / 2;
                    if (e % 2 == 0 && bi + kAheadB < 18) ldB(bi + kAheadB);
                    acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[j % kRing][e], Bq[bi][pos & 1], acc[pos], 0, 0, 0);
                }
                ldA(j % kRing, j + kRing < kRows ? aso + 1024 * (j + kRing) : aso_n + 1024 * (j + kRing - kRows));
                if (PAR == 0 && j == 4) {
                    st_halo(0, stg[0], pl2, pl0);
                    stg[0] = ld_halo(1, chs);
                    stg[1] = ld_halo(2, chs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PAR == 0) {
                st_halo(1, stg[0], pl2, pl0);
                st_halo(2, stg[1], pl2, pl0);
            } else {
                // first copy of the next even stage's pair (halo(E + 2), halo(E + 3), E = c + 1 or stage 0 of the next tile), issued HERE: loads complete in
                // order, so an A row issued after a copy cannot be used before the copy is back from HBM
                if (c + 3 == nS) set_halo_cursor(nxt);
                const int chn = c + 3 >= nS ? c + 3 - nS : c + 3;
                stg[0] = ld_halo(0, chn);
            }
            transform_q(pl1, kVw);
            __syncthreads();
            const int t0 = pl0; pl0 = pl1; pl1 = pl2; pl2 = t0;
        };
#pragma unroll 1
        for (int c = 0; c < nS; c += 2) {
            stage(std::integral_constant<int, 0>{}, c);
            stage(std::integral_constant<int, 1>{}, c + 1);
        }

#ifdef IDH_ABL_W4_NOEPI
#pragma unroll
        for (int p = 0; p < 18; ++p) asm volatile("" ::"v"(acc[p]));
        if (false)
#endif
        // ---- epilogue.  acc[9 b + 3 (xi % 3) + nu % 3] = M[xi = 3 ph + ..][nu = 3 b + ..]: row pass over the full rows (nu 0..5 -> 4 columns), column pass over this
        // half's three rows -> the partial output P[i][j] (i = pixel row); the pair of waves of a channel block swaps halves through LDS and wave ph
        // finishes pixel rows 2 ph, 2 ph + 1.
        {
            const int lane_e = fresh_lane();
            const int n = lane_e & 15, h = lane_e >> 4, ty = n >> 3, tx = n & 7;
            const int n0 = 16 * (4 * cur.nt + cbw);
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.img * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)cur.img * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                                  a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
            const int oy0 = cur.y0 + 4 * ty + 2 * ph, ox0 = cur.x0 + 4 * tx;  // first of this wave's two pixel rows
            const float slope_eff = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(a.act == IDH_ACT_LRELU ? __builtin_bit_cast(int, a.slope) : 0x3f800000));
            const bool elu = a.act == IDH_ACT_ELU;
            const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (n0 + 4 * h) * 4, 0, 0));
            auto pixel = [&](int i, int j) -> int { return ((oy0 + i < a.Ho) & (ox0 + j < a.Wo)) ? (oy0 + i) * a.Wo + ox0 + j : -1; };  // i = 0, 1
            // row pass: u[x][j], x = xi % 3
            f32x4 u[3][4];
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                at6(acc[3 * x], acc[3 * x + 1], acc[3 * x + 2], acc[9 + 3 * x], acc[9 + 3 * x + 1], acc[9 + 3 * x + 2], u[x][0], u[x][1], u[x][2], u[x][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {  // (pinned: see conv_wino4.hip)
                        float t = u[x][j][e];
                        asm volatile("" : "+v"(t));
                        u[x][j][e] = t;
                    }
            }
            // residual of this wave's 8 pixels: issued once the accumulators are dead, back by the time the exchange is through
            f32x4 r[2][4];
            if constexpr (RES) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int px = pixel(i, j);
                        r[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, px >= 0 ? (px * a.res_cs + n0 + 4 * h) * 4 : kOob, 0, 0));
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // column pass + exchange, instantiated per position half (ph is wave-uniform but not a compile-time constant: indexing P[][] with it would
            // put the array into scratch): this wave keeps pixel rows 2 ph, 2 ph + 1 of its partial output and sends the other two to its partner
            // (wave ^ 4), one row (4 quads) per round.  Free LDS at this point: the V buffer the last stage read (kV1), plane pl0 (the next tile's
            // halo(0), already transformed) and plane pl2.
            const int xw = (wave < 4 ? kV1 + 4096 * wave : (wave < 6 ? pl0 + 4096 * (wave - 4) : pl2 + 4096 * (wave - 6))) + 16 * lane_e;
            const int pw = wave ^ 4;
            const int xr = (pw < 4 ? kV1 + 4096 * pw : (pw < 6 ? pl0 + 4096 * (pw - 4) : pl2 + 4096 * (pw - 6))) + 16 * lane_e;
            f32x4 Y[2][4];
            auto finish = [&](auto phc) {
                constexpr bool PH = decltype(phc)::value;
                f32x4 P[4][4];  // P[i][j], i = pixel row
#pragma unroll
                for (int j = 0; j < 4; ++j) at3<PH>(u[0][j], u[1][j], u[2][j], P[0][j], P[1][j], P[2][j], P[3][j]);
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *(lds_f32x4 *)(lds + xw + 1024 * j) = P[(PH ? 0 : 2) + rd][j];
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 got = *(lds_cf32x4 *)(lds + xr + 1024 * j);
#pragma unroll
                        for (int e = 0; e < 4; ++e)  // (rows 0..2 + rows 3..5, in that order in both waves)
                            Y[rd][j][e] = PH ? got[e] + P[2 + rd][j][e] : P[rd][j][e] + got[e];
                    }
                    __syncthreads();
                }
            };
            if (ph == 0) finish(std::false_type{});
            else finish(std::true_type{});
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = Y[i][j][e] + b4[e];
                        if constexpr (RES) o[e] += r[i][j][e];
                        o[e] = fmaxf(o[e], o[e] * slope_eff);  // LeakyReLU / identity (slope_eff in [0, 1], wino4_supported): the sum above is canonical, so this is mul + v_max
                    }
                    if (elu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? __expf(o[e]) - 1.0f : o[e];
                    }
                    const int px = pixel(i, j);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsO, px >= 0 ? (px * a.out_cs + n0 + 4 * h) * 4 : kOob, 0, 0);
                }
        }
        if (!has_next) break;
        t_cur = t_next;
        cur = nxt;
        voffA = 16 * fresh_lane();
    }
}

int wino4p_cus() {
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

// (same preconditions as wino4_supported, checked by the caller; no second source)
int launch_conv_wino4p(const ConvArgs &a, int N, hipStream_t st) {
    Wino4pArgs wa{a, (a.Wo + 31) / 32, (a.Ho + 7) / 8, 0};
    wa.c.NT = a.Cout / 64;
    const long long tiles = (long long)N * wa.tiles_x * wa.tiles_y * wa.c.NT;
    if (tiles >= (1ll << 31) || a.s[1].in) return IDH_EUNSUPPORTED;
    wa.tiles = (int)tiles;
    long long grid = 2ll * wino4p_cus();
    if (grid > wa.tiles) grid = wa.tiles >= 8 ? wa.tiles / 8 * 8 : wa.tiles;
    if (a.res) hipLaunchKernelGGL(conv3x3_wino4p_k<true>, dim3((unsigned)grid), dim3(512), 0, st, wa);
    else hipLaunchKernelGGL(conv3x3_wino4p_k<false>, dim3((unsigned)grid), dim3(512), 0, st, wa);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

void launch_pack_wino4p(const float *w, float *dst, int Cout, int Cin, hipStream_t st) {
    const int nS = ((Cin + 15) / 16) * 2, nCB = (Cout + 15) / 16;
    const long long total = (long long)nS * nCB * kPanelFloats;
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_wino4p_weight_k, dim3(grid), dim3(256), 0, st, w, dst, Cout, Cin, nS, nCB);
}

}  // namespace idh_conv
