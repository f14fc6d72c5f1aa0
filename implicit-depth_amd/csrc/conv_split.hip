// 3x3 stride-1 convolution on the 16-bit matrix cores with fp32-equivalent results ("split
// precision").  gfx950 runs bf16 / f16 MFMA at 16x the fp32-MFMA rate (2.5 PFLOP/s vs 157 TFLOP/s
// dense) and has no TF32, so the fp32 operands are expanded into 16-bit pieces whose cross
// products are accumulated in fp32 by v_mfma_f32_32x32x16_f16.  Opt-in (IDH_OP_CONV with
// tile_m = 11); the default path stays on v_mfma_f32_16x16x4_f32.  (A three-piece bf16 variant, "bf16x6", existed in
// round 1; it was 1.5x slower than f16x3 at the same accuracy and was removed to shrink the surface.)
//
//  MODE_F16X3 (tile_m = 11): x/s = x0 + x1, two round-to-nearest f16 pieces (11+11 bits + sign:
//     |x/s - x0 - x1| <= 2^-23 |x/s|), products x0w0 + x0w1 + x1w0 (the dropped x1w1 is 2^-22).
//     f16 has a 5-bit exponent, so operands are scaled by exact powers of two into [2^14, 2^15):
//     weights per output channel at pack time, activations per 18x18x16 halo chunk by the running
//     maximum of the workgroup's chunk maxima (the accumulators are rescaled, again by an exact
//     power of two, when that maximum grows).  Elements more than 2^18 below the running maximum
//     fall into f16's subnormal range and keep an absolute error of 2^-40 of that maximum.
//  Measured against fp64 (tests/test_conv_split_gpu.py): it errs by ~4e-7 of the output
//  scale at K = 576..1728, the fp32-MFMA kernel by ~5e-7 (bar: 1e-4).
//
// Shape family: the layers that carry the flops of CVEncoder / UNet++ (layers.py:59-95): 3x3,
// stride 1, zero padding, Cout % 64 == 0, one source.
//
// Workgroup = 16x16 output pixels x 64 channels.  D^T = W * X^T: weights are the A operand, so a
// lane ends up with 4 consecutive output channels of one pixel -> 16-byte NHWC stores.
// Per 16-channel K chunk the 18x18 halo is split into pieces while it is written to LDS
//     sH[piece][kg 2][18*18 (+4 pad)] x 16 B   (kg = which 8 of the 16 channels; one slot = one
//                                                MFMA operand of one pixel)
// and per (chunk, tap row) the pre-split weight panel (packed by idh_pack_conv_weight_split in
// exactly this order) is copied to one of two LDS buffers
//     sW[buf 2][tap-in-row 3][piece][kg 2][co 64] x 16 B
// so a phase = 3 taps between barriers; the weight buffer is double-buffered (one barrier per
// phase), the halo single-buffered (one extra barrier per chunk); global loads for the next phase /
// chunk are register-prefetched under the MFMAs.
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"
#include "../../include/idh_ops.h"

using namespace idh_conv;

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int MODE_F16X3 = 1;
constexpr int kHalo = kSplitTile + 2;    // 18 halo columns
constexpr int kMinExp = -100;            // lower clamp of the scaling exponents (all-zero tiles)

constexpr int pieces_of(int) { return 2; }
constexpr int wslots_of(int mode) { return 3 * pieces_of(mode) * 2 * 64; }  // 16-B slots per (chunk, tap row, 64-channel tile)
// slots per (piece, kg) plane of a (rows+2) x 18 halo, padded to 8 mod 16 slots (bank offset 32 between the
// two kg planes -> conflict-free ds_write_b64): 324 -> 328 (16 rows), 180 -> 184 (8 rows)
constexpr int plane_of(int rows) { return (((rows + 2) * kHalo + 7) / 16) * 16 + 8; }
constexpr int lds_bytes_of(int mode, int rows) { return (pieces_of(mode) * 2 * plane_of(rows) + 2 * wslots_of(mode)) * 16 + 64; }

__device__ float g_zero16[16];

// x (already scaled into f16 range) ~= h0 + h1, both round-to-nearest-even
__device__ __forceinline__ void split2_f16(float x, _Float16 &h0, _Float16 &h1) {
    h0 = (_Float16)x;
    h1 = (_Float16)(x - (float)h0);
}
__device__ __forceinline__ unsigned pack2_f16(_Float16 lo, _Float16 hi) {
    f16x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float exp2_int(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }  // e in [-126, 127]
// floor(log2(v)) for finite v > 0 from the exponent field, clamped below; inf/nan -> 128
__device__ __forceinline__ int exponent_of(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xFF) - 127;
    return e < kMinExp ? kMinExp : e;
}

// Tile = (2 G WAVES) rows x 16 columns x 64 channels; each wave owns 2G rows = G 32-pixel MFMA column groups.
//   <4, 2>: 16 rows, 64 accumulator registers per wave, 166 VGPRs (f16x3) -> 3 workgroups / CU
//   <8, 1>: 16 rows, 32 accumulators, <= 128 VGPRs -> 2 workgroups x 8 waves
//   <4, 1>:  8 rows (twice the workgroups: small maps / small batches), 36 KiB (f16x3) -> 4 workgroups / CU
//   SRC2: compiled with the fused 1x1 second source (its extra loop costs registers: <4, 2> drops to 2 workgroups / CU)
template <int WAVES, int G, int MODE, bool SRC2>
__global__ __launch_bounds__(64 * WAVES, G == 1 ? 4 : 2) void conv3x3_split_k(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int NP = pieces_of(MODE);
    constexpr int kWSlots = wslots_of(MODE);
    constexpr int kRows = 2 * G * WAVES;
    constexpr int kHaloPix = (kRows + 2) * kHalo;
    constexpr int kPlane = plane_of(kRows);
    constexpr int kHaloSlots = NP * 2 * kPlane;
    constexpr int NT_ = 64 * WAVES;                          // threads
    constexpr int kHaloLoads = (kHaloPix * 4 + NT_ - 1) / NT_;
    constexpr int kWLoads = (kWSlots + NT_ - 1) / NT_;
    constexpr int kWFullWaves = (kWSlots - (kWLoads - 1) * NT_) / 64;  // waves that own a slot in the last round
    extern __shared__ u32x4 smem[];
    u32x4 *sH = smem;
    u32x4 *sW = smem + kHaloSlots;
    float *sMax = reinterpret_cast<float *>(smem + kHaloSlots + 2 * kWSlots);  // [2][8] chunk maxima (MODE_F16X3)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31, kg = lane >> 5;
    // lanes 16..31 (second tile row) take their 16 pixels rotated by 14: with the 18-slot row pitch
    // this puts every ds_read_b128 lane group {0-3,12-15,20-27}, ... on 16 distinct bank quads
    const int prow = p >> 4, px = (p - 2 * prow) & 15;

    unsigned blk = idh_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = blk % a.NT; blk /= a.NT;
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y;
    const int n = blk / tiles_y;
    const int y0 = ty * kRows, x0 = tx * kSplitTile;
    const int n0 = nt * 64;
    const ConvSrc &s = a.s[0];
    const int nC = s.cblocks;
    const int nPh = 3 * nC;
    // optional second source: 1x1 projection of another tensor (BasicBlock's downsample(x), layers.py:68-75),
    // one centre-tap phase per 16-channel chunk, accumulated into the same tile
    const int nC2 = SRC2 ? a.s[1].cblocks : 0;
    const int nChunks = nC + nC2, nPhases = nPh + nC2;
    constexpr int kW1Slots = kWSlots / 3;  // one tap

    f32x16 acc[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[g][j][e] = 0.f;
    int E = kMinExp - 20;  // MODE_F16X3: exponent of the running activation maximum (accumulator unit = 2^(E-14))

    f32x4 ph_[kHaloLoads];
    u32x4 pw_[kWLoads];
    auto issue_halo = [&](int cc) {  // cc indexes [source-0 chunks][source-1 chunks]
        const bool second = SRC2 && cc >= nC;
        const float *base = second ? a.s[1].in : s.in;
        const int cs = second ? a.s[1].cs : s.cs;
        const int ch = 16 * (second ? cc - nC : cc);
#pragma unroll
        for (int k = 0; k < kHaloLoads; ++k) {
            const int slot = tid + NT_ * k;
            const int q = slot & 3, pix = slot >> 2;
            const int hy = pix / kHalo, hx = pix - hy * kHalo;
            const int iy = y0 + hy - 1, ix = x0 + hx - 1;
            const bool ok = (pix < kHaloPix) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            const float *src = ok ? base + ((size_t)(n * s.H + iy) * s.W + ix) * cs + ch + 4 * q : g_zero16;
#ifdef IDH_ABL_NOLOAD  // pseudo-random operands made in registers (keeps the MFMA toggle rate realistic)
            const unsigned hsh = (unsigned)tid * 2654435761u + (unsigned)k * 40503u + (unsigned)cc * 9176u + (unsigned)(size_t)src;
            ph_[k] = (f32x4){__uint_as_float(0x3f800000u | (hsh & 0x7fffffu)) - 1.5f, __uint_as_float(0x3f800000u | ((hsh >> 3) & 0x7fffffu)) - 1.5f,
                             __uint_as_float(0x3f800000u | ((hsh >> 6) & 0x7fffffu)) - 1.5f, __uint_as_float(0x3f800000u | ((hsh >> 9) & 0x7fffffu)) - 1.5f};
#else
            ph_[k] = *reinterpret_cast<const f32x4 *>(src);
#endif
        }
    };
    // max |x| of this thread's prefetched halo values -> per-wave slot of parity `par`
    auto publish_max = [&](int par) {
        float m = 0.f;
        bool bad = false;  // inf / nan must reach the scale (fmaxf drops nan)
#pragma unroll
        for (int k = 0; k < kHaloLoads; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m = fmaxf(m, fabsf(ph_[k][e]));
                bad |= (__float_as_uint(ph_[k][e]) & 0x7F800000u) == 0x7F800000u;
            }
        if (bad) m = __uint_as_float(0x7F800000u);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) sMax[par * 8 + wave] = m;
    };
    auto commit_halo = [&](float mul) {
#ifdef IDH_ABL_NOCOMMIT
        return;
#endif
        u32x2 *sH2 = reinterpret_cast<u32x2 *>(sH);
#pragma unroll
        for (int k = 0; k < kHaloLoads; ++k) {
            const int slot = tid + NT_ * k;
            const int q = slot & 3, pix = slot >> 2;
            unsigned w[NP][2];
            {
                _Float16 h[4][2];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2_f16(ph_[k][e] * mul, h[e][0], h[e][1]);
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    w[pc][0] = pack2_f16(h[0][pc], h[1][pc]);
                    w[pc][1] = pack2_f16(h[2][pc], h[3][pc]);
                }
            }
            if (pix < kHaloPix) {
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    u32x2 v = {w[pc][0], w[pc][1]};
                    sH2[((pc * 2 + (q >> 1)) * kPlane + pix) * 2 + (q & 1)] = v;
                }
            }
        }
    };
    auto issue_w = [&](int ph) {  // ph indexes [3x3 tap-row panels][1x1 panels]
        const u32x4 *w0 = reinterpret_cast<const u32x4 *>(s.w);
        const bool second = SRC2 && ph >= nPh;
        const u32x4 *src = second ? w0 + (size_t)nPh * a.NT * kWSlots + ((size_t)(ph - nPh) * a.NT + nt) * kW1Slots
                                  : w0 + ((size_t)ph * a.NT + nt) * kWSlots;
        const int last = (second ? kW1Slots : kWSlots) - 1;
#pragma unroll
        for (int k = 0; k < kWLoads; ++k) {
            const int slot = tid + NT_ * k;
#ifdef IDH_ABL_NOLOAD
            const unsigned hsh = (unsigned)tid * 2246822519u + (unsigned)k * 3266489917u + (unsigned)ph * 668265263u + (unsigned)last;
            pw_[k] = (u32x4){0x3c003c00u ^ (hsh & 0x03ff03ffu), 0x3c003c00u ^ ((hsh >> 2) & 0x03ff03ffu), 0x3c003c00u ^ ((hsh >> 4) & 0x03ff03ffu),
                             0x3c003c00u ^ ((hsh >> 6) & 0x03ff03ffu)};
#else
            pw_[k] = src[slot < last ? slot : last];
#endif
        }
    };
#ifdef IDH_SPLIT_GLDS
    // weight panel of phase `ph` straight into LDS buffer `buf` (global_load_lds_dwordx4: no VGPRs, no ds_write);
    // 1 KiB pieces, piece i goes to wave i % WAVES
    auto dma_w = [&](int ph, int buf) {
        const u32x4 *w0 = reinterpret_cast<const u32x4 *>(s.w);
        const bool second = SRC2 && ph >= nPh;
        const u32x4 *src = second ? w0 + (size_t)nPh * a.NT * kWSlots + ((size_t)(ph - nPh) * a.NT + nt) * kW1Slots
                                  : w0 + ((size_t)ph * a.NT + nt) * kWSlots;
        const int pieces = (second ? kW1Slots : kWSlots) / 64;
#pragma unroll
        for (int i = 0; i < (kWSlots / 64 + WAVES - 1) / WAVES; ++i) {
            const int pc = wave + WAVES * i;
            if (pc < pieces)
                __builtin_amdgcn_global_load_lds(src + pc * 64 + lane,
                                                 (__attribute__((address_space(3))) void *)(sW + buf * kWSlots + pc * 64), 16, 0, 0);
        }
    };
#endif
    auto commit_w = [&](int buf) {
#if defined(IDH_ABL_NOCOMMIT) || defined(IDH_SPLIT_GLDS)
        return;
#endif
#pragma unroll
        for (int k = 0; k < kWLoads; ++k)
            if (k < kWLoads - 1 || wave < kWFullWaves) sW[buf * kWSlots + tid + NT_ * k] = pw_[k];
    };
    // taps (r, t0 .. t0 + NTAPS - 1) of the staged halo against weight taps 0 .. NTAPS - 1 of buffer `buf`
    auto compute = [&](int r, int buf, auto ntaps_c, int t0) {
        constexpr int NTAPS = decltype(ntaps_c)::value;
        const u32x4 *wb = sW + buf * kWSlots + kg * 64 + p;
        const u32x4 *hb = sH + kg * kPlane + (2 * G * wave + prow + r) * kHalo + px + t0;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            u32x4 A[2][NP], B[G][NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
                for (int j = 0; j < 2; ++j) A[j][pc] = wb[((t * NP + pc) * 2) * 64 + 32 * j];
#pragma unroll
                for (int g = 0; g < G; ++g) B[g][pc] = hb[pc * 2 * kPlane + 2 * g * kHalo + t];
            }
            // smallest terms first; independent accumulators between dependent MFMAs
            constexpr int kTerms = 3;
            constexpr int kPa3[3] = {1, 0, 0}, kPb3[3] = {0, 1, 0};
#pragma unroll
            for (int m = 0; m < kTerms; ++m)
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[j][kPa3[m]]),
                                                                               __builtin_bit_cast(f16x8, B[g][kPb3[m]]), acc[g][j], 0, 0, 0);
                    }
        }
    };

    // start of a chunk: (f16x3) fold the chunk maximum into the running scale, then split + stage the halo
    auto begin_chunk = [&](int c) {
        float mul = 1.f;
        if constexpr (MODE == MODE_F16X3) {
#ifndef IDH_ABL_NOCOMMIT
            publish_max(c & 1);
#endif
#ifndef IDH_ABL_NOBARRIER
            __syncthreads();  // every wave is done reading the previous chunk's halo; chunk maxima visible
#endif
            float bm = sMax[(c & 1) * 8];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) bm = fmaxf(bm, sMax[(c & 1) * 8 + w]);
            const int e = exponent_of(__builtin_amdgcn_readfirstlane(__float_as_uint(bm)));
            if (e > E) {  // workgroup-uniform: the running maximum grew -> shrink the accumulators to the new unit
                const int d = E - e;
                const float f = d < -126 ? 0.f : exp2_int(d);
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[g][j][i] *= f;
                E = e;
            }
            mul = exp2_int(14 - E);
        } else {
            if (c > 0) __syncthreads();  // every wave is done reading the previous chunk's halo
        }
        commit_halo(mul);
    };

    issue_halo(0);
#ifdef IDH_SPLIT_GLDS
    dma_w(0, 0);
#define IDH_NEXT_W(phn) dma_w((phn), (phn) & 1)
#else
    issue_w(0);
#define IDH_NEXT_W(phn) issue_w(phn)
#endif
    int ph = 0;
#pragma unroll 1
    for (int c = 0; c < nC; ++c) {
        begin_chunk(c);
#pragma unroll
        for (int r = 0; r < 3; ++r, ++ph) {
            commit_w(ph & 1);
#ifndef IDH_ABL_NOBARRIER
            __syncthreads();
#endif
            if (ph + 1 < nPhases || true) IDH_NEXT_W(ph + 1 < nPhases ? ph + 1 : ph);  // unconditional (re-reads the last panel at the end)
            if (r == 0) issue_halo(c + 1 < nChunks ? c + 1 : c);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch loads ahead of the MFMAs that hide them
            compute(r, ph & 1, std::integral_constant<int, 3>{}, 0);
        }
    }
    if constexpr (SRC2) {
#pragma unroll 1
        for (int c = nC; c < nChunks; ++c, ++ph) {  // 1x1 source: the centre tap of each chunk
            begin_chunk(c);
            commit_w(ph & 1);
            __syncthreads();
            IDH_NEXT_W(ph + 1 < nPhases ? ph + 1 : ph);
            issue_halo(c + 1 < nChunks ? c + 1 : c);
            __builtin_amdgcn_sched_barrier(0);
            compute(1, ph & 1, std::integral_constant<int, 1>{}, 1);
        }
    }

    // epilogue: C/D of 32x32: column = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (channel)
    const float *wscale =
        reinterpret_cast<const float *>(reinterpret_cast<const u32x4 *>(s.w) + (size_t)nPh * a.NT * kWSlots + (size_t)nC2 * a.NT * kW1Slots);
    const float sx = MODE == MODE_F16X3 ? exp2_int(E - 14 < -126 ? -126 : E - 14) : 1.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int oy = y0 + 2 * G * wave + 2 * g + prow, ox = x0 + px;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const size_t m = ((size_t)n * a.Ho + oy) * a.Wo + ox;
        float *o = a.out + m * a.out_cs;
#ifdef IDH_ABL_NORES
        const float *rp = nullptr;
#else
        const float *rp = a.res ? a.res + m * a.res_cs : nullptr;
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int co = n0 + 32 * j + 8 * gg + 4 * kg;
                f32x4 v = {acc[g][j][4 * gg], acc[g][j][4 * gg + 1], acc[g][j][4 * gg + 2], acc[g][j][4 * gg + 3]};
                if constexpr (MODE == MODE_F16X3) v = (v * sx) * *reinterpret_cast<const f32x4 *>(wscale + co);
                if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + co);
                if (rp) v += *reinterpret_cast<const f32x4 *>(rp + co);
                if (a.act != IDH_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], a.act, a.slope);
                }
#ifdef IDH_ABL_NOSTORE
                if (v[0] == 123.456f)
#endif
                *reinterpret_cast<f32x4 *>(o + co) = v;
            }
    }
}

// per output channel: exponent of max |w| (MODE_F16X3 weight scale); one workgroup per channel
__global__ __launch_bounds__(256) void weight_exponent_k(const float *__restrict__ w, const float *__restrict__ w2, int *__restrict__ wexp,
                                                         float *__restrict__ wscale, int per_cout, int per_cout2) {
    __shared__ float sm[256];
    const int co = blockIdx.x;
    float m = 0.f;
    for (int i = threadIdx.x; i < per_cout + per_cout2; i += 256) {
        const float v = i < per_cout ? w[(size_t)co * per_cout + i] : w2[(size_t)co * per_cout2 + (i - per_cout)];
        m = fmaxf(m, fabsf(v));
        if ((__float_as_uint(v) & 0x7F800000u) == 0x7F800000u) m = __uint_as_float(0x7F800000u);
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int e = exponent_of(__float_as_uint(sm[0]));
        wexp[co] = e;
        wscale[co] = exp2_int(e - 14 < -126 ? -126 : e - 14);  // epilogue factor: undoes w * 2^(14-e)
    }
}

// OIHW fp32 -> [chunk][tap row][co tile][tap in row][piece][kg][co 64][8 x 16 bit], zero padded.
// KS = 3: 3x3 weights; KS = 1: 1x1 weights -> [chunk][co tile][piece][kg][co 64][8] (one "tap row" of one tap)
template <int MODE, int KS>
__global__ __launch_bounds__(256) void pack_split_weight_k(const float *__restrict__ w, u32x4 *__restrict__ dst,
                                                           const int *__restrict__ wexp, int Cout, int Cin, int nC, int NT) {
    constexpr int NP = pieces_of(MODE);
    constexpr int kTapsRow = KS == 3 ? 3 : 1, kRowsK = KS == 3 ? 3 : 1;
    constexpr int kWSlots = wslots_of(MODE) / 3 * kTapsRow;
    const long long total = (long long)nC * kRowsK * NT * kTapsRow * 2 * 64;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
        long long rr = i;
        const int co = (int)(rr % 64); rr /= 64;
        const int kg = (int)(rr % 2); rr /= 2;
        const int t = (int)(rr % kTapsRow); rr /= kTapsRow;
        const int nt = (int)(rr % NT); rr /= NT;
        const int r = (int)(rr % kRowsK);
        const int c = (int)(rr / kRowsK);
        const int cout = 64 * nt + co, tap = 3 * r + t;
        unsigned h[8][NP];
        float mul = 1.f;
        if constexpr (MODE == MODE_F16X3) mul = exp2_int(14 - wexp[cout]);
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * c + 8 * kg + e;
            const float v = (ci < Cin && cout < Cout) ? w[((size_t)cout * Cin + ci) * (KS * KS) + tap] : 0.f;
            {
                _Float16 a0, a1;
                split2_f16(v * mul, a0, a1);
                h[e][0] = __builtin_bit_cast(unsigned short, a0);
                h[e][1] = __builtin_bit_cast(unsigned short, a1);
            }
        }
        const size_t base = ((size_t)(c * kRowsK + r) * NT + nt) * kWSlots;
        for (int pc = 0; pc < NP; ++pc) {
            u32x4 v;
            v = (u32x4){h[0][pc] | (h[1][pc] << 16), h[2][pc] | (h[3][pc] << 16), h[4][pc] | (h[5][pc] << 16), h[6][pc] | (h[7][pc] << 16)};
            dst[base + ((t * NP + pc) * 2 + kg) * 64 + co] = v;
        }
    }
}

inline int ceil16i(int v) { return (v + 15) & ~15; }
inline size_t panel_bytes(int mode, int Cout, int Cin) { return (size_t)(ceil16i(Cin) / 16) * 3 * (Cout / 64) * wslots_of(mode) * 16; }
inline size_t panel1_bytes(int mode, int Cout, int Cin2) { return Cin2 > 0 ? (size_t)(ceil16i(Cin2) / 16) * (Cout / 64) * (wslots_of(mode) / 3) * 16 : 0; }

template <int WAVES, int G, int MODE, bool SRC2>
int launch_one_src(const ConvArgs &a, int N, hipStream_t st);

template <int WAVES, int G, int MODE>
int launch_one(const ConvArgs &a, int N, hipStream_t st) {
    return a.s[1].in ? launch_one_src<WAVES, G, MODE, true>(a, N, st) : launch_one_src<WAVES, G, MODE, false>(a, N, st);
}

template <int WAVES, int G, int MODE, bool SRC2>
int launch_one_src(const ConvArgs &a, int N, hipStream_t st) {
    constexpr int kRows = 2 * G * WAVES;
    static IdhDeviceOnce attr_done;
    if (attr_done.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_split_k<WAVES, G, MODE, SRC2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_of(MODE, kRows)) != hipSuccess)
            return IDH_ELAUNCH;
        attr_done.mark();
    }
    const int tiles_x = (a.Wo + kSplitTile - 1) / kSplitTile, tiles_y = (a.Ho + kRows - 1) / kRows;
    const long long blocks = (long long)N * tiles_x * tiles_y * a.NT;
    if (blocks >= (1ll << 31)) return IDH_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_split_k<WAVES, G, MODE, SRC2>), dim3((unsigned)blocks), dim3(64 * WAVES), lds_bytes_of(MODE, kRows), st,
                       a, tiles_x, tiles_y);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

}  // namespace

namespace idh_conv {

int launch_conv_split(const ConvArgs &a, int N, int mode, int rows, hipStream_t st) {
    if (mode != IDH_SPLIT_F16X3) return IDH_EINVAL;
    if (rows == 8) return launch_one<4, 1, MODE_F16X3>(a, N, st);
    // measured on MI355X (tools/perf_split.py): 5-9 % faster with 4 waves x 2 tile groups (166 VGPRs, 44.5 KiB ->
    // 3 workgroups per CU) than with 8 waves
    return launch_one<4, 2, MODE_F16X3>(a, N, st);
}

}  // namespace idh_conv

extern "C" size_t idh_packed_split_weight_bytes(int Cout, int Cin, int Cin_1x1, int mode) {
    if (Cout <= 0 || Cin <= 0 || Cin_1x1 < 0 || Cout % 64 || mode != IDH_SPLIT_F16X3) return 0;
    const int m = MODE_F16X3;
    return panel_bytes(m, Cout, Cin) + panel1_bytes(m, Cout, Cin_1x1) + (size_t)Cout * 8;  // + per-channel scale floats + exponents
}

extern "C" int idh_pack_conv_weight_split(const float *w, const float *w_1x1, void *dst, int Cout, int Cin, int Cin_1x1, int mode,
                                          void *stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0 || Cin_1x1 < 0 || (Cin_1x1 > 0) != (w_1x1 != nullptr) ||
        mode != IDH_SPLIT_F16X3)
        return IDH_EINVAL;
    if (Cout % 64) return IDH_EUNSUPPORTED;
    const int m = MODE_F16X3;
    const int nC = ceil16i(Cin) / 16, NT = Cout / 64, nC2 = Cin_1x1 > 0 ? ceil16i(Cin_1x1) / 16 : 0;
    char *d8 = static_cast<char *>(dst);
    u32x4 *d3 = reinterpret_cast<u32x4 *>(d8), *d1 = reinterpret_cast<u32x4 *>(d8 + panel_bytes(m, Cout, Cin));
    float *wscale = reinterpret_cast<float *>(d8 + panel_bytes(m, Cout, Cin) + panel1_bytes(m, Cout, Cin_1x1));
    int *wexp = reinterpret_cast<int *>(wscale + Cout);
    hipStream_t st = idh_stream(stream);
    hipLaunchKernelGGL(weight_exponent_k, dim3(Cout), dim3(256), 0, st, w, w_1x1, wexp, wscale, Cin * 9, Cin_1x1);
    IDH_CHECK_LAUNCH();
    auto grid_of = [](long long total) { int g = idh_cdiv(total, 256); return g > 4096 ? 4096 : g; };
    const int g3 = grid_of((long long)nC * 3 * NT * 3 * 2 * 64);
    hipLaunchKernelGGL((pack_split_weight_k<MODE_F16X3, 3>), dim3(g3), dim3(256), 0, st, w, d3, wexp, Cout, Cin, nC, NT);
    IDH_CHECK_LAUNCH();
    if (nC2 > 0) {
        const int g1 = grid_of((long long)nC2 * NT * 2 * 64);
        hipLaunchKernelGGL((pack_split_weight_k<MODE_F16X3, 1>), dim3(g1), dim3(256), 0, st, w_1x1, d1, wexp, Cout, Cin_1x1, nC2, NT);
        IDH_CHECK_LAUNCH();
    }
    return IDH_OK;
}
