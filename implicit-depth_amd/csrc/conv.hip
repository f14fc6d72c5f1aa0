// Implicit-GEMM 2D convolution on the fp32 matrix cores of gfx950 + the small layout /
// upsample / reduce kernels that glue a network pass together (see include/idh_ops.h).
//
// Replaces every nn.Conv2d the reference runs inside BasicBlock (modules/layers.py:59-95),
// i.e. all of CVEncoder / BDDecoderPP / DepthDecoderPP (modules/networks.py:20-215).
//
// GEMM view:  out[m, co] = sum_{tap, ci} in[pix(m) + tap, ci] * w[tap, ci, co]
//   M = N*Ho*Wo output pixels, N = Cout, K = ks*ks*Cin (x2 sources for a fused projection).
// Numerics: v_mfma_f32_16x16x4_f32 — exact fp32 products, fp32 accumulate (bit-equal to an
// fmaf chain, cdna guide §3); gfx950 has no TF32/xf32 path, and bf16 would break the 1e-4
// parity bar, so fp32 MFMA (157 TFLOP/s dense) is the roofline of this kernel.
//
// Design ("direct-fragment" implicit GEMM, no LDS, no barriers):
//   * each 64-lane wave owns a (16*TM pixels) x (16*TN channels) output tile and keeps it in
//     TM*TN MFMA accumulators;
//   * MFMA 16x16x4 wants A[m=lane&15][k=lane>>4]: lane quarter h loads ONE float4 holding
//     channels 16c+4h..16c+4h+3 of its pixel (NHWC: contiguous), which feeds four MFMAs — MFMA
//     kk contracts channels {16c+kk, 16c+4+kk, 16c+8+kk, 16c+12+kk}; any K order is valid as long
//     as A and B agree.  A wave-load therefore reads 16 pixels x 64 contiguous bytes;
//   * weights are pre-packed [tap][ci/4][co][ci%4] so the matching B fragment is one float4 per
//     lane too (16 lanes x 16 B contiguous);
//   * fp32 MFMA is slow per byte (64 flop/clk/SIMD), so operand traffic is ~1 float4 per
//     256 MFMA-cycles per wave: it streams from L1/L2 without LDS staging; the K loop is
//     software-pipelined (loads of step s+1 in flight under the MFMAs of step s, ping-pong
//     register sets) and the waves of a workgroup are fully independent (no barriers);
//   * zero padding = out-of-image lanes read a zero page (branch-free K loop); torch.cat = channel-strided in/out pointers; the
//     residual projection of a BasicBlock is a second K-source of the same launch; bias +
//     residual + LeakyReLU are applied on the accumulators before the single store;
//   * layers with too few tiles to fill 256 CUs (12x16 / 24x32 maps) split K over `split_k`
//     waves that write raw partials; a reduce kernel applies the epilogue (deterministic, no atomics).
#include <algorithm>
#include "conv_args.h"
#include "../../include/idh_ops.h"

using namespace idh_conv;

namespace {

// Zero page: out-of-image taps (zero padding) read from here instead of being predicated,
// so the K loop is branch-free.  Must cover the widest Cin_pad (host-checked).
__device__ float g_zero_page[kZeroFloats];

// idh_count_launches() replays idh_run_ops' launch decisions without launching anything
thread_local bool t_dry_run = false;
thread_local int t_launches = 0;
#undef IDH_CHECK_LAUNCH
#define IDH_CHECK_LAUNCH()                                              \
    do {                                                                \
        if (!t_dry_run && hipGetLastError() != hipSuccess) return IDH_ELAUNCH; \
    } while (0)
#define IDH_LAUNCH(...)                         \
    do {                                        \
        if (t_dry_run) ++t_launches;            \
        else hipLaunchKernelGGL(__VA_ARGS__);   \
    } while (0)

template <int TM, int TN>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &a, unsigned blk_in, unsigned nblk) {
    const int lane = threadIdx.x & 63;
    // readfirstlane: tell the compiler the wave index is uniform, so the tile / split / step
    // bookkeeping below lives in SGPRs and the K loop uses scalar branches
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, h = lane >> 4;

    const unsigned blk = idh_xcd_remap(blk_in, nblk);
    const long long wg = (long long)blk * 4 + wave;
    const long long total = (long long)a.MT * a.NT * a.S;
    if (wg >= total) return;
    const int nt = (int)(wg % a.NT);
    const int mt = (int)((wg / a.NT) % a.MT);
    const int sp = (int)(wg / ((long long)a.NT * a.MT));

    const int m_base = mt * 16 * TM;
    const int n_base = nt * 16 * TN;

    // this wave's slice of the flattened (source, tap, channel-block) step list
    const int t0 = (int)((long long)a.steps_total * sp / a.S);
    const int t1 = (int)((long long)a.steps_total * (sp + 1) / a.S);

    // per-lane pixel coordinates for the A rows this lane loads (row ln of each sub-tile)
    int oy[TM], ox[TM], nb[TM];
    bool mv[TM];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = m_base + 16 * i + ln;
        mv[i] = m < a.M;
        m = mv[i] ? m : a.M - 1;
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        oy[i] = r / a.Wo;
        ox[i] = r - oy[i] * a.Wo;
        nb[i] = n;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- step iterator: (source si, tap, channel block cb), all wave-uniform ----------------
    const int steps0 = a.s[0].ks * a.s[0].ks * a.s[0].cblocks;
    int si = (t0 >= steps0) ? 1 : 0;
    ConvSrc s = a.s[si];
    int local = t0 - (si ? steps0 : 0);
    int tap = local / s.cblocks;
    int cb = local - tap * s.cblocks;
    const size_t w_cq_stride = (size_t)a.Cout_pad * 4;  // floats between consecutive ci/4 groups
    const float *ap[TM];
    const float *wp;
    auto set_tap = [&]() {
        const int pad = s.ks >> 1;
        const int dy = tap / s.ks, dx = tap - dy * s.ks;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int iy = oy[i] * s.stride + dy - pad;
            int ix = ox[i] * s.stride + dx - pad;
            // branch-free validity: unsigned compare folds the two-sided range test
            const bool inb = ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            const bool ok = mv[i] & (inb | (s.pad_mode == IDH_PAD_REPLICATE));
            iy = min(max(iy, 0), s.H - 1);
            ix = min(max(ix, 0), s.W - 1);
            const float *real = s.in + ((size_t)(nb[i] * s.H + iy) * s.W + ix) * s.cs + 4 * h;
            ap[i] = ok ? real : (g_zero_page + 4 * h);
        }
        wp = s.w + ((size_t)tap * s.cblocks * 4 + h) * w_cq_stride + (size_t)(n_base + ln) * 4;
    };
    set_tap();
    int remaining = t1 - t0;  // real steps not yet loaded
    // Loads are UNCONDITIONAL (so the compiler can use counted s_waitcnt vmcnt(N) and keep the
    // prefetch of step s+1 in flight under the MFMAs of step s).  Past the last real step the
    // iterator parks on a "null step": activations come from the zero page, so the extra MFMAs of
    // the ping-pong tail add exactly 0.
    auto load = [&](f32x4(&A)[TM], f32x4(&Bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) A[i] = *reinterpret_cast<const f32x4 *>(ap[i] + 16 * cb);
#pragma unroll
        for (int j = 0; j < TN; ++j) Bf[j] = *reinterpret_cast<const f32x4 *>(wp + (size_t)cb * 4 * w_cq_stride + 64 * j);
        --remaining;
        if (remaining > 0) {  // advance to the next real step (uniform control flow, VALU only)
            if (++cb == s.cblocks) {
                cb = 0;
                if (++tap == s.ks * s.ks) {
                    tap = 0;
                    si = 1;
                    s = a.s[1];
                }
                set_tap();
            }
        } else {
            cb = 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) ap[i] = g_zero_page + 4 * h;
        }
    };
    auto mma = [&](const f32x4(&A)[TM], const f32x4(&Bf)[TN]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bf[j][kk], A[i][kk], acc[i][j], 0, 0, 0);
    };

    // ---- software-pipelined K loop: the loads of step s+1 are in flight under the MFMAs of s ----
    f32x4 A0[TM], B0[TN], A1[TM], B1[TN];
    const int nsteps = t1 - t0;
    load(A0, B0);
#pragma unroll 1
    for (int st = 0; st < nsteps; st += 2) {
        load(A1, B1);
        mma(A0, B0);
        load(A0, B0);
        mma(A1, B1);
    }

    // ---- epilogue.  The MFMA was issued as D^T = W . X^T (weights as the A operand), so in the
    // 16x16x4 C/D layout (col = lane&15, row = 4*(lane>>4) + reg) a lane holds 4 CONSECUTIVE
    // output channels of ONE pixel — the pixel it loaded activations for — and the NHWC store is a
    // single 16-byte vector per accumulator.
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (!mv[i]) continue;
        const size_t m = (size_t)(m_base + 16 * i + ln);
        if (a.S > 1) {
            float *o = a.ws + ((size_t)sp * a.M + m) * a.Cout_pad + n_base + 4 * h;
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4 *>(o + 16 * j) = acc[i][j];
            continue;
        }
        float *o = a.out + m * a.out_cs;
        const float *rp = a.res ? a.res + m * a.res_cs : nullptr;
        act_dispatch(a.act, a.slope, [&](auto fn) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n_base + 16 * j + 4 * h;
                if (co >= a.Cout) continue;  // Cout % 4 == 0 (host-checked)
                f32x4 v = acc[i][j];
                if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + co);
                if (rp) v += *reinterpret_cast<const f32x4 *>(rp + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fn(v[r]);
                *reinterpret_cast<f32x4 *>(o + co) = v;
            }
        });
    }
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void conv_mfma_k(const ConvArgs a) {
    conv_mfma_body<TM, TN>(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------
// LDS-staged variant for the layers that carry ~95% of the flops: 3x3, stride 1, zero padding,
// Cout % 64 == 0, optionally fused with a 1x1 projection of a second tensor.
//
// Why: the direct-fragment kernel above re-reads every input pixel 9x (once per tap) and every
// weight once per wave through the vector L1, and that path sustains only ~30 B/clk/CU for these
// 16-byte-per-lane gathers — the MFMA pipe idles ~50% (rocprof: profiles/r01).  Here a 256-thread
// workgroup owns an 8x16-pixel x 64-channel output tile and, per 16-channel K chunk, stages
//   A: the 10x18-pixel halo of the input        (11.25 KiB, read from HBM/L2 ONCE for 9 taps)
//   B: the 9-tap x 16-ci x 64-co weight panel   (36 KiB, shared by the 4 waves)
// in LDS; each wave then issues 9 x 32 MFMAs (2 tile rows x 64 channels) fed by conflict-free
// ds_read_b128 fragment reads.  The next chunk's global loads are issued before the MFMA phase
// and parked in registers (classic register prefetch), so HBM/L2 latency hides under ~9k cycles
// of matrix work; 48 KiB of LDS per workgroup lets 3 workgroups share a CU and cover each
// other's barrier / staging phases.
//   A in LDS: [hy 10][q 4][hx 18] float4  (q = channel quad)  -> a tap shift is a pure offset and a
//             fragment read touches 16 consecutive float4 per quarter-wave: no bank conflicts
//   B in LDS: [tap 9][q 4][co 64] float4
// RW = tile rows per wave: RW = 2 -> 8x16 tile (best operand reuse), RW = 1 -> 4x16 tile (twice the
// workgroups: used when an 8-row grid cannot fill 256 CUs x 3 resident workgroups).
constexpr int kLT_W = 16;
constexpr int kHaloW = kLT_W + 2;

struct LdsConvArgs {
    ConvArgs c;
    int tiles_x, tiles_y;  // spatial tiles per image
};

// UP: some K chunks come from x2-upsampled low-resolution maps (idh_conv_src.up_*): the loader fetches the four
// low-resolution neighbours of every halo pixel and the commit blends them with exactly upsample2_k's expression
// (bit-identical to materialising the upsampled tensor first), so F.interpolate + torch.cat cost no launch and no
// HBM round trip.  Separate instantiation: the 4x register prefetch does not touch the plain kernel's occupancy.
// NJ = 16-channel output sub-tiles per workgroup: 4 (64 channels, the default), 2 or 1 — narrow layers (the matching
// encoder's 128 -> 16 conv) and small maps at small batch (twice / four times the workgroups) use the narrower tiles;
// the staged halo is then amortised over fewer MFMAs but still feeds all 9 taps from one HBM/L2 read.
// LDS image sizes (float4): halo of one 16-channel chunk / its 9-tap weight panel (36 KiB at NJ = 4)
constexpr int lds_a_slots(int RW) { return (4 * RW + 2) * 4 * kHaloW; }  // 720 (RW=2) / 432 (RW=1)
// stride-2 3x3 second source (BasicBlock's strided projection): the halo of a (4*RW) x 16 output tile is (8*RW + 1) x 33
// input pixels, kept de-interleaved by column parity — [hy][q][parity][17] float4 — so that a tap's fragment read is
// again 16 consecutive float4 per quarter-wave
constexpr int kHalo2W = 2 * kLT_W + 1, kHalo2Wh = kLT_W + 1;
constexpr int lds_a2_slots(int RW) { return (8 * RW + 1) * 4 * 2 * kHalo2Wh; }  // 2312 (RW=2) / 1224 (RW=1)
constexpr int lds_a_slots(int RW, bool S2) { return S2 ? (lds_a2_slots(RW) > lds_a_slots(RW) ? lds_a2_slots(RW) : lds_a_slots(RW)) : lds_a_slots(RW); }
constexpr int lds_b_slots(int NJ) { return 9 * 4 * 16 * NJ; }

// 1x1 second source (BasicBlock's fused projection): its 16-channel chunks are staged SEVERAL per barrier pair.  One chunk is only the tile's
// centre pixels + a 4 x kN panel and feeds 4 * RW * NJ MFMAs per wave - a ninth of a 3x3 chunk - so with one chunk per round the two
// barriers and the commit dominate (a 640-channel projection: 40 rounds).  Chunk 0 of a round keeps the halo image's place in sA, the
// panels of all chunks and the centre images of chunks 1.. share sB (idle between the 3x3 chunks and here): as many chunks as fit, at most 4.
constexpr int lds_c1_slots(int RW) { return 4 * RW * 4 * kHaloW; }  // centre image of one 1x1 chunk, rows padded like the halo's
constexpr int lds_g1(int RW, int NJ) {
    int g = 1;
    while (g < 4 && (g + 1) * 4 * 16 * NJ + g * lds_c1_slots(RW) <= lds_b_slots(NJ)) ++g;
    return g;
}

// sA / sB: the workgroup's LDS images (declared by the kernel so that a kernel hosting several instantiations
// of this body — level_k — allocates them once)
// NORM: source 0 is read through a per-(image, channel) normalisation + activation (idh_conv_src.norm): the statistics
// of the chunk's channels are prefetched with the halo and applied when the halo is committed to LDS.
// S2: source 1 is a 3x3 stride-2 convolution of the block input (the projection of a stride-2 BasicBlock,
// layers.py:71-74) instead of a 1x1: its (8*RW+1) x 33 halo is staged like source 0's, de-interleaved by column parity.
template <int RW, bool UP, int NJ, bool NORM = false, bool S2 = false>
__device__ __forceinline__ void conv3x3_lds_body(const LdsConvArgs &la, unsigned blk_in, unsigned nblk, f32x4 *__restrict__ sA,
                                                 f32x4 *__restrict__ sB) {
    constexpr int kN = 16 * NJ;                  // output channels per workgroup
    constexpr int kBSlots3 = lds_b_slots(NJ);    // weight panel of one 16-channel chunk, float4
    constexpr int kBLoads = (kBSlots3 + 255) / 256;
    constexpr int kLT_H = 4 * RW;
    constexpr int kASlots = lds_a_slots(RW);  // (kLT_H + 2)-row halo
    constexpr int kALoads = (kASlots + 255) / 256;
    constexpr int kCLoads = (kLT_H * kLT_W * 4) / 256;  // centre pixels for the 1x1 source
    const ConvArgs &a = la.c;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, h = lane >> 4;


    // block -> (split, image, tile_y, tile_x, channel tile); channel tile fastest so the blocks
    // sharing an input tile are neighbours (same XCD after the remap -> L2 hits on the halo)
    unsigned blk = idh_xcd_remap(blk_in, nblk);
    const int nt = blk % a.NT; blk /= a.NT;
    const int tx = blk % la.tiles_x; blk /= la.tiles_x;
    const int ty = blk % la.tiles_y; blk /= la.tiles_y;
    const int N_img = a.M / (a.Ho * a.Wo);
    const int n = blk % N_img;
    const int sp = blk / N_img;
    const int y0 = ty * kLT_H, x0 = tx * kLT_W;
    const int n0 = nt * kN;
    const bool replicate = a.s[0].pad_mode == IDH_PAD_REPLICATE;  // nn.Conv2d(padding_mode="replicate"): clamp instead of the zero page

    // the accumulators start from the bias (when this workgroup writes final values, i.e. no split-K): the 16-byte bias
    // loads replace the zero moves and the epilogue's adds — fp32 MFMA and VALU share the SIMD's datapath,
    // every vector instruction outside the K loop is paid in MFMA time
    f32x4 acc[RW][NJ];
    const bool bias_first = a.bias != nullptr && a.S == 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 b4 = bias_first ? *reinterpret_cast<const f32x4 *>(a.bias + n0 + 16 * j + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RW; ++i) acc[i][j] = b4;
    }

    // chunk list = [source-0 chunks][source-1 chunks]; this block's split owns [t0,t1)
    const int nc0 = a.s[0].cblocks;
    const int nc1 = a.s[1].in ? a.s[1].cblocks : 0;
    // split boundaries in COST units, not chunks: a 3x3 chunk is 9 taps of MFMAs + one staging round, a 1x1 chunk one tap and (batched,
    // lds_g1) a fraction of a round - about a ninth (measured flat between 4 and 9, tools/r05/job_c3.sh).  With equal weights the first split of a
    // BasicBlock's conv2 + projection got all 3x3 chunks and the last ones only 1x1 chunks: the launch took as long as the unsplit 3x3 part.
    constexpr int kW3 = 9;
    const int w1 = S2 ? kW3 : 1;
    const int T = kW3 * nc0 + w1 * nc1;
    auto split_bound = [&](int s_) {
        const int b = (int)((long long)T * s_ / a.S);
        return b <= kW3 * nc0 ? (b + kW3 - 1) / kW3 : nc0 + (b - kW3 * nc0 + w1 - 1) / w1;
    };
    const int t0 = split_bound(sp), t1 = split_bound(sp + 1);

    // ---- staging: global -> registers (prefetch) -> LDS -----------------------------------
    // A slots are enumerated (hy, hx, q) with q fastest so that 4 consecutive lanes read the 64
    // contiguous bytes of one pixel; the LDS image is [hy][q][hx].
    constexpr int kA2Slots = (8 * RW + 1) * kHalo2W * 4;  // real halo slots of the stride-2 source (hy, hx, q)
    constexpr int kA2Loads = (kA2Slots + 255) / 256;
    f32x4 pa[S2 ? kA2Loads : (UP ? 4 : 1) * kALoads], pb[kBLoads];
    f32x4 pmean[NORM ? kALoads : 1], prstd[NORM ? kALoads : 1];
    unsigned in_image = 0;  // NORM: bit k = halo slot k is a real pixel (a zero-padded tap stays 0 after normalisation)
    // Plain (not upsample-fused) source 0: buffer loads.  The byte offset of every halo / weight slot of this lane is
    // computed ONCE (one VGPR each); the chunk only moves the scalar offset, and padding is the descriptor's range check
    // (an out-of-image slot gets an offset past the image and reads 0).  The K loop then spends no vector instructions
    // on addresses — they would be paid in MFMA time.
    constexpr int kOob = 0x7fffffff;
    int voffA[UP ? 1 : kALoads], voffB[UP ? 1 : kBLoads];
    __amdgpu_buffer_rsrc_t rsA, rsW, rsN;
    if constexpr (!UP) {
        const ConvSrc &s = a.s[0];
        if constexpr (NORM) rsN = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.norm + (size_t)n * 2 * s.Cin), 0, 2 * s.Cin * 4, 0x00020000);
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)n * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
        rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, 9 * s.cblocks * 4 * a.Cout_pad * 16, 0x00020000);
#pragma unroll
        for (int k = 0; k < kALoads; ++k) {
            const int slot = tid + 256 * k;
            const int q = slot & 3, pix = slot >> 2;
            const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
            int iy = y0 + hy - 1, ix = x0 + hx - 1;
            bool ok = (slot < kASlots) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            if (replicate) { iy = min(max(iy, 0), s.H - 1); ix = min(max(ix, 0), s.W - 1); ok = slot < kASlots; }
            voffA[k] = ok ? ((iy * s.W + ix) * s.cs + 4 * q) * 4 : kOob;
            if (NORM) in_image = ok ? (in_image | (1u << k)) : in_image;
        }
#pragma unroll
        for (int k = 0; k < kBLoads; ++k) {  // slot = ((tap * 4 + q) * kN + co); NJ = 4: tap k, q = tid >> 6, co = tid & 63
            const int slot = tid + 256 * k;
            const int co = slot % kN, tq = slot / kN;
            const int tap = tq >> 2, q = tq & 3;
            voffB[k] = (slot < kBSlots3) ? ((tap * s.cblocks * 4 + q) * a.Cout_pad + co + n0) * 16 : kOob;
        }
    }
    // low-resolution neighbours + weights of hi-res pixel (iy, ix) under x2 bilinear, align_corners=False
    auto up_taps = [&](const ConvSrc &s, int c, int iy, int ix, int q, bool ok, const float *(&tp)[4]) {
        const int rel = 16 * c - s.up_c0;
        const int seg = rel >= s.up_C ? 1 : 0;
        const int Hl = s.H >> 1, Wl = s.W >> 1, ucs = s.up_cs[seg];
        const int yl = iy >> 1, xl = ix >> 1;
        int y0, y1, x0, x1;
        if (iy & 1) { y0 = yl; y1 = min(yl + 1, Hl - 1); } else { y0 = max(yl - 1, 0); y1 = yl; }
        if (ix & 1) { x0 = xl; x1 = min(xl + 1, Wl - 1); } else { x0 = max(xl - 1, 0); x1 = xl; }
        const float *b = s.up_in[seg] + (size_t)n * Hl * Wl * ucs + (rel - seg * s.up_C) + 4 * q;
        tp[0] = ok ? b + ((size_t)y0 * Wl + x0) * ucs : g_zero_page;
        tp[1] = ok ? b + ((size_t)y0 * Wl + x1) * ucs : g_zero_page;
        tp[2] = ok ? b + ((size_t)y1 * Wl + x0) * ucs : g_zero_page;
        tp[3] = ok ? b + ((size_t)y1 * Wl + x1) * ucs : g_zero_page;
    };
    auto up_blend = [&](int iy, int ix, const f32x4 &p00, const f32x4 &p01, const f32x4 &p10, const f32x4 &p11) {
        const float hy0 = (iy & 1) ? 0.75f : 0.25f, hy1 = (iy & 1) ? 0.25f : 0.75f;
        const float wx0 = (ix & 1) ? 0.75f : 0.25f, wx1 = (ix & 1) ? 0.25f : 0.75f;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hy0 * (wx0 * p00[e] + wx1 * p01[e]) + hy1 * (wx0 * p10[e] + wx1 * p11[e]);  // == upsample2_k
        return o;
    };
    auto issue3 = [&](int c) {  // 3x3 source 0, chunk c
        const ConvSrc &s = a.s[0];
        if (UP && s.up_in[0] != nullptr && 16 * c >= s.up_c0) {
#pragma unroll
            for (int k = 0; k < kALoads; ++k) {
                const int slot = tid + 256 * k;
                const int q = slot & 3, pix = slot >> 2;
                const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
                const int iy = y0 + hy - 1, ix = x0 + hx - 1;
                const bool ok = (slot < kASlots) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
                const float *tp[4];
                up_taps(s, c, iy, ix, q, ok, tp);
#pragma unroll
                for (int e = 0; e < 4; ++e) pa[(UP ? 4 : 1) * k + (UP ? e : 0)] = *reinterpret_cast<const f32x4 *>(tp[e]);
            }
        } else if constexpr (!UP) {
#pragma unroll
            for (int k = 0; k < kALoads; ++k) {
                pa[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[k], 64 * c, 0));
                if (NORM) {  // statistics of this slot's channel quad: offset 16 q bytes (tid & 3 == slot & 3), chunk and mean / rstd in the scalar offset
                    pmean[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsN, (tid & 3) * 16, 64 * c, 0));
                    prstd[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsN, (tid & 3) * 16, 64 * c + s.Cin * 4, 0));
                }
            }
#pragma unroll
            for (int k = 0; k < kBLoads; ++k)
                pb[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffB[k], c * 4 * a.Cout_pad * 16, 0));
            return;
        } else {
#pragma unroll
        for (int k = 0; k < kALoads; ++k) {
            const int slot = tid + 256 * k;
            const int q = slot & 3, pix = slot >> 2;
            const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
            int iy = y0 + hy - 1, ix = x0 + hx - 1;
            bool ok = (slot < kASlots) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            if (replicate) { iy = min(max(iy, 0), s.H - 1); ix = min(max(ix, 0), s.W - 1); ok = slot < kASlots; }
            const float *p = ok ? s.in + ((size_t)(n * s.H + iy) * s.W + ix) * s.cs + 16 * c + 4 * q : g_zero_page;
            pa[(UP ? 4 : 1) * k] = *reinterpret_cast<const f32x4 *>(p);
        }
        }
        const float *wb = s.w + ((size_t)(4 * c) * a.Cout_pad + n0) * 4;
#pragma unroll
        for (int k = 0; k < kBLoads; ++k) {  // slot = ((tap * 4 + q) * kN + co); NJ = 4: tap k, q = tid >> 6, co = tid & 63
            const int slot = tid + 256 * k;
            const int co = slot % kN, tq = slot / kN;
            const int tap = tq >> 2, q = tq & 3;
            const float *p = (slot < kBSlots3) ? wb + ((size_t)(tap * s.cblocks * 4 + q) * a.Cout_pad + co) * 4 : g_zero_page;
            pb[k] = *reinterpret_cast<const f32x4 *>(p);
        }
    };
    auto commit3 = [&](int c) {
        const bool up = UP && a.s[0].up_in[0] != nullptr && 16 * c >= a.s[0].up_c0;
#pragma unroll
        for (int k = 0; k < kALoads; ++k) {
            const int slot = tid + 256 * k;
            const int q = slot & 3, pix = slot >> 2;
            const int hy = pix / kHaloW, hx = pix - hy * kHaloW;
            f32x4 v = pa[(UP ? 4 : 1) * k];
            if (UP && up) v = up_blend(y0 + hy - 1, x0 + hx - 1, pa[4 * k], pa[4 * k + 1], pa[4 * k + 2], pa[4 * k + 3]);
            if (NORM) {  // == instnorm_apply_k
                const bool real = (in_image >> k) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = real ? act_apply((v[e] - pmean[k][e]) * prstd[k][e], a.s[0].norm_act, a.s[0].norm_slope) : 0.f;
            }
            if (slot < kASlots) sA[(hy * 4 + q) * kHaloW + hx] = v;
        }
#pragma unroll
        for (int k = 0; k < kBLoads; ++k)
            if (tid + 256 * k < kBSlots3) sB[k * 256 + tid] = pb[k];
    };
    auto compute3 = [&]() {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            f32x4 A[RW], Bf[NJ];
#pragma unroll
            for (int i = 0; i < RW; ++i) A[i] = sA[((RW * wave + i + dy) * 4 + h) * kHaloW + ln + dx];
#pragma unroll
            for (int j = 0; j < NJ; ++j) Bf[j] = sB[(tap * 4 + h) * kN + 16 * j + ln];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < RW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bf[j][kk], A[i][kk], acc[i][j], 0, 0, 0);
        }
    };
    // 1x1 source 1: A = the centre pixels of the tile (kCLoads float4 per thread), B = 4 x 64 float4
    // (plain variant: offsets computed once, as for source 0)
    int voffC[(UP || S2) ? 1 : kCLoads], voffB1 = kOob;
    __amdgpu_buffer_rsrc_t rsC, rsW1;
    if constexpr (!UP && !S2) {
        const ConvSrc &s = a.s[1];
        if (nc1 > 0) {
            rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)n * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
            rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, s.cblocks * 4 * a.Cout_pad * 16, 0x00020000);
#pragma unroll
            for (int k = 0; k < kCLoads; ++k) {
                const int slot = tid + 256 * k;
                const int q = slot & 3, pix = slot >> 2;
                const int iy = y0 + (pix >> 4), ix = x0 + (pix & 15);
                voffC[k] = ((iy < s.H) & (ix < s.W)) ? ((iy * s.W + ix) * s.cs + 4 * q) * 4 : kOob;
            }
            voffB1 = (tid < 4 * kN) ? ((tid / kN) * a.Cout_pad + n0 + (tid % kN)) * 16 : kOob;
        }
    }
    auto issue1 = [&](int c) {
        const ConvSrc &s = a.s[1];
        if constexpr (!UP && !S2) {
#pragma unroll
            for (int k = 0; k < kCLoads; ++k) pa[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsC, voffC[k], 64 * c, 0));
            pb[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW1, voffB1, c * 4 * a.Cout_pad * 16, 0));
            return;
        }
        const bool up = UP && s.up_in[0] != nullptr && 16 * c >= s.up_c0;
#pragma unroll
        for (int k = 0; k < kCLoads; ++k) {
            const int slot = tid + 256 * k;
            const int q = slot & 3, pix = slot >> 2;
            const int py = pix >> 4, px = pix & 15;
            const int iy = y0 + py, ix = x0 + px;
            const bool ok = (iy < s.H) & (ix < s.W);
            if (UP && up) {
                const float *tp[4];
                up_taps(s, c, iy, ix, q, ok, tp);
#pragma unroll
                for (int e = 0; e < 4; ++e) pa[4 * k + e] = *reinterpret_cast<const f32x4 *>(tp[e]);
            } else {
                const float *p = ok ? s.in + ((size_t)(n * s.H + iy) * s.W + ix) * s.cs + 16 * c + 4 * q : g_zero_page;
                pa[(UP ? 4 : 1) * k] = *reinterpret_cast<const f32x4 *>(p);
            }
        }
        // 1x1 panel: slot = q * kN + co (4 * kN float4)
        pb[0] = (tid < 4 * kN) ? *reinterpret_cast<const f32x4 *>(s.w + ((size_t)(4 * c + tid / kN) * a.Cout_pad + n0 + (tid % kN)) * 4)
                               : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto commit1 = [&](int c) {
        const bool up = UP && a.s[1].up_in[0] != nullptr && 16 * c >= a.s[1].up_c0;
#pragma unroll
        for (int k = 0; k < kCLoads; ++k) {
            const int slot = tid + 256 * k;
            const int q = slot & 3, pix = slot >> 2;
            const int py = pix >> 4, px = pix & 15;
            f32x4 v = pa[(UP ? 4 : 1) * k];
            if (UP && up) v = up_blend(y0 + py, x0 + px, pa[4 * k], pa[4 * k + 1], pa[4 * k + 2], pa[4 * k + 3]);
            sA[((py + 1) * 4 + q) * kHaloW + px + 1] = v;
        }
        if (tid < 4 * kN) sB[tid] = pb[0];
    };
    auto compute1 = [&]() {
        f32x4 A[RW], Bf[NJ];
#pragma unroll
        for (int i = 0; i < RW; ++i) A[i] = sA[((RW * wave + i + 1) * 4 + h) * kHaloW + ln + 1];
#pragma unroll
        for (int j = 0; j < NJ; ++j) Bf[j] = sB[h * kN + 16 * j + ln];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < RW; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bf[j][kk], A[i][kk], acc[i][j], 0, 0, 0);
    };

    // 3x3 stride-2 source 1 (S2): halo pixel (hy, hx) = input (2*y0 + hy - 1, 2*x0 + hx - 1); zero padding
    int voffA2[S2 ? kA2Loads : 1], voffB2[S2 ? kBLoads : 1];
    __amdgpu_buffer_rsrc_t rsA2, rsW2;
    if constexpr (S2) {
        const ConvSrc &s = a.s[1];
        rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.in + (size_t)n * s.H * s.W * s.cs), 0, s.H * s.W * s.cs * 4, 0x00020000);
        rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.w), 0, 9 * s.cblocks * 4 * a.Cout_pad * 16, 0x00020000);
#pragma unroll
        for (int k = 0; k < kA2Loads; ++k) {
            const int slot = tid + 256 * k;
            const int q = slot & 3, pix = slot >> 2;
            const int hy = pix / kHalo2W, hx = pix - hy * kHalo2W;
            const int iy = 2 * y0 + hy - 1, ix = 2 * x0 + hx - 1;
            const bool ok = (slot < kA2Slots) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
            voffA2[k] = ok ? ((iy * s.W + ix) * s.cs + 4 * q) * 4 : kOob;
        }
#pragma unroll
        for (int k = 0; k < kBLoads; ++k) {
            const int slot = tid + 256 * k;
            const int co = slot % kN, tq = slot / kN;
            const int tap = tq >> 2, q = tq & 3;
            voffB2[k] = (slot < kBSlots3) ? ((tap * s.cblocks * 4 + q) * a.Cout_pad + co + n0) * 16 : kOob;
        }
    }
    auto issue2 = [&](int c) {
        if constexpr (S2) {
#pragma unroll
            for (int k = 0; k < kA2Loads; ++k) pa[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA2, voffA2[k], 64 * c, 0));
#pragma unroll
            for (int k = 0; k < kBLoads; ++k) pb[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW2, voffB2[k], c * 4 * a.Cout_pad * 16, 0));
        }
    };
    auto commit2 = [&]() {
#pragma unroll
        for (int k = 0; k < kA2Loads; ++k) {
            const int slot = tid + 256 * k;
            const int q = slot & 3, pix = slot >> 2;
            const int hy = pix / kHalo2W, hx = pix - hy * kHalo2W;
            if (slot < kA2Slots) sA[((hy * 4 + q) * 2 + (hx & 1)) * kHalo2Wh + (hx >> 1)] = pa[k];
        }
#pragma unroll
        for (int k = 0; k < kBLoads; ++k)
            if (tid + 256 * k < kBSlots3) sB[k * 256 + tid] = pb[k];
    };
    auto compute2 = [&]() {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            f32x4 A[RW], Bf[NJ];
#pragma unroll
            for (int i = 0; i < RW; ++i) A[i] = sA[(((2 * (RW * wave + i) + dy) * 4 + h) * 2 + (dx & 1)) * kHalo2Wh + ln + (dx >> 1)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) Bf[j] = sB[(tap * 4 + h) * kN + 16 * j + ln];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < RW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bf[j][kk], A[i][kk], acc[i][j], 0, 0, 0);
        }
    };

    // ---- source 0 chunks ---------------------------------------------------------------------
    {
        const int lo = min(t0, nc0), hi = min(t1, nc0);
        if (lo < hi) issue3(lo);
#pragma unroll 1
        for (int c = lo; c < hi; ++c) {
            __syncthreads();  // every wave is done reading the previous chunk's LDS image
            commit3(c);
            __syncthreads();
            if (c + 1 < hi) issue3(c + 1);  // in flight under the 288 MFMAs below
            compute3();
        }
    }
    // ---- source 1 chunks (fused projection: 1x1, or 3x3 stride 2 with S2) -----------------------------
    if (nc1 > 0) {
        const int lo = max(t0 - nc0, 0), hi = max(t1 - nc0, 0);
        if constexpr (S2) {
            if (lo < hi) issue2(lo);
#pragma unroll 1
            for (int c = lo; c < hi; ++c) {
                __syncthreads();
                commit2();
                __syncthreads();
                if (c + 1 < hi) issue2(c + 1);
                compute2();
            }
        } else if constexpr (!UP && lds_g1(RW, NJ) > 1) {
            // rounds of G1 chunks (see lds_g1): same chunk order and MFMA order as one chunk per round, so the same bits
            constexpr int G1 = lds_g1(RW, NJ);
            constexpr int kB1 = 4 * kN, kC1 = lds_c1_slots(RW);
            f32x4 qa[G1][kCLoads], qb[G1];
            auto issue1b = [&](int c) {
#pragma unroll
                for (int g = 0; g < G1; ++g)
                    if (c + g < hi) {
#pragma unroll
                        for (int k = 0; k < kCLoads; ++k)
                            qa[g][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsC, voffC[k], 64 * (c + g), 0));
                        qb[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW1, voffB1, (c + g) * 4 * a.Cout_pad * 16, 0));
                    }
            };
            if (lo < hi) issue1b(lo);
#pragma unroll 1
            for (int c = lo; c < hi; c += G1) {
                __syncthreads();
#pragma unroll
                for (int g = 0; g < G1; ++g)
                    if (c + g < hi) {
#pragma unroll
                        for (int k = 0; k < kCLoads; ++k) {
                            const int slot = tid + 256 * k;
                            const int q = slot & 3, pix = slot >> 2;
                            const int py = pix >> 4, px = pix & 15;
                            if (g == 0) sA[((py + 1) * 4 + q) * kHaloW + px + 1] = qa[g][k];
                            else sB[G1 * kB1 + (g - 1) * kC1 + (py * 4 + q) * kHaloW + px] = qa[g][k];
                        }
                        if (tid < kB1) sB[g * kB1 + tid] = qb[g];
                    }
                __syncthreads();
                if (c + G1 < hi) issue1b(c + G1);
#pragma unroll
                for (int g = 0; g < G1; ++g)
                    if (c + g < hi) {
                        f32x4 A[RW], Bf[NJ];
#pragma unroll
                        for (int i = 0; i < RW; ++i)
                            A[i] = g == 0 ? sA[((RW * wave + i + 1) * 4 + h) * kHaloW + ln + 1]
                                          : sB[G1 * kB1 + (g - 1) * kC1 + ((RW * wave + i) * 4 + h) * kHaloW + ln];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) Bf[j] = sB[g * kB1 + h * kN + 16 * j + ln];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int i = 0; i < RW; ++i)
#pragma unroll
                                for (int j = 0; j < NJ; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bf[j][kk], A[i][kk], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            if (lo < hi) issue1(lo);
#pragma unroll 1
            for (int c = lo; c < hi; ++c) {
                __syncthreads();
                commit1(c);
                __syncthreads();
                if (c + 1 < hi) issue1(c + 1);
                compute1();
            }
        }
    }

    // ---- epilogue (same transposed C/D layout as above: 4 consecutive channels per lane) -------
    // buffer stores / residual loads: one byte offset per tile row of the lane, the channel sub-tile is the instruction's
    // immediate offset; rows and columns past the image are out of the descriptor's range (dropped / read as 0)
    if (a.S > 1) {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int oy = y0 + RW * wave + i, ox = x0 + ln;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const size_t m = ((size_t)n * a.Ho + oy) * a.Wo + ox;
            float *o = a.ws + ((size_t)sp * a.M + m) * a.Cout_pad + n0 + 4 * h;
#pragma unroll
            for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4 *>(o + 16 * j) = acc[i][j];
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)n * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)n * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                          a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    int voffO[RW], voffR[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int oy = y0 + RW * wave + i, ox = x0 + ln;
        const bool ok = (oy < a.Ho) & (ox < a.Wo);
        const int pixel = oy * a.Wo + ox;
        voffO[i] = ok ? (pixel * a.out_cs + n0 + 4 * h) * 4 : kOob;
        voffR[i] = ok ? (pixel * a.res_cs + n0 + 4 * h) * 4 : kOob;
    }
    if (a.res) {  // all residual loads of the tile in flight before the first add
        f32x4 rv[RW][NJ];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) rv[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, voffR[i] + 64 * j, 0, 0));
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] += rv[i][j];
    }
    // activation selector tested once, straight-line element code (act_dispatch, conv_args.h)
    act_dispatch(a.act, a.slope, [&](auto fn) {
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 v = acc[i][j];  // bias already inside (bias_first)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fn(v[r]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsO, voffO[i] + 64 * j, 0, 0);
            }
    });
}

template <int RW, bool UP, int NJ = 4, bool NORM = false, bool S2 = false>
__global__ __launch_bounds__(256) void conv3x3_lds_k(const LdsConvArgs la) {
    __shared__ f32x4 sA[lds_a_slots(RW, S2)];
    __shared__ f32x4 sB[lds_b_slots(NJ)];
    conv3x3_lds_body<RW, UP, NJ, NORM, S2>(la, blockIdx.x, gridDim.x, sA, sB);
}
// fused-upsample variants: keep 3 workgroups / CU (what the LDS footprint allows) although the 4x prefetch wants ~180 VGPRs
template <int RW>
__global__ __launch_bounds__(256, 3) void conv3x3_lds_up_k(const LdsConvArgs la) {
    __shared__ f32x4 sA[lds_a_slots(RW)];
    __shared__ f32x4 sB[lds_b_slots(4)];
    conv3x3_lds_body<RW, true, 4>(la, blockIdx.x, gridDim.x, sA, sB);
}

// Grouped launch: up to kMaxGroup INDEPENDENT convolutions (same dependency level of a plan, see
// implicit-depth_amd/nhwc.py:Plan.schedule) share one grid, so the small low-resolution layers of
// the UNet++ — each of which fills a fraction of the 256 CUs — run side by side.  Descriptors
// travel by value in the kernel arguments (no device-side table to upload).
constexpr int kMaxGroup = 12;
struct LdsGroupArgs {
    int n;
    unsigned start[kMaxGroup + 1];
    LdsConvArgs op[kMaxGroup];
};

template <int RW, bool UP, int NJ = 4>
__global__ __launch_bounds__(256) void conv3x3_lds_group_k(const LdsGroupArgs g) {
    __shared__ f32x4 sA[lds_a_slots(RW)];
    __shared__ f32x4 sB[lds_b_slots(NJ)];
    int idx = 0;
    for (int i = 1; i < g.n; ++i)
        if (blockIdx.x >= g.start[i]) idx = i;
    conv3x3_lds_body<RW, UP, NJ>(g.op[idx], blockIdx.x - g.start[idx], g.start[idx + 1] - g.start[idx], sA, sB);
}

struct ReduceDesc {
    const float *ws, *bias, *res;
    float *out;
    int M, Cout, Cout_pad, S, res_cs, out_cs, act;
    float slope;
};
struct ReduceGroupArgs {
    int n;
    unsigned start[kMaxGroup + 1];  // in units of 256-element blocks
    ReduceDesc d[kMaxGroup];
};

__global__ __launch_bounds__(256) void splitk_reduce_group_k(const ReduceGroupArgs g) {
    int idx = 0;
    for (int i = 1; i < g.n; ++i)
        if (blockIdx.x >= g.start[i]) idx = i;
    const ReduceDesc &d = g.d[idx];
    const long long t = (long long)(blockIdx.x - g.start[idx]) * 256 + threadIdx.x;
    const long long total = (long long)d.M * d.Cout;
    if (t >= total) return;
    const int co = (int)(t % d.Cout);
    const long long m = t / d.Cout;
    float v = 0.f;
    for (int s = 0; s < d.S; ++s) v += d.ws[((size_t)s * d.M + m) * d.Cout_pad + co];
    if (d.bias) v += d.bias[co];
    if (d.res) v += d.res[m * d.res_cs + co];
    d.out[m * d.out_cs + co] = act_apply(v, d.act, d.slope);
}

// split-K tail: out = act(sum_s ws[s] + bias + res)
__global__ __launch_bounds__(256) void splitk_reduce_k(const float *__restrict__ ws, const float *__restrict__ bias,
                                                       const float *__restrict__ res, float *__restrict__ out,
                                                       int M, int Cout, int Cout_pad, int S, int res_cs, int out_cs,
                                                       int act, float slope) {
    const long long total = (long long)M * Cout;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int co = (int)(t % Cout);
        const long long m = t / Cout;
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += ws[((size_t)s * M + m) * Cout_pad + co];
        if (bias) v += bias[co];
        if (res) v += res[m * res_cs + co];
        out[m * out_cs + co] = act_apply(v, act, slope);
    }
}

// OIHW -> [tap][ci/4][co][ci%4], zero padded
__global__ __launch_bounds__(256) void pack_weight_k(const float *__restrict__ w, float *__restrict__ dst, int Cout,
                                                     int Cin, int ks, int Cin_pad, int Cout_pad) {
    const int taps = ks * ks;
    const long long total = (long long)taps * Cin_pad * Cout_pad;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int c4 = (int)(t & 3);
        long long r = t >> 2;
        const int co = (int)(r % Cout_pad);
        r /= Cout_pad;
        const int cq = (int)(r % (Cin_pad / 4));
        const int tap = (int)(r / (Cin_pad / 4));
        const int ci = cq * 4 + c4;
        float v = 0.f;
        if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * taps + tap];
        dst[t] = v;
    }
}

// bilinear x2 (align_corners=False): out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] + .25 in[i+1],
// indices clamped at the border; evaluated as h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11).
// One thread = one low-resolution pixel x 4 channels -> the 2x2 output pixels it centres: 9 loads for 4 stores (4 per store
// with one thread per output pixel), the three horizontally blended rows shared by the two output rows (the same expression tree
// per output, so the same bits), and one 32-bit index decomposition per quad - with a thread per output the kernel was
// bound by its integer divisions, not by HBM (round 5: 3.7 -> TB/s of the 4.6 GB a 32-frame step moves through it).
__device__ __forceinline__ void upsample2_body(const float *__restrict__ in, float *__restrict__ out, int N, int H, int W, int C,
                                               int in_cs, int out_cs, unsigned blk, unsigned nblk) {
    const unsigned cq = (unsigned)C >> 2;
    const int Wo = 2 * W;
    const unsigned total = (unsigned)N * H * W * cq;  // (host-checked: < 2^31)
    constexpr float lo = 0.25f, hi = 0.75f;
    for (unsigned t = blk * 256u + threadIdx.x; t < total; t += nblk * 256u) {
        const unsigned q = t % cq;
        unsigned p = t / cq;
        const int ix = (int)(p % (unsigned)W);
        p /= (unsigned)W;
        const int iy = (int)(p % (unsigned)H);
        const int n = (int)(p / (unsigned)H);
        const int xm = max(ix - 1, 0), xp = min(ix + 1, W - 1);
        const int ym = max(iy - 1, 0), yp = min(iy + 1, H - 1);
        const float *b = in + (size_t)n * H * W * in_cs + 4 * q;
        const float *r0 = b + (size_t)ym * W * in_cs, *r1 = b + (size_t)iy * W * in_cs, *r2 = b + (size_t)yp * W * in_cs;
        float4 v[3][3];
        v[0][0] = *reinterpret_cast<const float4 *>(r0 + (size_t)xm * in_cs);
        v[0][1] = *reinterpret_cast<const float4 *>(r0 + (size_t)ix * in_cs);
        v[0][2] = *reinterpret_cast<const float4 *>(r0 + (size_t)xp * in_cs);
        v[1][0] = *reinterpret_cast<const float4 *>(r1 + (size_t)xm * in_cs);
        v[1][1] = *reinterpret_cast<const float4 *>(r1 + (size_t)ix * in_cs);
        v[1][2] = *reinterpret_cast<const float4 *>(r1 + (size_t)xp * in_cs);
        v[2][0] = *reinterpret_cast<const float4 *>(r2 + (size_t)xm * in_cs);
        v[2][1] = *reinterpret_cast<const float4 *>(r2 + (size_t)ix * in_cs);
        v[2][2] = *reinterpret_cast<const float4 *>(r2 + (size_t)xp * in_cs);
        // even output column 2ix: taps (ix-1, ix) with weights (.25, .75); odd column 2ix+1: taps (ix, ix+1) with (.75, .25)
        float4 e[3], o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            e[r].x = lo * v[r][0].x + hi * v[r][1].x; e[r].y = lo * v[r][0].y + hi * v[r][1].y;
            e[r].z = lo * v[r][0].z + hi * v[r][1].z; e[r].w = lo * v[r][0].w + hi * v[r][1].w;
            o[r].x = hi * v[r][1].x + lo * v[r][2].x; o[r].y = hi * v[r][1].y + lo * v[r][2].y;
            o[r].z = hi * v[r][1].z + lo * v[r][2].z; o[r].w = hi * v[r][1].w + lo * v[r][2].w;
        }
        // even output row 2iy: rows (iy-1, iy) with (.25, .75); odd row 2iy+1: rows (iy, iy+1) with (.75, .25)
        float4 o00, o01, o10, o11;
        o00.x = lo * e[0].x + hi * e[1].x; o00.y = lo * e[0].y + hi * e[1].y; o00.z = lo * e[0].z + hi * e[1].z; o00.w = lo * e[0].w + hi * e[1].w;
        o01.x = lo * o[0].x + hi * o[1].x; o01.y = lo * o[0].y + hi * o[1].y; o01.z = lo * o[0].z + hi * o[1].z; o01.w = lo * o[0].w + hi * o[1].w;
        o10.x = hi * e[1].x + lo * e[2].x; o10.y = hi * e[1].y + lo * e[2].y; o10.z = hi * e[1].z + lo * e[2].z; o10.w = hi * e[1].w + lo * e[2].w;
        o11.x = hi * o[1].x + lo * o[2].x; o11.y = hi * o[1].y + lo * o[2].y; o11.z = hi * o[1].z + lo * o[2].z; o11.w = hi * o[1].w + lo * o[2].w;
        float *d = out + (((size_t)n * 2 * H + 2 * iy) * Wo + 2 * ix) * out_cs + 4 * q;
        *reinterpret_cast<float4 *>(d) = o00;
        *reinterpret_cast<float4 *>(d + out_cs) = o01;
        *reinterpret_cast<float4 *>(d + (size_t)Wo * out_cs) = o10;
        *reinterpret_cast<float4 *>(d + (size_t)Wo * out_cs + out_cs) = o11;
    }
}

__global__ __launch_bounds__(256) void upsample2_k(const float *__restrict__ in, float *__restrict__ out, int N, int H,
                                                   int W, int C, int in_cs, int out_cs) {
    upsample2_body(in, out, N, H, W, C, in_cs, out_cs, blockIdx.x, gridDim.x);
}

// nearest x2 (F.interpolate(scale_factor=2, mode="nearest"), networks_fast.py:43): out[y,x] = in[y>>1, x>>1]
__global__ __launch_bounds__(256) void upsample2_nearest_k(const float *__restrict__ in, float *__restrict__ out, int N, int H,
                                                           int W, int C, int in_cs, int out_cs) {
    const int cq = C >> 2;
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * Ho * Wo * cq;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int q = (int)(t % cq);
        long long p = t / cq;
        const int x = (int)(p % Wo);
        p /= Wo;
        const int y = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const float4 v = *reinterpret_cast<const float4 *>(in + (((size_t)n * H + (y >> 1)) * W + (x >> 1)) * in_cs + 4 * q);
        *reinterpret_cast<float4 *>(out + (((size_t)n * Ho + y) * Wo + x) * out_cs + 4 * q) = v;
    }
}

// channel-strided NHWC copy (places an existing feature map into a slice of a concat buffer)
__global__ __launch_bounds__(256) void copy_nhwc_k(const float *__restrict__ in, float *__restrict__ out, long long npix, int C,
                                                   int in_cs, int out_cs) {
    const int cq = C >> 2;
    const long long total = npix * cq;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int q = (int)(t % cq);
        const long long p = t / cq;
        *reinterpret_cast<float4 *>(out + p * out_cs + 4 * q) = *reinterpret_cast<const float4 *>(in + p * in_cs + 4 * q);
    }
}

// (N,C,HW) dense -> NHWC slice with channel stride out_cs; LDS tile keeps both sides coalesced
__device__ __forceinline__ void import_nchw_body(const float *__restrict__ src, float *__restrict__ dst, int C, int HW, int out_cs, int bx, int by, int img,
                                                 float (&tile)[32][65]) {
    const int p0 = bx * 64, c0 = by * 32;
    const float *s = src + (size_t)img * C * HW;
    float *d = dst + (size_t)img * HW * out_cs;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int p = i & 63, c = i >> 6;
        if (p0 + p < HW && c0 + c < C) tile[c][p] = s[(size_t)(c0 + c) * HW + p0 + p];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int c = i & 31, p = i >> 5;
        if (p0 + p < HW && c0 + c < C) d[(size_t)(p0 + p) * out_cs + c0 + c] = tile[c][p];
    }
}

__global__ __launch_bounds__(256) void import_nchw_k(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                     int HW, int out_cs) {
    __shared__ float tile[32][65];
    import_nchw_body(src, dst, C, HW, out_cs, blockIdx.x, blockIdx.y, blockIdx.z, tile);
}

// The layout imports of one dependency level (the five maps of the image-encoder pyramid, bd_model.py:218) as ONE grid: at one frame each is a
// 5-us launch of a few dozen blocks.  Same body, same values.
constexpr int kMaxImport = 8;
struct ImportGroupArgs {
    int n;
    unsigned start[kMaxImport + 1];
    struct { const float *src; float *dst; int C, HW, out_cs, nbx, nby; } d[kMaxImport];
};
__global__ __launch_bounds__(256) void import_nchw_group_k(const ImportGroupArgs g) {
    __shared__ float tile[32][65];
    int idx = 0;
    for (int i = 1; i < g.n; ++i)
        if (blockIdx.x >= g.start[i]) idx = i;
    const auto &m = g.d[idx];
    const unsigned local = blockIdx.x - g.start[idx];
    const int bx = (int)(local % (unsigned)m.nbx);
    const unsigned r = local / (unsigned)m.nbx;
    import_nchw_body(m.src, m.dst, m.C, m.HW, m.out_cs, bx, (int)(r % (unsigned)m.nby), (int)(r / (unsigned)m.nby), tile);
}

__global__ __launch_bounds__(256) void export_nchw_k(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                     int HW, int in_cs) {
    __shared__ float tile[32][65];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 32;
    const float *s = src + (size_t)img * HW * in_cs;
    float *d = dst + (size_t)img * C * HW;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int c = i & 31, p = i >> 5;
        if (p0 + p < HW && c0 + c < C) tile[c][p] = s[(size_t)(p0 + p) * in_cs + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int p = i & 63, c = i >> 6;
        if (p0 + p < HW && c0 + c < C) d[(size_t)(c0 + c) * HW + p0 + p] = tile[c][p];
    }
}

// 1x1 convolution straight from an NCHW tensor into an NHWC slice: the first layer of the matching-encoder head
// (nn.Conv2d(64, 128, 1) on the ResNet layer1 map, reference modules/networks.py:279) without the layout-import pass
// in front of it.  HBM-bound (256 B in, 512 B out per pixel against 16 KFLOP): the point is to touch every byte once.
//   * D^T = W . X^T as in the kernels above: the weights are the A operand (whole packed matrix LDS-resident, 32 KiB
//     for 64 -> 128), the B operand of lane (ln, h) is channels 16c+4h..+3 of pixel ln — in NCHW four dword loads,
//     each a 64-byte run of 16 consecutive pixels of one channel plane;
//   * a wave owns 16 consecutive pixels x 128 channels (8 accumulators), a workgroup 64 pixels; persistent
//     grid-stride loop over tiles with the next tile's 16 input dwords prefetched under the current 128 MFMAs; 4
//     workgroups per CU keep ~50 KB of loads in flight per CU;
//   * epilogue: + bias, one 16-byte store per accumulator (4 consecutive channels of a pixel).
// Cin == 64 and Cout == 128 only (every ResNet-18/34 matching encoder); other widths keep import_nchw_k + conv_mfma_k.
constexpr int kPwCin = 64, kPwCout = 128, kPwTile = 64;  // pixels per workgroup
struct PwNchwArgs {
    const float *in, *w, *bias;
    float *out;
    int HW, out_cs, tiles_per_img;
    long long tiles;
};

__global__ __launch_bounds__(256, 4) void pointwise_nchw_k(const PwNchwArgs a) {
    constexpr int C16 = kPwCin / 16, NT = kPwCout / 16;
    __shared__ f32x4 sW[(kPwCin / 4) * kPwCout];  // [ci / 4][co] float4 = the packed layout of idh_pack_conv_weight(ks = 1)
    {
        const f32x4 *g = reinterpret_cast<const f32x4 *>(a.w);
        for (int i = threadIdx.x; i < (kPwCin / 4) * kPwCout; i += 256) sW[i] = g[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, h = lane >> 4;

    // buffer loads: one lane offset (VGPR) + a scalar channel offset per load instead of 16 64-bit addresses; pixels
    // past the end of the image are out of the descriptor's range and read as 0
    const unsigned img_bytes = (unsigned)kPwCin * (unsigned)a.HW * 4u;
    auto load = [&](long long tile, f32x4 (&x)[C16]) {
        const int n = (int)(tile / a.tiles_per_img);
        const int p = (int)(tile - (long long)n * a.tiles_per_img) * kPwTile + 16 * wave + ln;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in + (size_t)n * kPwCin * a.HW), 0, img_bytes, 0x00020000);
        const int voff = p < a.HW ? (4 * h * a.HW + p) * 4 : (int)0x7fffffff;
#pragma unroll
        for (int c = 0; c < C16; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                x[c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (16 * c + e) * a.HW * 4, 0));
    };
    auto compute = [&](long long tile, const f32x4 (&x)[C16]) {
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the weight fragments are the same for every tile: without this opaque zero the compiler hoists all 32 LDS
        // reads out of the tile loop (128 VGPRs) and spills
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const f32x4 *sWt = sW + opaque;
#pragma unroll
        for (int c = 0; c < C16; ++c)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 A = sWt[(4 * c + h) * kPwCout + 16 * j + ln];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], x[c][kk], acc[j], 0, 0, 0);
            }
        const int n = (int)(tile / a.tiles_per_img);
        const int p = (int)(tile - (long long)n * a.tiles_per_img) * kPwTile + 16 * wave + ln;
        if (p < a.HW) {
            float *o = a.out + ((size_t)n * a.HW + p) * a.out_cs + 4 * h;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 v = acc[j];
                if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + 16 * j + 4 * h);
                *reinterpret_cast<f32x4 *>(o + 16 * j) = v;
            }
        }
    };
    f32x4 xa[C16], xb[C16];
    long long t = blockIdx.x;
    if (t >= a.tiles) return;
    load(t, xa);
    for (;;) {
        const long long t1 = t + gridDim.x;
        if (t1 < a.tiles) load(t1, xb);
        compute(t, xa);
        if (t1 >= a.tiles) break;
        const long long t2 = t1 + gridDim.x;
        if (t2 < a.tiles) load(t2, xa);
        compute(t1, xb);
        if (t2 >= a.tiles) break;
        t = t2;
    }
}

// IDH_OP_POINTWISE_UP: out = W . x (+ bias) + up2(low) - a 1x1 convolution of an NHWC tensor plus the x2 bilinear upsampling (align_corners=False,
// upsample2_k's expression) of a half-resolution NHWC tensor of Cout channels.  It is the projection branch of the decoder blocks whose input is
// cat(right, up(lo), up(lo2)) (networks.py:52-77: BasicBlock's downsample(x), layers.py:86-92): a 1x1 convolution commutes with bilinear
// upsampling (the interpolation weights sum to one), so W . cat = W_a . right + up(W_b . lo + W_c . lo2) - two thirds of the projection run at a
// quarter of the pixels, and conv2 takes the sum as an ordinary residual instead of multiplying 192 channels in its epilogue.
//   * D^T = W . X^T as in pointwise_nchw_k: the packed weights ([ci / 4][co] float4, idh_pack_conv_weight ks = 1) LDS-resident, lane (ln, h) loads
//     channels 16c + 4h .. + 3 of pixel ln (one 16-byte load per 16-channel block), a wave owns 16 pixels x Cout channels, a workgroup 64 pixels;
//   * epilogue: the lane's 4 consecutive output channels of its pixel + the bilinear blend of the same channels of the 4 low-resolution
//     neighbours (4 x NT 16-byte loads, L2-resident: the low map is a quarter of the output), one 16-byte store per 16-channel block.
// HBM-bound (reads Cin, writes Cout floats per pixel).  Cin == Cout in {64, 128} (every decoder level that takes the F(4x4) kernel).
struct PwUpArgs {
    const float *in, *w, *bias, *low;
    float *out;
    int H, W, in_cs, low_cs, out_cs, tiles_per_img;
    long long tiles;
};

template <int C16, int NT>
__global__ __launch_bounds__(256) void pointwise_up_k(const PwUpArgs a) {
    constexpr int kCout = 16 * NT;
    extern __shared__ f32x4 sWu[];  // [4 * C16][kCout]
    {
        const f32x4 *g = reinterpret_cast<const f32x4 *>(a.w);
        for (int i = threadIdx.x; i < 4 * C16 * kCout; i += 256) sWu[i] = g[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, h = lane >> 4;
    const int HW = a.H * a.W, Hl = a.H >> 1, Wl = a.W >> 1;
    for (long long tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const int n = (int)(tile / a.tiles_per_img);
        const int p = (int)(tile - (long long)n * a.tiles_per_img) * kPwTile + 16 * wave + ln;
        const bool ok = p < HW;
        const int pc = ok ? p : 0;
        const float *ip = a.in + ((size_t)n * HW + pc) * a.in_cs + 4 * h;
        f32x4 x[C16];
#pragma unroll
        for (int c = 0; c < C16; ++c) x[c] = *reinterpret_cast<const f32x4 *>(ip + 16 * c);
        // low-resolution neighbours and weights of output pixel (y, xq): exactly upsample2_body's
        const int y = pc / a.W, xq = pc - y * a.W;
        const int iy = y >> 1, ix = xq >> 1;
        int y0, y1, x0, x1;
        float hy0, hy1, wx0, wx1;
        if (y & 1) { y0 = iy; y1 = min(iy + 1, Hl - 1); hy0 = 0.75f; hy1 = 0.25f; }
        else { y0 = max(iy - 1, 0); y1 = iy; hy0 = 0.25f; hy1 = 0.75f; }
        if (xq & 1) { x0 = ix; x1 = min(ix + 1, Wl - 1); wx0 = 0.75f; wx1 = 0.25f; }
        else { x0 = max(ix - 1, 0); x1 = ix; wx0 = 0.25f; wx1 = 0.75f; }
        const float *lb = a.low + (size_t)n * Hl * Wl * a.low_cs + 4 * h;
        const float *t00 = lb + ((size_t)y0 * Wl + x0) * a.low_cs, *t01 = lb + ((size_t)y0 * Wl + x1) * a.low_cs;
        const float *t10 = lb + ((size_t)y1 * Wl + x0) * a.low_cs, *t11 = lb + ((size_t)y1 * Wl + x1) * a.low_cs;
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + 16 * j + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
        int opaque = 0;  // (keeps the weight fragments' LDS reads inside the tile loop: hoisted they are 4 * C16 * NT registers)
        asm volatile("" : "+s"(opaque));
        const f32x4 *sWt = sWu + opaque;
#pragma unroll
        for (int c = 0; c < C16; ++c)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 A = sWt[(4 * c + h) * kCout + 16 * j + ln];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], x[c][kk], acc[j], 0, 0, 0);
            }
        if (ok) {
            float *o = a.out + ((size_t)n * HW + p) * a.out_cs + 4 * h;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 p00 = *reinterpret_cast<const f32x4 *>(t00 + 16 * j), p01 = *reinterpret_cast<const f32x4 *>(t01 + 16 * j);
                const f32x4 p10 = *reinterpret_cast<const f32x4 *>(t10 + 16 * j), p11 = *reinterpret_cast<const f32x4 *>(t11 + 16 * j);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][e] + (hy0 * (wx0 * p00[e] + wx1 * p01[e]) + hy1 * (wx0 * p10[e] + wx1 * p11[e]));
                *reinterpret_cast<f32x4 *>(o + 16 * j) = v;
            }
        }
    }
}

// 1x1 conv to a single channel: one thread per pixel
__global__ __launch_bounds__(256) void pointwise_head_k(const float *__restrict__ in, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ out,
                                                        float *__restrict__ out_exp, long long M, int C, int in_cs) {
    for (long long m = blockIdx.x * 256ll + threadIdx.x; m < M; m += gridDim.x * 256ll) {
        const float4 *p = reinterpret_cast<const float4 *>(in + m * in_cs);
        const float4 *wv = reinterpret_cast<const float4 *>(w);
        float s = 0.f;
        for (int q = 0; q < (C >> 2); ++q) {
            const float4 a = p[q], b = wv[q];
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
        const float v = s + (bias ? bias[0] : 0.f);
        out[m] = v;
        if (out_exp) out_exp[m] = expf(v);  // depth = exp(log-depth), depth_model.py:425-433
    }
}

// ---- InstanceNorm2d (no affine, eps 1e-5) on NHWC, optional LeakyReLU -------------------------
// reference modules/networks.py:279-283 (matching-encoder head).  Deterministic two-stage
// reduction: per-(image, pixel-chunk) channel sums -> per-thread combine in the apply kernel.
constexpr int kInChunk = 1024;  // pixels per partial

// blockDim.x = 256, or 1024 when the launch has too few (image, chunk) pairs to fill the chip (one frame: 8 images x 12 chunks): the chunk
// partition - and with it the workspace layout of the C ABI - stays, a chunk's pixels are spread over four times the threads.
// The reduction TREE does not depend on the block size: a chunk is always summed as SUB * blockDim.x / (C/4) "rows" (row r takes pixels
// p0 + r, p0 + r + rows, ...) that are then combined in row order; a 256-thread block of the C >= 64 case carries the four rows a
// 1024-thread block would give to four threads (SUB = 4), so a frame's statistics are bit-identical whether it is normalised alone
// (1024 threads) or inside a large batch (256 threads).
template <int SUB>
__global__ __launch_bounds__(1024) void instnorm_stats_k(const float *__restrict__ in, int cs, int C, int HW, int nchunks,
                                                         float *__restrict__ part) {  // part[n][chunk][2][C]
    extern __shared__ float red[];  // SUB * blockDim.x * 8 floats
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int cq = C >> 2;                 // channel quads
    const int q = threadIdx.x % cq;
    const int prow = threadIdx.x / cq, pstep = (int)blockDim.x / cq;  // host guarantees cq divides 256
    const int rows = SUB * pstep;
    const int p0 = chunk * kInChunk, p1 = min(HW, p0 + kInChunk);
    // sums of (x - pivot) and (x - pivot)^2 with pivot = the image's first pixel: E[x^2] - mean^2 on the raw values cancels
    // catastrophically for a channel whose |mean| is large against its spread (~(mean/std)^2 * 1e-7 relative; a biased 1x1 conv on
    // ReLU features); shifted by any sample of the distribution the two terms are both of the order of the variance
    const float4 pv = *reinterpret_cast<const float4 *>(in + (size_t)n * HW * cs + 4 * q);
    float s[SUB][4], ss[SUB][4];
#pragma unroll
    for (int u = 0; u < SUB; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[u][e] = ss[u][e] = 0.f;
    for (int p = p0 + prow; p < p1; p += rows) {
#pragma unroll
        for (int u = 0; u < SUB; ++u) {  // row prow + u * pstep
            const int pp = p + u * pstep;
            if (pp < p1) {
                const float4 v = *reinterpret_cast<const float4 *>(in + ((size_t)n * HW + pp) * cs + 4 * q);
                const float d0 = v.x - pv.x, d1 = v.y - pv.y, d2 = v.z - pv.z, d3 = v.w - pv.w;
                s[u][0] += d0; s[u][1] += d1; s[u][2] += d2; s[u][3] += d3;
                ss[u][0] = fmaf(d0, d0, ss[u][0]); ss[u][1] = fmaf(d1, d1, ss[u][1]); ss[u][2] = fmaf(d2, d2, ss[u][2]); ss[u][3] = fmaf(d3, d3, ss[u][3]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SUB; ++u) {
        float *mine = red + ((size_t)(prow + u * pstep) * cq + q) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mine[e] = s[u][e]; mine[4 + e] = ss[u][e]; }
    }
    __syncthreads();
    // fixed-order combine over the rows (deterministic), one thread per (channel quad, sum); 8 * cq may exceed the block (C >= 128 at 256 threads)
    for (int t = threadIdx.x; t < 8 * cq; t += blockDim.x) {
        const int qq = t >> 3, e = t & 7;
        float a = 0.f;
        for (int r = 0; r < rows; ++r) a += red[(r * cq + qq) * 8 + e];
        part[((size_t)(n * nchunks + chunk) * 2 + (e >> 2)) * C + 4 * qq + (e & 3)] = a;
    }
}

// mean / 1/sqrt(var + eps) per (image, channel) from the chunk partials (shifted by the image's first pixel, see above), in
// fixed chunk order (deterministic); stats[n][2][C] lives behind the partials in the same workspace
__global__ __launch_bounds__(256) void instnorm_finalize_k(const float *__restrict__ part, const float *__restrict__ in, int cs, int C, int HW,
                                                           int nchunks, float *__restrict__ stats) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f, ss = 0.f;
        for (int k = 0; k < nchunks; ++k) {
            const float *o = part + ((size_t)(n * nchunks + k) * 2) * C + c;
            s += o[0];
            ss += o[C];
        }
        const float ms = s / (float)HW;                                    // mean - pivot
        const float var = fmaxf(ss / (float)HW - ms * ms, 0.f);          // biased, as nn.InstanceNorm2d
        stats[(size_t)n * 2 * C + c] = in[(size_t)n * HW * cs + c] + ms;
        stats[(size_t)n * 2 * C + C + c] = 1.0f / sqrtf(var + 1e-5f);
    }
}

__global__ __launch_bounds__(256) void instnorm_apply_k(const float *__restrict__ in, int cs, float *__restrict__ out, int out_cs,
                                                        int C, int HW, long long total, const float *__restrict__ stats, int act,
                                                        float slope) {
    const int cq = C >> 2;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const int q = (int)(t % cq);
        const long long np = t / cq;          // n * HW + p
        const int n = (int)(np / HW);
        const float4 mean = *reinterpret_cast<const float4 *>(stats + (size_t)n * 2 * C + 4 * q);
        const float4 rstd = *reinterpret_cast<const float4 *>(stats + (size_t)n * 2 * C + C + 4 * q);
        const float4 v = *reinterpret_cast<const float4 *>(in + (size_t)np * cs + 4 * q);
        float r[4] = {(v.x - mean.x) * rstd.x, (v.y - mean.y) * rstd.y, (v.z - mean.z) * rstd.z, (v.w - mean.w) * rstd.w};
        for (int e = 0; e < 4; ++e) r[e] = act_apply(r[e], act, slope);
        *reinterpret_cast<float4 *>(out + (size_t)np * out_cs + 4 * q) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// One launch for a whole dependency level at small batch.  At one frame every level of the plan costs 10-35 us almost
// regardless of its flops (launch, ramp, one K-loop latency chain, drain), and a level of the UNet++ grid holds up to
// four different kinds of work — 4-row LDS convs with 64- and 32-channel tiles, a stride-2 direct conv, a bilinear
// upsample feeding the next level's concat — that the homogeneous group kernel above cannot mix.  level_k hosts those
// bodies behind one grid: blockIdx ranges select the member, so its independent ops fill the chip side by side.
// (A persistent kernel with device-wide dependency counters was measured first and rejected: 768 workgroups
// arriving at one agent-scope counter cost 33 us per round, 80 us with the release/acquire fences the non-coherent
// per-XCD L2s need, against 10 us for a kernel boundary — tools/micro/flow_sync.hip.)
enum { LV_LDS4 = 0, LV_LDS2 = 1, LV_MFMA14 = 2, LV_UP2 = 3 };
struct LevelArgs {
    int n;
    unsigned start[kMaxGroup + 1];
    int kind[kMaxGroup];
    LdsConvArgs op[kMaxGroup];  // LV_UP2 uses c.s[0].{in, H, W, Cin, cs}, c.out, c.out_cs and c.M (= images)
};

// WIDE = some member uses 64-channel tiles: without one the kernel keeps the 32-channel tile's LDS footprint (25 KiB,
// 6 workgroups per CU instead of 3) and register count.
template <bool WIDE>
__global__ __launch_bounds__(256) void level_k(const LevelArgs g) {
    __shared__ f32x4 sA[lds_a_slots(1)];
    __shared__ f32x4 sB[lds_b_slots(WIDE ? 4 : 2)];
    int idx = 0;
    for (int i = 1; i < g.n; ++i)
        if (blockIdx.x >= g.start[i]) idx = i;
    const unsigned blk = blockIdx.x - g.start[idx], nblk = g.start[idx + 1] - g.start[idx];
    const LdsConvArgs &la = g.op[idx];
    switch (g.kind[idx]) {
        case LV_LDS4:
            if constexpr (WIDE) conv3x3_lds_body<1, false, 4>(la, blk, nblk, sA, sB);
            break;
        case LV_LDS2: conv3x3_lds_body<1, false, 2>(la, blk, nblk, sA, sB); break;
        case LV_MFMA14: conv_mfma_body<1, 4>(la.c, blk, nblk); break;
        default: upsample2_body(la.c.s[0].in, la.c.out, la.c.M, la.c.s[0].H, la.c.s[0].W, la.c.s[0].Cin, la.c.s[0].cs, la.c.out_cs, blk, nblk); break;
    }
}

inline int ceil16(int v) { return (v + 15) & ~15; }

// A validated conv op, ready to launch: either the LDS-staged kernel (lds_rows = 8 / 4) or the
// direct-fragment kernel (lds_rows = 0, tile tm x tn).
struct PreparedConv {
    ConvArgs a;
    LdsConvArgs la;
    int lds_rows, tm, tn;  // lds_rows = 16: split-precision kernel, tm = IDH_SPLIT_* mode; 32: Winograd kernel, tn = tile rows; 36: Winograd F(4x4)
    int n_img;
    unsigned blocks;
    bool up;         // some source has fused x2-upsampled segments (LDS kernels only)
    bool norm;       // source 0 is normalised on load (LDS kernels with 16-channel tiles only)
    bool s2;         // source 1 is a 3x3 stride-2 projection (LDS kernels with 64- / 32-channel tiles)
    int nj;          // LDS kernels: 16-channel output sub-tiles per workgroup (4, 2, 1)
    ReduceDesc red;  // valid when a.S > 1
};

int prep_conv(const idh_op &op, PreparedConv &pc) {
    ConvArgs &a = pc.a;
    a = ConvArgs{};
    pc.up = false;
    pc.norm = op.src[0].norm != nullptr;
    pc.s2 = false;
    pc.nj = 4;
    int steps = 0;
    for (int i = 0; i < 2; ++i) {
        const idh_conv_src &s = op.src[i];
        ConvSrc &d = a.s[i];
        d.in = s.in;
        if (!s.in) continue;
        // the K loop reads whole 16-channel blocks: the buffer must be readable (and finite) up to
        // ceil16(Cin) channels per pixel; packed weights are zero there.
        if (!s.w || s.Cin <= 0 || s.cs < (s.up_in[0] ? s.up_c0 : ceil16(s.Cin)) || (s.cs & 3) || (s.ks != 1 && s.ks != 3) ||
            (s.stride != 1 && s.stride != 2) || s.H <= 0 || s.W <= 0)
            return IDH_EINVAL;
        if (ceil16(s.Cin) > kZeroFloats) return IDH_EUNSUPPORTED;
        const int pad = s.ks / 2;
        if ((s.H + 2 * pad - s.ks) / s.stride + 1 != op.Ho || (s.W + 2 * pad - s.ks) / s.stride + 1 != op.Wo)
            return IDH_EINVAL;
        d.w = s.w; d.cs = s.cs; d.H = s.H; d.W = s.W; d.Cin = s.Cin; d.ks = s.ks; d.stride = s.stride;
        d.pad_mode = s.pad_mode; d.cblocks = ceil16(s.Cin) / 16;
        d.up_in[0] = s.up_in[0]; d.up_in[1] = s.up_in[1]; d.up_cs[0] = s.up_cs[0]; d.up_cs[1] = s.up_cs[1];
        d.up_c0 = s.up_c0; d.up_C = s.up_C;
        d.norm = s.norm; d.norm_slope = s.norm_slope; d.norm_act = s.norm_act;
        if (s.norm && (i != 0 || s.up_in[0] || ((uintptr_t)s.norm & 15) || (s.Cin & 15))) return IDH_EUNSUPPORTED;
        if (s.up_in[0]) {
            const int nseg = s.up_in[1] ? 2 : 1;
            if (s.up_c0 < 0 || (s.up_c0 & 15) || s.up_C <= 0 || (s.up_C & 15) || s.up_c0 + nseg * s.up_C != s.Cin || (s.H & 1) || (s.W & 1) ||
                s.stride != 1 || (s.up_c0 > 0 && s.cs < s.up_c0) || s.up_cs[0] < s.up_C || (s.up_cs[0] & 3) ||
                (nseg == 2 && (s.up_cs[1] < s.up_C || (s.up_cs[1] & 3))))
                return IDH_EINVAL;
            pc.up = true;
        }
        steps += s.ks * s.ks * d.cblocks;
    }
    if (!a.s[0].in || !op.out || op.Cout <= 0 || op.N <= 0) return IDH_EINVAL;
    // vector epilogue: 4 consecutive channels per lane -> 16-byte aligned rows
    if ((op.Cout & 3) || (op.out_cs & 3) || ((uintptr_t)op.out & 15) || (op.bias && ((uintptr_t)op.bias & 15)) ||
        (op.res && ((op.res_cs & 3) || ((uintptr_t)op.res & 15))))
        return IDH_EINVAL;
    a.bias = op.bias; a.res = op.res; a.out = op.out; a.ws = op.ws;
    a.res_cs = op.res_cs; a.out_cs = op.out_cs; a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout;
    a.Cout_pad = ceil16(op.Cout);
    const long long M = (long long)op.N * op.Ho * op.Wo;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    a.M = (int)M; a.steps_total = steps; a.act = op.act; a.slope = op.slope;
    a.S = op.split_k > 1 ? op.split_k : 1;
    if (a.S > steps) a.S = steps;
    // A lone 3x3 stride-2 conv (conv1 of a stride-2 BasicBlock) asked onto the LDS-staged kernel (tile_m 8 / 9): it runs as the kernel's
    // stride-2 SECOND source (halo de-interleaved by column parity) behind an empty first source (no chunks: its loops do not execute,
    // its descriptors have zero range).
    if (a.s[0].ks == 3 && a.s[0].stride == 2 && !a.s[1].in && (op.tile_m == 8 || op.tile_m == 9) && a.s[0].pad_mode == IDH_PAD_ZEROS &&
        !a.s[0].up_in[0] && !a.s[0].norm && op.Wo >= kLT_W && (op.Cout % 32) == 0 && (op.tile_n == 0 ? (op.Cout % 64) == 0 : op.tile_n == 2)) {
        a.s[1] = a.s[0];
        ConvSrc e{};
        e.in = a.s[1].in; e.w = a.s[1].w; e.ks = 3; e.stride = 1; e.pad_mode = IDH_PAD_ZEROS;  // cblocks = H = W = cs = 0
        a.s[0] = e;
    }
    // LDS-staged kernel for the dominant shape family: tile_m 8 / 9 request the 8- / 4-row tile,
    // 0 = auto (8-row)
    // LDS kernels: tile_n = output sub-tiles of 16 channels per workgroup (0 = 4 -> 64 channels; 2; 1)
    const int nj = (op.tile_m == 8 || op.tile_m == 9 || op.tile_m == 0) ? (op.tile_n == 0 ? 4 : op.tile_n) : 4;
    const bool lds_ok = a.s[0].ks == 3 && a.s[0].stride == 1 && (nj == 4 || nj == 2 || nj == 1) && (op.Cout % (16 * nj)) == 0 &&
                        (a.s[0].pad_mode == IDH_PAD_ZEROS || (a.s[0].pad_mode == IDH_PAD_REPLICATE && !a.s[1].in)) &&
                        (!a.s[1].in || (a.s[1].ks == 1 && a.s[1].stride == 1) ||
                         (a.s[1].ks == 3 && a.s[1].stride == 2 && a.s[1].pad_mode == IDH_PAD_ZEROS && !a.s[1].up_in[0] && !a.s[0].up_in[0] &&
                          !op.src[0].norm && (nj == 4 || nj == 2))) &&
                        op.Wo >= kLT_W;
    pc.lds_rows = 0;
    // the LDS kernels address source 0 with 32-bit byte offsets inside one image / the packed weights
    const bool lds_fits = (long long)a.s[0].H * a.s[0].W * a.s[0].cs * 4 < (1ll << 31) && 9ll * a.s[0].cblocks * 4 * a.Cout_pad * 16 < (1ll << 31) &&
                          (!a.s[1].in || ((long long)a.s[1].H * a.s[1].W * a.s[1].cs * 4 < (1ll << 31) && 9ll * a.s[1].cblocks * 4 * a.Cout_pad * 16 < (1ll << 31))) &&
                          (long long)op.Ho * op.Wo * op.out_cs * 4 < (1ll << 31) && (!op.res || (long long)op.Ho * op.Wo * op.res_cs * 4 < (1ll << 31));
    if (op.tile_m == IDH_TILE_WINO) {
        // Winograd F(2x2,3x3) kernel (conv_wino.hip): src[0].w holds idh_pack_conv_weight_wino output
        if (!wino_supported(a)) return IDH_EUNSUPPORTED;
        pc.lds_rows = 32;
        pc.tm = op.tile_m;
        pc.tn = 0;
        pc.n_img = op.N;
        pc.blocks = 0;
    } else if (op.tile_m == IDH_TILE_WINO4) {
        // Winograd F(4x4,3x3) kernel (conv_wino4.hip): src[0].w holds idh_pack_conv_weight_wino4 output
        if (!wino4_supported(a)) return IDH_EUNSUPPORTED;
        pc.lds_rows = 36;
        pc.tm = op.tile_m;
        pc.tn = 0;
        pc.n_img = op.N;
        pc.blocks = 0;
    } else if (op.tile_m == IDH_SPLIT_F16X3) {
        // split-precision kernel (conv_split.hip): src[0].w holds idh_pack_conv_weight_split output
        if (!lds_ok || a.s[0].pad_mode != IDH_PAD_ZEROS || (op.Cout % 64) || a.S != 1 || op.Wo < kSplitTile || op.Ho < 1 || (op.tile_n != 0 && op.tile_n != 8 && op.tile_n != 16))
            return IDH_EUNSUPPORTED;
        a.NT = op.Cout / 64;
        pc.lds_rows = 16;
        pc.tm = op.tile_m;
        pc.tn = op.tile_n == 8 ? 8 : 16;  // tile rows
        pc.n_img = op.N;
        pc.blocks = 0;
    } else if (lds_ok && lds_fits && (op.tile_m == 8 || op.tile_m == 0 || op.tile_m == 9)) {
        const int rows = op.tile_m == 9 ? 4 : 8;
        const int chunks = a.s[0].cblocks + (a.s[1].in ? a.s[1].cblocks : 0);
        if (a.S > chunks) a.S = chunks;
        {   // the kernel draws the split boundaries in cost units (9 per 3x3 chunk, 1 per 1x1 chunk, conv3x3_lds_body): more splits than whole
            // 3x3-chunk costs would leave some of them empty (all-zero partials written and reduced for nothing)
            const int w1 = (a.s[1].in && a.s[1].ks == 3) ? 9 : 1;
            const int T = 9 * a.s[0].cblocks + (a.s[1].in ? w1 * a.s[1].cblocks : 0);
            const int smax = T / 9 > 1 ? T / 9 : 1;
            if (a.S > smax) a.S = smax;
        }
        a.NT = op.Cout / (16 * nj);
        pc.nj = nj;
        pc.la = LdsConvArgs{a, (op.Wo + kLT_W - 1) / kLT_W, (op.Ho + rows - 1) / rows};
        const long long blocks = (long long)a.S * op.N * pc.la.tiles_x * pc.la.tiles_y * a.NT;
        if (blocks >= (1ll << 31)) return IDH_EUNSUPPORTED;
        pc.blocks = (unsigned)blocks;
        pc.lds_rows = rows;
        pc.s2 = a.s[1].in && a.s[1].ks == 3;
    } else {
        int tm = op.tile_m, tn = op.tile_n;
        if (tm == 8 || tm == 9) tm = 0;
        const int nsub = a.Cout_pad / 16;
        if (tn == 0) tn = (nsub % 4 == 0) ? 4 : (nsub % 2 == 0 ? 2 : 1);
        if (tm == 0) tm = 4;
        if ((tm != 1 && tm != 2 && tm != 4) || (tn != 1 && tn != 2 && tn != 4) || nsub % tn) return IDH_EINVAL;
        a.MT = (int)((M + 16 * tm - 1) / (16 * tm));
        a.NT = nsub / tn;
        pc.tm = tm; pc.tn = tn;
        const long long waves = (long long)a.MT * a.NT * a.S;
        if ((waves + 3) / 4 >= (1ll << 31)) return IDH_EUNSUPPORTED;
        pc.blocks = (unsigned)((waves + 3) / 4);
    }
    if (pc.up && ((pc.lds_rows != 8 && pc.lds_rows != 4) || pc.nj != 4)) return IDH_EUNSUPPORTED;  // fused upsampling lives in the 64-channel LDS loader
    if (pc.norm && ((pc.lds_rows != 8 && pc.lds_rows != 4) || pc.nj != 1)) return IDH_EUNSUPPORTED;  // normalise-on-load: 16-channel LDS tiles
    if (a.S > 1) {
        if (!op.ws) return IDH_EWORKSPACE;
        pc.red = ReduceDesc{op.ws, op.bias, op.res, op.out, a.M, op.Cout, a.Cout_pad, a.S, op.res_cs, op.out_cs, op.act, op.slope};
    }
    return IDH_OK;
}

int launch_conv(const PreparedConv &pc, hipStream_t st) {
    if (pc.lds_rows == 16) {
        if (t_dry_run) { ++t_launches; return IDH_OK; }
        return launch_conv_split(pc.a, pc.n_img, pc.tm, pc.tn, st);
    }
    if (pc.lds_rows == 32) {
        if (t_dry_run) { ++t_launches; return IDH_OK; }
        return launch_conv_wino(pc.a, pc.n_img, pc.tn, st);
    }
    if (pc.lds_rows == 36) {
        if (t_dry_run) { ++t_launches; return IDH_OK; }
        return launch_conv_wino4(pc.a, pc.n_img, st);
    }
    if (pc.lds_rows == 8 && pc.up) IDH_LAUNCH(conv3x3_lds_up_k<2>, dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 8 && pc.s2 && pc.nj == 4) IDH_LAUNCH((conv3x3_lds_k<2, false, 4, false, true>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 8 && pc.s2) IDH_LAUNCH((conv3x3_lds_k<2, false, 2, false, true>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 4 && pc.s2 && pc.nj == 4) IDH_LAUNCH((conv3x3_lds_k<1, false, 4, false, true>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 4 && pc.s2) IDH_LAUNCH((conv3x3_lds_k<1, false, 2, false, true>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 8 && pc.norm) IDH_LAUNCH((conv3x3_lds_k<2, false, 1, true>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 4 && pc.norm) IDH_LAUNCH((conv3x3_lds_k<1, false, 1, true>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 8 && pc.nj == 2) IDH_LAUNCH((conv3x3_lds_k<2, false, 2>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 8 && pc.nj == 1) IDH_LAUNCH((conv3x3_lds_k<2, false, 1>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 4 && pc.nj == 2) IDH_LAUNCH((conv3x3_lds_k<1, false, 2>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 4 && pc.nj == 1) IDH_LAUNCH((conv3x3_lds_k<1, false, 1>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 8) IDH_LAUNCH((conv3x3_lds_k<2, false>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 4 && pc.up) IDH_LAUNCH(conv3x3_lds_up_k<1>, dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else if (pc.lds_rows == 4) IDH_LAUNCH((conv3x3_lds_k<1, false>), dim3(pc.blocks), dim3(256), 0, st, pc.la);
    else {
#define IDH_CASE(TM_, TN_) \
    if (pc.tm == TM_ && pc.tn == TN_) IDH_LAUNCH((conv_mfma_k<TM_, TN_>), dim3(pc.blocks), dim3(256), 0, st, pc.a);
        IDH_CASE(4, 4) IDH_CASE(2, 4) IDH_CASE(1, 4) IDH_CASE(4, 2) IDH_CASE(2, 2) IDH_CASE(1, 2) IDH_CASE(4, 1)
        IDH_CASE(2, 1) IDH_CASE(1, 1)
#undef IDH_CASE
    }
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

int launch_reduces(const PreparedConv *pcs, int n, hipStream_t st) {
    ReduceGroupArgs g{};
    unsigned cursor = 0;
    for (int i = 0; i < n; ++i) {
        if (pcs[i].a.S <= 1) continue;
        g.start[g.n] = cursor;
        g.d[g.n] = pcs[i].red;
        cursor += (unsigned)idh_cdiv((long long)pcs[i].red.M * pcs[i].red.Cout, 256);
        ++g.n;
    }
    if (g.n == 0) return IDH_OK;
    g.start[g.n] = cursor;
    IDH_LAUNCH(splitk_reduce_group_k, dim3(cursor), dim3(256), 0, st, g);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

// Launch ops[i..j) — all convs of one dependency level that map to the 4-row LDS kernel — as one grid.
int launch_group(const PreparedConv *pcs, int n, hipStream_t st) {
    LdsGroupArgs g{};
    unsigned cursor = 0;
    g.n = n;
    for (int i = 0; i < n; ++i) {
        g.start[i] = cursor;
        g.op[i] = pcs[i].la;
        cursor += pcs[i].blocks;
    }
    g.start[n] = cursor;
    bool any_up = false;
    for (int i = 0; i < n; ++i) any_up = any_up || pcs[i].up;
    if (any_up) IDH_LAUNCH((conv3x3_lds_group_k<1, true>), dim3(cursor), dim3(256), 0, st, g);
    else if (pcs[0].nj == 2) IDH_LAUNCH((conv3x3_lds_group_k<1, false, 2>), dim3(cursor), dim3(256), 0, st, g);
    else if (pcs[0].nj == 1) IDH_LAUNCH((conv3x3_lds_group_k<1, false, 1>), dim3(cursor), dim3(256), 0, st, g);
    else IDH_LAUNCH((conv3x3_lds_group_k<1, false>), dim3(cursor), dim3(256), 0, st, g);
    IDH_CHECK_LAUNCH();
    return launch_reduces(pcs, n, st);
}

// Launch a heterogeneous level (see level_k): kinds[i] = LV_*, pcs[i] prepared (LV_UP2: only la.c's upsample fields).
int launch_level(const PreparedConv *pcs, const int *kinds, int n, hipStream_t st) {
    LevelArgs g{};
    unsigned cursor = 0;
    g.n = n;
    for (int i = 0; i < n; ++i) {
        g.start[i] = cursor;
        g.kind[i] = kinds[i];
        g.op[i] = pcs[i].la;
        if (kinds[i] == LV_MFMA14) g.op[i].c = pcs[i].a;  // prep_conv fills la for the LDS kernels only
        cursor += pcs[i].blocks;
    }
    g.start[n] = cursor;
    bool wide = false;
    for (int i = 0; i < n; ++i) wide = wide || kinds[i] == LV_LDS4;
    if (wide) IDH_LAUNCH(level_k<true>, dim3(cursor), dim3(256), 0, st, g);
    else IDH_LAUNCH(level_k<false>, dim3(cursor), dim3(256), 0, st, g);
    IDH_CHECK_LAUNCH();
    return launch_reduces(pcs, n, st);
}

// Level-launch member kind of a prepared conv, or -1
int level_kind(const PreparedConv &pc) {
    if (pc.up || pc.norm || pc.s2) return -1;
    if (pc.lds_rows == 4 && pc.nj == 4) return LV_LDS4;
    if (pc.lds_rows == 4 && pc.nj == 2) return LV_LDS2;
    if (pc.lds_rows == 0 && pc.tm == 1 && pc.tn == 4) return LV_MFMA14;
    return -1;
}

int check_upsample(const idh_op &op) {
    const idh_conv_src &s = op.src[0];
    if (!s.in || !op.out || (s.Cin & 3) || (s.cs & 3) || (op.out_cs & 3) || op.N <= 0 || s.H <= 0 || s.W <= 0 || s.Cin <= 0) return IDH_EINVAL;
    if ((long long)op.N * s.H * s.W * (s.Cin >> 2) >= (1ll << 31)) return IDH_EUNSUPPORTED;  // (upsample2_body: 32-bit quad index)
    return IDH_OK;
}

// Members of a level launch are small by construction (one frame / low-resolution maps): a member that fills the chip
// for several rounds on its own gains nothing from sharing a grid and keeps its specialised kernel.
constexpr unsigned kLevelMaxBlocks = 512;
constexpr unsigned kLevelUpsampleBlocks = 512;
constexpr long long kLevelUpsampleNatural = 4096;  // one frame's largest upsample (64 ch, 96x128 -> 192x256) is 3072

}  // namespace

extern "C" size_t idh_sizeof_op(void) { return sizeof(idh_op); }

extern "C" size_t idh_packed_weight_floats(int Cout, int Cin, int ks) {
    if (Cout <= 0 || Cin <= 0 || ks <= 0) return 0;
    return (size_t)ks * ks * ceil16(Cin) * ceil16(Cout);
}

extern "C" int idh_pack_conv_weight(const float *w, float *dst, int Cout, int Cin, int ks, void *stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3)) return IDH_EINVAL;
    const long long total = (long long)idh_packed_weight_floats(Cout, Cin, ks);
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_weight_k, dim3(grid), dim3(256), 0, idh_stream(stream), w, dst, Cout, Cin, ks, ceil16(Cin),
                       ceil16(Cout));
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

extern "C" int idh_run_ops(const idh_op *ops, int n, void *stream) {
    if (n < 0 || (n > 0 && !ops)) return IDH_EINVAL;
    hipStream_t st = idh_stream(stream);
    for (int i = 0; i < n; ++i) {
        const idh_op &op = ops[i];
        const idh_conv_src &s = op.src[0];
        switch (op.kind) {
            case IDH_OP_CONV:
            case IDH_OP_UPSAMPLE2: {
                // gather the run of consecutive ops that share a (non-zero) group id: the host scheduler guarantees they
                // are mutually independent (same dependency level).  [i, i+cnt) = the homogeneous prefix of 4-row LDS convs
                // (one conv3x3_lds_group_k grid); [i, i+run) = the mixed run level_k can host when every member is small.
                PreparedConv pcs[kMaxGroup];
                int kinds[kMaxGroup];
                int run = 0;
                for (int j = i; j < n && run < kMaxGroup; ++j) {
                    const idh_op &o = ops[j];
                    if (j > i && (op.group == 0 || o.group != op.group)) break;
                    PreparedConv &pc = pcs[run];
                    if (o.kind == IDH_OP_CONV) {
                        const int rc = prep_conv(o, pc);
                        if (rc != IDH_OK) return rc;
                        kinds[run] = level_kind(pc);
                    } else if (o.kind == IDH_OP_UPSAMPLE2) {
                        const int rc = check_upsample(o);
                        if (rc != IDH_OK) return rc;
                        const idh_conv_src &us = o.src[0];
                        pc = PreparedConv{};
                        ConvArgs &c = pc.la.c;
                        c.s[0].in = us.in; c.s[0].H = us.H; c.s[0].W = us.W; c.s[0].Cin = us.Cin; c.s[0].cs = us.cs;
                        c.out = o.out; c.out_cs = o.out_cs; c.M = o.N; c.S = 1;
                        pc.a = c;
                        const long long tot = (long long)o.N * 4 * us.H * us.W * (us.Cin >> 2);
                        const long long natural = idh_cdiv(tot, 256);  // (output float4s: the measure the thresholds were tuned in)
                        pc.blocks = (unsigned)std::min<long long>(idh_cdiv(tot / 4, 256), kLevelUpsampleBlocks);  // one thread per 2x2 quad
                        // a large upsample (batch >= 2 at the top resolutions) runs at HBM speed on its own grid of thousands
                        // of blocks; squeezed into 512 grid-stride blocks it is slower than the launch it saves
                        kinds[run] = natural <= kLevelUpsampleNatural ? LV_UP2 : -1;
                    } else {
                        break;
                    }
                    ++run;
                }
                int cnt = 0;  // homogeneous prefix
                if (op.kind == IDH_OP_CONV && pcs[0].lds_rows == 4 && !pcs[0].norm && !pcs[0].s2) {
                    cnt = 1;
                    while (cnt < run && ops[i + cnt].kind == IDH_OP_CONV && pcs[cnt].lds_rows == 4 && pcs[cnt].nj == pcs[0].nj &&
                           (!pcs[cnt].up || pcs[0].nj == 4) && !pcs[cnt].norm && !pcs[cnt].s2)
                        ++cnt;
                }
                int mix = 0;  // mixed prefix of small members
                while (mix < run && kinds[mix] >= 0 && pcs[mix].blocks <= kLevelMaxBlocks) ++mix;
                int wg = 0;  // prefix of Winograd convs with the same kind of second source: one persistent grid
                if (op.kind == IDH_OP_CONV && pcs[0].lds_rows == 32) {
                    wg = 1;
                    while (wg < run && wg < wino_max_group() && ops[i + wg].kind == IDH_OP_CONV && pcs[wg].lds_rows == 32 &&
                           (pcs[wg].a.s[1].in != nullptr) == (pcs[0].a.s[1].in != nullptr))
                        ++wg;
                }
                int rc, used;
                if (wg > 1) {
                    if (t_dry_run) {
                        ++t_launches;
                        rc = IDH_OK;
                    } else {
                        const ConvArgs *as[kMaxGroup];
                        int ns[kMaxGroup];
                        for (int j = 0; j < wg; ++j) { as[j] = &pcs[j].a; ns[j] = pcs[j].n_img; }
                        rc = launch_conv_wino_group(as, ns, wg, st);
                        if (rc == IDH_EUNSUPPORTED) {  // (a device whose resident grid is not whole XCD octets): one launch per op
                            rc = IDH_OK;
                            for (int j = 0; j < wg && rc == IDH_OK; ++j) rc = launch_conv(pcs[j], st);
                        }
                    }
                    used = wg;
                } else if (mix > cnt && mix > 1) {
                    rc = launch_level(pcs, kinds, mix, st);
                    used = mix;
                } else if (cnt > 1) {
                    rc = launch_group(pcs, cnt, st);
                    used = cnt;
                } else if (op.kind == IDH_OP_CONV) {
                    rc = launch_conv(pcs[0], st);
                    if (rc == IDH_OK) rc = launch_reduces(pcs, 1, st);
                    used = 1;
                } else {
                    const long long quads = (long long)op.N * s.H * s.W * (s.Cin >> 2);  // one thread per 2x2 output quad x 4 channels
                    int grid = idh_cdiv(quads, 256);
                    if (grid > 32768) grid = 32768;
                    IDH_LAUNCH(upsample2_k, dim3(grid), dim3(256), 0, st, s.in, op.out, op.N, s.H, s.W, s.Cin, s.cs, op.out_cs);
                    IDH_CHECK_LAUNCH();
                    rc = IDH_OK;
                    used = 1;
                }
                if (rc != IDH_OK) return rc;
                i += used - 1;
                break;
            }
            case IDH_OP_UPSAMPLE2_NEAREST: {
                if (!s.in || !op.out || (s.Cin & 3) || (s.cs & 3) || (op.out_cs & 3) || op.N <= 0) return IDH_EINVAL;
                const long long tot = (long long)op.N * 4 * s.H * s.W * (s.Cin >> 2);
                int grid = idh_cdiv(tot, 256);
                if (grid > 8192) grid = 8192;
                IDH_LAUNCH(upsample2_nearest_k, dim3(grid), dim3(256), 0, st, s.in, op.out, op.N, s.H, s.W, s.Cin, s.cs, op.out_cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_NCHW_TO_NHWC: {
                // consecutive imports that share a (non-zero) group id are mutually independent (one dependency level): one grid
                ImportGroupArgs g{};
                unsigned cursor = 0;
                int run = 0;
                for (int j = i; j < n && run < kMaxImport; ++j) {
                    const idh_op &o = ops[j];
                    if (o.kind != IDH_OP_NCHW_TO_NHWC || (j > i && (op.group == 0 || o.group != op.group))) break;
                    const idh_conv_src &os = o.src[0];
                    if (!os.in || !o.out || o.N <= 0 || o.N > 65535 || os.H <= 0 || os.W <= 0 || os.Cin <= 0) return IDH_EINVAL;
                    const int HWj = os.H * os.W;
                    const long long blocks = (long long)idh_cdiv(HWj, 64) * idh_cdiv(os.Cin, 32) * o.N;
                    if (blocks > kLevelMaxBlocks * 8ll || cursor + blocks >= (1ll << 31)) break;  // a large import keeps its own 3-D grid
                    g.d[run].src = os.in; g.d[run].dst = o.out; g.d[run].C = os.Cin; g.d[run].HW = HWj; g.d[run].out_cs = o.out_cs;
                    g.d[run].nbx = idh_cdiv(HWj, 64); g.d[run].nby = idh_cdiv(os.Cin, 32);
                    g.start[run] = cursor;
                    cursor += (unsigned)blocks;
                    ++run;
                }
                if (run > 1) {
                    g.n = run;
                    g.start[run] = cursor;
                    IDH_LAUNCH(import_nchw_group_k, dim3(cursor), dim3(256), 0, st, g);
                    IDH_CHECK_LAUNCH();
                    i += run - 1;
                    break;
                }
                if (!s.in || !op.out || op.N <= 0 || op.N > 65535) return IDH_EINVAL;
                const int HW = s.H * s.W;
                IDH_LAUNCH(import_nchw_k, dim3(idh_cdiv(HW, 64), idh_cdiv(s.Cin, 32), op.N), dim3(256), 0, st, s.in,
                                   op.out, s.Cin, HW, op.out_cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_NHWC_TO_NCHW: {
                if (!s.in || !op.out || op.N <= 0 || op.N > 65535) return IDH_EINVAL;
                const int HW = s.H * s.W;
                IDH_LAUNCH(export_nchw_k, dim3(idh_cdiv(HW, 64), idh_cdiv(s.Cin, 32), op.N), dim3(256), 0, st, s.in,
                                   op.out, s.Cin, HW, s.cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_POINTWISE_HEAD: {
                if (!s.in || !s.w || !op.out || (s.Cin & 3) || (s.cs & 3)) return IDH_EINVAL;
                const long long M = (long long)op.N * s.H * s.W;
                int grid = idh_cdiv(M, 256);
                if (grid > 8192) grid = 8192;
                IDH_LAUNCH(pointwise_head_k, dim3(grid), dim3(256), 0, st, s.in, s.w, op.bias, op.out, op.ws, M, s.Cin,
                                   s.cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_POINTWISE_NCHW: {
                // src[0].in = (N, 64, H, W) dense NCHW, src[0].w = idh_pack_conv_weight(128, 64, 1), out = NHWC slice
                if (!s.in || !s.w || !op.out || op.N <= 0 || s.H <= 0 || s.W <= 0 || (op.out_cs & 3) || ((uintptr_t)op.out & 15) ||
                    (op.bias && ((uintptr_t)op.bias & 15)))
                    return IDH_EINVAL;
                if (s.Cin != kPwCin || op.Cout != kPwCout || op.out_cs < kPwCout || (long long)s.H * s.W * kPwCin * 4 >= (1ll << 31))
                    return IDH_EUNSUPPORTED;
                PwNchwArgs pa{};
                pa.in = s.in; pa.w = s.w; pa.bias = op.bias; pa.out = op.out;
                pa.HW = s.H * s.W; pa.out_cs = op.out_cs;
                pa.tiles_per_img = idh_cdiv(pa.HW, kPwTile);
                pa.tiles = (long long)op.N * pa.tiles_per_img;
                const int grid = (int)std::min<long long>(pa.tiles, 256 * 4 * 4);
                IDH_LAUNCH(pointwise_nchw_k, dim3(grid), dim3(256), 0, st, pa);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_POINTWISE_UP: {
                // src[0] = x (N, H, W, Cin) + idh_pack_conv_weight(Cout, Cin, 1); src[1].in = the half-resolution map (N, H/2, W/2, Cout), src[1].cs its stride
                const idh_conv_src &lo = op.src[1];
                if (!s.in || !s.w || !lo.in || !op.out || op.N <= 0 || s.H <= 0 || s.W <= 0 || (s.H & 1) || (s.W & 1) || (s.cs & 3) || (lo.cs & 3) || (op.out_cs & 3) ||
                    s.cs < s.Cin || lo.cs < op.Cout || op.out_cs < op.Cout || ((uintptr_t)s.in & 15) || ((uintptr_t)lo.in & 15) || ((uintptr_t)op.out & 15) ||
                    ((uintptr_t)s.w & 15) || (op.bias && ((uintptr_t)op.bias & 15)) || lo.H != s.H / 2 || lo.W != s.W / 2)
                    return IDH_EINVAL;
                if (s.Cin != op.Cout || (s.Cin != 64 && s.Cin != 128) || (long long)s.H * s.W >= (1ll << 31)) return IDH_EUNSUPPORTED;
                PwUpArgs pa{};
                pa.in = s.in; pa.w = s.w; pa.bias = op.bias; pa.low = lo.in; pa.out = op.out;
                pa.H = s.H; pa.W = s.W; pa.in_cs = s.cs; pa.low_cs = lo.cs; pa.out_cs = op.out_cs;
                pa.tiles_per_img = idh_cdiv(s.H * s.W, kPwTile);
                pa.tiles = (long long)op.N * pa.tiles_per_img;
                const int grid = (int)std::min<long long>(pa.tiles, 256 * 8);
                if (s.Cin == 64) IDH_LAUNCH((pointwise_up_k<4, 4>), dim3(grid), dim3(256), 16 * 64 * sizeof(f32x4), st, pa);
                else IDH_LAUNCH((pointwise_up_k<8, 8>), dim3(grid), dim3(256), 32 * 128 * sizeof(f32x4), st, pa);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_COPY: {
                if (!s.in || !op.out || (s.Cin & 3) || (s.cs & 3) || (op.out_cs & 3) || op.N <= 0) return IDH_EINVAL;
                const long long npix = (long long)op.N * s.H * s.W;
                int grid = idh_cdiv(npix * (s.Cin >> 2), 256);
                if (grid > 8192) grid = 8192;
                IDH_LAUNCH(copy_nhwc_k, dim3(grid), dim3(256), 0, st, s.in, op.out, npix, s.Cin, s.cs, op.out_cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_INSTNORM: {
                const int C = s.Cin, HW = s.H * s.W;
                if (!s.in || !op.ws || C <= 0 || (C & 3) || 256 % (C >> 2) || (s.cs & 3) || (op.out && (op.out_cs & 3)) ||
                    op.N <= 0 || op.N > 65535)
                    return IDH_EINVAL;
                const int nchunks = idh_cdiv(HW, kInChunk);
                // C >= 64: the chunk is summed as 1024 / (C/4) rows at either block size (bit-identical statistics at every batch size)
                const int sthreads = ((long long)nchunks * op.N < 512 && C >= 64) ? 1024 : 256;
                if (C >= 64 && sthreads == 256)
                    IDH_LAUNCH(instnorm_stats_k<4>, dim3(nchunks, op.N), dim3(256), 1024 * 8 * sizeof(float), st, s.in, s.cs, C, HW, nchunks, op.ws);
                else
                    IDH_LAUNCH(instnorm_stats_k<1>, dim3(nchunks, op.N), dim3(sthreads), sthreads * 8 * sizeof(float), st, s.in, s.cs, C, HW,
                                       nchunks, op.ws);
                IDH_CHECK_LAUNCH();
                float *stats = op.ws + (size_t)op.N * nchunks * 2 * C;
                IDH_LAUNCH(instnorm_finalize_k, dim3(op.N), dim3(256), 0, st, op.ws, s.in, s.cs, C, HW, nchunks, stats);
                IDH_CHECK_LAUNCH();
                if (!op.out) break;  // statistics only: the consumer conv normalises on load (idh_conv_src.norm)
                const long long total = (long long)op.N * HW * (C >> 2);
                int gx = idh_cdiv(total, 256 * 4);  // ~4 float4 per thread
                if (gx > 16384) gx = 16384;
                IDH_LAUNCH(instnorm_apply_k, dim3(gx), dim3(256), 0, st, s.in, s.cs, op.out, op.out_cs, C, HW, total, stats, op.act,
                                   op.slope);
                IDH_CHECK_LAUNCH();
                break;
            }
            default:
                return IDH_EINVAL;
        }
    }
    return IDH_OK;
}

extern "C" int idh_count_launches(const idh_op *ops, int n) {
    t_dry_run = true;
    t_launches = 0;
    const int rc = idh_run_ops(ops, n, nullptr);
    t_dry_run = false;
    return rc == IDH_OK ? t_launches : rc;
}
