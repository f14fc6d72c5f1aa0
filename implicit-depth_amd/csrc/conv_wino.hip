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
#include <type_traits>

#include "conv_args.h"
#include "../../include/idh_ops.h"

using namespace idh_conv;

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kWinoCH = 8;      // input channels per K step (the packed weight layout depends on it)
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
// (base, bytes) of a buffer kept as plain scalars: a source picked at run time is two s_cselects, not a branch
struct BufRef {
    const float *p;
    int bytes;
};
__device__ __forceinline__ void dma16(BufRef b, f32x4 *dst, int voff, int soff) {
    dma16(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b.p), 0, b.bytes, 0x00020000), dst, voff, soff);
}

// Developer build (-DIDH_ABL_WINO_TRACE, tools/trace_wino.sh): every wave logs s_memtime at its phase boundaries into
// ConvArgs.ws (64 x 8 bytes per wave: [0] entry, [1] first DMA issued, [2] first barrier passed, [3+2c] K step c computed,
// [4+2c] its barrier passed, [62] before the epilogue stores retire, [63] HW_ID).  Not part of the product library.
#ifdef IDH_ABL_WINO_TRACE
#define WINO_TRACE(idx) do { if (lane == 0) trace[(idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define WINO_TRACE(idx) do { } while (0)
#endif

struct WinoArgs {
    ConvArgs c;
    int tiles_x, tiles_y;
    int tiles;  // N * tiles_y * tiles_x * NT
};

// SRC2: a second source, the 1x1 projection of another tensor (BasicBlock's conv2(h) + downsample(x), layers.py:86-92), is
// accumulated in the OUTPUT domain: after the Winograd steps the tile runs "P steps" of 16 channels each — the same halo
// region of x (4 planes, only the patch centres are read) and a 2 KiB panel of the ordinary packed 1x1 weights in the same
// two LDS stages — whose MFMAs add W1 . x to the 2x2 output pixels directly (the registers that hold the residual tile).
// float4 slots of one of the two LDS stages
template <int WAVES, int NCO, int CH, bool SRC2>
constexpr int wino_stage_slots() {
    constexpr int halo = ((CH / 4) * wino_plane(WAVES) + 63) / 64 * 64;
    constexpr int s3 = halo + 64 * NCO * 4 * (CH / 4), sp = 8 * 32 * 2 * WAVES + 64 * 2 * NCO;
    return SRC2 && sp > s3 ? sp : s3;
}

// The persistent tile loop of ONE convolution as seen by workgroup `vblock` of `vgrid` (the launch's own block index for a
// single op; rotated per op in a grouped launch so that small ops land on different workgroups).
template <int WAVES, int NCO, int CH, bool SRC2>
__device__ __forceinline__ void wino_tiles(const WinoArgs &wa, f32x4 *lds, const unsigned vblock, const unsigned vgrid) {
    constexpr int KS = CH / 4;   // MFMA k-steps per K step = consecutive channels per lane
    constexpr int NP = CH / 4;   // 4-channel planes of the halo per K step
    typedef float vec __attribute__((ext_vector_type(KS)));
    constexpr int kRows = 2 * WAVES;
    constexpr int kPlane = wino_plane(WAVES);  // halo texels: (kRows + 2) rows x 2 column parities x 17
    // Halo in LDS, pixel-major: texel p = (2 hy + column parity) * 17 + column / 2 owns the NP consecutive 16-byte slots
    // NP p .. (its CH channels), plane q in slot NP p + (q ^ swz), swz = (column / 2 >> 3) & 1 — so that an LDS-DMA piece is the
    // contiguous 32 bytes of 32 texels (32 cache lines per piece instead of 64 scattered 16-byte granules) while the 32 lanes
    // of a ds_read_b64 group (texel columns n .. n+15 of one row) still cover all 64 banks.
    // Pieces (1 KiB each) of one K step: kHaloPieces of the halo — wave w copies pieces w, w + WAVES, .. (a wave with one fewer
    // repeats its first: same bytes, same place) — and kPanel per wave of the weight panel: one offset VGPR per halo piece of
    // the wave plus one for the panel
    static_assert(NP == 2, "halo swizzle written for 8-channel K steps");
    constexpr int kHaloPieces = (NP * kPlane + 63) / 64;
    constexpr int kHP = (kHaloPieces + WAVES - 1) / WAVES;  // halo pieces per wave
    constexpr int kPanelPieces = NCO * 4 * KS;
    static_assert(kPanelPieces % WAVES == 0, "panel pieces must split evenly over the waves");
    constexpr int kPanel = kPanelPieces / WAVES;
    constexpr int kTPW = kHP + kPanel;  // DMA pieces per wave and K step
    static_assert(CH == 8, "K steps of 8 channels: every tile has >= 2 steps (the first stores the previous tile's outputs, the last loads the residual)");
    static_assert(kTPW <= 8 && 4 * NCO <= 8, "one DMA piece per even position group, one output store per odd group");
    constexpr int kHalo = 64 * kHaloPieces;
    constexpr int kTight = 32 * kRows;                   // float4 slots of one 4-channel plane of the tile's own pixels (no halo)
    constexpr int kStage3 = kHalo + 64 * kPanelPieces;   // float4 slots of a Winograd step's stage
    constexpr int kStageP = 8 * kTight + 64 * 2 * NCO;   // ... of a P step's (SRC2): 32 channels of the tile + 2 x NCO panel pieces
    constexpr int kStage = SRC2 && kStageP > kStage3 ? kStageP : kStage3;
    static_assert(kStage == wino_stage_slots<WAVES, NCO, CH, SRC2>(), "stage size used by the kernels' LDS declaration");

    const ConvArgs &a = wa.c;
    const ConvSrc &s = a.s[0];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, h = lane >> 4;
#ifdef IDH_ABL_WINO_TRACE
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(a.ws) + ((size_t)blockIdx.x * WAVES + wave) * 64;
    int tr_i = 1;
    WINO_TRACE(0);
    if (lane == 0) trace[63] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
#endif
    const int nS = s.cblocks * (16 / CH);  // K steps per tile
    const int nCB = a.Cout_pad / 16;
    const BufRef rsW{s.w, s.cblocks * nCB * 16384};

    // ---- persistent workgroup: tiles t0, t0 + stride, ... < t_end.  The hardware places block b on XCD b % 8; every XCD
    // gets one contiguous range of tiles and its resident workgroups walk it side by side, so the workgroups that
    // share an input tile (the NT channel tiles, neighbouring halos) hit the same L2 at about the same time.
    const int T = wa.tiles;
    int t_cur, t_end, t_stride;
    if ((vgrid & 7) == 0) {
        const int xcd = vblock & 7;
        t_stride = vgrid >> 3;
        t_cur = (int)((long long)T * xcd / 8) + (int)(vblock >> 3);
        t_end = (int)((long long)T * (xcd + 1) / 8);
    } else {
        t_cur = vblock; t_end = T; t_stride = vgrid;
    }
    if (t_cur >= t_end) return;

    // tile -> (image, tile row, tile column, channel tile); channel tile fastest
    int y0, x0, cb0, img;
    auto decode = [&](int t) {
        unsigned blk = (unsigned)t;
        const int nt = blk % a.NT; blk /= a.NT;
        const int tx = blk % wa.tiles_x; blk /= wa.tiles_x;
        const int ty = blk % wa.tiles_y;
        img = blk / wa.tiles_y;
        y0 = ty * kRows; x0 = tx * kWinoTileW; cb0 = nt * NCO;
    };
    // ---- LDS-DMA descriptors of the tile whose K steps are being fetched --------------------------------------------
    BufRef rsA;
    int voffH[kHP], panel_so;
    const int voffP = 16 * lane;
    auto halo_piece = [&](int k) { return wave + WAVES * k < kHaloPieces ? wave + WAVES * k : wave; };  // k-th halo piece of this wave
    auto halo_voff = [&](int j, int lane) {  // byte offset of this lane's 16 bytes of piece j: texel p = 32 j + lane / 2, plane (lane & 1) ^ swz
        const int pt = 32 * j + (lane >> 1);
        const int row = pt / kWinoHalf, hxh = pt - row * kWinoHalf;
        const int q = (lane & 1) ^ ((hxh >> 3) & 1);
        const int iy = y0 - 1 + (row >> 1), ix = x0 - 1 + 2 * hxh + (row & 1);
        const bool ok = (pt < kPlane) & ((unsigned)iy < (unsigned)s.H) & ((unsigned)ix < (unsigned)s.W);
        return ok ? (iy * s.W + ix) * s.cs * 4 + 16 * q : kOob;  // out of the descriptor's range = zero padding
    };
    auto set_fetch_tile = [&]() {  // from (y0, x0, cb0, img)
        rsA = BufRef{s.in + (size_t)img * s.H * s.W * s.cs, s.H * s.W * s.cs * 4};
        // (the lane's row / column split of each piece is re-derived per tile from an opaque copy of the lane id: hoisted out of the tile
        // loop these ~10 values do not fit beside the accumulators and come back as scratch reloads in the tile header — each of which
        // waits, through the in-order vmcnt, for every LDS-DMA piece in flight)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
#pragma unroll
        for (int k = 0; k < kHP; ++k) voffH[k] = halo_voff(halo_piece(k), lane_o);
        panel_so = (cb0 * (4 * KS) + wave * kPanel) * 1024;
    };
    // ---- second source (P steps of 32 channels): the tile's own 32 x kRows pixels, pixel-major — slot 8 P + (plane ^ swz(P)),
    // P = (2 row + column parity) * 16 + column / 2, swz(P) = (P >> 1) & 7 — so that one LDS-DMA piece (64 slots) is the
    // 128 contiguous bytes of 8 pixels (8 cache lines per piece: plane-major pieces of 64 scattered 16-byte granules kept
    // the texture addresser busier than the matrix pipe) while the 16 lanes of a ds_read_b128 group still hit 16 different
    // bank groups (the XOR spreads a plane over the 8 slots of consecutive pixel pairs).  32 pixel pieces (8 per wave; the
    // piece only moves the scalar offset) and 2 x NCO panel pieces of the ordinary packed 1x1 weights (one per wave).
    static_assert(!SRC2 || (kTight == 64 * WAVES && 2 * NCO == WAVES), "P-step pieces: eight of the pixels and one of the panel per wave");
    constexpr int kTPW2 = SRC2 ? 9 : 0;
    const ConvSrc &s1 = a.s[1];
    const int nS2 = SRC2 ? (s1.cblocks + 1) / 2 : 0;  // P steps per tile
    BufRef rsA2 = rsW, rsW2 = rsW;
    int voffT = 0, voffP2 = 0;
    if constexpr (SRC2) {
        rsW2 = BufRef{s1.w, s1.cblocks * 4 * a.Cout_pad * 16};
        voffP2 = ((lane >> 4) * a.Cout_pad + (lane & 15)) * 16;  // idh_pack_conv_weight(ks = 1): [ci / 4][co][4]
    }
    int tile2_so = 0;  // byte offset of the tile's first pixel in its image (wave-uniform)
    auto set_fetch_tile2 = [&]() {  // from (y0, x0, img); same map size as source 0
        if constexpr (SRC2) {
            rsA2 = BufRef{s1.in + (size_t)img * s1.H * s1.W * s1.cs, s1.H * s1.W * s1.cs * 4};
            // lane L of a piece: pixel column 2 (L >> 3) (+ the piece's parity / half / row, in the scalar offset), plane (L & 7) ^ swz.
            // Rows below the image are past the descriptor's range (zeros); columns right of it alias the next row's pixels and
            // feed only tiles whose outputs are never stored (a 1x1 source mixes no pixels)
            tile2_so = (y0 * s1.W + x0) * s1.cs * 4;
        }
    };
    if constexpr (SRC2) voffT = 2 * (lane >> 3) * s1.cs * 4;
    // k-th DMA piece of this wave for the NEXT step into `stage`: Winograd step c of the fetch tile, or (p, SRC2 only) P step c
    // of the tile in flight.  Branch-free: both candidates are formed and selected (a uniform branch per piece costs more
    // in the K loop than the selects); wave-uniform operands forced into SGPRs
    auto issue_next = [&](int k, bool p, int c, int stage, int cbf) {
        int so, lo, vo;
        bool halo = true;
        const int k0 = k < kTPW ? k : kTPW - 1;  // (pieces kTPW.. exist for P steps only: the Winograd candidate repeats its last piece)
        if (k0 < kHP) {
            vo = voffH[k0]; so = 4 * CH * c; lo = 64 * halo_piece(k0);
        } else {
            const int i = k0 - kHP;
            halo = false;
            vo = voffP; so = panel_so + (c * nCB * (4 * KS) + i) * 1024; lo = kHalo + 64 * (wave * kPanel + i);
        }
#ifdef IDH_ABL_WINO_NOHALO
        if (halo && c + stage > 0) return;
#endif
#ifdef IDH_ABL_WINO_NOPANEL
        if (!halo && c + stage > 0) return;
#endif
        BufRef rs = halo ? rsA : rsW;
        if constexpr (SRC2) {
            int so1, lo1, vo1;
            bool halo1 = true;
            if (k < 8) {  // pixel piece t = 8 wave + k: row t >> 2, column parity (t >> 1) & 1, columns 16 (t & 1) ..; plane (lane & 7) ^ swz
                const int t = 8 * wave + k;
                const int plane = (lane & 7) ^ ((4 * (t & 1) + (lane >> 4)) & 7);  // swz(P) = (P >> 1) & 7, P & 15 = 8 (t & 1) + (lane >> 3)
                // the upper half of an odd last step lies past the packed weights: zeros
                vo1 = (plane >= 4 && 2 * c + 1 >= s1.cblocks) ? kOob : voffT + 16 * plane;
                so1 = tile2_so + (((t >> 2) * s1.W + 16 * (t & 1) + ((t >> 1) & 1)) * s1.cs + 32 * c) * 4;
                lo1 = 64 * t;
            } else {      // panel piece (16-channel half wave / NCO, output block wave % NCO)
                const int hh = wave / NCO, cbw = wave % NCO;
                halo1 = false;
                vo1 = voffP2; so1 = (4 * (2 * c + hh) * a.Cout_pad + 16 * (cbf + cbw)) * 16; lo1 = 8 * kTight + 64 * wave;
            }
            // bit blends, not ?: — hipcc turns selects of this much arithmetic back into a branch per piece
            const int m = -(int)p;
            const unsigned long long m64 = (unsigned long long)(long long)m;
            vo ^= (vo ^ vo1) & m; so ^= (so ^ so1) & m; lo ^= (lo ^ lo1) & m;
            const BufRef rs1 = halo1 ? rsA2 : rsW2;
            const unsigned long long p0 = (unsigned long long)rs.p, p1 = (unsigned long long)rs1.p;
            rs.p = reinterpret_cast<const float *>(p0 ^ ((p0 ^ p1) & m64));
            rs.bytes ^= (rs.bytes ^ rs1.bytes) & m;
        }
        dma16(rs, lds + __builtin_amdgcn_readfirstlane(stage * kStage + lo), vo, __builtin_amdgcn_readfirstlane(so));
    };

    f32x4 acc[16][NCO];
    // finished outputs of the previous tile, stored one 16-byte vector at a time under the next tile's first K step (a burst
    // of 8 stores per wave at the end of a tile stalls the wave ~3k cycles on the store path)
    f32x4 ost[2][2][NCO];  // ... and, under a tile's LAST K step, its residual tile (nS >= 2: never the step that stores)
    int voffS[2][2];
    __amdgpu_buffer_rsrc_t rsS;
    bool pending = false;

    // this lane's KS channels of a texel: plane q = (KS h) / 4, byte (KS h % 4) * 4 of the 16-byte slot NP p + (q ^ swz(column / 2)).
    // Patch origin = texel (row 4 wave, column n); patch column x sits at column n + (x >> 1) of parity x & 1: two lane bases
    const int pq = (KS * h) >> 2, pb8 = ((KS * h) & 3) * 4;
    const int patch0 = (NP * ((4 * wave) * kWinoHalf + n) + (pq ^ ((n >> 3) & 1))) * 16 + pb8;              // bytes, x >> 1 == 0
    const int frag0 = kHalo * 16 + (h * 16 + n) * (4 * KS);                                                   // bytes

    // One K step: 16 position groups of KS*NCO MFMAs, in the order xi = 1, 2, 0, 3 (rows 1 and 2 of the patch feed the
    // first eight groups, row 0 / row 3 are read while those run).  Every LDS read is issued kAhead groups ahead of its
    // first use and the DMA pieces of the NEXT step are spread over the first groups, so that neither the LDS latency nor
    // the ~100-cycle issue cost of an LDS-DMA instruction opens a gap in the matrix pipe; sched_barrier keeps the compiler
    // from sinking the reads back to their uses.
    constexpr int kAhead = CH == 16 ? 2 : 3;
    auto compute = [&](int stage, bool pnext, int cnext, int snext, int cbf, bool flush) {  // flush: the previous tile's outputs leave under this step
        // volatile LDS pointers: hipcc otherwise pairs the 8-byte reads into ds_read2_b64 (half rate, 2-way bank conflicts)
        typedef const __attribute__((address_space(3))) volatile char lds_cchar;
        typedef const __attribute__((address_space(3))) volatile vec lds_cvec;
        // (patch1 re-derived per step from an opaque copy of patch0's inputs: one address register fewer across the tile loop — a
        // scratch reload inside the step would wait for every LDS-DMA piece in flight)
        int n_o = n;
        asm volatile("" : "+v"(n_o));
        const int patch1 = patch0 + (NP + ((pq ^ (((n_o + 1) >> 3) & 1)) - (pq ^ ((n_o >> 3) & 1)))) * 16;
        lds_cchar *sH0 = (lds_cchar *)(lds + stage * kStage) + patch0, *sH1 = (lds_cchar *)(lds + stage * kStage) + patch1;
        lds_cchar *sW = (lds_cchar *)(lds + stage * kStage) + frag0;
        vec d[4][4], A[kAhead + 1][NCO], r[4];
        auto rd_frag = [&](int g) {
            const int p = ((g >> 2) == 0 ? 4 : (g >> 2) == 1 ? 8 : (g >> 2) == 2 ? 0 : 12) + (g & 3);
#pragma unroll
            for (int j = 0; j < NCO; ++j) A[g % (kAhead + 1)][j] = *(lds_cvec *)(sW + (j * 16 + p) * 64 * (4 * KS));
        };
        auto rd_row = [&](int y) {
#pragma unroll
            for (int x = 0; x < 4; ++x) d[y][x] = *(lds_cvec *)(((x >> 1) ? sH1 : sH0) + NP * (2 * y + (x & 1)) * kWinoHalf * 16);
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
            if (g == 5) rd_row(0);   // needed at g = 8; dead after it ...
            if (g == 9) rd_row(3);   // ... so that row 3 (needed at g = 12) can take its registers
#ifndef IDH_ABL_WINO_NODMA
            if ((g & 1) == 0 && (g >> 1) < kTPW) issue_next(g >> 1, pnext, cnext, snext, cbf);  // even groups: one DMA piece
            if (SRC2 && (g == 1 || g == 3) && kTPW + (g >> 1) < kTPW2) {  // a P step has more pieces than a Winograd step (odd groups
                if (pnext) issue_next(kTPW + (g >> 1), true, cnext, snext, cbf);  // carry output stores only in a tile's FIRST step)
            }
#endif
#ifndef IDH_ABL_WINO_NOSTORE
            if ((g & 1) == 1 && (g >> 1) < 4 * NCO && flush) {  // odd groups: one 16-byte store of the previous tile's outputs
                const int k = g >> 1;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ost[k >> 1 & 1][k & 1][k >> 2]), rsS, voffS[k >> 1 & 1][k & 1] + 64 * (k >> 2), 0, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifdef IDH_ABL_WINO_SCALARXF
            // scalar adds instead of v_pk_add_f32 (tools/micro/mfma_pk_valu.hip: a packed fp32 instruction costs ~3x a scalar one beside fp32 MFMAs)
            auto vadd = [](vec a, vec b) { vec o; for (int e = 0; e < KS; ++e) { float t = a[e] + b[e]; asm("" : "+v"(t)); o[e] = t; } return o; };
            auto vsub = [](vec a, vec b) { vec o; for (int e = 0; e < KS; ++e) { float t = a[e] - b[e]; asm("" : "+v"(t)); o[e] = t; } return o; };
#else
            auto vadd = [](vec a, vec b) { return a + b; };
            auto vsub = [](vec a, vec b) { return a - b; };
#endif
#ifndef IDH_ABL_WINO_NOXFORM
            if (nu == 0) {  // B^T d: rows combined for this xi
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    r[x] = xi == 0 ? vsub(d[0][x], d[2][x]) : xi == 1 ? vadd(d[1][x], d[2][x]) : xi == 2 ? vsub(d[2][x], d[1][x]) : vsub(d[1][x], d[3][x]);
            }
#endif
#ifdef IDH_ABL_WINO_NOXFORM
            const vec v = d[xi][nu];
#else
            const vec v = nu == 0 ? vsub(r[0], r[2]) : nu == 1 ? vadd(r[1], r[2]) : nu == 2 ? vsub(r[2], r[1]) : vsub(r[1], r[3]);
#endif
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int j = 0; j < NCO; ++j)
                    acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g % (kAhead + 1)][j][k], v[k], acc[p][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // One P step (SRC2): 32 channels of the 1x1 source.  Lane (n, h) reads channels 4h..4h+3 and 16+4h.. of the four pixels of
    // its tile (ds_read_b128 from planes h and 4+h) and the W1 fragments of both halves and each output block; 8 MFMA k-steps
    // per pixel and block.
    auto compute_p = [&](int stage, bool pnext, int cnext, int snext, int cbf) {
        // pixel P = ((2 wave + i) * 2 + j) * 16 + n -> slot 8 P + (plane ^ swz), swz = (n >> 1) & 7; plane h (channels 4h..) / 4 + h
        // (recomputed per step from an opaque copy of the lane id: hoisted out of the tile loop these addresses only add to the
        // register pressure of the Winograd steps and come back as scratch loads)
        int ln_o = lane;
        asm volatile("" : "+v"(ln_o));
        const int n_o = ln_o & 15, h_o = ln_o >> 4;
        const f32x4 *sH = lds + stage * kStage + 8 * ((4 * wave) * 16 + n_o);
        const int x0s = h_o ^ ((n_o >> 1) & 7);
        const f32x4 *sW = lds + stage * kStage + 8 * kTight + ln_o;
        f32x4 d2[2][2][2], A1[2][NCO];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int j = 0; j < NCO; ++j) A1[hh][j] = sW[64 * (hh * NCO + j)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) d2[i][j][hh] = sH[8 * 16 * (2 * i + j) + (x0s ^ (4 * hh))];
#pragma unroll
        for (int g = 0; g < 8; ++g) {  // (pixel g >> 1, channel half g & 1)
            // the next step's pieces go out in the first five of the eight groups: the last ones need ~1.5k cycles to land
            if (pnext) {
                if (g < 4) { issue_next(2 * g, true, cnext, snext, cbf); issue_next(2 * g + 1, true, cnext, snext, cbf); }
                if (g == 4) issue_next(8, true, cnext, snext, cbf);
            } else {
                if (g < 3) { issue_next(2 * g, false, cnext, snext, cbf); issue_next(2 * g + 1, false, cnext, snext, cbf); }
                if (g == 3 && kTPW > 6) issue_next(6, false, cnext, snext, cbf);
            }
            __builtin_amdgcn_sched_barrier(0);
            // Output pixel (i, j) of a tile is, in the Winograd domain, position (3i, 3j) with sign (-1)^(i+j): A^T P = I for
            // P = [1 0; 0 0; 0 0; 0 -1], so adding s W1.x to M[3i][3j] adds W1.x to Y[i][j] and nothing to the other three outputs.
            // The P steps therefore accumulate straight into four of the Winograd accumulators (negated operand for the two
            // mixed corners) and need no registers of their own.
            const int px = g >> 1, hh = g & 1;
            const int pos = 12 * (px >> 1) + 3 * (px & 1);
            f32x4 bop = d2[px >> 1][px & 1][hh];
            if ((px >> 1) != (px & 1)) bop = -bop;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < NCO; ++j)
                    acc[pos][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[hh][j][k], bop[k], acc[pos][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- first tile: fetch K step 0 -----------------------------------------------------------------------------------
    decode(t_cur);
    set_fetch_tile();
#pragma unroll
    for (int k = 0; k < kTPW; ++k) issue_next(k, false, 0, 0, 0);
    WINO_TRACE(tr_i++);
    // the barrier publishes LDS bytes written by LDS-DMA: every wave's copies must have LANDED before it.  hipcc (ROCm 7.2) happens
    // to drain vmcnt before s_barrier, but gfx950's back-off barriers do not oblige it to: say so explicitly.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    WINO_TRACE(tr_i++);

    int it = 0;  // K steps done by this workgroup: step `it` is computed from stage it & 1
#pragma unroll 1
    for (;;) {
        const int t_next = t_cur + t_stride;
        const bool has_next = t_next < t_end;
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int j = 0; j < NCO; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // epilogue operands of this tile, requested under its last K step: output / residual offsets (out-of-image pixels
        // are out of the descriptors' range: loads read 0, stores are dropped), bias, residual tile
        const int n0 = 16 * cb0;
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)img * a.Ho * a.Wo * a.out_cs, 0, a.Ho * a.Wo * a.out_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.res ? a.res + (size_t)img * a.Ho * a.Wo * a.res_cs : a.out), 0,
                                                                              a.res ? a.Ho * a.Wo * a.res_cs * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
        f32x4 b4[NCO];
        const int ey0 = y0, ex0 = x0;  // (the DMA descriptors move on to the next tile under the last step)

        set_fetch_tile2();     // P-step descriptors of THIS tile (its Winograd descriptors were set one tile ago)
        const int cb_cur = cb0;
        const int nT = nS + nS2;
#pragma unroll 1
        for (int c = 0; c < nT; ++c) {
            const bool last = c + 1 == nT;
            if (last) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int oy = y0 + 2 * wave + i, ox = x0 + 2 * n + j;
                        const bool ok = (oy < a.Ho) & (ox < a.Wo);
                        const int voffR = ok ? ((oy * a.Wo + ox) * a.res_cs + n0 + 4 * h) * 4 : kOob;
#pragma unroll
                        for (int cb = 0; cb < NCO; ++cb) ost[i][j][cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, voffR + 64 * cb, 0, 0));
                    }
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb) b4[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (n0 + 16 * cb + 4 * h) * 4, 0, 0));
                if (has_next) {  // from here on the Winograd DMA descriptors belong to the next tile: its step 0 lands under this step
                    decode(t_next);
                    set_fetch_tile();
                }
            }
            // next step: Winograd step c+1 of this tile, P step c+1-nS of this tile, or step 0 of the next tile (the DMA of the
            // step after the workgroup's last one is issued too — it re-reads a step 0 into the idle stage and is never
            // used: no branch in the K loop)
            const bool pnext = SRC2 && !last && c + 1 >= nS;
            const int cnext = last ? 0 : pnext ? c + 1 - nS : c + 1, sn = (it + 1) & 1;
            if (!SRC2 || c < nS) {
                compute(it & 1, pnext, cnext, sn, cb_cur, pending);
                pending = false;
            } else {
                compute_p(it & 1, pnext, cnext, sn, cb_cur);
            }
            ++it;
            WINO_TRACE(tr_i < 61 ? tr_i++ : 61);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's LDS-DMA pieces (next stage) have landed before the barrier publishes them
            __syncthreads();
            WINO_TRACE(tr_i < 61 ? tr_i++ : 61);
        }

        // ---- epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; lane = 4 consecutive channels of the 2x2 pixels of tile n
        act_dispatch(a.act, a.slope, [&](auto fn) {
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb) {
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
                        f32x4 o = y[j] + b4[cb] + ost[i][j][cb];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = fn(o[r]);
#ifdef IDH_ABL_WINO_NOSTORE
                        asm volatile("" ::"v"(o));
#endif
                        ost[i][j][cb] = o;
                    }
                }
            }
        });
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = ey0 + 2 * wave + i, ox = ex0 + 2 * n + j;
                voffS[i][j] = ((oy < a.Ho) & (ox < a.Wo)) ? ((oy * a.Wo + ox) * a.out_cs + n0 + 4 * h) * 4 : kOob;
            }
        rsS = rsO;
        pending = true;
        WINO_TRACE(tr_i < 61 ? tr_i++ : 61);
        if (!has_next) break;
        t_cur = t_next;
    }
#ifndef IDH_ABL_WINO_NOSTORE
#pragma unroll
    for (int k = 0; k < 4 * NCO; ++k)  // the last tile's outputs
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ost[k >> 1 & 1][k & 1][k >> 2]), rsS, voffS[k >> 1 & 1][k & 1] + 64 * (k >> 2), 0, 0);
#endif
    WINO_TRACE(62);
}

template <int WAVES, int NCO, int CH, bool SRC2>
__global__ __launch_bounds__(64 * WAVES, 2) void conv3x3_wino_k(const WinoArgs wa) {
    __shared__ f32x4 lds[2 * wino_stage_slots<WAVES, NCO, CH, SRC2>()];
    wino_tiles<WAVES, NCO, CH, SRC2>(wa, lds, blockIdx.x, gridDim.x);
}

// Several mutually independent convolutions of one dependency level (the UNet++ grid at small batch: 3-6 convs of 48-400 tiles
// each) behind ONE persistent grid: a workgroup walks its share of op 0, then of op 1, ... without a device-wide barrier in
// between, so that the tail of one op is filled by the next one's tiles instead of idling (and a launch boundary) per op.
constexpr int kWinoMaxGroup = 6;
struct WinoGroupArgs {
    WinoArgs op[kWinoMaxGroup];
    int rot[kWinoMaxGroup];  // per-XCD rotation of the workgroup index: an op's tiles start where the previous op's ended
    int n;
};
template <int WAVES, int NCO, int CH, bool SRC2>
__global__ __launch_bounds__(64 * WAVES, 2) void conv3x3_wino_group_k(const WinoGroupArgs g) {
    __shared__ f32x4 lds[2 * wino_stage_slots<WAVES, NCO, CH, SRC2>()];
    const unsigned per = gridDim.x >> 3;  // the grid is a multiple of the 8 XCDs
    for (int i = 0; i < g.n; ++i) {
        const unsigned vblock = (((blockIdx.x >> 3) + per - (unsigned)g.rot[i] % per) % per) << 3 | (blockIdx.x & 7);
        wino_tiles<WAVES, NCO, CH, SRC2>(g.op[i], lds, vblock, gridDim.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the trailing, unused LDS-DMA of the op's last step must not land after the next op's)
        __syncthreads();  // the next op's first copies overwrite the LDS stages
    }
}

template <int WAVES, int NCO>
int wino_args(const ConvArgs &a, int N, WinoArgs &wa) {
    wa = WinoArgs{a, (a.Wo + kWinoTileW - 1) / kWinoTileW, (a.Ho + 2 * WAVES - 1) / (2 * WAVES), 0};
    wa.c.NT = a.Cout / (16 * NCO);
    const long long tiles = (long long)N * wa.tiles_x * wa.tiles_y * wa.c.NT;
    if (tiles >= (1ll << 31)) return IDH_EUNSUPPORTED;
    wa.tiles = (int)tiles;
    return IDH_OK;
}

// persistent grid: as many workgroups as the chip holds at once (LDS-limited)
template <int WAVES, int NCO, int CH, bool SRC2>
long long wino_resident() {
    static int resident = 0;
    if (resident == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = cus > 0 ? cus : 256;
    }
    constexpr int kLdsBytes = 2 * wino_stage_slots<WAVES, NCO, CH, SRC2>() * 16;
    return (long long)resident * (kLdsBytes * 2 <= 160 * 1024 ? 2 : 1);
}

template <int WAVES, int NCO, int CH, bool SRC2>
int launch_wino(const ConvArgs &a, int N, hipStream_t st) {
    WinoArgs wa;
    if (int rc = wino_args<WAVES, NCO>(a, N, wa)) return rc;
    long long grid = wino_resident<WAVES, NCO, CH, SRC2>();  // a multiple of the 8 XCDs
    if (grid > wa.tiles) grid = wa.tiles >= 8 ? wa.tiles / 8 * 8 : wa.tiles;
    hipLaunchKernelGGL((conv3x3_wino_k<WAVES, NCO, CH, SRC2>), dim3((unsigned)grid), dim3(64 * WAVES), 0, st, wa);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

template <int WAVES, int NCO, int CH, bool SRC2>
int launch_wino_group(const ConvArgs *const *as, const int *Ns, int n, hipStream_t st) {
    WinoGroupArgs g{};
    long long total = 0;
    int rot = 0;
    const long long res = wino_resident<WAVES, NCO, CH, SRC2>();
    for (int i = 0; i < n; ++i) {
        if (int rc = wino_args<WAVES, NCO>(*as[i], Ns[i], g.op[i])) return rc;
        g.rot[i] = rot;
        rot = (int)((rot + (g.op[i].tiles + 7) / 8) % (res / 8));  // tiles per XCD of this op
        total += g.op[i].tiles;
    }
    g.n = n;
    long long grid = res;
    if (grid > total) grid = total >= 8 ? total / 8 * 8 : 8;
    if ((grid & 7) || (res & 7)) return IDH_EUNSUPPORTED;  // the rotation assumes whole XCD octets (callers fall back to single launches)
    hipLaunchKernelGGL((conv3x3_wino_group_k<WAVES, NCO, CH, SRC2>), dim3((unsigned)grid), dim3(64 * WAVES), 0, st, g);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

}  // namespace

namespace idh_conv {

bool wino_supported(const ConvArgs &a) {
    const ConvSrc &s = a.s[0];
    const ConvSrc &s1 = a.s[1];
    const bool src2_ok = !s1.in || (s1.ks == 1 && s1.stride == 1 && !s1.up_in[0] && !s1.norm && s1.H == s.H && s1.W == s.W &&
                                    (long long)s1.H * s1.W * s1.cs * 4 < (1ll << 31) && (long long)s1.cblocks * a.Cout_pad * 64 < (1ll << 31));
    return s.ks == 3 && s.stride == 1 && s.pad_mode == IDH_PAD_ZEROS && !s.up_in[0] && !s.norm && src2_ok && a.S == 1 && (a.Cout % 32) == 0 &&
           (long long)s.H * s.W * s.cs * 4 < (1ll << 31) && (long long)a.Ho * a.Wo * a.out_cs * 4 < (1ll << 31) &&
           (!a.res || (long long)a.Ho * a.Wo * a.res_cs * 4 < (1ll << 31)) && (long long)s.cblocks * a.Cout_pad * 1024 < (1ll << 31);
}

int wino_max_group() { return kWinoMaxGroup; }

// n mutually independent Winograd convs (all with, or all without, a fused 1x1 second source) as one persistent grid
int launch_conv_wino_group(const ConvArgs *const *as, const int *Ns, int n, hipStream_t st) {
    if (n < 1 || n > kWinoMaxGroup) return IDH_EINVAL;
    const bool src2 = as[0]->s[1].in != nullptr;
    for (int i = 0; i < n; ++i)
        if (!wino_supported(*as[i]) || (as[i]->s[1].in != nullptr) != src2) return IDH_EUNSUPPORTED;
    return src2 ? launch_wino_group<4, 2, kWinoCH, true>(as, Ns, n, st) : launch_wino_group<4, 2, kWinoCH, false>(as, Ns, n, st);
}

int launch_conv_wino(const ConvArgs &a, int N, int rows, hipStream_t st) {
    if (!wino_supported(a)) return IDH_EUNSUPPORTED;
    (void)rows;
    // 8-row tiles, 8-channel K steps: 56 KiB of LDS -> 2 workgroups / CU
    return a.s[1].in ? launch_wino<4, 2, kWinoCH, true>(a, N, st) : launch_wino<4, 2, kWinoCH, false>(a, N, st);
}

}  // namespace idh_conv

extern "C" size_t idh_packed_wino_weight_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)((Cin + 15) & ~15) * ((Cout + 15) & ~15) * 16;
}

extern "C" int idh_pack_conv_weight_wino(const float *w, float *dst, int Cout, int Cin, void *stream) {
    constexpr int ch = kWinoCH;
    if (!w || !dst || Cout <= 0 || Cin <= 0) return IDH_EINVAL;
    const int nC = (Cin + 15) / 16, nCB = (Cout + 15) / 16;
    const long long total = (long long)nC * nCB * 4096;
    int grid = idh_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_wino_weight_k, dim3(grid), dim3(256), 0, idh_stream(stream), w, dst, Cout, Cin, nC * (16 / ch), nCB, ch / 4);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}
