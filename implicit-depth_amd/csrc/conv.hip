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
//     256 MFMA-cycles per wave: it streams from L1/L2 without LDS staging, and the waves of a
//     workgroup are independent — latency is hidden by wave-level parallelism (2-4 waves/SIMD);
//   * zero padding = predicated loads; torch.cat = channel-strided in/out pointers; the
//     residual projection of a BasicBlock is a second K-source of the same launch; bias +
//     residual + LeakyReLU are applied on the accumulators before the single store;
//   * layers with too few tiles to fill 256 CUs (12x16 / 24x32 maps) split K over `split_k`
//     waves that write raw partials; a reduce kernel applies the epilogue (deterministic, no atomics).
#include "idh_common.h"
#include "../../include/idh_ops.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvSrc {
    const float *in;
    const float *w;
    int cs, H, W, Cin;
    int ks, stride, pad_mode, cblocks;  // cblocks = Cin_pad / 16
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
    return (act == IDH_ACT_LRELU && v < 0.f) ? v * slope : v;
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void conv_mfma_k(const ConvArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ln = lane & 15, h = lane >> 4;

    const unsigned blk = idh_xcd_remap(blockIdx.x, gridDim.x);
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

    int step_base = 0;
#pragma unroll 1
    for (int si = 0; si < 2; ++si) {
        const ConvSrc s = a.s[si];
        if (s.in == nullptr) break;
        const int ntaps = s.ks * s.ks;
        const int pad = s.ks >> 1;
        const int src_steps = ntaps * s.cblocks;
        // intersect [t0,t1) with this source's steps
        const int lo_s = max(t0 - step_base, 0), hi_s = min(t1 - step_base, src_steps);
        step_base += src_steps;
        if (lo_s >= hi_s) continue;
        const int tap_lo = lo_s / s.cblocks, tap_hi = (hi_s - 1) / s.cblocks;
        const float *wl = s.w + ((size_t)h * a.Cout_pad + n_base + ln) * 4;
        const size_t w_cq_stride = (size_t)a.Cout_pad * 4;  // floats between consecutive ci/4 groups
#pragma unroll 1
        for (int tap = tap_lo; tap <= tap_hi; ++tap) {
            const int dy = tap / s.ks, dx = tap - dy * s.ks;
            const int c_lo = (tap == tap_lo) ? lo_s - tap * s.cblocks : 0;
            const int c_hi = (tap == tap_hi) ? hi_s - tap * s.cblocks : s.cblocks;
            const float *ap[TM];
            bool av[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int iy = oy[i] * s.stride + dy - pad;
                int ix = ox[i] * s.stride + dx - pad;
                bool ok = mv[i];
                if (s.pad_mode == IDH_PAD_REPLICATE) {
                    iy = min(max(iy, 0), s.H - 1);
                    ix = min(max(ix, 0), s.W - 1);
                } else {
                    ok = ok && iy >= 0 && iy < s.H && ix >= 0 && ix < s.W;
                    iy = min(max(iy, 0), s.H - 1);
                    ix = min(max(ix, 0), s.W - 1);
                }
                av[i] = ok;
                ap[i] = s.in + ((size_t)(nb[i] * s.H + iy) * s.W + ix) * s.cs + 4 * h;
            }
            const float *wt = wl + (size_t)tap * s.cblocks * 4 * w_cq_stride;
#pragma unroll 1
            for (int cb = c_lo; cb < c_hi; ++cb) {
                const bool cok = (16 * cb + 4 * h) < s.Cin;  // Cin % 4 == 0 (host-checked)
                f32x4 A[TM], Bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    A[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (av[i] && cok) A[i] = *reinterpret_cast<const f32x4 *>(ap[i] + 16 * cb);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    Bf[j] = *reinterpret_cast<const f32x4 *>(wt + (size_t)cb * 4 * w_cq_stride + 64 * j);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i][kk], Bf[j][kk], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg ----------
    if (a.S > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + 16 * i + 4 * h + r;
                if (m >= a.M) continue;
                float *o = a.ws + ((size_t)sp * a.M + m) * a.Cout_pad + n_base + ln;
#pragma unroll
                for (int j = 0; j < TN; ++j) o[16 * j] = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n_base + 16 * j + ln;
        if (co >= a.Cout) continue;
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + 16 * i + 4 * h + r;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bv;
                if (a.res) v += a.res[(size_t)m * a.res_cs + co];
                a.out[(size_t)m * a.out_cs + co] = act_apply(v, a.act, a.slope);
            }
    }
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
__global__ __launch_bounds__(256) void upsample2_k(const float *__restrict__ in, float *__restrict__ out, int N, int H,
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
        const int iy = y >> 1, ix = x >> 1;
        int y0, y1, x0, x1;
        float hy0, hy1, wx0, wx1;
        if (y & 1) { y0 = iy; y1 = min(iy + 1, H - 1); hy0 = 0.75f; hy1 = 0.25f; }
        else { y0 = max(iy - 1, 0); y1 = iy; hy0 = 0.25f; hy1 = 0.75f; }
        if (x & 1) { x0 = ix; x1 = min(ix + 1, W - 1); wx0 = 0.75f; wx1 = 0.25f; }
        else { x0 = max(ix - 1, 0); x1 = ix; wx0 = 0.25f; wx1 = 0.75f; }
        const float *b = in + (size_t)n * H * W * in_cs + 4 * q;
        const float4 p00 = *reinterpret_cast<const float4 *>(b + ((size_t)y0 * W + x0) * in_cs);
        const float4 p01 = *reinterpret_cast<const float4 *>(b + ((size_t)y0 * W + x1) * in_cs);
        const float4 p10 = *reinterpret_cast<const float4 *>(b + ((size_t)y1 * W + x0) * in_cs);
        const float4 p11 = *reinterpret_cast<const float4 *>(b + ((size_t)y1 * W + x1) * in_cs);
        float4 o;
        o.x = hy0 * (wx0 * p00.x + wx1 * p01.x) + hy1 * (wx0 * p10.x + wx1 * p11.x);
        o.y = hy0 * (wx0 * p00.y + wx1 * p01.y) + hy1 * (wx0 * p10.y + wx1 * p11.y);
        o.z = hy0 * (wx0 * p00.z + wx1 * p01.z) + hy1 * (wx0 * p10.z + wx1 * p11.z);
        o.w = hy0 * (wx0 * p00.w + wx1 * p01.w) + hy1 * (wx0 * p10.w + wx1 * p11.w);
        *reinterpret_cast<float4 *>(out + (((size_t)n * Ho + y) * Wo + x) * out_cs + 4 * q) = o;
    }
}

// (N,C,HW) dense -> NHWC slice with channel stride out_cs; LDS tile keeps both sides coalesced
__global__ __launch_bounds__(256) void import_nchw_k(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                     int HW, int out_cs) {
    __shared__ float tile[32][65];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 32;
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

// 1x1 conv to a single channel: one thread per pixel
__global__ __launch_bounds__(256) void pointwise_head_k(const float *__restrict__ in, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ out,
                                                        long long M, int C, int in_cs) {
    for (long long m = blockIdx.x * 256ll + threadIdx.x; m < M; m += gridDim.x * 256ll) {
        const float4 *p = reinterpret_cast<const float4 *>(in + m * in_cs);
        const float4 *wv = reinterpret_cast<const float4 *>(w);
        float s = 0.f;
        for (int q = 0; q < (C >> 2); ++q) {
            const float4 a = p[q], b = wv[q];
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
        out[m] = s + (bias ? bias[0] : 0.f);
    }
}

inline int ceil16(int v) { return (v + 15) & ~15; }

template <int TM, int TN>
void launch_conv(const ConvArgs &a, hipStream_t st) {
    const long long waves = (long long)a.MT * a.NT * a.S;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    hipLaunchKernelGGL((conv_mfma_k<TM, TN>), dim3(grid), dim3(256), 0, st, a);
}

int run_conv(const idh_op &op, hipStream_t st) {
    ConvArgs a{};
    int steps = 0;
    for (int i = 0; i < 2; ++i) {
        const idh_conv_src &s = op.src[i];
        ConvSrc &d = a.s[i];
        d.in = s.in;
        if (!s.in) continue;
        if (!s.w || s.Cin <= 0 || (s.Cin & 3) || s.cs < s.Cin || (s.cs & 3) || (s.ks != 1 && s.ks != 3) ||
            (s.stride != 1 && s.stride != 2) || s.H <= 0 || s.W <= 0)
            return IDH_EINVAL;
        const int pad = s.ks / 2;
        if ((s.H + 2 * pad - s.ks) / s.stride + 1 != op.Ho || (s.W + 2 * pad - s.ks) / s.stride + 1 != op.Wo)
            return IDH_EINVAL;
        d.w = s.w; d.cs = s.cs; d.H = s.H; d.W = s.W; d.Cin = s.Cin; d.ks = s.ks; d.stride = s.stride;
        d.pad_mode = s.pad_mode; d.cblocks = ceil16(s.Cin) / 16;
        steps += s.ks * s.ks * d.cblocks;
    }
    if (!a.s[0].in || !op.out || op.Cout <= 0 || op.N <= 0) return IDH_EINVAL;
    a.bias = op.bias; a.res = op.res; a.out = op.out; a.ws = op.ws;
    a.res_cs = op.res_cs; a.out_cs = op.out_cs; a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout;
    a.Cout_pad = ceil16(op.Cout);
    const long long M = (long long)op.N * op.Ho * op.Wo;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    a.M = (int)M; a.steps_total = steps; a.act = op.act; a.slope = op.slope;
    a.S = op.split_k > 1 ? op.split_k : 1;
    if (a.S > steps) a.S = steps;
    if (a.S > 1 && !op.ws) return IDH_EWORKSPACE;
    int tm = op.tile_m, tn = op.tile_n;
    const int nsub = a.Cout_pad / 16;
    if (tn == 0) tn = (nsub % 4 == 0) ? 4 : (nsub % 2 == 0 ? 2 : 1);
    if (tm == 0) tm = 4;
    if ((tm != 1 && tm != 2 && tm != 4) || (tn != 1 && tn != 2 && tn != 4) || nsub % tn) return IDH_EINVAL;
    a.MT = (int)((M + 16 * tm - 1) / (16 * tm));
    a.NT = nsub / tn;
#define IDH_CASE(TM_, TN_) \
    if (tm == TM_ && tn == TN_) launch_conv<TM_, TN_>(a, st);
    IDH_CASE(4, 4) IDH_CASE(2, 4) IDH_CASE(1, 4) IDH_CASE(4, 2) IDH_CASE(2, 2) IDH_CASE(1, 2) IDH_CASE(4, 1)
    IDH_CASE(2, 1) IDH_CASE(1, 1)
#undef IDH_CASE
    IDH_CHECK_LAUNCH();
    if (a.S > 1) {
        const long long tot = M * op.Cout;
        int grid = idh_cdiv(tot, 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(splitk_reduce_k, dim3(grid), dim3(256), 0, st, op.ws, op.bias, op.res, op.out, a.M, op.Cout,
                           a.Cout_pad, a.S, op.res_cs, op.out_cs, op.act, op.slope);
        IDH_CHECK_LAUNCH();
    }
    return IDH_OK;
}

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
            case IDH_OP_CONV: {
                const int rc = run_conv(op, st);
                if (rc != IDH_OK) return rc;
                break;
            }
            case IDH_OP_UPSAMPLE2: {
                if (!s.in || !op.out || (s.Cin & 3) || (s.cs & 3) || (op.out_cs & 3) || op.N <= 0) return IDH_EINVAL;
                const long long tot = (long long)op.N * 4 * s.H * s.W * (s.Cin >> 2);
                int grid = idh_cdiv(tot, 256);
                if (grid > 8192) grid = 8192;
                hipLaunchKernelGGL(upsample2_k, dim3(grid), dim3(256), 0, st, s.in, op.out, op.N, s.H, s.W, s.Cin, s.cs,
                                   op.out_cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_NCHW_TO_NHWC: {
                if (!s.in || !op.out || op.N <= 0 || op.N > 65535) return IDH_EINVAL;
                const int HW = s.H * s.W;
                hipLaunchKernelGGL(import_nchw_k, dim3(idh_cdiv(HW, 64), idh_cdiv(s.Cin, 32), op.N), dim3(256), 0, st, s.in,
                                   op.out, s.Cin, HW, op.out_cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_NHWC_TO_NCHW: {
                if (!s.in || !op.out || op.N <= 0 || op.N > 65535) return IDH_EINVAL;
                const int HW = s.H * s.W;
                hipLaunchKernelGGL(export_nchw_k, dim3(idh_cdiv(HW, 64), idh_cdiv(s.Cin, 32), op.N), dim3(256), 0, st, s.in,
                                   op.out, s.Cin, HW, s.cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            case IDH_OP_POINTWISE_HEAD: {
                if (!s.in || !s.w || !op.out || (s.Cin & 3) || (s.cs & 3)) return IDH_EINVAL;
                const long long M = (long long)op.N * s.H * s.W;
                int grid = idh_cdiv(M, 256);
                if (grid > 8192) grid = 8192;
                hipLaunchKernelGGL(pointwise_head_k, dim3(grid), dim3(256), 0, st, s.in, s.w, op.bias, op.out, M, s.Cin,
                                   s.cs);
                IDH_CHECK_LAUNCH();
                break;
            }
            default:
                return IDH_EINVAL;
        }
    }
    return IDH_OK;
}
