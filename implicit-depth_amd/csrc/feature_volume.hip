// Fused MLP feature volume (gfx950): plane-sweep warp + per-voxel metadata + 3-layer MLP.
//
// Replaces FeatureVolumeManager.build_cost_volume + forward (reference
// modules/cost_volume.py:437-706, 324-358; MLP modules/networks.py:218-233; geometry
// utils/geometry_utils.py:55-89, 149-195).  The reference materialises, for each of D planes, a
// (B,H,W,16(K+1)+10K+4) tensor (202 channels for K=7: 635 MB over 64 planes per frame) and runs
// three nn.Linear over it.  Here the 202-vector of a voxel only ever exists in registers:
//
//   * 4 lanes cooperate on one voxel (pixel x plane): lane quarter q owns channels 4q..4q+3 of
//     every feature vector, so one bilinear tap is a 64-byte segment read by 4 adjacent lanes,
//     and — by construction — what a lane holds is exactly its B-operand fragment of
//     v_mfma_f32_16x16x4_f32 (lane (col, q) supplies k = 16c + 4q + kk).  K order is free as long
//     as the weights are packed to match, so W1's columns are re-ordered on the host into
//         [ K blocks: warped features of view k ][ 4 blocks: per-voxel metadata ]
//     where quarter q carries the metadata (mask, z, dot, ray angle, ray xyz) of views q and q+4
//     (+ the plane depth), i.e. each lane builds rays only for "its" two views.
//   * everything that does not depend on the plane is folded out of the per-voxel GEMM:
//       - cur features and cur ray -> a per-pixel pre-activation computed once per task,
//       - the 3K pose-distance inputs -> a per-batch-element bias (setup kernel).
//   * layers are computed transposed (out^T = W . act^T, see csrc/mlp.hip) so activations stay in
//     registers from the gather to the final dot product; W1 (per-voxel part) and W2 live in LDS in
//     MFMA fragment order (152 KiB for K = 7, all 160 KiB for K = 8) and are shared by the 8 waves of a persistent
//     workgroup; per plane a wave issues (K+4+8)*32 MFMAs.
// Roofline: fp32 MFMA (2*D*N*(16K+64+128)*128 + ... ~ 66.6 GFLOP per 96x128x64 frame).
#include <stdlib.h>

#include "idh_common.h"
#include "split_f16.h"

namespace {

using namespace idh_f16;

constexpr int kC = 16;
constexpr int kHid = 128;
constexpr int kNS = 8;            // 128 / 16
constexpr int kMaxK = 8;          // LDS budget: (K+4)*8 KiB + 64 KiB = 160 KiB exactly at K = 8 (b2 / w3 stay in L1)

// per-b workspace layout (floats), sized for IDH_MAX_SOURCE_VIEWS = 16 views: [0,192) hom[k][12]   [192,256) tsrc[k][4]
//                                  [256,265) invK 3x3   [272, 272+128) bias1_b
constexpr int kWsHom = 0, kWsT = 192, kWsInvK = 256, kWsBias = 272;
static_assert(kWsT == 12 * IDH_MAX_SOURCE_VIEWS && kWsInvK == kWsT + 4 * IDH_MAX_SOURCE_VIEWS && kWsBias + kHid <= 400, "workspace layout");
constexpr int kWsStrideReal = 400;

// LeakyReLU(0.01) as max(x, 0.01 x) in TWO vector instructions: v_mul + gfx950's v_maximum3_f32 (IEEE-754-2019 maximum: NaN propagates, -0 < +0), which
// the compiler emits for __builtin_elementwise_maximum without the canonicalising v_max that fmaxf() puts in front of an MFMA result (three
// instructions, like multiply / compare / select).  Same values as the select form for every input, NaN and -0 included.  (A bare v_max_f32
// through inline asm is NOT an option: hipcc does not track the MFMA -> VALU wait states of inline-asm operands, and fv_mlp_k<8> produced
// wrong values with it; profiles/r05/experiments.md.)
__device__ __forceinline__ float lrelu01(float x) {
#ifdef IDH_ABL_FV2_NOACT
    return x;
#endif
    return __builtin_elementwise_maximum(x, x * 0.01f);
}
// (the split-precision kernel keeps the select form: there the two-instruction forms - this one or fmaxf on the canonical fma result - measure
// 8.60 ms against 8.38 at 32 frames, with 16 B more scratch; profiles/r05/experiments.md)
__device__ __forceinline__ float lrelu01_sel(float x) { return x >= 0.f ? x : x * 0.01f; }

__device__ __forceinline__ float fv_depth_plane(int i, int D, float dmin, float dmax) {
    float ramp = 0.f;
    if (D > 1) {
        const float step = 1.0f / (float)(D - 1);
        ramp = (i < D / 2) ? step * (float)i : 1.0f - step * (float)(D - 1 - i);
    }
    return expf(logf(dmin) + logf(dmax / dmin) * ramp);
}

// ---- setup: homographies, source camera centres, pose-distance bias (one block per b) --------
// w1_pose: (128, 3K) row-major = W1[:, pose-distance | R-measure | t-measure columns]
__global__ void fv_setup_k(const float *__restrict__ src_K, const float *__restrict__ src_E,
                           const float *__restrict__ src_poses, const float *__restrict__ cur_invK,
                           const float *__restrict__ w1_pose, const float *__restrict__ b1, int K,
                           float *__restrict__ ws) {
    const int b = blockIdx.x, t = threadIdx.x;
    float *o = ws + (size_t)b * kWsStrideReal;
    __shared__ float s_pd[3 * IDH_MAX_SOURCE_VIEWS];
    if (t < K) {
        const float *Km = src_K + (size_t)(b * K + t) * 16, *Em = src_E + (size_t)(b * K + t) * 16;
        const float *iK = cur_invK + (size_t)b * 16, *Pm = src_poses + (size_t)(b * K + t) * 16;
        float P[3][4];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
                for (int m = 0; m < 4; ++m) s = fmaf(Km[i * 4 + m], Em[m * 4 + j], s);
                P[i][j] = s;
            }
        float *h = o + kWsHom + 12 * t;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                float s = 0.f;
                for (int m = 0; m < 3; ++m) s = fmaf(P[i][m], iK[m * 4 + j], s);
                h[i * 3 + j] = s;
            }
            h[9 + i] = P[i][3];
        }
        // source camera centre in the current frame (cost_volume.py:630-633)
        o[kWsT + 4 * t + 0] = Pm[3]; o[kWsT + 4 * t + 1] = Pm[7]; o[kWsT + 4 * t + 2] = Pm[11]; o[kWsT + 4 * t + 3] = 0.f;
        // DVMVS pose distance (geometry_utils.py:183-195)
        const float tr = Pm[0] + Pm[5] + Pm[10];
        const float rm = sqrtf(2.f * (1.f - fminf(3.f, tr) / 3.f));
        const float tm = sqrtf(Pm[3] * Pm[3] + Pm[7] * Pm[7] + Pm[11] * Pm[11]);
        s_pd[t] = sqrtf(tm * tm + rm * rm);
        s_pd[K + t] = rm;
        s_pd[2 * K + t] = tm;
    }
    if (t < 9) o[kWsInvK + t] = cur_invK[(size_t)b * 16 + (t / 3) * 4 + (t % 3)];
    __syncthreads();
    if (t < kHid) {
        float s = b1[t];
        for (int j = 0; j < 3 * K; ++j) s = fmaf(w1_pose[(size_t)t * 3 * K + j], s_pd[j], s);
        o[kWsBias + t] = s;
    }
}

struct FvArgs {
    const float *cur;      // B,N,16
    const float *src;      // B,K,N,16
    const float *ws;       // per-b constants
    const float *w1v;      // packed per-voxel part of W1: (K+4) blocks x 8 x 64 x float4
    const float *w1p;      // packed per-pixel part of W1:   2 blocks x 8 x 64 x float4
    const float *w2;       // packed W2: 8 x 8 x 64 x float4
    const float *vecs;     // b2[128], w3[128], b3
    float *vol;            // (B,D,N) if vol_cs == 0 else NHWC (B,N,vol_cs)
    unsigned char *mask;   // (B,N) or null
    int vol_cs;
    int B, K, H, W, D;
    int tiles_per_img;     // ceil(N/16)
    int DP, G;             // planes per task, plane groups
    int J, MB;             // generic kernel: source views per lane quarter (view q + 4j), metadata blocks = 2 J
    float dmin, dmax;
    // idh_volume_opts, resolved: batch strides (floats) and caller-supplied planes (null = log-spaced)
    long long cur_bs, src_bs;
    const float *planes;
    long long planes_sb, planes_sd;
    int planes_sp;
};

__device__ __forceinline__ float fv_plane(const FvArgs &a, int b, int p, int d) {
    return a.planes ? a.planes[(size_t)b * a.planes_sb + (size_t)d * a.planes_sd + (size_t)p * a.planes_sp]
                    : fv_depth_plane(d, a.D, a.dmin, a.dmax);
}

// KT = compile-time view count (7, 8) or 0 = run-time: with KT > 0 the view loop is fully unrolled so the
// scheduler can slot view k+1's projection / blend VALU work between view k's 32 MFMAs.
template <int KT>
__global__ __launch_bounds__(512) void fv_mlp_k(const FvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *sW1 = reinterpret_cast<f32x4 *>(smem_raw);            // (K+4)*8*64
    f32x4 *sW2 = sW1 + (a.K + 4) * kNS * 64;                       // 8*8*64
    {
        const f32x4 *g1 = reinterpret_cast<const f32x4 *>(a.w1v);
        const f32x4 *g2 = reinterpret_cast<const f32x4 *>(a.w2);
        const int n1 = (a.K + 4) * kNS * 64, n2 = kNS * kNS * 64;
        for (int i = threadIdx.x; i < n1; i += 512) sW1[i] = g1[i];
        for (int i = threadIdx.x; i < n2; i += 512) sW2[i] = g2[i];
    }
    __syncthreads();
    // b2 / w3 / b3 (1 KiB) are read through L1: at K = 8 the weights use the whole 160 KiB of LDS
    const float *s_b2 = a.vecs, *s_w3 = a.vecs + kHid;
    const float b3 = a.vecs[2 * kHid];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, q = lane >> 4;
    const int N = a.H * a.W;
    const int K = KT > 0 ? KT : a.K;
    const float Wf = (float)a.W, Hf = (float)a.H;
    const long long ntasks = (long long)a.B * a.tiles_per_img * a.G;

    // XCD-aware: hardware puts workgroup b on XCD b % 8; after the remap each XCD walks a contiguous run of tasks
    // (= a band of image rows) per round, so the taps of its 32 workgroups share that XCD's 4 MB L2
    for (long long task = (long long)idh_xcd_remap(blockIdx.x, gridDim.x) * 8 + wave; task < ntasks; task += (long long)gridDim.x * 8) {
        // wave-uniform task coordinates: pin to SGPRs so the per-b constants come through the scalar cache
        const int g = __builtin_amdgcn_readfirstlane((int)(task % a.G));
        const int tile = __builtin_amdgcn_readfirstlane((int)((task / a.G) % a.tiles_per_img));
        const int b = __builtin_amdgcn_readfirstlane((int)(task / ((long long)a.G * a.tiles_per_img)));
        const int p_raw = tile * 16 + ln;
        const bool live = p_raw < N;
        const int p = live ? p_raw : N - 1;
        const int py = p / a.W, px = p - py * a.W;
        const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
        const float *pb = a.ws + (size_t)b * kWsStrideReal;

        const f32x4 cur4 = *reinterpret_cast<const f32x4 *>(a.cur + (size_t)b * a.cur_bs + (size_t)p * kC + 4 * q);
        // back-projected ray of the pixel (geometry_utils.py:60) and its direction (cost_volume.py:618)
        const float *iK = pb + kWsInvK;
        const float rx = fmaf(iK[0], pxf, fmaf(iK[1], pyf, iK[2]));
        const float ry = fmaf(iK[3], pxf, fmaf(iK[4], pyf, iK[5]));
        const float rz = fmaf(iK[6], pxf, fmaf(iK[7], pyf, iK[8]));

        // ---- per-pixel pre-activation: bias_b + W1[:,cur].cur + W1[:,cur_ray].ray -------------
        f32x4 pre[kNS];
#pragma unroll
        for (int i = 0; i < kNS; ++i) pre[i] = *reinterpret_cast<const f32x4 *>(pb + kWsBias + 16 * i + 4 * q);
        const int d0 = g * a.DP, d1 = min(a.D, d0 + a.DP);
        if (d0 >= d1) continue;
        // cur ray = normalize(depth * r): plane independent for depth > 0 (F.normalize eps 1e-12)
        const float rn = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
        const float crx = rx / rn, cry = ry / rn, crz = rz / rn;
        {
            const f32x4 rayB = (q == 0) ? (f32x4){crx, cry, crz, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 *w1p = reinterpret_cast<const f32x4 *>(a.w1p);
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
                const f32x4 A0 = w1p[(0 * kNS + i) * 64 + lane];
                const f32x4 A1 = w1p[(1 * kNS + i) * 64 + lane];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pre[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[kk], cur4[kk], pre[i], 0, 0, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pre[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[kk], rayB[kk], pre[i], 0, 0, 0);
            }
        }

        // ---- the projection of a source view is computed ONCE per voxel, by the lane quarter that owns the view's metadata (quarter q: views q and
        // q + 4), and handed to the other three quarters through ds_bpermute_b32: packed tap address + 4 bilinear weights = 5 values per view
        // instead of ~65 vector instructions repeated by all four lanes of the voxel (the fp32 MFMAs and the vector ALU share the SIMD: every
        // instruction removed is matrix time).  Plane-independent parts of the own two views' homographies stay in registers.
        const int ov[2] = {min(q, K - 1), min(q + 4, K - 1)};  // (absent views: a valid stand-in whose results nobody reads)
        float oq[2][3], oh[2][3];
#pragma unroll
        for (int jv = 0; jv < 2; ++jv) {
            const float *hm = pb + kWsHom + 12 * ov[jv];
            oq[jv][0] = fmaf(hm[0], pxf, fmaf(hm[1], pyf, hm[2]));
            oq[jv][1] = fmaf(hm[3], pxf, fmaf(hm[4], pyf, hm[5]));
            oq[jv][2] = fmaf(hm[6], pxf, fmaf(hm[7], pyf, hm[8]));
            oh[jv][0] = hm[9]; oh[jv][1] = hm[10]; oh[jv][2] = hm[11];
        }
        // taps through a buffer descriptor of this frame's K source maps: 32-bit offsets (view k: scalar offset k * N * 64 bytes)
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.src + (size_t)b * a.src_bs), 0, K * N * kC * 4, 0x00020000);
        const int bp0 = 4 * ln;  // ds_bpermute address of this voxel's lane in quarter 0 (+ 64 per quarter)

        f32x4 ob = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool vec_ok = ((d0 & 3) == 0) && ((a.vol_cs & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.vol) & 15) == 0);
        // ---- per-plane prologue of a voxel: the own two views' projection (bilinear weights, packed tap address: pixel index of tap 00 | x step << 30 |
        // y step << 31) and their viewing rays / ray angles (cost_volume.py:630-659).  It runs ONE PLANE AHEAD, inside the previous plane's layer 2
        // (256 MFMAs to hide under), together with the first view's tap loads: computed at the top of its own plane, the dependent chain
        // projection -> ds_bpermute -> address -> tap loads (L2 latency) -> blend stood in front of every plane's first MFMA.
        struct Own { int pk; float w00, w01, w10, w11, z; };
        struct Pro { Own own[2]; float depth, m3, m4, m5, m6, m10, m11, m12, m13; };
        struct Tap { f32x4 t00, t01, t10, t11; };
        struct Wts { float w00, w01, w10, w11; };
        float oc[2][3];  // source camera centres of the own views (cost_volume.py:630-633)
#pragma unroll
        for (int jv = 0; jv < 2; ++jv) {
            const float *t = pb + kWsT + 4 * ov[jv];
            oc[jv][0] = t[0]; oc[jv][1] = t[1]; oc[jv][2] = t[2];
        }
        auto prologue = [&](int d) {
            Pro P;
            const float depth = fv_plane(a, b, p, d);
            P.depth = depth;
#pragma unroll
            for (int jv = 0; jv < 2; ++jv) {
#ifdef IDH_ABL_FV2_NOPRO
                P.own[jv].pk = p; P.own[jv].w00 = P.own[jv].w01 = P.own[jv].w10 = P.own[jv].w11 = depth; P.own[jv].z = depth;
                continue;
#endif
                const float cx = fmaf(depth, oq[jv][0], oh[jv][0]);
                const float cy = fmaf(depth, oq[jv][1], oh[jv][1]);
                const float cz = fmaf(depth, oq[jv][2], oh[jv][2]);
                const float z = fmaxf(cz, 1e-5f);
                float r = __builtin_amdgcn_rcpf(z);
                r = r * fmaf(-z, r, 2.0f);
                const float su = cx * r, sv = cy * r;
                const float sx = fminf(fmaxf(su - 0.5f, -1.0f), Wf);
                const float sy = fminf(fmaxf(sv - 0.5f, -1.0f), Hf);
                const float x0f = floorf(sx), y0f = floorf(sy);
                const float fx = sx - x0f, fy = sy - y0f;
                const int x0 = (int)x0f, y0 = (int)y0f;
                const float wx0 = (x0 >= 0 && x0 < a.W) ? 1.0f - fx : 0.f;
                const float wx1 = (x0 + 1 < a.W) ? fx : 0.f;
                const float wy0 = (y0 >= 0 && y0 < a.H) ? 1.0f - fy : 0.f;
                const float wy1 = (y0 + 1 < a.H) ? fy : 0.f;
                const int xa0 = min(max(x0, 0), a.W - 1), xa1 = min(x0 + 1, a.W - 1);
                const int ya0 = min(max(y0, 0), a.H - 1), ya1 = min(y0 + 1, a.H - 1);
                P.own[jv].pk = (ya0 * a.W + xa0) | ((xa1 - xa0) << 30) | ((ya1 - ya0) << 31);
                P.own[jv].w00 = wx0 * wy0; P.own[jv].w01 = wx1 * wy0; P.own[jv].w10 = wx0 * wy1; P.own[jv].w11 = wx1 * wy1;
                P.own[jv].z = z;
            }
            // unit ray e = (X - c) / |X - c| through v_rsq_f32 + one Newton step (<= 1 ulp), and the ray angle as the plain dot product cr . e: the
            // reference divides it by max(|cr|, 1e-5) max(|e|, 1e-5) (cosine_similarity), both 1 up to rounding for unit vectors - a 1e-7 relative
            // difference, three orders below the 1e-4 bar.  (Absent views: computed on the stand-in view, multiplied by zero weight columns.)
            const float Xx = depth * rx, Xy = depth * ry, Xz = depth * rz;
            auto ray = [&](int jv, float &ang, float &e0, float &e1, float &e2) {
                const float ax = Xx - oc[jv][0], ay = Xy - oc[jv][1], az = Xz - oc[jv][2];
                const float n2 = fmaf(az, az, fmaf(ay, ay, ax * ax));
                float in = __builtin_amdgcn_rsqf(fmaxf(n2, 1e-24f));
                in = in * fmaf(-0.5f * n2 * in, in, 1.5f);
                e0 = ax * in; e1 = ay * in; e2 = az * in;
                ang = fmaf(crz, e2, fmaf(cry, e1, crx * e0));
            };
#ifdef IDH_ABL_FV2_NORAY
            P.m3 = P.m4 = P.m5 = P.m6 = P.m10 = P.m11 = P.m12 = P.m13 = depth;
            return P;
#endif
            ray(0, P.m3, P.m4, P.m5, P.m6);
            ray(1, P.m10, P.m11, P.m12, P.m13);
            return P;
        };
        auto issue = [&](const Pro &P, int k) {  // k: compile-time after unrolling (selects own[k >> 2] and the source quarter k & 3)
            const int pk = __builtin_amdgcn_ds_bpermute(bp0 + 64 * (k & 3), P.own[k >> 2].pk);
            const int o00 = ((pk & 0x3fffffff) << 6) + 16 * q;
            const int o01 = o00 + (((pk >> 30) & 1) << 6);
            const int ystep = (int)((unsigned)pk >> 31) * (a.W * 64);
            Tap t;
#ifdef IDH_ABL_FV2_NOTAPS  // (timing experiments, tools/abl_fv32.sh: results are meaningless)
            t.t00 = t.t01 = t.t10 = t.t11 = (f32x4){1.f + (float)o00, 2.f + (float)o01, 3.f + (float)ystep, 4.f};
            return t;
#endif
            const int so = __builtin_amdgcn_readfirstlane(k * N * (kC * 4));
            t.t00 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o00, so, 0));
            t.t01 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o01, so, 0));
            t.t10 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o00 + ystep, so, 0));
            t.t11 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o01 + ystep, so, 0));
            return t;
        };
        auto weights = [&](const Pro &P, int k) {
            const Own &o = P.own[k >> 2];
            auto bperm = [&](float v) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp0 + 64 * (k & 3), __builtin_bit_cast(int, v))); };
            Wts w;
            w.w00 = bperm(o.w00); w.w01 = bperm(o.w01); w.w10 = bperm(o.w10); w.w11 = bperm(o.w11);
            return w;
        };
        Pro P = prologue(d0);
        Tap cur = issue(P, 0);
        Wts wc = weights(P, 0);
#pragma unroll 1
        for (int d = d0; d < d1; ++d) {
            const float depth = P.depth;
            f32x4 acc1[kNS];
#pragma unroll
            for (int i = 0; i < kNS; ++i) acc1[i] = pre[i];
            // metadata registers of this lane: six per view - z, dot, ray angle, ray xyz - [1..6] view q, [8..13] view q+4, and the plane depth.  The
            // per-view "valid" input of the reference is identically 1 (z is clamped to 1e-5 BEFORE the z > 0 test, geometry_utils.py:86,
            // cost_volume.py:216; NaN depths included): its weight columns are folded into the layer-1 bias on the host
            // (implicit-depth_amd/cost_volume.py), which leaves 12 slots per lane quarter = three 16-column blocks, plus the plane depth: in the
            // (unused) first slot of view 7 in quarter 3 when K < 8, else as the only column of a fourth block.
            float m2 = 0.f, m9 = 0.f;
            const float m1 = q < K ? P.own[0].z : 0.f;
            float m8 = q + 4 < K ? P.own[1].z : 0.f;
            // Software-pipelined view loop: the tap address of view k+1 is fetched from its owner quarter and its 4 tap loads are issued before
            // the bilinear blend / 32 MFMAs of view k, so L2 latency hides under matrix work; the weights follow with the address.
            // Wave priority: layer 1 (taps, blend, metadata, activation: the vector-heavy half of a plane) runs at priority 2, layers 2 / 3 at 0.
            // The SIMD's two waves drift into opposite halves and the arbiter then gives the vector-heavy wave its issue slots while the other
            // wave's back-to-back MFMAs fill the matrix pipe: 15.2 -> 14.6 ms at 32 frames; raising layer 3 as well, or only the view loop, or the
            // next plane's prologue: 14.8 - 15.1 (profiles/r05/experiments.md).
            __builtin_amdgcn_s_setprio(2);
            constexpr int KU = KT > 0 ? KT : kMaxK;
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                if (KT == 0 && k >= K) break;
                const int kn = (KT > 0) ? (k + 1 < KT ? k + 1 : k) : k + 1;  // (KT = 0: the stand-in of an absent view k + 1 < 8 is fetched, never used)
                Tap nxt = cur;
                Wts wn = wc;
                if (kn != k && kn < KU) { nxt = issue(P, kn); wn = weights(P, kn); }
                f32x4 wv;
#ifdef IDH_ABL_FV2_NOBLEND
                wv = cur.t00 + cur.t11;
                float part = wc.w00 + wc.w11 + cur.t01[0] + cur.t10[0];
#else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    wv[e] = fmaf(wc.w11, cur.t11[e], fmaf(wc.w10, cur.t10[e], fmaf(wc.w01, cur.t01[e], wc.w00 * cur.t00[e])));
                float part = wv[0] * cur4[0];
                part = fmaf(wv[1], cur4[1], part); part = fmaf(wv[2], cur4[2], part); part = fmaf(wv[3], cur4[3], part);
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
#endif
                const float dotv = part;  // * mask (== 1)
                const bool s0 = (k == q), s1 = (k == q + 4);
                m2 = s0 ? dotv : m2;
                m9 = s1 ? dotv : m9;
                // layer-1 block k: warped features of view k
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
#ifdef IDH_ABL_FV2_NOLDSA
                    const f32x4 A = (f32x4){1.f, 2.f, 3.f, 4.f} * (float)(i + 1);
#else
                    const f32x4 A = sW1[(k * kNS + i) * 64 + lane];
#endif
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], wv[kk], acc1[i], 0, 0, 0);
                }
                if constexpr (KT > 0) {
                    // Scheduling hint (unrolled view loop = one scheduling region): small alternating groups of MFMAs and vector instructions
                    // instead of a vector block followed by 32 back-to-back matrix ops (history of the pattern: profiles/r02/experiments.md)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#ifndef IDH_ABL_FV_VALU_PER_GROUP
#define IDH_ABL_FV_VALU_PER_GROUP 3
#endif
                        __builtin_amdgcn_sched_group_barrier(0x002, IDH_ABL_FV_VALU_PER_GROUP, 0);
                    }
                }
                cur = nxt;
                wc = wn;
            }
            const float m3 = P.m3, m4 = P.m4, m5 = P.m5, m6 = P.m6, m10 = P.m10, m11 = P.m11, m12 = P.m12, m13 = P.m13;
            const bool depth_in_q3 = K < 8;
            if (depth_in_q3 && q == 3) m8 = depth;
            const f32x4 mb[4] = {(f32x4){m1, m2, m3, m4}, (f32x4){m5, m6, m8, m9}, (f32x4){m10, m11, m12, m13}, (f32x4){depth, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const f32x4 A = sW1[((K + c) * kNS + i) * 64 + lane];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], mb[c][kk], acc1[i], 0, 0, 0);
                }
            }
            if (!depth_in_q3) {  // K = 8: the plane depth is the only column of block 3 (quarter 0, k-step 0)
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const float A0 = reinterpret_cast<const float *>(&sW1[((K + 3) * kNS + i) * 64 + lane])[0];
                    acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, mb[3][0], acc1[i], 0, 0, 0);
                }
            }
            // ---- LeakyReLU(0.01) -> layer 2 (weights from LDS) -> LeakyReLU -> layer 3 ----------
            f32x4 acc2[kNS];
            Pro Pn;
            Tap curn;
            Wts wcn;
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[i][r] = lrelu01(acc1[i][r]);
                acc2[i] = *reinterpret_cast<const f32x4 *>(s_b2 + 16 * i + 4 * q);
            }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int c = 0; c < kNS; ++c) {
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
#ifdef IDH_ABL_FV2_NOLDSA
                    const f32x4 A = (f32x4){1.f, 2.f, 3.f, 4.f} * (float)(i + 1);
#else
                    const f32x4 A = sW2[(c * kNS + i) * 64 + lane];
#endif
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], acc1[c][kk], acc2[i], 0, 0, 0);
                }
                // The NEXT plane's prologue and first taps ride on layer 2 (the last plane of a task repeats its own: harmless).  Measured placements,
                // 32 frames (profiles/r05/experiments.md): prologue at the top of its own plane 15.45 ms; here with a sched_barrier per chunk 15.48 (no
                // gain: the scheduler then runs the chunk's 32 MFMAs first and the whole dependent chain after them); without the per-chunk barriers,
                // in chunk 0 / 1 / 4: 14.9 / 15.1 / 15.3; before the activation 15.5; before the metadata MFMAs 15.4.
                // (Run-time view count, KT = 0: chunk 1 and a barrier per chunk - without them that variant spills 400 B per lane.)
                if (c == (KT > 0 ? 0 : 1)) {
                    Pn = prologue(min(d + 1, d1 - 1));
                    curn = issue(Pn, 0);
                    wcn = weights(Pn, 0);
                }
                if constexpr (KT == 0) __builtin_amdgcn_sched_barrier(0);
            }
            float s = 0.f;
#ifdef IDH_ABL_FV2_NOL3
#pragma unroll
            for (int i = 0; i < kNS; ++i) s += acc2[i][0];
#else
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
                const f32x4 w3 = *reinterpret_cast<const f32x4 *>(s_w3 + 16 * i + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fmaf(w3[r], lrelu01(acc2[i][r]), s);
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
#endif
            const float val = s + b3;
            if (a.vol_cs > 0) {
                // NHWC output: collect 4 consecutive planes and write one 16-byte vector (a 4-byte store
                // per plane into 256-byte pixel rows cost ~10x write amplification in the PMC counters)
                const int e = (d - d0) & 3;
                ob[0] = e == 0 ? val : ob[0]; ob[1] = e == 1 ? val : ob[1]; ob[2] = e == 2 ? val : ob[2]; ob[3] = e == 3 ? val : ob[3];
                if (q == 0 && live && (e == 3 || d == d1 - 1)) {
                    float *o = a.vol + ((size_t)b * N + p) * a.vol_cs + (d - e);
                    if (e == 3 && vec_ok) *reinterpret_cast<f32x4 *>(o) = ob;
                    else { o[0] = ob[0]; if (e >= 1) o[1] = ob[1]; if (e >= 2) o[2] = ob[2]; if (e >= 3) o[3] = ob[3]; }
                }
            } else if (q == 0 && live) {
                a.vol[((size_t)b * a.D + d) * N + p] = val;
            }
            // overall mask: the reference overwrites it every plane, the LAST plane survives
            // (only the last plane's mask is ever visible, so it is not tracked in the view loop — 10 vector instructions per
            // view and plane — but recomputed here from the same projection expressions, once per pixel tile)
            if (a.mask != nullptr && d == a.D - 1) {
                bool any_inb = false, any_front = false;
                for (int k = 0; k < K; ++k) {
                    const float *hm = pb + kWsHom + 12 * k;
                    const float qx = fmaf(hm[0], pxf, fmaf(hm[1], pyf, hm[2]));
                    const float qy = fmaf(hm[3], pxf, fmaf(hm[4], pyf, hm[5]));
                    const float qz = fmaf(hm[6], pxf, fmaf(hm[7], pyf, hm[8]));
                    const float cx = fmaf(depth, qx, hm[9]);
                    const float cy = fmaf(depth, qy, hm[10]);
                    const float cz = fmaf(depth, qz, hm[11]);
                    const float z = fmaxf(cz, 1e-5f);
                    float r = __builtin_amdgcn_rcpf(z);
                    r = r * fmaf(-z, r, 2.0f);
                    const float u = cx * r, v = cy * r;
                    any_inb |= (u > 2.f) & (u < Wf - 2.f) & (v > 2.f) & (v < Hf - 2.f);
                    any_front |= z > 0.f;
                }
                if (q == 0 && live) a.mask[(size_t)b * N + p] = (any_front && any_inb) ? 1 : 0;
            }
            P = Pn; cur = curn; wc = wcn;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Generic variant: any source-view count up to IDH_MAX_SOURCE_VIEWS (FeatureVolumeManager takes num_source_views from
// model_num_views - 1, reference cost_volume.py:382-435 / depth_model.py:206-212) and matching_feature_dims = 16 * CB
// (options.py:138).  Same arithmetic as fv_mlp_k; what changes is where things live:
//   * W1's per-voxel blocks - K*CB feature blocks + 2J metadata blocks, up to 40 x 8 KiB - do not fit next to W2 in
//     LDS and are read through L1 / L2 as MFMA fragments (every wave of the chip reads the same 8 KiB per block);
//   * lane quarter q carries the metadata of views q, q+4, .. q+4(J-1), J = ceil(K/4) (2 for K <= 8), EIGHT slots per view
//     group j = two 16-column blocks: [valid, z, dot, ray angle | ray xyz, plane depth (group 0, quarter 0) / 0]
//     (implicit-depth_amd/cost_volume.py: feature_mlp_column_maps(layout="gen8")).  With eight slots a view group's
//     metadata MFMAs are issued inside ITS iteration of a run-time loop over j, so no metadata array lives across
//     iterations (round 4's seven-slot packing needed one indexed by the run-time group: 428-508 B of scratch per lane);
//   * as in fv_mlp_k (round 5) the projection of a view is computed once per voxel, by its owner quarter, and handed to
//     the other three through ds_bpermute_b32; the four views of a group are unrolled, the next view's tap loads are
//     issued before the current view's blend.
template <int CB>
__global__ __launch_bounds__(512) void fv_mlp_gen_k(const FvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *sW2 = reinterpret_cast<f32x4 *>(smem_raw);  // 8*8*64
    {
        const f32x4 *g2 = reinterpret_cast<const f32x4 *>(a.w2);
        for (int i = threadIdx.x; i < kNS * kNS * 64; i += 512) sW2[i] = g2[i];
    }
    __syncthreads();
    const f32x4 *gW1 = reinterpret_cast<const f32x4 *>(a.w1v);
    const float *s_b2 = a.vecs, *s_w3 = a.vecs + kHid;
    const float b3 = a.vecs[2 * kHid];
    constexpr int kCc = kC * CB;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, q = lane >> 4;
    const int N = a.H * a.W;
    const int K = a.K, J = a.J;
    const float Wf = (float)a.W, Hf = (float)a.H;
    const long long ntasks = (long long)a.B * a.tiles_per_img * a.G;
    const int bp0 = 4 * ln;  // ds_bpermute address of this voxel's lane in quarter 0 (+ 64 per quarter)

    for (long long task = (long long)idh_xcd_remap(blockIdx.x, gridDim.x) * 8 + wave; task < ntasks; task += (long long)gridDim.x * 8) {
        const int g = __builtin_amdgcn_readfirstlane((int)(task % a.G));
        const int tile = __builtin_amdgcn_readfirstlane((int)((task / a.G) % a.tiles_per_img));
        const int b = __builtin_amdgcn_readfirstlane((int)(task / ((long long)a.G * a.tiles_per_img)));
        const int p_raw = tile * 16 + ln;
        const bool live = p_raw < N;
        const int p = live ? p_raw : N - 1;
        const int py = p / a.W, px = p - py * a.W;
        const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
        const float *pb = a.ws + (size_t)b * kWsStrideReal;

        f32x4 cur4[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) cur4[cb] = *reinterpret_cast<const f32x4 *>(a.cur + (size_t)b * a.cur_bs + (size_t)p * kCc + 16 * cb + 4 * q);
        const float *iK = pb + kWsInvK;
        const float rx = fmaf(iK[0], pxf, fmaf(iK[1], pyf, iK[2]));
        const float ry = fmaf(iK[3], pxf, fmaf(iK[4], pyf, iK[5]));
        const float rz = fmaf(iK[6], pxf, fmaf(iK[7], pyf, iK[8]));

        f32x4 pre[kNS];
#pragma unroll
        for (int i = 0; i < kNS; ++i) pre[i] = *reinterpret_cast<const f32x4 *>(pb + kWsBias + 16 * i + 4 * q);
        const int d0 = g * a.DP, d1 = min(a.D, d0 + a.DP);
        if (d0 >= d1) continue;
        const float rn = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
        const float crx = rx / rn, cry = ry / rn, crz = rz / rn;
        {
            const f32x4 rayB = (q == 0) ? (f32x4){crx, cry, crz, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 *w1p = reinterpret_cast<const f32x4 *>(a.w1p);
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const f32x4 A0 = w1p[(cb * kNS + i) * 64 + lane];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) pre[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[kk], cur4[cb][kk], pre[i], 0, 0, 0);
                }
                const f32x4 A1 = w1p[(CB * kNS + i) * 64 + lane];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pre[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[kk], rayB[kk], pre[i], 0, 0, 0);
            }
        }
        // taps through a buffer descriptor of this frame's K source maps (32-bit offsets; view k: scalar offset k * N * kCc * 4)
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.src + (size_t)b * a.src_bs), 0, K * N * kCc * 4, 0x00020000);

        f32x4 ob = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool vec_ok = ((d0 & 3) == 0) && ((a.vol_cs & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.vol) & 15) == 0);
#pragma unroll 1
        for (int d = d0; d < d1; ++d) {
            const float depth = fv_plane(a, b, p, d);
            const float Xx = depth * rx, Xy = depth * ry, Xz = depth * rz;
            __builtin_amdgcn_s_setprio(2);  // (layer 1 high, layers 2 / 3 low: see fv_mlp_k)
            f32x4 acc1[kNS];
#pragma unroll
            for (int i = 0; i < kNS; ++i) acc1[i] = pre[i];
            bool any_inb = false, any_front = false;  // (of this quarter's own views; OR-ed over the quarters at the last plane)
#pragma unroll 1
            for (int j = 0; j < J; ++j) {
                // ---- this quarter's view of group j: projection, bilinear weights, packed tap address (pixel index | x step << 30 | y step << 31),
                // viewing ray and ray angle (cost_volume.py:630-659).  An absent view (q + 4j >= K) computes on a stand-in nobody reads.
                const int v = q + 4 * j, vs = min(v, K - 1);
                const bool mine = v < K;
                const float *hm = pb + kWsHom + 12 * vs;
                const float qx = fmaf(hm[0], pxf, fmaf(hm[1], pyf, hm[2]));
                const float qy = fmaf(hm[3], pxf, fmaf(hm[4], pyf, hm[5]));
                const float qz = fmaf(hm[6], pxf, fmaf(hm[7], pyf, hm[8]));
                const float cx = fmaf(depth, qx, hm[9]);
                const float cy = fmaf(depth, qy, hm[10]);
                const float cz = fmaf(depth, qz, hm[11]);
                const float z = fmaxf(cz, 1e-5f);
                float rc = __builtin_amdgcn_rcpf(z);
                rc = rc * fmaf(-z, rc, 2.0f);
                const float su = cx * rc, sv = cy * rc;
                any_inb |= mine & (su > 2.f) & (su < Wf - 2.f) & (sv > 2.f) & (sv < Hf - 2.f);
                any_front |= mine & (z > 0.f);
                const float sx = fminf(fmaxf(su - 0.5f, -1.0f), Wf);
                const float sy = fminf(fmaxf(sv - 0.5f, -1.0f), Hf);
                const float x0f = floorf(sx), y0f = floorf(sy);
                const float fx = sx - x0f, fy = sy - y0f;
                const int x0 = (int)x0f, y0 = (int)y0f;
                const float wx0 = (x0 >= 0 && x0 < a.W) ? 1.0f - fx : 0.f;
                const float wx1 = (x0 + 1 < a.W) ? fx : 0.f;
                const float wy0 = (y0 >= 0 && y0 < a.H) ? 1.0f - fy : 0.f;
                const float wy1 = (y0 + 1 < a.H) ? fy : 0.f;
                const int xa0 = min(max(x0, 0), a.W - 1), xa1 = min(x0 + 1, a.W - 1);
                const int ya0 = min(max(y0, 0), a.H - 1), ya1 = min(y0 + 1, a.H - 1);
                const int own_pk = (ya0 * a.W + xa0) | ((xa1 - xa0) << 30) | ((ya1 - ya0) << 31);
                const float own_w00 = wx0 * wy0, own_w01 = wx1 * wy0, own_w10 = wx0 * wy1, own_w11 = wx1 * wy1;
                float e0 = 0.f, e1 = 0.f, e2 = 0.f, ang = 0.f;
                {
                    const float *t = pb + kWsT + 4 * vs;
                    const float ax = Xx - t[0], ay = Xy - t[1], az = Xz - t[2];
                    const float in = 1.0f / fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-12f);
                    e0 = ax * in; e1 = ay * in; e2 = az * in;
                    const float n1 = fmaxf(sqrtf(crx * crx + cry * cry + crz * crz), 1e-5f);
                    const float n2 = fmaxf(sqrtf(e0 * e0 + e1 * e1 + e2 * e2), 1e-5f);
                    ang = (crx * e0 + cry * e1 + crz * e2) / (n1 * n2);
                }
                const float maskv = z > 0.f ? 1.f : 0.f;
                float own_dot = 0.f;

                struct Tap { f32x4 t00[CB], t01[CB], t10[CB], t11[CB]; };
                struct Wts { float w00, w01, w10, w11; };
                auto issue = [&](int kq) {  // view 4 j + kq: address from its owner quarter kq
                    const int pk = __builtin_amdgcn_ds_bpermute(bp0 + 64 * kq, own_pk);
                    const int o00 = (pk & 0x3fffffff) * (kCc * 4) + 16 * q;
                    const int o01 = o00 + ((pk >> 30) & 1) * (kCc * 4);
                    const int ystep = (int)((unsigned)pk >> 31) * (a.W * kCc * 4);
                    const int so = __builtin_amdgcn_readfirstlane((4 * j + kq) * N * (kCc * 4));
                    Tap t;
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        t.t00[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o00 + 64 * cb, so, 0));
                        t.t01[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o01 + 64 * cb, so, 0));
                        t.t10[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o00 + ystep + 64 * cb, so, 0));
                        t.t11[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, o01 + ystep + 64 * cb, so, 0));
                    }
                    return t;
                };
                auto weights = [&](int kq) {
                    auto bperm = [&](float x) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp0 + 64 * kq, __builtin_bit_cast(int, x))); };
                    Wts w;
                    w.w00 = bperm(own_w00); w.w01 = bperm(own_w01); w.w10 = bperm(own_w10); w.w11 = bperm(own_w11);
                    return w;
                };
                const int nv = min(4, K - 4 * j);  // views of this group (wave-uniform)
                Tap cur = issue(0);
                Wts wc = weights(0);
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    if (kq >= nv) break;
                    Tap nxt = cur;
                    Wts wn = wc;
                    if (kq + 1 < 4 && kq + 1 < nv) { nxt = issue(kq + 1); wn = weights(kq + 1); }
                    float part = 0.f;
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        f32x4 wv;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            wv[e] = fmaf(wc.w11, cur.t11[cb][e], fmaf(wc.w10, cur.t10[cb][e], fmaf(wc.w01, cur.t01[cb][e], wc.w00 * cur.t00[cb][e])));
                        // per-quarter partial of <warped, cur> in channel order: block cb's 4 channels of this quarter
                        float pc = wv[0] * cur4[cb][0];
                        pc = fmaf(wv[1], cur4[cb][1], pc); pc = fmaf(wv[2], cur4[cb][2], pc); pc = fmaf(wv[3], cur4[cb][3], pc);
                        part += pc;
#pragma unroll
                        for (int i = 0; i < kNS; ++i) {
                            const f32x4 A = gW1[((size_t)((4 * j + kq) * CB + cb) * kNS + i) * 64 + lane];
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], wv[kk], acc1[i], 0, 0, 0);
                        }
                    }
                    part += __shfl_xor(part, 16, 64);
                    part += __shfl_xor(part, 32, 64);
                    own_dot = (kq == q) ? part : own_dot;
                    cur = nxt;
                    wc = wn;
                }
                // ---- metadata of view group j: two blocks of four slots per quarter (zero weights where a view is absent, but keep the operands finite)
                const f32x4 mb0 = mine ? (f32x4){maskv, z, own_dot * maskv, ang} : (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 mb1 = mine ? (f32x4){e0, e1, e2, (j == 0 && q == 0) ? depth : 0.f} : (f32x4){0.f, 0.f, 0.f, (j == 0 && q == 0) ? depth : 0.f};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int i = 0; i < kNS; ++i) {
                        const f32x4 A = gW1[((size_t)(K * CB + 2 * j + c) * kNS + i) * 64 + lane];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], c == 0 ? mb0[kk] : mb1[kk], acc1[i], 0, 0, 0);
                    }
                }
            }
            f32x4 acc2[kNS];
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[i][r] = lrelu01(acc1[i][r]);
                acc2[i] = *reinterpret_cast<const f32x4 *>(s_b2 + 16 * i + 4 * q);
            }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int c = 0; c < kNS; ++c) {
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const f32x4 A = sW2[(c * kNS + i) * 64 + lane];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], acc1[c][kk], acc2[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
                const f32x4 w3 = *reinterpret_cast<const f32x4 *>(s_w3 + 16 * i + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc = fmaf(w3[r], lrelu01(acc2[i][r]), sacc);
            }
            sacc += __shfl_xor(sacc, 16, 64);
            sacc += __shfl_xor(sacc, 32, 64);
            const float val = sacc + b3;
            if (a.vol_cs > 0) {
                const int e = (d - d0) & 3;
                ob[0] = e == 0 ? val : ob[0]; ob[1] = e == 1 ? val : ob[1]; ob[2] = e == 2 ? val : ob[2]; ob[3] = e == 3 ? val : ob[3];
                if (q == 0 && live && (e == 3 || d == d1 - 1)) {
                    float *o = a.vol + ((size_t)b * N + p) * a.vol_cs + (d - e);
                    if (e == 3 && vec_ok) *reinterpret_cast<f32x4 *>(o) = ob;
                    else { o[0] = ob[0]; if (e >= 1) o[1] = ob[1]; if (e >= 2) o[2] = ob[2]; if (e >= 3) o[3] = ob[3]; }
                }
            } else if (q == 0 && live) {
                a.vol[((size_t)b * a.D + d) * N + p] = val;
            }
            // overall mask: the reference overwrites it every plane, the LAST plane survives; the flags of the four quarters' own views are OR-ed
            if (a.mask != nullptr && d == a.D - 1) {
                int fl = (any_inb ? 1 : 0) | (any_front ? 2 : 0);
                fl |= __shfl_xor(fl, 16, 64);
                fl |= __shfl_xor(fl, 32, 64);
                if (q == 0 && live) a.mask[(size_t)b * N + p] = (fl == 3) ? 1 : 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Split-precision variant (IDH "f16x3", see csrc/conv_split.hip for the arithmetic): the two
// 128-wide layers run on v_mfma_f32_16x16x32_f16 with every fp32 operand expanded into two
// round-to-nearest f16 pieces (x/s = x0 + x1, |err| <= 2^-23) and the three significant cross
// products accumulated in fp32.  Scaling is by exact powers of two: weights per hidden unit (row)
// at pack time, activations per voxel (= per MFMA column, so the scale factors out of the GEMM) from
// the maximum of that voxel's own input / hidden vector.  Operand order: a 32-wide K block is two
// consecutive 16-blocks of the fp32 kernel's order — lane quarter q holds k = 16(2c)+4q+e and
// 16(2c+1)+4q+e — and the voxel inputs are ordered [4 metadata blocks][K view blocks] so every
// register index is a compile-time constant.  C/D layout of 16x16x32 equals 16x16x4's, so the
// layer-1 accumulators are again exactly the layer-2 B operand (after scale + split).
// LDS: ceil((K+4)/2) x 16 KiB + 64 KiB = 160 KiB for K = 7, 8.
// KT = compile-time source-view count (7: every BDModel config, 8), 0 = run-time count: with KT > 0 the
// 8 unrolled view iterations lose their `k < K` guards and become ONE basic block the scheduler can
// software-pipeline across views.
template <int KT>
__global__ __launch_bounds__(512) void fv_mlp_f16_k(const FvArgs a, const float *__restrict__ sw1g, const float *__restrict__ sw2g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = KT > 0 ? KT : a.K;
    const int nb32 = (K + 5) / 2;                                 // 32-wide K blocks of layer 1
    u32x4 *sW1 = reinterpret_cast<u32x4 *>(smem_raw);             // nb32 * 8 * 2 * 64
    u32x4 *sW2 = sW1 + nb32 * kNS * 2 * 64;                       // 4 * 8 * 2 * 64
    {
        const u32x4 *g1 = reinterpret_cast<const u32x4 *>(a.w1v);
        const u32x4 *g2 = reinterpret_cast<const u32x4 *>(a.w2);
        const int n1 = nb32 * kNS * 2 * 64, n2 = 4 * kNS * 2 * 64;
        for (int i = threadIdx.x; i < n1; i += 512) sW1[i] = g1[i];
        for (int i = threadIdx.x; i < n2; i += 512) sW2[i] = g2[i];
    }
    __syncthreads();
    const float *s_b2 = a.vecs, *s_w3 = a.vecs + kHid;
    const float b3 = a.vecs[2 * kHid];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, q = lane >> 4;
    const int N = a.H * a.W;
    const float Wf = (float)a.W, Hf = (float)a.H;
    const long long ntasks = (long long)a.B * a.tiles_per_img * a.G;

    // XCD-aware: hardware puts workgroup b on XCD b % 8; after the remap each XCD walks a contiguous run of tasks
    // (= a band of image rows) per round, so the taps of its 32 workgroups share that XCD's 4 MB L2
    for (long long task = (long long)idh_xcd_remap(blockIdx.x, gridDim.x) * 8 + wave; task < ntasks; task += (long long)gridDim.x * 8) {
        const int g = __builtin_amdgcn_readfirstlane((int)(task % a.G));
        const int tile = __builtin_amdgcn_readfirstlane((int)((task / a.G) % a.tiles_per_img));
        const int b = __builtin_amdgcn_readfirstlane((int)(task / ((long long)a.G * a.tiles_per_img)));
        const int p_raw = tile * 16 + ln;
        const bool live = p_raw < N;
        const int p = live ? p_raw : N - 1;
        const int py = p / a.W, px = p - py * a.W;
        const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
        const float *pb = a.ws + (size_t)b * kWsStrideReal;

        const f32x4 cur4 = *reinterpret_cast<const f32x4 *>(a.cur + (size_t)b * a.cur_bs + (size_t)p * kC + 4 * q);
        const float *iK = pb + kWsInvK;
        const float rx = fmaf(iK[0], pxf, fmaf(iK[1], pyf, iK[2]));
        const float ry = fmaf(iK[3], pxf, fmaf(iK[4], pyf, iK[5]));
        const float rz = fmaf(iK[6], pxf, fmaf(iK[7], pyf, iK[8]));

        // per-pixel pre-activation (fp32 MFMA, once per task): bias_b + W1[:,cur].cur + W1[:,cur_ray].ray
        f32x4 pre[kNS];
#pragma unroll
        for (int i = 0; i < kNS; ++i) pre[i] = *reinterpret_cast<const f32x4 *>(pb + kWsBias + 16 * i + 4 * q);
        const int d0 = g * a.DP, d1 = min(a.D, d0 + a.DP);
        if (d0 >= d1) continue;
        const float rn = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
        const float crx = rx / rn, cry = ry / rn, crz = rz / rn;
        {
            const f32x4 rayB = (q == 0) ? (f32x4){crx, cry, crz, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 *w1p = reinterpret_cast<const f32x4 *>(a.w1p);
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
                const f32x4 A0 = w1p[(0 * kNS + i) * 64 + lane];
                const f32x4 A1 = w1p[(1 * kNS + i) * 64 + lane];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pre[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[kk], cur4[kk], pre[i], 0, 0, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pre[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[kk], rayB[kk], pre[i], 0, 0, 0);
            }
        }

        f32x4 ob = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool vec_ok = ((d0 & 3) == 0) && ((a.vol_cs & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.vol) & 15) == 0);
#pragma unroll 1
        for (int d = d0; d < d1; ++d) {
            const float depth = fv_plane(a, b, p, d);
            const float Xx = depth * rx, Xy = depth * ry, Xz = depth * rz;
            __builtin_amdgcn_s_setprio(2);  // (gather and layer 1 high, layers 2 / 3 low: see fv_mlp_k; dropping before layer 1 instead: half the gain)
            f32x4 X[12];  // [0..3] metadata blocks, [4 + k] warped features of view k
#pragma unroll
            for (int j = 0; j < 12; ++j) X[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float m0 = 0.f, m1 = 0.f, m2 = 0.f, m7 = 0.f, m8 = 0.f, m9 = 0.f;
            bool any_inb = false, any_front = false;
            struct Tap {
                f32x4 t00, t01, t10, t11;
                float w00, w01, w10, w11, z, u, v;
            };
            auto issue = [&](int k) {
                Tap t;
                const float *hm = pb + kWsHom + 12 * k;
                const float qx = fmaf(hm[0], pxf, fmaf(hm[1], pyf, hm[2]));
                const float qy = fmaf(hm[3], pxf, fmaf(hm[4], pyf, hm[5]));
                const float qz = fmaf(hm[6], pxf, fmaf(hm[7], pyf, hm[8]));
                const float cx = fmaf(depth, qx, hm[9]);
                const float cy = fmaf(depth, qy, hm[10]);
                const float cz = fmaf(depth, qz, hm[11]);
                t.z = fmaxf(cz, 1e-5f);
                float r = __builtin_amdgcn_rcpf(t.z);
                r = r * fmaf(-t.z, r, 2.0f);
                t.u = cx * r;
                t.v = cy * r;
                const float sx = fminf(fmaxf(t.u - 0.5f, -1.0f), Wf);
                const float sy = fminf(fmaxf(t.v - 0.5f, -1.0f), Hf);
                const float x0f = floorf(sx), y0f = floorf(sy);
                const float fx = sx - x0f, fy = sy - y0f;
                const int x0 = (int)x0f, y0 = (int)y0f;
                const float wx0 = (x0 >= 0 && x0 < a.W) ? 1.0f - fx : 0.f;
                const float wx1 = (x0 + 1 < a.W) ? fx : 0.f;
                const float wy0 = (y0 >= 0 && y0 < a.H) ? 1.0f - fy : 0.f;
                const float wy1 = (y0 + 1 < a.H) ? fy : 0.f;
                const int xa0 = min(max(x0, 0), a.W - 1), xa1 = min(x0 + 1, a.W - 1);
                const int ya0 = min(max(y0, 0), a.H - 1), ya1 = min(y0 + 1, a.H - 1);
                const float *sb = a.src + (size_t)b * a.src_bs + (size_t)k * N * kC;  // wave-uniform
#ifdef IDH_ABL_NOGATHER
                t.t00 = t.t01 = t.t10 = t.t11 = (f32x4){(float)xa0, (float)ya0, (float)xa1, (float)ya1 + sb[0]};
#else
                // 32-bit offsets from a wave-uniform base: the loads take the SGPR-base + VGPR-offset form
                const int r0 = ya0 * a.W, r1 = ya1 * a.W;
                t.t00 = *reinterpret_cast<const f32x4 *>(sb + (r0 + xa0) * kC + 4 * q);
                t.t01 = *reinterpret_cast<const f32x4 *>(sb + (r0 + xa1) * kC + 4 * q);
                t.t10 = *reinterpret_cast<const f32x4 *>(sb + (r1 + xa0) * kC + 4 * q);
                t.t11 = *reinterpret_cast<const f32x4 *>(sb + (r1 + xa1) * kC + 4 * q);
#endif
                t.w00 = wx0 * wy0; t.w01 = wx1 * wy0; t.w10 = wx0 * wy1; t.w11 = wx1 * wy1;
                return t;
            };
            Tap cur = issue(0);
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (KT > 0 ? k < KT : k < K) {  // compile-time when KT > 0, else wave-uniform
                    const Tap nxt = issue(min(k + 1, K - 1));
                    any_inb |= (cur.u > 2.f) & (cur.u < Wf - 2.f) & (cur.v > 2.f) & (cur.v < Hf - 2.f);
                    const float z = cur.z;
                    const float maskv = z > 0.f ? 1.f : 0.f;
                    any_front |= z > 0.f;
                    f32x4 wv;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        wv[e] = fmaf(cur.w11, cur.t11[e], fmaf(cur.w10, cur.t10[e], fmaf(cur.w01, cur.t01[e], cur.w00 * cur.t00[e])));
                    float part = wv[0] * cur4[0];
                    part = fmaf(wv[1], cur4[1], part); part = fmaf(wv[2], cur4[2], part); part = fmaf(wv[3], cur4[3], part);
                    part += __shfl_xor(part, 16, 64);
                    part += __shfl_xor(part, 32, 64);
                    const float dotv = part * maskv;
                    const bool s0 = (k == q), s1 = (k == q + 4);
                    m0 = s0 ? maskv : m0; m1 = s0 ? z : m1; m2 = s0 ? dotv : m2;
                    m7 = s1 ? maskv : m7; m8 = s1 ? z : m8; m9 = s1 ? dotv : m9;
                    X[4 + k] = wv;
                    cur = nxt;
                }
            }
            float m3 = 0.f, m4 = 0.f, m5 = 0.f, m6 = 0.f, m10 = 0.f, m11 = 0.f, m12 = 0.f, m13 = 0.f;
            {
                const int v0 = q, v1 = q + 4;
                if (v0 < K) {
                    const float *t = pb + kWsT + 4 * v0;
                    const float ax = Xx - t[0], ay = Xy - t[1], az = Xz - t[2];
                    const float in = 1.0f / fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-12f);
                    m4 = ax * in; m5 = ay * in; m6 = az * in;
                    const float n1 = fmaxf(sqrtf(crx * crx + cry * cry + crz * crz), 1e-5f);
                    const float n2 = fmaxf(sqrtf(m4 * m4 + m5 * m5 + m6 * m6), 1e-5f);
                    m3 = (crx * m4 + cry * m5 + crz * m6) / (n1 * n2);
                }
                if (v1 < K) {
                    const float *t = pb + kWsT + 4 * v1;
                    const float ax = Xx - t[0], ay = Xy - t[1], az = Xz - t[2];
                    const float in = 1.0f / fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-12f);
                    m11 = ax * in; m12 = ay * in; m13 = az * in;
                    const float n1 = fmaxf(sqrtf(crx * crx + cry * cry + crz * crz), 1e-5f);
                    const float n2 = fmaxf(sqrtf(m11 * m11 + m12 * m12 + m13 * m13), 1e-5f);
                    m10 = (crx * m11 + cry * m12 + crz * m13) / (n1 * n2);
                }
            }
            X[0] = (f32x4){m0, m1, m2, m3};
            X[1] = (f32x4){m4, m5, m6, m7};
            X[2] = (f32x4){m8, m9, m10, m11};
            X[3] = (f32x4){m12, m13, depth, 0.f};

            // ---- layer 1: scale + split the voxel's inputs, 3 f16 products per K block ------------
            f32x4 h[kNS];
            {
                const int ex = column_exponent<12>(X);
                const float mul = exp2_int(14 - ex), sx = exp2_int(ex - 14);
                f32x4 acc[kNS];
#pragma unroll
                for (int i = 0; i < kNS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if (c < nb32) {
                        u32x4 Bh, Bl;  // split right before use: X dies block by block
                        split_block(X[2 * c], X[2 * c + 1], mul, Bh, Bl);
#pragma unroll
                        for (int i4 = 0; i4 < kNS; i4 += 4) {
                            u32x4 Ah[4], Al[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                Ah[i] = sW1[((c * kNS + i4 + i) * 2 + 0) * 64 + lane];
                                Al[i] = sW1[((c * kNS + i4 + i) * 2 + 1) * 64 + lane];
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i4 + i] = mfma_f16(Al[i], Bh, acc[i4 + i]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i4 + i] = mfma_f16(Ah[i], Bl, acc[i4 + i]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i4 + i] = mfma_f16(Ah[i], Bh, acc[i4 + i]);
                            __builtin_amdgcn_sched_barrier(0);  // no hoisting of later blocks' LDS reads (register pressure)
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const f32x4 sw = *reinterpret_cast<const f32x4 *>(sw1g + 16 * i + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[i][r] = lrelu01_sel(fmaf(acc[i][r], sx * sw[r], pre[i][r]));
                }
            }
            __builtin_amdgcn_s_setprio(0);
            // ---- layer 2 (same scheme on the hidden vector) -> LeakyReLU -> layer 3 ----------------
            float s = 0.f;
            {
                const int ex = column_exponent<kNS>(h);
                const float mul = exp2_int(14 - ex), sx = exp2_int(ex - 14);
                f32x4 acc[kNS];
#pragma unroll
                for (int i = 0; i < kNS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    u32x4 Bh, Bl;
                    split_block(h[2 * c], h[2 * c + 1], mul, Bh, Bl);
#pragma unroll
                    for (int i4 = 0; i4 < kNS; i4 += 4) {
                        u32x4 Ah[4], Al[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            Ah[i] = sW2[((c * kNS + i4 + i) * 2 + 0) * 64 + lane];
                            Al[i] = sW2[((c * kNS + i4 + i) * 2 + 1) * 64 + lane];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i4 + i] = mfma_f16(Al[i], Bh, acc[i4 + i]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i4 + i] = mfma_f16(Ah[i], Bl, acc[i4 + i]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i4 + i] = mfma_f16(Ah[i], Bh, acc[i4 + i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const f32x4 sw = *reinterpret_cast<const f32x4 *>(sw2g + 16 * i + 4 * q);
                    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(s_b2 + 16 * i + 4 * q);
                    const f32x4 w3 = *reinterpret_cast<const f32x4 *>(s_w3 + 16 * i + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s = fmaf(w3[r], lrelu01_sel(fmaf(acc[i][r], sx * sw[r], b2[r])), s);
                }
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float val = s + b3;
            if (a.vol_cs > 0) {
                const int e = (d - d0) & 3;
                ob[0] = e == 0 ? val : ob[0]; ob[1] = e == 1 ? val : ob[1]; ob[2] = e == 2 ? val : ob[2]; ob[3] = e == 3 ? val : ob[3];
                if (q == 0 && live && (e == 3 || d == d1 - 1)) {
                    float *o = a.vol + ((size_t)b * N + p) * a.vol_cs + (d - e);
                    if (e == 3 && vec_ok) *reinterpret_cast<f32x4 *>(o) = ob;
                    else { o[0] = ob[0]; if (e >= 1) o[1] = ob[1]; if (e >= 2) o[2] = ob[2]; if (e >= 3) o[3] = ob[3]; }
                }
            } else if (q == 0 && live) {
                a.vol[((size_t)b * a.D + d) * N + p] = val;
            }
            if (q == 0 && live && a.mask != nullptr && d == a.D - 1) a.mask[(size_t)b * N + p] = (any_front && any_inb) ? 1 : 0;
        }
    }
}

// lowest[b,p] = plane_{argmax_d vol[b,d,p]} (first maximum wins), reference cost_volume.py:352-356
__global__ __launch_bounds__(256) void argmax_planes_k(const float *__restrict__ vol, int vol_cs, int B, int N, int D,
                                                       float dmin, float dmax, float *__restrict__ lowest,
                                                       float *__restrict__ planes_out, const float *__restrict__ planes,
                                                       long long planes_sb, long long planes_sd, int planes_sp) {
    auto plane_of = [&](long long t, int bi) {
        if (planes == nullptr) return fv_depth_plane(bi, D, dmin, dmax);
        const long long b = t / N, p = t - b * N;
        return planes[b * planes_sb + (long long)bi * planes_sd + p * planes_sp];
    };
    const long long total = (long long)B * N;
    if (planes_out && planes == nullptr && blockIdx.x == 0)
        for (int i = threadIdx.x; i < D; i += 256) planes_out[i] = fv_depth_plane(i, D, dmin, dmax);
    if (vol_cs > 0 && (vol_cs & 3) == 0 && (D & 3) == 0 && (reinterpret_cast<uintptr_t>(vol) & 15) == 0) {
        // NHWC: 16 lanes share a pixel and read its D planes as coalesced 16-byte pieces (a lane per pixel would
        // stride 4*cs bytes between lanes: 2.5x read amplification at the HBM counters)
        const int sub = threadIdx.x & 15;
        for (long long t = blockIdx.x * 16ll + (threadIdx.x >> 4); t < total; t += gridDim.x * 16ll) {
            float best = -INFINITY;
            int bi = 0;
            for (int d0 = 4 * sub; d0 < D; d0 += 64) {
                const float4 v = *reinterpret_cast<const float4 *>(vol + t * vol_cs + d0);
                if (v.x > best) { best = v.x; bi = d0; }
                if (v.y > best) { best = v.y; bi = d0 + 1; }
                if (v.z > best) { best = v.z; bi = d0 + 2; }
                if (v.w > best) { best = v.w; bi = d0 + 3; }
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {  // first maximum wins: on ties keep the lower plane index
                const float ov = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (sub == 0) lowest[t] = plane_of(t, bi);
        }
        return;
    }
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const long long b = t / N, p = t - b * N;
        float best = -INFINITY;
        int bi = 0;
        for (int d = 0; d < D; ++d) {
            const float v = vol_cs > 0 ? vol[t * vol_cs + d] : vol[(b * D + d) * N + p];
            if (v > best) { best = v; bi = d; }
        }
        lowest[t] = plane_of(t, bi);
    }
}

}  // namespace

extern "C" size_t idh_feature_volume_workspace_bytes(int B) { return B <= 0 ? 0 : (size_t)B * kWsStrideReal * sizeof(float); }

static int feature_volume_impl(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                               const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                               float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                               const void *w1_voxel_packed, const float *w1_pixel_packed,
                               const float *w1_pose_rowmajor, const float *b1, const void *w2_packed,
                               const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                               unsigned char *mask_bhw, float *planes_d, void *workspace,
                               size_t workspace_bytes, void *stream, bool f16x3, const idh_volume_opts *opts);

extern "C" int idh_feature_volume_f16x3_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                                            const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                                            float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                                            const void *w1_voxel_f16, const float *w1_pixel_packed,
                                            const float *w1_pose_rowmajor, const float *b1, const void *w2_f16,
                                            const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                                            unsigned char *mask_bhw, float *planes_d, void *workspace,
                                            size_t workspace_bytes, void *stream) {
    return feature_volume_impl(cur_nhwc, src_nhwc, src_K_44, src_E_44, src_poses_44, cur_invK_44, dmin, dmax, B, K, C, H, W, D,
                               w1_voxel_f16, w1_pixel_packed, w1_pose_rowmajor, b1, w2_f16, vecs_b2_w3_b3, vol, vol_nhwc_cs,
                               lowest_bhw, mask_bhw, planes_d, workspace, workspace_bytes, stream, true, nullptr);
}

extern "C" int idh_feature_volume_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                                      const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                                      float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                                      const float *w1_voxel_packed, const float *w1_pixel_packed,
                                      const float *w1_pose_rowmajor, const float *b1, const float *w2_packed,
                                      const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                                      unsigned char *mask_bhw, float *planes_d, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    return feature_volume_impl(cur_nhwc, src_nhwc, src_K_44, src_E_44, src_poses_44, cur_invK_44, dmin, dmax, B, K, C, H, W, D,
                               w1_voxel_packed, w1_pixel_packed, w1_pose_rowmajor, b1, w2_packed, vecs_b2_w3_b3, vol, vol_nhwc_cs,
                               lowest_bhw, mask_bhw, planes_d, workspace, workspace_bytes, stream, false, nullptr);
}

extern "C" int idh_feature_volume_ex_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                                         const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                                         float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                                         const void *w1_voxel, const float *w1_pixel_packed,
                                         const float *w1_pose_rowmajor, const float *b1, const void *w2,
                                         const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                                         unsigned char *mask_bhw, float *planes_d, void *workspace,
                                         size_t workspace_bytes, int f16x3, const idh_volume_opts *opts, void *stream) {
    return feature_volume_impl(cur_nhwc, src_nhwc, src_K_44, src_E_44, src_poses_44, cur_invK_44, dmin, dmax, B, K, C, H, W, D,
                               w1_voxel, w1_pixel_packed, w1_pose_rowmajor, b1, w2, vecs_b2_w3_b3, vol, vol_nhwc_cs,
                               lowest_bhw, mask_bhw, planes_d, workspace, workspace_bytes, stream, f16x3 != 0, opts);
}

static int feature_volume_impl(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                               const float *src_E_44, const float *src_poses_44, const float *cur_invK_44,
                               float dmin, float dmax, int B, int K, int C, int H, int W, int D,
                               const void *w1_voxel_packed, const float *w1_pixel_packed,
                               const float *w1_pose_rowmajor, const float *b1, const void *w2_packed,
                               const float *vecs_b2_w3_b3, float *vol, int vol_nhwc_cs, float *lowest_bhw,
                               unsigned char *mask_bhw, float *planes_d, void *workspace,
                               size_t workspace_bytes, void *stream, bool f16x3, const idh_volume_opts *opts) {
    const bool own_planes = opts && opts->planes;
    if (own_planes) { dmin = dmax = 1.f; planes_d = nullptr; }
    if (B < 0 || K <= 0 || H <= 0 || W <= 0 || D <= 0 || !(dmin > 0.f) || !(dmax > 0.f)) return IDH_EINVAL;
    const bool generic = C != kC || K > kMaxK;  // fv_mlp_gen_k: matching_feature_dims 32, up to IDH_MAX_SOURCE_VIEWS views
    if ((long long)K * H * W * C * 4 >= (1ll << 31)) return IDH_EUNSUPPORTED;  // (taps through a buffer descriptor per frame: 32-bit byte offsets)
    if ((C != kC && C != 2 * kC) || K > IDH_MAX_SOURCE_VIEWS || D > 4096 || (generic && f16x3)) return IDH_EUNSUPPORTED;
    if (B == 0) return IDH_OK;
    if (!cur_nhwc || !src_nhwc || !src_K_44 || !src_E_44 || !src_poses_44 || !cur_invK_44 || !w1_voxel_packed ||
        !w1_pixel_packed || !w1_pose_rowmajor || !b1 || !w2_packed || !vecs_b2_w3_b3 || !vol)
        return IDH_EINVAL;
    if (vol_nhwc_cs != 0 && vol_nhwc_cs < D) return IDH_EINVAL;
    if (!workspace || workspace_bytes < idh_feature_volume_workspace_bytes(B)) return IDH_EWORKSPACE;
    hipStream_t st = idh_stream(stream);
    float *ws = static_cast<float *>(workspace);
    hipLaunchKernelGGL(fv_setup_k, dim3(B), dim3(128), 0, st, src_K_44, src_E_44, src_poses_44, cur_invK_44,
                       w1_pose_rowmajor, b1, K, ws);
    IDH_CHECK_LAUNCH();

    FvArgs a{};
    a.cur = cur_nhwc; a.src = src_nhwc; a.ws = ws; a.w1v = static_cast<const float *>(w1_voxel_packed); a.w1p = w1_pixel_packed;
    a.w2 = static_cast<const float *>(w2_packed);
    a.vecs = vecs_b2_w3_b3; a.vol = vol; a.mask = mask_bhw; a.vol_cs = vol_nhwc_cs;
    a.B = B; a.K = K; a.H = H; a.W = W; a.D = D; a.dmin = dmin; a.dmax = dmax;
    const int N = H * W;
    a.cur_bs = (long long)N * C; a.src_bs = (long long)K * N * C;
    a.J = K <= kMaxK ? 2 : (K + 3) / 4;  // views per lane quarter (the packed W1 columns follow the same rule: feature_mlp_column_maps)
    a.MB = 2 * a.J;  // fv_mlp_gen_k: eight metadata slots per view group (two 16-column blocks)
    if (opts) {
        if (opts->cur_batch_stride) a.cur_bs = opts->cur_batch_stride;
        if (opts->src_batch_stride) a.src_bs = opts->src_batch_stride;
        if (a.cur_bs < (long long)N * C || a.src_bs < (long long)K * N * C || (a.cur_bs & 3) || (a.src_bs & 3)) return IDH_EINVAL;
        if (opts->planes) {
            if (opts->planes_pixel_stride != 0 && opts->planes_pixel_stride != 1) return IDH_EINVAL;
            a.planes = opts->planes; a.planes_sb = opts->planes_batch_stride; a.planes_sd = opts->planes_plane_stride;
            a.planes_sp = opts->planes_pixel_stride;
        }
    }
    a.tiles_per_img = (N + 15) / 16;
    // plane groups: tasks are uniform in cost and run on 256 CUs x 8 persistent waves, so pick the
    // number of groups G (>= 4 planes per task, so the per-pixel pre-activation stays amortised) that
    // wastes the least of the last round: minimise ceil(tasks/slots)*slots/tasks, smallest G on ties.
    const long long pix_tasks = (long long)B * a.tiles_per_img;
    const long long slots = 256ll * 8;
    int bestG = 1;
    double bestWaste = 1e30;
    for (int G = 1; G <= (D + 3) / 4; ++G) {
        const int DP = (D + G - 1) / G;
        const int Gr = (D + DP - 1) / DP;
        if (Gr != G) continue;
        // planes are not always divisible: cost of a task ~ DP (+1 for the pre-activation)
        const long long tasks = pix_tasks * G;
        const double rounds = (double)((tasks + slots - 1) / slots);
        const double waste = rounds * slots * (DP + 1) / ((double)pix_tasks * (D + G));
        if (waste < bestWaste - 1e-9) { bestWaste = waste; bestG = G; }
    }
    a.G = bestG;
    a.DP = (D + bestG - 1) / bestG;
    const long long ntasks = pix_tasks * a.G;
    int grid = (int)((ntasks + 7) / 8);
    if (grid > 256) grid = 256;  // persistent: one 512-thread workgroup per CU (LDS-resident weights)
    static IdhDeviceOnce attr_set;
    if (attr_set.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_gen_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_gen_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_k<7>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_k<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_f16_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_f16_k<7>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fv_mlp_f16_k<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return IDH_ELAUNCH;
        attr_set.mark();
    }
    if (generic) {
        const size_t lds = (size_t)kNS * kNS * 64 * sizeof(f32x4);
        if (C == kC) hipLaunchKernelGGL(fv_mlp_gen_k<1>, dim3(grid), dim3(512), lds, st, a);
        else hipLaunchKernelGGL(fv_mlp_gen_k<2>, dim3(grid), dim3(512), lds, st, a);
    } else if (f16x3) {
        const int nb32 = (K + 5) / 2;
        const size_t lds = ((size_t)nb32 * kNS * 2 * 64 + 4 * kNS * 2 * 64) * 16;
        const float *sw1 = reinterpret_cast<const float *>(static_cast<const char *>(w1_voxel_packed) + (size_t)nb32 * kNS * 2 * 64 * 16);
        const float *sw2 = reinterpret_cast<const float *>(static_cast<const char *>(w2_packed) + (size_t)4 * kNS * 2 * 64 * 16);
        if (K == 7) hipLaunchKernelGGL(fv_mlp_f16_k<7>, dim3(grid), dim3(512), lds, st, a, sw1, sw2);
        else if (K == 8) hipLaunchKernelGGL(fv_mlp_f16_k<8>, dim3(grid), dim3(512), lds, st, a, sw1, sw2);
        else hipLaunchKernelGGL(fv_mlp_f16_k<0>, dim3(grid), dim3(512), lds, st, a, sw1, sw2);
    } else {
        const size_t lds = ((size_t)(K + 4) * kNS * 64 + kNS * kNS * 64) * sizeof(f32x4);
        if (K == 7) hipLaunchKernelGGL(fv_mlp_k<7>, dim3(grid), dim3(512), lds, st, a);
        else if (K == 8) hipLaunchKernelGGL(fv_mlp_k<8>, dim3(grid), dim3(512), lds, st, a);  // (BASELINE.json's literal 8 source views)
        else hipLaunchKernelGGL(fv_mlp_k<0>, dim3(grid), dim3(512), lds, st, a);
    }
    IDH_CHECK_LAUNCH();
    if (lowest_bhw) {
        int g2 = idh_cdiv((long long)B * N, vol_nhwc_cs > 0 ? 16 : 256);
        if (g2 > 8192) g2 = 8192;
        hipLaunchKernelGGL(argmax_planes_k, dim3(g2), dim3(256), 0, st, vol, vol_nhwc_cs, B, N, D, dmin, dmax, lowest_bhw,
                           planes_d, a.planes, a.planes_sb, a.planes_sd, a.planes_sp);
        IDH_CHECK_LAUNCH();
    }
    return IDH_OK;
}
