// Fused plane-sweep warp + dot-product matching (gfx950).
//
// Replaces the reference's per-plane Python loop (modules/cost_volume.py:287-313) —
// BackprojectDepth -> repeat_interleave -> Project3D -> grid_sample -> mul/sum/mask/sum ->
// cat -> argmax/gather — with ONE launch that never materialises a warped feature map.
//
// Arithmetic follows SURVEY.md §8(a'):
//   q_k   = (P_k[:3,:3] invK[:3,:3]) (x+.5, y+.5, 1)^T         (homography, per pixel & view)
//   c     = depth_d * q_k + P_k[:3,3];  z = max(c_z, 1e-5);  (u,v) = c_xy / z
//   (sx,sy) = (u-.5, v-.5)  -> 4 bilinear taps, taps outside the image contribute 0
//   cost[b,d,y,x] = sum_k sum_c cur[c] * tap-blend(src_k)[c]
// The reference's "mask = depth > 0" is identically 1 because depth is clamped to >= 1e-5
// first (geometry_utils.py:86, cost_volume.py:216) — reproduced by construction.
//
// Data layout: features are NHWC with C = 16, so one tap = one 64-byte line segment read as
// 4 x dwordx4.  Work decomposition: a 256-thread workgroup owns 32 consecutive pixels and ALL
// D planes: thread (px, g) sweeps planes [g*DP, (g+1)*DP) for its pixel, g = 0..7, so the
// arg-max over planes finishes inside the workgroup (LDS reduce, first maximum wins) and the
// cost volume is written exactly once.  (cv_dot_k below keeps this one-lane-per-tap form; the launcher
// prefers cv_dot_quad_k, the quad-coalesced form further down, whenever 32-bit tap offsets suffice.)  The per-(b,k) 3x4 homographies are built once per
// workgroup into LDS and read back as same-address (broadcast) LDS reads.
#include <cstddef>
#include <stdlib.h>

#include "idh_common.h"

namespace {

constexpr int kC = 16;
constexpr int kTilePx = 32;
constexpr int kGroups = 8;
constexpr int kMaxPlanes = 512;

// strides / caller-supplied planes (idh_volume_opts resolved to concrete values by the launcher)
struct CvExt {
    const float *planes;          // null: log-spaced planes from dmin/dmax (s_planes)
    long long planes_sb, planes_sd;
    int planes_sp;                // 0: planes constant over the image, 1: per-pixel (B,D,H,W)
    long long cur_bs, src_bs;     // floats between consecutive batch elements
};

// ---- per-workgroup prologue: homographies + depth planes into LDS ------------------------
// One thread per source view builds the 3x4 map  [M | t] = [P[:3,:3] invK[:3,:3] | P[:3,3]],
// P = K_src E (geometry_utils.py:82); ~100 flops per view, redundant per workgroup but it
// removes a separate launch and any workspace.
__device__ __forceinline__ void build_homography(const float *__restrict__ Km, const float *__restrict__ Em,
                                                 const float *__restrict__ iK, float *__restrict__ o) {
    float P[3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int m = 0; m < 4; ++m) s = fmaf(Km[i * 4 + m], Em[m * 4 + j], s);
            P[i][j] = s;
        }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int m = 0; m < 3; ++m) s = fmaf(P[i][m], iK[m * 4 + j], s);
            o[i * 3 + j] = s;
        }
        o[9 + i] = P[i][3];
    }
}

// depth planes: exp(log(dmin) + log(dmax/dmin) * linspace(0,1,D)) (cost_volume.py:123-126);
// torch's linspace is evaluated from both ends (start + i*step below the middle,
// end - (D-1-i)*step above it).
__device__ __forceinline__ float depth_plane(int i, int D, float dmin, float dmax) {
    float ramp = 0.f;
    if (D > 1) {
        const float step = 1.0f / (float)(D - 1);
        ramp = (i < D / 2) ? step * (float)i : 1.0f - step * (float)(D - 1 - i);
    }
    return expf(logf(dmin) + logf(dmax / dmin) * ramp);
}

__device__ __forceinline__ float dot16(const float4 &a0, const float4 &a1, const float4 &a2,
                                       const float4 &a3, const float4 *__restrict__ p) {
    const float4 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    float s = a0.x * b0.x;
    s = fmaf(a0.y, b0.y, s); s = fmaf(a0.z, b0.z, s); s = fmaf(a0.w, b0.w, s);
    s = fmaf(a1.x, b1.x, s); s = fmaf(a1.y, b1.y, s); s = fmaf(a1.z, b1.z, s); s = fmaf(a1.w, b1.w, s);
    s = fmaf(a2.x, b2.x, s); s = fmaf(a2.y, b2.y, s); s = fmaf(a2.z, b2.z, s); s = fmaf(a2.w, b2.w, s);
    s = fmaf(a3.x, b3.x, s); s = fmaf(a3.y, b3.y, s); s = fmaf(a3.z, b3.z, s); s = fmaf(a3.w, b3.w, s);
    return s;
}

// CQ = matching feature channels / 16: 1 for every shipped configuration (options.py:138 matching_feature_dims = 16); 2 / 4 cover
// matching_feature_dims = 32 / 64 on this kernel only (the quad / window kernels are specialised for 64-byte texels)
template <int CQ>
__global__ __launch_bounds__(256) void cv_dot_k(const float *__restrict__ cur,   // B,N,16 CQ
                                                const float *__restrict__ src,   // B,K,N,16 CQ
                                                const float *__restrict__ src_K, // B,K,4,4
                                                const float *__restrict__ src_E, // B,K,4,4
                                                const float *__restrict__ cur_invK,  // B,4,4
                                                float dmin, float dmax,
                                                int B, int K, int H, int W, int D, int tiles_per_img,
                                                int cost_cs,                     // 0: (B,D,N) planes; >0: NHWC, floats per pixel
                                                float *__restrict__ cost,
                                                float *__restrict__ lowest,      // B,N or null
                                                float *__restrict__ planes_out,  // D or null
                                                const CvExt ext) {
    __shared__ float s_planes[kMaxPlanes];
    __shared__ __attribute__((aligned(16))) float s_h[IDH_MAX_SOURCE_VIEWS][12];
    __shared__ float s_best[kGroups][kTilePx];
    __shared__ int s_bidx[kGroups][kTilePx];

    const int N = H * W;
    const unsigned lin = idh_xcd_remap(blockIdx.x, gridDim.x);
    const int b = lin / tiles_per_img;
    const int tile = lin - b * tiles_per_img;
    const int px = threadIdx.x & (kTilePx - 1);
    const int g = threadIdx.x >> 5;

    if (threadIdx.x < K)
        build_homography(src_K + (size_t)(b * K + threadIdx.x) * 16, src_E + (size_t)(b * K + threadIdx.x) * 16,
                         cur_invK + (size_t)b * 16, s_h[threadIdx.x]);
    for (int i = threadIdx.x; i < D; i += 256) {
        const float dp = depth_plane(i, D, dmin, dmax);
        s_planes[i] = dp;
        if (planes_out != nullptr && blockIdx.x == 0) planes_out[i] = dp;
    }
    __syncthreads();

    const int p_raw = tile * kTilePx + px;
    const bool live = p_raw < N;
    const int p = live ? p_raw : N - 1;
    const int y = p / W, x = p - y * W;
    const float pxf = (float)x + 0.5f, pyf = (float)y + 0.5f;

    constexpr int kCc = kC * CQ;  // channels of this instantiation
    const float4 *cp = reinterpret_cast<const float4 *>(cur + (size_t)b * ext.cur_bs + (size_t)p * kCc);
    const float *pl = ext.planes ? ext.planes + (size_t)b * ext.planes_sb + (size_t)p * ext.planes_sp : nullptr;
    float4 cc[4 * CQ];
#pragma unroll
    for (int i = 0; i < 4 * CQ; ++i) cc[i] = cp[i];
    auto tapdot = [&](const float *t) {  // <cur, texel> in channel order, one fmaf chain (as dot16)
        const float4 *tp = reinterpret_cast<const float4 *>(t);
        float sacc = dot16(cc[0], cc[1], cc[2], cc[3], tp);
#pragma unroll
        for (int i = 1; i < CQ; ++i) {
            const float4 b0 = tp[4 * i], b1 = tp[4 * i + 1], b2 = tp[4 * i + 2], b3 = tp[4 * i + 3];
            const float4 a0 = cc[4 * i], a1 = cc[4 * i + 1], a2 = cc[4 * i + 2], a3 = cc[4 * i + 3];
            sacc = fmaf(a0.x, b0.x, sacc); sacc = fmaf(a0.y, b0.y, sacc); sacc = fmaf(a0.z, b0.z, sacc); sacc = fmaf(a0.w, b0.w, sacc);
            sacc = fmaf(a1.x, b1.x, sacc); sacc = fmaf(a1.y, b1.y, sacc); sacc = fmaf(a1.z, b1.z, sacc); sacc = fmaf(a1.w, b1.w, sacc);
            sacc = fmaf(a2.x, b2.x, sacc); sacc = fmaf(a2.y, b2.y, sacc); sacc = fmaf(a2.z, b2.z, sacc); sacc = fmaf(a2.w, b2.w, sacc);
            sacc = fmaf(a3.x, b3.x, sacc); sacc = fmaf(a3.y, b3.y, sacc); sacc = fmaf(a3.z, b3.z, sacc); sacc = fmaf(a3.w, b3.w, sacc);
        }
        return sacc;
    };

    const int DP = (D + kGroups - 1) / kGroups;
    const int d0 = g * DP;
    const int d1 = min(D, d0 + DP);
    const float Wf = (float)W, Hf = (float)H;

    float best = -INFINITY;
    int bidx = d0 < D ? d0 : 0;
    // NHWC output: batch 4 consecutive planes into one 16-byte store (4-byte stores into 256-byte
    // pixel rows are partial-line writes: ~10x write amplification at the HBM counters)
    float ob0 = 0.f, ob1 = 0.f, ob2 = 0.f, ob3 = 0.f;
    const bool vec_ok = cost_cs > 0 && ((d0 & 3) == 0) && ((cost_cs & 3) == 0) && ((reinterpret_cast<uintptr_t>(cost) & 15) == 0);
    for (int d = d0; d < d1; ++d) {
        const float depth = pl ? pl[(size_t)d * ext.planes_sd] : s_planes[d];
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const float *hm = s_h[k];  // same address in every lane: LDS broadcast read
            const float qx = fmaf(hm[0], pxf, fmaf(hm[1], pyf, hm[2]));
            const float qy = fmaf(hm[3], pxf, fmaf(hm[4], pyf, hm[5]));
            const float qz = fmaf(hm[6], pxf, fmaf(hm[7], pyf, hm[8]));
            const float cx = fmaf(depth, qx, hm[9]);
            const float cy = fmaf(depth, qy, hm[10]);
            const float cz = fmaf(depth, qz, hm[11]);
            const float z = fmaxf(cz, 1e-5f);
            float r = __builtin_amdgcn_rcpf(z);
            r = r * fmaf(-z, r, 2.0f);  // one Newton step: <= 1 ulp
            // clamp in float BEFORE any int conversion: behind-camera points give |u| ~ 1e8
            const float sx = fminf(fmaxf(fmaf(cx, r, -0.5f), -1.0f), Wf);
            const float sy = fminf(fmaxf(fmaf(cy, r, -0.5f), -1.0f), Hf);
            const float x0f = floorf(sx), y0f = floorf(sy);
            const float fx = sx - x0f, fy = sy - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;  // in [-1, W] / [-1, H]
            const float wx0 = (x0 >= 0 && x0 < W) ? 1.0f - fx : 0.f;
            const float wx1 = (x0 + 1 < W) ? fx : 0.f;
            const float wy0 = (y0 >= 0 && y0 < H) ? 1.0f - fy : 0.f;
            const float wy1 = (y0 + 1 < H) ? fy : 0.f;
            const int xa0 = min(max(x0, 0), W - 1), xa1 = min(x0 + 1, W - 1);
            const int ya0 = min(max(y0, 0), H - 1), ya1 = min(y0 + 1, H - 1);
            const float *sb = src + (size_t)b * ext.src_bs + (size_t)k * N * kCc;
            const float t00 = tapdot(sb + (size_t)(ya0 * W + xa0) * kCc);
            const float t01 = tapdot(sb + (size_t)(ya0 * W + xa1) * kCc);
            const float t10 = tapdot(sb + (size_t)(ya1 * W + xa0) * kCc);
            const float t11 = tapdot(sb + (size_t)(ya1 * W + xa1) * kCc);
            const float top = fmaf(wx1, t01, wx0 * t00);
            const float bot = fmaf(wx1, t11, wx0 * t10);
            acc += fmaf(wy1, bot, wy0 * top);
        }
        if (cost_cs > 0) {
            const int e = (d - d0) & 3;
            ob0 = e == 0 ? acc : ob0; ob1 = e == 1 ? acc : ob1; ob2 = e == 2 ? acc : ob2; ob3 = e == 3 ? acc : ob3;
            if (live && (e == 3 || d == d1 - 1)) {
                float *o = cost + ((size_t)b * N + p) * cost_cs + (d - e);
                if (e == 3 && vec_ok) *reinterpret_cast<float4 *>(o) = make_float4(ob0, ob1, ob2, ob3);
                else { o[0] = ob0; if (e >= 1) o[1] = ob1; if (e >= 2) o[2] = ob2; if (e >= 3) o[3] = ob3; }
            }
        } else if (live) {
            cost[((size_t)b * D + d) * N + p] = acc;
        }
        if (acc > best) { best = acc; bidx = d; }
    }
    if (lowest == nullptr) return;
    s_best[g][px] = best;
    s_bidx[g][px] = bidx;
    __syncthreads();
    if (g == 0 && live) {
        float bv = s_best[0][px];
        int bi = s_bidx[0][px];
#pragma unroll
        for (int j = 1; j < kGroups; ++j) {
            const float v = s_best[j][px];
            if (v > bv) { bv = v; bi = s_bidx[j][px]; }  // strict: first maximum wins
        }
        lowest[(size_t)b * N + p] = pl ? pl[(size_t)bi * ext.planes_sd] : s_planes[bi];
    }
}

// ------------------------------------------------------------------------------------------
// Quad-coalesced variant.  The kernel above issues one 64-byte tap per lane as four dwordx4 loads; the vector
// L1 serves that pattern at ~28 B/clk/CU, but 43 B/clk/CU when the four 16-byte pieces of a tap sit in four
// ADJACENT lanes (tools/micro/gather_bw.hip).  Here a quad of lanes owns one pixel and four consecutive planes:
// lane q projects plane d+q (no redundant geometry), then for j = 0..3 the quad takes plane d+j's tap
// offsets / weights from lane j (DPP quad broadcast), each lane loads ITS 16-byte quarter of the four taps
// (one coalesced 64-byte segment per tap per quad) and accumulates a partial dot product; after the K views
// the four partial sums of each plane are added across the quad and lane 0 stores the four planes as one
// 16-byte NHWC vector.  A 256-thread workgroup = 4 pixels x 16 plane quads (D <= 64 per pass; larger D loops).
template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

__global__ __launch_bounds__(256) void cv_dot_quad_k(const float *__restrict__ cur, const float *__restrict__ src,
                                                     const float *__restrict__ src_K, const float *__restrict__ src_E,
                                                     const float *__restrict__ cur_invK, float dmin, float dmax, int B, int K,
                                                     int H, int W, int D, int tiles_per_img, int cost_cs,
                                                     float *__restrict__ cost, float *__restrict__ lowest,
                                                     float *__restrict__ planes_out, const CvExt ext) {
    __shared__ float s_planes[kMaxPlanes];
    __shared__ __attribute__((aligned(16))) float s_h[IDH_MAX_SOURCE_VIEWS][12];
    __shared__ float s_best[16][4];
    __shared__ int s_bidx[16][4];

    const int N = H * W;
    const unsigned lin = idh_xcd_remap(blockIdx.x, gridDim.x);
    const int b = lin / tiles_per_img;
    const int tile = lin - b * tiles_per_img;
    const int q = threadIdx.x & 3;           // channel quarter / plane within the quad's group of four
    const int quad = threadIdx.x >> 2;       // 0..63
    const int pxl = quad & 3;                // pixel of the tile
    const int pg = quad >> 2;                // plane-quad slot 0..15

    if (threadIdx.x < K)
        build_homography(src_K + (size_t)(b * K + threadIdx.x) * 16, src_E + (size_t)(b * K + threadIdx.x) * 16,
                         cur_invK + (size_t)b * 16, s_h[threadIdx.x]);
    for (int i = threadIdx.x; i < D; i += 256) {
        const float dp = depth_plane(i, D, dmin, dmax);
        s_planes[i] = dp;
        if (planes_out != nullptr && blockIdx.x == 0) planes_out[i] = dp;
    }
    __syncthreads();

    const int p_raw = tile * 4 + pxl;
    const bool live = p_raw < N;
    const int p = live ? p_raw : N - 1;
    const int y = p / W, x = p - y * W;
    const float pxf = (float)x + 0.5f, pyf = (float)y + 0.5f;
    const float4 cq = *reinterpret_cast<const float4 *>(cur + (size_t)b * ext.cur_bs + (size_t)p * kC + 4 * q);
    const float *pl = ext.planes ? ext.planes + (size_t)b * ext.planes_sb + (size_t)p * ext.planes_sp : nullptr;
    const float Wf = (float)W, Hf = (float)H;

    float best = -INFINITY;
    int bidx = 0;
    for (int dbase = 4 * pg; dbase < D; dbase += 64) {  // this quad's planes dbase .. dbase+3
        const int dmine = min(dbase + q, D - 1);       // the plane this lane projects
        const float depth = pl ? pl[(size_t)dmine * ext.planes_sd] : s_planes[dmine];
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        for (int k = 0; k < K; ++k) {
            const float *hm = s_h[k];
            const float qx = fmaf(hm[0], pxf, fmaf(hm[1], pyf, hm[2]));
            const float qy = fmaf(hm[3], pxf, fmaf(hm[4], pyf, hm[5]));
            const float qz = fmaf(hm[6], pxf, fmaf(hm[7], pyf, hm[8]));
            const float cx = fmaf(depth, qx, hm[9]);
            const float cy = fmaf(depth, qy, hm[10]);
            const float cz = fmaf(depth, qz, hm[11]);
            const float z = fmaxf(cz, 1e-5f);
            float r = __builtin_amdgcn_rcpf(z);
            r = r * fmaf(-z, r, 2.0f);
            const float sx = fminf(fmaxf(fmaf(cx, r, -0.5f), -1.0f), Wf);
            const float sy = fminf(fmaxf(fmaf(cy, r, -0.5f), -1.0f), Hf);
            const float x0f = floorf(sx), y0f = floorf(sy);
            const float fx = sx - x0f, fy = sy - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;
            const float wx0 = (x0 >= 0 && x0 < W) ? 1.0f - fx : 0.f;
            const float wx1 = (x0 + 1 < W) ? fx : 0.f;
            const float wy0 = (y0 >= 0 && y0 < H) ? 1.0f - fy : 0.f;
            const float wy1 = (y0 + 1 < H) ? fy : 0.f;
            const int xa0 = min(max(x0, 0), W - 1), xa1 = min(x0 + 1, W - 1);
            const int ya0 = min(max(y0, 0), H - 1), ya1 = min(y0 + 1, H - 1);
            // own sample: four tap weights and four element offsets (32-bit: B*K*N*16 < 2^31 is checked on the host)
            const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
            const int o00 = (ya0 * W + xa0) * kC, o01 = (ya0 * W + xa1) * kC, o10 = (ya1 * W + xa0) * kC, o11 = (ya1 * W + xa1) * kC;
            const float *sb = src + (size_t)b * ext.src_bs + (size_t)k * N * kC + 4 * q;
            // round j: every lane of the quad works on plane dbase + j with lane j's geometry
#define IDH_QUAD_ROUND(CTRL, ACC)                                                                                      \
    {                                                                                                                  \
        const float4 t00 = *reinterpret_cast<const float4 *>(sb + quad_bcast_i<CTRL>(o00));                            \
        const float4 t01 = *reinterpret_cast<const float4 *>(sb + quad_bcast_i<CTRL>(o01));                            \
        const float4 t10 = *reinterpret_cast<const float4 *>(sb + quad_bcast_i<CTRL>(o10));                            \
        const float4 t11 = *reinterpret_cast<const float4 *>(sb + quad_bcast_i<CTRL>(o11));                            \
        const float d00 = fmaf(cq.w, t00.w, fmaf(cq.z, t00.z, fmaf(cq.y, t00.y, cq.x * t00.x)));                       \
        const float d01 = fmaf(cq.w, t01.w, fmaf(cq.z, t01.z, fmaf(cq.y, t01.y, cq.x * t01.x)));                       \
        const float d10 = fmaf(cq.w, t10.w, fmaf(cq.z, t10.z, fmaf(cq.y, t10.y, cq.x * t10.x)));                       \
        const float d11 = fmaf(cq.w, t11.w, fmaf(cq.z, t11.z, fmaf(cq.y, t11.y, cq.x * t11.x)));                       \
        ACC += fmaf(quad_bcast<CTRL>(w11), d11, fmaf(quad_bcast<CTRL>(w10), d10, fmaf(quad_bcast<CTRL>(w01), d01, quad_bcast<CTRL>(w00) * d00))); \
    }
            IDH_QUAD_ROUND(0x00, acc0)  // quad_perm [0,0,0,0]
            IDH_QUAD_ROUND(0x55, acc1)  // [1,1,1,1]
            IDH_QUAD_ROUND(0xAA, acc2)  // [2,2,2,2]
            IDH_QUAD_ROUND(0xFF, acc3)  // [3,3,3,3]
#undef IDH_QUAD_ROUND
        }
        // add the four channel-quarter partials of every plane across the quad (xor 1, xor 2 within the quad)
#define IDH_QUAD_SUM(v)                                                                                                 \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)); /* [1,0,3,2] */      \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)); /* [2,3,0,1] */
        IDH_QUAD_SUM(acc0) IDH_QUAD_SUM(acc1) IDH_QUAD_SUM(acc2) IDH_QUAD_SUM(acc3)
#undef IDH_QUAD_SUM
        if (q == 0) {
            const float a4[4] = {acc0, acc1, acc2, acc3};
            const int nd = min(4, D - dbase);
            if (live) {
                if (cost_cs > 0) {
                    float *o = cost + ((size_t)b * N + p) * cost_cs + dbase;
                    if (nd == 4 && (cost_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(cost) & 15) == 0)
                        *reinterpret_cast<float4 *>(o) = make_float4(acc0, acc1, acc2, acc3);
                    else
                        for (int j = 0; j < nd; ++j) o[j] = a4[j];
                } else {
                    for (int j = 0; j < nd; ++j) cost[((size_t)b * D + dbase + j) * N + p] = a4[j];
                }
            }
            for (int j = 0; j < nd; ++j)
                if (a4[j] > best) { best = a4[j]; bidx = dbase + j; }
        }
    }
    if (lowest == nullptr) return;
    if (q == 0) { s_best[pg][pxl] = best; s_bidx[pg][pxl] = bidx; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int px2 = threadIdx.x;
        const int pp = tile * 4 + px2;
        if (pp < N) {
            // first maximum wins: candidates are compared in plane order (slot pg covers planes 4pg + 64m)
            float bv = -INFINITY;
            int bi = 0;
            for (int j = 0; j < 16; ++j) {
                const float v = s_best[j][px2];
                const int i = s_bidx[j][px2];
                if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
            }
            lowest[(size_t)b * N + pp] = ext.planes ? ext.planes[(size_t)b * ext.planes_sb + (size_t)pp * ext.planes_sp + (size_t)bi * ext.planes_sd]
                                                    : s_planes[bi];
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS-window variant (default when the matching map is at least 48 x 12 texels).
//
// Both kernels above fetch every bilinear tap through the vector L1 (TA/TCP): D*K*N*4 taps x 64 B = 1.6 GB per
// 96x128x64x8 frame at <= 43 B/clk/CU, and their dot products run on unpacked v_fma_f32.  Here a 256-thread workgroup
// owns a 32x8 pixel tile and works on one source view and a RUN of consecutive planes at a time: it first copies
// the source window the tile sweeps over those planes — the bounding box of the tile corners' samples at every plane
// of the run (for points in front of the camera the tile x depth-slab frustum projects into the convex hull of those
// samples) — into LDS with coalesced `global_load_lds_dwordx4` rows (no VGPRs, no ds_write), then every lane owns ONE
// pixel and reads its 4 taps x 64 B from LDS with `ds_read_b128` (256 B/clk/CU, 6x the L1 gather rate) and
// contracts them against the pixel's feature vector with v_pk_fma_f32 (two fp32 FMAs per lane per issue).  Runs are
// as long as the 48-texel window allows: planes come in super-groups of 16; a (super-group, view) pair is served by
// one window when the sweep fits (far planes / short baselines), else by two 8-plane or four 4-plane windows, else
// plane by plane.  L1 traffic drops from 256 B to 12-100 B per sample and is fully coalesced; runs whose samples
// all fall outside the image are skipped outright; whatever a window cannot cover (views behind the camera, strong
// rotation, caller-supplied per-pixel planes) is gathered from global memory lane by lane, so the result never
// depends on the window guess.
//
// LDS layout: window rows are 48 texels x 64 B plus 16 B of padding.  `ds_read_b128` is served in four fixed groups
// of 16 lanes; the lane -> pixel map below makes each group a 4x4 pixel block, whose texels (4 neighbouring columns in
// 4 neighbouring rows for the near-identity warps of an MVS tuple) then fall into 16 different 16-byte bank slots:
// (64*col + 16*row + 16*j) mod 256.  One buffer (36 KB) + the window table -> 3 workgroups per CU; the other two
// cover a workgroup's stage -> barrier latency.
constexpr int kTileW = 32, kTileH = 8;            // pixels per workgroup, one per lane
constexpr int kWW = 48, kWH = 12;                 // window, texels (3 x 16-texel DMA chunks per row)
constexpr int kRowPitch = kWW * 64 + 16;          // bytes
constexpr int kWinBytes = kWH * kRowPitch;        // 37,056
constexpr int kSG = 16;                           // planes per super-group
constexpr int kNodes = 1 + 2 + 4 + 16;            // run tree of one (super-group, view): 16 | 8 8 | 4 4 4 4 | 16 x 1
constexpr int kMaxPairs = 16;                     // (super-groups of one workgroup) x K: window + run list + tables stay below 40 KiB (4 workgroups / CU)
enum { WM_SKIP = 0, WM_WINDOW = 1, WM_GLOBAL = 2, WM_DESCEND = 3 };

struct WinEntry {  // x0 | y0 << 16 ; wneed | hneed << 8 | mode << 16
    int xy, whm;
};
struct PlaneBox {  // per (pair, plane): clamped tap bounding box of the 4 tile corners, flags
    short x0, x1, y0, y1;
    int flags;     // bit 0: all corners in front; bits 1-4: all corners left / right / above / below the image
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Sample {  // one lane's bilinear cell in one source view
    float w00, w01, w10, w11;
    int xa0, xa1, ya0, ya1;
};

// min(max(x, 0), hi) for a wave-uniform hi >= 0 as one v_med3_i32 (the compiler cannot prove 0 <= hi and emits max + min)
__device__ __forceinline__ int cv_clamp0(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

__device__ __forceinline__ Sample cv_project(float depth, float qx, float qy, float qz, float h9, float h10, float h11, float Wf, float Hf, int W, int H) {
    const float cx = fmaf(depth, qx, h9);
    const float cy = fmaf(depth, qy, h10);
    const float cz = fmaf(depth, qz, h11);
    const float z = fmaxf(cz, 1e-5f);
    float r = __builtin_amdgcn_rcpf(z);
    r = r * fmaf(-z, r, 2.0f);
    const float sx = fminf(fmaxf(fmaf(cx, r, -0.5f), -1.0f), Wf);
    const float sy = fminf(fmaxf(fmaf(cy, r, -0.5f), -1.0f), Hf);
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    // 0 <= x0 < W as one unsigned compare; clamps as v_med3 (every vector instruction here is paid 64 planes x 8 views per pixel)
    const float wx0 = (unsigned)x0 < (unsigned)W ? 1.0f - fx : 0.f;
    const float wx1 = (x0 + 1 < W) ? fx : 0.f;
    const float wy0 = (unsigned)y0 < (unsigned)H ? 1.0f - fy : 0.f;
    const float wy1 = (y0 + 1 < H) ? fy : 0.f;
    Sample s;
    s.xa0 = cv_clamp0(x0, W - 1); s.xa1 = min(x0 + 1, W - 1);
    s.ya0 = cv_clamp0(y0, H - 1); s.ya1 = min(y0 + 1, H - 1);
    s.w00 = wx0 * wy0; s.w01 = wx1 * wy0; s.w10 = wx0 * wy1; s.w11 = wx1 * wy1;
    return s;
}

// cur . tap with packed FMAs: c[i] = channels 2i, 2i+1 of the pixel's feature vector
__device__ __forceinline__ float cv_dot16(const f32x2 (&c)[8], const float4 &t0, const float4 &t1, const float4 &t2, const float4 &t3) {
    f32x2 a = c[0] * (f32x2){t0.x, t0.y};
    a = __builtin_elementwise_fma(c[1], (f32x2){t0.z, t0.w}, a);
    a = __builtin_elementwise_fma(c[2], (f32x2){t1.x, t1.y}, a);
    a = __builtin_elementwise_fma(c[3], (f32x2){t1.z, t1.w}, a);
    a = __builtin_elementwise_fma(c[4], (f32x2){t2.x, t2.y}, a);
    a = __builtin_elementwise_fma(c[5], (f32x2){t2.z, t2.w}, a);
    a = __builtin_elementwise_fma(c[6], (f32x2){t3.x, t3.y}, a);
    a = __builtin_elementwise_fma(c[7], (f32x2){t3.z, t3.w}, a);
    return a.x + a.y;
}

__device__ __forceinline__ float cv_tap_lds(const unsigned char *win, int off, const f32x2 (&c)[8]) {
#ifdef IDH_ABL_FIXEDADDR  // timing only: conflict-free lane-linear addresses instead of the gathered cell
    off = (threadIdx.x & 63) * IDH_ABL_FIXEDADDR;
    asm volatile("" : "+v"(off));
#endif
    const float4 *p = reinterpret_cast<const float4 *>(win + off);
#ifdef IDH_ABL_NOMATH  // timing only: the reads without the packed FMAs
    return p[0].x + p[1].y + p[2].z + p[3].w;
#else
    return cv_dot16(c, p[0], p[1], p[2], p[3]);
#endif
}

__device__ __forceinline__ float cv_tap_global(const float *sb, int xa, int ya, int W, const f32x2 (&c)[8]) {
    const float4 *p = reinterpret_cast<const float4 *>(sb + ((size_t)ya * W + xa) * kC);
    return cv_dot16(c, p[0], p[1], p[2], p[3]);
}

// One plane of one view for this lane's pixel from the staged window, branch-free (the four planes of a unit are then
// one basic block the scheduler can interleave: 64 ds_read_b128 against 128 packed FMAs).  Lanes whose cell is not
// inside the window read address 0 and are masked; `fb` = the cell has weight but lies outside the window -> the
// caller gathers it from global memory.
struct WinCell {
    int a00, dx, dy;
    bool ok, fb;
};
__device__ __forceinline__ WinCell cv_cell(const Sample &s, int wx0, int wy0, int wneed, int hneed) {
    const int cx0 = s.xa0 - wx0, cy0 = s.ya0 - wy0, cx1 = s.xa1 - wx0, cy1 = s.ya1 - wy0;
    const bool inside = cx0 >= 0 && cy0 >= 0 && cx1 < wneed && cy1 < hneed;
    const bool any_w = fmaxf(__builtin_fmaxf(s.w00, s.w01), __builtin_fmaxf(s.w10, s.w11)) > 0.f;  // weights are >= 0 (v_max3 + v_max)
    WinCell c;
    c.ok = inside && any_w;
    c.fb = any_w && !inside;
    c.a00 = c.ok ? cy0 * kRowPitch + cx0 * 64 : 0;
    // (a masked lane reads texels 0 / 1 of rows 0 / 1: its cell is at most one texel wide — no select needed on the steps)
    c.dx = (cx1 - cx0) * 64;
    c.dy = (cy1 - cy0) * kRowPitch;
    return c;
}
__device__ __forceinline__ float cv_sample_win(const Sample &s, const WinCell &w, const unsigned char *win, const f32x2 (&c)[8]) {
    const float d00 = cv_tap_lds(win, w.a00, c);
    const float d01 = cv_tap_lds(win, w.a00 + w.dx, c);
    const float d10 = cv_tap_lds(win, w.a00 + w.dy, c);
    const float d11 = cv_tap_lds(win, w.a00 + w.dy + w.dx, c);
    const float v = fmaf(s.w11, d11, fmaf(s.w10, d10, fmaf(s.w01, d01, s.w00 * d00)));
    return w.ok ? v : 0.f;  // select, not multiply: a masked lane may have read stale LDS bits
}
__device__ __forceinline__ float cv_sample_global(const Sample &s, const float *sb, int W, const f32x2 (&c)[8]) {
    if (s.w00 + s.w01 + s.w10 + s.w11 == 0.f) return 0.f;  // whole cell outside the image
    const float d00 = cv_tap_global(sb, s.xa0, s.ya0, W, c);
    const float d01 = cv_tap_global(sb, s.xa1, s.ya0, W, c);
    const float d10 = cv_tap_global(sb, s.xa0, s.ya1, W, c);
    const float d11 = cv_tap_global(sb, s.xa1, s.ya1, W, c);
    return fmaf(s.w11, d11, fmaf(s.w10, d10, fmaf(s.w01, d01, s.w00 * d00)));
}

struct WinArgs {
    const float *cur, *src, *src_K, *src_E, *cur_invK;
    float *cost, *lowest, *planes_out;
    float dmin, dmax;
    int B, K, H, W, D;
    int tiles_x, tiles_y, psplit, units_per_split;  // 4-plane units per workgroup: 1, 2 or a multiple of 4
    int list_bytes, planes_bytes;                   // dynamic LDS: window | run list | plane table | homographies
    float *partial;                                 // [psplit][B N][2] (best cost, plane index) of each plane group, or null
    int cost_cs;
    CvExt ext;
    const int *runs;                                // run lists left by cv_runs_k: [task][runs_stride] = {count, (xy, info) x count}, task = ((b ty) tx) sp; or null
    int runs_stride;
};

// One entry of the flattened run list: a window serving L consecutive planes of one (super-group, view) pair, or one
// plane that is gathered from global memory (views behind the camera, oversized sweeps).
struct RunEntry {
    int xy;    // window origin x0 | y0 << 16
    int info;  // wneed | hneed << 8 | mode << 16 | j0 << 18 | (L - 1) << 22 | pair << 26
};
constexpr int kRunsPerPair = 16;  // worst case: sixteen single planes
constexpr int kCntOff = 16384;  // prologue scratch inside the window: PlaneBox at 0, per-pair counts here

// The run list of one (frame, tile, plane group) task: pass 1 = per (pair, plane) the tap bounding box of the four tile corners, pass 2 = one
// thread per (super-group, view) pair builds its run tree and walks it into the flat, ordered list (count in *s_total).  Called by the
// workgroup that will consume the list (cv_dot_win_k without run-list scratch) or by cv_runs_k ahead of it.  `nthr` threads take part; the
// caller's next __syncthreads() publishes list and count.  s_h / s_planes must be visible (a barrier behind their writers).
template <typename Args>
__device__ __forceinline__ void cv_build_runs(const Args &a, const float (*s_h)[12], const float *s_planes, PlaneBox *s_pb, int *s_cnt, RunEntry *s_list,
                                              int *s_total, int b, int px0, int py0, int sg0, int nsg, int ua, int ub, int tid, int nthr) {
    const int W = a.W, H = a.H, K = a.K, D = a.D;
    const float Wf = (float)W, Hf = (float)H;
    const int npairs = nsg * K;
    // ---- window table, pass 1: per (pair, plane) the tap bounding box of the four tile corners ------------------
    const int px1 = min(px0 + kTileW, W) - 1, py1 = min(py0 + kTileH, H) - 1;  // last live pixel of the tile
    for (int it = tid; it < npairs * kSG; it += nthr) {
        const int pair = it / kSG, j = it - pair * kSG;
        const int sgi = pair / K, k = pair - sgi * K;
        const int dj = kSG * (sg0 + sgi) + j;
        PlaneBox pb;
        pb.x0 = pb.y0 = 32767; pb.x1 = pb.y1 = -1; pb.flags = 0;
        if (dj < D) {
            const float *hm = s_h[k];
            bool front = true, lft = true, rgt = true, top = true, bot = true;
            int x0m = 1 << 20, x1m = -1, y0m = 1 << 20, y1m = -1;
            for (int cidx = 0; cidx < 4; ++cidx) {
                const int cxp = (cidx & 1) ? px1 : px0, cyp = (cidx & 2) ? py1 : py0;
                const float xf = (float)cxp + 0.5f, yf = (float)cyp + 0.5f;
                const float depth = a.ext.planes ? a.ext.planes[(size_t)b * a.ext.planes_sb + (size_t)dj * a.ext.planes_sd +
                                                                 (size_t)(cyp * W + cxp) * a.ext.planes_sp]
                                                 : s_planes[dj];
                const float qx = fmaf(hm[0], xf, fmaf(hm[1], yf, hm[2]));
                const float qy = fmaf(hm[3], xf, fmaf(hm[4], yf, hm[5]));
                const float qz = fmaf(hm[6], xf, fmaf(hm[7], yf, hm[8]));
                const float cz = fmaf(depth, qz, hm[11]);
                front = front && cz > 1e-4f;
                const float r = 1.0f / fmaxf(cz, 1e-5f);
                const float rx = fmaf(depth, qx, hm[9]) * r - 0.5f, ry = fmaf(depth, qy, hm[10]) * r - 0.5f;  // unclamped sample position
                lft = lft && rx <= -1.01f; rgt = rgt && rx >= Wf + 0.01f; top = top && ry <= -1.01f; bot = bot && ry >= Hf + 0.01f;
                const int x0 = (int)floorf(fminf(fmaxf(rx, -1.0f), Wf)), y0 = (int)floorf(fminf(fmaxf(ry, -1.0f), Hf));
                // no slack: a lane whose cell rounds one texel outside the box takes the per-lane global path
                x0m = min(x0m, max(x0, 0)); x1m = max(x1m, min(x0 + 1, W - 1));
                y0m = min(y0m, max(y0, 0)); y1m = max(y1m, min(y0 + 1, H - 1));
            }
            pb.x0 = (short)x0m; pb.x1 = (short)x1m; pb.y0 = (short)y0m; pb.y1 = (short)y1m;
            pb.flags = (front ? 1 : 0) | (lft ? 2 : 0) | (rgt ? 4 : 0) | (top ? 8 : 0) | (bot ? 16 : 0) | 32;  // 32: plane exists
        }
        s_pb[it] = pb;
    }
    __syncthreads();
    // ---- pass 2: one thread per (super-group, view) pair builds its run tree 16 | 8 8 | 4 4 4 4 | 1 x 16 in registers (one
    // LDS round trip for the 16 plane boxes) and walks it into the flat, ordered run list: count, prefix, entries -------------
    const bool exact = a.ext.planes_sp == 0;  // skipping needs the convex-hull argument: image-constant planes only
    struct Box { int x0, x1, y0, y1, andf, cnt; };
    auto join = [](const Box &p, const Box &q) {
        return Box{min(p.x0, q.x0), max(p.x1, q.x1), min(p.y0, q.y0), max(p.y1, q.y1), p.andf & q.andf, p.cnt + q.cnt};
    };
    auto classify = [&](const Box &n, bool leaf) {
        WinEntry w;
        w.xy = 0;
        if (n.cnt == 0 || (exact && (n.andf & 1) && (n.andf & 30))) {
            w.whm = WM_SKIP << 16;  // no plane, or every vertex of the slab's hull is outside the image on the SAME side
        } else if ((n.andf & 1) && n.x1 - n.x0 + 1 <= kWW && n.y1 - n.y0 + 1 <= kWH) {
            const int ox = min(n.x0, W - kWW), oy = min(n.y0, H - kWH);
            w.xy = ox | (oy << 16);
            w.whm = (n.x1 - ox + 1) | ((n.y1 - oy + 1) << 8) | (WM_WINDOW << 16);
        } else {
            w.whm = (leaf ? WM_GLOBAL : WM_DESCEND) << 16;
        }
        return w;
    };
    WinEntry n16{}, n8[2]{}, n4[4]{}, n1[16]{};
    int ulo = 0, uhi = 0, my_cnt = 0;
    // the walk, twice over the same registers: emit(entry, first plane, planes)
    auto walk = [&](auto &&emit) {
        const int m16 = n16.whm >> 16;
        if (m16 == WM_SKIP) return;
        if (m16 == WM_WINDOW && ulo == 0 && uhi == 4) { emit(n16, 0, 16); return; }
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
            if (2 * h8 + 2 <= ulo || 2 * h8 >= uhi) continue;  // not this workgroup's planes
            const int m8 = n8[h8].whm >> 16;
            if (m8 == WM_SKIP) continue;
            if (m8 == WM_WINDOW && 2 * h8 >= ulo && 2 * h8 + 2 <= uhi) { emit(n8[h8], 8 * h8, 8); continue; }
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const int u = 2 * h8 + q4;
                if (u < ulo || u >= uhi) continue;
                const int m4 = n4[u].whm >> 16;
                if (m4 == WM_SKIP) continue;
                if (m4 == WM_WINDOW) { emit(n4[u], 4 * u, 4); continue; }
#pragma unroll
                for (int j = 0; j < 4; ++j)  // plane by plane: own window, or (behind the camera / oversized) global taps
                    if ((n1[4 * u + j].whm >> 16) != WM_SKIP) emit(n1[4 * u + j], 4 * u + j, 1);
            }
        }
    };
    if (tid < npairs) {
        const int sgi = tid / K;
        ulo = max(ua - 4 * (sg0 + sgi), 0); uhi = min(ub - 4 * (sg0 + sgi), 4);  // units of this super-group owned by this workgroup
        Box l4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Box l1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const PlaneBox pb = s_pb[tid * kSG + 4 * u + j];
                const bool ex = (pb.flags & 32) != 0;
                l1[j] = ex ? Box{pb.x0, pb.x1, pb.y0, pb.y1, pb.flags & 31, 1} : Box{1 << 20, -1, 1 << 20, -1, 31, 0};
                n1[4 * u + j] = classify(l1[j], true);
            }
            l4[u] = join(join(l1[0], l1[1]), join(l1[2], l1[3]));
            n4[u] = classify(l4[u], false);
        }
        Box l8[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            l8[h] = join(l4[2 * h], l4[2 * h + 1]);
            n8[h] = classify(l8[h], false);
        }
        n16 = classify(join(l8[0], l8[1]), false);
        walk([&](const WinEntry &, int, int) { ++my_cnt; });
        s_cnt[tid] = my_cnt;
    }
    __syncthreads();
    if (tid < npairs) {
        int off = 0;
        for (int i = 0; i < npairs; ++i) off += i < tid ? s_cnt[i] : 0;  // independent reads: one round trip
        if (tid == npairs - 1) *s_total = off + my_cnt;
        walk([&](const WinEntry &e, int j0, int L) {
            RunEntry r;
            r.xy = e.xy;
            r.info = (e.whm & 0x3ffff) | (j0 << 18) | ((L - 1) << 22) | (tid << 26);
            s_list[off++] = r;
        });
    }
    if (npairs == 0 && tid == 0) *s_total = 0;

}

// PLANES: caller-supplied per-pixel depth planes (idh_volume_opts.planes) instead of the uniform plane table.
// Four workgroups per CU (<= 128 VGPRs, <= 40 KiB of LDS for the bench shape): the kernel is a chain of latency-bound phases
// (prologue, window copies, barriers, LDS reads) around VALU work, and other workgroups' waves are what fills them.
template <bool PLANES>
__global__ __launch_bounds__(256, 4) void cv_dot_win_k(const WinArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_win = smem;                                                     // kWinBytes
    RunEntry *s_list = reinterpret_cast<RunEntry *>(smem + kWinBytes);               // [pairs * kRunsPerPair]
    int *s_cnt = reinterpret_cast<int *>(smem + kCntOff);                            // [pairs + 1], prologue only
    float *s_planes = reinterpret_cast<float *>(smem + kWinBytes + a.list_bytes);    // [D]
    float (*s_h)[12] = reinterpret_cast<float (*)[12]>(smem + kWinBytes + a.list_bytes + a.planes_bytes);  // [K][12]
    __shared__ int s_total;

    const int N = a.H * a.W, W = a.W, H = a.H, K = a.K, D = a.D;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef IDH_ABL_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long tr_wait = 0, tr_comp = 0, tr_prolog = 0;
#endif
    // Task order = dispatch order: each XCD (blocks b % 8) walks one contiguous range of tiles, the plane groups of a tile
    // back to back (they share its source maps in that XCD's L2).  A plane group's work grows with its depth — near planes have
    // the larger parallax, most of their samples leave the source images and are skipped: 50 k ... 235 k cycles per workgroup
    // from the nearest to the farthest quarter of 64 planes — and the dispatcher hands consecutive workgroups of an XCD to its
    // 32 CUs in turn, so with 2, 4 or 8 plane groups a CU would only ever see ONE group (a quarter of the CUs gets all the far
    // planes).  Rotating the group order by one every 32 workgroups gives every CU every group in turn.
    unsigned lin = idh_xcd_remap(blockIdx.x, gridDim.x);
    int sp = lin % a.psplit;
    lin /= a.psplit;
#ifndef IDH_ABL_DOT_NOROTATE
    sp = (int)((sp + lin * a.psplit / 32) % a.psplit);
#endif
#ifdef IDH_ABL_DOT_FARFIRST_CHUNKED  // all far-plane groups of a frame's tiles first: loses the interleave (D = 96: +12 %)
    {
        const unsigned ntiles = (unsigned)(a.B * a.tiles_x * a.tiles_y);
        if ((ntiles & 7) == 0) {
            const unsigned per = ntiles >> 3, slot = blockIdx.x >> 3;
            const unsigned frame = (unsigned)(a.tiles_x * a.tiles_y);
            const unsigned chunk = per % frame == 0 ? frame : per;
            const unsigned c = slot / (chunk * a.psplit), within = slot - c * chunk * a.psplit;
            sp = a.psplit - 1 - (int)(within / chunk);
            lin = (blockIdx.x & 7) * per + c * chunk + within % chunk;
        }
    }
#endif
    int tx = lin % a.tiles_x; lin /= a.tiles_x;
#ifndef IDH_ABL_DOT_NOROTATE_TX
    // the same for the tile columns (a tile's work depends on where it lies in the image: here 70 k ... 205 k cycles from the left to
    // the right column): the column order of a tile row advances by one every 32 tiles, i.e. once per full turn of the plane groups
    tx = (int)((tx + lin * a.tiles_x / 32) % a.tiles_x);
#endif
    const int ty = lin % a.tiles_y;
    const int b = lin / a.tiles_y;
    const int nunits_all = (D + 3) >> 2;
    const int ua = sp * a.units_per_split, ub = min(ua + a.units_per_split, nunits_all);  // this workgroup's 4-plane units
    const int sg0 = ua >> 2;
    const int nsg = ((ub + 3) >> 2) - sg0;
    const int px0 = tx * kTileW, py0 = ty * kTileH;
    const float Wf = (float)W, Hf = (float)H;

    if (tid < K)
        build_homography(a.src_K + (size_t)(b * K + tid) * 16, a.src_E + (size_t)(b * K + tid) * 16, a.cur_invK + (size_t)b * 16, s_h[tid]);
    for (int i = tid; i < D; i += 256) {
        const float dp = depth_plane(i, D, a.dmin, a.dmax);
        s_planes[i] = dp;
        if (a.planes_out != nullptr && blockIdx.x == 0) a.planes_out[i] = dp;
    }
    __syncthreads();

    // ---- run list of this (tile, plane group): built here, or (a.runs: idh_volume_opts.scratch large enough) read from the list cv_runs_k left ----
    if (a.runs != nullptr) {
        const int *rl = a.runs + (size_t)((((size_t)b * a.tiles_y + ty) * a.tiles_x + tx) * a.psplit + sp) * a.runs_stride;
        const int cnt = rl[0];
        if (tid == 0) s_total = cnt;
        if (tid < cnt) s_list[tid] = RunEntry{rl[1 + 2 * tid], rl[2 + 2 * tid]};
    } else {
        cv_build_runs(a, s_h, s_planes, reinterpret_cast<PlaneBox *>(s_win), s_cnt, s_list, &s_total, b, px0, py0, sg0, nsg, ua, ub, tid, 256);
    }

    // ---- this lane's pixel: ds_read_b128 lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) -> 4x4 pixel blocks ----
    const int m5 = lane & 31, hi = lane >> 5;
    const bool g1 = (m5 >= 4 && m5 < 12) || (m5 >= 16 && m5 < 20) || m5 >= 28;
    const int rank = g1 ? (m5 < 12 ? m5 - 4 : (m5 < 20 ? m5 - 8 : m5 - 16)) : (m5 < 4 ? m5 : (m5 < 16 ? m5 - 8 : m5 - 12));
    const int lx = 16 * (wave & 1) + 4 * (2 * hi + (g1 ? 1 : 0)) + (rank & 3), ly = 4 * (wave >> 1) + (rank >> 2);
    const int x_raw = px0 + lx, y_raw = py0 + ly;
    const bool live = x_raw < W && y_raw < H;
    const int x = min(x_raw, W - 1), y = min(y_raw, H - 1);
    const int p = y * W + x;
    const float pxf = (float)x + 0.5f, pyf = (float)y + 0.5f;
    f32x2 c[8];
    {
        const float4 *cp = reinterpret_cast<const float4 *>(a.cur + (size_t)b * a.ext.cur_bs + (size_t)p * kC);
        const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
        c[0] = (f32x2){c0.x, c0.y}; c[1] = (f32x2){c0.z, c0.w}; c[2] = (f32x2){c1.x, c1.y}; c[3] = (f32x2){c1.z, c1.w};
        c[4] = (f32x2){c2.x, c2.y}; c[5] = (f32x2){c2.z, c2.w}; c[6] = (f32x2){c3.x, c3.y}; c[7] = (f32x2){c3.z, c3.w};
    }
    const float *pl = PLANES ? a.ext.planes + (size_t)b * a.ext.planes_sb + (size_t)p * a.ext.planes_sp : nullptr;
    // staging: a window row is 3 chunks of 16 texels (1 KiB = one global_load_lds_dwordx4 of a wave); the 36 chunks
    // of a window go round-robin to the 4 waves (chunk i = wave + 4m -> row i / 3, columns 16 (i % 3) ..), lane l
    // copies 16-byte quarter (l & 3) of texel column (l >> 2) of its chunk
    const int slane_col = lane >> 2, squart = lane & 3;
    __attribute__((address_space(3))) unsigned char *lds_win = (__attribute__((address_space(3))) unsigned char *)s_win;
    constexpr int kChunksPerRow = kWW / 16, kChunks = kWH * kChunksPerRow;

    auto stage = [&](const RunEntry &e, int buf) {
#ifndef IDH_ABL_NOSTAGE
        const int wx0 = e.xy & 0xffff, wy0 = e.xy >> 16;
        const int wneed = e.info & 0xff, hneed = (e.info >> 8) & 0xff;
        const int k = (e.info >> 26) % K;
        const float *g = a.src + (size_t)b * a.ext.src_bs + (size_t)k * N * kC + ((size_t)wy0 * W + wx0 + slane_col) * kC + 4 * squart;
#pragma unroll
        for (int m = 0; m < (kChunks + 3) / 4; ++m) {
            const int ch = wave + 4 * m;                       // wave-uniform
            const int row = ch / kChunksPerRow, cc = ch - row * kChunksPerRow;
            if (ch < kChunks && row < hneed && 16 * cc < wneed) {
                if (16 * cc + slane_col < wneed)
                    __builtin_amdgcn_global_load_lds(g + ((size_t)row * W + 16 * cc) * kC, lds_win + buf * kWinBytes + row * kRowPitch + cc * 1024, 16, 0, 0);
            }
        }
#endif
    };
    auto entry = [&](int r) {  // wave-uniform copy of list entry r
        const RunEntry e = s_list[r];
        return RunEntry{__builtin_amdgcn_readfirstlane(e.xy), __builtin_amdgcn_readfirstlane(e.info)};
    };

    __syncthreads();  // run list visible, prologue scratch (aliasing the windows) dead
#ifdef IDH_ABL_TRACE
    tr_prolog = __builtin_amdgcn_s_memtime() - tr_t0;
#endif
#ifdef IDH_ABL_NORUNS
    const int nruns = 0;
#else
    const int nruns = __builtin_amdgcn_readfirstlane(s_total);
#endif
    float best = -INFINITY;
    int bidx = 0;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    // a super-group's 16 planes are complete when the list moves on to the next one: store them, fold them into the arg-max
    auto flush = [&](int sgi) {
        const int dbase = kSG * (sg0 + sgi);
        const int ulo = max(ua - 4 * (sg0 + sgi), 0), uhi = min(ub - 4 * (sg0 + sgi), 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d0 = dbase + 4 * u, n4 = min(4, D - d0);
            if (n4 > 0 && u >= ulo && u < uhi) {
                const float a4[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
                if (live) {
                    if (a.cost_cs > 0) {
                        float *o = a.cost + ((size_t)b * N + p) * a.cost_cs + d0;
                        if (n4 == 4 && (a.cost_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.cost) & 15) == 0) *reinterpret_cast<float4 *>(o) = acc[u];
                        else
                            for (int j = 0; j < n4; ++j) o[j] = a4[j];
                    } else {
                        for (int j = 0; j < n4; ++j) a.cost[((size_t)b * D + d0 + j) * N + p] = a4[j];
                    }
                }
                for (int j = 0; j < n4; ++j)
                    if (a4[j] > best) { best = a4[j]; bidx = d0 + j; }
            }
            acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // acc[u] += r with a run-time u: 16 selects + 16 adds keep acc in registers and the unit loop one basic block (a scalar branch on the
    // wave-uniform u with 4 adds per arm is 28 vector instructions shorter and 8 % slower: 0.613 vs 0.566 ms)
    auto add_unit = [&](int u, const float4 &r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool m = (u == i);
            acc[i].x += m ? r.x : 0.f; acc[i].y += m ? r.y : 0.f; acc[i].z += m ? r.z : 0.f; acc[i].w += m ? r.w : 0.f;
        }
    };

    int sg_done = 0;     // super-groups [0, sg_done) have been flushed
#ifdef IDH_ABL_COUNT
    int dbg_planes = 0, dbg_runs = 0;
#endif
    int cur_pair = -1;
    float qx = 0.f, qy = 0.f, qz = 0.f, h9 = 0.f, h10 = 0.f, h11 = 0.f, dlane = 0.f;
    const float *sb = a.src;
    int dbase = 0;
    RunEntry e = nruns > 0 ? entry(0) : RunEntry{0, 0};
    for (int r = 0; r < nruns; ++r) {
        const RunEntry en = r + 1 < nruns ? entry(r + 1) : RunEntry{0, 0};  // read one run ahead: off the critical path
        const int pair = e.info >> 26 & 63, mode = (e.info >> 16) & 3, j0 = (e.info >> 18) & 15, L = ((e.info >> 22) & 15) + 1;
#ifdef IDH_ABL_COUNT
        dbg_planes += L; ++dbg_runs;
#endif
        if (pair != cur_pair) {  // wave-uniform
            cur_pair = pair;
            const int sgi = pair / K, k = pair - sgi * K;
            while (sg_done < sgi) flush(sg_done++);
            dbase = kSG * (sg0 + sgi);
            dlane = s_planes[min(dbase + (lane & 15), D - 1)];  // lane j holds plane j of the super-group: depth_of = v_readlane
            const float *hm = s_h[k];
            qx = fmaf(hm[0], pxf, fmaf(hm[1], pyf, hm[2]));
            qy = fmaf(hm[3], pxf, fmaf(hm[4], pyf, hm[5]));
            qz = fmaf(hm[6], pxf, fmaf(hm[7], pyf, hm[8]));
            h9 = hm[9]; h10 = hm[10]; h11 = hm[11];
            sb = a.src + (size_t)b * a.ext.src_bs + (size_t)k * N * kC;
        }
        const int nd = min(kSG, D - dbase);
        auto depth_of = [&](int j) {  // plane j of this super-group (clamped to the last real plane)
            if constexpr (PLANES) return pl[(size_t)min(dbase + j, D - 1) * a.ext.planes_sd];
            else return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlane), j));
        };
        if (mode == WM_GLOBAL) {
            const Sample sm = cv_project(depth_of(j0), qx, qy, qz, h9, h10, h11, Wf, Hf, W, H);
            const float v = cv_sample_global(sm, sb, W, c);
            const int jj = j0 & 3;
            add_unit(j0 >> 2, make_float4(jj == 0 ? v : 0.f, jj == 1 ? v : 0.f, jj == 2 ? v : 0.f, jj == 3 ? v : 0.f));
            e = en;
            continue;
        }
#ifdef IDH_ABL_TRACE
        const unsigned long long tr_a = __builtin_amdgcn_s_memtime();
#endif
#ifndef IDH_ABL_NOBARRIER
        __syncthreads();  // every wave is done with the previous window
#endif
        stage(e, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef IDH_ABL_NOBARRIER
        __syncthreads();
#endif
        const unsigned char *win = s_win;
#ifdef IDH_ABL_TRACE
        const unsigned long long tr_b = __builtin_amdgcn_s_memtime();
        tr_wait += tr_b - tr_a;
#endif
#ifndef IDH_ABL_NOCOMPUTE
        const int wx0 = e.xy & 0xffff, wy0 = e.xy >> 16, wneed = e.info & 0xff, hneed = (e.info >> 8) & 0xff;
        // one plane of this view from the window; `fb`: the cell has weight but lies outside the window -> global gather (rare)
        auto plane = [&](int j) {
#ifdef IDH_ABL_NOPREP  // timing only: one projection per run
            Sample sm = cv_project(depth_of(j0), qx, qy, qz, h9, h10, h11, Wf, Hf, W, H);
            asm volatile("" : "+v"(sm.w00), "+v"(sm.xa0) : "s"(j));
            const WinCell wc = cv_cell(sm, wx0, wy0, wneed, hneed);
#else
            const Sample sm = cv_project(depth_of(j), qx, qy, qz, h9, h10, h11, Wf, Hf, W, H);
            const WinCell wc = cv_cell(sm, wx0, wy0, wneed, hneed);
#endif
#ifdef IDH_ABL_NODOT  // timing only
            float v = sm.w00 + sm.w01 + sm.w10 + sm.w11 + __builtin_bit_cast(float, wc.a00 + wc.dx + wc.dy);
            v = wc.ok ? v : 0.f;
#else
            float v = cv_sample_win(sm, wc, win, c);
#endif
            if (__builtin_amdgcn_ballot_w64(wc.fb) != 0) {
#ifdef IDH_ABL_DOT_EAGER_FALLBACK
                if (wc.fb) v = cv_sample_global(sm, sb, W, c);
#else
                // rare: re-project behind an opaque copy of the plane index so that none of the fallback's 64-bit address
                // arithmetic is hoisted into the common path
                int jo = j;
                asm volatile("" : "+s"(jo));
                if (wc.fb) v = cv_sample_global(cv_project(depth_of(jo), qx, qy, qz, h9, h10, h11, Wf, Hf, W, H), sb, W, c);
#endif
            }
            return v;
        };
        if (L == 1) {
            const float v = plane(j0);
            const int jj = j0 & 3;
            add_unit(j0 >> 2, make_float4(jj == 0 ? v : 0.f, jj == 1 ? v : 0.f, jj == 2 ? v : 0.f, jj == 3 ? v : 0.f));
        } else {
            const int u0 = j0 >> 2, uend = min(u0 + (L >> 2), (nd + 3) >> 2);  // units with at least one real plane
#ifdef IDH_DOT_DYNAMIC_UNITS  // (rounds 3-5: one loop body, the unit's accumulator picked with 16 selects + 16 adds)
#pragma unroll 1
            for (int u = u0; u < uend; ++u) {
                float4 rr;
                rr.x = plane(4 * u); rr.y = plane(4 * u + 1); rr.z = plane(4 * u + 2); rr.w = plane(4 * u + 3);
                add_unit(u, rr);
            }
#else
            // Four copies of the unit body behind wave-uniform guards: the unit index is a compile-time constant inside each copy, so its
            // results go to acc[u] with 4 adds (the dynamic index cost 32 vector operations per unit = 8 of ~102 per plane) and the plane
            // depths are v_readlane with an immediate lane.  Each copy is still ONE basic block of four planes (a branch INSIDE the unit
            // body - round 3's variant of this idea - split it and lost 8 %).
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= u0 && u < uend) {
                    float4 rr;
                    rr.x = plane(4 * u); rr.y = plane(4 * u + 1); rr.z = plane(4 * u + 2); rr.w = plane(4 * u + 3);
                    acc[u].x += rr.x; acc[u].y += rr.y; acc[u].z += rr.z; acc[u].w += rr.w;
                }
            }
#endif
        }
#endif
#ifdef IDH_ABL_TRACE
        tr_comp += __builtin_amdgcn_s_memtime() - tr_b;
#endif
        e = en;
    }
    while (sg_done < nsg) flush(sg_done++);
#ifdef IDH_ABL_TRACE
    if (a.lowest != nullptr && lane < 4) {
        const unsigned long long tot = __builtin_amdgcn_s_memtime() - tr_t0;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned cu = ((xcc & 15) << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15);
        const unsigned long long vals[4] = {tot, (tr_t0 >> 6) & 0xffffff, (unsigned long long)cu, tr_comp};
        a.lowest[(size_t)b * N + (((ty * a.tiles_x + tx) * a.psplit + sp) * 4 + wave) * 4 + lane] = (float)vals[lane];
    }
    return;
#endif
#ifdef IDH_ABL_COUNT
    if (a.lowest != nullptr && live) a.lowest[(size_t)b * N + p] = (float)(dbg_planes + 1000 * dbg_runs);
    return;
#endif
    if (a.lowest != nullptr && a.psplit == 1 && live) a.lowest[(size_t)b * N + p] = PLANES ? pl[(size_t)bidx * a.ext.planes_sd] : s_planes[bidx];
    // planes split over workgroups: this group's (best cost, plane) per pixel for the combine pass (cv_argmax_partials_k) — 8 bytes
    // instead of the pass re-reading the group's 4 D / psplit bytes of the volume
    if (a.partial != nullptr && a.psplit > 1 && live)
        *reinterpret_cast<float2 *>(a.partial + (((size_t)sp * a.B + b) * N + p) * 2) = make_float2(best, __builtin_bit_cast(float, bidx));
}

// Run lists ahead of the volume kernel (round 6).  cv_dot_win_k's workgroups spent ~10 % of their life building the list of their own (tile,
// plane group) - three barrier-separated phases in front of the first window copy, on 8 of 256 threads in the last one.  One 64-thread workgroup
// per task builds the same list here (the same code: cv_build_runs) and leaves it in the caller's scratch; the volume kernel then starts with one
// coalesced read.  6144 tasks for 32 frames: a few microseconds for the whole batch.
__global__ __launch_bounds__(64) void cv_runs_k(const WinArgs a, int *__restrict__ runs_out) {
    __shared__ PlaneBox s_pb[kMaxPairs * kSG];
    __shared__ int s_cnt[kMaxPairs + 1];
    __shared__ RunEntry s_list[kMaxPairs * kRunsPerPair];
    __shared__ float s_planes[kMaxPlanes];
    __shared__ float s_h[IDH_MAX_SOURCE_VIEWS][12];
    __shared__ int s_total;
    const int tid = threadIdx.x, K = a.K, D = a.D;
    unsigned lin = blockIdx.x;
    const int sp = lin % a.psplit; lin /= a.psplit;
    const int tx = lin % a.tiles_x; lin /= a.tiles_x;
    const int ty = lin % a.tiles_y;
    const int b = lin / a.tiles_y;
    const int nunits_all = (D + 3) >> 2;
    const int ua = sp * a.units_per_split, ub = min(ua + a.units_per_split, nunits_all);
    const int sg0 = ua >> 2, nsg = ((ub + 3) >> 2) - sg0;
    if (tid < K)
        build_homography(a.src_K + (size_t)(b * K + tid) * 16, a.src_E + (size_t)(b * K + tid) * 16, a.cur_invK + (size_t)b * 16, s_h[tid]);
    for (int i = tid; i < D; i += 64) s_planes[i] = depth_plane(i, D, a.dmin, a.dmax);
    __syncthreads();
    cv_build_runs(a, s_h, s_planes, s_pb, s_cnt, s_list, &s_total, b, tx * kTileW, ty * kTileH, sg0, nsg, ua, ub, tid, 64);
    __syncthreads();
    int *o = runs_out + (size_t)blockIdx.x * a.runs_stride;
    const int cnt = s_total;
    if (tid == 0) o[0] = cnt;
    for (int i = tid; i < cnt; i += 64) { o[1 + 2 * i] = s_list[i].xy; o[2 + 2 * i] = s_list[i].info; }
}

// lowest[b,p] = plane_{argmax_d cost[b,d,p]} (first maximum wins) for launches that split the planes over workgroups
__global__ __launch_bounds__(256) void cv_argmax_k(const float *__restrict__ cost, int cost_cs, int B, int N, int D, float dmin, float dmax,
                                                   float *__restrict__ lowest, const CvExt ext) {
    const long long total = (long long)B * N;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const long long b = t / N, p = t - b * N;
        float best = -INFINITY;
        int bi = 0;
        if (cost_cs > 0 && (cost_cs & 3) == 0 && (D & 3) == 0 && (reinterpret_cast<uintptr_t>(cost) & 15) == 0) {
            for (int d = 0; d < D; d += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(cost + t * cost_cs + d);
                if (v.x > best) { best = v.x; bi = d; }
                if (v.y > best) { best = v.y; bi = d + 1; }
                if (v.z > best) { best = v.z; bi = d + 2; }
                if (v.w > best) { best = v.w; bi = d + 3; }
            }
        } else {
            for (int d = 0; d < D; ++d) {
                const float v = cost_cs > 0 ? cost[t * cost_cs + d] : cost[(b * D + d) * N + p];
                if (v > best) { best = v; bi = d; }
            }
        }
        lowest[t] = ext.planes ? ext.planes[b * ext.planes_sb + (long long)bi * ext.planes_sd + p * ext.planes_sp] : depth_plane(bi, D, dmin, dmax);
    }
}

// the same from the plane groups' partial results (ascending groups, strict >: the first maximum still wins)
__global__ __launch_bounds__(256) void cv_argmax_partials_k(const float *__restrict__ partial, int psplit, int B, int N, int D, float dmin, float dmax,
                                                            float *__restrict__ lowest, const CvExt ext) {
    const long long total = (long long)B * N;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        const long long b = t / N, p = t - b * N;
        float best = -INFINITY;
        int bi = 0;
        for (int g = 0; g < psplit; ++g) {
            const float2 v = *reinterpret_cast<const float2 *>(partial + ((size_t)g * total + t) * 2);
            if (v.x > best) { best = v.x; bi = __builtin_bit_cast(int, v.y); }
        }
        lowest[t] = ext.planes ? ext.planes[b * ext.planes_sb + (long long)bi * ext.planes_sd + p * ext.planes_sp] : depth_plane(bi, D, dmin, dmax);
    }
}

}  // namespace

// plane split of the window kernel: keep the per-workgroup table inside kMaxPairs and give small batches enough
// workgroups for 256 CUs x 3 (the arg-max then runs as a second, tiny kernel)
static void cv_win_split(int B, int K, int H, int W, int D, int *psplit, int *units_per_split) {
    const int nunits = (D + 3) / 4;
    const int nsg = (nunits + 3) / 4;
    const long long tiles = (long long)B * idh_cdiv(W, kTileW) * idh_cdiv(H, kTileH);
    // whole super-groups per workgroup (longest runs) unless that leaves CUs idle: then 8- or 4-plane slices
    int per = 4 * nsg;  // units per workgroup
#ifndef IDH_DOT_SPLIT_MIN
#define IDH_DOT_SPLIT_MIN 6144  // >= 6 rounds of 256 CUs x 4 workgroups: a tile's work varies 8x with its position (fewer, longer workgroups leave CUs idle at the tail)
#endif
    while (per > 4 && (tiles * idh_cdiv(nunits, per) < IDH_DOT_SPLIT_MIN || (long long)(per / 4) * K > kMaxPairs)) per = 4 * idh_cdiv(per / 4, 2);
    // small batches: 8- and 4-plane slices until the grid has two rounds of 256 CUs x 4 workgroups (measured, 96x128 map, K = 8, D = 64:
    // B = 8: 29.2 -> 27.2 us/frame with 8-plane slices, B = 4: 39.9 -> 36.4 and B = 2: 72.8 -> 49.3 with 4-plane slices)
    if (per == 4 && tiles * idh_cdiv(nunits, per) < 2048) per = 2;
    if (per == 2 && tiles * idh_cdiv(nunits, per) < 2048) per = 1;
#ifdef IDH_ABL_DOT_PER
    per = IDH_ABL_DOT_PER;
#endif
    *units_per_split = per;
    *psplit = idh_cdiv(nunits, per);
}

// 0 = automatic; otherwise the caller's choice when that kernel covers the shape (else -1)
static int cv_pick_kernel(int forced, int B, int K, int H, int W, int D, int C = kC) {
    if (C != kC) return (forced == 0 || forced == IDH_CV_KERNEL_LANE) ? IDH_CV_KERNEL_LANE : -1;  // wider texels: the lane kernel only
    const bool quad_ok = (long long)K * H * W * kC < (1ll << 31) && K > 0;  // 32-bit tap offsets within one (b,k) image
    const bool win_ok = quad_ok && W >= kWW && H >= kWH && W < 32768 && H < 32768 && K <= kMaxPairs;  // PlaneBox / WinEntry pack coordinates in 15 / 16 bits
    (void)D;
    // measured (tools/perf_dot.py, 96x128 map, K=8, D=64): window 18.6 us/frame at B=32, 27 at B=8, 36 at B=4, 49 at B=2, 72 at B=1;
    // quad 51 / 57 / 57 / 58 / 58 -> a single frame (48 tiles) cannot fill 256 CUs and stays on the quad kernel
    const bool win_pays = (long long)B * idh_cdiv(W, kTileW) * idh_cdiv(H, kTileH) >= 96;
    switch (forced) {
        case 0: return (win_ok && win_pays) ? IDH_CV_KERNEL_WINDOW : (quad_ok ? IDH_CV_KERNEL_QUAD : IDH_CV_KERNEL_LANE);
        case IDH_CV_KERNEL_LANE: return IDH_CV_KERNEL_LANE;
        case IDH_CV_KERNEL_QUAD: return quad_ok ? IDH_CV_KERNEL_QUAD : -1;
        case IDH_CV_KERNEL_WINDOW: return win_ok ? IDH_CV_KERNEL_WINDOW : -1;
        default: return -1;
    }
}

static int cv_resolve_ext(const idh_volume_opts *o, int K, int H, int W, CvExt *e, int C = kC) {
    const long long N = (long long)H * W;
    e->planes = nullptr; e->planes_sb = e->planes_sd = 0; e->planes_sp = 0;
    e->cur_bs = N * C; e->src_bs = (long long)K * N * C;
    if (!o) return IDH_OK;
    if (o->cur_batch_stride) e->cur_bs = o->cur_batch_stride;
    if (o->src_batch_stride) e->src_bs = o->src_batch_stride;
    if (e->cur_bs < N * C || e->src_bs < (long long)K * N * C || (e->cur_bs & 3) || (e->src_bs & 3)) return IDH_EINVAL;
    if (o->planes) {
        if (o->planes_pixel_stride != 0 && o->planes_pixel_stride != 1) return IDH_EINVAL;
        e->planes = o->planes; e->planes_sb = o->planes_batch_stride; e->planes_sd = o->planes_plane_stride;
        e->planes_sp = o->planes_pixel_stride;
    }
    return IDH_OK;
}

extern "C" int idh_cost_volume_dot_ex_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                                          const float *src_E_44, const float *cur_invK_44, float dmin,
                                          float dmax, int B, int K, int C, int H, int W, int D,
                                          float *cost, int cost_nhwc_cs, float *lowest_bhw, float *planes_d,
                                          const idh_volume_opts *opts, void *stream) {
    const bool own_planes = opts && opts->planes;
    if (B < 0 || K < 0 || H <= 0 || W <= 0 || D <= 0) return IDH_EINVAL;
    if (!own_planes && (!(dmin > 0.f) || !(dmax > 0.f))) return IDH_EINVAL;
    if ((C != kC && C != 2 * kC && C != 4 * kC) || D > kMaxPlanes || K > IDH_MAX_SOURCE_VIEWS) return IDH_EUNSUPPORTED;
    if (B == 0) return IDH_OK;
    if (cost_nhwc_cs != 0 && cost_nhwc_cs < D) return IDH_EINVAL;
    if (!cur_nhwc || !cost || !cur_invK_44 || (K > 0 && (!src_nhwc || !src_K_44 || !src_E_44)))
        return IDH_EINVAL;
    CvExt ext;
    if (int rc = cv_resolve_ext(opts, K, H, W, &ext, C)) return rc;
    if (own_planes) { planes_d = nullptr; dmin = dmax = 1.f; }
    const int which = cv_pick_kernel(opts ? opts->kernel : 0, B, K, H, W, D, C);
    if (which < 0) return IDH_EINVAL;
    if (which == IDH_CV_KERNEL_WINDOW) {
        WinArgs a{};
        a.cur = cur_nhwc; a.src = src_nhwc; a.src_K = src_K_44; a.src_E = src_E_44; a.cur_invK = cur_invK_44;
        a.cost = cost; a.lowest = lowest_bhw; a.planes_out = planes_d; a.dmin = dmin; a.dmax = dmax;
        a.B = B; a.K = K; a.H = H; a.W = W; a.D = D; a.cost_cs = cost_nhwc_cs; a.ext = ext;
        a.tiles_x = idh_cdiv(W, kTileW); a.tiles_y = idh_cdiv(H, kTileH);
        cv_win_split(B, K, H, W, D, &a.psplit, &a.units_per_split);
        // optional scratch for the arg-max over split planes (idh_volume_opts.scratch: >= idh_cost_volume_dot_scratch_floats)
        a.partial = (opts && opts->struct_size >= (int64_t)(offsetof(idh_volume_opts, struct_size) + sizeof(int64_t)) && opts->scratch && lowest_bhw && a.psplit > 1 &&
                     opts->scratch_floats >= 2ll * a.psplit * B * H * W) ? opts->scratch : nullptr;
        a.list_bytes = ((a.units_per_split + 3) / 4) * K * kRunsPerPair * (int)sizeof(RunEntry);
        // ... and, behind the arg-max partials, for the run lists of every (frame, tile, plane group) task, built by cv_runs_k ahead of the volume kernel
        {
            const bool scratch_ok = opts && opts->struct_size >= (int64_t)(offsetof(idh_volume_opts, struct_size) + sizeof(int64_t)) && opts->scratch;
            const long long part = a.psplit > 1 ? 2ll * a.psplit * B * H * W : 0;
            const long long tasks = (long long)B * a.tiles_x * a.tiles_y * a.psplit;
            a.runs_stride = 1 + 2 * (a.list_bytes / (int)sizeof(RunEntry));
            a.runs = (scratch_ok && tasks < (1ll << 31) && opts->scratch_floats >= part + tasks * a.runs_stride) ? reinterpret_cast<const int *>(opts->scratch + part) : nullptr;
#ifdef IDH_DOT_NO_PREPASS
            a.runs = nullptr;
#endif
        }
        a.planes_bytes = ((D + 3) & ~3) * (int)sizeof(float);
        const size_t lds = (size_t)kWinBytes + a.list_bytes + a.planes_bytes + (size_t)K * 12 * sizeof(float);
        const long long blocks = (long long)B * a.tiles_x * a.tiles_y * a.psplit;
        if (blocks >= (1ll << 31)) return IDH_EUNSUPPORTED;
        static IdhDeviceOnce attr_set;
        if (attr_set.first()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(cv_dot_win_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(cv_dot_win_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
                return IDH_ELAUNCH;
            attr_set.mark();
        }
        if (a.runs) {
            hipLaunchKernelGGL(cv_runs_k, dim3((unsigned)blocks), dim3(64), 0, idh_stream(stream), a, const_cast<int *>(a.runs));
            IDH_CHECK_LAUNCH();
        }
        if (ext.planes) hipLaunchKernelGGL(cv_dot_win_k<true>, dim3((unsigned)blocks), dim3(256), lds, idh_stream(stream), a);
        else hipLaunchKernelGGL(cv_dot_win_k<false>, dim3((unsigned)blocks), dim3(256), lds, idh_stream(stream), a);
        IDH_CHECK_LAUNCH();
#if defined(IDH_ABL_TRACE) || defined(IDH_ABL_COUNT)
        if (false) {
#else
        if (a.psplit > 1 && lowest_bhw) {
#endif
            int g2 = idh_cdiv((long long)B * H * W, 256);
            if (g2 > 8192) g2 = 8192;
            if (a.partial)
                hipLaunchKernelGGL(cv_argmax_partials_k, dim3(g2), dim3(256), 0, idh_stream(stream), a.partial, a.psplit, B, H * W, D, dmin, dmax, lowest_bhw, ext);
            else
                hipLaunchKernelGGL(cv_argmax_k, dim3(g2), dim3(256), 0, idh_stream(stream), cost, cost_nhwc_cs, B, H * W, D, dmin, dmax, lowest_bhw, ext);
            IDH_CHECK_LAUNCH();
        }
        return IDH_OK;
    }
    if (which == IDH_CV_KERNEL_QUAD) {  // 1.4-1.7x faster than the one-lane-per-tap kernel at every batch size measured
        const int tiles4 = idh_cdiv((long long)H * W, 4);
        hipLaunchKernelGGL(cv_dot_quad_k, dim3((unsigned)(B * tiles4)), dim3(256), 0, idh_stream(stream), cur_nhwc, src_nhwc, src_K_44,
                           src_E_44, cur_invK_44, dmin, dmax, B, K, H, W, D, tiles4, cost_nhwc_cs, cost, lowest_bhw, planes_d, ext);
        IDH_CHECK_LAUNCH();
        return IDH_OK;
    }
    const int tiles = idh_cdiv((long long)H * W, kTilePx);
#define IDH_LANE(CQ_)                                                                                                         \
    hipLaunchKernelGGL(cv_dot_k<CQ_>, dim3((unsigned)(B * tiles)), dim3(256), 0, idh_stream(stream), cur_nhwc, src_nhwc, src_K_44, \
                       src_E_44, cur_invK_44, dmin, dmax, B, K, H, W, D, tiles, cost_nhwc_cs, cost, lowest_bhw, planes_d, ext)
    if (C == kC) IDH_LANE(1);
    else if (C == 2 * kC) IDH_LANE(2);
    else IDH_LANE(4);
#undef IDH_LANE
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

extern "C" long long idh_cost_volume_dot_scratch_floats(int B, int K, int C, int H, int W, int D) {
    if (B <= 0 || K < 0 || H <= 0 || W <= 0 || D <= 0 || cv_pick_kernel(0, B, K, H, W, D, C) != IDH_CV_KERNEL_WINDOW) return 0;
    int psplit = 1, per = 0;
    cv_win_split(B, K, H, W, D, &psplit, &per);
    // arg-max partials of the split planes + the run lists cv_runs_k builds ahead of the volume kernel (either part is optional: a shorter scratch
    // only forgoes what does not fit)
    const long long tasks = (long long)B * idh_cdiv(W, kTileW) * idh_cdiv(H, kTileH) * psplit;
    const long long stride = 1 + 2ll * ((per + 3) / 4) * K * kRunsPerPair;
    return (psplit > 1 ? 2ll * psplit * B * H * W : 0) + tasks * stride;
}

extern "C" const char *idh_cost_volume_dot_kernel_name(int B, int K, int H, int W, int D) {
    const int which = cv_pick_kernel(0, B, K, H, W, D);
    return which == IDH_CV_KERNEL_WINDOW ? "cv_dot_win_k<false>" : (which == IDH_CV_KERNEL_QUAD ? "cv_dot_quad_k" : "cv_dot_k");
}

extern "C" int idh_cost_volume_dot_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                                       const float *src_E_44, const float *cur_invK_44, float dmin,
                                       float dmax, int B, int K, int C, int H, int W, int D,
                                       float *cost, int cost_nhwc_cs, float *lowest_bhw, float *planes_d,
                                       void *stream) {
    return idh_cost_volume_dot_ex_fwd(cur_nhwc, src_nhwc, src_K_44, src_E_44, cur_invK_44, dmin, dmax, B, K, C, H, W, D, cost,
                                      cost_nhwc_cs, lowest_bhw, planes_d, nullptr, stream);
}
