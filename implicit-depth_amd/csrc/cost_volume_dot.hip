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

__global__ __launch_bounds__(256) void cv_dot_k(const float *__restrict__ cur,   // B,N,16
                                                const float *__restrict__ src,   // B,K,N,16
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

    const float4 *cp = reinterpret_cast<const float4 *>(cur + (size_t)b * ext.cur_bs + (size_t)p * kC);
    const float *pl = ext.planes ? ext.planes + (size_t)b * ext.planes_sb + (size_t)p * ext.planes_sp : nullptr;
    const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];

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
            const float *sb = src + (size_t)b * ext.src_bs + (size_t)k * N * kC;
            const float t00 = dot16(c0, c1, c2, c3, reinterpret_cast<const float4 *>(sb + (size_t)(ya0 * W + xa0) * kC));
            const float t01 = dot16(c0, c1, c2, c3, reinterpret_cast<const float4 *>(sb + (size_t)(ya0 * W + xa1) * kC));
            const float t10 = dot16(c0, c1, c2, c3, reinterpret_cast<const float4 *>(sb + (size_t)(ya1 * W + xa0) * kC));
            const float t11 = dot16(c0, c1, c2, c3, reinterpret_cast<const float4 *>(sb + (size_t)(ya1 * W + xa1) * kC));
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

}  // namespace

static int cv_resolve_ext(const idh_volume_opts *o, int K, int H, int W, CvExt *e) {
    const long long N = (long long)H * W;
    e->planes = nullptr; e->planes_sb = e->planes_sd = 0; e->planes_sp = 0;
    e->cur_bs = N * kC; e->src_bs = (long long)K * N * kC;
    if (!o) return IDH_OK;
    if (o->cur_batch_stride) e->cur_bs = o->cur_batch_stride;
    if (o->src_batch_stride) e->src_bs = o->src_batch_stride;
    if (e->cur_bs < N * kC || e->src_bs < (long long)K * N * kC || (e->cur_bs & 3) || (e->src_bs & 3)) return IDH_EINVAL;
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
    if (C != kC || D > kMaxPlanes || K > IDH_MAX_SOURCE_VIEWS) return IDH_EUNSUPPORTED;
    if (B == 0) return IDH_OK;
    if (cost_nhwc_cs != 0 && cost_nhwc_cs < D) return IDH_EINVAL;
    if (!cur_nhwc || !cost || !cur_invK_44 || (K > 0 && (!src_nhwc || !src_K_44 || !src_E_44)))
        return IDH_EINVAL;
    CvExt ext;
    if (int rc = cv_resolve_ext(opts, K, H, W, &ext)) return rc;
    if (own_planes) { planes_d = nullptr; dmin = dmax = 1.f; }
    const bool quad_ok = (long long)K * H * W * kC < (1ll << 31) && K > 0;  // 32-bit tap offsets within one (b,k) image
    if (quad_ok) {  // default: 1.4-1.7x faster than the one-lane-per-tap kernel at every batch size measured
        const int tiles4 = idh_cdiv((long long)H * W, 4);
        hipLaunchKernelGGL(cv_dot_quad_k, dim3((unsigned)(B * tiles4)), dim3(256), 0, idh_stream(stream), cur_nhwc, src_nhwc, src_K_44,
                           src_E_44, cur_invK_44, dmin, dmax, B, K, H, W, D, tiles4, cost_nhwc_cs, cost, lowest_bhw, planes_d, ext);
        IDH_CHECK_LAUNCH();
        return IDH_OK;
    }
    const int tiles = idh_cdiv((long long)H * W, kTilePx);
    hipLaunchKernelGGL(cv_dot_k, dim3((unsigned)(B * tiles)), dim3(256), 0, idh_stream(stream), cur_nhwc,
                       src_nhwc, src_K_44, src_E_44, cur_invK_44, dmin, dmax, B, K, H, W, D, tiles, cost_nhwc_cs,
                       cost, lowest_bhw, planes_d, ext);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

extern "C" const char *idh_cost_volume_dot_kernel_name(int B, int K, int H, int W, int D) {
    (void)B; (void)D;
    const bool quad_ok = (long long)K * H * W * kC < (1ll << 31) && K > 0;
    return quad_ok ? "cv_dot_quad_k" : "cv_dot_k";
}

extern "C" int idh_cost_volume_dot_fwd(const float *cur_nhwc, const float *src_nhwc, const float *src_K_44,
                                       const float *src_E_44, const float *cur_invK_44, float dmin,
                                       float dmax, int B, int K, int C, int H, int W, int D,
                                       float *cost, int cost_nhwc_cs, float *lowest_bhw, float *planes_d,
                                       void *stream) {
    return idh_cost_volume_dot_ex_fwd(cur_nhwc, src_nhwc, src_K_44, src_E_44, cur_invK_44, dmin, dmax, B, K, C, H, W, D, cost,
                                      cost_nhwc_cs, lowest_bhw, planes_d, nullptr, stream);
}
