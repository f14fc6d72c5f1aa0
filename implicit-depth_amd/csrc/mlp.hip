// Per-pixel occlusion MLP over all query-depth planes in one launch (gfx950, fp32 MFMA).
//
// Replaces the reference's per-plane loop  bd_model.py:293-304 -> run_mlp_val :412-442 ->
// BinaryMLPNetwork (modules/networks.py:98-115):
//     for each rendered-depth plane p:  cat([depth_p, feature_s0(64), (prior_p)]) -> permute ->
//         Linear(65|66,128) -> ELU -> Linear(128,128) -> ELU -> Linear(128,1) -> permute -> cat
// i.e. P passes over a (B,192,256,66) tensor that is materialised P times.
//
// Here:  pre1 = W1[:,feat] . feat + b1 is plane independent and is computed ONCE per pixel;
// per plane only the rank-1 terms  w_depth*depth_p (+ w_prior*prior_p)  are added before the ELU.
// Everything is computed TRANSPOSED (channels x pixels) so that the C/D register layout of one
// layer's MFMA result is exactly the B-operand layout of the next layer:
//     out^T[n, m] = W[n, k] . act^T[k, m]       A operand = weights, B operand = activations
// With v_mfma_f32_16x16x4_f32, lane (col = lane&15, q = lane>>4) holds rows 4q..4q+3 of each 16-row
// sub-tile in its 4 accumulator registers, and as a B operand for k-block c it must supply
// k = 16c + 4q + kk for MFMA kk = 0..3 (K order is free as long as A and B agree) — the same
// registers.  So activations never leave the register file between layers: no LDS transpose.
// Weights are pre-packed in "fragment order" so every A-fragment load is one coalesced 1 KiB
// wave read (idh_pack_mlp_weight).
#include <type_traits>

#include "idh_common.h"
#include "split_f16.h"

namespace {

using namespace idh_f16;

constexpr int kHidden = 128;           // mlp_size (networks.py:88)
constexpr int kNS = kHidden / 16;      // 8 sub-tiles of 16 hidden units

// nn.ELU(alpha = 1): x > 0 ? x : exp(x) - 1, formed exactly as torch's kernel forms it (exp, then subtract), with the
// exponential through v_exp_f32 (1 ulp; exp(x) = exp2(x * log2 e): relative error < 1e-6 for the |x| < 10 that occur)
// instead of ocml's expm1f (~25 VALU per activation against 5 here).  fp32 MFMA and VALU serialise on a SIMD (DESIGN
// 4.3), and the 64 activations per pixel and plane were ~1000 vector instructions against 256 MFMAs: 4.63 -> 4.23 ms
// with a polynomial near 0, -> this form.
__device__ __forceinline__ float elu1(float x) {
#ifdef IDH_ELU_OCML
    return x > 0.f ? x : expm1f(x);
#else
    return x > 0.f ? x : __expf(x) - 1.0f;
#endif
}
// ELU of the split-precision kernel: exp via v_exp_f32 (2 ulp of a value near 1 -> |err| ~1e-7
// absolute, the size of one fp32 rounding of the O(1) sums around it) instead of ocml expm1f (~25 VALU)
__device__ __forceinline__ float elu1_fast(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }
__device__ __forceinline__ float lrelu(float x, float s) { return x >= 0.f ? x : x * s; }

// One 128 -> 128 layer on register-resident activations (transposed form), TM pixel sub-tiles.
//   hin[c][t]  : B operands, c = k-block (8), t = pixel sub-tile
//   wfrag      : packed weights [c][i][lane][4]   (i = output sub-tile)
//   acc[i][t]  : results in C/D layout (= next layer's B operands)
template <int TM>
__device__ __forceinline__ void dense128(const f32x4 (&hin)[kNS][TM], const f32x4 *wfrag, int lane,
                                         f32x4 (&acc)[kNS][TM]) {
#pragma unroll
    for (int c = 0; c < kNS; ++c) {
#pragma unroll
        for (int i = 0; i < kNS; ++i) {
            const f32x4 A = wfrag[(c * kNS + i) * 64 + lane];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], hin[c][t][kk], acc[i][t], 0, 0, 0);
        }
        // keep the scheduler from hoisting all 64 weight-fragment loads (256 VGPRs) to the top
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct BinArgs {
    const float *feat;    // NHWC rows, M x cs
    const float *depth;   // B,P,HW
    const float *prior;   // B,P,HW or null
    const float *w1f;     // packed [Cf/16][8][64][4]
    const float *w2;      // packed [8][8][64][4]
    const float *vecs;    // 6 x 128: b1, w_depth, w_prior, b2, w3, (b3 at [5*128])
    float *out;           // B,P,HW
    int M, HW, P, cs, Cf;
    int has_prior;        // W1 has a prior column
    float prior_const;    // used when has_prior && prior == null
    // per-pixel binary depth search (bd_model.py:273-292): when search_iters > 0 the P planes are
    // replaced by search_iters dependent evaluations at the pixel's current search depth
    int search_iters;
    float search_lo, search_hi, thr_logit;
    float *search_out;    // B,1,HW final search depths
    const float *thr_bins;    // per-depth Thresholder (binary_metrics_utils.py:42-52): n_thr ascending bin edges ...
    const float *thr_logits;  // ... and logit(threshold) per bin; null -> the constant thr_logit
    int n_thr;
    const float *sw2;     // F16 kernel: per-row scale of the f16-packed W2 (w2 then points to idh_pack_mlp_weight_f16 output)
    int feat_unaligned;   // 1: feature rows are not 16-byte aligned (cs % 4 != 0 or an odd base: the reference's [depth | feat | prior] rows
                          // of 65 / 66 floats, networks.py:106-115, read in place): dword loads instead of one dwordx4.
                          // 2: fully strided features, element (b, pix, c) at feat[b * feat_bs + pix * feat_ps + c * feat_chs] - the permuted view of
                          // an NCHW tensor that run_mlp_val hands to the network (bd_model.py:415-439: cat along dim 1, then permute(0, 2, 3, 1))
    long long feat_bs;
    int feat_ps, feat_chs;
};

// Persistent 512- / 768-thread workgroups (one per CU): W2 (64 KiB) and, when it fits, the feature part
// of W1 (8 KiB per 16 input channels) are staged once into LDS in MFMA fragment order and shared
// by the 8 waves; each wave then streams 16-pixel tiles.
// 8 waves (2 per SIMD) for the split-precision variant (183 VGPRs); 12 waves (3 per SIMD) for the fp32 kernel, whose 142
// VGPRs allow it: a third wave per SIMD gives the MFMA pipe something to do while the other two evaluate ELUs
// (measured: see DESIGN.md 4.3).
constexpr int bin_threads(bool f16) { return f16 ? 512 : 768; }
constexpr int kW1LdsMaxBlocks = 4;  // Cf <= 64 -> W1f in LDS (32 KiB); wider scales read it via L1/L2

// F16 = true: layer 2 (the per-plane 128x128 GEMM, ~95 % of the flops) runs in "f16x3" split precision
// on v_mfma_f32_16x16x32_f16 (csrc/split_f16.h): W2 in LDS as two f16 pieces (same 64 KiB), the hidden
// vector of each pixel scaled by its own power of two, 96 MFMAs per plane instead of 256 fp32 ones.
template <int TM, bool F16>
__global__ __launch_bounds__(bin_threads(F16)) void binary_mlp_k(const BinArgs a) {
    constexpr int kBinThreads = bin_threads(F16), kBinWaves = kBinThreads / 64;
    static_assert(!F16 || TM == 1, "split-precision path is written for one pixel sub-tile per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *sW2 = reinterpret_cast<f32x4 *>(smem_raw);
    const int cblocks = (a.Cf + 15) >> 4;
    const bool w1_lds = cblocks <= kW1LdsMaxBlocks;
    f32x4 *sW1 = sW2 + kNS * kNS * 64;
    float *s_vec = reinterpret_cast<float *>(sW1 + (w1_lds ? cblocks * kNS * 64 : 0));
    {
        const f32x4 *g2 = reinterpret_cast<const f32x4 *>(a.w2), *g1 = reinterpret_cast<const f32x4 *>(a.w1f);
        for (int i = threadIdx.x; i < kNS * kNS * 64; i += kBinThreads) sW2[i] = g2[i];
        if (w1_lds)
            for (int i = threadIdx.x; i < cblocks * kNS * 64; i += kBinThreads) sW1[i] = g1[i];
        for (int i = threadIdx.x; i < 6 * kHidden; i += kBinThreads) s_vec[i] = a.vecs[i];
    }
    __syncthreads();
    const float *s_b1 = s_vec, *s_wd = s_vec + kHidden, *s_wp = s_vec + 2 * kHidden, *s_b2 = s_vec + 3 * kHidden,
                *s_w3 = s_vec + 4 * kHidden;
    const float b3 = s_vec[5 * kHidden];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, q = lane >> 4;
    const int tiles = (a.M + 16 * TM - 1) / (16 * TM);

    for (int tile = blockIdx.x * kBinWaves + wave; tile < tiles; tile += gridDim.x * kBinWaves) {
        const int m0 = tile * 16 * TM;
        // ---- layer 1, plane-independent part: pre1^T = W1f . feat^T + b1 ------------------
        f32x4 pre1[kNS][TM];
#pragma unroll
        for (int i = 0; i < kNS; ++i) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(s_b1 + 16 * i + 4 * q);
#pragma unroll
            for (int t = 0; t < TM; ++t) pre1[i][t] = bv;
        }
        int mrow[TM];
        bool mok[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int m = m0 + 16 * t + ln;
            mok[t] = m < a.M;
            mrow[t] = mok[t] ? m : a.M - 1;
        }
#pragma unroll 1
        for (int c = 0; c < cblocks; ++c) {
            f32x4 Bf[TM];
            const bool cok = (16 * c + 4 * q) < a.Cf;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                Bf[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (cok) {
                    if (a.feat_unaligned == 2) {  // planar / strided features: 16 consecutive pixels of a channel are one 64-byte run
                        const int b = mrow[t] / a.HW;
                        const float *fp = a.feat + (size_t)b * a.feat_bs + (size_t)(mrow[t] - b * a.HW) * a.feat_ps + (size_t)(16 * c + 4 * q) * a.feat_chs;
                        Bf[t] = (f32x4){fp[0], fp[(size_t)a.feat_chs], fp[2 * (size_t)a.feat_chs], fp[3 * (size_t)a.feat_chs]};
                    } else {
                        const float *fp = a.feat + (size_t)mrow[t] * a.cs + 16 * c + 4 * q;
                        if (a.feat_unaligned) Bf[t] = (f32x4){fp[0], fp[1], fp[2], fp[3]};
                        else Bf[t] = *reinterpret_cast<const f32x4 *>(fp);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
                const f32x4 A = w1_lds ? sW1[(c * kNS + i) * 64 + lane]
                                       : *reinterpret_cast<const f32x4 *>(a.w1f + ((size_t)(c * kNS + i) * 64 + lane) * 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int t = 0; t < TM; ++t)
                        pre1[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], Bf[t][kk], pre1[i][t], 0, 0, 0);
            }
        }
        // per-lane pixel bookkeeping for the depth / prior / output planes (pixel = column ln)
        size_t poff[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int b = mrow[t] / a.HW;
            const int pix = mrow[t] - b * a.HW;
            poff[t] = (size_t)b * a.P * a.HW + pix;
        }
        // ---- per query plane ----------------------------------------------------------------
        const bool search = a.search_iters > 0;
        const int n_eval = search ? a.search_iters : a.P;
        float lo[TM], hi[TM], sd[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) { lo[t] = a.search_lo; hi[t] = a.search_hi; sd[t] = (a.search_hi - a.search_lo) * 0.5f; }
#pragma unroll 1
        for (int p = 0; p < n_eval; ++p) {
            float dv[TM], pv[TM];
            const int pp = search ? 0 : p;  // the search reads plane 0 of the prior (P = 1 there)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                dv[t] = search ? sd[t] : a.depth[poff[t] + (size_t)p * a.HW];
                pv[t] = a.has_prior ? (a.prior ? a.prior[poff[t] + (size_t)pp * a.HW] : a.prior_const) : 0.f;
            }
            f32x4 h1[kNS][TM], acc[kNS][TM];
            // layer-1 pre-activation = per-pixel part + w_depth * d (+ w_prior * p): the prior term only when the network has
            // a prior column (uniform branch: a vector instruction saved is MFMA time saved)
            auto layer1 = [&](auto with_prior) {
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const f32x4 wd = *reinterpret_cast<const f32x4 *>(s_wd + 16 * i + 4 * q);
                    f32x4 wp = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (decltype(with_prior)::value) wp = *reinterpret_cast<const f32x4 *>(s_wp + 16 * i + 4 * q);
                    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(s_b2 + 16 * i + 4 * q);
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = fmaf(wd[r], dv[t], pre1[i][t][r]);
                            if (decltype(with_prior)::value) v = fmaf(wp[r], pv[t], v);
                            h1[i][t][r] = F16 ? elu1_fast(v) : elu1(v);
                        }
                        acc[i][t] = b2;
                    }
                }
            };
            if (a.has_prior) layer1(std::true_type{});
            else layer1(std::false_type{});
            if constexpr (F16) {
                f32x4 hv[kNS], a2[kNS];
#pragma unroll
                for (int i = 0; i < kNS; ++i) { hv[i] = h1[i][0]; a2[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                const int ex = column_exponent<kNS>(hv);
                const float mul = exp2_int(14 - ex), sx = exp2_int(ex - 14);
                const u32x4 *w2 = reinterpret_cast<const u32x4 *>(sW2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    u32x4 Bh, Bl;
                    split_block(hv[2 * c], hv[2 * c + 1], mul, Bh, Bl);
#pragma unroll
                    for (int i4 = 0; i4 < kNS; i4 += 4) {
                        u32x4 Ah[4], Al[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            Ah[i] = w2[((c * kNS + i4 + i) * 2 + 0) * 64 + lane];
                            Al[i] = w2[((c * kNS + i4 + i) * 2 + 1) * 64 + lane];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) a2[i4 + i] = mfma_f16(Al[i], Bh, a2[i4 + i]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) a2[i4 + i] = mfma_f16(Ah[i], Bl, a2[i4 + i]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) a2[i4 + i] = mfma_f16(Ah[i], Bh, a2[i4 + i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const f32x4 sw = *reinterpret_cast<const f32x4 *>(a.sw2 + 16 * i + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][0][r] = fmaf(a2[i][r], sx * sw[r], acc[i][0][r]);
                }
            } else {
                // wave priority: the MFMA block at 0, the activations either side of it at 2 (3.66 -> 3.60 ms at 32 frames x 8 planes; the
                // other way round 3.62; profiles/r05/experiments.md - the larger effect of the same idea is in fv_mlp_k)
                __builtin_amdgcn_s_setprio(0);
                dense128<TM>(h1, sW2, lane, acc);
                __builtin_amdgcn_s_setprio(2);
            }
            // ---- layer 3: logit = w3 . ELU(h2) + b3; reduce over the 4 lane quarters ----------
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
                    const f32x4 w3 = *reinterpret_cast<const f32x4 *>(s_w3 + 16 * i + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s = fmaf(w3[r], F16 ? elu1_fast(acc[i][t][r]) : elu1(acc[i][t][r]), s);
                }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const float logit = s + b3;
                if (!search) {
                    if (q == 0 && mok[t]) a.out[poff[t] + (size_t)p * a.HW] = logit;
                } else {
                    // sigmoid(logit) < threshold  <=>  logit < logit(threshold): "visible" -> move the far bound
                    float thr = a.thr_logit;
                    if (a.n_thr > 0) {  // torch.bucketize(depth, bins): number of edges strictly below the query depth
                        int idx = 0;
                        for (int e = 0; e < a.n_thr; ++e) idx += a.thr_bins[e] < sd[t] ? 1 : 0;
                        thr = a.thr_logits[idx < a.n_thr ? idx : a.n_thr - 1];
                    }
                    if (logit < thr) hi[t] = sd[t]; else lo[t] = sd[t];
                    if (p == n_eval - 1 && q == 0 && mok[t]) a.out[poff[t]] = logit;  // pred_0 of the last evaluation
                    sd[t] = (hi[t] + lo[t]) * 0.5f;
                }
            }
        }
        if (search) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
                if (q == 0 && mok[t]) a.search_out[poff[t]] = sd[t];
        }
    }
}

// ---- f16x3 weight packing: rows scaled by 2^(14 - e_row), two f16 pieces, 32-wide K blocks ----
// dst: [ceil(n_in/32)][8 n-subtiles][piece 2][lane 64][8 halves], then 128 floats 2^(e_row - 14)
__global__ __launch_bounds__(256) void pack_mlp_weight_f16_k(const float *__restrict__ w, u32x4 *__restrict__ dst, int ld, int col0,
                                                             int n_in, int nb32) {
    __shared__ float s_mul[kHidden];
    float *scale_out = reinterpret_cast<float *>(dst + (size_t)nb32 * kNS * 2 * 64);
    if (threadIdx.x < kHidden) {
        const int n = threadIdx.x;
        float m = 0.f;
        for (int k = 0; k < n_in; ++k) {
            const float v = w[(size_t)n * ld + col0 + k];
            m = fmaxf(m, fabsf(v));
            if ((__float_as_uint(v) & 0x7F800000u) == 0x7F800000u) m = __uint_as_float(0x7F800000u);
        }
        const int e = exponent_of(__float_as_uint(m));
        s_mul[n] = exp2_int(14 - e);
        if (blockIdx.x == 0) scale_out[n] = exp2_int(e - 14 < -126 ? -126 : e - 14);
    }
    __syncthreads();
    const int total = nb32 * kNS * 64;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int lane = t & 63, i = (t >> 6) % kNS, c = (t >> 6) / kNS;
        const int n = 16 * i + (lane & 15), q = lane >> 4;
        f32x4 x0, x1;
        for (int e = 0; e < 4; ++e) {
            const int k0 = 16 * (2 * c) + 4 * q + e, k1 = 16 * (2 * c + 1) + 4 * q + e;
            x0[e] = k0 < n_in ? w[(size_t)n * ld + col0 + k0] : 0.f;
            x1[e] = k1 < n_in ? w[(size_t)n * ld + col0 + k1] : 0.f;
        }
        u32x4 hi, lo;
        split_block(x0, x1, s_mul[n], hi, lo);
        dst[((size_t)(c * kNS + i) * 2 + 0) * 64 + lane] = hi;
        dst[((size_t)(c * kNS + i) * 2 + 1) * 64 + lane] = lo;
    }
}

// W (n_out=128, n_in) row-major, input columns [col0, col0+n_in_used) -> fragment order
// dst[c][i][lane][4] = W[16i + (lane&15)][col0 + 16c + 4(lane>>4) + e], zero padded in k.
__global__ __launch_bounds__(256) void pack_mlp_weight_k(const float *__restrict__ w, float *__restrict__ dst, int ld,
                                                         int col0, int n_in, int cblocks) {
    const int total = cblocks * kNS * 64 * 4;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int e = t & 3, lane = (t >> 2) & 63, i = (t >> 8) % kNS, c = (t >> 8) / kNS;
        const int n = 16 * i + (lane & 15), k = 16 * c + 4 * (lane >> 4) + e;
        dst[t] = (k < n_in) ? w[(size_t)n * ld + col0 + k] : 0.f;
    }
}

// ---- temporal prior: warp the previous frame's occlusion prediction into the current view ----
// reference experiment_modules/bd_model.py:395-410 (BackprojectDepth -> Project3D ->
// grid_sample(mode="nearest", zeros, align_corners=False) -> invalid := -1)
__global__ __launch_bounds__(256) void sample_prior_k(const float *__restrict__ depth, const float *__restrict__ prior,
                                                      int Q, const float *__restrict__ cur_world_T_cam,
                                                      const float *__restrict__ prior_cam_T_world,
                                                      const float *__restrict__ Kmat, const float *__restrict__ invK, int P,
                                                      int H, int W, float *__restrict__ out) {
    __shared__ float sP[12], sI[9];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        const float *A = prior_cam_T_world + (size_t)b * 16, *Bm = cur_world_T_cam + (size_t)b * 16, *Kb = Kmat + (size_t)b * 16;
        float T[4][4], Pm[3][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
                for (int m = 0; m < 4; ++m) s = fmaf(A[i * 4 + m], Bm[m * 4 + j], s);
                T[i][j] = s;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
                for (int m = 0; m < 4; ++m) s = fmaf(Kb[i * 4 + m], T[m][j], s);
                Pm[i][j] = s;
            }
        for (int i = 0; i < 12; ++i) sP[i] = Pm[i / 4][i % 4];
        for (int i = 0; i < 9; ++i) sI[i] = invK[(size_t)b * 16 + (i / 3) * 4 + (i % 3)];
    }
    __syncthreads();
    const int N = H * W;
    const float Wf = (float)W, Hf = (float)H;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < P * N; t += gridDim.x * 256) {
        const int p = t / N, pix = t - p * N;
        const int y = pix / W, x = pix - y * W;
        const float pxf = (float)x + 0.5f, pyf = (float)y + 0.5f;
        const float d = depth[((size_t)b * P + p) * N + pix];
        const float X0 = d * fmaf(sI[0], pxf, fmaf(sI[1], pyf, sI[2]));
        const float X1 = d * fmaf(sI[3], pxf, fmaf(sI[4], pyf, sI[5]));
        const float X2 = d * fmaf(sI[6], pxf, fmaf(sI[7], pyf, sI[8]));
        const float cx = fmaf(sP[0], X0, fmaf(sP[1], X1, fmaf(sP[2], X2, sP[3])));
        const float cy = fmaf(sP[4], X0, fmaf(sP[5], X1, fmaf(sP[6], X2, sP[7])));
        const float cz = fmaf(sP[8], X0, fmaf(sP[9], X1, fmaf(sP[10], X2, sP[11])));
        const float z = fmaxf(cz, 1e-5f);
        const float u = cx / z, v = cy / z;
        // normalise exactly as the reference does, then grid_sample's un-normalisation + nearbyint
        const float gx = (u / Wf - 0.5f) * 2.f, gy = (v / Hf - 0.5f) * 2.f;
        const float sx = ((gx + 1.f) * Wf - 1.f) * 0.5f, sy = ((gy + 1.f) * Hf - 1.f) * 0.5f;
        const float xr = rintf(sx), yr = rintf(sy);  // round-half-even, like std::nearbyint
        float val = 0.f;
        if (xr >= 0.f && xr <= Wf - 1.f && yr >= 0.f && yr <= Hf - 1.f) {
            const int q = p < Q ? p : Q - 1;
            val = prior[((size_t)b * Q + q) * N + (int)yr * W + (int)xr];
        }
        // (cam z > 0) is always true after the clamp — same quirk as the cost-volume mask
        out[((size_t)b * P + p) * N + pix] = (d > 0.f && z > 0.f) ? val : -1.f;
    }
}

}  // namespace

extern "C" int idh_sample_prior_fwd(const float *rendered_depth_bphw, const float *prior_pred_bqhw, int Q,
                                    const float *cur_world_T_cam_44, const float *prior_cam_T_world_44, const float *K_44,
                                    const float *invK_44, int B, int P, int H, int W, float *out_bphw, void *stream) {
    if (B < 0 || P <= 0 || Q <= 0 || H <= 0 || W <= 0) return IDH_EINVAL;
    if (B == 0) return IDH_OK;
    if (!rendered_depth_bphw || !prior_pred_bqhw || !cur_world_T_cam_44 || !prior_cam_T_world_44 || !K_44 || !invK_44 || !out_bphw ||
        B > 65535)
        return IDH_EINVAL;
    int gx = idh_cdiv((long long)P * H * W, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(sample_prior_k, dim3(gx, B), dim3(256), 0, idh_stream(stream), rendered_depth_bphw, prior_pred_bqhw, Q,
                       cur_world_T_cam_44, prior_cam_T_world_44, K_44, invK_44, P, H, W, out_bphw);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

extern "C" size_t idh_packed_mlp_weight_floats(int n_in) {
    return n_in <= 0 ? 0 : (size_t)((n_in + 15) / 16) * kNS * 64 * 4;
}

// Repack columns [col0, col0+n_in) of a (128, ld) row-major Linear weight into MFMA A-fragment order.
extern "C" int idh_pack_mlp_weight(const float *w, float *dst, int ld, int col0, int n_in, void *stream) {
    if (!w || !dst || ld <= 0 || col0 < 0 || n_in <= 0 || col0 + n_in > ld) return IDH_EINVAL;
    const int cblocks = (n_in + 15) / 16;
    hipLaunchKernelGGL(pack_mlp_weight_k, dim3(idh_cdiv(cblocks * kNS * 256, 256)), dim3(256), 0, idh_stream(stream), w,
                       dst, ld, col0, n_in, cblocks);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

extern "C" size_t idh_packed_mlp_weight_f16_bytes(int n_in) {
    if (n_in <= 0) return 0;
    return (size_t)((n_in + 31) / 32) * kNS * 2 * 64 * 16 + kHidden * sizeof(float);
}

extern "C" int idh_pack_mlp_weight_f16(const float *w_row_major, void *dst, int ld, int col0, int n_in, void *stream) {
    if (!w_row_major || !dst || n_in <= 0 || ld < col0 + n_in || col0 < 0) return IDH_EINVAL;
    const int nb32 = (n_in + 31) / 32;
    hipLaunchKernelGGL(pack_mlp_weight_f16_k, dim3(idh_cdiv(nb32 * kNS * 64, 256)), dim3(256), 0, idh_stream(stream), w_row_major,
                       reinterpret_cast<u32x4 *>(dst), ld, col0, n_in, nb32);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

static int binary_mlp_launch(BinArgs a, int B, void *stream, bool f16 = false);

extern "C" int idh_binary_mlp_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *depth_bphw,
                                  const float *prior_bphw, int has_prior, float prior_const, const float *w1f_packed,
                                  const float *w2_packed, const float *vecs6x128, int B, int P, int HW,
                                  float *out_bphw, void *stream) {
    if (B < 0 || P < 0 || HW <= 0 || Cf <= 0 || (Cf & 3) || feat_cs < Cf) return IDH_EINVAL;
    if (B == 0 || P == 0) return IDH_OK;
    if (!feat_nhwc || !depth_bphw || !w1f_packed || !w2_packed || !vecs6x128 || !out_bphw) return IDH_EINVAL;
    if (reinterpret_cast<uintptr_t>(feat_nhwc) & 3) return IDH_EINVAL;
    const long long M = (long long)B * HW;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    BinArgs a{feat_nhwc, depth_bphw, prior_bphw, w1f_packed, w2_packed, vecs6x128, out_bphw,
              (int)M, HW, P, feat_cs, Cf, has_prior, prior_const, 0, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, 0, nullptr, 0};
    // any row stride / any 4-byte-aligned base (ABI 105): rows that are not 16-byte aligned are read with dword loads
    a.feat_unaligned = ((feat_cs & 3) || (reinterpret_cast<uintptr_t>(feat_nhwc) & 15)) ? 1 : 0;
    return binary_mlp_launch(a, B, stream);
}

extern "C" int idh_binary_mlp_strided_fwd(const float *feat, long long feat_batch_stride, int feat_pixel_stride, int feat_channel_stride, int Cf,
                                          const float *depth_bphw, const float *prior_bphw, int has_prior, float prior_const,
                                          const float *w1f_packed, const float *w2_packed, const float *vecs6x128, int B, int P, int HW,
                                          float *out_bphw, void *stream) {
    if (B < 0 || P < 0 || HW <= 0 || Cf <= 0 || (Cf & 3) || feat_batch_stride < 0 || feat_pixel_stride <= 0 || feat_channel_stride <= 0) return IDH_EINVAL;
    if (B == 0 || P == 0) return IDH_OK;
    if (!feat || !depth_bphw || !w1f_packed || !w2_packed || !vecs6x128 || !out_bphw || (reinterpret_cast<uintptr_t>(feat) & 3)) return IDH_EINVAL;
    const long long M = (long long)B * HW;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    BinArgs a{feat, depth_bphw, prior_bphw, w1f_packed, w2_packed, vecs6x128, out_bphw,
              (int)M, HW, P, 0, Cf, has_prior, prior_const, 0, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, 0, nullptr, 2,
              feat_batch_stride, feat_pixel_stride, feat_channel_stride};
    return binary_mlp_launch(a, B, stream);
}

extern "C" int idh_binary_mlp_f16x3_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *depth_bphw,
                                        const float *prior_bphw, int has_prior, float prior_const, const float *w1f_packed,
                                        const void *w2_f16, const float *vecs6x128, int B, int P, int HW,
                                        float *out_bphw, void *stream) {
    if (B < 0 || P < 0 || HW <= 0 || Cf <= 0 || (Cf & 3) || (feat_cs & 3) || feat_cs < Cf) return IDH_EINVAL;
    if (B == 0 || P == 0) return IDH_OK;
    if (!feat_nhwc || !depth_bphw || !w1f_packed || !w2_f16 || !vecs6x128 || !out_bphw) return IDH_EINVAL;
    const long long M = (long long)B * HW;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    BinArgs a{feat_nhwc, depth_bphw, prior_bphw, w1f_packed, static_cast<const float *>(w2_f16), vecs6x128, out_bphw,
              (int)M, HW, P, feat_cs, Cf, has_prior, prior_const, 0, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, 0, nullptr};
    return binary_mlp_launch(a, B, stream, true);
}

// Per-pixel binary search for the depth at which the occlusion MLP flips (reference
// experiment_modules/bd_model.py:273-292, `infer_depth=True`): `iters` dependent MLP evaluations
// fused into one launch; bounds [lo, hi], first query (hi - lo)/2 as in the reference (0.5, 8.0, 3.75),
// "visible" when sigmoid(logit) < threshold.
extern "C" int idh_binary_mlp_search_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *prior_b1hw,
                                         int has_prior, float prior_const, const float *w1f_packed,
                                         const float *w2_packed, const float *vecs6x128, int B, int HW, int iters,
                                         float lo, float hi, float threshold, float *search_depths_b1hw,
                                         float *last_logits_b1hw, void *stream) {
    if (B < 0 || HW <= 0 || iters <= 0 || Cf <= 0 || (Cf & 3) || (feat_cs & 3) || feat_cs < Cf || !(threshold > 0.f) ||
        !(threshold < 1.f) || !(hi > lo))
        return IDH_EINVAL;
    if (B == 0) return IDH_OK;
    if (!feat_nhwc || !w1f_packed || !w2_packed || !vecs6x128 || !search_depths_b1hw || !last_logits_b1hw) return IDH_EINVAL;
    const long long M = (long long)B * HW;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    BinArgs a{feat_nhwc, nullptr, prior_b1hw, w1f_packed, w2_packed, vecs6x128, last_logits_b1hw,
              (int)M, HW, 1, feat_cs, Cf, has_prior, prior_const, iters, lo, hi, logf(threshold / (1.f - threshold)),
              search_depths_b1hw, nullptr, nullptr, 0, nullptr};
    return binary_mlp_launch(a, B, stream);
}

// Same search with the reference's per-depth Thresholder (utils/binary_metrics_utils.py:42-52, used by
// bd_model.py:282-283): threshold = thresholds[bucketize(query depth, bins)].  `thr_logits` holds
// logit(threshold) per bin (host-side transform, so the kernel compares logits like above).
extern "C" int idh_binary_mlp_search_thr_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *prior_b1hw,
                                             int has_prior, float prior_const, const float *w1f_packed,
                                             const float *w2_packed, const float *vecs6x128, int B, int HW, int iters,
                                             float lo, float hi, const float *bins, const float *thr_logits, int n_bins,
                                             float *search_depths_b1hw, float *last_logits_b1hw, void *stream) {
    if (B < 0 || HW <= 0 || iters <= 0 || Cf <= 0 || (Cf & 3) || (feat_cs & 3) || feat_cs < Cf || n_bins <= 0 || !(hi > lo))
        return IDH_EINVAL;
    if (B == 0) return IDH_OK;
    if (!feat_nhwc || !w1f_packed || !w2_packed || !vecs6x128 || !search_depths_b1hw || !last_logits_b1hw || !bins || !thr_logits)
        return IDH_EINVAL;
    const long long M = (long long)B * HW;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    BinArgs a{feat_nhwc, nullptr, prior_b1hw, w1f_packed, w2_packed, vecs6x128, last_logits_b1hw,
              (int)M, HW, 1, feat_cs, Cf, has_prior, prior_const, iters, lo, hi, 0.f,
              search_depths_b1hw, bins, thr_logits, n_bins, nullptr};
    return binary_mlp_launch(a, B, stream);
}

// f16x3 variants of the two search entry points (w2_f16 from idh_pack_mlp_weight_f16): the same kernel
// template with the per-plane layer in split precision
extern "C" int idh_binary_mlp_search_f16x3_fwd(const float *feat_nhwc, int feat_cs, int Cf, const float *prior_b1hw,
                                               int has_prior, float prior_const, const float *w1f_packed,
                                               const void *w2_f16, const float *vecs6x128, int B, int HW, int iters,
                                               float lo, float hi, float threshold, const float *bins,
                                               const float *thr_logits, int n_bins, float *search_depths_b1hw,
                                               float *last_logits_b1hw, void *stream) {
    if (B < 0 || HW <= 0 || iters <= 0 || Cf <= 0 || (Cf & 3) || (feat_cs & 3) || feat_cs < Cf || !(hi > lo) || n_bins < 0) return IDH_EINVAL;
    if (n_bins == 0 && (!(threshold > 0.f) || !(threshold < 1.f))) return IDH_EINVAL;
    if (n_bins > 0 && (!bins || !thr_logits)) return IDH_EINVAL;
    if (B == 0) return IDH_OK;
    if (!feat_nhwc || !w1f_packed || !w2_f16 || !vecs6x128 || !search_depths_b1hw || !last_logits_b1hw) return IDH_EINVAL;
    const long long M = (long long)B * HW;
    if (M >= (1ll << 31)) return IDH_EUNSUPPORTED;
    BinArgs a{feat_nhwc, nullptr, prior_b1hw, w1f_packed, static_cast<const float *>(w2_f16), vecs6x128, last_logits_b1hw,
              (int)M, HW, 1, feat_cs, Cf, has_prior, prior_const, iters, lo, hi,
              n_bins == 0 ? logf(threshold / (1.f - threshold)) : 0.f, search_depths_b1hw, bins, thr_logits, n_bins, nullptr};
    return binary_mlp_launch(a, B, stream, true);
}

static int binary_mlp_launch(BinArgs a, int B, void *stream, bool f16) {
    const long long M = a.M;
    constexpr int TM = 1;
    const int tiles = (int)((M + 16 * TM - 1) / (16 * TM));
    const int waves = bin_threads(f16) / 64;
    int grid = (tiles + waves - 1) / waves;
    if (grid > 256) grid = 256;  // persistent: one workgroup per CU (LDS-resident weights)
    (void)B;
    const int cblocks = (a.Cf + 15) >> 4;
    const size_t lds = ((size_t)kNS * kNS * 64 + (cblocks <= kW1LdsMaxBlocks ? (size_t)cblocks * kNS * 64 : 0)) * sizeof(f32x4) +
                       6 * kHidden * sizeof(float);
    static IdhDeviceOnce attr_set;
    if (attr_set.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(binary_mlp_k<TM, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(binary_mlp_k<TM, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return IDH_ELAUNCH;
        attr_set.mark();
    }
    if (f16) {
        a.sw2 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.w2) + (size_t)4 * kNS * 2 * 64 * 16);
        hipLaunchKernelGGL((binary_mlp_k<TM, true>), dim3(grid), dim3(bin_threads(true)), lds, idh_stream(stream), a);
    } else
        hipLaunchKernelGGL((binary_mlp_k<TM, false>), dim3(grid), dim3(bin_threads(false)), lds, idh_stream(stream), a);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}
