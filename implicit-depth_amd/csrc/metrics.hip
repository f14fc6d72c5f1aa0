// GPU-side evaluation metrics: the step immediately after the hot path, producing the per-frame
// metric rows that are all-gathered across ranks.
//
//   plane IoU      reference utils/binary_metrics_utils.py:59-192 (PlaneEvaluator.compute_batch_scores /
//                  compute_batch_scores_test): per (frame, query plane, threshold) IoU of the predicted
//                  occlusion mask vs (query_depth < gt_depth), over pixels with gt > 0 and query > 0;
//                  positive-class, negative-class and their harmonic mean.
//   depth metrics  reference utils/metrics_utils.py:52-120 (compute_depth_metrics_batched): abs_diff,
//                  abs_rel, sq_rel, rmse, rmse_log and the a5..a3 inlier ratios over a validity mask.
// The reference builds ~10 full-size temporaries per call and abuses NaNs for masking; here each
// metric family is one counting / summing pass plus a tiny finalise kernel.  IoU counts are integer
// (exact, order independent); depth sums are accumulated per block and combined in a fixed order
// in double, so results are deterministic.
#include "idh_common.h"

namespace {

constexpr int kMaxThr = 8;

struct IouArgs {
    const float *query;   // B,D,N
    const float *gt;      // B,1,N
    const float *pred;    // B,D,N
    const float *thr;     // T constant thresholds, or per-bin thresholds when bins != null
    const float *bins;    // nb sorted bin edges (Thresholder.bins) or null
    int nb;
    int B, D, N, T;
    unsigned *counts;     // B*D*(2 + 2T): valid, target, pred[T], inter[T]
};

__global__ __launch_bounds__(256) void iou_count_k(const IouArgs a) {
    const int bd = blockIdx.y;
    const int b = bd / a.D;
    const float *q = a.query + (size_t)bd * a.N;
    const float *p = a.pred + (size_t)bd * a.N;
    const float *g = a.gt + (size_t)b * a.N;
    unsigned nv = 0, nt = 0, np[kMaxThr], ni[kMaxThr];
#pragma unroll
    for (int t = 0; t < kMaxThr; ++t) np[t] = ni[t] = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.N; i += gridDim.x * 256) {
        const float qd = q[i], gd = g[i], pv = p[i];
        if (!(gd > 0.f && qd > 0.f)) continue;  // valid mask (:70-72); NaN gt is invalid too
        ++nv;
        const bool tgt = qd < gd;
        nt += tgt;
        if (a.bins) {  // per-depth threshold: thresholds[bucketize(query, bins)] (:49-51), right=False
            int lo = 0, hi = a.nb;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.bins[mid] < qd) lo = mid + 1; else hi = mid; }
            if (lo >= a.nb) lo = a.nb - 1;  // query beyond the last bin edge (the reference would raise): last threshold, as csrc/mlp.hip
            const bool pr = pv > a.thr[lo];
            np[0] += pr; ni[0] += pr && tgt;
        } else {
#pragma unroll
            for (int t = 0; t < kMaxThr; ++t)
                if (t < a.T) { const bool pr = pv > a.thr[t]; np[t] += pr; ni[t] += pr && tgt; }
        }
    }
    // wave reduce then one integer atomic per counter per wave
    auto wred = [](unsigned v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    unsigned *c = a.counts + (size_t)bd * (2 + 2 * a.T);
    nv = wred(nv); nt = wred(nt);
    if ((threadIdx.x & 63) == 0) { atomicAdd(c, nv); atomicAdd(c + 1, nt); }
#pragma unroll
    for (int t = 0; t < kMaxThr; ++t)
        if (t < a.T) {
            const unsigned x = wred(np[t]), y = wred(ni[t]);
            if ((threadIdx.x & 63) == 0) { atomicAdd(c + 2 + t, x); atomicAdd(c + 2 + a.T + t, y); }
        }
}

// out[b,d,t,{iou, iou_pos, iou_neg}] with the reference's float arithmetic (0/0 -> NaN kept)
__global__ void iou_finalise_k(const unsigned *__restrict__ counts, int BD, int T, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BD * T) return;
    const int bd = i / T, t = i - bd * T;
    const unsigned *c = counts + (size_t)bd * (2 + 2 * T);
    const float nv = (float)c[0], nt = (float)c[1], np = (float)c[2 + t], ni = (float)c[2 + T + t];
    const float pos = ni / (nt + np - ni);
    const float nn_t = nv - nt, nn_p = nv - np, nn_i = nv - nt - np + ni;  // counts of the negated masks
    const float neg = nn_i / (nn_t + nn_p - nn_i);
    const float iou = 2.f * (pos * neg) / (pos + neg);
    out[(size_t)i * 3 + 0] = iou;
    out[(size_t)i * 3 + 1] = pos;
    out[(size_t)i * 3 + 2] = neg;
}

constexpr int kDM = 12;  // abs_diff abs_rel sq_rel rmse rmse_log a5 a10 a25 a0 a1 a2 a3
constexpr int kDmChunk = 4096;

__global__ __launch_bounds__(256) void depth_metrics_partial_k(const float *__restrict__ gt, const float *__restrict__ pred,
                                                               const unsigned char *__restrict__ valid, int N, int nchunks,
                                                               double *__restrict__ part) {  // part[b][chunk][13]
    __shared__ double red[4][kDM + 1];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int i0 = chunk * kDmChunk, i1 = min(N, i0 + kDmChunk);
    double s[kDM + 1];
    for (int k = 0; k <= kDM; ++k) s[k] = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        if (!valid[(size_t)b * N + i]) continue;
        const float g = gt[(size_t)b * N + i], p = pred[(size_t)b * N + i];
        const float d = g - p;
        const float th = fmaxf(g / p, p / g);
        const float lg = logf(g) - logf(p);
        s[0] += fabsf(d); s[1] += fabsf(d) / g; s[2] += d * d / g; s[3] += d * d; s[4] += lg * lg;
        s[5] += th < 1.05f; s[6] += th < 1.10f; s[7] += th < 1.25f; s[8] += th < 1.10f; s[9] += th < 1.25f;
        s[10] += th < 1.25f * 1.25f; s[11] += th < 1.25f * 1.25f * 1.25f;
        s[12] += 1.0;
    }
    for (int k = 0; k <= kDM; ++k) {
        double v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x <= kDM)
        part[((size_t)b * nchunks + chunk) * (kDM + 1) + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void depth_metrics_finalise_k(const double *__restrict__ part, int nchunks, int mult_a, float *__restrict__ out) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= kDM) return;
    double s = 0.0, n = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        s += part[((size_t)b * nchunks + c) * (kDM + 1) + k];
        n += part[((size_t)b * nchunks + c) * (kDM + 1) + kDM];
    }
    double m = s / n;  // nanmean over the valid pixels (0/0 -> NaN like torch.nanmean of an all-NaN row)
    if (k == 3 || k == 4) m = sqrt(m);
    if (k >= 5 && mult_a) m *= 100.0;
    out[(size_t)b * kDM + k] = (float)m;
}

}  // namespace

extern "C" size_t idh_metrics_workspace_bytes(int B, int D, int N, int T) {
    if (B <= 0 || N <= 0) return 0;
    const size_t iou = (size_t)B * (D > 0 ? D : 1) * (2 + 2 * (T > 0 ? T : 1)) * sizeof(unsigned);
    const size_t dm = (size_t)B * ((N + kDmChunk - 1) / kDmChunk) * (kDM + 1) * sizeof(double);
    return (iou > dm ? iou : dm) + 64;
}

extern "C" int idh_plane_iou_fwd(const float *query_depth_bdn, const float *gt_depth_b1n, const float *prediction_bdn,
                                 const float *thresholds, int T, const float *bins, int n_bins, int B, int D, int N,
                                 float *out_bdt3, void *workspace, size_t workspace_bytes, void *stream) {
    if (B < 0 || D <= 0 || N <= 0 || T <= 0 || T > kMaxThr || (bins && (T != 1 || n_bins <= 0))) return IDH_EINVAL;
    if (B == 0) return IDH_OK;
    if (!query_depth_bdn || !gt_depth_b1n || !prediction_bdn || !thresholds || !out_bdt3) return IDH_EINVAL;
    const size_t need = (size_t)B * D * (2 + 2 * T) * sizeof(unsigned);
    if (!workspace || workspace_bytes < need) return IDH_EWORKSPACE;
    if ((long long)B * D > 65535) return IDH_EUNSUPPORTED;
    hipStream_t st = idh_stream(stream);
    if (hipMemsetAsync(workspace, 0, need, st) != hipSuccess) return IDH_ELAUNCH;
    IouArgs a{query_depth_bdn, gt_depth_b1n, prediction_bdn, thresholds, bins, n_bins, B, D, N, T, static_cast<unsigned *>(workspace)};
    int gx = idh_cdiv(N, 256 * 8);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(iou_count_k, dim3(gx, B * D), dim3(256), 0, st, a);
    IDH_CHECK_LAUNCH();
    hipLaunchKernelGGL(iou_finalise_k, dim3(idh_cdiv((long long)B * D * T, 128)), dim3(128), 0, st, a.counts, B * D, T, out_bdt3);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

extern "C" int idh_depth_metrics_fwd(const float *gt_bn, const float *pred_bn, const unsigned char *valid_bn, int B, int N,
                                     int mult_a, float *out_b12, void *workspace, size_t workspace_bytes, void *stream) {
    if (B < 0 || N <= 0) return IDH_EINVAL;
    if (B == 0) return IDH_OK;
    if (!gt_bn || !pred_bn || !valid_bn || !out_b12) return IDH_EINVAL;
    const int nchunks = (N + kDmChunk - 1) / kDmChunk;
    const size_t need = (size_t)B * nchunks * (kDM + 1) * sizeof(double);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 7)) return IDH_EWORKSPACE;
    if (B > 65535) return IDH_EUNSUPPORTED;
    hipStream_t st = idh_stream(stream);
    double *part = static_cast<double *>(workspace);
    hipLaunchKernelGGL(depth_metrics_partial_k, dim3(nchunks, B), dim3(256), 0, st, gt_bn, pred_bn, valid_bn, N, nchunks, part);
    IDH_CHECK_LAUNCH();
    hipLaunchKernelGGL(depth_metrics_finalise_k, dim3(B), dim3(64), 0, st, part, nchunks, mult_a, out_b12);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}
