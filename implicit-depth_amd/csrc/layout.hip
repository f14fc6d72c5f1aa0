// NCHW <-> NHWC conversion.  The reference keeps everything NCHW; the gfx950 kernels want
// channels innermost so that one bilinear tap / one conv K-slice is a single contiguous
// 16-byte-vector read.  Converted once at the module boundary.
#include "idh_common.h"

namespace {

// One thread = (pixel, group of 4 channels): reads 4 strided floats (coalesced across the
// 16 lanes that share a channel group), writes one float4 (fully coalesced across the wave).
__global__ __launch_bounds__(256) void nchw_to_nhwc_k(const float *__restrict__ src,
                                                      float *__restrict__ dst, int C, int HW,
                                                      long long total_q) {
    const int cq = C >> 2;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total_q; t += gridDim.x * 256ll) {
        int q = (int)(t % cq);
        long long pimg = t / cq;  // img*HW + pixel
        long long img = pimg / HW;
        int p = (int)(pimg - img * HW);
        const float *s = src + (img * C + 4 * q) * (long long)HW + p;
        float4 v = make_float4(s[0], s[HW], s[2ll * HW], s[3ll * HW]);
        *reinterpret_cast<float4 *>(dst + pimg * C + 4 * q) = v;
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_generic_k(const float *__restrict__ src,
                                                              float *__restrict__ dst, int C, int HW,
                                                              long long total) {
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
        int c = (int)(t % C);
        long long pimg = t / C;
        long long img = pimg / HW;
        int p = (int)(pimg - img * HW);
        dst[t] = src[(img * C + c) * (long long)HW + p];
    }
}

// NHWC -> NCHW through an LDS tile so both sides are coalesced: tile = 64 pixels x 32 channels.
__global__ __launch_bounds__(256) void nhwc_to_nchw_k(const float *__restrict__ src,
                                                      float *__restrict__ dst, int C, int HW) {
    __shared__ float tile[32][65];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 32;
    const float *s = src + (long long)img * HW * C;
    float *d = dst + (long long)img * HW * C;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        int c = i & 31, p = i >> 5;
        if (p0 + p < HW && c0 + c < C) tile[c][p] = s[(long long)(p0 + p) * C + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        int p = i & 63, c = i >> 6;
        if (p0 + p < HW && c0 + c < C) d[(long long)(c0 + c) * HW + p0 + p] = tile[c][p];
    }
}

}  // namespace

extern "C" int idh_nchw_to_nhwc_f32(const float *src, float *dst, int n_img, int C, int HW, void *stream) {
    if (!src || !dst || n_img < 0 || C <= 0 || HW <= 0) return IDH_EINVAL;
    if (n_img == 0) return IDH_OK;
    if ((C & 3) == 0) {
        long long total = (long long)n_img * HW * (C >> 2);
        int grid = idh_cdiv(total, 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(nchw_to_nhwc_k, dim3(grid), dim3(256), 0, idh_stream(stream), src, dst, C, HW, total);
    } else {
        long long total = (long long)n_img * HW * C;
        int grid = idh_cdiv(total, 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(nchw_to_nhwc_generic_k, dim3(grid), dim3(256), 0, idh_stream(stream), src, dst, C, HW, total);
    }
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}

extern "C" int idh_nhwc_to_nchw_f32(const float *src, float *dst, int n_img, int C, int HW, void *stream) {
    if (!src || !dst || n_img < 0 || C <= 0 || HW <= 0) return IDH_EINVAL;
    if (n_img == 0) return IDH_OK;
    if (n_img > 65535) return IDH_EUNSUPPORTED;
    dim3 grid(idh_cdiv(HW, 64), idh_cdiv(C, 32), n_img);
    hipLaunchKernelGGL(nhwc_to_nchw_k, grid, dim3(256), 0, idh_stream(stream), src, dst, C, HW);
    IDH_CHECK_LAUNCH();
    return IDH_OK;
}
