// L1-gather micro-benchmark: how fast can a CU pull 64-byte "taps" (16 fp32 channels of one pixel)
// out of an L2/MALL-resident NHWC image, depending on how the 64 bytes are spread over lanes?
//   mode 0: one lane per tap, 4 sequential float4 loads            (cv_dot_k v1)
//   mode 1: 4 ADJACENT lanes per tap, one float4 each              (quad-coalesced)
//   mode 2: 4 lanes per tap at lane stride 16 (q = lane>>4)        (conv_mfma_k / fv_mlp_k fragment shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void gather_k(const float4 *__restrict__ img, int npix, int iters, int mode, float *out) {
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned s = gw * 9781u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned base = (s >> 8) % (npix - 128);   // wave-uniform window start, lanes spread over 64 px
        if (mode == 0) {
            const unsigned p = base + lane + ((lane * 7) & 15);
            for (int q = 0; q < 4; ++q) { float4 v = img[p * 4 + q]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        } else if (mode == 1) {
            for (int r = 0; r < 4; ++r) {
                const unsigned p = base + (lane >> 2) + 16 * r + (((lane >> 2) * 7) & 15);
                float4 v = img[p * 4 + (lane & 3)]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        } else {
            for (int r = 0; r < 4; ++r) {
                const unsigned p = base + (lane & 15) + 16 * r + (((lane & 15) * 7) & 15);
                float4 v = img[p * 4 + (lane >> 4)]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
int main() {
    const int npix = 8 * 96 * 128 * 4;  // 25 MB image set: L2/MALL resident like the source features of 4 frames
    float4 *img; float *out;
    (void)hipMalloc(&img, (size_t)npix * 64); (void)hipMalloc(&out, 64);
    (void)hipMemset(img, 0, (size_t)npix * 64);
    const int iters = 2000, blocks = 256 * 8;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        gather_k<<<blocks, 256>>>(img, npix, 50, mode, out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        gather_k<<<blocks, 256>>>(img, npix, iters, mode, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)blocks * 4 * iters * 64 * 64.0;  // waves * iters * 64 lanes... each wave-iter moves 4 KiB
        printf("mode %d: %.3f ms  %.1f TB/s  %.1f B/clk/CU @2.4GHz\n", mode, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
