// How fast can ONE wave issue fp32 MFMAs, and how many waves per SIMD does the matrix pipe need to be full?
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_issue_rate.hip -o tools/micro/mfma_issue_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// SHAPE 0: 16x16x4 f32 (2048 flop), 1: 32x32x2 f32 (4096 flop).  WAVES = waves per SIMD (workgroup = 256 * WAVES threads).
template <int SHAPE, int NACC>
__global__ void k(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float s = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NACC>
void run(float *out, int waves) {
    const int iters = 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, NACC><<<256, 256 * waves>>>(out, 10);
    hipEventRecord(e0);
    k<SHAPE, NACC><<<256, 256 * waves>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)iters * 32 * waves;
    const double cyc = ms * 1e-3 * 2.4e9 / mfmas_per_simd;
    const double flop = SHAPE == 0 ? 2048.0 : 4096.0;
    printf("%s  %d independent accumulators  %d wave(s) / SIMD: %6.1f cycles per MFMA per SIMD -> %5.1f %% of the 64 flop/cycle/SIMD peak\n",
           SHAPE == 0 ? "16x16x4 " : "32x32x2 ", NACC, waves, cyc, 100.0 * flop / cyc / 64.0);
}

int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {1, 2, 3, 4}) run<0, 8>(out, w);
    for (int w : {1, 2}) run<0, 2>(out, w);
    for (int w : {1, 2}) run<0, 1>(out, w);
    for (int w : {1, 2, 3}) run<1, 4>(out, w);
    for (int w : {1, 2}) run<1, 1>(out, w);
    return 0;
}
