// Can the vector ALU of a SIMD issue one wave's VALU instructions while another wave's fp32 MFMAs execute?
// 512-thread workgroups = 2 waves per SIMD.  Modes: 0 = both waves MFMA only, 1 = both VALU only,
// 2 = waves 0-3 MFMA / waves 4-7 VALU, i.e. one of each per SIMD (waves go to SIMD wave % 4) (specialised), 3 = every wave alternates blocks of 32 MFMAs and 64 VALU.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_coissue.hip -o tools/micro/mfma_valu_coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int VR>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && ((wave >> 2) & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && ((wave >> 2) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < VR; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], b, a);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int VR>
float run(float *out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, VR><<<256, 512>>>(out, 10);
    hipEventRecord(e0);
    k<MODE, VR><<<256, 512>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    {
        const float m = run<0, 8>(out, iters), v = run<1, 8>(out, iters), sp = run<2, 8>(out, iters), alt = run<3, 8>(out, iters);
        printf("64 VALU per 32 MFMAs:  MFMA only (2 waves / SIMD) %.2f ms, VALU only %.2f ms, specialised 1+1 %.2f ms (MFMA half alone %.2f), alternating %.2f ms (sum %.2f)\n", m, v, sp, m / 2, alt, m + v);
    }
    {
        const float m = run<0, 40>(out, iters), v = run<1, 40>(out, iters), sp = run<2, 40>(out, iters), alt = run<3, 40>(out, iters);
        printf("320 VALU per 32 MFMAs: MFMA only (2 waves / SIMD) %.2f ms, VALU only %.2f ms, specialised 1+1 %.2f ms (MFMA half alone %.2f, VALU half alone %.2f), alternating %.2f ms (sum %.2f)\n", m, v, sp, m / 2, v / 2, alt, m + v);
    }
    return 0;
}
