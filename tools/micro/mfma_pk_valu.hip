// One wave per SIMD (256-thread workgroups, one per CU): what do vector instructions cost beside v_mfma_f32_16x16x4_f32 when they are
// interleaved ONE BY ONE with independent MFMAs — plain v_fma_f32, packed v_pk_fma_f32 (two FMAs per lane), integer v_add_u32, and
// LDS reads?  Per iteration: 12 MFMAs + N fillers placed after each MFMA (sched_barrier keeps the order).
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_pk_valu.hip -o tools/micro/mfma_pk_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 none, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_add_u32, 4 ds_read_b64;  PER = fillers per MFMA
template <int KIND, int PER>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters) {
    __shared__ f32x2 lds[1024];
    lds[threadIdx.x] = (f32x2){1.f, 2.f};
    __syncthreads();
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8];
    f32x2 p[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; p[i] = (f32x2){a + i, a - i}; u[i] = threadIdx.x + i; }
    const f32x2 pb = (f32x2){b, b}, pa = (f32x2){a, a};
    f32x2 ld = (f32x2){0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int j = (i * PER + r) & 7;
                if (KIND == 1) v[j] = __builtin_fmaf(v[j], b, a);
                if (KIND == 2) p[j] = __builtin_elementwise_fma(p[j], pb, pa);
                if (KIND == 3) u[j] = u[j] + 0x9e3779b9u * (unsigned)it;
                if (KIND == 4) { ld += *(volatile f32x2 *)&lds[(threadIdx.x + 8 * j) & 1023]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = ld[0] + ld[1];
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int PER>
float run(float *out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, PER><<<256, 256>>>(out, 10);
    hipEventRecord(e0);
    k<KIND, PER><<<256, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    const float base = run<0, 1>(out, iters);
    printf("12 MFMAs per iteration, one wave per SIMD: %.2f ms = %.1f ns per MFMA\n", base, base * 1e6 / iters / 12);
#define ROW(name, KIND) { const float t1 = run<KIND, 1>(out, iters), t2 = run<KIND, 2>(out, iters), t4 = run<KIND, 4>(out, iters); \
    printf("%-14s 1 / 2 / 4 per MFMA: %.2f / %.2f / %.2f ms  -> +%.1f / +%.1f / +%.1f %% ; per filler %.2f / %.2f / %.2f ns (an MFMA: %.1f ns)\n", name, t1, t2, t4, \
           100 * (t1 / base - 1), 100 * (t2 / base - 1), 100 * (t4 / base - 1), (t1 - base) * 1e6 / iters / 12, (t2 - base) * 1e6 / iters / 24, (t4 - base) * 1e6 / iters / 48, base * 1e6 / iters / 12); }
    ROW("v_fma_f32", 1)
    ROW("v_pk_fma_f32", 2)
    ROW("v_add_u32/mul", 3)
    ROW("ds_read_b64", 4)
    return 0;
}
