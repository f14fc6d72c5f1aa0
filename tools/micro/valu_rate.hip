// Issue cost of the vector instructions cv_dot_win_k is made of, in SIMD cycles per wave64 instruction (2.4 GHz assumed):
// 8 independent registers, NI instructions per loop iteration, 1 or 2 waves per SIMD (256- / 512-thread workgroups, one per CU).
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, int iters) {
    float v[8];
    f32x2 p[8];
    int n[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = (f32x2){v[i], v[i] + 1.f}; n[i] = threadIdx.x + i; }
    float a = 1.0001f, b = 0.5f;
    unsigned long long smask;
    asm volatile("s_mov_b64 %0, exec" : "=s"(smask));
    f32x2 pa = {1.0001f, 0.9999f}, pb = {0.5f, 0.25f};
    int m = 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 3) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 4) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                REP8(X)
#undef X
            } else if (KIND == 5) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : );
                REP8(X)
#undef X
            } else if (KIND == 6) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(a));
                REP8(X)
#undef X
            } else if (KIND == 7) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
                REP8(X)
#undef X
            } else if (KIND == 8) {  // one dependent chain of packed FMAs
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                             "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pa), "v"(pb));
            } else if (KIND == 9) {
#define X(i) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(n[i]), "v"(m) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == 10) {
#define X(i) asm volatile("v_floor_f32 %0, %0" : "+v"(v[i]));
                REP8(X)
#undef X
            } else if (KIND == 11) {
#define X(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(n[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 12) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "s"(smask));
                REP8(X)
#undef X
            } else if (KIND == 13) {  // compare + select through vcc (counted as 2 instructions)
#define X(i) asm volatile("v_cmp_gt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(v[i]) : "v"(n[i]), "v"(m), "v"(a) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == 14) {  // compare + select through an SGPR pair (2 instructions)
#define X(i) { unsigned long long t; asm volatile("v_cmp_gt_i32_e64 %1, %2, %3\n v_cndmask_b32_e64 %0, %0, %4, %1" : "+v"(v[i]), "=&s"(t) : "v"(n[i]), "v"(m), "v"(a)); }
                REP8(X)
#undef X
            } else if (KIND == 15) {  // two compares, s_and, select (the shape of an "inside the window" test), 4 instructions
#define X(i) { unsigned long long t, u; asm volatile("v_cmp_gt_i32_e64 %1, %3, %4\n v_cmp_lt_i32_e64 %2, %3, %4\n s_and_b64 %1, %1, %2\n v_cndmask_b32_e64 %0, %0, %5, %1" : "+v"(v[i]), "=&s"(t), "=&s"(u) : "v"(n[i]), "v"(m), "v"(a)); }
                REP8(X)
#undef X
            } else if (KIND == 16) {  // v_max_f32 / v_med3 style branch-free alternatives
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + n[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int KIND, int THREADS>
float run(float *out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, THREADS><<<256, THREADS>>>(out, 10);
    hipEventRecord(e0);
    k<KIND, THREADS><<<256, THREADS>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_mul_lo_u32", "v_rcp_f32", "v_cndmask_b32", "v_mov_b32", "v_pk_mul_f32",
                           "v_pk_fma_f32 (one dependent chain)", "v_cmp_gt_i32", "v_floor_f32", "v_min_i32", "v_cndmask_b32_e64 (sgpr mask)",
                           "v_cmp vcc + v_cndmask vcc (x2)", "v_cmp_e64 s + v_cndmask_e64 s (x2)", "2 v_cmp + s_and + v_cndmask (x4)", "v_med3_f32"};
#define ROW(K) { const float m1 = run<K, 256>(out, iters), m2 = run<K, 512>(out, iters); \
    printf("%-36s 1 wave/SIMD %6.2f cycles/instr   2 waves/SIMD %6.2f cycles/instr (per SIMD)\n", names[K], m1 * 1e-3 * 2.4e9 / (64.0 * iters), m2 * 1e-3 * 2.4e9 / (128.0 * iters)); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16)
    return 0;
}
