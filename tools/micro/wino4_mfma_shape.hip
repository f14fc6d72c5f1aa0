// What would v_mfma_f32_32x32x2_f32 buy the K loop of conv3x3_wino4_k?  (round-5 review, item 3.)
//
// Same flops per wave and stage (72 x 16x16x4 = 36 x 32x32x2 = 147,456 flop), same per-wave input transform (25 ds_read_b64 of the raw halo, 108 scalar
// fp32 operations, 18 ds_write_b32 of the transformed quadrant), one barrier per stage, 144 accumulator registers, two waves per SIMD - but
//   S16: the shipped shape.  256-thread workgroups, two per CU; a wave = 16 channels x 16 tiles x 36 positions; per stage 18 rows of
//        { 1 KiB A row from L2 (buffer_load_b128, ring of 3), 16-byte B read from LDS, 4 MFMAs 16x16x4 }; 6 halo copies (global -> registers ->
//        ds_write_b128) per pair of stages.
//   S32: what the 32x32 instruction allows with 144 accumulators: a wave = 32 channels x 32 tiles x 9 positions (a quadrant), 512-thread
//        workgroups (2 channel blocks x 4 quadrants), ONE per CU; per stage 9 rows of { 1 KiB A row, 16-byte B read, 4 MFMAs 32x32x2 }: HALF the
//        A rows, B reads and MFMA issue slots per flop; the halo of 32 tiles is shared by eight waves (3 copies per wave and pair of stages).
// The K loop only: the 32x32 shape's epilogue would additionally have to sum the four quadrants' partial output transforms through LDS (a wave holds
// 9 of a tile's 36 positions), which the shipped kernel does not need - see profiles/r06/experiments.md for the bill.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/wino4_mfma_shape.hip -o tools/micro/wino4_mfma_shape.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) char lds_char;

template <int SHAPE, int ABL>  // ABL bit 0: no A rows, bit 1: no transform, bit 2: no halo copies, bit 3: no B reads, bit 4: transform without its raw-halo reads, bit 5: ... without its V writes, bit 6: V writes as 2 x b128 + b32 per k-step
__global__ __launch_bounds__(SHAPE == 16 ? 256 : 512, 2) void k(const float *__restrict__ panel, const float *__restrict__ act, float *out, int stages, int panel_rows, int act_floats) {
    constexpr int kV = 2 * 64 * 36 * 4 * (SHAPE == 16 ? 1 : 2);  // transformed input of one stage (bytes)
    constexpr int kHalo = (SHAPE == 16 ? 340 : 612) * 32 + 256;
    __shared__ __attribute__((aligned(16))) char lds_raw[2 * kV + 3 * kHalo];
    lds_char *lds = (lds_char *)lds_raw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(panel), 0, panel_rows * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(act), 0, act_floats * 4, 0x00020000);
    f32x4 Af[3];
    float accs[144];
#pragma unroll
    for (int i = 0; i < 144; ++i) accs[i] = 0.f;
    for (int i = threadIdx.x; i < (int)sizeof(lds_raw) / 4; i += blockDim.x) ((__attribute__((address_space(3))) float *)lds)[i] = 1e-3f * i;
    __syncthreads();
    constexpr int kRows = SHAPE == 16 ? 18 : 9;
    int row = (blockIdx.x * 8 + wave) * kRows;
    auto ldA = [&](int slot) {
        if (ABL & 1) { Af[slot] = (f32x4){1.f, 2.f, 3.f, 4.f}; return; }
        Af[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, lane * 16, __builtin_amdgcn_readfirstlane((row % panel_rows) * 1024), 0));
        ++row;
    };
#pragma unroll
    for (int j = 0; j < 3; ++j) ldA(j);
    const int vbase = lane * 144;
    int hoff = (blockIdx.x * 131 + threadIdx.x) * 64;
    f32x4 stg[3];
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
        const int vr = (s & 1) * kV, vw = ((s + 1) & 1) * kV;
        f32x4 Bf[kRows];
        auto ldB = [&](int j) {
            if (ABL & 8) { Bf[j] = (f32x4){1.f, 2.f, 3.f, 4.f}; return; }
            Bf[j] = *(const __attribute__((address_space(3))) f32x4 *)(lds + vr + (j / 9) * (64 * 144) + vbase + 16 * (j % 9));
        };
        ldB(0); ldB(1);
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            if (j + 2 < kRows) ldB(j + 2);
            if constexpr (SHAPE == 16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x4 c = {accs[16 * (j % 9) + 4 * e], accs[16 * (j % 9) + 4 * e + 1], accs[16 * (j % 9) + 4 * e + 2], accs[16 * (j % 9) + 4 * e + 3]};
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[j % 3][e], Bf[j][e], c, 0, 0, 0);
                    accs[16 * (j % 9) + 4 * e] = c[0]; accs[16 * (j % 9) + 4 * e + 1] = c[1]; accs[16 * (j % 9) + 4 * e + 2] = c[2]; accs[16 * (j % 9) + 4 * e + 3] = c[3];
                }
            } else {
                f32x16 c;
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = accs[16 * j + i];
#pragma unroll
                for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[j % 3][e], Bf[j][e], c, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) accs[16 * j + i] = c[i];
            }
            ldA(j % 3);
            if (!(ABL & 4) && (s & 1) == 0 && j == kRows / 2) {  // halo copies of a pair of stages: loads here, stores after the loop
#pragma unroll
                for (int q = 0; q < 3; ++q) stg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, (hoff + 4096 * q) % (act_floats * 4 - 64), 0, 0));
                hoff += 64 * 1031;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(ABL & 4) && (s & 1) == 0) {
            constexpr int kCopies = SHAPE == 16 ? 2 : 1;  // S16: 6 copies per wave and pair of stages, S32: 3
#pragma unroll
            for (int rep = 0; rep < kCopies; ++rep)
#pragma unroll
                for (int q = 0; q < 3; ++q) *(__attribute__((address_space(3))) f32x4 *)(lds + 2 * kV + ((s >> 1) % 3) * kHalo + ((threadIdx.x * 48 + 16 * q + 1024 * rep) % (kHalo - 16) & ~15)) = stg[q];
        }
        if (!(ABL & 2)) {  // input transform of this wave's share: 25 ds_read_b64, 108 operations, 18 ds_write_b32
            const int rb = 2 * kV + (s % 3) * kHalo + (lane & 15) * 32 + (lane >> 4) * 8;
            float S[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) S[i] = 0.f;
#pragma unroll
            for (int c5 = 0; c5 < 5; ++c5) {
                f32x2 d[5];
#pragma unroll
                for (int r5 = 0; r5 < 5; ++r5) {
                    if (ABL & 16) { d[r5] = (f32x2){S[r5] + 1.f, S[r5 + 5] - 1.f}; asm volatile("" : "+v"(d[r5])); }
                    else d[r5] = *(const __attribute__((address_space(3))) volatile f32x2 *)(lds + rb + 544 * (5 * c5 + r5) % (kHalo - 1024));
                }
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float a = __builtin_fmaf(-4.f, d[2][ch], d[4][ch]), b = __builtin_fmaf(-4.f, d[1][ch], d[3][ch]);
                    const float w0 = __builtin_fmaf(-4.25f, d[2][ch], d[0][ch]) + d[4][ch], w1 = __builtin_fmaf(0.5f, b, a), w2 = __builtin_fmaf(-0.5f, b, a);
                    S[9 * ch + 0] = __builtin_fmaf(-4.25f, w0, S[9 * ch + 0]); S[9 * ch + 1] = __builtin_fmaf(-4.f, S[9 * ch + 1], w0); S[9 * ch + 2] += w0;
                    S[9 * ch + 3] = __builtin_fmaf(-4.25f, w1, S[9 * ch + 3]); S[9 * ch + 4] = __builtin_fmaf(-4.f, S[9 * ch + 4], w1); S[9 * ch + 5] += w1;
                    S[9 * ch + 6] = __builtin_fmaf(-4.25f, w2, S[9 * ch + 6]); S[9 * ch + 7] = __builtin_fmaf(-4.f, S[9 * ch + 7], w2); S[9 * ch + 8] += w2;
                }
            }
            if (ABL & 32) {
#pragma unroll
                for (int i = 0; i < 18; ++i) asm volatile("" ::"v"(S[i]));
            } else if (ABL & 64) {  // 8 + 1 floats per k-step: two aligned 16-byte stores + one dword
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    lds_char *o = lds + vw + ks * (64 * 144) + vbase;
                    *(__attribute__((address_space(3))) f32x4 *)(o + 32 * (wave & 3)) = (f32x4){S[9 * ks], S[9 * ks + 1], S[9 * ks + 2], S[9 * ks + 3]};
                    *(__attribute__((address_space(3))) f32x4 *)(o + 32 * (wave & 3) + 16) = (f32x4){S[9 * ks + 4], S[9 * ks + 5], S[9 * ks + 6], S[9 * ks + 7]};
                    *(__attribute__((address_space(3))) float *)(o + 128 + 4 * (wave & 3)) = S[9 * ks + 8];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 18; ++i) *(__attribute__((address_space(3))) float *)(lds + vw + (i / 9) * (64 * 144) + vbase + 4 * ((wave & 3) * 9 + i % 9)) = S[i];
            }
        }
        __syncthreads();
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 144; ++i) t += accs[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t + Af[0][0] + Af[1][1] + Af[2][2];
}

template <int SHAPE, int ABL>
double run(const float *panel, const float *act, float *out, int stages, int panel_rows, int act_floats, const char *what) {
    const int grid = SHAPE == 16 ? 512 : 256, threads = SHAPE == 16 ? 256 : 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, ABL>), dim3(grid), dim3(threads), 0, 0, panel, act, out, 64, panel_rows, act_floats);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, ABL>), dim3(grid), dim3(threads), 0, 0, panel, act, out, stages, panel_rows, act_floats);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
    // per SIMD and stage: 2 waves x 72 x 32 (= 36 x 64) cycles of matrix issue = 4608 cycles
    const double us_per_stage = ms * 1e3 / stages, tflops = 2048.0 * 72 * 8 * 256 * stages / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %6.3f us/stage  %6.1f TFLOP/s  = %.3f of 157.3\n", what, ms, us_per_stage, tflops, tflops / 157.3);
    return ms;
}

int main() {
    const int panel_rows = 8 * 4 * 18 * 2, act_floats = 64 << 20;  // 1.1 MiB of A rows (L2-resident), 256 MiB of activations (HBM)
    float *panel, *act, *out;
    hipMalloc(&panel, panel_rows * 1024); hipMalloc(&act, (size_t)act_floats * 4); hipMalloc(&out, 512 * 512 * 4);
    hipMemset(panel, 0, panel_rows * 1024); hipMemset(act, 0, (size_t)act_floats * 4);
    const int stages = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        run<16, 0>(panel, act, out, stages, panel_rows, act_floats, "S16 16x16x4 (shipped shape), all streams");
        run<32, 0>(panel, act, out, stages, panel_rows, act_floats, "S32 32x32x2, all streams");
    }
    run<16, 15>(panel, act, out, stages, panel_rows, act_floats, "S16 MFMAs + barrier only");
    run<32, 15>(panel, act, out, stages, panel_rows, act_floats, "S32 MFMAs + barrier only");
    run<16, 1>(panel, act, out, stages, panel_rows, act_floats, "S16 without the A rows");
    run<32, 1>(panel, act, out, stages, panel_rows, act_floats, "S32 without the A rows");
    run<16, 2>(panel, act, out, stages, panel_rows, act_floats, "S16 without the transform");
    run<32, 2>(panel, act, out, stages, panel_rows, act_floats, "S32 without the transform");
    run<16, 4>(panel, act, out, stages, panel_rows, act_floats, "S16 without the halo copies");
    run<32, 4>(panel, act, out, stages, panel_rows, act_floats, "S32 without the halo copies");
    run<16, 8>(panel, act, out, stages, panel_rows, act_floats, "S16 without the B reads");
    run<16, 16>(panel, act, out, stages, panel_rows, act_floats, "S16 transform without its 25 halo reads");
    run<16, 32>(panel, act, out, stages, panel_rows, act_floats, "S16 transform without its 18 V writes");
    run<16, 48>(panel, act, out, stages, panel_rows, act_floats, "S16 transform: the 108 operations only");
    run<16, 64>(panel, act, out, stages, panel_rows, act_floats, "S16 V writes as 2 x b128 + b32 per k-step");
    run<16, 0>(panel, act, out, stages, panel_rows, act_floats, "S16 all streams (again)");
    run<32, 8>(panel, act, out, stages, panel_rows, act_floats, "S32 without the B reads");
    return 0;
}
