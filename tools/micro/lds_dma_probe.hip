#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// probe: buffer_load_dwordx4 ... lds — (1) OOB lanes write zeros? (2) LDS destination above 64 KiB? (3) does the immediate offset move the LDS side too?
__global__ __launch_bounds__(64) void probe_k(const float *src, int nbytes, float *out) {
    __shared__ f32x4 lds[9216];  // 144 KiB
    const int lane = threadIdx.x;
    for (int i = lane; i < 9216; i += 64) lds[i] = (f32x4){-7.f, -7.f, -7.f, -7.f};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, nbytes, 0x00020000);
    // test 1: lanes 0..31 in range, 32..63 OOB, dest slot 0
    int voff = lane < 32 ? lane * 16 : 0x7fffffff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)&lds[0], 16, voff, 0, 0, 0);
    // test 2: dest at 100 KiB (slot 6400), all lanes in range, soffset 1024
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)&lds[6400], 16, lane * 16, 1024, 0, 0);
    // test 3: imm offset 32 bytes: dest slot 128
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)&lds[128], 16, lane * 16, 0, 32, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64; i += 64) {
        for (int e = 0; e < 4; ++e) {
            out[(0 * 64 + i) * 4 + e] = lds[i][e];
            out[(1 * 64 + i) * 4 + e] = lds[6400 + i][e];
            out[(2 * 64 + i) * 4 + e] = lds[128 + i][e];
            out[(3 * 64 + i) * 4 + e] = lds[128 + 2 + i][e];
        }
    }
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 4 * 64 * 4 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_k, dim3(1), dim3(64), 0, 0, d, n * 4, o);
    std::vector<float> r(4 * 64 * 4);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    printf("err %s\n", hipGetErrorString(hipGetLastError()));
    for (int t = 0; t < 4; ++t) {
        printf("test %d:", t);
        for (int i : {0, 1, 31, 32, 33, 63}) printf(" [%d]=%g,%g", i, r[(t * 64 + i) * 4], r[(t * 64 + i) * 4 + 3]);
        printf("\n");
    }
    return 0;
}
