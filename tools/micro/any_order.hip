// Does hipExtAnyOrderLaunch let two kernels of one stream overlap on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/micro/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin_k(float *out, long long ticks) {
    const long long t0 = wall_clock64();
    float acc = 0.f;
    while (wall_clock64() - t0 < ticks) acc += 1.f;
    if (acc < 0.f) out[blockIdx.x] = acc;
}

static float run(int pairs, int flags, int wgs, long long ticks, hipStream_t s, float *buf) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < pairs; ++i) {
        hipExtLaunchKernelGGL(spin_k, dim3(wgs), dim3(256), 0, s, nullptr, nullptr, 0, buf, ticks);
        hipExtLaunchKernelGGL(spin_k, dim3(wgs), dim3(256), 0, s, nullptr, nullptr, flags, buf, ticks);
    }
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / pairs;
}

int main() {
    float *buf;
    hipMalloc(&buf, 1 << 20);
    hipStream_t s;
    hipStreamCreate(&s);
    const long long ticks = 5000;  // 100 MHz wall clock: 50 us
    for (int wgs : {64, 256, 1024}) {
        run(4, 0, wgs, ticks, s, buf);
        printf("wgs=%4d: in-order pair %.1f us, second launch any-order %.1f us (one kernel = 50 us)\n", wgs,
               run(20, 0, wgs, ticks, s, buf), run(20, hipExtAnyOrderLaunch, wgs, ticks, s, buf));
    }
    // empty-kernel launch cadence
    printf("back-to-back 1 us kernels: in-order %.2f us / pair, any-order %.2f us / pair\n", run(200, 0, 256, 100, s, buf), run(200, hipExtAnyOrderLaunch, 256, 100, s, buf));
    return 0;
}
