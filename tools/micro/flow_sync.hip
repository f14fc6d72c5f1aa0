// What does a device-wide dependency wait cost inside one persistent kernel on gfx950 (8 XCDs, non-coherent L2s)?
// 768 workgroups run `levels` rounds; in each round a workgroup reads 40 KB of shared "weights" + a 7 KB tile another
// workgroup (other XCD) wrote in the previous round, writes its own tile, then release-increments a counter and
// acquire-waits until all workgroups of the round have arrived.  Compare with the same rounds as separate launches.
// hipcc --offload-arch=gfx950 -O3 tools/micro/flow_sync.hip -o tools/micro/flow_sync.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kTile = 1792;   // floats per workgroup tile (7 KB)
constexpr int kW = 10240;     // floats of shared weights (40 KB)

__device__ __forceinline__ float round_body(const float *w, const float *src, float *dst, int tid) {
    float acc = 0.f;
    for (int i = tid; i < kW; i += 256) acc += w[i];
    for (int i = tid; i < kTile; i += 256) acc += src[i];
    for (int i = tid; i < kTile; i += 256) dst[i] = acc * 1e-9f + 1.0f;
    return acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void flow_k(const float *w, float *buf0, float *buf1, unsigned *counter, int levels, int *err) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    for (int l = 0; l < levels; ++l) {
        const float *src = ((l & 1) ? buf1 : buf0) + (size_t)((b * 37 + 101) % G) * kTile;  // some other workgroup's tile
        float *dst = ((l & 1) ? buf0 : buf1) + (size_t)b * kTile;
        const float a = round_body(w + (size_t)(l % 8) * kW, src, dst, tid);
        if (l > 0 && tid == 0 && src[5] != 1.0f + 0.f * a && !(src[5] > 0.99f && src[5] < 1.01f)) atomicAdd(err, 1);
        __syncthreads();
        if (tid == 0) {
            if (MODE != 1) __threadfence();  // release: this workgroup's tile is visible device-wide
            atomicAdd(counter, 1u);
            const unsigned want = (unsigned)G * (l + 1);
            if (MODE == 2) { while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(40); }
            else if (MODE == 3) { while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want - (unsigned)G + G / 2) __builtin_amdgcn_s_sleep(2); }
            else { while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2); }
            if (MODE != 1) __threadfence();  // acquire
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void round_k(const float *w, const float *bufs, float *bufd, int l) {
    const int G = gridDim.x, b = blockIdx.x;
    round_body(w + (size_t)(l % 8) * kW, bufs + (size_t)((b * 37 + 101) % G) * kTile, bufd + (size_t)b * kTile, threadIdx.x);
}

// split-K style "last arriver reduces": every workgroup writes its tile, fences, bumps the counter of its group of S
// workgroups; the last one fences again and re-reads the S tiles.  One launch per round (no device-wide wait).
__global__ __launch_bounds__(256) void last_arriver_k(const float *w, const float *bufs, float *bufd, unsigned *cnt, int l, int S, float *out) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned s_prev;
    const float a = round_body(w + (size_t)(l % 8) * kW, bufs + (size_t)((b * 37 + 101) % G) * kTile, bufd + (size_t)b * kTile, tid);
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_prev = atomicAdd(&cnt[b / S], 1u);
    }
    __syncthreads();
    if (s_prev % S != (unsigned)S - 1) return;
    __threadfence();
    float acc = a * 0.f;
    for (int s = 0; s < S; ++s)
        for (int i = tid; i < kTile; i += 256) acc += bufd[(size_t)((b / S) * S + s) * kTile + i];
    out[(size_t)(b / S) * 256 + tid] = acc;
}

int main() {
    const int G = 768, levels = 100;
    float *w, *b0, *b1; unsigned *cnt; int *err;
    hipMalloc(&w, 8 * kW * 4); hipMalloc(&b0, (size_t)G * kTile * 4); hipMalloc(&b1, (size_t)G * kTile * 4);
    hipMalloc(&cnt, 4); hipMalloc(&err, 4);
    std::vector<float> ones((size_t)G * kTile, 1.0f);
    hipMemcpy(b0, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
    hipMemset(w, 0, 8 * kW * 4); hipMemset(err, 0, 4);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, flow_k<0>, 256, 0);
    printf("occupancy %d workgroups / CU -> %d resident (grid %d)\n", occ, occ * 256, G);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int mode = 0; mode < 4; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cnt, 0, 4);
        hipEventRecord(e0);
        if (mode == 0) flow_k<0><<<G, 256>>>(w, b0, b1, cnt, levels, err);
        if (mode == 1) flow_k<1><<<G, 256>>>(w, b0, b1, cnt, levels, err);
        if (mode == 2) flow_k<2><<<G, 256>>>(w, b0, b1, cnt, levels, err);
        if (mode == 3) flow_k<3><<<G, 256>>>(w, b0, b1, cnt, levels, err);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("persistent mode %d (0 fences+poll, 1 no fences, 2 slow poll, 3 wait for half): %.2f us / round\n", mode, ms * 1e3 / levels);
    }
    int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("stale reads seen: %d\n", herr);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < levels; ++l) round_k<<<G, 256>>>(w, (l & 1) ? b1 : b0, (l & 1) ? b0 : b1, l);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch per round: %.2f us / round\n", ms * 1e3 / levels);
    }
    unsigned *cnt2; float *out;
    hipMalloc(&cnt2, 768 * 4); hipMemset(cnt2, 0, 768 * 4); hipMalloc(&out, 768 * 256 * 4);
    for (int S : {2, 4, 8})
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < levels; ++l) last_arriver_k<<<G, 256>>>(w, (l & 1) ? b1 : b0, (l & 1) ? b0 : b1, cnt2, l, S, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("launch per round + last-arriver reduce over S=%d: %.2f us / round\n", S, ms * 1e3 / levels);
        }
    return 0;
}
