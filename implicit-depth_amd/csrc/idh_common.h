// Shared helpers for the gfx950 kernels (device + host side).  gfx950 only: wave64,
// 256 CUs in 8 XCDs, 160 KiB LDS per CU.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/idh.h"

#define IDH_WAVE 64

#define IDH_CHECK_LAUNCH()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return IDH_ELAUNCH; \
    } while (0)

// Per-device "already configured" flag for per-function attributes (hipFuncSetAttribute is a
// per-device setting).  True exactly once per (flag set, device); devices beyond the table are
// configured on every call.  Relaxed atomics: a race only repeats an idempotent call.
struct IdhDeviceOnce {
    unsigned char done[64] = {};
    bool first() {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) return false;
        return true;
    }
    void mark() {
        int dev = -1;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) __atomic_store_n(&done[dev], 1, __ATOMIC_RELEASE);
    }
};

static inline hipStream_t idh_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int idh_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// XCD-aware block remap (guide T1): hardware places block b on XCD b % 8, so consecutive
// logical tiles land on different L2s.  Remap so each XCD owns a contiguous run of logical
// tiles (neighbouring tiles share halo rows / weight panels in its private L2).  Bijective for
// any grid size.
__device__ __forceinline__ unsigned idh_xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned NX = 8;
    unsigned full = nblocks / NX, rem = nblocks % NX;
    unsigned xcd = bid % NX, slot = bid / NX;
    // XCDs [0,rem) own full+1 tiles, the rest own `full`
    unsigned start = xcd * full + (xcd < rem ? xcd : rem);
    return start + slot;
}
