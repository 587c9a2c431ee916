// Host-side helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "stof_common.h"

namespace stof {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting, so a process that drives several GPUs
// has to make it on each of them.  One instance per kernel: a bit per device ordinal, set after a successful call.
// Relaxed atomics are enough -- losing a race only repeats an idempotent runtime call -- and this is the only state
// the library keeps (it caches a runtime setting, not data).
struct LdsLimitOnce {
    std::atomic<uint64_t> done[4];
    constexpr LdsLimitOnce() : done{} {}
    int ensure(const void* kernel, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return STOF_ERR_HIP;
        const bool tracked = dev >= 0 && dev < 256;
        if (tracked && (done[dev >> 6].load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) return STOF_OK;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return STOF_ERR_HIP;
        if (tracked) done[dev >> 6].fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        return STOF_OK;
    }
};

inline int device_cu_count() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        return v;
    return 256;
}

}  // namespace stof
