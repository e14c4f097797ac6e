// Shared host/device helpers for libmvs_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mvs_hip.h"

namespace mvs {

void set_error(const char* fmt, ...);

inline int finish_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return -(1000 + (int)e);
    }
    return MVS_OK;
}

// out[j] = sum_p part[p * n + j], p = 0..nparts-1, in a fixed order (one wave per j: lanes stride over p, then an xor-shuffle tree).
// The deterministic second stage of every block-partial reduction in the library (BatchNorm statistics and their backward).
void launch_partials_reduce(const float* part, int nparts, int n, float* out, hipStream_t stream);

// Grouped form over partial rows of width 2*C ([sum | sum of squares]) written by (block bx of sample n) at row n*bps + bx:
// out[g*C + c] / out[groups*C + g*C + c] = sums over the samples n = g (mod groups) and their bps blocks, fixed order.
void launch_partials_reduce_grouped(const float* part, int bps, int nsamples, int groups, int C, float* out, hipStream_t stream);

// Per-device facts.  The library keeps no other mutable global state (header contract: re-entrant across threads and devices), and these
// two are caches keyed by the CURRENT device id (hipGetDevice), so a single process driving several GPUs gets each device's own answer.
int device_cus();                                              // compute units of the current device (256 on MI355X)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for `func`, once per (function, device); MVS_OK or a negative error with set_error()
int ensure_dynamic_lds(const void* func, int bytes, const char* who);

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace mvs

#define MVS_REQUIRE(cond, ...)               \
    do {                                     \
        if (!(cond)) {                       \
            mvs::set_error(__VA_ARGS__);     \
            return MVS_EINVAL;               \
        }                                    \
    } while (0)

#define MVS_STREAM(s) reinterpret_cast<hipStream_t>(s)
