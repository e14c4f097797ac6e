// Error channel + ABI version of libmvs_hip.so.
#include "common.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace mvs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
constexpr int MAX_DEVICES = 64;
std::atomic<int> g_cus[MAX_DEVICES];                            // 0 = not asked yet
struct LdsKey { const void* func; int dev; int bytes; };
std::mutex g_lds_mutex;
std::vector<LdsKey> g_lds_done;
}  // namespace

int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
    int n = g_cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cus[dev].store(n, std::memory_order_relaxed);       // idempotent: a race only repeats the query
    }
    return n;
}

int ensure_dynamic_lds(const void* func, int bytes, const char* who) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    for (const LdsKey& k : g_lds_done)
        if (k.func == func && k.dev == dev && k.bytes >= bytes) return MVS_OK;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %d bytes of dynamic LDS (this library needs gfx950's 160 KB per CU): %s", who, bytes, hipGetErrorString(e));
        return -(1000 + (int)e);
    }
    g_lds_done.push_back(LdsKey{func, dev, bytes});
    return MVS_OK;
}

static __global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ out) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    float s = 0.0f;
    for (int p = lane; p < nparts; p += 64) s += part[(size_t)p * n + j];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[j] = s;
}

static __global__ __launch_bounds__(256) void partials_reduce_grouped_kernel(const float* __restrict__ part, int bps, int nsamples, int groups,
                                                                             int C, float* __restrict__ out) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;          // output (g, j), j in [0, 2C)
    if (o >= groups * 2 * C) return;
    const int g = o / (2 * C), j = o % (2 * C);
    const int per = (nsamples / groups) * bps;              // partial rows of this group
    float s = 0.0f;
    for (int q = lane; q < per; q += 64) {
        const int n = g + (q / bps) * groups, bx = q % bps;
        s += part[((size_t)n * bps + bx) * 2 * C + j];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[(j >= C ? groups * C : 0) + g * C + (j % C)] = s;
}

void launch_partials_reduce_grouped(const float* part, int bps, int nsamples, int groups, int C, float* out, hipStream_t stream) {
    if (groups == 1 && nsamples == 1) return launch_partials_reduce(part, bps, 2 * C, out, stream);
    hipLaunchKernelGGL(partials_reduce_grouped_kernel, dim3((groups * 2 * C + 3) / 4), dim3(256), 0, stream, part, bps, nsamples, groups, C, out);
}

void launch_partials_reduce(const float* part, int nparts, int n, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, part, nparts, n, out);
}
}  // namespace mvs

// out[j] = sum_p part[p*n + j] in a fixed order (the second stage of every split reduction of the library, exposed for callers that
// produce their own partial rows: the FPN weight gradients' split-K partial matrices)
extern "C" int mvs_partials_reduce(const float* part, int nparts, int n, float* out, mvs_stream_t stream) {
    MVS_REQUIRE(part && out && nparts >= 1 && n >= 1, "mvs_partials_reduce: bad arguments");
    mvs::launch_partials_reduce(part, nparts, n, out, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_partials_reduce");
}

extern "C" int mvs_version(void) { return MVS_ABI_VERSION; }
extern "C" const char* mvs_last_error(void) { return mvs::g_err; }
