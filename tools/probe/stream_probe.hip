// What bounds the HBM rate of the "x3" kernels' staging phase (conv1 / tail / deconv sit at 2.2-2.6 TB/s while a float4 copy runs at 6.5)?
// Blocks of 256 threads alternate a BURST of global loads (bytes per thread fixed at 128: 32 dwords | 16 x 2 | 8 x 4, the dword form laid
// out like the staging: a wavefront-load = 256 contiguous bytes of one of 8 channel planes) with a compute phase of `delay` dependent FMAs,
// like a pass of x3_conv_kernel; blocks per CU through the dynamic LDS size.  Prints GB/s per variant.
//     hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe && ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int WIDTH>
struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef f32x2 T; };
template <> struct Vec<4> { typedef f32x4 T; };
__device__ inline float hsum(float v) { return v; }
__device__ inline float hsum(f32x2 v) { return v[0] + v[1]; }
__device__ inline float hsum(f32x4 v) { return v[0] + v[1] + v[2] + v[3]; }

// plane = floats per channel plane (8 planes); a pass of a block reads [8 planes][1024 floats]: 4 / 2 / 1 loads per thread and plane
template <int WIDTH>
__global__ __launch_bounds__(256) void burst(const float* __restrict__ x, float* __restrict__ out, size_t plane, int passes, int delay) {
    extern __shared__ float pad[];
    typedef typename Vec<WIDTH>::T V;
    constexpr int NL = 32 / WIDTH;
    const int tid = threadIdx.x;
    float acc = 0.0f;
    for (int pass = 0; pass < passes; ++pass) {
        const size_t base = ((size_t)blockIdx.x * passes + pass) * 1024;
        V v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = i % 8, j = i / 8;
            const float* p = x + (size_t)e * plane + base + (size_t)j * 256 * WIDTH + (size_t)tid * WIDTH;
            v[i] = *reinterpret_cast<const V*>(p);
        }
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < NL; ++i) s += hsum(v[i]);
        for (int d = 0; d < delay; ++d) s = fmaf(s, 1.0000001f, 1e-9f);
        acc += s;
        __syncthreads();
    }
    if (acc == 123.456f) out[blockIdx.x * 256 + tid] = acc + pad[tid];
}

template <int WIDTH>
void run(const float* x, float* out, size_t plane, int blocks_per_cu, int delay, int passes) {
    const int lds = blocks_per_cu == 2 ? 70 * 1024 : blocks_per_cu == 3 ? 50 * 1024 : blocks_per_cu == 4 ? 36 * 1024 : 18 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&burst<WIDTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const size_t per_block = (size_t)passes * 1024;
    const int blocks = (int)(plane / per_block);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(burst<WIDTH>, dim3(blocks), dim3(256), lds, 0, x, out, plane, passes, delay);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r > 0 && ms < best) best = ms;
    }
    const double bytes = (double)blocks * per_block * 8 * 4;
    printf("  width %d dwords x %2d loads, %d blocks/CU, delay %5d, %d passes/block: %.3f ms  %.0f GB/s\n", WIDTH, 32 / WIDTH, blocks_per_cu, delay, passes,
           best, bytes / best / 1e6);
}

int main() {
    const size_t plane = (size_t)16 * 1024 * 1024;        // 8 planes x 64 MB = 512 MB (> the 256 MB Infinity Cache)
    float *x, *out;
    hipMalloc(&x, plane * 8 * 4);
    hipMalloc(&out, 64 << 20);
    hipMemset(x, 0, plane * 8 * 4);
    for (int passes : {4, 1})
        for (int delay : {0, 400, 1600})
            for (int bpc : {2, 3, 4, 8}) {
                run<1>(x, out, plane, bpc, delay, passes);
                run<2>(x, out, plane, bpc, delay, passes);
                run<4>(x, out, plane, bpc, delay, passes);
            }
    return 0;
}
