// Can vector-ALU work ride under bf16 MFMAs?  (1) inside ONE wavefront: a loop of {1 v_mfma_f32_16x16x32_bf16 + n independent v_fma_f32},
// n = 0..6, one wavefront per SIMD; (2) the same with TWO such wavefronts per SIMD; (3) an MFMA-only wavefront next to a VALU-only wavefront
// on the same SIMD (r03 ubench: those serialize).  Prints clocks per loop iteration of the slowest wavefront.
//     hipcc -O3 --offload-arch=gfx950 coissue_probe.hip -o coissue_probe && ./coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int CHAINS>
__global__ __launch_bounds__(512) void mix(float* out, unsigned long long* cyc, int iters, float seed) {
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + i); b[i] = (__bf16)(0.5f + i); }
    float v[6] = {seed, seed + 1, seed + 2, seed + 3, seed + 4, seed + 5};
    const float k1 = seed * 0.999f, k2 = seed * 1e-3f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u % CHAINS], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(k1), "v"(k2));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    for (int j = 0; j < 6; ++j) r += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}

template <int NV, int CHAINS>
void run(float* out, unsigned long long* cyc, int threads) {
    const int iters = 2000;
    hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL((mix<NV, CHAINS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %d chains, %d v_fma per MFMA, %d wavefront(s)/SIMD: %6.2f clocks per {MFMA + %d v_fma}\n", CHAINS, NV, threads / 256, (double)c / (iters * 8.0), NV);
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    printf("independent MFMA chains (4 accumulators):\n");
    run<0, 4>(out, cyc, 256); run<1, 4>(out, cyc, 256); run<2, 4>(out, cyc, 256); run<3, 4>(out, cyc, 256); run<4, 4>(out, cyc, 256); run<6, 4>(out, cyc, 256);
    printf("one dependent MFMA chain:\n");
    run<0, 1>(out, cyc, 256); run<2, 1>(out, cyc, 256); run<3, 1>(out, cyc, 256); run<4, 1>(out, cyc, 256);
    printf("two wavefronts per SIMD, each running the mix:\n");
    run<0, 4>(out, cyc, 512); run<2, 4>(out, cyc, 512); run<3, 4>(out, cyc, 512); run<6, 4>(out, cyc, 512);
    run<0, 1>(out, cyc, 512); run<3, 1>(out, cyc, 512);
    return 0;
}
