// Torch-free reproducer attempt for the round-3 multi-stream nondeterminism: do packed-fp32 vector instructions
// (v_pk_fma_f32 / v_pk_add_f32) or fp32 MFMAs give wrong results when wavefronts issuing back-to-back v_mfma_f32_16x16x32_bf16 are
// resident on the same SIMDs?
//
//   victim    : a miniature of conv3d_wino.hip's inner loop - an LDS-staged 4x4 patch per lane, the Winograd input transform B^T d B with
//               packed fp32 adds / fmas, the 16 results fed straight into v_mfma_f32_16x16x4_f32 (A = small integer weights from LDS),
//               then the output transform with packed ops on the accumulators.  Every value is a small INTEGER held in fp32, so the
//               answer is exact, order-independent and computed on the host with integer arithmetic.
//   aggressor : wavefronts that only issue dependent-free chains of v_mfma_f32_16x16x32_bf16 (and check their own known answer).
//
//   modes     : 1  victim alone, one stream
//               2  victim on stream A, aggressor kernel back to back on stream B
//               3  both roles in ONE block of 8 wavefronts (4 + 4: every SIMD holds one of each for the whole run)
// Each mode runs REPS launches and bit-compares every output word with the host answer.  Built twice by the Makefile: with the compiler
// free to use packed fp32 (pk_mfma_race) and with -packed-fp32-ops (pk_mfma_race_nopk: the transforms become scalar v_add / v_fma).
//
// Build: make -C tools/probe      Run (GPU box): tools/probe/pk_mfma_race [reps]     Output: one line per mode, mismatching words per launch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NCH = 4;                 // input chunks resident in LDS (each: 4 k-slices x 64 tiles x 16 patch values)
constexpr int CHUNK = 4 * 64 * 16;     // floats per chunk
constexpr int WCH = 16 * 4 * 16;       // weights per chunk: [xi][k][m]
constexpr int OUT_PER_WAVE = 64 * 4 * 4;   // per lane: 4 accumulator rows x (2 x 2 output tile)

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    float m1;
    asm("s_mov_b32 %0, -1.0" : "=s"(m1));
    return pk_fma(b, f32x2{m1, m1}, a);
}

// ---- the victim role of one wavefront: `iters` passes over the NCH chunks, accumulating Z; out[lane][r][a][b] ----
__device__ __forceinline__ void victim_wave(const float* s_in, const float* s_w, int wave, int lane, int iters, float* __restrict__ out) {
    const int i16 = lane & 15, kk = lane >> 4;
    f32x4 Z[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) Z[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            const float* p = s_in + c * CHUNK + (kk * 64 + wave * 16 + i16) * 16;
            // the patch rows as even / odd column pairs: E[q] = (d[q][0], d[q][2]), O[q] = (d[q][1], d[q][3])
            f32x2 E[4], O[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                E[q] = f32x2{p[q * 4 + 0], p[q * 4 + 2]};
                O[q] = f32x2{p[q * 4 + 1], p[q * 4 + 3]};
            }
            float X[16];
            {
                f32x2 T[4], U2[4];
                T[0] = pk_sub(E[0], E[2]);  U2[0] = pk_sub(O[0], O[2]);
                T[1] = E[1] + E[2];         U2[1] = O[1] + O[2];
                T[2] = pk_sub(E[2], E[1]);  U2[2] = pk_sub(O[2], O[1]);
                T[3] = pk_sub(E[1], E[3]);  U2[3] = pk_sub(O[1], O[3]);
#ifdef VICTIM_VGPR_CONST
                // hypothesis test 3: the (1, -1) multiplier of the packed fma below normally sits in an SGPR PAIR whose HIGH half is selected
                // by op_sel_hi (v_pk_fma_f32 ..., s[0:1], ... op_sel_hi:[0,1,1]); here it is forced into vector registers
                float c_one, c_minus;
                asm volatile("v_mov_b32 %0, 1.0\n v_mov_b32 %1, -1.0" : "=v"(c_one), "=v"(c_minus));
                const f32x2 pm = {c_one, c_minus};
#else
                const f32x2 pm = {1.0f, -1.0f};
#endif
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x2 m = pk_fma(f32x2{U2[a].x, U2[a].x}, pm, f32x2{T[a].y, T[a].y});
                    X[a * 4 + 0] = T[a].x - T[a].y;
                    X[a * 4 + 1] = m.x;
                    X[a * 4 + 2] = m.y;
                    X[a * 4 + 3] = U2[a].x - U2[a].y;
                }
            }
            const float* u = s_w + c * WCH + kk * 16 + i16;
#ifdef VICTIM_RAW_FENCE
            // hypothesis test 2: a read-after-write window between the packed ops that produce X[] and the MFMAs that read it - all 16
            // results are pinned in registers, the wave idles 32 wait states, then the MFMAs start
            asm volatile("s_nop 15\n s_nop 15"
                         : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]), "+v"(X[8]), "+v"(X[9]),
                           "+v"(X[10]), "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15]));
#endif
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) Z[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[xi * 64], X[xi], Z[xi], 0, 0, 0);
#ifdef VICTIM_FENCE
            // hypothesis test: the next chunk's packed ops OVERWRITE registers (X[]) that the MFMAs above read as their B operand; if the
            // matrix unit fetches operands late while another kernel's bf16 MFMAs hold it, idling the wave here must make the errors vanish
            asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
        }
    }
    // output transform A^T Z A with packed ops on the accumulators (rows r = 0..3 of the lane)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f32x2 s0[2], s1[2];                                   // s[a][bcol] packed over bcol pairs (0,1) and (2,3)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 z0 = {Z[0 + 2 * h][r], Z[1 + 2 * h][r]}, z1 = {Z[4 + 2 * h][r], Z[5 + 2 * h][r]}, z2 = {Z[8 + 2 * h][r], Z[9 + 2 * h][r]},
                        z3 = {Z[12 + 2 * h][r], Z[13 + 2 * h][r]};
            s0[h] = (z0 + z1) + z2;
            s1[h] = pk_sub(pk_sub(z1, z2), z3);
        }
        float* o = out + (lane * 4 + r) * 4;
        o[0] = s0[0].x + s0[0].y + s0[1].x;
        o[1] = s0[0].y - s0[1].x - s0[1].y;
        o[2] = s1[0].x + s1[0].y + s1[1].x;
        o[3] = s1[0].y - s1[1].x - s1[1].y;
    }
}

// ---- the aggressor role of one wavefront: 8 independent chains of bf16 MFMAs on all-ones operands; returns the sum of one accumulator ----
__device__ __forceinline__ float aggressor_wave(int iters) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)1.0f; }
    f32x4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    return s;                                                  // = 8 * 4 * 32 * iters
}

// ---- a victim with NO matrix instruction: packed fp32 FMAs over the LDS-staged patches only; out[lane][0..15] (exact small integers) ----
__device__ __forceinline__ void pkonly_wave(const float* s_in, int wave, int lane, int iters, float* __restrict__ out) {
    const int i16 = lane & 15, kk = lane >> 4;
    f32x2 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x2{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            const float* p = s_in + c * CHUNK + (kk * 64 + wave * 16 + i16) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 E = {p[q * 4 + 0], p[q * 4 + 2]}, O = {p[q * 4 + 1], p[q * 4 + 3]};
                acc[2 * q] = pk_fma(E, O, acc[2 * q]);                       // += (d0*d1, d2*d3)
                acc[2 * q + 1] = pk_sub(acc[2 * q + 1], pk_sub(E, O));        // -= (d0-d1, d2-d3)
            }
        }
    }
    float* o = out + lane * 16;
#pragma unroll
    for (int q = 0; q < 8; ++q) { o[2 * q] = acc[q].x; o[2 * q + 1] = acc[q].y; }
}

// fp32-MFMA and VALU-only aggressors (does it take the bf16 matrix instruction, any matrix instruction, or just a busy neighbour?)
__device__ __forceinline__ float aggressor_f32mfma_wave(int iters) {
    f32x4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    return s;                                                  // = 8 * 4 * 4 * iters
}
__device__ __forceinline__ float aggressor_valu_wave(int iters, float one) {
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = fmaf(one, one, c[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i];
    return s;                                                  // = 8 * iters
}

__device__ __forceinline__ void stage(float* s_in, float* s_w, const float* in, const float* w, int tid, int nthreads) {
    for (int i = tid; i < NCH * CHUNK; i += nthreads) s_in[i] = in[i];
    for (int i = tid; i < NCH * WCH; i += nthreads) s_w[i] = w[i];
}

__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ in, const float* __restrict__ w, int iters, float* __restrict__ out) {
    extern __shared__ float smem[];
    float* s_in = smem;
    float* s_w = smem + NCH * CHUNK;
    stage(s_in, s_w, in, w, threadIdx.x, 256);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    victim_wave(s_in, s_w, wave, lane, iters, out + ((size_t)blockIdx.x * 4 + wave) * OUT_PER_WAVE);
}

__global__ __launch_bounds__(256) void pkonly_kernel(const float* __restrict__ in, const float* __restrict__ w, int iters, float* __restrict__ out) {
    extern __shared__ float smem[];
    float* s_in = smem;
    float* s_w = smem + NCH * CHUNK;
    stage(s_in, s_w, in, w, threadIdx.x, 256);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    pkonly_wave(s_in, wave, lane, iters, out + ((size_t)blockIdx.x * 4 + wave) * OUT_PER_WAVE);
}

// kind 0: bf16 MFMA, 1: fp32 MFMA, 2: VALU only
__global__ __launch_bounds__(256) void aggressor_kernel(int kind, int iters, float one, float* __restrict__ out) {
    const float s = kind == 0 ? aggressor_wave(iters) : kind == 1 ? aggressor_f32mfma_wave(iters) : aggressor_valu_wave(iters, one);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

// 8 wavefronts: wave w and wave w + 4 share a SIMD; waves 0-3 victim, 4-7 aggressor
__global__ __launch_bounds__(512) void mixed_kernel(const float* __restrict__ in, const float* __restrict__ w, int iters, int agg_iters,
                                                    float* __restrict__ out, float* __restrict__ agg_out) {
    extern __shared__ float smem[];
    float* s_in = smem;
    float* s_w = smem + NCH * CHUNK;
    stage(s_in, s_w, in, w, threadIdx.x, 512);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        victim_wave(s_in, s_w, wave, lane, iters, out + ((size_t)blockIdx.x * 4 + wave) * OUT_PER_WAVE);
    } else {
        const float s = aggressor_wave(agg_iters);
        if (lane == 0) agg_out[blockIdx.x * 4 + wave - 4] = s;
    }
}

// ---- host reference in integers ----
static void reference(const std::vector<float>& in, const std::vector<float>& w, int iters, std::vector<float>& out /* one block */) {
    out.assign(4 * OUT_PER_WAVE, 0.f);
    static const int BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
    static const int AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
    for (int wave = 0; wave < 4; ++wave) {
        // Z[xi][m][n] = iters * sum_c sum_k W[c][xi][k][m] * X[c][k][n][xi]
        std::vector<long long> Z(16 * 16 * 16, 0);
        for (int c = 0; c < NCH; ++c)
            for (int k = 0; k < 4; ++k)
                for (int n = 0; n < 16; ++n) {
                    const float* p = &in[c * CHUNK + (k * 64 + wave * 16 + n) * 16];
                    long long d[4][4], t[4][4], X[4][4];
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) d[i][j] = (long long)p[i * 4 + j];
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { t[i][j] = 0; for (int q = 0; q < 4; ++q) t[i][j] += BT[i][q] * d[q][j]; }
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { X[i][j] = 0; for (int q = 0; q < 4; ++q) X[i][j] += t[i][q] * BT[j][q]; }
                    for (int xi = 0; xi < 16; ++xi)
                        for (int m = 0; m < 16; ++m) Z[(xi * 16 + m) * 16 + n] += (long long)w[c * WCH + xi * 64 + k * 16 + m] * X[xi / 4][xi % 4];
                }
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int n = lane & 15, m = (lane >> 4) * 4 + r;
                long long z[4][4], t[2][4];
                for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) z[a][b] = Z[((a * 4 + b) * 16 + m) * 16 + n] * iters;
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) { t[a][b] = 0; for (int q = 0; q < 4; ++q) t[a][b] += AT[a][q] * z[q][b]; }
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        long long v = 0;
                        for (int q = 0; q < 4; ++q) v += t[a][q] * AT[b][q];
                        out[wave * OUT_PER_WAVE + (lane * 4 + r) * 4 + a * 2 + b] = (float)v;
                    }
            }
    }
}

static void reference_pkonly(const std::vector<float>& in, int iters, std::vector<float>& out /* one block */) {
    out.assign(4 * OUT_PER_WAVE, 0.f);
    for (int wave = 0; wave < 4; ++wave)
        for (int lane = 0; lane < 64; ++lane) {
            long long acc[16] = {0};
            const int i16 = lane & 15, kk = lane >> 4;
            for (int c = 0; c < NCH; ++c) {
                const float* p = &in[c * CHUNK + (kk * 64 + wave * 16 + i16) * 16];
                for (int q = 0; q < 4; ++q) {
                    const long long d0 = (long long)p[q * 4], d1 = (long long)p[q * 4 + 1], d2 = (long long)p[q * 4 + 2], d3 = (long long)p[q * 4 + 3];
                    acc[4 * q + 0] += d0 * d1;
                    acc[4 * q + 1] += d2 * d3;
                    acc[4 * q + 2] -= d0 - d1;
                    acc[4 * q + 3] -= d2 - d3;
                }
            }
            for (int j = 0; j < 16; ++j) out[wave * OUT_PER_WAVE + lane * 16 + j] = (float)(acc[j] * iters);
        }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10000;
    const int iters = 16, agg_iters = 4096, blocks = 512;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; victim: %d blocks x %d passes over %d LDS chunks; %d launches per mode\n", prop.gcnArchName, prop.multiProcessorCount, blocks, iters, NCH, reps);
    std::vector<float> in(NCH * CHUNK), w(NCH * WCH), ref;
    unsigned s = 12345u;
    auto rnd = [&](int lo, int hi) { s = s * 1664525u + 1013904223u; return (float)(lo + (int)((s >> 10) % (unsigned)(hi - lo + 1))); };
    for (float& v : in) v = rnd(-4, 4);
    for (float& v : w) v = rnd(-2, 2);
    reference(in, w, iters, ref);
    float maxabs = 0;
    for (float v : ref) maxabs = fmaxf(maxabs, fabsf(v));
    printf("largest exact output %.0f (< 2^24: every intermediate is an exactly representable integer)\n", maxabs);
    float *d_in, *d_w, *d_out, *d_agg;
    const size_t out_floats = (size_t)blocks * 4 * OUT_PER_WAVE;
    CK(hipMalloc(&d_in, in.size() * 4));
    CK(hipMalloc(&d_w, w.size() * 4));
    CK(hipMalloc(&d_out, out_floats * 4));
    CK(hipMalloc(&d_agg, 4096 * 4 * 4));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    const int lds = (NCH * CHUNK + NCH * WCH) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(victim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mixed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    std::vector<float> host(out_floats), hagg(4096 * 4), ref_pk;
    reference_pkonly(in, iters, ref_pk);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pkonly_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    // victim: 0 = Winograd miniature (packed fp32 + fp32 MFMA), 1 = packed fp32 only
    // neighbour on stream B: -1 none, 0 bf16-MFMA kernel, 1 fp32-MFMA kernel, 2 VALU-only kernel, 3 the victim kernel itself (second copy, own output),
    //                        9 = none, but both roles inside one 8-wavefront block (mixed_kernel)
    struct Cfg { int victim, neighbour; const char* name; };
    const Cfg cfgs[] = {
        {0, -1, "wino victim alone, one stream"},
        {0, 0, "wino victim on stream A + bf16-MFMA kernel on stream B"},
        {0, 1, "wino victim on stream A + fp32-MFMA kernel on stream B"},
        {0, 2, "wino victim on stream A + VALU-only kernel on stream B"},
        {0, 3, "wino victim on stream A + a second copy of itself on stream B"},
        {0, 9, "wino victim and bf16-MFMA wavefronts in ONE block (same SIMDs, one kernel)"},
        {1, -1, "pk-only victim alone, one stream"},
        {1, 0, "pk-only victim on stream A + bf16-MFMA kernel on stream B"},
        {1, 2, "pk-only victim on stream A + VALU-only kernel on stream B"},
    };
    float* d_out2;
    CK(hipMalloc(&d_out2, out_floats * 4));
    for (const Cfg& cf : cfgs) {
        const std::vector<float>& want_out = cf.victim == 0 ? ref : ref_pk;
        long long bad_launches = 0, bad_words = 0, bad_agg = 0, nan_words = 0;
        int printed = 0;
        const int check_every = 50;                           // outputs are overwritten in place; check a launch in every 50 + the last
        for (int rep = 0; rep < reps; ++rep) {
            if (cf.neighbour >= 0 && cf.neighbour <= 2)
                hipLaunchKernelGGL(aggressor_kernel, dim3(1024), dim3(256), 0, sb, cf.neighbour, cf.neighbour == 2 ? 8 * agg_iters : agg_iters, 1.0f, d_agg);
            if (cf.neighbour == 3) hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(256), lds, sb, d_in, d_w, iters, d_out2);
            if (cf.neighbour == 9)
                hipLaunchKernelGGL(mixed_kernel, dim3(blocks), dim3(512), lds, sa, d_in, d_w, iters, 4 * iters * NCH, d_out, d_agg);
            else if (cf.victim == 0)
                hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(256), lds, sa, d_in, d_w, iters, d_out);
            else
                hipLaunchKernelGGL(pkonly_kernel, dim3(blocks), dim3(256), lds, sa, d_in, d_w, iters, d_out);
            CK(hipGetLastError());
            if (rep % check_every == check_every - 1 || rep == reps - 1) {
                CK(hipStreamSynchronize(sa));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(host.data(), d_out, out_floats * 4, hipMemcpyDeviceToHost));
                long long bw = 0;
                for (int b = 0; b < blocks; ++b)
                    bw += memcmp(&host[(size_t)b * 4 * OUT_PER_WAVE], want_out.data(), 4 * OUT_PER_WAVE * 4) ? 1 : 0;
                if (bw) {
                    ++bad_launches;
                    for (size_t i = 0; i < out_floats; ++i) {
                        const float wv = want_out[i % (4 * OUT_PER_WAVE)];
                        if (host[i] != wv) {
                            ++bad_words;
                            nan_words += host[i] != host[i];
                            if (printed < 6) {
                                const size_t j = i % (4 * OUT_PER_WAVE);
                                printf("    launch %d: block %zu wave %zu lane %zu word %zu: got %.9g (0x%08x), want %.9g\n", rep, i / (4 * OUT_PER_WAVE), j / OUT_PER_WAVE,
                                       (j % OUT_PER_WAVE) / 16, j % 16, host[i], *reinterpret_cast<const unsigned*>(&host[i]), wv);
                                ++printed;
                            }
                        }
                    }
                }
                if (cf.neighbour == 0 || cf.neighbour == 1 || cf.neighbour == 2 || cf.neighbour == 9) {
                    const int nagg = cf.neighbour == 9 ? blocks * 4 : 1024 * 4;
                    const float want = cf.neighbour == 9 ? 8.f * 4 * 32 * (4 * iters * NCH) : cf.neighbour == 0 ? 8.f * 4 * 32 * agg_iters
                                     : cf.neighbour == 1 ? 8.f * 4 * 4 * agg_iters : 8.f * 8 * agg_iters;
                    CK(hipMemcpy(hagg.data(), d_agg, nagg * 4, hipMemcpyDeviceToHost));
                    for (int i = 0; i < nagg; ++i) bad_agg += hagg[i] != want;
                }
                CK(hipMemsetAsync(d_out, 0xFF, out_floats * 4, sa));
                CK(hipStreamSynchronize(sa));
            }
        }
        CK(hipDeviceSynchronize());
        printf("%-80s: %d launches, %d checked: %lld with a wrong block, %lld wrong words (%lld of them NaN = never written), %lld wrong neighbour sums\n", cf.name, reps,
               (reps + check_every - 1) / check_every, bad_launches, bad_words, nan_words, bad_agg);
    }
    return 0;
}
