// Stand-alone gfx950 micro-benchmarks behind the design decisions of round 3 (no torch, no libmvs_hip):
//   1. issue cost of the vector instructions the sweeps are made of (cycles per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD),
//   2. whether an fp32 MFMA wave and an fp32 VALU wave on the SAME SIMD overlap (separate pipes) or add (one pipe),
//   3. LDS-DMA (buffer_load_dwordx4 ... lds) against register-staged loads on an L2-resident linear stream.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip      Run: ./ubench  (prints a table; tools/README in profiles/)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

// ---------------------------------------------------------------------------------------------------------
// 1. instruction issue cost
// ---------------------------------------------------------------------------------------------------------
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
// each OP(i) is one instruction on its own accumulator a{i}; 8 independent chains, 8 repeats = 64 instructions per asm block
#define BLOCK64(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define OP_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_MUL(i) "v_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %8, %" #i "\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_DPPADD(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP_DPPMIR(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %8, %" #i "\n"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 5, %8\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define OP_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define OP_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define OP_MAX(i) "v_max_f32 %" #i ", %8, %" #i "\n"
#define OP_CMP(i) "v_cmp_gt_u32 vcc, %8, %" #i "\n"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_CNDS(i) "v_cndmask_b32_e64 %" #i ", %8, %" #i ", s[10:11]\n"
#define OP_CMPCND(i) "v_cmp_gt_u32 vcc, %8, %" #i "\n v_cndmask_b32 %" #i ", %9, %" #i ", vcc\n"
#define OP_CMPSCND(i) "v_cmp_gt_u32_e64 s[10:11], %8, %" #i "\n v_cndmask_b32_e64 %" #i ", %9, %" #i ", s[10:11]\n"
#define OP_AND(i) "v_and_b32 %" #i ", %8, %" #i "\n"
#define OP_ASHR(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define OP_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 5, %8\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n"
#define OP_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define OP_MINU(i) "v_min_u32 %" #i ", %8, %" #i "\n"
#define OP_FMAS(i) "v_fma_f32 %" #i ", s12, %9, %" #i "\n"
#define OP_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"

enum { K_FMA, K_MUL, K_ADD, K_MOV, K_DPPADD, K_DPPMIR, K_CNDMASK, K_MULLO, K_LSHLOR, K_ADD3, K_RCP, K_RSQ, K_EXP, K_FLOOR, K_CVTI, K_MAX, K_CMP,
       K_MAD24, K_PKFMA, K_PKMUL, K_PKADD, K_CNDS, K_CMPCND, K_CMPSCND, K_AND, K_ASHR, K_BFI, K_LSHLADD, K_MED3, K_FRACT, K_SUBU, K_MINU, K_FMAS, K_FMAC, K_NOPS };
static const char* kNames[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_add_f32_dpp quad_perm", "v_add_f32_dpp row_mirror", "v_cndmask_b32",
                               "v_mul_lo_u32", "v_lshl_or_b32", "v_add3_u32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_floor_f32", "v_cvt_i32_f32",
                               "v_max_f32", "v_cmp_gt_u32", "v_mad_u32_u24", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cndmask_b32_e64 (sgpr pair)",
                               "v_cmp+v_cndmask vcc (2 instr)", "v_cmp_e64+v_cndmask_e64 (2 instr)", "v_and_b32", "v_ashrrev_i32", "v_bfi_b32", "v_lshl_add_u32", "v_med3_f32",
                               "v_fract_f32", "v_sub_u32", "v_min_u32", "v_fma_f32 (sgpr operand)", "v_fmac_f32"};

template <int K>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, unsigned long long* cycles, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float x = 1.0000001f, y = 1e-9f;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 p0 = {seed, seed}, p1 = p0 + 1, p2 = p0 + 2, p3 = p0 + 3, p4 = p0 + 4, p5 = p0 + 5, p6 = p0 + 6, p7 = p0 + 7, px = {x, x}, py = {y, y};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define RUN(OP) asm volatile(BLOCK64(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc")
        if constexpr (K == K_FMA) RUN(OP_FMA);
        else if constexpr (K == K_MUL) RUN(OP_MUL);
        else if constexpr (K == K_ADD) RUN(OP_ADD);
        else if constexpr (K == K_MOV) RUN(OP_MOV);
        else if constexpr (K == K_DPPADD) RUN(OP_DPPADD);
        else if constexpr (K == K_DPPMIR) RUN(OP_DPPMIR);
        else if constexpr (K == K_CNDMASK) RUN(OP_CNDMASK);
        else if constexpr (K == K_MULLO) RUN(OP_MULLO);
        else if constexpr (K == K_LSHLOR) RUN(OP_LSHLOR);
        else if constexpr (K == K_ADD3) RUN(OP_ADD3);
        else if constexpr (K == K_RCP) RUN(OP_RCP);
        else if constexpr (K == K_RSQ) RUN(OP_RSQ);
        else if constexpr (K == K_EXP) RUN(OP_EXP);
        else if constexpr (K == K_FLOOR) RUN(OP_FLOOR);
        else if constexpr (K == K_CVTI) RUN(OP_CVTI);
        else if constexpr (K == K_MAX) RUN(OP_MAX);
        else if constexpr (K == K_CMP) RUN(OP_CMP);
        else if constexpr (K == K_MAD24) RUN(OP_MAD24);
#define RUNS(OP) asm volatile(BLOCK64(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc", "s10", "s11", "s12")
        else if constexpr (K == K_CNDS) RUNS(OP_CNDS);
        else if constexpr (K == K_CMPCND) RUNS(OP_CMPCND);
        else if constexpr (K == K_CMPSCND) RUNS(OP_CMPSCND);
        else if constexpr (K == K_AND) RUNS(OP_AND);
        else if constexpr (K == K_ASHR) RUNS(OP_ASHR);
        else if constexpr (K == K_BFI) RUNS(OP_BFI);
        else if constexpr (K == K_LSHLADD) RUNS(OP_LSHLADD);
        else if constexpr (K == K_MED3) RUNS(OP_MED3);
        else if constexpr (K == K_FRACT) RUNS(OP_FRACT);
        else if constexpr (K == K_SUBU) RUNS(OP_SUBU);
        else if constexpr (K == K_MINU) RUNS(OP_MINU);
        else if constexpr (K == K_FMAS) RUNS(OP_FMAS);
        else if constexpr (K == K_FMAC) RUNS(OP_FMAC);
#undef RUNS
#undef RUN
        else {
#define PK(OPS) asm volatile(OPS : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px), "v"(py))
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_PKMUL(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_PKADD(i) "v_pk_add_f32 %" #i ", %8, %" #i "\n"
            if constexpr (K == K_PKFMA) PK(BLOCK64(OP_PKFMA));
            else if constexpr (K == K_PKMUL) PK(BLOCK64(OP_PKMUL));
            else PK(BLOCK64(OP_PKADD));
#undef PK
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

// ---------------------------------------------------------------------------------------------------------
// 2. fp32 MFMA next to fp32 VALU on one SIMD.  512-thread blocks, one per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 run
// v_fma chains.  mode bit 0: MFMA waves work, bit 1: VALU waves work (idle waves leave at once).
// ---------------------------------------------------------------------------------------------------------
template <int KIND>   // 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_32x32x2_f32, 2: v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(512) void coissue_kernel(float* out, unsigned long long* cycles, int iters, int mode, float seed) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    if ((mf && !(mode & 1)) || (!mf && !(mode & 2))) return;
    float r = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mf) {
        if constexpr (KIND == 0) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            float a = seed, b = seed * 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                }
            }
            r = c0[0] + c1[1] + c2[2] + c3[3];
        } else if constexpr (KIND == 1) {
            using f32x16 = __attribute__((ext_vector_type(16))) float;
            f32x16 c0 = {}, c1 = {};
            float a = seed, b = seed * 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                }
            }
            r = c0[0] + c1[1];
        } else {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            bf16x8 a, b;
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3f00 + i); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
                }
            }
            r = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
        float x = 1.0000001f, y = 1e-9f;
        for (int it = 0; it < iters; ++it)
            asm volatile(BLOCK64(OP_FMA) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cycles[wave >> 2] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---------------------------------------------------------------------------------------------------------
// 3. L2/MALL-resident linear stream: register loads (buffer_load_dwordx4) vs LDS-DMA (buffer_load_dwordx4 ... lds) + ds_read_b128 back
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
using rsrc_t = __amdgpu_buffer_rsrc_t;

template <bool DMA>
__global__ __launch_bounds__(256) void stream_kernel(const float* __restrict__ src, float* out, unsigned bytes_per_block, int rounds) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + (size_t)blockIdx.x * (bytes_per_block / 4), 0, bytes_per_block, 0x00020000);
    float* my = lds + wave * 4096;                      // 16 KB per wave: 16 pieces of 1 KB
    f32x4 acc = {0, 0, 0, 0};
    const unsigned per_wave = bytes_per_block / 4;      // bytes streamed by this wave per round
    for (int rd = 0; rd < rounds; ++rd) {
        for (unsigned off = 0; off < per_wave; off += 16 * 1024) {
            const unsigned base = wave * per_wave + off + lane * 16;
            if constexpr (DMA) {
#pragma unroll
                for (int p = 0; p < 16; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(my + p * 256), 16, base + p * 1024, 0, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int p = 0; p < 16; ++p) acc += *reinterpret_cast<const f32x4*>(my + p * 256 + lane * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                f32x4 v[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) v[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, base + p * 1024, 0, 0));
#pragma unroll
                for (int p = 0; p < 16; ++p) acc += v[p];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int K>
static void run_issue(float* out, unsigned long long* cyc, int threads) {
    const int iters = 2000;
    hipLaunchKernelGGL((issue_kernel<K>), dim3(256), dim3(threads), 0, 0, out, cyc, 10, 1.0f);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((issue_kernel<K>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const int wps = threads / 256;
    // wave 0 of a block is the oldest wave of its SIMD and wins the issue arbitration, so its own cycle count under-states what a
    // SIMD shared by `wps` waves needs; the wall clock of the whole launch (one block per CU) gives the SIMD throughput
    static double clock_ghz = 2.4;
    if (wps == 1) clock_ghz = (double)c / (ms * 1e6);          // one wave per SIMD: its cycles span the launch
    const double simd_cyc = ms * 1e6 * clock_ghz / (iters * 64.0 * wps);
    printf("  %-34s waves/SIMD %d : oldest wave %5.2f counter ticks per asm line; launch %.3f ms -> %5.2f ns per asm line per SIMD\n", kNames[K], wps,
           (double)c / (iters * 64.0), ms, ms * 1e6 / (iters * 64.0 * wps));
    (void)simd_cyc;
}

template <int K>
static void run_issue_all(float* out, unsigned long long* cyc) {
    run_issue<K>(out, cyc, 256);
    run_issue<K>(out, cyc, 512);
    run_issue<K>(out, cyc, 1024);
}

template <int KIND>
static void run_coissue(float* out, unsigned long long* cyc, const char* name) {
    const int iters = 4000;
    for (int mode = 1; mode <= 3; ++mode) {
        CK(hipMemset(cyc, 0, 16));
        hipLaunchKernelGGL((coissue_kernel<KIND>), dim3(256), dim3(512), 0, 0, out, cyc, 10, mode, 1.0f);
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((coissue_kernel<KIND>), dim3(256), dim3(512), 0, 0, out, cyc, iters, mode, 1.0f);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long c[2];
        CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
        const int nm = (KIND == 1) ? 8 : 16;
        printf("  %-26s mode %s : %.3f ms; MFMA wave %8llu cycles (%.1f per MFMA), VALU wave %8llu cycles (%.2f per v_fma)\n", name,
               mode == 1 ? "MFMA only " : mode == 2 ? "VALU only " : "both      ", ms, c[0], (double)c[0] / (iters * (double)nm), c[1], (double)c[1] / (iters * 64.0));
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    float* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMalloc(&cyc, 64));

    printf("[1] issue cost (256 blocks; one wave sees N cycles for 64*iters instructions)\n");
    run_issue_all<K_FMA>(out, cyc);
    run_issue_all<K_PKFMA>(out, cyc);
    run_issue_all<K_MUL>(out, cyc);
    run_issue_all<K_PKMUL>(out, cyc);
    run_issue_all<K_ADD>(out, cyc);
    run_issue_all<K_PKADD>(out, cyc);
    run_issue_all<K_MOV>(out, cyc);
    run_issue_all<K_DPPADD>(out, cyc);
    run_issue_all<K_DPPMIR>(out, cyc);
    run_issue_all<K_CNDMASK>(out, cyc);
    run_issue_all<K_CMP>(out, cyc);
    run_issue_all<K_MULLO>(out, cyc);
    run_issue_all<K_MAD24>(out, cyc);
    run_issue_all<K_LSHLOR>(out, cyc);
    run_issue_all<K_ADD3>(out, cyc);
    run_issue_all<K_MAX>(out, cyc);
    run_issue_all<K_FLOOR>(out, cyc);
    run_issue_all<K_CVTI>(out, cyc);
    run_issue_all<K_RCP>(out, cyc);
    run_issue_all<K_RSQ>(out, cyc);
    run_issue_all<K_EXP>(out, cyc);
    run_issue_all<K_CNDS>(out, cyc);
    run_issue_all<K_CMPCND>(out, cyc);
    run_issue_all<K_CMPSCND>(out, cyc);
    run_issue_all<K_AND>(out, cyc);
    run_issue_all<K_ASHR>(out, cyc);
    run_issue_all<K_BFI>(out, cyc);
    run_issue_all<K_LSHLADD>(out, cyc);
    run_issue_all<K_MED3>(out, cyc);
    run_issue_all<K_FRACT>(out, cyc);
    run_issue_all<K_SUBU>(out, cyc);
    run_issue_all<K_MINU>(out, cyc);
    run_issue_all<K_FMAS>(out, cyc);
    run_issue_all<K_FMAC>(out, cyc);

    printf("[2] fp32 MFMA wave + fp32 VALU wave on the same SIMD (512-thread blocks, one per CU)\n");
    run_coissue<0>(out, cyc, "v_mfma_f32_16x16x4_f32");
    run_coissue<1>(out, cyc, "v_mfma_f32_32x32x2_f32");
    run_coissue<2>(out, cyc, "v_mfma_f32_16x16x32_bf16");

    printf("[3] L2/MALL-resident linear stream, 1024 blocks x 256 threads, 64 KB per block per round (64 MB working set), 20 rounds\n");
    {
        const unsigned bpb = 64 * 1024;
        const int blocks = 1024, rounds = 20;
        float* src;
        CK(hipMalloc(&src, (size_t)blocks * bpb));
        CK(hipMemset(src, 0, (size_t)blocks * bpb));
        for (int dma = 0; dma < 2; ++dma) {
            hipEvent_t a, b;
            CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
                if (dma) hipLaunchKernelGGL((stream_kernel<true>), dim3(blocks), dim3(256), 64 * 1024, 0, src, out, bpb, rounds);
                else hipLaunchKernelGGL((stream_kernel<false>), dim3(blocks), dim3(256), 64 * 1024, 0, src, out, bpb, rounds);
                CK(hipEventRecord(b));
                CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("  %-34s %.3f ms  %.2f TB/s\n", dma ? "buffer_load_dwordx4 lds + ds_read" : "buffer_load_dwordx4 (registers)", ms,
                   (double)blocks * bpb * rounds / ms * 1e-9);
        }
    }
    return 0;
}
