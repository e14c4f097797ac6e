// The transformer blocks of the DINO ViT branch (models/vision_transformer.py:123-154,194-214) on PRE-SPLIT operands.
//
// vit.hip's GEMM splits both operands into (h, m, l) bf16 terms on the way into LDS, in EVERY block: each of the ~136 row blocks re-splits the
// same constant weight tile, each of the 6-24 column blocks re-splits the same activation tile ("weak" 2).  Here every matrix
// operand lives in memory already split and already in MFMA fragment order ("packed"):
//
//     packed X [R rows][K]:  piece (rt = r / 16, ks = k / 32, term) = 1 KiB = 64 lanes x 8 bf16, lane = ((k % 32) / 8) * 16 + r % 16, element k % 8
//                            at byte ((rt * K/32 + ks) * 3 + term) * 1024          (term 0 = h, 1 = m, 2 = l of split3.h: x = h + m + l exactly)
//
// so one piece IS the A- or B-operand fragment of v_mfma_f32_16x16x32_bf16 for 16 rows x 32 k, lane-linear: it moves global -> LDS by LDS-DMA
// (buffer_load ... lds, no registers, no VALU) and LDS -> registers by one conflict-free ds_read_b128.  Weights are packed once
// (mvs_x3p_pack at _prepared() time); activations are packed by their PRODUCER: LayerNorm (mvs_layernorm_x3p), the GEMM epilogue
// (fc1 -> GELU -> packed; qkv -> packed Q (pre-scaled), K and V^T per head) and the attention epilogue.  The main loops contain no
// conversion and no split: LDS-DMA, ds_read_b128 and six MFMAs per fragment pair.
//
//   mvs_x3p_pack          fp32 [R][K] -> packed (weights; tests)
//   mvs_layernorm_x3p     LayerNorm rows -> packed (the A operand of qkv / fc1)
//   mvs_gemm_x3p          C = epi(A . B^T), A [M][K] and B [N][K] packed; 128 x 128 x 32 tiles, two LDS stages filled by LDS-DMA one
//                         K step ahead; epilogue: scale / shift, GELU, residual -> fp32 C and / or packed output, or the qkv form
//   mvs_attention_x3p     flash attention on packed Q / K / V^T -> packed [M][C]; S^T = K Q^T so that a lane's accumulators ARE its
//                         fragment of P^T (no LDS round trip of P), online softmax with two cross-lane steps per tile
//   mvs_cls_attention_x3p the CLS row of softmax(Q K^T) per head from the packed operands (mvsformer_model.py:257 reads only that row)
#include <stdlib.h>

#include "common.h"
#include "split3.h"

namespace {
using mvsx3::bf16x8;
using rsrc_t = __amdgpu_buffer_rsrc_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int PIECE = 1024;                                  // bytes of one fragment (16 rows x 32 k, one term)
constexpr int KSTEP = 3 * PIECE;                             // the three terms of one (row tile, k step)

__device__ __forceinline__ rsrc_t rsrc_of(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// LDS-DMA of the three terms of one (row tile, k step) - 3 KiB contiguous in memory AND in LDS: lane l moves 16 bytes from src + voff(l) +
// soff (+ immediate) to lds_dst + 16 l (+ immediate).  The instruction's immediate offset is added to BOTH addresses, so one M0 (LDS base)
// write serves the three pieces.  (In a __device__ helper on purpose: with the builtin directly in a __global__ template body hipcc drops the
// kernel's host stub, tools/probe/gather_probe.hip.)
__device__ __forceinline__ void dma16x3(rsrc_t src, unsigned char* lds_dst, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_ptr_t)lds_dst, 16, voff_bytes, soff_bytes, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_ptr_t)lds_dst, 16, voff_bytes, soff_bytes, PIECE, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_ptr_t)lds_dst, 16, voff_bytes, soff_bytes, 2 * PIECE, 0);
}
__device__ __forceinline__ bf16x8 ldfrag(rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// four fp32 values -> three 8-byte words (4 bf16 each): the h, m, l terms
__device__ __forceinline__ void split4(const float (&v)[4], u32x2& h, u32x2& m, u32x2& l) {
    unsigned a, b, c;
    mvsx3::split3_pair<true>(v[0], v[1], a, b, c);
    h[0] = a, m[0] = b, l[0] = c;
    mvsx3::split3_pair<true>(v[2], v[3], a, b, c);
    h[1] = a, m[1] = b, l[1] = c;
}

// ---------------------------------------------------------------------------------------------------------------- pack
// one thread = (row tile, k step, lane): 8 values -> 3 x 16 bytes; rows >= R and k >= K are zeros
__global__ __launch_bounds__(256) void x3p_pack_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int R, int K, int ld, int RT, int KS) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)RT * KS * 64) return;
    const int lane = (int)(idx & 63), ks = (int)((idx >> 6) % KS), rt = (int)((idx >> 6) / KS);
    const int r = rt * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r < R && k0 + e < K) ? x[(size_t)r * ld + k0 + e] : 0.0f;
    const mvsx3::Split3 s = mvsx3::split3(v);
    unsigned char* d = out + ((size_t)rt * KS + ks) * KSTEP + lane * 16;
    *reinterpret_cast<bf16x8*>(d) = s.h;
    *reinterpret_cast<bf16x8*>(d + PIECE) = s.m;
    *reinterpret_cast<bf16x8*>(d + 2 * PIECE) = s.l;
}

// packed -> fp32 (h + m + l is exact): tests and the CLS row
__global__ __launch_bounds__(256) void x3p_unpack_kernel(const unsigned char* __restrict__ in, float* __restrict__ x, int R, int K, int ld, int RT, int KS) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)RT * KS * 64) return;
    const int lane = (int)(idx & 63), ks = (int)((idx >> 6) % KS), rt = (int)((idx >> 6) / KS);
    const int r = rt * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
    if (r >= R) return;
    const unsigned char* s = in + ((size_t)rt * KS + ks) * KSTEP + lane * 16;
    const bf16x8 h = *reinterpret_cast<const bf16x8*>(s), m = *reinterpret_cast<const bf16x8*>(s + PIECE), l = *reinterpret_cast<const bf16x8*>(s + 2 * PIECE);
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (k0 + e < K) x[(size_t)r * ld + k0 + e] = ((float)h[e] + (float)m[e]) + (float)l[e];
}

// ---------------------------------------------------------------------------------------------------------------- LayerNorm -> packed
// A block owns ONE row tile of 16 rows (four wavefronts x four rows each, one row at a time per wavefront: lane l holds features 8 l ..
// 8 l + 7, C <= 512, a multiple of 32); statistics as vit.hip's layernorm_kernel (mean, then the centered sum of squares).  The split rows
// are assembled in LDS in the packed layout ([k step][term][lane][8]) and leave as whole 1 KiB pieces, one coalesced 16-byte store per lane
// (12.8 us per launch at 8800 x 384 = 34 MB: the same as the form that stored each row's 16-byte chunks straight from registers, and as the
// fp32-output kernel of vit.hip - the launch is bound by its load -> reduce -> reduce -> store chain, not by the store pattern).
// Rows are laid out [images][Np]: rows t >= N of an image are padding and written as zeros (finite keys / values for the attention's
// masked tail).
__global__ __launch_bounds__(256) void layernorm_x3p_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                            unsigned char* __restrict__ out, int rows, int C, int Np, int N, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];           // [C / 32][3][64 lanes][16 bytes]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool active = lane * 8 < C;
    const int KS = C >> 5, ks = lane >> 2, kb = lane & 3;
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, b0 = g0, b1 = g0;
    if (active) {
        g0 = *reinterpret_cast<const f32x4*>(g + lane * 8), g1 = *reinterpret_cast<const f32x4*>(g + lane * 8 + 4);
        b0 = *reinterpret_cast<const f32x4*>(b + lane * 8), b1 = *reinterpret_cast<const f32x4*>(b + lane * 8 + 4);
    }
    // the wavefront's four rows: loads of all four in flight before the first reduction
    float v[4][8];
    bool pad[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = blockIdx.x * 16 + wave * 4 + r;
        pad[r] = row >= rows || (row % Np) >= N;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[r][e] = 0.0f;
        if (active && !pad[r]) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + lane * 8), hi = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + lane * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[r][e] = lo[e], v[r][4 + e] = hi[e];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[r][e];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        const float mean = s / (float)C;
        float q = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = active ? v[r][e] - mean : 0.0f;
            q = fmaf(d, d, q);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        if (!active) continue;
        float y[8];
        if (pad[r]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = 0.0f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = fmaf((v[r][e] - mean) * rstd, g0[e], b0[e]);
                y[4 + e] = fmaf((v[r][4 + e] - mean) * rstd, g1[e], b1[e]);
            }
        }
        const mvsx3::Split3 sp = mvsx3::split3(y);
        unsigned char* d = tile + (size_t)ks * KSTEP + (kb * 16 + wave * 4 + r) * 16;
        *reinterpret_cast<bf16x8*>(d) = sp.h;
        *reinterpret_cast<bf16x8*>(d + PIECE) = sp.m;
        *reinterpret_cast<bf16x8*>(d + 2 * PIECE) = sp.l;
    }
    __syncthreads();
    unsigned char* dst = out + (size_t)blockIdx.x * KS * KSTEP;
    for (int i = threadIdx.x; i < KS * 3 * 64; i += 256) *reinterpret_cast<u32x4*>(dst + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(tile + (size_t)i * 16);
}

// ---------------------------------------------------------------------------------------------------------------- GEMM
struct PGemmArgs {
    const void* Ap;           // packed activations [Mp][K], Mp = art * 16 rows allocated (>= the 128-row tiles the grid touches, or zero-filled by the range check)
    const void* Bp;           // packed weights [Npad][K]
    int M, N, K;              // logical sizes: rows m >= M and columns n >= N are not stored
    int art, brt;             // row tiles allocated in Ap / Bp
    int nbm, nbn;             // 128-row / 128-column blocks
    float* C;                 // optional fp32 output [M][ldc]
    int ldc;
    const float* scale;       // [N] or null
    const float* shift;       // [N] or null
    const float* res;         // [M][ldc] or null
    int act;                  // 0 none, 1 GELU (erf)
    void* Op;                 // optional packed output [Mp][N] (K' = N): the A operand of the next GEMM
    // the qkv form (MODE 1): columns [0, Cd) = q, [Cd, 2 Cd) = k, [2 Cd, 3 Cd) = v, heads of 64 columns; rows = [images][Np]
    void* Qp;                 // [image][head][Np / 16][2][3] pieces: q * qscale
    void* Kp;                 // the same for k
    void* Vp;                 // V^T [image][head][4 row tiles of d][Np / 32][3] pieces, keys of a 32-step in the order pi (see attention)
    int Cd, NH, Np;
    float qscale;
    // implicit convolutions over a packed channel-last map [images * cH * cW pixels][Cp] (MODE 0; models/module.py:353-368,450-466):
    //   a_mode 1: 3 x 3, padding 1: row m = output pixel, K = 9 * Cp, k = tap * Cp + c (tap = ky * 3 + kx)
    //   a_mode 2: ConvTranspose2d(kernel 4, stride 2, padding 1), output-parity class blockIdx.y = ph * 2 + pw: row m = INPUT pixel (y, x),
    //             K = 4 * Cp, k = (th * 2 + tw) * Cp + c, input pixel (y + (ph ? 1 - th : -th), x + (pw ? 1 - tw : -tw)); the class's weights
    //             are b_class_bytes apart; its output pixel is (2y + ph, 2x + pw): output ROW = image * 4 cH cW + (2y + ph) * 2 cW + 2x + pw
    // taps outside the image read row `zero_row` of the map (a row of zeros the caller keeps in its padding)
    int a_mode, cH, cW, Cp, zero_row;
    long long b_class_bytes;
    const float* mul;         // [M][ldc] or null: v *= mul after the activation (before res)
};

// MODE 0: fp32 C and / or plain packed output (and the implicit convolutions).  MODE 1: the qkv form.
// The MFMA's first operand supplies the accumulator's ROW index i (a lane holds four consecutive i), the second the column j (lane & 15).
// First slot = weights (i = n, TI columns per block), second = activations (j = m, GTT * 16 rows per block): a lane holds FOUR CONSECUTIVE n
// of ONE row m - 16-byte fp32 stores, 8-byte packed stores (half a lane's 8-k chunk of the next GEMM's A operand).
//   GTT row tiles of 16 per block: 8 (128 rows) or 7 (112 rows: at M = 8800 that is 79 row blocks, and 79 x 3 / 9 / 12 column blocks fill
//       92.6 % of the 256 CUs' rounds where 69 x 3 / 9 / 12 fill 81 %)
//   TI  columns per block (128 or 64): 64 -> 33-36 KiB stages -> two blocks per CU (a block's epilogue under the other's main loop)
//   NW  wavefronts (4 or 8) = (NW / WJ) along i x WJ along j; GTT = 7 does not split along j (WJ = 1)
// Two LDS stages: the barrier of step k drains the LDS-DMA of step k, issued one step earlier, under step k-1's MFMAs.  (Three stages with
// counted vmcnt waits and a raw s_barrier - step k+2 issued during step k - were built and measured no faster: profiles/r06_bench_x3p.txt.)
template <int MODE, int GTT, int TI, int NW, int WJ>
__global__ __launch_bounds__(NW * 64, (TI == 64 || NW == 8) ? 2 : 1) void gemm_x3p_kernel(const PGemmArgs a) {
    constexpr int GT = GTT * 16;
    constexpr int RT1 = TI / 16, RTS = RT1 + GTT;            // row tiles of the weights / of a stage ([weights | activations], three terms each)
    constexpr int STAGE = RTS * KSTEP;
    constexpr int TPW = (RTS + NW - 1) / NW;                 // row tiles one wavefront fills per K step (the last slot may be empty)
    constexpr int WI = NW / WJ, IT = RT1 / WI, JT = GTT / WJ;  // the wavefront's tile: IT i tiles x JT j tiles of 16
    static_assert(RT1 % WI == 0 && GTT % WJ == 0 && IT >= 1, "tile / wavefront combination");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: block id -> XCD id % 8; the column blocks of one row block run on ONE XCD back to back (its A panel stays in that L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mb = (slot / a.nbn) * 8 + xcd, nb = slot % a.nbn;
    if (mb >= a.nbm) return;
    const int KS = a.K >> 5;
    const rsrc_t rA = rsrc_of(a.Ap, (unsigned)((size_t)a.art * (MODE == 0 && a.a_mode ? a.Cp >> 5 : KS) * KSTEP));
    const rsrc_t rB = rsrc_of(reinterpret_cast<const unsigned char*>(a.Bp) + (MODE == 0 ? (size_t)blockIdx.y * a.b_class_bytes : 0), (unsigned)((size_t)a.brt * KS * KSTEP));
    // LDS-DMA roles: the stage's RTS row tiles are dealt to the wavefronts TPW at a time
    const unsigned voff = lane * 16;
    // implicit convolution: the pixel (image base row, y, x) of this lane's row in each activation row tile the wavefront fills
    const int CB = MODE == 0 && a.a_mode ? a.Cp >> 5 : 1;    // k steps per tap
    int pix_base[TPW], pix_y[TPW], pix_x[TPW];
    if (MODE == 0 && a.a_mode) {
#pragma unroll
        for (int r = 0; r < TPW; ++r) {
            const int g = wave * TPW + r;
            const int m = mb * GT + (g - RT1) * 16 + (lane & 15);
            const int hw = a.cH * a.cW, img = m / hw, rem = m - img * hw;
            pix_base[r] = g >= RT1 && m < a.M ? img * hw : -1;
            pix_y[r] = rem / a.cW;
            pix_x[r] = rem - pix_y[r] * a.cW;
        }
    }
    auto fill = [&](int stage, int ks) {
#pragma unroll
        for (int r = 0; r < TPW; ++r) {
            const int g = wave * TPW + r;                    // row tile of the stage (scalar)
            if (RTS % NW != 0 && g >= RTS) continue;
            const bool first = g < RT1;                      // a weight tile
            if (MODE == 0 && a.a_mode && !first) {           // gathered rows of the map: per-lane source rows
                const int tap = ks / CB, cb = ks - tap * CB;
                int dy, dx;
                if (a.a_mode == 1) {
                    dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                } else {
                    const int ph = (int)blockIdx.y >> 1, pw = (int)blockIdx.y & 1, th = tap >> 1, tw = tap & 1;
                    dy = ph ? 1 - th : -th, dx = pw ? 1 - tw : -tw;
                }
                const int iy = pix_y[r] + dy, ix = pix_x[r] + dx;
                const bool inb = pix_base[r] >= 0 && (unsigned)iy < (unsigned)a.cH && (unsigned)ix < (unsigned)a.cW;
                const int row = inb ? pix_base[r] + iy * a.cW + ix : a.zero_row;
                const unsigned vo = (unsigned)((row >> 4) * CB) * (unsigned)KSTEP + (unsigned)(((lane >> 4) << 8) + ((row & 15) << 4));
                dma16x3(rA, lds + stage * STAGE + g * KSTEP, vo, (unsigned)(cb * KSTEP));
                continue;
            }
            const int rt = first ? nb * RT1 + g : mb * GTT + (g - RT1);
            dma16x3(first ? rB : rA, lds + stage * STAGE + g * KSTEP, voff, (unsigned)((rt * KS + ks) * KSTEP));
        }
    };
    const int wi = wave / WJ, wj = wave % WJ;
    const unsigned char* const f1 = lds + (wi * IT) * KSTEP + lane * 16;
    const unsigned char* const f2 = lds + (RT1 + wj * JT) * KSTEP + lane * 16;

    f32x4 acc[JT][IT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int it = 0; it < IT; ++it) acc[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifndef X3P_ABLATE
#define X3P_ABLATE 0                                         // experiment builds only (make exp EXPFLAGS=-DX3P_ABLATE=n): 1 no LDS-DMA in the loop, 2 no MFMAs in the loop,
#endif                                                       // 3 MFMAs only (no LDS-DMA, barriers, fragment reads in the loop), 4 fragment reads + MFMAs (no LDS-DMA, no barriers)
#if X3P_ABLATE == 3
    bf16x8 t1[IT][3], t2[JT][3];
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int t = 0; t < 3; ++t) t1[it][t] = *reinterpret_cast<const bf16x8*>(f1 + (it * 3 + t) * PIECE);
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int t = 0; t < 3; ++t) t2[jt][t] = *reinterpret_cast<const bf16x8*>(f2 + (jt * 3 + t) * PIECE);
#endif
    auto compute = [&](int stage) {
#if X3P_ABLATE != 3
        bf16x8 t1[IT][3], t2[JT][3];
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int t = 0; t < 3; ++t) t1[it][t] = *reinterpret_cast<const bf16x8*>(f1 + stage * STAGE + (it * 3 + t) * PIECE);
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int t = 0; t < 3; ++t) t2[jt][t] = *reinterpret_cast<const bf16x8*>(f2 + stage * STAGE + (jt * 3 + t) * PIECE);
#else
        asm volatile("" ::: "memory");
#endif
        // smallest products first (split3.h's mfma6 order), each term pair swept over all accumulators: no two consecutive MFMAs share one
#define X3P_SWEEP(TA, TB)                                                                                                              \
    _Pragma("unroll") for (int jt = 0; jt < JT; ++jt) _Pragma("unroll") for (int it = 0; it < IT; ++it) acc[jt][it] =                  \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1[it][TA], t2[jt][TB], acc[jt][it], 0, 0, 0);
        X3P_SWEEP(1, 1)
        X3P_SWEEP(0, 2)
        X3P_SWEEP(2, 0)
        X3P_SWEEP(0, 1)
        X3P_SWEEP(1, 0)
        X3P_SWEEP(0, 0)
#undef X3P_SWEEP
    };

    fill(0, 0);
    int ks = 0;
    for (; ks + 2 <= KS; ks += 2) {                          // two K steps per trip: the stage index is a literal
        if (X3P_ABLATE < 3 || ks == 0) __syncthreads();      // stage 0 has landed (the barrier drains the LDS-DMA queue); stage 1's readers are done
        if (X3P_ABLATE == 0 || X3P_ABLATE == 2) fill(1, ks + 1);
        if (X3P_ABLATE != 2 || ks == 0) compute(0);
        if (X3P_ABLATE < 3) __syncthreads();
        if ((X3P_ABLATE == 0 || X3P_ABLATE == 2) && ks + 2 < KS) fill(0, ks + 2);
        if (X3P_ABLATE != 2) compute(1);
    }
    if (ks < KS) {
        __syncthreads();
        compute(0);
    }

    // ---- epilogue: lane (jl = lane & 15, kb = lane >> 4) holds acc[jt][it][r] = D[n = n0 + it*16 + 4 kb + r][m = m0 + jt*16 + jl]
    const int jl = lane & 15, kb = lane >> 4;
    const int n0 = nb * TI + wi * (IT * 16), m0 = mb * GT + wj * (JT * 16);
    int sec = 0, head = 0, dt0 = 0;
    if (MODE == 1) {                                          // the block's columns lie in ONE of q | k | v (Cd % 128 == 0), the wavefront's in one head
        sec = n0 / a.Cd;
        head = (n0 - sec * a.Cd) >> 6;
        dt0 = (n0 & 63) >> 4;                                 // first d tile (of the head's four) of this wavefront
    }
    const bool has_sc = a.scale != nullptr, has_sh = a.shift != nullptr;
    if (MODE == 1 && sec == 2) {
        // v columns -> V^T pieces [image][head][d tile][key step][term]: lane (kb', jl' = d % 16) element e <-> key pi(kb', e) =
        // (e < 4 ? 4 kb' + e : 16 + 4 kb' + e - 4) of the step.  A lane holds four d of one token; V^T wants four TOKENS of one d: each
        // 16 x 16 accumulator tile goes through a wavefront-private LDS tile (written [token][d], read [d][token]) - the stages are free now.
        __syncthreads();                                      // every wavefront's last fragment reads are done
        float* tile = reinterpret_cast<float*>(lds + wave * 2048);       // 16 rows of 20 floats: rows 4 kb' + r land 16 banks apart
        const int KSV = a.Np >> 5;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int mt = m0 + jt * 16;                      // first row of the token tile (scalar)
            if (mt >= a.M) continue;
            const int img = mt / a.Np, tok = mt - img * a.Np; // (Np % 16 == 0: the tile stays in one image)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int n = n0 + it * 16 + 4 * kb;
                f32x4 sh = {0.f, 0.f, 0.f, 0.f};
                if (has_sh) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[jt][it][r] + sh[r];
                *reinterpret_cast<f32x4*>(tile + jl * 20 + 4 * kb) = v;
                __builtin_amdgcn_wave_barrier();              // a wavefront's LDS operations execute in order
                float t4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) t4[r] = tile[(4 * kb + r) * 20 + jl];
                __builtin_amdgcn_wave_barrier();
                u32x2 h, mm, l;
                split4(t4, h, mm, l);
                unsigned char* d = reinterpret_cast<unsigned char*>(a.Vp) + ((((size_t)img * a.NH + head) * 4 + dt0 + it) * KSV + (tok >> 5)) * KSTEP +
                                   lane * 16 + ((tok >> 4) & 1) * 8;
                *reinterpret_cast<u32x2*>(d) = h;
                *reinterpret_cast<u32x2*>(d + PIECE) = mm;
                *reinterpret_cast<u32x2*>(d + 2 * PIECE) = l;
            }
        }
        return;
    }
    int orow[JT];                                             // output row of the lane's row m (the transposed convolution interleaves its parity classes)
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int m = m0 + jt * 16 + jl;
        orow[jt] = m;
        if (MODE == 0 && a.a_mode == 2) {
            const int hw = a.cH * a.cW, img = m / hw, rem = m - img * hw, y = rem / a.cW, x = rem - y * a.cW;
            orow[jt] = img * 4 * hw + (2 * y + ((int)blockIdx.y >> 1)) * 2 * a.cW + 2 * x + ((int)blockIdx.y & 1);
        }
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int m = m0 + jt * 16 + jl;
        if (m >= a.M) continue;
        int img = 0, tok0 = 0;
        if (MODE == 1) {
            img = (m0 + jt * 16) / a.Np;
            tok0 = (m0 + jt * 16) - img * a.Np;               // first token of this 16-row tile (Np % 16 == 0: the tile stays in one image)
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int n = n0 + it * 16 + 4 * kb;
            if (n >= a.N) continue;
            float v[4];
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (has_sc) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
            if (has_sh) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = fmaf(acc[jt][it][r], sc[r], sh[r]);
                if (MODE == 0) {
                    if (a.act == 1) v[r] = gelu_erf(v[r]);
                    else if (a.act == 2) v[r] = v[r] / (1.0f + __expf(-v[r]));
                    else if (a.act == 3) v[r] = fmaxf(v[r], 0.0f);
                }
            }
            if (MODE == 0) {
                const size_t o = (size_t)orow[jt] * a.ldc + n;
                if (a.mul) {
                    const f32x4 mm = *reinterpret_cast<const f32x4*>(a.mul + o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= mm[r];
                }
                if (a.res) {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                }
                if (a.C) *reinterpret_cast<f32x4*>(a.C + o) = f32x4{v[0], v[1], v[2], v[3]};
                if (a.Op) {
                    u32x2 h, mm, l;
                    split4(v, h, mm, l);
                    const int KSo = a.N >> 5;
                    unsigned char* d = reinterpret_cast<unsigned char*>(a.Op) + ((size_t)(orow[jt] >> 4) * KSo + (n >> 5)) * KSTEP +
                                       ((((n >> 3) & 3) * 16 + (orow[jt] & 15)) * 16) + ((n >> 2) & 1) * 8;
                    *reinterpret_cast<u32x2*>(d) = h;
                    *reinterpret_cast<u32x2*>(d + PIECE) = mm;
                    *reinterpret_cast<u32x2*>(d + 2 * PIECE) = l;
                }
            } else {                                          // q (scaled) or k of head `head`: packed [image][head][token tile][2 k steps]
                if (sec == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= a.qscale;
                }
                u32x2 h, mm, l;
                split4(v, h, mm, l);
                const int dd = (dt0 + it) * 16 + 4 * kb;      // first of the lane's four head dimensions
                unsigned char* base = reinterpret_cast<unsigned char*>(sec == 0 ? a.Qp : a.Kp);
                unsigned char* d = base + ((((size_t)img * a.NH + head) * (a.Np >> 4) + (tok0 >> 4)) * 2 + (dd >> 5)) * KSTEP +
                                   ((((dd >> 3) & 3) * 16 + jl) * 16) + ((dd >> 2) & 1) * 8;
                *reinterpret_cast<u32x2*>(d) = h;
                *reinterpret_cast<u32x2*>(d + PIECE) = mm;
                *reinterpret_cast<u32x2*>(d + 2 * PIECE) = l;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- attention
// softmax(Q K^T) V per (image, head) in flash form on the packed operands of the qkv GEMM (Q already scaled).  A block owns 128 queries
// (4 wavefronts x 2 query tiles of 16) and walks the keys in steps of 32; K and V^T steps (12 + 12 pieces = 24 KiB) arrive by LDS-DMA one
// step ahead (two stages).  Per step and query tile:
//   S^T [key][query] = K Q^T      first slot = K fragment (i = key), second = Q fragment (j = query): a lane holds keys 4 kb + r of both
//                                 16-key tiles for ONE query (lane & 15) - eight values
//   online softmax                row max = the lane's 8 values, then two cross-lane steps (lanes j, j+16, j+32, j+48 share a query);
//                                 the row SUM stays a per-lane partial until the end
//   P^T fragment                  = those eight values, split: element e of lane (kb, j) is key pi(kb, e) - the order V^T was packed in,
//                                 so the fragment never leaves the lane's registers
//   O^T [d][query] += V^T P^T     first slot = V^T fragment (i = d), second = P^T (j = query)
// Output: packed [image * Np + token][heads * 64] (the A operand of the projection GEMM): a lane holds four consecutive d of one token.
constexpr int ASTAGE = 24 * PIECE;

__device__ __forceinline__ float xmax16(float x) { return fmaxf(x, __shfl_xor(x, 16, 64)); }
__device__ __forceinline__ float xmax32(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }

__global__ __launch_bounds__(256, 2) void attention_x3p_kernel(const void* __restrict__ Qp, const void* __restrict__ Kp, const void* __restrict__ Vp,
                                                               void* __restrict__ Op, int N, int Np, int NH) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * ASTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y, img = blockIdx.z;
    const int RT = Np >> 4, KSV = Np >> 5;
    const size_t bh = (size_t)img * NH + head;
    const unsigned char* kbase = reinterpret_cast<const unsigned char*>(Kp) + bh * RT * 2 * KSTEP;
    const unsigned char* vbase = reinterpret_cast<const unsigned char*>(Vp) + bh * 4 * KSV * KSTEP;
    const rsrc_t rq = rsrc_of(reinterpret_cast<const unsigned char*>(Qp) + bh * RT * 2 * KSTEP, (unsigned)((size_t)RT * 2 * KSTEP));
    const rsrc_t rk = rsrc_of(kbase, (unsigned)((size_t)RT * 2 * KSTEP)), rv = rsrc_of(vbase, (unsigned)((size_t)4 * KSV * KSTEP));
    const unsigned voff = lane * 16;
    const int qt0 = blockIdx.x * 8 + wave * 2;                // the wavefront's two query tiles (tiles >= RT read zeros, store nothing)

    bf16x8 qf[2][2][3];                                       // [query tile][k step of d][term]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t) qf[qt][s][t] = ldfrag(rq, voff, (unsigned)(((qt0 + qt) * 2 + s) * KSTEP + t * PIECE));

    f32x4 o[2][4];                                            // [query tile][d tile]
    float mrow[2], lrow[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        mrow[qt] = -INFINITY, lrow[qt] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // LDS-DMA roles: per step, K = 12 contiguous pieces (2 key tiles x 2 d steps x 3 terms): wavefront w moves pieces 3w .. 3w+2;
    // V^T: d tile w's three terms.  Stage = [K 12][V 12].
    auto fill = [&](int stage, int kt) {
        unsigned char* dk = lds + stage * ASTAGE + wave * KSTEP;
        const unsigned ko = (unsigned)((kt * 4 + wave) * KSTEP), vo = (unsigned)((wave * KSV + kt) * KSTEP);
        dma16x3(rk, dk, voff, ko);
        dma16x3(rv, dk + 12 * PIECE, voff, vo);
    };
    const int kb = lane >> 4;
    // One key step: S for both query tiles | reference maxima | P (exponentials, split) | PV.  (Measured and not kept, NOTEBOOK.md round 6:
    // tile 1's S / tile 0's PV MFMAs placed over the other tile's vector work - 0.1197 against 0.1204 ms, within 1 %; s_setprio around the
    // MFMA phases - slower; the order below, term pairs swept over independent accumulators, 0.1186.)
    auto mask_tail = [&](int kt, f32x4 (&s)[2]) {               // the tail step: keys >= N are padding
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kt * 32 + nt * 16 + 4 * kb + r >= N) s[nt][r] = -INFINITY;
    };
    // The reference maximum mrow is shared by the four lanes of a query (their P values meet in one MFMA contraction) but need not be the
    // exact running maximum: any common reference gives the same softmax.  It is raised only when some lane of the wavefront sees a score
    // more than 8 (base-2 exponent: a factor 256) above it - after the first few key steps almost never - so the cross-lane maximum, the
    // rescale of O and their LDS-crossbar waits leave the steady state (P <= 256, exact in the split).
    auto raise_ref = [&](int qt, const f32x4 (&s)[2]) {
        const float lmax = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        if (__builtin_amdgcn_ballot_w64(lmax > mrow[qt] + 8.0f) != 0) {
            const float mnew = fmaxf(mrow[qt], xmax32(xmax16(lmax)));
            const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);          // 0 at the first step (mrow = -inf)
            mrow[qt] = mnew;
            lrow[qt] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
        }
    };
    auto p_tile = [&](int qt, const f32x4 (&s)[2], bf16x8& ph, bf16x8& pm, bf16x8& pl) {
        float p[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s[0][r] - mrow[qt]), p[4 + r] = __builtin_amdgcn_exp2f(s[1][r] - mrow[qt]);
        lrow[qt] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        u32x4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned a, b, c;
            mvsx3::split3_pair<false>(p[2 * e], p[2 * e + 1], a, b, c);
            h[e] = a, m[e] = b, l[e] = c;
        }
        ph = __builtin_bit_cast(bf16x8, h), pm = __builtin_bit_cast(bf16x8, m), pl = __builtin_bit_cast(bf16x8, l);
    };
    // MFMA order: each term pair is swept over ALL independent accumulators (4 in the S phase: 2 query x 2 key tiles; 8 in the PV phase:
    // 2 query x 4 d tiles) before the next pair, so no two consecutive MFMAs share an accumulator (a chain of six dependent MFMAs per
    // accumulator waits out the matrix pipe's latency at every link)
    auto step = [&](int stage, int kt) {
        const unsigned char* sk = lds + stage * ASTAGE + lane * 16;
        const unsigned char* sv = sk + 12 * PIECE;
        const bool tail = kt * 32 + 32 > N;                   // (block-uniform)
        f32x4 s0[2], s1[2];
        bf16x8 ph[2], pm[2], pl[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) s0[nt] = s1[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
            bf16x8 kf[2][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int t = 0; t < 3; ++t) kf[nt][t] = *reinterpret_cast<const bf16x8*>(sk + (nt * 2 + ds) * KSTEP + t * PIECE);
#define X3P_S_SWEEP(TA, TB)                                                                                                     \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                                           \
        s0[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[nt][TA], qf[0][ds][TB], s0[nt], 0, 0, 0);                              \
        s1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[nt][TA], qf[1][ds][TB], s1[nt], 0, 0, 0);                              \
    }
            X3P_S_SWEEP(1, 1)
            X3P_S_SWEEP(0, 2)
            X3P_S_SWEEP(2, 0)
            X3P_S_SWEEP(0, 1)
            X3P_S_SWEEP(1, 0)
            X3P_S_SWEEP(0, 0)
#undef X3P_S_SWEEP
        }
        if (tail) mask_tail(kt, s0), mask_tail(kt, s1);
        raise_ref(0, s0);
        raise_ref(1, s1);
        p_tile(0, s0, ph[0], pm[0], pl[0]);
        p_tile(1, s1, ph[1], pm[1], pl[1]);
        bf16x8 vf[4][3];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int t = 0; t < 3; ++t) vf[dt][t] = *reinterpret_cast<const bf16x8*>(sv + dt * KSTEP + t * PIECE);
#define X3P_PV_SWEEP(TA, TB)                                                                                                    \
    _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                                           \
        o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt][TA], (TB == 0 ? ph[0] : TB == 1 ? pm[0] : pl[0]), o[0][dt], 0, 0, 0); \
        o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt][TA], (TB == 0 ? ph[1] : TB == 1 ? pm[1] : pl[1]), o[1][dt], 0, 0, 0); \
    }
        X3P_PV_SWEEP(1, 1)
        X3P_PV_SWEEP(0, 2)
        X3P_PV_SWEEP(2, 0)
        X3P_PV_SWEEP(0, 1)
        X3P_PV_SWEEP(1, 0)
        X3P_PV_SWEEP(0, 0)
#undef X3P_PV_SWEEP
    };
    const int KT = (N + 31) >> 5;
    fill(0, 0);
    int kt = 0;
    for (; kt + 2 <= KT; kt += 2) {
        __syncthreads();
        fill(1, kt + 1);
        step(0, kt);
        __syncthreads();
        if (kt + 2 < KT) fill(0, kt + 2);
        step(1, kt + 1);
    }
    if (kt < KT) {
        __syncthreads();
        step(0, kt);
    }
    // ---- output: o[qt][dt][r] = O[token (qt0 + qt)*16 + jl][d = dt*16 + 4 kb + r] / row sum
    const int jl = lane & 15;
    const int KSo = (NH * 64) >> 5;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if (qt0 + qt >= RT) continue;
        float l = lrow[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const size_t rt = (size_t)img * RT + qt0 + qt;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const float v[4] = {o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv};
            u32x2 h, m, lo;
            split4(v, h, m, lo);
            const int n = head * 64 + dt * 16 + 4 * kb;
            unsigned char* d = reinterpret_cast<unsigned char*>(Op) + (rt * KSo + (n >> 5)) * KSTEP + ((((n >> 3) & 3) * 16 + jl) * 16) + ((n >> 2) & 1) * 8;
            *reinterpret_cast<u32x2*>(d) = h;
            *reinterpret_cast<u32x2*>(d + PIECE) = m;
            *reinterpret_cast<u32x2*>(d + 2 * PIECE) = lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- CLS row
// att[image][head][key] = softmax_key(q_cls . k_key) over keys < N (q already scaled), from the packed Q / K (h + m + l is exact): one block
// per (image, head), fp32 FMA chains.  The only attention values MVSFormer reads (mvsformer_model.py:257: vit_att[:, :, 0, 1:]).
__global__ __launch_bounds__(256) void cls_attention_kernel(const unsigned char* __restrict__ Qp, const unsigned char* __restrict__ Kp, float* __restrict__ att,
                                                            int N, int Np, int NH) {
    __shared__ float q[64];
    __shared__ float red[4];
    __shared__ float sc[8192];                                 // the row's scores (N <= 8192)
    const int head = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int RT = Np >> 4;
    const size_t bh = (size_t)img * NH + head;
    const unsigned char* qb = Qp + bh * RT * 2 * KSTEP;
    const unsigned char* kbp = Kp + bh * RT * 2 * KSTEP;
    auto elem = [](const unsigned char* base, int row, int d) {
        const unsigned char* p = base + ((size_t)(row >> 4) * 2 + (d >> 5)) * KSTEP + ((((d >> 3) & 3) * 16 + (row & 15)) * 16) + (d & 7) * 2;
        return ((float)*reinterpret_cast<const __bf16*>(p) + (float)*reinterpret_cast<const __bf16*>(p + PIECE)) +
               (float)*reinterpret_cast<const __bf16*>(p + 2 * PIECE);
    };
    if (tid < 64) q[tid] = elem(qb, 0, tid);
    __syncthreads();
    float mx = -INFINITY;
#pragma unroll 1
    for (int key = tid; key < N; key += 256) {
        float acc = 0.0f;
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {                          // 16-byte chunk c of the key's row: d = 8 c .. 8 c + 7
            const unsigned char* p = kbp + ((size_t)(key >> 4) * 2 + (c >> 2)) * KSTEP + (((c & 3) * 16 + (key & 15)) * 16);
            const bf16x8 h = *reinterpret_cast<const bf16x8*>(p), m = *reinterpret_cast<const bf16x8*>(p + PIECE),
                         l = *reinterpret_cast<const bf16x8*>(p + 2 * PIECE);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(q[c * 8 + e], ((float)h[e] + (float)m[e]) + (float)l[e], acc);
        }
        sc[key] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.0f;
#pragma unroll 1
    for (int key = tid; key < N; key += 256) {                 // (a thread re-reads only its own scores)
        const float e = __builtin_amdgcn_exp2f(sc[key] - mx);
        sc[key] = e;
        sum += e;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll 1
    for (int key = tid; key < N; key += 256) att[bh * N + key] = sc[key] * inv;
}
}  // namespace

extern "C" int64_t mvs_x3p_bytes(int64_t rows, int K) {
    if (rows < 1 || K < 1) return 0;
    return ((rows + 15) / 16) * ((K + 31) / 32) * (int64_t)KSTEP;
}

extern "C" int mvs_x3p_pack(const float* x, void* out, int64_t R, int K, int ld, int64_t rows_alloc, mvs_stream_t stream) {
    MVS_REQUIRE(x && out && R >= 1 && K >= 1 && ld >= K && rows_alloc >= R && rows_alloc % 16 == 0 && K % 32 == 0,
                "mvs_x3p_pack: K (%d) must be a multiple of 32, rows_alloc a multiple of 16 >= R", K);
    const int RT = (int)(rows_alloc / 16), KS = K / 32;
    MVS_REQUIRE((int64_t)RT * KS * KSTEP < ((int64_t)1 << 32), "mvs_x3p_pack: packed operand exceeds 4 GiB");
    const long long n = (long long)RT * KS * 64;
    hipLaunchKernelGGL(x3p_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), x, reinterpret_cast<unsigned char*>(out), (int)R, K, ld, RT, KS);
    return mvs::finish_launch("mvs_x3p_pack");
}

extern "C" int mvs_x3p_unpack(const void* in, float* x, int64_t R, int K, int ld, int64_t rows_alloc, mvs_stream_t stream) {
    MVS_REQUIRE(x && in && R >= 1 && K >= 1 && ld >= K && rows_alloc >= R && rows_alloc % 16 == 0 && K % 32 == 0, "mvs_x3p_unpack: bad shape");
    const int RT = (int)(rows_alloc / 16), KS = K / 32;
    const long long n = (long long)RT * KS * 64;
    hipLaunchKernelGGL(x3p_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), reinterpret_cast<const unsigned char*>(in), x, (int)R, K, ld, RT, KS);
    return mvs::finish_launch("mvs_x3p_unpack");
}

extern "C" int mvs_layernorm_x3p(const float* x, const float* gamma, const float* beta, void* out, int64_t rows, int C, int Np, int N, float eps,
                                 mvs_stream_t stream) {
    MVS_REQUIRE(x && gamma && beta && out && rows >= 1 && rows < ((int64_t)1 << 31) && C >= 32 && C <= 512 && C % 32 == 0,
                "mvs_layernorm_x3p: 32 <= C <= 512, a multiple of 32 (got %d)", C);
    MVS_REQUIRE(Np >= 16 && Np % 16 == 0 && N >= 1 && N <= Np && rows % Np == 0, "mvs_layernorm_x3p: rows = images * Np, Np %% 16 == 0, N <= Np");
    // (the packed buffer holds whole row tiles: rows rounded up to 16 are written - zeros beyond `rows`)
    hipLaunchKernelGGL(layernorm_x3p_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), (C / 32) * KSTEP, MVS_STREAM(stream), x, gamma, beta,
                       reinterpret_cast<unsigned char*>(out), (int)rows, C, Np, N, eps);
    return mvs::finish_launch("mvs_layernorm_x3p");
}

namespace {
// MVS_X3P_CFG / MVS_X3P_CFG_QKV (diagnostics, read once): "GTT,TI,NW,WJ" of the plain GEMM / of the qkv form instead of the choice below
struct PCfg { int gtt, ti, nw, wj; };
PCfg env_cfg(const char* name) {
    PCfg c{0, 0, 0, 0};
    if (const char* e = getenv(name)) sscanf(e, "%d,%d,%d,%d", &c.gtt, &c.ti, &c.nw, &c.wj);
    return c;
}
template <int MODE, int GTT, int TI, int NW, int WJ>
void launch_one(const PGemmArgs& a, hipStream_t s) {
    const unsigned grid = (unsigned)(((a.nbm + 7) / 8) * 8 * a.nbn);
    hipLaunchKernelGGL((gemm_x3p_kernel<MODE, GTT, TI, NW, WJ>), dim3(grid, a.a_mode == 2 ? 4 : 1), dim3(NW * 64), 0, s, a);
}
int launch_pgemm(PGemmArgs& a, int mode, int64_t a_rows_alloc, int64_t b_rows_alloc, hipStream_t s) {
    static const PCfg e0 = env_cfg("MVS_X3P_CFG"), e1 = env_cfg("MVS_X3P_CFG_QKV");
    a.art = (int)(a_rows_alloc / 16), a.brt = (int)(b_rows_alloc / 16);
    const int KS = a.K / 32;
    MVS_REQUIRE((int64_t)a.art * KS * KSTEP < ((int64_t)1 << 32) && (int64_t)a.brt * KS * KSTEP < ((int64_t)1 << 32), "mvs_gemm_x3p: a packed operand exceeds 4 GiB");
    PCfg c = mode == 0 ? e0 : e1;
    // measured at the ViT-small shapes (profiles/r06_bench_x3p.txt): eight wavefronts on a 128 x 128 tile (two per SIMD: one's barrier / LDS-DMA
    // issue under the other's MFMAs) for the plain epilogues; the GELU + packed epilogue of fc1 prefers 64-column tiles, two blocks per CU
    // (a block's epilogue under the other block's main loop)
    if (c.gtt == 0) {
        c = (mode == 0 && ((a.act == 1 && a.Op) || a.N <= 64)) ? PCfg{8, 64, 4, 2} : PCfg{8, 128, 8, 4};
        if (c.ti == 128) {                                   // 112-row blocks when they fill the CUs' rounds better (M = 8800: 79 x nbn against 69 x nbn blocks)
            const auto fill = [&](int rows) {
                const long long t = (long long)((a.M + rows - 1) / rows) * ((a.N + 127) / 128);
                return (double)t / (double)((t + 255) / 256 * 256);
            };
            if (fill(112) > fill(128) + 0.05) c = PCfg{7, 128, 8, 1};
        }
    }
    a.nbm = (a.M + c.gtt * 16 - 1) / (c.gtt * 16), a.nbn = (a.N + c.ti - 1) / c.ti;
#define X3P_CASE(M_, G_, TI_, NW_, WJ_) if (mode == M_ && c.gtt == G_ && c.ti == TI_ && c.nw == NW_ && c.wj == WJ_) { launch_one<M_, G_, TI_, NW_, WJ_>(a, s); return mvs::finish_launch("mvs_gemm_x3p"); }
    X3P_CASE(0, 8, 128, 8, 4) X3P_CASE(0, 8, 128, 4, 2) X3P_CASE(0, 8, 64, 4, 2) X3P_CASE(0, 8, 128, 8, 1) X3P_CASE(0, 8, 64, 4, 1)
    X3P_CASE(0, 7, 128, 8, 1) X3P_CASE(0, 7, 128, 4, 1) X3P_CASE(0, 7, 64, 4, 1)
    X3P_CASE(1, 8, 128, 8, 4) X3P_CASE(1, 8, 128, 4, 2) X3P_CASE(1, 7, 128, 8, 1) X3P_CASE(1, 7, 128, 4, 1) X3P_CASE(1, 8, 128, 8, 1)
#undef X3P_CASE
    mvs::set_error("mvs_gemm_x3p: no kernel instance for GTT=%d TI=%d NW=%d WJ=%d", c.gtt, c.ti, c.nw, c.wj);
    return MVS_EINVAL;
}
}  // namespace

extern "C" int mvs_gemm_x3p(const void* Ap, const void* Bp, int M, int N, int K, int64_t a_rows_alloc, int64_t b_rows_alloc, float* C, int ldc,
                            const float* scale, const float* shift, int act, const float* res, void* Op, mvs_stream_t stream) {
    MVS_REQUIRE(Ap && Bp && M >= 1 && N >= 1 && K >= 32 && K % 32 == 0 && N % 4 == 0, "mvs_gemm_x3p: K (%d) %% 32 == 0 and N (%d) %% 4 == 0 required", K, N);
    MVS_REQUIRE(a_rows_alloc % 16 == 0 && b_rows_alloc % 16 == 0 && a_rows_alloc >= M && b_rows_alloc >= N, "mvs_gemm_x3p: packed operands hold fewer rows than M / N");
    MVS_REQUIRE((C || Op) && (!C || ldc >= N) && (!res || C) && act >= 0 && act <= 1 && (!Op || N % 32 == 0) && (!C || ldc % 4 == 0),
                "mvs_gemm_x3p: needs C and / or a packed output (N %% 32 == 0), res only with C, ldc %% 4 == 0, act 0 / 1");
    PGemmArgs a{};
    a.Ap = Ap, a.Bp = Bp, a.M = M, a.N = N, a.K = K, a.C = C, a.ldc = ldc, a.scale = scale, a.shift = shift, a.act = act, a.res = res, a.Op = Op;
    return launch_pgemm(a, 0, a_rows_alloc, b_rows_alloc, MVS_STREAM(stream));
}

extern "C" int mvs_conv_x3p(const void* Xp, int64_t x_rows_alloc, int zero_row, const void* Wp, int64_t w_rows_alloc, int mode, int images, int H, int W,
                            int Cp, int N, float* C, int ldc, const float* scale, const float* shift, int act, const float* mul, void* Op, mvs_stream_t stream) {
    MVS_REQUIRE(Xp && Wp && (mode == 1 || mode == 2) && images >= 1 && H >= 1 && W >= 1 && Cp >= 32 && Cp % 32 == 0 && N >= 4 && N % 4 == 0,
                "mvs_conv_x3p: mode 1 (3x3) / 2 (transposed 4x4 stride 2), Cp %% 32 == 0, N %% 4 == 0");
    const int64_t M = (int64_t)images * H * W;
    MVS_REQUIRE(M < ((int64_t)1 << 29) && x_rows_alloc % 16 == 0 && x_rows_alloc >= M && zero_row >= 0 && zero_row < x_rows_alloc && w_rows_alloc % 16 == 0 && w_rows_alloc >= N,
                "mvs_conv_x3p: the packed map holds fewer rows than pixels, or zero_row outside it");
    MVS_REQUIRE((C || Op) && (!C || (ldc >= N && ldc % 4 == 0)) && (!mul || C || Op) && act >= 0 && act <= 3 && (!Op || N % 32 == 0), "mvs_conv_x3p: bad outputs");
    PGemmArgs a{};
    a.Ap = Xp, a.Bp = Wp, a.M = (int)M, a.N = N, a.K = (mode == 1 ? 9 : 4) * Cp, a.C = C, a.ldc = C ? ldc : N, a.scale = scale, a.shift = shift, a.act = act, a.mul = mul, a.Op = Op;
    a.a_mode = mode, a.cH = H, a.cW = W, a.Cp = Cp, a.zero_row = zero_row;
    a.b_class_bytes = mode == 2 ? (long long)(w_rows_alloc / 16) * (a.K / 32) * KSTEP : 0;
    return launch_pgemm(a, 0, x_rows_alloc, w_rows_alloc, MVS_STREAM(stream));
}

extern "C" int mvs_gemm_x3p_qkv(const void* Ap, const void* Bp, int images, int Np, int C, int heads, int64_t a_rows_alloc, int64_t b_rows_alloc,
                                const float* bias, float qscale, void* Qp, void* Kp, void* Vtp, mvs_stream_t stream) {
    MVS_REQUIRE(Ap && Bp && Qp && Kp && Vtp && images >= 1 && Np >= 32 && Np % 32 == 0 && heads >= 1 && C == heads * 64 && C % 128 == 0,
                "mvs_gemm_x3p_qkv: head dimension 64, C %% 128 == 0, Np %% 32 == 0 (got C=%d heads=%d Np=%d)", C, heads, Np);
    MVS_REQUIRE(a_rows_alloc % 16 == 0 && b_rows_alloc % 16 == 0 && a_rows_alloc >= (int64_t)images * Np && b_rows_alloc >= 3 * C, "mvs_gemm_x3p_qkv: packed operands too small");
    PGemmArgs a{};
    a.Ap = Ap, a.Bp = Bp, a.M = images * Np, a.N = 3 * C, a.K = C, a.shift = bias, a.Qp = Qp, a.Kp = Kp, a.Vp = Vtp, a.Cd = C, a.NH = heads, a.Np = Np,
    a.qscale = qscale;
    return launch_pgemm(a, 1, a_rows_alloc, b_rows_alloc, MVS_STREAM(stream));
}

extern "C" int mvs_attention_x3p(const void* Qp, const void* Kp, const void* Vtp, void* Op, int images, int N, int Np, int heads, mvs_stream_t stream) {
    MVS_REQUIRE(Qp && Kp && Vtp && Op && images >= 1 && images <= 65535 && heads >= 1 && heads <= 65535 && N >= 1 && Np >= N && Np % 32 == 0 && (heads * 64) % 32 == 0,
                "mvs_attention_x3p: Np %% 32 == 0, N <= Np");
    MVS_REQUIRE((int64_t)Np * 8 * KSTEP < ((int64_t)1 << 32), "mvs_attention_x3p: one head exceeds the buffer range");
    hipLaunchKernelGGL(attention_x3p_kernel, dim3((Np / 16 + 7) / 8, heads, images), dim3(256), 0, MVS_STREAM(stream), Qp, Kp, Vtp, Op, N, Np, heads);
    return mvs::finish_launch("mvs_attention_x3p");
}

extern "C" int mvs_cls_attention_x3p(const void* Qp, const void* Kp, float* att, int images, int N, int Np, int heads, mvs_stream_t stream) {
    MVS_REQUIRE(Qp && Kp && att && images >= 1 && images <= 65535 && heads >= 1 && N >= 1 && N <= 8192 && Np >= N && Np % 16 == 0, "mvs_cls_attention_x3p: N <= 8192, Np %% 16 == 0");
    hipLaunchKernelGGL(cls_attention_kernel, dim3(heads, images), dim3(256), 0, MVS_STREAM(stream), reinterpret_cast<const unsigned char*>(Qp),
                       reinterpret_cast<const unsigned char*>(Kp), att, N, Np, heads);
    return mvs::finish_launch("mvs_cls_attention_x3p");
}
