// The DINO ViT-small feature branch of MVSFormer-P (models/vision_transformer.py:104-154,194-214,324-451,
// models/module.py:353-368,450-466, models/mvsformer_model.py:243-262) on gfx950, eval mode, fp32 in / fp32 out / fp32-EQUIVALENT arithmetic:
// every matrix product runs on the bf16 matrix cores in the three-term split form of split3.h (x = h + m + l exactly, six
// v_mfma_f32_16x16x32_bf16 per K = 32 step, fp32 accumulation), like the regularizer's convolutions.
//
//   mvs_gemm_x3        C = epi(alpha * A . B^T): batched over two batch axes with independent strides per operand (a linear layer, one
//                      attention head's Q.K^T or P.V, ...).  A is read plainly, as the im2col of a 3x3 / pad-1 convolution over a
//                      channel-last map, or as the 2x2 taps of one output-parity class of a ConvTranspose2d(k 4, s 2, p 1) (implicit
//                      GEMM: no im2col buffer).  Epilogue: per-column scale / shift (bias, folded BatchNorm), GELU(erf) / Swish,
//                      elementwise product with a second tensor, residual add.
//   mvs_layernorm      rows of up to 1024 features, one wavefront per row, two passes in registers
//   mvs_softmax_rows   y = softmax(scale * x) over rows of up to 8192 elements, one block per row
//   mvs_bicubic_resize ATen's upsample_bicubic2d (align_corners = False, A = -0.75, clamped taps) with explicit scale factors - the image
//                      resize of mvsformer_model.py:246-247 and the position-table resize of vision_transformer.py:394-416
#include <stdlib.h>

#include "common.h"
#include "geometry.h"
#include "split3.h"

namespace {
using mvsx3::bf16x8;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int ROWB = BK * 2 + 16;                            // LDS bytes per tile row (32 bf16 + 16 bytes: rows 20 banks apart)
constexpr int TERMB = BM * ROWB;                             // one term of one operand tile

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* scale;       // [N] or null
    const float* shift;       // [N] or null
    const float* mul;         // C-shaped or null
    const float* res;         // C-shaped or null
    long long sA1, sA2, sB1, sB2, sC1, sC2;                  // batch strides (elements) of the two batch axes
    int M, N, K, lda, ldb, ldc, nb2;
    int b_kn;                 // 0: B[n][k] (k contiguous), 1: B[k][n] (n contiguous)
    int a_mode;               // 0 plain; 1 conv3x3 pad 1 over [H][W][Cp] (k = tap*Cp + c, m = y*W + x); 2 ConvTranspose2d k4 s2 p1 parity class
    int H, W, Cp;             //    (class = second batch index: ph = class / 2, pw = class % 2; k = (th*2 + tw)*Cp + c; m = y*W + x of the INPUT grid)
    int act;                  // 0 none, 1 GELU (erf), 2 Swish
    float alpha;
    // B as the implicit patch matrix of a 2-D convolution over an NCHW map (the FPN's training convolutions, fp32 [C][H][W] per batch item):
    //   b_mode 1  B[k = c*KS*KS + tap][n = output pixel]   = x[c][oy*S - P + ky][ox*S - P + kx]           forward:  A = weight [Cout][Cin*KS*KS]
    //   b_mode 2  B[k = c*KS*KS + tap][n = INPUT pixel]    = dy[c][(iy + P - ky)/S][(ix + P - kx)/S]      data gradient: A = weight as [Cin][Cout*KS*KS]
    //   b_mode 3  B[n = c*KS*KS + tap][k = output pixel]   = x[c][oy*S - P + ky][ox*S - P + kx]           weight gradient: A = dy [Cout][Ho*Wo]
    // (zero outside the map / where the stride does not divide); cH x cW = the gathered map, cHo x cWo = the convolution's output grid.
    int b_mode, cH, cW, cHo, cWo, cKS, cS, cP;
    int ksplit;               // > 0: batch index b2 owns k in [b2*ksplit, (b2+1)*ksplit) (split-K partial products, reduced by the caller)
};

// element (channel*KS*KS + tap, pixel) of the implicit patch matrices above
__device__ __forceinline__ float conv_elem(const GemmArgs& a, const float* __restrict__ base, int ct, int pix) {
    const int KS2 = a.cKS * a.cKS, c = ct / KS2, tap = ct % KS2, ky = tap / a.cKS, kx = tap % a.cKS;
    if (a.b_mode == 2) {
        const int iy = pix / a.cW, ix = pix % a.cW;         // pixel of the INPUT grid; gathered map = dy [C][cHo][cWo]
        const int ty = iy + a.cP - ky, tx = ix + a.cP - kx;
        if (ty < 0 || tx < 0 || ty % a.cS || tx % a.cS) return 0.0f;
        const int oy = ty / a.cS, ox = tx / a.cS;
        if (oy >= a.cHo || ox >= a.cWo) return 0.0f;
        return base[((size_t)c * a.cHo + oy) * a.cWo + ox];
    }
    const int oy = pix / a.cWo, ox = pix % a.cWo;
    const int iy = oy * a.cS - a.cP + ky, ix = ox * a.cS - a.cP + kx;
    if ((unsigned)iy >= (unsigned)a.cH || (unsigned)ix >= (unsigned)a.cW) return 0.0f;
    return base[((size_t)c * a.cH + iy) * a.cW + ix];
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// 8 consecutive k of row m of the A operand (zeros beyond M / K / the image)
__device__ __forceinline__ void load_a8(const GemmArgs& a, const float* __restrict__ Ab, int m, int k0, int cls, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    if (m >= a.M || k0 >= a.K) return;
    const float* p;
    if (a.a_mode == 0) {
        p = Ab + (size_t)m * a.lda + k0;
    } else {
        const int tap = k0 / a.Cp, c = k0 % a.Cp;            // Cp is a multiple of 8: the 8 values share the tap
        const int y = m / a.W, x = m % a.W;
        int iy, ix;
        if (a.a_mode == 1) {
            iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        } else {
            const int ph = cls >> 1, pw = cls & 1, th = tap >> 1, tw = tap & 1;
            iy = y + (ph ? (th == 0 ? 1 : 0) : (th == 0 ? 0 : -1));          // ph = 0: ky = 1, 3 -> iy = y, y - 1;  ph = 1: ky = 0, 2 -> y + 1, y
            ix = x + (pw ? (tw == 0 ? 1 : 0) : (tw == 0 ? 0 : -1));
        }
        if ((unsigned)iy >= (unsigned)a.H || (unsigned)ix >= (unsigned)a.W) return;
        p = Ab + ((size_t)iy * a.W + ix) * a.Cp + c;
    }
    if (k0 + 8 <= a.K && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = lo[e], v[4 + e] = hi[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < a.K) v[e] = p[e];
    }
}

__device__ __forceinline__ void store_split(unsigned char* tile, int term_bytes, int row, int kseg, const float (&v)[8]) {
    const mvsx3::Split3 s = mvsx3::split3(v);
    unsigned char* d = tile + row * ROWB + kseg * 2;
    *reinterpret_cast<bf16x8*>(d) = s.h;
    *reinterpret_cast<bf16x8*>(d + term_bytes) = s.m;
    *reinterpret_cast<bf16x8*>(d + 2 * term_bytes) = s.l;
}

// RT = row tiles of 16 per wavefront: the block's tile is (64 * RT) x 64.  RT = 2 (the large GEMMs: M >= 2048) reads every B fragment
// for two row tiles - 18 fragment reads per 48 MFMAs instead of 15 per 24 - and halves the barriers per MFMA.
template <int RT>
__global__ __launch_bounds__(256) void gemm_x3_kernel(const GemmArgs a) {
    constexpr int TERMA = RT * TERMB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * TERMA + 3 * TERMB];          // A terms h|m|l, then B terms h|m|l
    unsigned char* tA = lds;
    unsigned char* tB = lds + 3 * TERMA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b1 = blockIdx.z / a.nb2, b2 = blockIdx.z % a.nb2;
    const float* Ab = a.A + b1 * a.sA1 + b2 * a.sA2;
    const float* Bb = a.B + b1 * a.sB1 + b2 * a.sB2;
    const int m0 = blockIdx.y * (BM * RT), n0 = blockIdx.x * BN;
    const int j = lane & 15, kb = lane >> 4;

    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging roles: A (and B stored [n][k]): thread -> (row = tid / 4, 8 k at (tid % 4) * 8); B stored [k][n]: thread -> (k = tid / 8, 8 n)
    const int srow = tid >> 2, skseg = (tid & 3) * 8;
    const int tk = tid >> 3, tn = (tid & 7) * 8;
    float pa[RT][8], pb[8];
    const int k_begin = a.ksplit > 0 ? b2 * a.ksplit : 0, k_end = a.ksplit > 0 ? min(a.K, k_begin + a.ksplit) : a.K;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            load_a8(a, Ab, m0 + rt * 64 + srow, k0 + skseg, b2, pa[rt]);
            if (a.ksplit > 0) {                              // a split's last tile may run past its range
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k0 + skseg + e >= k_end) pa[rt][e] = 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[e] = 0.0f;
        if (a.b_mode == 3) {                                 // rows n = (channel, tap), 8 consecutive k = output pixels
            const int n = n0 + srow, k = k0 + skseg;
            if (n < a.N) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k + e < k_end) pb[e] = conv_elem(a, Bb, n, k + e);
            }
        } else if (a.b_mode != 0) {                          // row k = (channel, tap), 8 consecutive n = pixels
            const int k = k0 + tk, n = n0 + tn;
            if (k < k_end) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < a.N) pb[e] = conv_elem(a, Bb, k, n + e);
            }
        } else if (a.b_kn == 0) {
            const int n = n0 + srow, k = k0 + skseg;
            if (n < a.N && k < a.K) {
                const float* p = Bb + (size_t)n * a.ldb + k;
                if (k + 8 <= a.K && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pb[e] = lo[e], pb[4 + e] = hi[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (k + e < a.K) pb[e] = p[e];
                }
            }
        } else {
            const int k = k0 + tk, n = n0 + tn;
            if (k < a.K) {
                const float* p = Bb + (size_t)k * a.ldb + n;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < a.N) pb[e] = p[e];
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) store_split(tA, TERMA, rt * 64 + srow, skseg, pa[rt]);
        if (a.b_mode == 3 || (a.b_mode == 0 && a.b_kn == 0)) {
            store_split(tB, TERMB, srow, skseg, pb);
        } else {                                             // transposed into [n][k]: 2-byte stores
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 h, m, l;
                mvsx3::split3(pb[e], h, m, l);
                unsigned char* d = tB + (tn + e) * ROWB + tk * 2;
                *reinterpret_cast<__bf16*>(d) = h;
                *reinterpret_cast<__bf16*>(d + TERMB) = m;
                *reinterpret_cast<__bf16*>(d + 2 * TERMB) = l;
            }
        }
    };

    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();                                     // the previous step's fragment reads are done
        commit();
        __syncthreads();
        if (k0 + BK < k_end) fetch(k0 + BK);                 // the next tile's loads travel under this step's MFMAs
        bf16x8 ah[RT], am[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned char* ap = tA + (rt * 64 + wave * 16 + j) * ROWB + kb * 16;
            ah[rt] = *reinterpret_cast<const bf16x8*>(ap), am[rt] = *reinterpret_cast<const bf16x8*>(ap + TERMA),
            al[rt] = *reinterpret_cast<const bf16x8*>(ap + 2 * TERMA);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned char* bp = tB + (t * 16 + j) * ROWB + kb * 16;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp), bm = *reinterpret_cast<const bf16x8*>(bp + TERMB),
                         bl = *reinterpret_cast<const bf16x8*>(bp + 2 * TERMB);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mvsx3::mfma6(ah[rt], am[rt], al[rt], bh, bm, bl, acc[rt][t]);
        }
    }
    // D[i = 4*kb + r (row of the wave's 16)][j = column of the tile]
    float* Cb = a.C + b1 * a.sC1 + b2 * a.sC2;
    const size_t eoff = (size_t)(b1 * a.sC1 + b2 * a.sC2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + t * 16 + j;
        if (n >= a.N) continue;
        const float sc = a.scale ? a.scale[n] : 1.0f, sh = a.shift ? a.shift[n] : 0.0f;
#pragma unroll
        for (int rr = 0; rr < 4 * RT; ++rr) {
            const int rt = rr >> 2, r = rr & 3;
            const int m = m0 + rt * 64 + wave * 16 + kb * 4 + r;
            if (m >= a.M) continue;
            float v = acc[rt][t][r] * a.alpha;
            v = fmaf(v, sc, sh);
            if (a.act == 1) v = gelu_erf(v);
            else if (a.act == 2) v = v / (1.0f + __expf(-v));
            else if (a.act == 3) v = fmaxf(v, 0.0f);
            const size_t o = (size_t)m * a.ldc + n;
            if (a.mul) v *= a.mul[eoff + o];
            if (a.res) v += a.res[eoff + o];
            Cb[o] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- plain GEMM, pipelined
// The plain case (A [M][K] and B [N][K] row-major, K a multiple of 32, rows 16-byte aligned: the transformer's linear layers) with TWO
// tiles of global loads in flight.  The general kernel above prefetches one tile ahead and its loads branch (alignment / range / mode),
// so every wait is a full drain and a K step pays most of a global round trip: 18 % of the split-form rate.  Here every load is a
// 16-byte buffer load whose out-of-range cases (rows beyond M / N, tiles beyond K) are OFFSETS beyond the descriptor - no branches, a
// fixed number of loads per tile, so the wait before a tile's commit leaves the next tile's loads flying.
__device__ __forceinline__ f32x4 buf_load4(mvs::rsrc_t r, unsigned voff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, 0, 0));
}

__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

template <int RT>
__global__ __launch_bounds__(256) void gemm_x3_fast_kernel(const GemmArgs a) {
    using mvs::rsrc_t;
    constexpr int TERMA = RT * TERMB;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * TERMA + 3 * TERMB];
    unsigned char* tA = lds;
    unsigned char* tB = lds + 3 * TERMA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b1 = blockIdx.z / a.nb2, b2 = blockIdx.z % a.nb2;
    const float* Ab = a.A + b1 * a.sA1 + b2 * a.sA2;
    const float* Bb = a.B + b1 * a.sB1 + b2 * a.sB2;
    const int m0 = blockIdx.y * (BM * RT), n0 = blockIdx.x * BN;
    const int j = lane & 15, kb = lane >> 4;
    // (the batch item's base address is block-uniform, but its 64-bit arithmetic may be done on the vector ALU: a descriptor word that lives
    //  in a vector register turns EVERY buffer load into a readfirstlane waterfall loop - seen in the RT = 2 instance; pin them scalar)
    const rsrc_t ra = mvs::make_rsrc(uniform_ptr(Ab), (unsigned)((size_t)a.M * a.lda * 4)), rb = mvs::make_rsrc(uniform_ptr(Bb), (unsigned)((size_t)a.N * a.ldb * 4));
    const int srow = tid >> 2, skseg = (tid & 3) * 8;
    unsigned offA[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m0 + rt * 64 + srow;
        offA[rt] = m < a.M ? (unsigned)(((size_t)m * a.lda + skseg) * 4) : OOB;
    }
    const unsigned offB = n0 + srow < a.N ? (unsigned)(((size_t)(n0 + srow) * a.ldb + skseg) * 4) : OOB;

    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 pa[2][RT][2], pb[2][2];                            // two tiles of loads: [set][row tile][half of the 8 k]
    auto fetch = [&](int set, int k0) {                      // `set` is a literal at every call: the sets stay in registers
        const unsigned kofs = k0 < a.K ? (unsigned)k0 * 4u : OOB;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned o = (offA[rt] | kofs) & OOB ? OOB : offA[rt] + kofs;
            pa[set][rt][0] = buf_load4(ra, o);
            pa[set][rt][1] = buf_load4(ra, o == OOB ? OOB : o + 16u);
        }
        const unsigned o = (offB | kofs) & OOB ? OOB : offB + kofs;
        pb[set][0] = buf_load4(rb, o);
        pb[set][1] = buf_load4(rb, o == OOB ? OOB : o + 16u);
    };
    auto commit = [&](int set) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float v[8] = {pa[set][rt][0][0], pa[set][rt][0][1], pa[set][rt][0][2], pa[set][rt][0][3],
                                pa[set][rt][1][0], pa[set][rt][1][1], pa[set][rt][1][2], pa[set][rt][1][3]};
            store_split(tA, TERMA, rt * 64 + srow, skseg, v);
        }
        const float v[8] = {pb[set][0][0], pb[set][0][1], pb[set][0][2], pb[set][0][3], pb[set][1][0], pb[set][1][1], pb[set][1][2], pb[set][1][3]};
        store_split(tB, TERMB, srow, skseg, v);
    };
    auto mfmas = [&]() {
        bf16x8 ah[RT], am[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned char* ap = tA + (rt * 64 + wave * 16 + j) * ROWB + kb * 16;
            ah[rt] = *reinterpret_cast<const bf16x8*>(ap), am[rt] = *reinterpret_cast<const bf16x8*>(ap + TERMA),
            al[rt] = *reinterpret_cast<const bf16x8*>(ap + 2 * TERMA);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned char* bp = tB + (t * 16 + j) * ROWB + kb * 16;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp), bm = *reinterpret_cast<const bf16x8*>(bp + TERMB),
                         bl = *reinterpret_cast<const bf16x8*>(bp + 2 * TERMB);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mvsx3::mfma6(ah[rt], am[rt], al[rt], bh, bm, bl, acc[rt][t]);
        }
    };
    // sched_barrier: the machine scheduler otherwise sinks a tile's loads below the MFMAs towards their use (one tile ahead at best, and
    // the wait at the loop head becomes a full drain); pinned in program order the waits are counted - vmcnt(loads of the younger tile)
    fetch(0, 0);
    __builtin_amdgcn_sched_barrier(0);                       // (tile 0's loads first: the loop head waits for the OLDER tile only)
    fetch(1, BK);
    __builtin_amdgcn_sched_barrier(0);
    int k0 = 0;
    for (; k0 + 2 * BK <= a.K; k0 += 2 * BK) {               // pairs of tiles (no exit between the halves: the wait counts stay exact)
        __syncthreads();                                     // the previous tile's fragment reads are done
        commit(0);
        __syncthreads();
        fetch(0, k0 + 2 * BK);
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
        __syncthreads();
        commit(1);
        __syncthreads();
        fetch(1, k0 + 3 * BK);
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
    }
    if (k0 < a.K) {                                          // an odd tile count: the last tile sits in set 0
        __syncthreads();
        commit(0);
        __syncthreads();
        mfmas();
    }
    float* Cb = a.C + b1 * a.sC1 + b2 * a.sC2;
    const size_t eoff = (size_t)(b1 * a.sC1 + b2 * a.sC2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + t * 16 + j;
        if (n >= a.N) continue;
        const float sc = a.scale ? a.scale[n] : 1.0f, sh = a.shift ? a.shift[n] : 0.0f;
#pragma unroll
        for (int rr = 0; rr < 4 * RT; ++rr) {
            const int rt = rr >> 2, r = rr & 3;
            const int m = m0 + rt * 64 + wave * 16 + kb * 4 + r;
            if (m >= a.M) continue;
            float v = acc[rt][t][r] * a.alpha;
            v = fmaf(v, sc, sh);
            if (a.act == 1) v = gelu_erf(v);
            else if (a.act == 2) v = v / (1.0f + __expf(-v));
            else if (a.act == 3) v = fmaxf(v, 0.0f);
            const size_t o = (size_t)m * a.ldc + n;
            if (a.mul) v *= a.mul[eoff + o];
            if (a.res) v += a.res[eoff + o];
            Cb[o] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- attention
// softmax(Q K^T * scale) V per (image, head) without materializing the N x N matrix (flash form, online softmax), split-form arithmetic:
//   block = 64 queries (4 wavefronts x 16), loop over tiles of 32 keys: S = Q K^T (Q's split fragments live in registers, K's in LDS),
//   running row max / sum in the accumulator layout (a lane's four rows 4*kb + r; the 16 lanes of a group hold 16 keys: xor butterflies),
//   P -> split -> the wavefront's own LDS tile -> A operand of P V (V^T tile [d][key] in LDS as the B operand), O rescaled per row.
// qkv = [B][N][3C] packed rows (q | k | v, head h at columns h*64), vt = V transposed [B][heads][64][N], out = [B][N][C] (head h at h*64).
// hd = 64 only (ViT-small / base); scale is applied to Q before the split (exact for 64^-0.5 = 0.125).
constexpr int AT_KT = 32;                                    // keys per tile
// LDS tiles WITHOUT row padding (36 KB per block: FOUR blocks per CU - 840 blocks of the half-size ViT input then run as one round of 1024
// slots instead of a full round of 768 and a nearly empty second one); bank conflicts of the 16-byte fragment reads (16 lanes = 16 rows,
// same chunk) are avoided by storing 16-byte chunk c of row r at chunk c ^ f(r):
//   K tile  [32 keys][64 d]  bf16, 128-byte rows (8 chunks):  f(r) = (r >> 1) & 7   (two rows per 256 bytes of banks)
//   V^T     [64 d][32 keys]  bf16,  64-byte rows (4 chunks):  f(r) = (r >> 2) & 3   (four rows per 256 bytes)
//   P       [16 q][32 keys]  bf16,  64-byte rows, per wavefront: as V^T
constexpr int AT_KROW = 64 * 2, AT_VROW = AT_KT * 2, AT_PROW = AT_KT * 2;
constexpr int AT_KTERM = AT_KT * AT_KROW, AT_VTERM = 64 * AT_VROW, AT_PTERM = 16 * AT_PROW;
__device__ __forceinline__ int at_koff(int row, int chunk) { return row * AT_KROW + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int at_voff(int row, int chunk) { return row * AT_VROW + ((chunk ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(256, 4) void attention_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ vt, float* __restrict__ out, int N,
                                                           int NH, int ldv, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * AT_KTERM + 3 * AT_VTERM + 4 * 3 * AT_PTERM];
    unsigned char* kl = lds;
    unsigned char* vl = lds + 3 * AT_KTERM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* pl = lds + 3 * AT_KTERM + 3 * AT_VTERM + wave * 3 * AT_PTERM;
    const int j = lane & 15, kb = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z, C = NH * 64;
    const float* qrow = qkv + (size_t)b * N * 3 * C + h * 64;
    const int q0 = blockIdx.x * 64 + wave * 16;

    // Q fragments (A operand of S): lane (i = query q0 + j, kb) holds d = 32*s + 8*kb .. + 7
    bf16x8 qa[2][3];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float v[8];
        const int q = q0 + j;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = q < N ? qrow[(size_t)q * 3 * C + 32 * st + 8 * kb + e] * scale : 0.0f;
        const mvsx3::Split3 sp = mvsx3::split3(v);
        qa[st][0] = sp.h, qa[st][1] = sp.m, qa[st][2] = sp.l;
    }
    f32x4 acc_o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc_o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow[4], lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mrow[r] = -INFINITY, lrow[r] = 0.0f;

    // staging roles: K tile: thread -> (key = tid / 8, 8 d at (tid % 8) * 8); V^T tile: thread -> (d = tid / 4, 8 keys at (tid % 4) * 8).
    // All loads are 16-byte BUFFER loads (keys beyond N / tiles beyond the last: offsets beyond the descriptor -> zeros): no branches, a
    // fixed number of loads per tile, so TWO tiles of loads can be in flight with counted waits (the scheduler's order is pinned below).
    const int kkey = tid >> 3, kd = (tid & 7) * 8, vd = tid >> 2, vk = (tid & 3) * 8;
    constexpr unsigned OOB = 0x80000000u;
    const mvs::rsrc_t rk = mvs::make_rsrc(qrow, (unsigned)((size_t)N * 3 * C * 4));
    const mvs::rsrc_t rv = mvs::make_rsrc(vt + ((size_t)b * NH + h) * 64 * ldv, (unsigned)((size_t)64 * ldv * 4));
    const unsigned kbase = (unsigned)(((size_t)kkey * 3 * C + C + kd) * 4), vbase = (unsigned)(((size_t)vd * ldv + vk) * 4);
    f32x4 pk[1][2], pv[1][2];
    auto fetch = [&](int set, int kt) {                      // `set` is a literal at every call
        const unsigned ko = kt + kkey < N ? kbase + (unsigned)((size_t)kt * 3 * C * 4) : OOB;
        const unsigned vo = kt < N ? vbase + (unsigned)kt * 4u : OOB;
        pk[set][0] = buf_load4(rk, ko);
        pk[set][1] = buf_load4(rk, ko == OOB ? OOB : ko + 16u);
        pv[set][0] = buf_load4(rv, vo);
        pv[set][1] = buf_load4(rv, vo == OOB ? OOB : vo + 16u);
    };
    auto commit = [&](int set) {
        const float kv[8] = {pk[set][0][0], pk[set][0][1], pk[set][0][2], pk[set][0][3], pk[set][1][0], pk[set][1][1], pk[set][1][2], pk[set][1][3]};
        const float vv[8] = {pv[set][0][0], pv[set][0][1], pv[set][0][2], pv[set][0][3], pv[set][1][0], pv[set][1][1], pv[set][1][2], pv[set][1][3]};
        const mvsx3::Split3 a = mvsx3::split3(kv), c = mvsx3::split3(vv);
        unsigned char* d0 = kl + at_koff(kkey, kd >> 3);
        *reinterpret_cast<bf16x8*>(d0) = a.h;
        *reinterpret_cast<bf16x8*>(d0 + AT_KTERM) = a.m;
        *reinterpret_cast<bf16x8*>(d0 + 2 * AT_KTERM) = a.l;
        unsigned char* d1 = vl + at_voff(vd, vk >> 3);
        *reinterpret_cast<bf16x8*>(d1) = c.h;
        *reinterpret_cast<bf16x8*>(d1 + AT_VTERM) = c.m;
        *reinterpret_cast<bf16x8*>(d1 + 2 * AT_VTERM) = c.l;
    };
    auto tile = [&](int kt) {
    // ---- S = Q K^T for 2 x 16 keys: D[i = query 4*kb + r][j = key]
        f32x4 sacc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const unsigned char* bp = kl + at_koff(nt * 16 + j, 4 * st + kb);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp), bm = *reinterpret_cast<const bf16x8*>(bp + AT_KTERM),
                             bl = *reinterpret_cast<const bf16x8*>(bp + 2 * AT_KTERM);
                c = mvsx3::mfma6(qa[st][0], qa[st][1], qa[st][2], bh, bm, bl, c);
            }
            if (kt + nt * 16 + j >= N) c = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            sacc[nt] = c;
        }
        // ---- online softmax over the tile's 32 keys, per row r of this lane group
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = fmaxf(sacc[0][r], sacc[1][r]);
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
            const float mnew = fmaxf(mrow[r], mx);
            const float p0 = __expf(sacc[0][r] - mnew), p1 = __expf(sacc[1][r] - mnew);
            float sum = p0 + p1;
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
            alpha[r] = __expf(mrow[r] - mnew);
            lrow[r] = lrow[r] * alpha[r] + sum;
            mrow[r] = mnew;
            sacc[0][r] = p0, sacc[1][r] = p1;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[t][r] *= alpha[r];
        // ---- P (this wavefront's 16 x 32 tile) -> split -> own LDS tile [term][query][key]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __bf16 ph, pm, plo;
                mvsx3::split3_bounded(sacc[nt][r], ph, pm, plo);
                unsigned char* d = pl + at_voff(4 * kb + r, (nt * 16 + j) >> 3) + ((nt * 16 + j) & 7) * 2;
                *reinterpret_cast<__bf16*>(d) = ph;
                *reinterpret_cast<__bf16*>(d + AT_PTERM) = pm;
                *reinterpret_cast<__bf16*>(d + 2 * AT_PTERM) = plo;
            }
        __builtin_amdgcn_wave_barrier();                     // a wavefront's LDS operations execute in order: its own stores are visible to its loads
        const unsigned char* ap = pl + at_voff(j, kb);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap), am = *reinterpret_cast<const bf16x8*>(ap + AT_PTERM),
                     al = *reinterpret_cast<const bf16x8*>(ap + 2 * AT_PTERM);
        // ---- O += P V: B operand = V^T tile [d = 16*t + j][keys 8*kb .. + 7]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned char* bp = vl + at_voff(t * 16 + j, kb);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp), bm = *reinterpret_cast<const bf16x8*>(bp + AT_VTERM),
                         bl = *reinterpret_cast<const bf16x8*>(bp + 2 * AT_VTERM);
            acc_o[t] = mvsx3::mfma6(ah, am, al, bh, bm, bl, acc_o[t]);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // one tile of loads ahead (two were built: the counted waits work - vmcnt(4) with the younger tile flying - but the tile is bound by its
    // dependent chain S -> row max / sum butterflies -> P -> LDS -> P V, not by the loads: 3.59 -> 3.53 ms, and the second register set
    // costs the fourth block per CU)
    fetch(0, 0);
    for (int kt = 0; kt < N; kt += AT_KT) {
        __syncthreads();                                     // the previous tile's K / V^T fragments are consumed
        commit(0);
        __syncthreads();
        fetch(0, kt + AT_KT);
        __builtin_amdgcn_sched_barrier(0);
        tile(kt);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * kb + r;
        if (q >= N) continue;
        const float inv = 1.0f / lrow[r];
        float* o = out + ((size_t)b * N + q) * C + h * 64 + j;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t * 16] = acc_o[t][r] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------------------- LayerNorm
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                        float* __restrict__ y, int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    float v[16];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        v[i] = c < C ? xr[c] : 0.0f;
        s += v[i];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        const float d = c < C ? v[i] - mean : 0.0f;
        q = fmaf(d, d, q);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    float* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        if (c < C) yr[c] = fmaf((v[i] - mean) * rstd, g[c], b[c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------- softmax over rows
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int N, float scale) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * N;
    float* yr = y + (size_t)blockIdx.x * N;
    float v[32];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = threadIdx.x + i * 256;
        v[i] = c < N ? xr[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = threadIdx.x + i * 256;
        v[i] = c < N ? __expf(v[i] - mx) : 0.0f;
        s += v[i];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < N) yr[c] = v[i] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------------------- bicubic resize
// ATen upsample_bicubic2d, align_corners = False: src = (dst + 0.5) * rscale - 0.5, rscale = 1 / scale_factor when one was given (else
// in / out - the caller passes whichever applies), taps floor(src) - 1 .. + 2 clamped to the image, cubic convolution with A = -0.75.
__device__ __forceinline__ float cc1(float x) { return ((-0.75f + 2.0f) * x - (-0.75f + 3.0f)) * x * x + 1.0f; }
__device__ __forceinline__ float cc2(float x) { return ((-0.75f * x - 5.0f * -0.75f) * x + 8.0f * -0.75f) * x - 4.0f * -0.75f; }
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    c[0] = cc2(t + 1.0f), c[1] = cc1(t), c[2] = cc1(1.0f - t), c[3] = cc2(2.0f - t);
}
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int Ho, int Wo,
                                                      float rscale_h, float rscale_w) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= Wo) return;
    const float* ip = in + (size_t)blockIdx.z * H * W;
    const float sy = ((float)oy + 0.5f) * rscale_h - 0.5f, sx = ((float)ox + 0.5f) * rscale_w - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    float cy[4], cx[4];
    cubic_coeffs(sy - fy, cy);
    cubic_coeffs(sx - fx, cx);
    const int iy = (int)fy, ix = (int)fx;
    float rows[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), H - 1);
        const float* r = ip + (size_t)yy * W;
        float v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = r[min(max(ix - 1 + b, 0), W - 1)];
        rows[a] = v[0] * cx[0] + v[1] * cx[1] + v[2] * cx[2] + v[3] * cx[3];
    }
    out[((size_t)blockIdx.z * Ho + oy) * Wo + ox] = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
}
}  // namespace

namespace {
void launch_gemm(const GemmArgs& a, int nbatch, hipStream_t s) {
    static const int big = [] { const char* e = getenv("MVS_GEMM_BIG_M"); return e ? atoi(e) : 2048; }();     // rows from which the 128-row tile is used
    static const bool fast_on = [] { const char* e = getenv("MVS_GEMM_FAST"); return !e || atoi(e) != 0; }();
    const bool aligned = a.K % 32 == 0 && a.lda % 4 == 0 && a.ldb % 4 == 0 && a.sA1 % 4 == 0 && a.sA2 % 4 == 0 && a.sB1 % 4 == 0 && a.sB2 % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0 &&
                         (int64_t)a.M * a.lda * 4 < ((int64_t)1 << 31) && (int64_t)a.N * a.ldb * 4 < ((int64_t)1 << 31);
    if (fast_on && a.a_mode == 0 && a.b_mode == 0 && a.b_kn == 0 && a.ksplit == 0 && aligned) {        // the linear layers: pipelined kernel
        if (a.M >= big) hipLaunchKernelGGL(gemm_x3_fast_kernel<2>, dim3((a.N + BN - 1) / BN, (a.M + 2 * BM - 1) / (2 * BM), nbatch), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(gemm_x3_fast_kernel<1>, dim3((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nbatch), dim3(256), 0, s, a);
        return;
    }
    if (a.M >= big) hipLaunchKernelGGL(gemm_x3_kernel<2>, dim3((a.N + BN - 1) / BN, (a.M + 2 * BM - 1) / (2 * BM), nbatch), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gemm_x3_kernel<1>, dim3((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nbatch), dim3(256), 0, s, a);
}
}  // namespace

extern "C" int mvs_gemm_x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int nb1, int nb2,
                           int64_t sA1, int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, int b_kn, int a_mode, int H, int W, int Cp,
                           float alpha, const float* scale, const float* shift, int act, const float* mul, const float* res,
                           mvs_stream_t stream) {
    MVS_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && nb1 >= 1 && nb2 >= 1, "mvs_gemm_x3: bad shape M=%d N=%d K=%d", M, N, K);
    MVS_REQUIRE((int64_t)nb1 * nb2 <= 65535 && (b_kn == 0 || b_kn == 1) && act >= 0 && act <= 3, "mvs_gemm_x3: bad batch / flags");
    MVS_REQUIRE(a_mode == 0 || ((a_mode == 1 || a_mode == 2) && H >= 1 && W >= 1 && Cp >= 8 && Cp % 8 == 0 && M == H * W &&
                                K == (a_mode == 1 ? 9 : 4) * Cp && (a_mode == 1 || nb2 == 4)),
                "mvs_gemm_x3: implicit convolution needs M = H*W, K = taps*Cp, Cp a multiple of 8 (and nb2 = 4 parity classes for the transposed form)");
    GemmArgs a{};
    a.A = A, a.B = B, a.C = C, a.scale = scale, a.shift = shift, a.mul = mul, a.res = res;
    a.sA1 = sA1, a.sA2 = sA2, a.sB1 = sB1, a.sB2 = sB2, a.sC1 = sC1, a.sC2 = sC2;
    a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldb = ldb, a.ldc = ldc, a.nb2 = nb2, a.b_kn = b_kn, a.a_mode = a_mode, a.H = H, a.W = W, a.Cp = Cp;
    a.act = act, a.alpha = alpha;
    launch_gemm(a, nb1 * nb2, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_gemm_x3");
}

// The FPN's / ViT decoder's 2-D convolutions in training (fp32 NCHW, any kernel size / stride 1 or 2 / padding; a ConvTranspose2d is the same three
// GEMMs with the roles swapped: its forward is mode 2, its data gradient mode 1): forward, data gradient and weight
// gradient as the same split-form GEMM with an implicit patch matrix (mode = 1 / 2 / 3, see GemmArgs).  Per batch item b1:
//   mode 1: y [Cout][Ho*Wo]      = w [Cout][Cin*KS*KS] . patches(x [Cin][H][W])
//   mode 2: dx [Cin][H*W]        = wT [Cin][Cout*KS*KS] . gather(dy [Cout][Ho][Wo])         (wT = w.permute(1,0,2,3), made by the caller)
//   mode 3: dWpart [b1][b2][Cout][Cin*KS*KS] = dy [Cout][Ho*Wo] . patches(x)^T over the k range of split b2 (ksplit pixels per split,
//           a multiple of 32); the caller adds the nb1 * nsplit partial matrices (mvs::launch_partials_reduce order)
extern "C" int mvs_conv2d_gemm_x3(int mode, const float* A, const float* Bmap, float* C, int nb1, int Cin, int Cout, int H, int W, int Ho, int Wo,
                                  int KS, int S, int P, int ksplit, mvs_stream_t stream) {
    MVS_REQUIRE(A && Bmap && C && nb1 >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && KS >= 1 && (S == 1 || S == 2) && P >= 0,
                "mvs_conv2d_gemm_x3: bad shape");
    MVS_REQUIRE(mode >= 1 && mode <= 3 && (mode != 3 || (ksplit >= 32 && ksplit % 32 == 0)), "mvs_conv2d_gemm_x3: mode 1..3 (mode 3 needs ksplit %% 32 == 0)");
    GemmArgs a{};
    a.A = A, a.B = Bmap, a.C = C, a.alpha = 1.0f, a.nb2 = 1;
    a.b_mode = mode, a.cH = H, a.cW = W, a.cHo = Ho, a.cWo = Wo, a.cKS = KS, a.cS = S, a.cP = P;
    const int KK = KS * KS;
    int nb2 = 1;
    if (mode == 1) {
        a.M = Cout, a.N = Ho * Wo, a.K = Cin * KK, a.lda = a.K, a.ldc = a.N;
        a.sB1 = (long long)Cin * H * W, a.sC1 = (long long)Cout * Ho * Wo;
    } else if (mode == 2) {
        a.M = Cin, a.N = H * W, a.K = Cout * KK, a.lda = a.K, a.ldc = a.N;
        a.sB1 = (long long)Cout * Ho * Wo, a.sC1 = (long long)Cin * H * W;
    } else {
        a.M = Cout, a.N = Cin * KK, a.K = Ho * Wo, a.lda = a.K, a.ldc = a.N;
        nb2 = (a.K + ksplit - 1) / ksplit;
        a.ksplit = ksplit, a.nb2 = nb2;
        a.sA1 = (long long)Cout * Ho * Wo, a.sB1 = (long long)Cin * H * W;
        a.sC2 = (long long)a.M * a.N, a.sC1 = a.sC2 * nb2;
    }
    MVS_REQUIRE((int64_t)nb1 * nb2 <= 65535, "mvs_conv2d_gemm_x3: too many batch items x splits (%d x %d)", nb1, nb2);
    launch_gemm(a, nb1 * nb2, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_conv2d_gemm_x3");
}

// out [B][N][heads*64] = softmax(scale * Q K^T) V per (image, head), flash form (the N x N matrix is never written); qkv = [B][N][3*heads*64]
// packed (q | k | v), vt = V transposed [B][heads][64][ldv] (row stride ldv >= N floats, a multiple of 4; the padding must be finite).
// head dimension 64.
extern "C" int mvs_attention_x3(const float* qkv, const float* vt, float* out, int B, int N, int heads, int head_dim, int ldv, float scale,
                                mvs_stream_t stream) {
    MVS_REQUIRE(qkv && vt && out && B >= 1 && B <= 65535 && N >= 1 && heads >= 1 && heads <= 65535, "mvs_attention_x3: bad shape");
    MVS_REQUIRE(head_dim == 64, "mvs_attention_x3: head dimension 64 only (got %d)", head_dim);
    MVS_REQUIRE(ldv >= N && ldv % 4 == 0 && (reinterpret_cast<uintptr_t>(vt) & 15) == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0,
                "mvs_attention_x3: V^T rows must be 16-byte aligned (row stride %d floats, a multiple of 4, >= N = %d; padding finite)", ldv, N);
    MVS_REQUIRE((int64_t)N * 3 * heads * 64 * 4 < ((int64_t)1 << 31) && (int64_t)64 * ldv * 4 < ((int64_t)1 << 31), "mvs_attention_x3: one image exceeds the 2 GiB buffer range");
    hipLaunchKernelGGL(attention_x3_kernel, dim3((N + 63) / 64, heads, B), dim3(256), 0, MVS_STREAM(stream), qkv, vt, out, N, heads, ldv, scale);
    return mvs::finish_launch("mvs_attention_x3");
}

extern "C" int mvs_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps, mvs_stream_t stream) {
    MVS_REQUIRE(x && gamma && beta && y && rows >= 1 && C >= 1 && C <= 1024, "mvs_layernorm: rows >= 1, 1 <= C <= 1024 (got %d)", C);
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, MVS_STREAM(stream), x, gamma, beta, y, (int)rows, C, eps);
    return mvs::finish_launch("mvs_layernorm");
}

extern "C" int mvs_softmax_rows(const float* x, float* y, int64_t rows, int N, float scale, mvs_stream_t stream) {
    MVS_REQUIRE(x && y && rows >= 1 && rows < ((int64_t)1 << 31) && N >= 1 && N <= 8192, "mvs_softmax_rows: 1 <= N <= 8192 (got %d)", N);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, MVS_STREAM(stream), x, y, N, scale);
    return mvs::finish_launch("mvs_softmax_rows");
}

extern "C" int mvs_bicubic_resize(const float* in, float* out, int planes, int H, int W, int Ho, int Wo, float rscale_h, float rscale_w,
                                  mvs_stream_t stream) {
    MVS_REQUIRE(in && out && planes >= 1 && planes <= 65535 && H >= 1 && W >= 1 && Ho >= 1 && Ho <= 65535 && Wo >= 1, "mvs_bicubic_resize: bad shape");
    hipLaunchKernelGGL(bicubic_kernel, dim3((Wo + 255) / 256, Ho, planes), dim3(256), 0, MVS_STREAM(stream), in, out, H, W, Ho, Wo, rscale_h, rscale_w);
    return mvs::finish_launch("mvs_bicubic_resize");
}
