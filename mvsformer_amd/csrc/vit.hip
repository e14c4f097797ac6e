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
#include "common.h"
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
};

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

__device__ __forceinline__ void store_split(unsigned char* tile, int row, int kseg, const float (&v)[8]) {
    const mvsx3::Split3 s = mvsx3::split3(v);
    unsigned char* d = tile + row * ROWB + kseg * 2;
    *reinterpret_cast<bf16x8*>(d) = s.h;
    *reinterpret_cast<bf16x8*>(d + TERMB) = s.m;
    *reinterpret_cast<bf16x8*>(d + 2 * TERMB) = s.l;
}

__global__ __launch_bounds__(256) void gemm_x3_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * TERMB];          // A terms h|m|l, then B terms h|m|l
    unsigned char* tA = lds;
    unsigned char* tB = lds + 3 * TERMB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b1 = blockIdx.z / a.nb2, b2 = blockIdx.z % a.nb2;
    const float* Ab = a.A + b1 * a.sA1 + b2 * a.sA2;
    const float* Bb = a.B + b1 * a.sB1 + b2 * a.sB2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int j = lane & 15, kb = lane >> 4;

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging roles: A (and B stored [n][k]): thread -> (row = tid / 4, 8 k at (tid % 4) * 8); B stored [k][n]: thread -> (k = tid / 8, 8 n)
    const int srow = tid >> 2, skseg = (tid & 3) * 8;
    const int tk = tid >> 3, tn = (tid & 7) * 8;
    float pa[8], pb[8];
    auto fetch = [&](int k0) {
        load_a8(a, Ab, m0 + srow, k0 + skseg, b2, pa);
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[e] = 0.0f;
        if (a.b_kn == 0) {
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
        store_split(tA, srow, skseg, pa);
        if (a.b_kn == 0) {
            store_split(tB, srow, skseg, pb);
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

    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        __syncthreads();                                     // the previous step's fragment reads are done
        commit();
        __syncthreads();
        if (k0 + BK < a.K) fetch(k0 + BK);                   // the next tile's loads travel under this step's MFMAs
        const unsigned char* ap = tA + (wave * 16 + j) * ROWB + kb * 16;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap), am = *reinterpret_cast<const bf16x8*>(ap + TERMB),
                     al = *reinterpret_cast<const bf16x8*>(ap + 2 * TERMB);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned char* bp = tB + (t * 16 + j) * ROWB + kb * 16;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp), bm = *reinterpret_cast<const bf16x8*>(bp + TERMB),
                         bl = *reinterpret_cast<const bf16x8*>(bp + 2 * TERMB);
            acc[t] = mvsx3::mfma6(ah, am, al, bh, bm, bl, acc[t]);
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
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + kb * 4 + r;
            if (m >= a.M) continue;
            float v = acc[t][r] * a.alpha;
            v = fmaf(v, sc, sh);
            if (a.act == 1) v = gelu_erf(v);
            else if (a.act == 2) v = v / (1.0f + __expf(-v));
            const size_t o = (size_t)m * a.ldc + n;
            if (a.mul) v *= a.mul[eoff + o];
            if (a.res) v += a.res[eoff + o];
            Cb[o] = v;
        }
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

extern "C" int mvs_gemm_x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int nb1, int nb2,
                           int64_t sA1, int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, int b_kn, int a_mode, int H, int W, int Cp,
                           float alpha, const float* scale, const float* shift, int act, const float* mul, const float* res,
                           mvs_stream_t stream) {
    MVS_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && nb1 >= 1 && nb2 >= 1, "mvs_gemm_x3: bad shape M=%d N=%d K=%d", M, N, K);
    MVS_REQUIRE((int64_t)nb1 * nb2 <= 65535 && (b_kn == 0 || b_kn == 1) && act >= 0 && act <= 2, "mvs_gemm_x3: bad batch / flags");
    MVS_REQUIRE(a_mode == 0 || ((a_mode == 1 || a_mode == 2) && H >= 1 && W >= 1 && Cp >= 8 && Cp % 8 == 0 && M == H * W &&
                                K == (a_mode == 1 ? 9 : 4) * Cp && (a_mode == 1 || nb2 == 4)),
                "mvs_gemm_x3: implicit convolution needs M = H*W, K = taps*Cp, Cp a multiple of 8 (and nb2 = 4 parity classes for the transposed form)");
    GemmArgs a{};
    a.A = A, a.B = B, a.C = C, a.scale = scale, a.shift = shift, a.mul = mul, a.res = res;
    a.sA1 = sA1, a.sA2 = sA2, a.sB1 = sB1, a.sB2 = sB2, a.sC1 = sC1, a.sC2 = sC2;
    a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldb = ldb, a.ldc = ldc, a.nb2 = nb2, a.b_kn = b_kn, a.a_mode = a_mode, a.H = H, a.W = W, a.Cp = Cp;
    a.act = act, a.alpha = alpha;
    hipLaunchKernelGGL(gemm_x3_kernel, dim3((N + BN - 1) / BN, (M + BM - 1) / BM, nb1 * nb2), dim3(256), 0, MVS_STREAM(stream), a);
    return mvs::finish_launch("mvs_gemm_x3");
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
