// FPN encoder, the layers below full resolution: the six 3x3 stride-1 layers (and, further down, the three stride-2 ones) (models/module.py:208-240: conv10/11 16 -> 16, conv20/21 32 -> 32, conv30/31 64 -> 64;
// ``Conv2d`` = conv (no bias) -> BatchNorm2d -> leaky_relu(0.1), module.py:40-73) in the THREE-TERM BF16 SPLIT form (split3.h): fp32 NCHW in and out,
// fp32-equivalent.  conv2d_kernel (conv2d.hip) runs them on v_mfma_f32_16x16x4_f32 at 84-104 TFLOP/s = 0.55-0.66 of THAT pipe's peak; six
// v_mfma_f32_16x16x32_bf16 per fp32-equivalent K = 32 step have 2.65x its rate.
//
// Block = 4 x 32 output pixels, 4 wavefronts (one output row each, two 16-column halves), the input channels in chunks of 16: the thread that owns
// a halo pixel loads its 16 channels, splits them and writes two octets of [term][octet][pixel][8 bf16] (a K block of the B operand = one
// ds_read_b128); the next chunk's loads fly during the MFMAs.  M = 16 output channels per tile, K = 9 taps x 2 octets = 18 K blocks = 5 steps per
// chunk; the weights arrive pre-split in fragment order (BatchNorm scale folded in) from L1 / L2, one step ahead.
#include "conv_common.h"
#include "split3.h"

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;

constexpr int TH = 4, TW = 32, HR = TH + 2, HC = TW + 2, NPIX = HR * HC;   // 204 halo pixels <= 256 threads
constexpr int CS = 208;                                                    // pixel slots of the split tile
constexpr int OCTB = CS * 16, TERMB = 2 * OCTB, STEPS = 5;

__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], c, 0, 0, 0);
    return c;
}

// prepared[(((chunk * STEPS + step) * NT + nt) * 3 + term)][lane][8]: the MFMA A operand, lane = kb * 16 + m: output channel 16 nt + m,
// K block t = 4 step + kb = (tap = t / 2, octet = t % 2) (t >= 18: zero), input channel 16 chunk + 8 octet + e; scale[co] multiplied in
__global__ void conv2d_x3s_prepare_kernel(const float* __restrict__ w /*[Cout,Cin,3,3]*/, const float* __restrict__ scale, int Cin, int Cout,
                                          bf16x8* __restrict__ out) {
    const int NT = Cout / 16, total = (Cin / 16) * STEPS * NT * 3 * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, term = (idx >> 6) % 3, nt = (idx / 192) % NT, step = (idx / (192 * NT)) % STEPS, chunk = idx / (192 * NT * STEPS);
    const int m = lane & 15, kb = lane >> 4, co = 16 * nt + m, t = 4 * step + kb, tap = t >> 1, oct = t & 1;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 16 * chunk + 8 * oct + e;
        const float f = t < 18 ? w[((co * Cin + c) * 3 + tap / 3) * 3 + tap % 3] * scale[co] : 0.0f;
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv2d_x3s_kernel(const float* __restrict__ x /*[N,CIN,H,W]*/, const bf16x8* __restrict__ wprep,
                                                            const float* __restrict__ shift, int H, int W, float slope, float* __restrict__ y /*[N,COUT,H,W]*/) {
    constexpr int NT = COUT / 16, NCH = CIN / 16;
    __shared__ __attribute__((aligned(256))) unsigned char s_b[3 * TERMB];

    unsigned bx, by, bz;
    xcd_block_coords(bx, by, bz);
    const int x0 = (int)bx * TW, y0 = (int)by * TH, img = (int)bz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;

    // ---- this thread's halo pixel: 16 channel loads per chunk (zero outside the image = the convolution's padding) ----
    const int p = tid, gy = y0 - 1 + p / HC, gx = x0 - 1 + p % HC;
    const bool inimg = p < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const rsrc_t rx = make_rsrc(x + (size_t)img * CIN * H * W, (unsigned)(CIN * H * W) * 4u);
    const unsigned poff = inimg ? (unsigned)(gy * W + gx) * 4u : OOB, chb = (unsigned)(H * W) * 4u;
    float xr[16];
    auto prefetch = [&](int cc) {
#pragma unroll
        for (int c = 0; c < 16; ++c) xr[c] = buf_load(rx, poff, (unsigned)(cc * 16 + c) * chb);
    };
    auto commit = [&]() {
        if (p < NPIX) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 th, tm, tl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned xh, xm, xl;
                    mvsx3::split3_pair<true>(xr[8 * o + 2 * e], xr[8 * o + 2 * e + 1], xh, xm, xl);
                    th[e] = xh; tm[e] = xm; tl[e] = xl;
                }
                unsigned char* dst = s_b + o * OCTB + p * 16;
                *reinterpret_cast<u32x4*>(dst) = th;
                *reinterpret_cast<u32x4*>(dst + TERMB) = tm;
                *reinterpret_cast<u32x4*>(dst + 2 * TERMB) = tl;
            }
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B operand of this lane's K block per step: t = 4 s + kb = (tap, octet): octet * OCTB + ((wv + kh) * HC + kw + n) * 16 (+ 256 for the second half)
    unsigned boff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int t = min(4 * s + kb, 17), tap = t >> 1;
        boff[s] = (unsigned)((t & 1) * OCTB + ((wv + tap / 3) * HC + tap % 3 + n) * 16);
    }

    prefetch(0);
    for (int cc = 0; cc < NCH; ++cc) {
        if (cc) __syncthreads();                            // the previous chunk's MFMA phase has finished reading LDS
        commit();
        __syncthreads();
        if (cc + 1 < NCH) prefetch(cc + 1);                 // in flight during this chunk's MFMAs
        const bf16x8* wc = wprep + (size_t)cc * STEPS * NT * 3 * 64 + lane;
        bf16x8 wa[2][NT][3];
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int t = 0; t < 3; ++t) wa[0][q][t] = wc[(q * 3 + t) * 64];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (s + 1 < STEPS) {
#pragma unroll
                for (int q = 0; q < NT; ++q)
#pragma unroll
                    for (int t = 0; t < 3; ++t) wa[(s + 1) & 1][q][t] = wc[(((s + 1) * NT + q) * 3 + t) * 64];
            }
            bf16x8 x0f[3], x1f[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                x0f[t] = *reinterpret_cast<const bf16x8*>(s_b + t * TERMB + boff[s]);
                x1f[t] = *reinterpret_cast<const bf16x8*>(s_b + t * TERMB + boff[s] + 256);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                acc[0][q] = mfma6(wa[s & 1][q], x0f, acc[0][q]);
                acc[1][q] = mfma6(wa[s & 1][q], x1f, acc[1][q]);
            }
        }
    }

    // ---- epilogue: BatchNorm shift (the scale sits in the weights) + leaky ReLU; D[m = channel][n = column]: this lane's 4 channels of a pixel ----
    const int yy = y0 + wv;
    float* y_img = y + (size_t)img * COUT * H * W;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int co = 16 * q + 4 * kb;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + co);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int xx = x0 + 16 * t + n;
            if (yy < H && xx < W) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[t][q][r] + sh[r];
                    y_img[(size_t)(co + r) * H * W + (size_t)yy * W + xx] = v > 0.0f ? v : v * slope;
                }
            }
        }
    }
}

// ---- the three stride-2 layers (downsample1 8 -> 16 k5, downsample2 16 -> 32 k5, downsample3 32 -> 64 k3; padding K / 2) ----------------------------
// Same tile and roles; the B operand of 16 consecutive OUTPUT pixels is every second input pixel, so the halo tile's columns are de-interleaved by parity
// at staging ([term][row][column parity][column / 2][8 bf16]: a tap's 16 pixels are 256 contiguous bytes again); input channels in chunks of 8 (one octet:
// the 11 x 67 halo of a 5 x 5 layer is 36 KB per chunk); K blocks = the K^2 taps of the chunk's octet.  downsample1 reads conv01's channel-last companion
// (two 16-byte loads per pixel).
template <int KS>
struct S2 {
    static constexpr int P = KS / 2, IR = 2 * TH + KS - 2, IC = 2 * TW + KS - 2, PS = 34;      // halo rows / columns, slots per parity plane
    static constexpr int NSLOT = IR * 2 * PS, TERM2 = NSLOT * 16;
    static constexpr int STEPS2 = (KS * KS + 3) / 4;                                            // 7 | 3
    static constexpr int PPT = (IR * IC + 255) / 256;                                           // halo pixels per thread (3)
};

__global__ void conv2d_x3s2_prepare_kernel(const float* __restrict__ w /*[Cout,Cin,K,K]*/, const float* __restrict__ scale, int Cin, int Cout, int KS,
                                           bf16x8* __restrict__ out) {
    const int NT = Cout / 16, ST = (KS * KS + 3) / 4, total = (Cin / 8) * ST * NT * 3 * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, term = (idx >> 6) % 3, nt = (idx / 192) % NT, step = (idx / (192 * NT)) % ST, chunk = idx / (192 * NT * ST);
    const int m = lane & 15, kb = lane >> 4, co = 16 * nt + m, t = 4 * step + kb;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 8 * chunk + e;
        const float f = t < KS * KS ? w[((co * Cin + c) * KS + t / KS) * KS + t % KS] * scale[co] : 0.0f;
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

template <int CIN, int COUT, int KS>
__global__ __launch_bounds__(256, 2) void conv2d_x3s2_kernel(const float* __restrict__ x /*[N,CIN,H,W] | [N,H,W,8] with in_nhwc*/, const bf16x8* __restrict__ wprep,
                                                             const float* __restrict__ shift, int H, int W, int Ho, int Wo, float slope, int in_nhwc,
                                                             float* __restrict__ y /*[N,COUT,Ho,Wo]*/) {
    using G = S2<KS>;
    constexpr int NT = COUT / 16, NCH = CIN / 8, ST = G::STEPS2;
    extern __shared__ __attribute__((aligned(256))) unsigned char s_b[];     // [term][row][parity][PS][16 B]

    unsigned bx, by, bz;
    xcd_block_coords(bx, by, bz);
    const int x0 = (int)bx * TW, y0 = (int)by * TH, img = (int)bz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;
    const int iy0 = 2 * y0 - G::P, ix0 = 2 * x0 - G::P;

    const rsrc_t rx = make_rsrc(x + (size_t)img * CIN * H * W, (unsigned)(CIN * H * W) * 4u);
    const unsigned chb = (unsigned)(H * W) * 4u;
    unsigned poff[G::PPT], slot[G::PPT];
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int u = tid + 256 * i, r = u / G::IC, c = u % G::IC, gy = iy0 + r, gx = ix0 + c;
        const bool ok = u < G::IR * G::IC;
        poff[i] = (ok && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)(gy * W + gx) * 4u : OOB;
        slot[i] = ok ? (unsigned)(((r * 2 + (c & 1)) * G::PS + (c >> 1)) * 16) : 0xffffffffu;
    }
    float xr[G::PPT][8];
    auto prefetch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            if (CIN == 8 && in_nhwc) {
                const unsigned o8 = (poff[i] & OOB) ? OOB : poff[i] * 8u;
                const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o8, 0, 0));
                const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o8, 16, 0));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    xr[i][c] = a[c];
                    xr[i][4 + c] = b[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) xr[i][c] = buf_load(rx, poff[i], (unsigned)(cc * 8 + c) * chb);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            if (slot[i] == 0xffffffffu) continue;
            const mvsx3::Split3 sp = mvsx3::split3(xr[i]);
            unsigned char* dst = s_b + slot[i];
            *reinterpret_cast<bf16x8*>(dst) = sp.h;
            *reinterpret_cast<bf16x8*>(dst + G::TERM2) = sp.m;
            *reinterpret_cast<bf16x8*>(dst + 2 * G::TERM2) = sp.l;
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B operand of this lane's K block per step: tap t = 4 s + kb = (kh, kw): row 2 wv + kh, parity kw & 1, slot n + (kw >> 1) (+ 16 for the second half)
    unsigned boff[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int t = min(4 * s + kb, KS * KS - 1), kh = t / KS, kw = t % KS;
        boff[s] = (unsigned)((((2 * wv + kh) * 2 + (kw & 1)) * G::PS + n + (kw >> 1)) * 16);
    }

    prefetch(0);
    for (int cc = 0; cc < NCH; ++cc) {
        if (cc) __syncthreads();
        commit();
        __syncthreads();
        if (cc + 1 < NCH) prefetch(cc + 1);
        const bf16x8* wc = wprep + (size_t)cc * ST * NT * 3 * 64 + lane;
        bf16x8 wa[2][NT][3];
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int t = 0; t < 3; ++t) wa[0][q][t] = wc[(q * 3 + t) * 64];
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            if (s + 1 < ST) {
#pragma unroll
                for (int q = 0; q < NT; ++q)
#pragma unroll
                    for (int t = 0; t < 3; ++t) wa[(s + 1) & 1][q][t] = wc[(((s + 1) * NT + q) * 3 + t) * 64];
            }
            bf16x8 x0f[3], x1f[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                x0f[t] = *reinterpret_cast<const bf16x8*>(s_b + t * G::TERM2 + boff[s]);
                x1f[t] = *reinterpret_cast<const bf16x8*>(s_b + t * G::TERM2 + boff[s] + 256);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                acc[0][q] = mfma6(wa[s & 1][q], x0f, acc[0][q]);
                acc[1][q] = mfma6(wa[s & 1][q], x1f, acc[1][q]);
            }
        }
    }

    const int yy = y0 + wv;
    float* y_img = y + (size_t)img * COUT * Ho * Wo;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int co = 16 * q + 4 * kb;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + co);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int xx = x0 + 16 * t + n;
            if (yy < Ho && xx < Wo) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[t][q][r] + sh[r];
                    y_img[(size_t)(co + r) * Ho * Wo + (size_t)yy * Wo + xx] = v > 0.0f ? v : v * slope;
                }
            }
        }
    }
}

template <int CIN, int COUT, int KS>
int launch_s2(const float* x, int in_nhwc, const void* prepared, const float* shift, int N, int H, int W, float slope, float* y, hipStream_t s) {
    using G = S2<KS>;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const dim3 grid(mvs::ceil_div(Wo, TW), mvs::ceil_div(Ho, TH), N);
    constexpr int LDS = 3 * G::TERM2;
    hipLaunchKernelGGL((conv2d_x3s2_kernel<CIN, COUT, KS>), grid, dim3(256), LDS, s, x, static_cast<const bf16x8*>(prepared), shift, H, W, Ho, Wo, slope, in_nhwc, y);
    return mvs::finish_launch("mvs_conv2d_x3s_bn_lrelu");
}

template <int CIN, int COUT>
int launch(const float* x, const void* prepared, const float* shift, int N, int H, int W, float slope, float* y, hipStream_t s) {
    const dim3 grid(mvs::ceil_div(W, TW), mvs::ceil_div(H, TH), N);
    hipLaunchKernelGGL((conv2d_x3s_kernel<CIN, COUT>), grid, dim3(256), 0, s, x, static_cast<const bf16x8*>(prepared), shift, H, W, slope, y);
    return mvs::finish_launch("mvs_conv2d_x3s_bn_lrelu");
}

}  // namespace

static bool is_s1(int Cin, int Cout, int KS, int stride) { return KS == 3 && stride == 1 && Cin == Cout && (Cin == 16 || Cin == 32 || Cin == 64); }
static bool is_s2(int Cin, int Cout, int KS, int stride) {
    return stride == 2 && ((Cin == 8 && Cout == 16 && KS == 5) || (Cin == 16 && Cout == 32 && KS == 5) || (Cin == 32 && Cout == 64 && KS == 3));
}

extern "C" int mvs_conv2d_x3s_supported(int Cin, int Cout, int KS, int stride) { return (is_s1(Cin, Cout, KS, stride) || is_s2(Cin, Cout, KS, stride)) ? 1 : 0; }

extern "C" int64_t mvs_conv2d_x3s_prepared_bytes(int Cin, int Cout, int KS, int stride) {
    if (is_s1(Cin, Cout, KS, stride)) return (int64_t)(Cin / 16) * STEPS * (Cout / 16) * 3 * 64 * 16;
    if (is_s2(Cin, Cout, KS, stride)) return (int64_t)(Cin / 8) * ((KS * KS + 3) / 4) * (Cout / 16) * 3 * 64 * 16;
    return -1;
}

extern "C" int mvs_conv2d_x3s_prepare(const float* w, const float* scale, int Cin, int Cout, int KS, int stride, void* prepared, mvs_stream_t stream) {
    MVS_REQUIRE(w && scale && prepared, "mvs_conv2d_x3s_prepare: null pointer");
    MVS_REQUIRE(mvs_conv2d_x3s_supported(Cin, Cout, KS, stride), "mvs_conv2d_x3s_prepare: (Cin,Cout,K,stride)=(%d,%d,%d,%d) is not a layer of the FPN encoder below full resolution",
                Cin, Cout, KS, stride);
    const int total = (int)(mvs_conv2d_x3s_prepared_bytes(Cin, Cout, KS, stride) / 16);
    if (stride == 1)
        hipLaunchKernelGGL(conv2d_x3s_prepare_kernel, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w, scale, Cin, Cout, static_cast<bf16x8*>(prepared));
    else
        hipLaunchKernelGGL(conv2d_x3s2_prepare_kernel, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w, scale, Cin, Cout, KS, static_cast<bf16x8*>(prepared));
    return mvs::finish_launch("mvs_conv2d_x3s_prepare");
}

extern "C" int mvs_conv2d_x3s_bn_lrelu(const float* x, int x_nhwc, const void* prepared, const float* shift, int N, int Cin, int Cout, int KS, int stride, int H,
                                       int W, float slope, float* y, mvs_stream_t stream) {
    MVS_REQUIRE(x && prepared && shift && y, "mvs_conv2d_x3s_bn_lrelu: null pointer");
    MVS_REQUIRE(mvs_conv2d_x3s_supported(Cin, Cout, KS, stride), "mvs_conv2d_x3s_bn_lrelu: (Cin,Cout,K,stride)=(%d,%d,%d,%d) is not a layer of the FPN encoder below full resolution",
                Cin, Cout, KS, stride);
    MVS_REQUIRE(!x_nhwc || (Cin == 8 && stride == 2), "mvs_conv2d_x3s_bn_lrelu: a channel-last input is read by the 8-channel stride-2 layer only");
    MVS_REQUIRE(N >= 1 && N <= 65535 && H >= 1 && W >= 1 && (int64_t)H <= 4 * 65535, "mvs_conv2d_x3s_bn_lrelu: bad shape N=%d H=%d W=%d", N, H, W);
    // 32-bit byte offsets inside one image: Cin * H * W * 4 < 2^31; the channel-last 8-channel input scales a pixel offset by 32: H * W * 32 < 2^28 keeps it exact
    MVS_REQUIRE((int64_t)Cin * H * W * 4 < ((int64_t)1 << 31) && (!x_nhwc || (int64_t)H * W * 32 < ((int64_t)1 << 28)),
                "mvs_conv2d_x3s_bn_lrelu: one image is too large for 32-bit offsets");
    hipStream_t s = MVS_STREAM(stream);
    if (stride == 1) {
        if (Cin == 16) return launch<16, 16>(x, prepared, shift, N, H, W, slope, y, s);
        if (Cin == 32) return launch<32, 32>(x, prepared, shift, N, H, W, slope, y, s);
        return launch<64, 64>(x, prepared, shift, N, H, W, slope, y, s);
    }
    if (Cin == 8) return launch_s2<8, 16, 5>(x, x_nhwc ? 1 : 0, prepared, shift, N, H, W, slope, y, s);
    if (Cin == 16) return launch_s2<16, 32, 5>(x, 0, prepared, shift, N, H, W, slope, y, s);
    return launch_s2<32, 64, 3>(x, 0, prepared, shift, N, H, W, slope, y, s);
}
