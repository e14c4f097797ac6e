// FPN encoder layers (models/module.py:40-73 ``Conv2d`` = conv (no bias) -> BatchNorm2d -> leaky_relu(0.1), as stacked by
// ``FPNEncoder``, module.py:208-240): one fused kernel per layer, eval-mode BatchNorm folded into (scale, shift), NCHW in and out.
//
// Implicit GEMM on v_mfma_f32_16x16x4_f32 with the fragment conventions of conv3d.hip / fpn.hip: M = 16 output pixels along x,
// K = taps x 4 input channels, N = output channels.  Block = 4 x 32 output pixels, 4 wavefronts; the input tile with its halo and the
// packed weights of 8 input channels at a time are staged in LDS (channel stride == 16 mod 32 for unit-stride fragment reads, odd for
// the stride-2 layers so the two k-halves of a 32-lane access group fall on disjoint banks).  Cout = 8 (conv00, conv01: the two
// full-resolution layers, a third of the encoder's FLOPs) would fill half of the N tile, so N = (2 output rows) x (8 channels): the
// KS+1 input rows feeding an output row pair are each multiplied against a weight matrix holding tap row j for the upper output row
// and j-1 for the lower one - KS/(KS+1) of the MFMA work is useful instead of 1/2.
//
// Layer shapes of the shipped feat_chs = [8,16,32,64]: (Cin,Cout,K,stride) = (3,8,7,1) (8,8,5,1) (8,16,5,2) (16,16,3,1) (16,32,5,2)
// (32,32,3,1) (32,64,3,2) (64,64,3,1); padding = K/2.  Algorithmic FLOPs per output pixel: 2*K*K*Cin*Cout.
#include "conv_common.h"

namespace {
using namespace mvsconv;

constexpr int TH = 4, TW = 32;               // output tile

template <int CIN, int COUT, int KS, int S>
struct Cfg {
    static constexpr int CP = (CIN + 3) / 4 * 4;                 // input channels padded to the MFMA k-step
    static constexpr int CC = CP < 8 ? CP : 8;                   // input channels per LDS chunk
    static constexpr bool ROWS2 = (COUT == 8);
    static constexpr int NT = ROWS2 ? 1 : (COUT + 15) / 16;
    static constexpr int NP = np_of(NT);
    static constexpr int T = ROWS2 ? (KS + 1) * KS : KS * KS;    // weight matrices per 4 input channels
    static constexpr int IR = (TH - 1) * S + KS, IC = (TW - 1) * S + KS;
    static constexpr int CS = pad_cs(IR * IC, S);
    static constexpr int WCH = (CC / 4) * T * 4 * NP;            // packed weight floats per chunk
    static_assert(!ROWS2 || S == 1, "the row-pair form is built for stride 1");
    static_assert(NT == 1 || NT == 2 || NT == 4, "np_of");
};

__host__ __device__ inline int packed_taps(int Cout, int KS) { return Cout == 8 ? (KS + 1) * KS : KS * KS; }
__host__ __device__ inline int packed_np(int Cout) { return np_of(Cout == 8 ? 1 : (Cout + 15) / 16); }

// packed image: [slab = cin/4][tap][cin%4][NP], zero-padded in cin and n
__global__ void conv2d_pack_kernel(const float* __restrict__ w /*[Cout,Cin,KS,KS]*/, int Cin, int Cout, int KS, float* __restrict__ out) {
    const int T = packed_taps(Cout, KS), NP = packed_np(Cout), CP = (Cin + 3) / 4 * 4;
    const int total = (CP / 4) * T * 4 * NP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int n = idx % NP, kk = (idx / NP) % 4, tap = (idx / (4 * NP)) % T, slab = idx / (4 * NP * T);
        const int c = slab * 4 + kk;
        float v = 0.0f;
        if (c < Cin) {
            if (Cout == 8) {                 // tap = j*KS + kx over the KS+1 input rows j of an output row pair; n = h*8 + co
                const int j = tap / KS, kx = tap % KS, h = n >> 3, co = n & 7, ky = j - h;
                if (n < 16 && ky >= 0 && ky < KS) v = w[((size_t)(co * Cin + c) * KS + ky) * KS + kx];
            } else if (n < Cout) {
                v = w[(size_t)(n * Cin + c) * KS * KS + tap];
            }
        }
        out[idx] = v;
    }
}

template <int CIN, int COUT, int KS, int S>
__global__ __launch_bounds__(256) void conv2d_kernel(const float* __restrict__ x /*[N,CIN,H,W]*/, const float* __restrict__ wp,
                                                     const float* __restrict__ scale, const float* __restrict__ shift, int H, int W,
                                                     int Ho, int Wo, float slope, float* __restrict__ y /*[N,COUT,Ho,Wo]*/) {
    using C = Cfg<CIN, COUT, KS, S>;
    constexpr int NT = C::NT, NP = C::NP, T = C::T, IC = C::IC, CS = C::CS, CC = C::CC, WCH = C::WCH;
    constexpr bool ROWS2 = C::ROWS2;
    constexpr int MT = ROWS2 ? 1 : 2;
    constexpr int P = KS / 2;
    __shared__ __attribute__((aligned(16))) float s_in[CC * CS];
    __shared__ __attribute__((aligned(16))) float s_w[WCH];

    unsigned bx, by, bz;
    xcd_block_coords(bx, by, bz);
    const int x0 = (int)bx * TW, y0 = (int)by * TH, img = (int)bz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int iy0 = y0 * S - P, ix0 = x0 * S - P;            // input coordinates of the tile's first halo element
    const float* x_img = x + (size_t)img * CIN * H * W;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging roles: element i of this thread is tile slot tid + i*256; its LDS slot and its in-image offset do not depend on
    // the chunk, so both are computed once.  A chunk's loads are all issued before any is consumed, and the NEXT chunk's loads are in
    // flight during the current chunk's MFMA phase (a rolled load -> wait -> store loop serializes one global latency per element).
    constexpr int NIT = (CC * C::IR * IC + 255) / 256, NWV = (WCH / 4 + 255) / 256;
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + i * 256;
        const int c = idx / (C::IR * IC), r = idx % (C::IR * IC);
        const int gy = iy0 + r / IC, gx = ix0 + r % IC;
        const bool slot = idx < CC * C::IR * IC;
        loff[i] = slot ? c * CS + r : -1;
        goff[i] = (slot && c < CIN && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (c * H + gy) * W + gx : -1;
    }
    float sreg[NIT];
    f32x4 wreg[NWV];
    auto prefetch = [&](int ch) {
        const float* xc = x_img + (size_t)ch * CC * H * W;
#pragma unroll
        for (int i = 0; i < NIT; ++i) sreg[i] = goff[i] >= 0 ? xc[(unsigned)goff[i]] : 0.0f;
        const f32x4* src = reinterpret_cast<const f32x4*>(wp + (size_t)ch * WCH);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            wreg[i] = idx < WCH / 4 ? src[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (loff[i] >= 0) s_in[loff[i]] = sreg[i];
        f32x4* dst = reinterpret_cast<f32x4*>(s_w);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            if (idx < WCH / 4) dst[idx] = wreg[i];
        }
    };

    prefetch(0);
    for (int ch = 0; ch < C::CP / CC; ++ch) {
        if (ch) __syncthreads();                            // the previous chunk's MFMA phase has finished reading LDS
        commit();
        __syncthreads();
        if (ch + 1 < C::CP / CC) prefetch(ch + 1);
#pragma unroll
        for (int ks = 0; ks < CC / 4; ++ks) {
            const float* abase = s_in + (ks * 4 + kk) * CS + S * i16;
            const float* bbase = s_w + (ks * T * 4 + kk) * NP + i16;
            if constexpr (ROWS2) {
                const int pq = wv >> 1, mt = wv & 1;        // output row pair, 16-column half
#pragma unroll
                for (int j = 0; j < KS + 1; ++j)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
                        acc[0][0] = mfma4(abase[(2 * pq + j) * IC + mt * 16 + kx], bbase[(j * KS + kx) * 4 * NP], acc[0][0]);
            } else {
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        float a[MT];
#pragma unroll
                        for (int t = 0; t < MT; ++t) a[t] = abase[(wv * S + ky) * IC + t * 16 * S + kx];
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const float b = bbase[(ky * KS + kx) * 4 * NP + n * 16];
#pragma unroll
                            for (int t = 0; t < MT; ++t) acc[t][n] = mfma4(a[t], b, acc[t][n]);
                        }
                    }
            }
        }
    }

    // ---- epilogue: folded BatchNorm + leaky ReLU; a lane holds 4 consecutive pixels of one channel -> 16-byte NCHW stores ----
    float* y_img = y + (size_t)img * COUT * Ho * Wo;
    const bool vec = (Wo % 4) == 0;
    auto store4 = [&](f32x4 a, int co, int yy, int xx) {
        if (yy >= Ho || xx >= Wo) return;
        const float sc = scale[co], sh = shift[co];
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = fmaf(a[r], sc, sh);
            o[r] = v > 0.0f ? v : v * slope;
        }
        float* dst = y_img + (unsigned)((co * Ho + yy) * Wo + xx);
        if (vec) {
            *reinterpret_cast<f32x4*>(dst) = o;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (xx + r < Wo) dst[r] = o[r];
        }
    };
    if constexpr (ROWS2) {
        const int pq = wv >> 1, mt = wv & 1;
        store4(acc[0][0], i16 & 7, y0 + 2 * pq + (i16 >> 3), x0 + mt * 16 + 4 * kk);
    } else {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                if (n * 16 + i16 < COUT) store4(acc[t][n], n * 16 + i16, y0 + wv, x0 + t * 16 + 4 * kk);
    }
}

template <int CIN, int COUT, int KS, int S>
void launch(const float* x, const float* wp, const float* scale, const float* shift, int N, int H, int W, int Ho, int Wo, float slope,
            float* y, hipStream_t s) {
    const dim3 grid(mvs::ceil_div(Wo, TW), mvs::ceil_div(Ho, TH), N);
    hipLaunchKernelGGL((conv2d_kernel<CIN, COUT, KS, S>), grid, dim3(256), 0, s, x, wp, scale, shift, H, W, Ho, Wo, slope, y);
}

bool supported(int Cin, int Cout, int KS, int S) {
    const int k[8][4] = {{3, 8, 7, 1}, {8, 8, 5, 1}, {8, 16, 5, 2}, {16, 16, 3, 1}, {16, 32, 5, 2}, {32, 32, 3, 1}, {32, 64, 3, 2}, {64, 64, 3, 1}};
    for (auto& e : k)
        if (e[0] == Cin && e[1] == Cout && e[2] == KS && e[3] == S) return true;
    return false;
}
}  // namespace

extern "C" int64_t mvs_conv2d_packed_floats(int Cin, int Cout, int KS) {
    if (Cin < 1 || Cout < 8 || Cout > 64 || Cout % 8 || (KS != 3 && KS != 5 && KS != 7)) return -1;
    return (int64_t)((Cin + 3) / 4) * packed_taps(Cout, KS) * 4 * packed_np(Cout);
}

extern "C" int mvs_conv2d_pack_weights(const float* w, int Cin, int Cout, int KS, float* packed, mvs_stream_t stream) {
    MVS_REQUIRE(w && packed, "mvs_conv2d_pack_weights: null pointer");
    const int64_t total = mvs_conv2d_packed_floats(Cin, Cout, KS);
    MVS_REQUIRE(total > 0, "mvs_conv2d_pack_weights: unsupported Cin=%d Cout=%d K=%d", Cin, Cout, KS);
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3(mvs::ceil_div((int)total, 256)), dim3(256), 0, MVS_STREAM(stream), w, Cin, Cout, KS, packed);
    return mvs::finish_launch("mvs_conv2d_pack_weights");
}

extern "C" int mvs_conv2d_bn_lrelu(const float* x, const float* packed, const float* scale, const float* shift, int N, int Cin, int Cout,
                                   int KS, int stride, int H, int W, float slope, float* y, mvs_stream_t stream) {
    MVS_REQUIRE(x && packed && scale && shift && y, "mvs_conv2d_bn_lrelu: null pointer");
    MVS_REQUIRE(supported(Cin, Cout, KS, stride),
                "mvs_conv2d_bn_lrelu: (Cin,Cout,K,stride)=(%d,%d,%d,%d) is not one of the FPN encoder's layer shapes", Cin, Cout, KS, stride);
    MVS_REQUIRE(N >= 1 && N <= 65535 && H >= 1 && W >= 1 && (int64_t)H <= 4 * 65535, "mvs_conv2d_bn_lrelu: bad shape N=%d H=%d W=%d", N, H, W);
    MVS_REQUIRE((int64_t)(Cin > Cout ? Cin : Cout) * H * W < ((int64_t)1 << 31), "mvs_conv2d_bn_lrelu: one image exceeds 2^31 elements");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    hipStream_t s = MVS_STREAM(stream);
#define MVS_C2D(ci, co, k, st)                                                                   \
    if (Cin == ci && Cout == co && KS == k && stride == st) {                                     \
        launch<ci, co, k, st>(x, packed, scale, shift, N, H, W, Ho, Wo, slope, y, s);             \
        return mvs::finish_launch("mvs_conv2d_bn_lrelu");                                         \
    }
    MVS_C2D(3, 8, 7, 1)
    MVS_C2D(8, 8, 5, 1)
    MVS_C2D(8, 16, 5, 2)
    MVS_C2D(16, 16, 3, 1)
    MVS_C2D(16, 32, 5, 2)
    MVS_C2D(32, 32, 3, 1)
    MVS_C2D(32, 64, 3, 2)
    MVS_C2D(64, 64, 3, 1)
#undef MVS_C2D
    return MVS_EINVAL;
}
