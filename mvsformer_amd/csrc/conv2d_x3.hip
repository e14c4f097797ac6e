// FPN encoder, the two full-resolution layers (models/module.py:208-240: conv00 = Conv2d(3, 8, 7, 1, padding 3), conv01 = Conv2d(8, 8, 5, 1,
// padding 2); ``Conv2d`` = conv (no bias) -> BatchNorm2d -> leaky_relu(0.1), module.py:40-73) in the THREE-TERM BF16 SPLIT form of
// conv3d_x3.hip / vis_net_x3.hip / fpn_x3.hip: fp32 NCHW in and out, every product as six v_mfma_f32_16x16x32_bf16 of the exact splits
// x = h + m + l with fp32 accumulation - fp32-equivalent.  They replace conv2d_kernel<3,8,7,1> / <8,8,5,1> (conv2d.hip: fp32 matrix
// cores, 0.32 + 0.39 ms of the encoder's 1.8 ms at 5 views of 1152 x 1536).
//
//   * one WAVEFRONT owns a 16-column strip and walks down it: a ring of input rows lives in wave-private LDS, the weights (8 K steps x 3 terms
//     = 96 VGPRs) in registers for the whole launch; no block barrier anywhere (a wavefront's DS instructions execute in order);
//   * eight output channels would fill half of an MFMA's M: M = (output row parity, channel), the K+1 input rows a row pair sees are each
//     multiplied against tap row j for the upper output row and j-1 for the lower one - KS / (KS + 1) of the A operand is useful;
//   * conv01 (8 channels): a K block = the 8 channels of one tap, the B operand one aligned ds_read_b128 at pixel n + kw;
//     6 rows x 5 columns = 30 K blocks = 8 steps;
//   * conv00 (3 channels): a pixel's record is 4 bf16 (the fourth zero), a K block = TWO adjacent pixels (taps kw, kw+1): 8 rows x 4 pairs =
//     32 K blocks = 8 steps.  A 16-byte read at an odd pixel would be misaligned, so every row is stored twice, the second copy one pixel
//     ahead: even n reads the first copy, odd n the second, both aligned.
#include "conv_common.h"
#include "split3.h"

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;

constexpr int TW = 16, RB = 4, STEPS = 8, CO = 8;

template <int KS>
struct Enc {
    static constexpr int P = KS / 2;
    static constexpr int CIN = KS == 7 ? 3 : 8;
    static constexpr int HC = TW + 2 * P;                            // halo columns: 22 | 20
    static constexpr int RINGN = RB + 2 * P;                         // rows Y-P .. Y+P+3 of a batch: 10 | 8
    static constexpr int PXB = KS == 7 ? 8 : 16;                     // bytes of a pixel's record per term
    static constexpr int COPYB = KS == 7 ? 24 * 8 : 0;               // conv00: second copy of the row, one pixel ahead (24 pixel slots: 22 + the zero tap's)
    static constexpr int ROWB = KS == 7 ? 2 * COPYB : (HC + 2) * 16; // 384 | 352 (conv01: two more pixel slots, read by the last step's two zero K blocks)
    static constexpr int TERM = RINGN * ROWB;
    static constexpr int WAVE_BYTES = ((3 * TERM + 255) / 256) * 256;
    static constexpr int UNITS = RB * HC;                            // staging units of a batch: one pixel of the four new rows
    static constexpr int PASSES = (UNITS + 63) / 64;                 // 2
    static constexpr int PRO = KS == 7 ? 2 : 1;                      // prologue builds (2P rows in chunks of four)
};

__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], c, 0, 0, 0);
    return c;
}

// prepared[(step * 3 + term)][lane][8]: the MFMA A operand, lane = kb * 16 + m, m = (dy = m >> 3, co = m & 7), K block t = 4 * step + kb; the
// folded BatchNorm scale of the output channel is multiplied in before the split.
//   KS = 5: t = (input row j, kw) = (t / 5, t % 5) (t >= 30: zero), element e = input channel, weight of kh = j - dy
//   KS = 7: t = (j, pair) = (t / 4, t % 4), element e = (kw = 2 * pair + e / 4, channel e % 4) (channel 3 and kw 7: zero)
template <int KS>
__global__ void enc_x3_prepare_kernel(const float* __restrict__ w /*[8,CIN,KS,KS]*/, const float* __restrict__ scale, bf16x8* __restrict__ out) {
    using E = Enc<KS>;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= STEPS * 3 * 64) return;
    const int lane = idx & 63, term = (idx >> 6) % 3, step = idx / 192;
    const int m = lane & 15, kb = lane >> 4, dy = m >> 3, co = m & 7, t = 4 * step + kb;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = 0.0f;
        int j, kw, c;
        if (KS == 5) { j = t / 5; kw = t % 5; c = e; }
        else { j = t / 4; kw = 2 * (t % 4) + e / 4; c = e % 4; }
        const int kh = j - dy;
        if (j <= KS && kh >= 0 && kh < KS && kw < KS && c < E::CIN) f = w[((co * E::CIN + c) * KS + kh) * KS + kw] * scale[co];
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

template <int KS>
__global__ __launch_bounds__(256, 2) void enc_x3_kernel(const float* __restrict__ x /*[N,CIN,H,W]*/, const bf16x8* __restrict__ prep,
                                                        const float* __restrict__ shift /*[8]*/, int H, int W, int ngroups, int nseg, int seg_rows,
                                                        float slope, float* __restrict__ y /*[N,8,H,W] or null*/, float* __restrict__ ycl /*[N,H,W,8] or null*/, int in_nhwc /*x is [N,H,W,8] (conv01 only)*/) {
    using E = Enc<KS>;
    constexpr int P = E::P, HC = E::HC, RINGN = E::RINGN, ROWB = E::ROWB, TERM = E::TERM, CIN = E::CIN;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;
    unsigned char* ring = smem + wave * E::WAVE_BYTES;

    const unsigned item = xcd_linear_block_id();
    const int group = (int)(item % (unsigned)ngroups), seg = (int)((item / (unsigned)ngroups) % (unsigned)nseg), img = (int)(item / (unsigned)(ngroups * nseg));
    const int x0 = (group * 4 + wave) * TW, ys = seg * seg_rows, yend = min(ys + seg_rows, H);
    if (x0 >= W) return;                                     // (no block barrier in this kernel)

    bf16x8 wgt[STEPS][3];
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) wgt[s][t] = prep[(s * 3 + t) * 64 + lane];

    const rsrc_t rx = make_rsrc(x + (size_t)img * CIN * H * W, (unsigned)(CIN * H * W) * 4u);
    const unsigned chb = (unsigned)(H * W) * 4u;

    // ---- staging: unit u = 64 p + lane = (row of the four new ones, halo column); CIN channel loads, split, the pixel's record(s) ----
    unsigned sgo[E::PASSES], smeta[E::PASSES];               // global offset without the batch's row part | LDS offset in the row, row << 16, valid << 20
#pragma unroll
    for (int p = 0; p < E::PASSES; ++p) {
        const int u = 64 * p + lane;
        const bool ok = u < E::UNITS;
        const int uu = ok ? u : 0, row = uu / HC, col = uu % HC, gx = x0 - P + col;
        sgo[p] = (ok && gx >= 0 && gx < W) ? (unsigned)(row * W + gx) * 4u : OOB;     // (wraps for rows above the image: masked per batch)
        smeta[p] = (unsigned)(col * E::PXB) | (unsigned)row << 16 | (ok ? 1u << 20 : 0u);
    }
    float sreg[E::PASSES][CIN];
    auto stage_issue = [&](int g0) {
#pragma unroll
        for (int p = 0; p < E::PASSES; ++p) {
            const int g = g0 + (int)(smeta[p] >> 16 & 15u);
            const unsigned off = (g >= 0 && g < H) ? sgo[p] + (unsigned)(g0 * W) * 4u : OOB;
            if (CIN == 8 && in_nhwc) {                        // a pixel's 8 channels are 32 contiguous bytes: two 16-byte loads instead of 8 scattered dwords
                const unsigned off8 = (off & OOB) ? OOB : off * 8u;
                const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off8, 0, 0));
                const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off8, 16, 0));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sreg[p][c] = a[c];
                    sreg[p][(4 + c) % CIN] = b[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < CIN; ++c) sreg[p][c] = buf_load(rx, off, (unsigned)c * chb);
            }
        }
    };
    auto stage_commit = [&](int slot0) {
#pragma unroll
        for (int p = 0; p < E::PASSES; ++p) {
            if (!(smeta[p] >> 20 & 1u)) continue;
            int slot = slot0 + (int)(smeta[p] >> 16 & 15u);
            slot = slot >= RINGN ? slot - RINGN : slot;
            unsigned char* dst = ring + slot * ROWB + (smeta[p] & 0xffffu);
            if constexpr (KS == 5) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 th, tm, tl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned xh, xm, xl;
                    mvsx3::split3_pair<true>(sreg[p][2 * e], sreg[p][2 * e + 1], xh, xm, xl);
                    th[e] = xh; tm[e] = xm; tl[e] = xl;
                }
                *reinterpret_cast<u32x4*>(dst) = th;
                *reinterpret_cast<u32x4*>(dst + TERM) = tm;
                *reinterpret_cast<u32x4*>(dst + 2 * TERM) = tl;
            } else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 th, tm, tl;
                unsigned xh, xm, xl;
                mvsx3::split3_pair<true>(sreg[p][0], sreg[p][1], xh, xm, xl);
                th[0] = xh; tm[0] = xm; tl[0] = xl;
                mvsx3::split3_pair<true>(sreg[p][2], 0.0f, xh, xm, xl);
                th[1] = xh; tm[1] = xm; tl[1] = xl;
                *reinterpret_cast<u32x2*>(dst) = th;                          // first copy: pixel c at 8 c
                *reinterpret_cast<u32x2*>(dst + TERM) = tm;
                *reinterpret_cast<u32x2*>(dst + 2 * TERM) = tl;
                if ((smeta[p] & 0xffffu) != 0) {                              // second copy: pixel c at 8 (c - 1)
                    *reinterpret_cast<u32x2*>(dst + E::COPYB - 8) = th;
                    *reinterpret_cast<u32x2*>(dst + E::COPYB - 8 + TERM) = tm;
                    *reinterpret_cast<u32x2*>(dst + E::COPYB - 8 + 2 * TERM) = tl;
                }
            }
        }
    };

    // ---- B operand of this lane's K block t = 4 s + kb ----
    //   KS = 5: row j = t / 5, pixel n + t % 5: lane part (n + kb) * 16, step part (4 s - 5 j) * 16 + the row's ring offset (j changes at most once inside a step)
    //   KS = 7: row j = s, pixels n + 2 kb, n + 2 kb + 1: copy n & 1, offset ((n & ~1) + 2 kb) * 8
    const unsigned blane = KS == 5 ? (unsigned)((n + kb) * 16) : (unsigned)((n & 1) * E::COPYB + ((n & ~1) + 2 * kb) * 8);
    auto multiply = [&](int sb, f32x4& c0, f32x4& c1) {
        int ro[RINGN];
#pragma unroll
        for (int k = 0; k < RINGN; ++k) {
            const int s_ = sb + k;
            ro[k] = (s_ >= RINGN ? s_ - RINGN : s_) * ROWB;
        }
        bf16x8 xa[2][3], xb[2][3];
        auto fetch = [&](int s, bf16x8 (&a)[3], bf16x8 (&b)[3]) {
            const unsigned char *pa, *pb;
            if constexpr (KS == 5) {
                const int jlo = min((4 * s) / 5, KS), jhi = min((4 * s + 3) / 5, KS), kcut = 5 * jhi - 4 * s;      // lanes with kb >= kcut are in row jhi
                const bool up = jhi != jlo && kb >= kcut;
                pa = ring + blane + (up ? ro[jhi] + (4 * s - 5 * jhi) * 16 : ro[jlo] + (4 * s - 5 * jlo) * 16);
                pb = ring + blane + (up ? ro[2 + jhi] + (4 * s - 5 * jhi) * 16 : ro[2 + jlo] + (4 * s - 5 * jlo) * 16);
            } else {
                pa = ring + blane + ro[s];
                pb = ring + blane + ro[2 + s];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                a[t] = *reinterpret_cast<const bf16x8*>(pa + t * TERM);
                b[t] = *reinterpret_cast<const bf16x8*>(pb + t * TERM);
            }
        };
        fetch(0, xa[0], xb[0]);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (s + 1 < STEPS) fetch(s + 1, xa[(s + 1) & 1], xb[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            c0 = mfma6(wgt[s], xa[s & 1], c0);
            c1 = mfma6(wgt[s], xb[s & 1], c1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x4 shv;
#pragma unroll
    for (int r = 0; r < 4; ++r) shv[r] = shift[(kb & 1) * 4 + r];
    float* y_img = y + (size_t)img * CO * H * W;
    // the pixel slots only zero weights ever multiply (conv00: the eighth tap's; conv01: the last step's two empty K blocks) must hold finite data
    for (int i = lane; i < 3 * TERM / 16; i += 64) reinterpret_cast<f32x4*>(ring)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" ::: "memory");

    // ---- prologue: rows ys-P .. ys+P-1; then per batch: rows Y+P .. Y+P+3 in, output rows Y .. Y+3 out ----
    const int nb = (yend - ys + RB - 1) / RB;
    const int gfirst = ys - P - (E::PRO * RB - 2 * P);       // the first prologue build's first row (ring slot 0)
    stage_issue(gfirst);
    int slot0 = 0;
#pragma unroll
    for (int q = 0; q < E::PRO; ++q) {
        stage_commit(slot0);
        stage_issue(gfirst + RB * (q + 1));
        slot0 = slot0 + RB >= RINGN ? slot0 + RB - RINGN : slot0 + RB;
    }
    int sb = (E::PRO * RB - 2 * P);                          // ring slot of row ys - P
    for (int b = 0; b < nb; ++b) {
        const int Y = ys + RB * b;
        asm volatile("" ::: "memory");
        stage_commit(slot0);
        if (b + 1 < nb) stage_issue(Y + RB + P);
        slot0 = slot0 + RB >= RINGN ? slot0 + RB - RINGN : slot0 + RB;
        asm volatile("" ::: "memory");
        f32x4 c0 = shv, c1 = shv;
        multiply(sb, c0, c1);
        sb = sb + RB >= RINGN ? sb + RB - RINGN : sb + RB;
        // D[m = (dy, co)][n]: this lane holds row parity kb >> 1, channels (kb & 1) * 4 + r of column x0 + n
        const int gx = x0 + n;
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const f32x4& c = rp ? c1 : c0;
            const int gy = Y + 2 * rp + (kb >> 1);
            if (gy < yend && gx < W) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = c[r];
                    o[r] = v > 0.0f ? v : v * slope;
                    if (y) y_img[(size_t)((kb & 1) * 4 + r) * H * W + (size_t)gy * W + gx] = o[r];
                }
                // the channel-last companion (what the decoder's last level stages with 16-byte loads): this lane's 4 channels are 16 contiguous bytes
                if (ycl) *reinterpret_cast<f32x4*>(ycl + (((size_t)img * H + gy) * W + gx) * CO + (kb & 1) * 4) = o;
            }
        }
    }
}

template <int KS>
int launch_enc(const float* x, int in_nhwc, const void* prepared, const float* shift, int N, int H, int W, float slope, float* y, float* ycl, hipStream_t s) {
    using E = Enc<KS>;
    const int ngroups = mvs::ceil_div(W, 4 * TW), slots = 2 * mvs::device_cus();
    int nseg = 1;
    double best = 0.0;
    for (int k = 1; k <= 32; ++k) {
        const int rows = mvs::ceil_div(mvs::ceil_div(H, k), RB) * RB, ns = mvs::ceil_div(H, rows);
        if (k > 1 && rows < 32) break;
        const long long items = (long long)N * ngroups * ns;
        const double fill = (double)items / (double)(mvs::ceil_div(items, (long long)slots) * slots);
        if (fill > best + 0.02) { best = fill; nseg = ns; }
    }
    const int seg_rows = mvs::ceil_div(mvs::ceil_div(H, nseg), RB) * RB;
    nseg = mvs::ceil_div(H, seg_rows);
    const long long items = (long long)N * ngroups * nseg;
    MVS_REQUIRE(items < (1ll << 31), "mvs_conv2d_x3_bn_lrelu: too many strips");
    constexpr int LDS = 4 * E::WAVE_BYTES;
    {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(&enc_x3_kernel<KS>), LDS, "mvs_conv2d_x3_bn_lrelu");
        if (rc != MVS_OK) return rc;
    }
    hipLaunchKernelGGL((enc_x3_kernel<KS>), dim3((unsigned)items), dim3(256), LDS, s, x, static_cast<const bf16x8*>(prepared), shift, H, W, ngroups, nseg,
                       seg_rows, slope, y, ycl, in_nhwc);
    return mvs::finish_launch("mvs_conv2d_x3_bn_lrelu");
}

}  // namespace

extern "C" int mvs_conv2d_x3_supported(int Cin, int Cout, int KS, int stride) {
    return (stride == 1 && Cout == 8 && ((Cin == 3 && KS == 7) || (Cin == 8 && KS == 5))) ? 1 : 0;
}

extern "C" int64_t mvs_conv2d_x3_prepared_bytes(int Cin, int Cout, int KS) {
    return mvs_conv2d_x3_supported(Cin, Cout, KS, 1) ? (int64_t)STEPS * 3 * 64 * 16 : -1;
}

extern "C" int mvs_conv2d_x3_prepare(const float* w, const float* scale, int Cin, int Cout, int KS, void* prepared, mvs_stream_t stream) {
    MVS_REQUIRE(w && scale && prepared, "mvs_conv2d_x3_prepare: null pointer");
    MVS_REQUIRE(mvs_conv2d_x3_supported(Cin, Cout, KS, 1), "mvs_conv2d_x3_prepare: (Cin,Cout,K)=(%d,%d,%d) is not conv00 / conv01 of the FPN encoder", Cin, Cout, KS);
    constexpr int total = STEPS * 3 * 64;
    if (KS == 7)
        hipLaunchKernelGGL(enc_x3_prepare_kernel<7>, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w, scale, static_cast<bf16x8*>(prepared));
    else
        hipLaunchKernelGGL(enc_x3_prepare_kernel<5>, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w, scale, static_cast<bf16x8*>(prepared));
    return mvs::finish_launch("mvs_conv2d_x3_prepare");
}

extern "C" int mvs_conv2d_x3_bn_lrelu_layout(const float* x, int x_nhwc, const void* prepared, const float* shift, int N, int Cin, int Cout, int KS,
                                             int stride, int H, int W, float slope, float* y, float* y_nhwc, mvs_stream_t stream);

extern "C" int mvs_conv2d_x3_bn_lrelu(const float* x, const void* prepared, const float* shift, int N, int Cin, int Cout, int KS, int stride, int H,
                                      int W, float slope, float* y, mvs_stream_t stream) {
    return mvs_conv2d_x3_bn_lrelu_layout(x, 0, prepared, shift, N, Cin, Cout, KS, stride, H, W, slope, y, nullptr, stream);
}

extern "C" int mvs_conv2d_x3_bn_lrelu_layout(const float* x, int x_nhwc, const void* prepared, const float* shift, int N, int Cin, int Cout, int KS,
                                             int stride, int H, int W, float slope, float* y, float* y_nhwc, mvs_stream_t stream) {
    MVS_REQUIRE(x && prepared && shift && (y || y_nhwc), "mvs_conv2d_x3_bn_lrelu: null pointer");
    MVS_REQUIRE(mvs_conv2d_x3_supported(Cin, Cout, KS, stride), "mvs_conv2d_x3_bn_lrelu: (Cin,Cout,K,stride)=(%d,%d,%d,%d) is not conv00 / conv01 of the FPN encoder",
                Cin, Cout, KS, stride);
    MVS_REQUIRE(!x_nhwc || Cin == 8, "mvs_conv2d_x3_bn_lrelu: a channel-last input needs Cin = 8 (got %d)", Cin);
    MVS_REQUIRE(N >= 1 && H >= 1 && W >= 1, "mvs_conv2d_x3_bn_lrelu: bad shape N=%d H=%d W=%d", N, H, W);
    MVS_REQUIRE((int64_t)8 * H * W * 4 < ((int64_t)1 << 28), "mvs_conv2d_x3_bn_lrelu: one image exceeds 256 MiB");
    hipStream_t s = MVS_STREAM(stream);
    return KS == 7 ? launch_enc<7>(x, 0, prepared, shift, N, H, W, slope, y, y_nhwc, s) : launch_enc<5>(x, x_nhwc ? 1 : 0, prepared, shift, N, H, W, slope, y, y_nhwc, s);
}
