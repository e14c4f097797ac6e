// FPN decoder levels 1 and 2 (models/module.py:260-265: intra_k = up2(intra_{k-1}) + inner_k(lateral_k); out_k = Swish(BN(conv3x3 64 -> C_k)),
// C_k = 32 | 16) with the 3x3 convolution in the THREE-TERM BF16 SPLIT form (split3.h): the structure of fpn_level_kernel (fpn.hip: 4 x 32 output
// tile, the 64 top-down channels in four chunks of 16, the intra tile built on the vector ALU from an LDS window of the coarser level) with the
// matrix part moved from v_mfma_f32_16x16x4_f32 - which shares the issue port with every vector instruction, so its 36 MFMAs x 32 clk per
// wavefront and chunk ADD to the tile's vector work - to six v_mfma_f32_16x16x32_bf16 per fp32-equivalent K = 32 step on the matrix pipe
// (5 steps per chunk and 16 x 16 output tile: 0.42 of the matrix time).  The chunk's intra values are split where they are made (the thread that
// interpolated a halo pixel's 16 channels writes its two octets as [term][octet][pixel][8 bf16]: a K block of the B operand = one ds_read_b128);
// the weights arrive pre-split in MFMA fragment order (BatchNorm scale folded in) straight from L1/L2, one step ahead.  The intra map itself still
// leaves as fp32 (NCHW for the next level of this kind, channel-last for fpn_cp.hip).  fp32 in / fp32 out, fp32-equivalent.
#include "conv_common.h"
#include "split3.h"

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;

constexpr int FC = 64;
constexpr int TH = 4, TW = 32;               // output tile
constexpr int HR = TH + 2, HC = TW + 2;      // intra tile with the 3x3 halo
constexpr int NPIX = HR * HC;                // 204 <= 256 threads: one thread per halo pixel
constexpr int CCH = 16;                      // top-down channels per chunk
constexpr int CS = 208;                      // channel stride of the fp32 LDS tile / pixel slots of the split tile
constexpr int SH = 6, SW = 20, SS = SH * SW; // LDS window of the coarser level
constexpr int OCTB = CS * 16, TERMB = 2 * OCTB;   // split tile: [term][octet][pixel][16 B]
constexpr int STEPS = 5;                     // 9 taps x 2 octets = 18 K blocks (+ 2 zero)

__device__ __forceinline__ float swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

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
// K block t = 4 step + kb = (tap = t / 2, octet = t % 2) (t >= 18: zero), channel 16 chunk + 8 octet + e; scale[co] multiplied in
__global__ void fpn_lvl_x3_prepare_kernel(const float* __restrict__ w3 /*[CK,64,3,3]*/, const float* __restrict__ scale, int CK, bf16x8* __restrict__ out) {
    const int NT = CK / 16, total = 4 * STEPS * NT * 3 * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, term = (idx >> 6) % 3, nt = (idx / 192) % NT, step = (idx / (192 * NT)) % STEPS, chunk = idx / (192 * NT * STEPS);
    const int m = lane & 15, kb = lane >> 4, co = 16 * nt + m, t = 4 * step + kb, tap = t >> 1, oct = t & 1;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 16 * chunk + 8 * oct + e;
        const float f = t < 18 ? w3[((co * FC + c) * 3 + tap / 3) * 3 + tap % 3] * scale[co] : 0.0f;
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

template <int CK>
__global__ __launch_bounds__(256, (CK == 32 ? 2 : 3)) void fpn_level_x3s_kernel(const float* __restrict__ prev /*[N,64,h,w]*/, const float* __restrict__ lat /*[N,CK,2h,2w]*/,
                                                                                const float* __restrict__ w_in_p /*[32,CK,2]*/, const float* __restrict__ b_in /*[64]*/,
                                                                                const bf16x8* __restrict__ wprep, const float* __restrict__ shift, int h, int w,
                                                                                float* __restrict__ intra_out /*[N,64,2h,2w] ([N,2h,2w,64] with intra_nhwc) or null*/,
                                                                                float* __restrict__ out /*[N,2h,2w,CK]*/, int intra_nhwc) {
    constexpr int NT = CK / 16;
    __shared__ __attribute__((aligned(16))) float s_src[CCH * SS];
    __shared__ __attribute__((aligned(16))) float s_tile[CCH * CS];
    __shared__ __attribute__((aligned(256))) unsigned char s_b[3 * TERMB];

    unsigned bx, by, bz;
    xcd_block_coords(bx, by, bz);
    const int H = 2 * h, W = 2 * w;
    const int x0 = (int)bx * TW, y0 = (int)by * TH, img = (int)bz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;

    // ---- this thread's halo pixel: bilinear taps into the coarse window (ATen upsample_bilinear2d, align_corners=True) ----
    const int p = tid;
    const int gy = y0 - 1 + p / HC, gx = x0 - 1 + p % HC;
    const bool inimg = (p < NPIX) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float sy = (float)(h - 1) / (float)(H - 1), sx = (float)(w - 1) / (float)(W - 1);
    const int wy0 = (int)(sy * (float)max(y0 - 1, 0)), wx0 = (int)(sx * (float)max(x0 - 1, 0));
    int o00 = 0, o01 = 0, o10 = 0, o11 = 0;
    float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
    const float gate = inimg ? 1.0f : 0.0f;
    float lv[CK];
#pragma unroll
    for (int j = 0; j < CK; ++j) lv[j] = 0.0f;
    if (inimg) {
        const float fy = sy * (float)gy, fx = sx * (float)gx;
        const int iy0 = (int)fy, ix0 = (int)fx;
        const int iy1 = iy0 + (iy0 < h - 1 ? 1 : 0), ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
        const float ly1 = fy - (float)iy0, lx1 = fx - (float)ix0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
        w00 = ly0 * lx0;
        w01 = ly0 * lx1;
        w10 = ly1 * lx0;
        w11 = ly1 * lx1;
        const int ry0 = min(iy0 - wy0, SH - 1), ry1 = min(iy1 - wy0, SH - 1), rx0 = min(ix0 - wx0, SW - 1), rx1 = min(ix1 - wx0, SW - 1);
        o00 = ry0 * SW + rx0;
        o01 = ry0 * SW + rx1;
        o10 = ry1 * SW + rx0;
        o11 = ry1 * SW + rx1;
        const float* lat_img = lat + (size_t)img * CK * H * W;
        const unsigned lo = (unsigned)(gy * W + gx), HW = (unsigned)(H * W);
#pragma unroll
        for (int j = 0; j < CK; ++j) lv[j] = lat_img[j * HW + lo];
    }

    // ---- staging roles: thread = (window slot r, channel parity); the slot's global offset is the same for every chunk ----
    const int sr = tid & 127, shalf = tid >> 7;
    const int spy = wy0 + sr / SW, spx = wx0 + sr % SW;
    const bool svalid = sr < SS && spy < h && spx < w;
    const float* prev_img = prev + (size_t)img * FC * h * w;
    const unsigned hw = (unsigned)(h * w), soff = svalid ? (unsigned)(shalf * (h * w) + spy * w + spx) : 0u;
    constexpr int NSR = CCH / 2;
    float sreg[NSR];
    auto prefetch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < NSR; ++i) sreg[i] = svalid ? prev_img[(unsigned)(cc * CCH + 2 * i) * hw + soff] : 0.0f;
    };
    auto commit = [&]() {
        if (sr < SS) {
#pragma unroll
            for (int i = 0; i < NSR; ++i) s_src[(2 * i + shalf) * SS + sr] = sreg[i];
        }
    };

    // this wavefront's output row wv, two 16-column halves; D[m = channel][n = column]
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
    {   // consume the lateral values once BEFORE the loop (see fpn.hip: the wait-count pass otherwise drains the prefetch with them)
        float guard = 0.0f;
#pragma unroll
        for (int j = 0; j < CK; ++j) guard += lv[j];
        asm volatile("" ::"v"(guard));
    }
    for (int cc = 0; cc < FC / CCH; ++cc) {
        __syncthreads();                                    // the previous chunk's MFMA phase has finished reading LDS
        commit();
        __syncthreads();
        if (cc + 1 < FC / CCH) prefetch(cc + 1);            // in flight during this chunk's two phases
        // ---- intra tile of this chunk: upsampled coarse level + lateral 1x1 convolution (zero outside the image: the 3x3 conv's padding), kept as
        //      fp32 for the intra map's store and split for the matrix cores ----
        if (p < NPIX) {
            float v16[CCH];
#pragma unroll
            for (int c = 0; c < CCH; ++c) {
                const int ch = cc * CCH + c;
                float v = b_in[ch] * gate;
#pragma unroll
                for (int j = 0; j < CK; ++j) v = fmaf(w_in_p[((ch >> 1) * CK + j) * 2 + (ch & 1)], lv[j], v);
                const float* S = s_src + c * SS;
                v = fmaf(w00, S[o00], v);
                v = fmaf(w01, S[o01], v);
                v = fmaf(w10, S[o10], v);
                v = fmaf(w11, S[o11], v);
                s_tile[c * CS + p] = v;
                v16[c] = v;
                if (c % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // else all 16 channels' window reads are hoisted
            }
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 th, tm, tl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned xh, xm, xl;
                    mvsx3::split3_pair<true>(v16[8 * o + 2 * e], v16[8 * o + 2 * e + 1], xh, xm, xl);
                    th[e] = xh; tm[e] = xm; tl[e] = xl;
                }
                unsigned char* dst = s_b + o * OCTB + p * 16;
                *reinterpret_cast<u32x4*>(dst) = th;
                *reinterpret_cast<u32x4*>(dst + TERMB) = tm;
                *reinterpret_cast<u32x4*>(dst + 2 * TERMB) = tl;
            }
        }
        __syncthreads();
        if (intra_out && intra_nhwc) {                      // interior of the tile -> channel-last: a thread takes 4 channels of one pixel -> one 16-byte store
#pragma unroll
            for (int i = 0; i < CCH * TH * TW / 1024; ++i) {
                const int idx = tid + i * 256;
                const int pix = idx % (TH * TW), g = idx / (TH * TW), col = pix % TW, row = pix / TW;
                const int yy = y0 + row, xx = x0 + col;
                const float* src = s_tile + (4 * g) * CS + (row + 1) * HC + col + 1;
                const f32x4 v = {src[0], src[CS], src[2 * CS], src[3 * CS]};
                if (yy < H && xx < W)
                    *reinterpret_cast<f32x4*>(intra_out + ((size_t)img * H * W + (unsigned)(yy * W + xx)) * FC + cc * CCH + 4 * g) = v;
            }
        } else if (intra_out) {                             // interior of the tile -> NCHW, 128-byte row segments
#pragma unroll
            for (int i = 0; i < CCH * TH * TW / 256; ++i) {
                const int idx = tid + i * 256;
                const int col = idx % TW, row = (idx / TW) % TH, c = idx / (TW * TH);
                const int yy = y0 + row, xx = x0 + col;
                if (yy < H && xx < W)
                    intra_out[(size_t)img * FC * H * W + (unsigned)(((cc * CCH + c) * H + yy) * W + xx)] = s_tile[c * CS + (row + 1) * HC + col + 1];
            }
        }
        // ---- 3x3 convolution of this chunk on the bf16 matrix cores, split form; weights one step ahead from L1 / L2 ----
        {
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
    }

    // ---- epilogue: BatchNorm shift (the scale sits in the weights, the conv bias in the shift) + Swish; this lane's 4 channels of a pixel = 16 bytes ----
    const int yy = y0 + wv;
    float* out_img = out + (size_t)img * H * W * CK;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int co = 16 * q + 4 * kb;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + co);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int xx = x0 + 16 * t + n;
            if (yy < H && xx < W) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = swish(acc[t][q][r] + sh[r]);
                *reinterpret_cast<f32x4*>(out_img + (unsigned)((yy * W + xx) * CK + co)) = o;
            }
        }
    }
}

}  // namespace

extern "C" int64_t mvs_fpn_level_x3s_prepared_bytes(int Ck) { return (Ck == 16 || Ck == 32) ? (int64_t)4 * STEPS * (Ck / 16) * 3 * 64 * 16 : -1; }

extern "C" int mvs_fpn_level_x3s_prepare(const float* w3, const float* scale, int Ck, void* prepared, mvs_stream_t stream) {
    MVS_REQUIRE(w3 && scale && prepared, "mvs_fpn_level_x3s_prepare: null pointer");
    MVS_REQUIRE(Ck == 16 || Ck == 32, "mvs_fpn_level_x3s_prepare: built for the levels with Ck = 16 or 32 (got %d)", Ck);
    const int total = 4 * STEPS * (Ck / 16) * 3 * 64;
    hipLaunchKernelGGL(fpn_lvl_x3_prepare_kernel, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w3, scale, Ck, static_cast<bf16x8*>(prepared));
    return mvs::finish_launch("mvs_fpn_level_x3s_prepare");
}

extern "C" int mvs_fpn_level_x3s(const float* intra_prev, const float* lateral, const float* w_inner_p, const float* b_inner, const void* prepared,
                                 const float* shift, int N, int Ck, int h, int w, float* intra_out, int intra_nhwc, float* out, mvs_stream_t stream) {
    MVS_REQUIRE(intra_prev && lateral && w_inner_p && b_inner && prepared && shift && out, "mvs_fpn_level_x3s: null pointer");
    MVS_REQUIRE(Ck == 16 || Ck == 32, "mvs_fpn_level_x3s: lateral channels must be 16 or 32 (got %d)", Ck);
    MVS_REQUIRE(N >= 1 && N <= 65535 && h >= 1 && w >= 1 && (int64_t)2 * h <= 4 * 65535, "mvs_fpn_level_x3s: bad shape N=%d h=%d w=%d", N, h, w);
    MVS_REQUIRE((int64_t)FC * 4 * h * w < ((int64_t)1 << 31), "mvs_fpn_level_x3s: one image's 64-channel level exceeds 2^31 elements (32-bit in-image offsets)");
    const dim3 grid(mvs::ceil_div(2 * w, TW), mvs::ceil_div(2 * h, TH), N), block(256);
    hipStream_t s = MVS_STREAM(stream);
    if (Ck == 16)
        hipLaunchKernelGGL(fpn_level_x3s_kernel<16>, grid, block, 0, s, intra_prev, lateral, w_inner_p, b_inner, static_cast<const bf16x8*>(prepared), shift, h, w,
                           intra_out, out, intra_nhwc);
    else
        hipLaunchKernelGGL(fpn_level_x3s_kernel<32>, grid, block, 0, s, intra_prev, lateral, w_inner_p, b_inner, static_cast<const bf16x8*>(prepared), shift, h, w,
                           intra_out, out, intra_nhwc);
    return mvs::finish_launch("mvs_fpn_level_x3s");
}
