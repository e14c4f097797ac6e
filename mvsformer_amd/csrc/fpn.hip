// FPN decoder (models/module.py:242-270), eval mode: the step that hands the four feature maps to the plane sweeps
// (SURVEY.md §8 f1/f4).  Reference per level k = 1..3:
//     intra_k = interpolate(intra_{k-1}, x2, bilinear, align_corners=True) + inner_k(lateral_k)        (1x1 conv, C_k -> 64)
//     out_k   = Swish(BatchNorm2d(conv3x3(intra_k)))                                                     (64 -> C_k)
// materializes intra_3 = [N,64,H,W] (453 MB per 1152x1536 view) only to convolve it down to 8 channels.  Here one kernel per
// level builds the intra tile in LDS (bilinear taps from an LDS-staged window of intra_{k-1}, the lateral 1x1 convolution on
// the packed fp32 VALU with wave-uniform weights in SGPRs), runs the 3x3 convolution from LDS on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32, same fragment conventions as conv3d.hip) and stores out_k CHANNEL-LAST [N,H,W,C_k] - the layout the
// sweeps gather from, so the four nchw_to_nhwc passes of the reference layout disappear.  intra_k goes back to HBM only for
// k < 3 (the next level upsamples it); intra_3 never exists.
//
// Layouts: every input (encoder outputs, intra) is NCHW like the reference's tensors; outputs out_k are NHWC.
// Work split: block = 4 x 32 output pixels, 4 wavefronts; the 64 top-down channels are processed in 4 chunks of 16 so the LDS
// footprint stays at 30-50 KB (3-5 blocks per CU).  GEMM view per chunk: M = 16 pixels along x, K = taps x 16 channels,
// N = output channels.  C_k = 8 would fill half of the 16-wide N tile, so there N = (2 output rows) x (8 channels): the four
// input rows that feed an output row pair are each multiplied against a 16-column weight matrix holding tap row j for the upper
// output row and tap row j-1 for the lower one (zero where that tap does not exist) - 12 MFMAs per 4 channels for two rows
// instead of 18.
//
// Algorithmic FLOPs per output pixel of level k: 2*64*(C_k + 9*C_k); bytes: 4*(C_k in + C_k out + 64/4 upsampled source).
#include "conv_common.h"

namespace {
using namespace mvsconv;

constexpr int FC = 64;                       // channels of the top-down path (feat_chs[-1])
constexpr int TH = 4, TW = 32;               // output tile
constexpr int HR = TH + 2, HC = TW + 2;      // intra tile with the 3x3 halo
constexpr int NPIX = HR * HC;                // 204 <= 256 threads: one thread per halo pixel
constexpr int CCH = 16;                      // top-down channels per chunk
constexpr int CS = pad_cs(NPIX, 1);          // 208: channel stride of the LDS tile (== 16 mod 32, see conv_common.h)
constexpr int SH = 6, SW = 20, SS = SH * SW; // LDS window of the coarser level: (TH+1)/2 + 2 (+1 slack) rows, (TW+1)/2 + 2 (+1) columns

__host__ __device__ constexpr int fpn_nt(int ck) { return ck == 32 ? 2 : 1; }
__host__ __device__ constexpr int fpn_taps(int ck) { return ck == 8 ? 12 : 9; }
__host__ __device__ constexpr int fpn_chunk_floats(int ck) { return 4 * fpn_taps(ck) * 4 * np_of(fpn_nt(ck)); }

// x * sigmoid(x) with the hardware exp2 / reciprocal (a few ulp; the epilogue shares the fp32 pipe with the MFMAs, an IEEE divide costs 10 slots)
__device__ __forceinline__ float swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

// packed image: [slab = cin/4 (16)][tap (9 | 12)][cin%4][NP]; a chunk of 16 input channels = 4 consecutive slabs
__global__ void fpn_pack_kernel(const float* __restrict__ w /*[Cout,64,3,3]*/, int Cout, float* __restrict__ out) {
    const int T = fpn_taps(Cout), NP = np_of(fpn_nt(Cout));
    const int total = (FC / 4) * T * 4 * NP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int n = idx % NP, kk = (idx / NP) % 4, tap = (idx / (4 * NP)) % T, slab = idx / (4 * NP * T);
        const int c = slab * 4 + kk;
        float v = 0.0f;
        if (Cout == 8) {                     // tap = j*3 + kx over the 4 input rows j of an output row pair; n = h*8 + co
            const int j = tap / 3, kx = tap % 3, h = n >> 3, co = n & 7, ky = j - h;
            if (n < 16 && ky >= 0 && ky <= 2) v = w[((size_t)(co * FC + c) * 3 + ky) * 3 + kx];
        } else if (n < Cout) {
            v = w[(size_t)(n * FC + c) * 9 + tap];
        }
        out[idx] = v;
    }
}

// out0 = Swish(BN(conv1x1 64->64 (conv31)))  (module.py:246,259): one thread per pixel, weights wave-uniform
__global__ __launch_bounds__(64) void fpn_out0_kernel(const float* __restrict__ x /*[N,64,hw]*/, const float* __restrict__ w /*[64,64]*/,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int hw,
                                                      float* __restrict__ out /*[N,hw,64]*/) {
    const int pix = blockIdx.x * 64 + threadIdx.x, img = blockIdx.y;
    if (pix >= hw) return;
    const float* xp = x + (size_t)img * FC * hw + pix;
    float acc[FC];
#pragma unroll
    for (int co = 0; co < FC; ++co) acc[co] = 0.0f;
    for (int ci = 0; ci < FC; ++ci) {
        const float xv = xp[(size_t)ci * hw];
#pragma unroll
        for (int co = 0; co < FC; ++co) acc[co] = fmaf(w[co * FC + ci], xv, acc[co]);
    }
    f32x4* o = reinterpret_cast<f32x4*>(out + ((size_t)img * hw + pix) * FC);
#pragma unroll
    for (int q = 0; q < FC / 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = swish(fmaf(acc[q * 4 + i], scale[q * 4 + i], shift[q * 4 + i]));
        o[q] = v;
    }
}

template <int CK>
__global__ __launch_bounds__(256, (CK == 32 ? 2 : (CK == 16 ? 3 : 4))) void fpn_level_kernel(const float* __restrict__ prev /*[N,64,h,w]*/, const float* __restrict__ lat /*[N,CK,2h,2w]*/,
                                                        const float* __restrict__ w_in_p /*[32,CK,2]*/, const float* __restrict__ b_in /*[64]*/,
                                                        const float* __restrict__ wp, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int h, int w,
                                                        float* __restrict__ intra_out /*[N,64,2h,2w] ([N,2h,2w,64] with intra_nhwc) or null*/,
                                                        float* __restrict__ out /*[N,2h,2w,CK]*/, int intra_nhwc) {
    constexpr bool ROWS2 = (CK == 8);
    constexpr int NT = fpn_nt(CK), NP = np_of(NT), T = fpn_taps(CK), WCH = fpn_chunk_floats(CK);
    constexpr int MT = ROWS2 ? 1 : 2;        // M tiles (16 pixels) per wavefront
    __shared__ __attribute__((aligned(16))) float s_src[CCH * SS];
    __shared__ __attribute__((aligned(16))) float s_tile[CCH * CS];
    __shared__ __attribute__((aligned(16))) float s_w[WCH];

    unsigned bx, by, bz;
    xcd_block_coords(bx, by, bz);
    const int H = 2 * h, W = 2 * w;
    const int x0 = (int)bx * TW, y0 = (int)by * TH, img = (int)bz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;

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
        // wave-uniform 64-bit base + 32-bit lane offset: the loads take the saddr form, no per-lane 64-bit address arithmetic
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
    constexpr int NSR = CCH / 2, NWV = (WCH / 4 + 255) / 256;
    float sreg[NSR];
    f32x4 wreg[NWV];
    auto prefetch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < NSR; ++i) sreg[i] = svalid ? prev_img[(unsigned)(cc * CCH + 2 * i) * hw + soff] : 0.0f;
        const f32x4* src = reinterpret_cast<const f32x4*>(wp + (size_t)cc * WCH);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            wreg[i] = (idx < WCH / 4) ? src[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() {
        if (sr < SS) {
#pragma unroll
            for (int i = 0; i < NSR; ++i) s_src[(2 * i + shalf) * SS + sr] = sreg[i];
        }
        f32x4* dst = reinterpret_cast<f32x4*>(s_w);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            if (idx < WCH / 4) dst[idx] = wreg[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    prefetch(0);
    {   // consume the lateral values once BEFORE the loop: otherwise the wait-count pass keeps them "pending" around the back edge
        // and every chunk's first use waits vmcnt down to 0 - which also drains the (younger) prefetch of the next chunk
        float guard = 0.0f;
#pragma unroll
        for (int j = 0; j < CK; ++j) guard += lv[j];
        asm volatile("" ::"v"(guard));
    }
    for (int cc = 0; cc < FC / CCH; ++cc) {
        __syncthreads();                                    // the previous chunk's MFMA phase has finished reading LDS
        commit();                                           // coarse window of this chunk's 16 channels + its packed 3x3 weights
        __syncthreads();
        if (cc + 1 < FC / CCH) prefetch(cc + 1);            // in flight during this chunk's two phases
        // ---- intra tile: upsampled coarse level + lateral 1x1 convolution; zero outside the image (the 3x3 conv's padding) ----
        // two channels per step on the packed fp32 pipe (v_pk_fma_f32): the pair's lateral weights are 2*CK consecutive floats of w_in_p, the pair's
        // window values arrive as one ds_read2_b32.  Branch-free: a pixel outside the image has zero taps, zero lateral values, gate 0.
        if (p < NPIX) {
#pragma unroll
            for (int c = 0; c < CCH; c += 2) {
                const int ch = cc * CCH + c;
                f32x2 v = f32x2{b_in[ch], b_in[ch + 1]} * f32x2{gate, gate};
#pragma unroll
                for (int j = 0; j < CK; ++j) v = pk_fma(f32x2{w_in_p[(ch * CK) + 2 * j], w_in_p[(ch * CK) + 2 * j + 1]}, f32x2{lv[j], lv[j]}, v);
                const float* S = s_src + c * SS;
                v = pk_fma(f32x2{w00, w00}, f32x2{S[o00], S[SS + o00]}, v);
                v = pk_fma(f32x2{w01, w01}, f32x2{S[o01], S[SS + o01]}, v);
                v = pk_fma(f32x2{w10, w10}, f32x2{S[o10], S[SS + o10]}, v);
                v = pk_fma(f32x2{w11, w11}, f32x2{S[o11], S[SS + o11]}, v);
                s_tile[c * CS + p] = v.x;
                s_tile[(c + 1) * CS + p] = v.y;
                if (c % 4 == 2) __builtin_amdgcn_sched_barrier(0);      // else all 16 channels' window reads are hoisted (187 VGPRs)
            }
        }
        __syncthreads();
        if (intra_out && intra_nhwc) {                      // interior of the tile -> channel-last (what fpn_cp.hip stages with 16-byte loads): 64-byte pixel segments
            // a thread takes 4 channels of one pixel (LDS reads: consecutive lanes = consecutive pixels, conflict-free) -> one 16-byte store
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
        // ---- 3x3 convolution of this chunk on the matrix cores ----
#pragma unroll
        for (int ks = 0; ks < CCH / 4; ++ks) {
            const float* abase = s_tile + (ks * 4 + kk) * CS + i16;
            const float* bbase = s_w + (ks * T * 4 + kk) * NP + i16;
            if constexpr (ROWS2) {
                const int pq = wv >> 1, mt = wv & 1;        // output row pair, 16-column half
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float a = abase[(2 * pq + j) * HC + mt * 16 + kx];
                        const float b = bbase[(j * 3 + kx) * 4 * NP];
                        acc[0][0] = mfma4(a, b, acc[0][0]);
                    }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float a[MT];
#pragma unroll
                        for (int t = 0; t < MT; ++t) a[t] = abase[(wv + ky) * HC + t * 16 + kx];
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const float b = bbase[(ky * 3 + kx) * 4 * NP + n * 16];
#pragma unroll
                            for (int t = 0; t < MT; ++t) acc[t][n] = mfma4(a[t], b, acc[t][n]);
                        }
                    }
            }
        }
    }

    // ---- epilogue: BatchNorm (folded with the conv bias) + Swish, channel-last store ----
    float* out_img = out + (size_t)img * H * W * CK;
    if constexpr (ROWS2) {
        const int pq = wv >> 1, mt = wv & 1, co = i16 & 7, yy = y0 + 2 * pq + (i16 >> 3);
        const float sc = scale[co], sh = shift[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int xx = x0 + mt * 16 + 4 * kk + r;
            if (yy < H && xx < W) out_img[(unsigned)((yy * W + xx) * CK + co)] = swish(fmaf(acc[0][0][r], sc, sh));
        }
    } else {
        const int yy = y0 + wv;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n * 16 + i16;
            const float sc = scale[co], sh = shift[co];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int xx = x0 + t * 16 + 4 * kk + r;
                    if (yy < H && xx < W) out_img[(unsigned)((yy * W + xx) * CK + co)] = swish(fmaf(acc[t][n][r], sc, sh));
                }
        }
    }
}

}  // namespace

extern "C" int64_t mvs_fpn_packed_floats(int Cout) {
    if (Cout != 8 && Cout != 16 && Cout != 32) return -1;
    return (int64_t)(FC / CCH) * fpn_chunk_floats(Cout);
}

extern "C" int mvs_fpn_pack_weights(const float* w, int Cout, float* packed, mvs_stream_t stream) {
    MVS_REQUIRE(w && packed, "mvs_fpn_pack_weights: null pointer");
    MVS_REQUIRE(Cout == 8 || Cout == 16 || Cout == 32, "mvs_fpn_pack_weights: Cout must be 8, 16 or 32 (got %d)", Cout);
    const int total = (int)mvs_fpn_packed_floats(Cout);
    hipLaunchKernelGGL(fpn_pack_kernel, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w, Cout, packed);
    return mvs::finish_launch("mvs_fpn_pack_weights");
}

extern "C" int mvs_fpn_out0(const float* x, const float* w, const float* scale, const float* shift, int N, int h, int wd, float* out,
                            mvs_stream_t stream) {
    MVS_REQUIRE(x && w && scale && shift && out, "mvs_fpn_out0: null pointer");
    MVS_REQUIRE(N >= 1 && N <= 65535 && h >= 1 && wd >= 1, "mvs_fpn_out0: bad shape N=%d h=%d w=%d", N, h, wd);
    const int hw = h * wd;
    hipLaunchKernelGGL(fpn_out0_kernel, dim3(mvs::ceil_div(hw, 64), N), dim3(64), 0, MVS_STREAM(stream), x, w, scale, shift, hw, out);
    return mvs::finish_launch("mvs_fpn_out0");
}

extern "C" int mvs_fpn_level_layout(const float* intra_prev, const float* lateral, const float* w_inner_p, const float* b_inner,
                                    const float* w_packed, const float* scale, const float* shift, int N, int Ck, int h, int w,
                                    float* intra_out, int intra_nhwc, float* out, mvs_stream_t stream);

extern "C" int mvs_fpn_level(const float* intra_prev, const float* lateral, const float* w_inner_p, const float* b_inner,
                             const float* w_packed, const float* scale, const float* shift, int N, int Ck, int h, int w,
                             float* intra_out, float* out, mvs_stream_t stream) {
    return mvs_fpn_level_layout(intra_prev, lateral, w_inner_p, b_inner, w_packed, scale, shift, N, Ck, h, w, intra_out, 0, out, stream);
}

extern "C" int mvs_fpn_level_layout(const float* intra_prev, const float* lateral, const float* w_inner_p, const float* b_inner,
                                    const float* w_packed, const float* scale, const float* shift, int N, int Ck, int h, int w,
                                    float* intra_out, int intra_nhwc, float* out, mvs_stream_t stream) {
    MVS_REQUIRE(intra_prev && lateral && w_inner_p && b_inner && w_packed && scale && shift && out, "mvs_fpn_level: null pointer");
    MVS_REQUIRE(Ck == 8 || Ck == 16 || Ck == 32, "mvs_fpn_level: lateral channels must be 8, 16 or 32 (got %d)", Ck);
    MVS_REQUIRE(N >= 1 && N <= 65535 && h >= 1 && w >= 1 && (int64_t)2 * h <= 4 * 65535, "mvs_fpn_level: bad shape N=%d h=%d w=%d", N, h, w);
    MVS_REQUIRE((int64_t)FC * 4 * h * w < ((int64_t)1 << 31), "mvs_fpn_level: one image's 64-channel level exceeds 2^31 elements (32-bit in-image offsets)");
    const dim3 grid(mvs::ceil_div(2 * w, TW), mvs::ceil_div(2 * h, TH), N), block(256);
    hipStream_t s = MVS_STREAM(stream);
    if (Ck == 8)
        hipLaunchKernelGGL(fpn_level_kernel<8>, grid, block, 0, s, intra_prev, lateral, w_inner_p, b_inner, w_packed, scale, shift, h, w, intra_out, out, intra_nhwc);
    else if (Ck == 16)
        hipLaunchKernelGGL(fpn_level_kernel<16>, grid, block, 0, s, intra_prev, lateral, w_inner_p, b_inner, w_packed, scale, shift, h, w, intra_out, out, intra_nhwc);
    else
        hipLaunchKernelGGL(fpn_level_kernel<32>, grid, block, 0, s, intra_prev, lateral, w_inner_p, b_inner, w_packed, scale, shift, h, w, intra_out, out, intra_nhwc);
    return mvs::finish_launch("mvs_fpn_level");
}
