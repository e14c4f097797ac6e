// FPN decoder, full-resolution level (models/module.py:266-268: out3 = Swish(BN(conv3x3(up2(intra2) + inner3(conv01))))), second split-form
// formulation: THE CHANNEL CONTRACTION MOVES IN FRONT OF THE UPSAMPLING.  fpn_x3.hip interpolates 64 channels to full resolution, splits every
// value into three bf16 terms and convolves - it is bound by that vector work (profiles/r06_pmc_fpn_level.txt: ~600 vector instructions per
// 84 MFMAs).  Both the convolution and ATen's bilinear interpolation are linear, so
//     conv3x3(up(prev))[p] = sum_taps W_tap . sum_q u(p + tap, q) prev[q] = sum_taps sum_q u(p + tap, q) (W_tap . prev[q])
// with u the (per-pixel, align_corners=True) interpolation weights: the 64 -> 8 contraction P_tap[q] = W_tap . prev[q] runs at the COARSE
// resolution on the bf16 matrix cores (M = 9 taps x 8 channels, K = 64, N = coarse pixels: a quarter of the pixels, no per-pixel split of an
// upsampled value), and the fine level only blends 8-vectors: 36 (tap, corner) pairs x 8 fp32 fmas per pixel, zero weight where p + tap falls
// into intra3's zero padding.  The lateral path stays the composed 8 -> 8 3x3 convolution of fpn_x3.hip (its operands, shift and border table
// are shared), as 18 MFMAs per row pair.
//
// One block = one 16 x 16 fine tile at a time (persistent, a wavefront's weights of the contraction in 48 VGPRs):
//   stage   the 11 x 11 coarse window of all 64 channels (CHANNEL-LAST sources: 16-byte loads), split, as the B operand [term][octet][q][8 bf16];
//           the 18 x 18 lateral tile, split
//   phase 1 P[(tap, co)][q]: 5 M tiles x 2 K steps per 16 coarse pixels; wavefront w owns M tile w over all eight N tiles + the fifth over two
//   phase 1b P -> LDS [q][tap][co] fp32 over the operand it came from; the lateral 3x3 as two row pairs per wavefront -> LDS [pixel][co]
//   phase 2 one thread per fine pixel: 9 taps x 4 corners of P, BatchNorm shift (+ the bias' border correction), Swish, 32-byte store
// The next tile's global loads are in flight during phase 2.
#include "conv_common.h"
#include "split3.h"

// experiment builds only (make exp EXPFLAGS=-DFPNCP_TIMING): per-phase cycle totals of one block's wavefronts, printed by the launcher
#ifdef FPNCP_TIMING
#define TSTAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;

constexpr int FC = 64, CK = 8, T = 16, HT = T + 2;       // fine tile, with the 3x3 halo
constexpr int WQ = 11, NQ = 128;                          // coarse window edge; q slots (121 used)
constexpr int OCTB = NQ * 16, TERMB = 8 * OCTB;           // B operand: [term][octet][q][16 B]
constexpr int B_BYTES = 3 * TERMB;                        // 49152
constexpr int PST = 84 * 4;                               // bytes of a q's row of P (72 floats used; 84 keeps 16-byte alignment and spreads the banks)
static_assert(NQ * PST <= B_BYTES, "P lives over the operand it was computed from");
constexpr int LT = HT * HT * 16;                          // lateral tile, one term: [18][18][8 bf16]
constexpr int LAT_BYTES = ((3 * LT + 255) / 256) * 256;   // 15616
constexpr int L_BYTES = T * T * CK * 4;                   // lateral convolution's result [pixel][co]
constexpr int LDS_BYTES = B_BYTES + LAT_BYTES + L_BYTES + 512;
constexpr int NW = 5 * 2 * 3, NWL = 3 * 3;                // weight fragments: contraction [M tile][K step][term], lateral [step][term]

__device__ __forceinline__ float swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// the block exchanges data through LDS only: no wait for global loads / stores at a barrier (__syncthreads() is a workgroup-scope release)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], c, 0, 0, 0);
    return c;
}

// prepared[frag][lane][8], lane = kb * 16 + m; the BatchNorm scale of the output channel is multiplied in before the split
//   frag < 30: (M tile mt, K step ks, term): row idx = 16 mt + m = (tap = idx / 8, co = idx % 8) (idx >= 72: zero), channel 32 ks + 8 kb + e
//   frag >= 30: lateral (step s, term): m = (dy, co), K block 4 s + kb = (input row j, kw) of a row pair, lateral channel e, weight of kh = j - dy
__global__ void fpn8_cp_prepare_kernel(const float* __restrict__ w3 /*[8,64,3,3]*/, const float* __restrict__ wc /*[8,8,3,3]*/,
                                       const float* __restrict__ scale, bf16x8* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (NW + NWL) * 64) return;
    const int lane = idx & 63, frag = idx >> 6, m = lane & 15, kb = lane >> 4;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = 0.0f;
        int term;
        if (frag < NW) {
            term = frag % 3;
            const int ks = (frag / 3) % 2, mt = frag / 6, row = 16 * mt + m, tap = row / 8, co = row % 8, c = 32 * ks + 8 * kb + e;
            if (row < 72) f = w3[((co * FC + c) * 3 + tap / 3) * 3 + tap % 3] * scale[co];
        } else {
            term = (frag - NW) % 3;
            const int s = (frag - NW) / 3, t = 4 * s + kb, j = t / 3, kw = t % 3, dy = m >> 3, co = m & 7, kh = j - dy;
            if (kh >= 0 && kh <= 2) f = wc[((co * CK + e) * 3 + kh) * 3 + kw] * scale[co];
        }
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

__global__ __launch_bounds__(256, 2) void fpn8_cp_kernel(const float* __restrict__ prev /*[N,h,w,64]*/, const float* __restrict__ lat /*[N,2h,2w,8]*/,
                                                         const bf16x8* __restrict__ prep, const float* __restrict__ shift /*[8]*/,
                                                         const float* __restrict__ border /*[9][8]*/, int h, int w, int ntx, int nty, int ntiles,
                                                         float* __restrict__ out /*[N,2h,2w,8]*/, long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned char* sB = smem;                                 // the B operand, then P
    unsigned char* sLat = smem + B_BYTES;
    unsigned char* sL = sLat + LAT_BYTES;
    float* sborder = reinterpret_cast<float*>(sL + L_BYTES);  // [9][8] + shift [8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;
    const int H = 2 * h, W = 2 * w;
    // ATen's upsample_bilinear2d, align_corners=True: source index = dst * (in - 1) / (out - 1), in float
    const float sy = (float)(h - 1) / (float)(H - 1), sx = (float)(w - 1) / (float)(W - 1);
    if (tid < 9 * CK) sborder[tid] = border[tid];
    if (tid < CK) sborder[9 * CK + tid] = shift[tid];

    // this wavefront's rows of P: M tile `wave` (taps 2 wave, 2 wave + 1) for every coarse pixel, and the half-empty fifth M tile (tap 8) for the
    // coarse pixels 32 wave .. 32 wave + 31: 12 weight fragments = 48 VGPRs (all five M tiles per wavefront would be 120)
    bf16x8 wA[2][3], wE[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            wA[ks][t] = prep[((wave * 2 + ks) * 3 + t) * 64 + lane];
            wE[ks][t] = prep[((4 * 2 + ks) * 3 + t) * 64 + lane];
        }

    // ---- staging roles: both sources are CHANNEL-LAST, so a unit's 8 channels are 32 contiguous bytes = two 16-byte loads (from NCHW tensors the
    //      same data took 48 scattered dword loads per thread and tile: 44-byte runs, a quarter of every line fetched used - measured 2.3x slower) ----
    // coarse window: unit u = tid + 256 i = (q = u % 128, octet = u / 128 = 2 i + (wave >> 1))
    const int q = tid & 127, qr = q / WQ, qc = q % WQ;
    const unsigned qoff = (unsigned)(qr * w + qc) * (FC * 4u);
    // lateral tile: unit u = tid + 256 i < 324 = (halo row, halo column)
    int lhr[2], lhc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 256 * i;
        lhr[i] = u / HT;
        lhc[i] = u % HT;
    }
    f32x4 preg[4][2], lreg[2][2];
    auto tile_origin = [&](int tile, int& img, int& x0, int& y0) {
        img = tile / (ntx * nty);
        y0 = ((tile / ntx) % nty) * T;
        x0 = (tile % ntx) * T;
    };
    auto issue = [&](int tile) {
        int img, x0, y0;
        tile_origin(tile, img, x0, y0);
        const int wy0 = (int)(sy * (float)max(y0 - 1, 0)), wx0 = (int)(sx * (float)max(x0 - 1, 0));
        const rsrc_t rprev = make_rsrc(prev + (size_t)img * FC * h * w, (unsigned)(FC * h * w) * 4u);
        const rsrc_t rlat = make_rsrc(lat + (size_t)img * CK * H * W, (unsigned)(CK * H * W) * 4u);
        const unsigned poff = (q < WQ * WQ && wy0 + qr < h && wx0 + qc < w) ? qoff + (unsigned)(wy0 * w + wx0) * (FC * 4u) : OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned oct = (unsigned)(2 * i + (wave >> 1));
            preg[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rprev, poff, oct * 32u, 0));
            preg[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rprev, poff, oct * 32u + 16u, 0));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gy = y0 - 1 + lhr[i], gx = x0 - 1 + lhc[i];
            const unsigned loff = (tid + 256 * i < HT * HT && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)(gy * W + gx) * (CK * 4u) : OOB;
            lreg[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rlat, loff, 0, 0));
            lreg[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rlat, loff, 16u, 0));
        }
    };
    auto split8 = [&](const f32x4 (&v)[2]) {
        const float f[8] = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
        return mvsx3::split3(f);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const mvsx3::Split3 s = split8(preg[i]);
            unsigned char* dst = sB + (2 * i + (wave >> 1)) * OCTB + q * 16;
            *reinterpret_cast<bf16x8*>(dst) = s.h;
            *reinterpret_cast<bf16x8*>(dst + TERMB) = s.m;
            *reinterpret_cast<bf16x8*>(dst + 2 * TERMB) = s.l;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (tid + 256 * i < HT * HT) {
                const mvsx3::Split3 s = split8(lreg[i]);
                unsigned char* dst = sLat + (tid + 256 * i) * 16;
                *reinterpret_cast<bf16x8*>(dst) = s.h;
                *reinterpret_cast<bf16x8*>(dst + LT) = s.m;
                *reinterpret_cast<bf16x8*>(dst + 2 * LT) = s.l;
            }
        }
    };

    // lateral B operand of this lane's K block per step: t = 4 s + kb = (row j, kw) -> ((j * 18) + kw + n) * 16
    unsigned loff3[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) loff3[s] = (unsigned)((((4 * s + kb) / 3) * HT + (4 * s + kb) % 3 + n) * 16);
    // phase 2: this thread's pixel
    const int py = tid >> 4, px = tid & 15;

    // XCD-aware tile order: consecutive block ids go round-robin to the 8 XCDs (each with its own L2), so XCD x takes the x-th eighth of the tiles
    // and its blocks walk it together - the window rows neighbouring tiles share are fetched into ONE L2, at about the same time
    const int nx = (int)gridDim.x / 8, xcd = (int)blockIdx.x % 8, jx = (int)blockIdx.x / 8;
    const bool by_xcd = nx >= 1 && (int)gridDim.x % 8 == 0;
    const int tbeg = by_xcd ? (int)((long long)ntiles * xcd / 8) : 0, tend = by_xcd ? (int)((long long)ntiles * (xcd + 1) / 8) : ntiles;
    const int tstep = by_xcd ? nx : (int)gridDim.x;
    int tile = tbeg + (by_xcd ? jx : (int)blockIdx.x);
    if (tile >= tend) return;
    issue(tile);
#ifdef FPNCP_TIMING
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    int ntl = 0;
#endif
    for (;;) {
        int img, x0, y0;
        tile_origin(tile, img, x0, y0);
        const int wy0 = (int)(sy * (float)max(y0 - 1, 0)), wx0 = (int)(sx * (float)max(x0 - 1, 0));
        TSTAMP(0);
        commit();
        TSTAMP(1);
        lds_barrier();
        TSTAMP(2);

        // ---- phase 1: P[(tap, co)][q]: this wavefront's M tile over the eight N tiles, the fifth M tile over its own two ----
        f32x4 acc[8], accE[2];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        accE[0] = accE[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            bf16x8 xb[2][2][3];                               // [buffer][K step][term]
            auto fetch = [&](int nt, bf16x8 (&x)[2][3]) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < 3; ++t) x[ks][t] = *reinterpret_cast<const bf16x8*>(sB + t * TERMB + (ks * 4 + kb) * OCTB + (nt * 16 + n) * 16);
            };
            fetch(0, xb[0]);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                if (nt + 1 < 8) fetch(nt + 1, xb[(nt + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                acc[nt] = mfma6(wA[0], xb[nt & 1][0], acc[nt]);
                if ((nt >> 1) == wave) accE[nt & 1] = mfma6(wE[0], xb[nt & 1][0], accE[nt & 1]);
                acc[nt] = mfma6(wA[1], xb[nt & 1][1], acc[nt]);
                if ((nt >> 1) == wave) accE[nt & 1] = mfma6(wE[1], xb[nt & 1][1], accE[nt & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        TSTAMP(3);
        lds_barrier();                                        // every wavefront has read its B fragments: P may overwrite them
        TSTAMP(4);

        // ---- phase 1b: P -> LDS [q][tap * 8 + co]; the lateral 3x3 of row pairs 2 wave, 2 wave + 1 -> LDS [pixel][co] ----
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<f32x4*>(sB + (nt * 16 + n) * PST + (16 * wave + 4 * kb) * 4) = acc[nt];
        if (kb < 2) {                                         // rows 72..79 do not exist
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(sB + ((2 * wave + j) * 16 + n) * PST + (64 + 4 * kb) * 4) = accE[j];
        }
        {
            bf16x8 wl[3][3];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int t = 0; t < 3; ++t) wl[s][t] = prep[(NW + s * 3 + t) * 64 + lane];
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                bf16x8 xa[3], xc[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    xa[t] = *reinterpret_cast<const bf16x8*>(sLat + t * LT + (4 * wave) * (HT * 16) + loff3[s]);
                    xc[t] = *reinterpret_cast<const bf16x8*>(sLat + t * LT + (4 * wave + 2) * (HT * 16) + loff3[s]);
                }
                c0 = mfma6(wl[s], xa, c0);
                c1 = mfma6(wl[s], xc, c1);
            }
            // D[m = (dy, co)][n = x]: this lane holds row parity kb >> 1, channels (kb & 1) * 4 + r
            *reinterpret_cast<f32x4*>(sL + (((4 * wave + (kb >> 1)) * T + n) * CK + (kb & 1) * 4) * 4) = c0;
            *reinterpret_cast<f32x4*>(sL + (((4 * wave + 2 + (kb >> 1)) * T + n) * CK + (kb & 1) * 4) * 4) = c1;
        }
        TSTAMP(5);
        lds_barrier();
        TSTAMP(6);

        __builtin_amdgcn_sched_barrier(0);
        // ---- the next tile's loads fly during phase 2 ----
        const int next = tile + tstep;
        if (next < tend) issue(next);

        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 2: blend the partials at this thread's pixel ----
        {
            const int gy = y0 + py, gx = x0 + px;
            float o[8];
            {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sL + (tid * CK) * 4), b = *reinterpret_cast<const f32x4*>(sL + (tid * CK + 4) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = a[r] + sborder[9 * CK + r];
                    o[4 + r] = b[r] + sborder[9 * CK + 4 + r];
                }
            }
            // the row loop stays rolled on purpose: fully unrolled, the 36 (tap, corner) terms of this phase cost > 100 registers and everything else
            // spills; the three columns' taps are computed once and unrolled with a scheduling fence between them
            unsigned cx0[3], cx1[3];
            float wxa[3], wxb[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = gx + kw - 1;
                const float fx = sx * (float)min(max(xx, 0), W - 1);
                const int ix0 = (int)fx;
                const float gx_ = (xx >= 0 && xx < W) ? 1.0f : 0.0f;
                wxb[kw] = (fx - (float)ix0) * gx_;
                wxa[kw] = (1.0f - (fx - (float)ix0)) * gx_;
                const int rx0 = min(max(ix0 - wx0, 0), WQ - 1), rx1 = min(rx0 + (ix0 < w - 1 ? 1 : 0), WQ - 1);
                cx0[kw] = (unsigned)(rx0 * PST + kw * 32);
                cx1[kw] = (unsigned)(rx1 * PST + kw * 32);
            }
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh) {
                const int yy = gy + kh - 1;
                const float fy = sy * (float)min(max(yy, 0), H - 1);
                const int iy0 = (int)fy;
                const float gy_ = (yy >= 0 && yy < H) ? 1.0f : 0.0f;
                const float wy1 = (fy - (float)iy0) * gy_, wy0_ = (1.0f - (fy - (float)iy0)) * gy_;
                const int ry0 = min(max(iy0 - wy0, 0), WQ - 1), ry1 = min(ry0 + (iy0 < h - 1 ? 1 : 0), WQ - 1);
                const unsigned char* row0 = sB + ry0 * (WQ * PST) + kh * 96;
                const unsigned char* row1 = sB + ry1 * (WQ * PST) + kh * 96;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const f32x4 a00 = *reinterpret_cast<const f32x4*>(row0 + cx0[kw]), b00 = *reinterpret_cast<const f32x4*>(row0 + cx0[kw] + 16);
                    const f32x4 a01 = *reinterpret_cast<const f32x4*>(row0 + cx1[kw]), b01 = *reinterpret_cast<const f32x4*>(row0 + cx1[kw] + 16);
                    const f32x4 a10 = *reinterpret_cast<const f32x4*>(row1 + cx0[kw]), b10 = *reinterpret_cast<const f32x4*>(row1 + cx0[kw] + 16);
                    const f32x4 a11 = *reinterpret_cast<const f32x4*>(row1 + cx1[kw]), b11 = *reinterpret_cast<const f32x4*>(row1 + cx1[kw] + 16);
                    const float w00 = wy0_ * wxa[kw], w01 = wy0_ * wxb[kw], w10 = wy1 * wxa[kw], w11 = wy1 * wxb[kw];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = fmaf(w11, a11[r], fmaf(w10, a10[r], fmaf(w01, a01[r], fmaf(w00, a00[r], o[r]))));
                        o[4 + r] = fmaf(w11, b11[r], fmaf(w10, b10[r], fmaf(w01, b01[r], fmaf(w00, b00[r], o[4 + r]))));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (gy < H && gx < W) {
                if (gy == 0 || gy == H - 1 || gx == 0 || gx == W - 1) {   // the bias' response loses the taps that fall into the zero padding
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int yy = gy + tap / 3 - 1, xx = gx + tap % 3 - 1;
                        if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) o[r] -= sborder[tap * CK + r];
                        }
                    }
                }
                f32x4 v0, v1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = swish(o[r]);
                    v1[r] = swish(o[4 + r]);
                }
                float* dst = out + (((size_t)img * H + gy) * W + gx) * CK;
                *reinterpret_cast<f32x4*>(dst) = v0;
                *reinterpret_cast<f32x4*>(dst + 4) = v1;
            }
        }
        TSTAMP(7);
        lds_barrier();                                        // P, the lateral tile and its result are consumed
        TSTAMP(8);
#ifdef FPNCP_TIMING
        ++ntl;
#endif
        if (next >= tend) break;
        tile = next;
    }
#ifdef FPNCP_TIMING
    if (dbg && blockIdx.x == gridDim.x / 2 + 3 && lane == 0) {
        for (int i = 0; i < 9; ++i) dbg[wave * 10 + i] = tacc[i];
        dbg[wave * 10 + 9] = ntl;
    }
#endif
}

}  // namespace

extern "C" int64_t mvs_fpn_level_cp_prepared_bytes(int Ck) { return Ck == 8 ? (int64_t)(NW + NWL) * 64 * 16 : -1; }

extern "C" int mvs_fpn_level_cp_prepare(const float* w3, const float* wc, const float* scale, int Ck, void* prepared, mvs_stream_t stream) {
    MVS_REQUIRE(w3 && wc && scale && prepared, "mvs_fpn_level_cp_prepare: null pointer");
    MVS_REQUIRE(Ck == 8, "mvs_fpn_level_cp_prepare: built for the full-resolution level, Ck = 8 (got %d)", Ck);
    constexpr int total = (NW + NWL) * 64;
    hipLaunchKernelGGL(fpn8_cp_prepare_kernel, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w3, wc, scale,
                       static_cast<bf16x8*>(prepared));
    return mvs::finish_launch("mvs_fpn_level_cp_prepare");
}

extern "C" int mvs_fpn_level_cp(const float* intra_prev, const float* lateral, const void* prepared, const float* shift, const float* border,
                                int N, int Ck, int h, int w, float* out, mvs_stream_t stream) {
    MVS_REQUIRE(intra_prev && lateral && prepared && shift && border && out, "mvs_fpn_level_cp: null pointer");
    MVS_REQUIRE(Ck == 8, "mvs_fpn_level_cp: built for the full-resolution level, Ck = 8 (got %d)", Ck);
    MVS_REQUIRE(N >= 1 && h >= 1 && w >= 1, "mvs_fpn_level_cp: bad shape N=%d h=%d w=%d", N, h, w);
    MVS_REQUIRE((int64_t)FC * h * w * 4 < ((int64_t)1 << 31), "mvs_fpn_level_cp: one image's 64-channel level exceeds 2 GiB");
    const int H = 2 * h, W = 2 * w, ntx = mvs::ceil_div(W, T), nty = mvs::ceil_div(H, T);
    const int64_t ntiles = (int64_t)N * ntx * nty;
    MVS_REQUIRE(ntiles < ((int64_t)1 << 31), "mvs_fpn_level_cp: too many tiles");
    const int slots = 2 * mvs::device_cus();
    const int blocks = ntiles < slots ? (int)ntiles : slots;   // persistent: two resident blocks per CU
    {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(&fpn8_cp_kernel), LDS_BYTES, "mvs_fpn_level_cp");
        if (rc != MVS_OK) return rc;
    }
    long long* dbg = nullptr;
#ifdef FPNCP_TIMING
    static long long* dbg_buf = nullptr;
    static int dbg_calls = 0;
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 64 * sizeof(long long));
    dbg = dbg_buf;
#endif
    hipLaunchKernelGGL(fpn8_cp_kernel, dim3(blocks), dim3(256), LDS_BYTES, MVS_STREAM(stream), intra_prev, lateral,
                       static_cast<const bf16x8*>(prepared), shift, border, h, w, ntx, nty, (int)ntiles, out, dbg);
#ifdef FPNCP_TIMING
    if (dbg_calls++ == 3) {
        long long hbuf[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hbuf, dbg_buf, sizeof(hbuf), hipMemcpyDeviceToHost);
        fprintf(stderr, "fpn8_cp, a middle block: cycles per tile [loop top, commit, barrier, phase 1, barrier, P + lateral, barrier, phase 2 (+issue), barrier]\n");
        for (int wv = 0; wv < 4; ++wv) {
            fprintf(stderr, "  wave %d (%lld tiles):", wv, hbuf[wv * 10 + 9]);
            for (int i = 0; i < 9; ++i) fprintf(stderr, " %7lld", hbuf[wv * 10 + i] / (hbuf[wv * 10 + 9] ? hbuf[wv * 10 + 9] : 1));
            fprintf(stderr, "\n");
        }
    }
#endif
    return mvs::finish_launch("mvs_fpn_level_cp");
}
