// FPN decoder, full-resolution level (models/module.py:266-268: intra3 = up2(intra2) + inner3(conv01); out3 = Swish(BN(conv3x3 64 -> 8)))
// in the THREE-TERM BF16 SPLIT form of conv3d_x3.hip / vis_net_x3.hip: fp32 in, fp32 out, every product as six v_mfma_f32_16x16x32_bf16 of
// the exact splits x = h + m + l, fp32 accumulation (split3.h).  It replaces fpn_level_kernel<8> (fpn.hip: fp32 matrix cores, the lateral
// 1x1 convolution on the vector ALU), the largest single launch of the reference's timed region in images -> depth (1.45 ms of 13.3).
//
// Three reformulations carry it:
//   * conv3x3 is linear, so conv3x3(up + W_in . lat + b_in) = conv3x3(up) + (W3 o W_in)(lat) + the bias' own response: the lateral path is
//     an 8 -> 8 3x3 convolution with weights composed ONCE on the host in fp64 (12 more K blocks beside the 96 of the 64 top-down channels),
//     not 64 x 8 vector fmas per pixel; the bias' response is a constant folded into the BatchNorm shift, corrected at the image border
//     (where the zero padding of intra3 removes taps) by a 9 x 8 table;
//   * K SPLIT OVER THE BLOCK'S FOUR WAVEFRONTS: wavefront w owns top-down channels 16w .. 16w+15 and input row j = w of the lateral path.
//     Its weights (7 steps x 3 terms = 84 VGPRs) stay in registers for the whole launch; it builds, in a WAVE-PRIVATE LDS ring, the
//     upsampled rows of its own channels (bilinear taps from a wave-private window of the coarser level, split, stored as
//     [term][octet][row][pixel][8 bf16] - the B operand of a K block is one conflict-free ds_read_b128) and multiplies them: no block
//     barrier between building and multiplying, so the wavefronts of a CU drift apart and one's vector work runs under another's MFMAs.
//     The four partial tiles meet once per four output rows through LDS (fixed order: deterministic);
//   * a block walks DOWN a 16-column strip: the ring keeps the two rows the next four output rows share with the previous ones, so the
//     halo is recomputed only sideways (18 / 16).
// Output rows come in pairs: M = (row parity, 8 channels), K = 4 input rows x 3 columns x channels (a quarter of the A operand is
// structural zeros instead of half, as layer 3 of vis_net_x3.hip).
#include "conv_common.h"
#include "split3.h"

#ifndef FPNX3_ABLATE
#define FPNX3_ABLATE 0      // experiment builds only (make exp EXPFLAGS=-DFPNX3_ABLATE=bits): skip a phase to time the others
#endif

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;

constexpr int FC = 64, CK = 8;
constexpr int TW = 16, HC = TW + 2;                   // strip width, with the halo
constexpr int RB = 4;                                 // output rows per batch (two row pairs)
constexpr int ROWB = HC * 16;                         // bytes of one (term, octet, row): 18 pixels x 8 bf16
constexpr int RING = 6;                               // rows Y-1 .. Y+4 of a batch
constexpr int OCT = 1792;                             // 6 rows = 1728 B, padded to a multiple of 256 (the two octets of a read hit disjoint banks)
constexpr int TERM = 2 * OCT;
constexpr int RING_BYTES = 3 * TERM;                  // 10752
constexpr int LAT_TERM = 2 * ROWB;                    // the wave's lateral row of each of the two row pairs
constexpr int LAT_BYTES = 1792;                       // 3 * 576 = 1728, padded
constexpr int WIN_ROWS = 4, WIN_COLS = 12, WIN_ROWB = WIN_COLS * 16 + 16, WIN_QUAD = WIN_ROWS * WIN_ROWB + 16;   // [channel quad][row][col][4 fp32], padded against bank conflicts
constexpr int WIN_BYTES = 3584;                       // 4 * 848 = 3392, padded to a multiple of 256
constexpr int WREG = RING_BYTES + LAT_BYTES + WIN_BYTES;   // 16128 per wavefront
constexpr int RED_BUF = 4 * 2 * 1024;                 // [wave][row pair][lane][16 B]
constexpr int LDS_BYTES = 4 * WREG + 2 * RED_BUF;     // 80896: two blocks per CU
constexpr int STEPS = 7;                              // 24 K blocks of the wave's 16 channels + 3 of its lateral row (+1 zero block)
constexpr int UNITS = RB * HC * 4;                    // build units of a batch: (pixel of the four new rows, channel quad)
constexpr int PASSES = (UNITS + 63) / 64;             // 5 (the last one half full)
static_assert(WREG % 256 == 0 && OCT % 256 == 0, "bank alignment of the fragment reads");

__device__ __forceinline__ float swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// LDS traffic between the phases of ONE wavefront needs no barrier (a wavefront's DS instructions execute in order); this keeps the compiler
// from moving accesses across the phase boundary and drains the queue
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// prepared[(wave * STEPS + step) * 3 + term][lane][8]: the MFMA A operand, lane = kb * 16 + m, m = (dy = m >> 3, co = m & 7); the BatchNorm
// scale of the output channel is multiplied in before the split.
//   step < 6: K block 2*step + (kb >> 1) = (input row j, kw) of a row pair, channels 16*wave + (kb & 1)*8 + e, weight of kh = j - dy
//   step 6:   kb = kw of the composed lateral weights at input row j = wave (kb 3: zero)
__global__ void fpn8_x3_prepare_kernel(const float* __restrict__ w3 /*[8,64,3,3]*/, const float* __restrict__ wc /*[8,8,3,3]*/,
                                       const float* __restrict__ scale, bf16x8* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * STEPS * 3 * 64) return;
    const int lane = idx & 63, term = (idx >> 6) % 3, step = (idx / 192) % STEPS, wave = idx / (192 * STEPS);
    const int m = lane & 15, kb = lane >> 4, dy = m >> 3, co = m & 7;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = 0.0f;
        if (step < 6) {
            const int t = 2 * step + (kb >> 1), j = t / 3, kw = t % 3, kh = j - dy, c = 16 * wave + (kb & 1) * 8 + e;
            if (kh >= 0 && kh <= 2) f = w3[((co * FC + c) * 3 + kh) * 3 + kw] * scale[co];
        } else if (kb < 3) {
            const int kh = wave - dy;
            if (kh >= 0 && kh <= 2) f = wc[((co * CK + e) * 3 + kh) * 3 + kb] * scale[co];
        }
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], c, 0, 0, 0);
    return c;
}

__global__ __launch_bounds__(256, 2) void fpn8_x3_kernel(const float* __restrict__ prev /*[N,64,h,w]*/, const float* __restrict__ lat /*[N,8,2h,2w]*/,
                                                         const bf16x8* __restrict__ prep, const float* __restrict__ shift /*[8]*/,
                                                         const float* __restrict__ border /*[9][8]*/, int h, int w, int nstrips, int nseg,
                                                         int seg_rows, float* __restrict__ out /*[N,2h,2w,8]*/) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;
    unsigned char* ring = smem + wave * WREG;
    unsigned char* slat = ring + RING_BYTES;
    unsigned char* swin = slat + LAT_BYTES;
    unsigned char* sred = smem + 4 * WREG;

    const int H = 2 * h, W = 2 * w;
    const unsigned item = xcd_linear_block_id();
    const int strip = (int)(item % (unsigned)nstrips), seg = (int)((item / (unsigned)nstrips) % (unsigned)nseg), img = (int)(item / (unsigned)(nstrips * nseg));
    const int x0 = strip * TW, ys = seg * seg_rows, yend = min(ys + seg_rows, H);
    // ATen's upsample_bilinear2d, align_corners=True: source index = dst * (in - 1) / (out - 1), in float
    const float sy = (float)(h - 1) / (float)(H - 1), sx = (float)(w - 1) / (float)(W - 1);
    const int wx0 = (int)(sx * (float)max(x0 - 1, 0));

    bf16x8 wgt[STEPS][3];
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) wgt[s][t] = prep[((wave * STEPS + s) * 3 + t) * 64 + lane];

    const rsrc_t rprev = make_rsrc(prev + (size_t)img * FC * h * w, (unsigned)(FC * h * w) * 4u);
    const rsrc_t rlat = make_rsrc(lat + (size_t)img * CK * H * W, (unsigned)(CK * H * W) * 4u);

    // ---- the coarse window of a batch's four new rows: unit v = 64 i + lane = (channel, window row, column quad), one 16-byte load each
    //      (4 columns of a row), scattered into [channel quad][row][col][4 channels].  Nothing is masked: window rows >= h / columns >= w hold
    //      whatever follows in memory (zero past the tensor's end) and are never referenced (the taps clamp like ATen's) ----
    unsigned wgo[3], wlds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v = 64 * i + lane, ch = v / 12, rem = v % 12, row = rem / 3, cq = rem % 3;
        wgo[i] = (unsigned)(((16 * wave + ch) * h + row) * w + wx0 + 4 * cq) * 4u;
        wlds[i] = (unsigned)((ch >> 2) * WIN_QUAD + row * WIN_ROWB + 4 * cq * 16 + (ch & 3) * 4);
    }
    f32x4 wreg[3];
    auto window_issue = [&](int g0) {
        const int wy0 = (int)(sy * (float)min(max(g0, 0), H - 1));
        const unsigned rowpart = (unsigned)(wy0 * w) * 4u;
#pragma unroll
        for (int i = 0; i < 3; ++i) wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rprev, wgo[i] + rowpart, 0, 0));
    };
    auto window_commit = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(swin + wlds[i] + e * 16) = wreg[i][e];
    };
    // ---- the wave's lateral row of each row pair: lane = (row pair, halo column), 8 channel loads (the channel in the scalar offset) ----
    const int lrp = lane / HC, lgx = x0 - 1 + lane % HC;
    const unsigned lgo = (lane < 2 * HC && lgx >= 0 && lgx < W) ? (unsigned)((2 * lrp - 1 + wave) * W + lgx) * 4u : OOB;   // (wraps for the row above the image: masked below)
    const unsigned lchb = (unsigned)(H * W) * 4u;
    float lreg[8];
    auto lat_issue = [&](int Y) {
        const int r = Y + 2 * lrp - 1 + wave;
        const unsigned off = (r >= 0 && r < H) ? lgo + (unsigned)(Y * W) * 4u : OOB;
#pragma unroll
        for (int c = 0; c < 8; ++c) lreg[c] = buf_load(rlat, off, (unsigned)c * lchb);
    };
    auto lat_commit = [&]() {
        if (lane < 2 * HC) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 th, tm, tl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned xh, xm, xl;
                mvsx3::split3_pair<false>(lreg[2 * e], lreg[2 * e + 1], xh, xm, xl);
                th[e] = xh; tm[e] = xm; tl[e] = xl;
            }
            *reinterpret_cast<u32x4*>(slat + lane * 16) = th;
            *reinterpret_cast<u32x4*>(slat + LAT_TERM + lane * 16) = tm;
            *reinterpret_cast<u32x4*>(slat + 2 * LAT_TERM + lane * 16) = tl;
        }
    };

    // ---- build roles: unit u = 64 p + lane = (channel quad, pixel of the four new rows); everything horizontal is fixed for the strip ----
    unsigned bA[PASSES], bB[PASSES];
    float bl[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int u = 64 * p + lane;
        const bool ok = u < UNITS;
        const int uu = ok ? u : 0;
        const int quad = uu / (RB * HC), px = uu % (RB * HC), row = px / HC, col = px % HC;
        const int gx = x0 - 1 + col;
        const bool colok = ok && gx >= 0 && gx < W;
        const float fx = sx * (float)min(max(gx, 0), W - 1);
        const int ix0 = (int)fx, ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
        const int rx0 = min(ix0 - wx0, WIN_COLS - 1), rx1 = min(ix1 - wx0, WIN_COLS - 1);
        bl[p] = colok ? fx - (float)ix0 : -1.0f;          // (-1: a halo column outside the image - zero padding)
        bA[p] = (unsigned)(quad * WIN_QUAD + rx0 * 16) | (unsigned)(quad * WIN_QUAD + rx1 * 16) << 16;     // (+ the window row * WIN_ROWB per batch)
        bB[p] = (unsigned)((quad >> 1) * OCT + col * 16 + (quad & 1) * 8) | (unsigned)(row * 4) << 16 | (ok ? 1u << 21 : 0u);
    }
    auto build = [&](int g0, int slot0) {
        const int wy0 = (int)(sy * (float)min(max(g0, 0), H - 1));
        // vertical taps of the batch's four rows, computed once by lanes 0..3 (every lane evaluates row lane & 3) and fetched per pass by
        // ds_bpermute (a pass's lanes span all four rows): window row offset | row step << 16, and the two weights (0 for a row outside the image)
        int vpk;
        float vl0, vl1;
        {
            const int g = g0 + (lane & 3);
            const float fy = sy * (float)min(max(g, 0), H - 1);
            const int iy0 = (int)fy;
            const int ry0 = min(max(iy0 - wy0, 0), WIN_ROWS - 1), ry1 = min(ry0 + (iy0 < h - 1 ? 1 : 0), WIN_ROWS - 1);
            const float gate = (g >= 0 && g < H) ? 1.0f : 0.0f;
            vl1 = (fy - (float)iy0) * gate;
            vl0 = (1.0f - (fy - (float)iy0)) * gate;
            vpk = ry0 * WIN_ROWB | ((ry1 - ry0) * WIN_ROWB) << 16;
        }
        f32x4 va[2][4];
        float wv[2][2];
        auto taps = [&](int p, f32x4 (&v)[4], float (&ly)[2]) {
            const int sel = (int)(bB[p] >> 16 & 15u);                 // 4 * row = the byte address of lane `row` for ds_bpermute
            const int pk = __builtin_amdgcn_ds_bpermute(sel, vpk);
            ly[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, vl0)));
            ly[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, vl1)));
            const unsigned char* s0 = swin + (bA[p] & 0xffffu) + (pk & 0xffff);
            const unsigned char* s1 = swin + (bA[p] >> 16) + (pk & 0xffff);
            const int dr = pk >> 16;
            v[0] = *reinterpret_cast<const f32x4*>(s0);
            v[1] = *reinterpret_cast<const f32x4*>(s1);
            v[2] = *reinterpret_cast<const f32x4*>(s0 + dr);
            v[3] = *reinterpret_cast<const f32x4*>(s1 + dr);
        };
        taps(0, va[0], wv[0]);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if (p + 1 < PASSES) taps(p + 1, va[(p + 1) & 1], wv[(p + 1) & 1]);     // the next pass's window reads fly during this pass's arithmetic
            __builtin_amdgcn_sched_barrier(0);
            if (bB[p] >> 21 & 1u) {
                const f32x4(&v)[4] = va[p & 1];
                const bool colok = bl[p] >= 0.0f;
                const float lx1 = colok ? bl[p] : 0.0f, lx0 = colok ? 1.0f - bl[p] : 0.0f;
                const float w00 = wv[p & 1][0] * lx0, w01 = wv[p & 1][0] * lx1, w10 = wv[p & 1][1] * lx0, w11 = wv[p & 1][1] * lx1;
                int slot4 = 4 * slot0 + (int)(bB[p] >> 16 & 15u);
                slot4 = slot4 >= 4 * RING ? slot4 - 4 * RING : slot4;
                unsigned char* dst = ring + (bB[p] & 0xffffu) + slot4 * (ROWB / 4);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 th, tm, tl;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float a = fmaf(w11, v[3][2 * r], fmaf(w10, v[2][2 * r], fmaf(w01, v[1][2 * r], w00 * v[0][2 * r])));
                    const float b = fmaf(w11, v[3][2 * r + 1], fmaf(w10, v[2][2 * r + 1], fmaf(w01, v[1][2 * r + 1], w00 * v[0][2 * r + 1])));
                    unsigned xh, xm, xl;
                    mvsx3::split3_pair<false>(a, b, xh, xm, xl);    // convex combinations of finite feature values: no clamp (split3.h)
                    th[r] = xh; tm[r] = xm; tl[r] = xl;
                }
                *reinterpret_cast<u32x2*>(dst) = th;
                *reinterpret_cast<u32x2*>(dst + TERM) = tm;
                *reinterpret_cast<u32x2*>(dst + 2 * TERM) = tl;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- B operand of this lane's K block: (kb & 1) = octet, (kb >> 1) picks one of a step's two (input row, kw) pairs ----
    const unsigned bbase = (unsigned)((kb & 1) * OCT + n * 16);
    const unsigned boff6 = (unsigned)((min(kb, 2) + n) * 16);        // (kb 3 multiplies zero weights: any finite data)
    const bool hi = (kb >> 1) != 0;
    auto multiply = [&](int sbase, f32x4& c0, f32x4& c1) {
        int ro[RING];
#pragma unroll
        for (int k = 0; k < RING; ++k) {
            const int s_ = sbase + k;
            ro[k] = (s_ >= RING ? s_ - RING : s_) * ROWB;
        }
        bf16x8 xa[2][3], xb[2][3];
        auto fetch = [&](int s, bf16x8 (&a)[3], bf16x8 (&b)[3]) {
            if (s < 6) {
                const int jlo = (2 * s) / 3, jhi = (2 * s + 1) / 3, klo = ((2 * s) % 3) * 16, khi = ((2 * s + 1) % 3) * 16;
                const unsigned char* pa = ring + bbase + (hi ? ro[jhi] + khi : ro[jlo] + klo);
                const unsigned char* pb = ring + bbase + (hi ? ro[2 + jhi] + khi : ro[2 + jlo] + klo);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    a[t] = *reinterpret_cast<const bf16x8*>(pa + t * TERM);
                    b[t] = *reinterpret_cast<const bf16x8*>(pb + t * TERM);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    a[t] = *reinterpret_cast<const bf16x8*>(slat + boff6 + t * LAT_TERM);
                    b[t] = *reinterpret_cast<const bf16x8*>(slat + boff6 + ROWB + t * LAT_TERM);
                }
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
    float* out_img = out + (size_t)img * H * W * CK;

    const int nb = (yend - ys + RB - 1) / RB;
    window_issue(ys - 3);
    window_commit();
    window_issue(ys + 1);
    lat_issue(ys);
    wave_lds_fence();
    build(ys - 3, 0);                                        // rows ys-1, ys of the first batch (and two it never reads)
    int slot0 = 4, sbase = 2;
    for (int b = 0; b < nb; ++b) {
        const int Y = ys + RB * b;
        wave_lds_fence();                                    // the previous build / multiply of this wavefront are done with the window / lateral rows
#if !(FPNX3_ABLATE & 8)
        window_commit();
        lat_commit();
        if (b + 1 < nb) {                                    // in flight during this batch's two phases
            window_issue(Y + RB + 1);
            lat_issue(Y + RB);
        }
#endif
        wave_lds_fence();
#if !(FPNX3_ABLATE & 1)
        build(Y + 1, slot0);
#endif
        wave_lds_fence();
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
#if !(FPNX3_ABLATE & 2)
        multiply(sbase, c0, c1);
#endif
        __builtin_amdgcn_s_setprio(0);
        slot0 = slot0 + 4 >= RING ? slot0 + 4 - RING : slot0 + 4;
        sbase = sbase + 4 >= RING ? sbase + 4 - RING : sbase + 4;

        // ---- the four K quarters meet: wavefront w finishes row pair w >> 1, output row parity w & 1 ----
        f32x4* red = reinterpret_cast<f32x4*>(sred + (b & 1) * RED_BUF);
        red[(wave * 2 + 0) * 64 + lane] = c0;
        red[(wave * 2 + 1) * 64 + lane] = c1;
#if !(FPNX3_ABLATE & 4)
        __syncthreads();
#endif
        const int rp = wave >> 1;
        f32x4 s = red[(0 * 2 + rp) * 64 + lane];
        s += red[(1 * 2 + rp) * 64 + lane];
        s += red[(2 * 2 + rp) * 64 + lane];
        s += red[(3 * 2 + rp) * 64 + lane];
        const int gy = Y + 2 * rp + (wave & 1), gx = x0 + n;
        if ((kb >> 1) == (wave & 1) && gy < yend && gx < W) {
            f32x4 v = s + shv;
            if (gy == 0 || gy == H - 1 || gx == 0 || gx == W - 1) {   // the bias' response loses the taps that fall into the zero padding
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int yy = gy + tap / 3 - 1, xx = gx + tap % 3 - 1;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] -= border[tap * CK + (kb & 1) * 4 + r];
                    }
                }
            }
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = swish(v[r]);
            *reinterpret_cast<f32x4*>(out_img + ((size_t)gy * W + gx) * CK + (kb & 1) * 4) = o;
        }
    }
}

}  // namespace

extern "C" int64_t mvs_fpn_level_x3_prepared_bytes(int Ck) { return Ck == 8 ? (int64_t)4 * STEPS * 3 * 64 * 16 : -1; }

extern "C" int mvs_fpn_level_x3_prepare(const float* w3, const float* wc, const float* scale, int Ck, void* prepared, mvs_stream_t stream) {
    MVS_REQUIRE(w3 && wc && scale && prepared, "mvs_fpn_level_x3_prepare: null pointer");
    MVS_REQUIRE(Ck == 8, "mvs_fpn_level_x3_prepare: built for the full-resolution level, Ck = 8 (got %d)", Ck);
    constexpr int total = 4 * STEPS * 3 * 64;
    hipLaunchKernelGGL(fpn8_x3_prepare_kernel, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), w3, wc, scale,
                       static_cast<bf16x8*>(prepared));
    return mvs::finish_launch("mvs_fpn_level_x3_prepare");
}

extern "C" int mvs_fpn_level_x3(const float* intra_prev, const float* lateral, const void* prepared, const float* shift, const float* border,
                                int N, int Ck, int h, int w, float* out, mvs_stream_t stream) {
    MVS_REQUIRE(intra_prev && lateral && prepared && shift && border && out, "mvs_fpn_level_x3: null pointer");
    MVS_REQUIRE(Ck == 8, "mvs_fpn_level_x3: built for the full-resolution level, Ck = 8 (got %d)", Ck);
    MVS_REQUIRE(N >= 1 && h >= 1 && w >= 1, "mvs_fpn_level_x3: bad shape N=%d h=%d w=%d", N, h, w);
    MVS_REQUIRE((int64_t)FC * h * w * 4 < ((int64_t)1 << 31), "mvs_fpn_level_x3: one image's 64-channel level exceeds 2 GiB");
    const int H = 2 * h, W = 2 * w, nstrips = mvs::ceil_div(W, TW), slots = 2 * mvs::device_cus();
    // vertical segments only where they fill the chip's block slots better (each pays a two-row prologue)
    int nseg = 1;
    double best = 0.0;
    for (int s = 1; s <= 16; ++s) {
        const int rows = mvs::ceil_div(mvs::ceil_div(H, s), RB) * RB, ns = mvs::ceil_div(H, rows);
        if (s > 1 && rows < 64) break;
        const int64_t items = (int64_t)N * nstrips * ns;
        const double fill = (double)items / (double)(mvs::ceil_div((long long)items, (long long)slots) * slots);
        if (fill > best + 0.03) { best = fill; nseg = ns; }
    }
    const int seg_rows = mvs::ceil_div(mvs::ceil_div(H, nseg), RB) * RB;
    nseg = mvs::ceil_div(H, seg_rows);
    const int64_t items = (int64_t)N * nstrips * nseg;
    MVS_REQUIRE(items < ((int64_t)1 << 31), "mvs_fpn_level_x3: too many strips");
    {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(&fpn8_x3_kernel), LDS_BYTES, "mvs_fpn_level_x3");
        if (rc != MVS_OK) return rc;
    }
    hipLaunchKernelGGL(fpn8_x3_kernel, dim3((unsigned)items), dim3(256), LDS_BYTES, MVS_STREAM(stream), intra_prev, lateral,
                       static_cast<const bf16x8*>(prepared), shift, border, h, w, nstrips, nseg, seg_rows, out);
    return mvs::finish_launch("mvs_fpn_level_x3");
}
