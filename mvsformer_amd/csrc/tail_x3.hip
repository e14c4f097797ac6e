// The tail of CostRegNet3D in ONE launch on the bf16 matrix cores (three-term split form, split3.h): conv11 = ConvTranspose3d(16 -> 8,
// k 3, stride (1,2,2), padding 1, output_padding (0,1,1), bias=False) -> BatchNorm3d -> ReLU, + the skip tensor (the regularizer's input
// volume), then prob = Conv3d(8, 1, 1) (reference models/module.py:575-576,582,590-592).  The 8-channel feature volume is never
// written: the kernel reads conv9's output [B,16,D,H,W] and the skip volume [B,8,D,2H,2W] and writes the logits [B,D,2H,2W].
//
// It replaces deconv3d_s1_kernel<8, prob> (fp32 MFMA: 0.235 + 0.128 ms per depth map at config 2, the largest regularizer kernel of round
// 3).  Why the generic split-form transposed conv (conv3d_x3.hip: one N tile per output-parity class, 16 channels wide) does not serve it:
// 8 output channels leave half of every tile empty.  Here the N = 16 side of a tile is (column parity pw, channel): the two classes of an
// output ROW parity share the input pixels they read,
//     row parity 0:  classes 00 | 01 read in[i][j], in[i][j+1]                                   -> K = 2 pixels x 16 channels = 1 step of 32
//     row parity 1:  classes 10 | 11 read in[i][j], in[i][j+1], in[i+1][j], in[i+1][j+1]        -> K = 4 pixels x 16 channels = 2 steps
// so a (16 input pixels, depth tap) unit is 3 steps x 6 MFMAs with 18 of 24 K blocks useful (the generic form: 18 of 40).  The weights
// are the MFMA's A operand (M = (pw, channel)), the activations the B operand (N = 16 consecutive input pixels of a row), so
//   * a B fragment depends only on (row, term): the fragment of row i serves tile P of row i, step 0 of tile Q of row i and step 1 of
//     tile Q of row i-1 - a wavefront reads R + 1 row fragments for R rows (3 x 16-byte LDS reads per 54 MFMAs);
//   * ALL pre-split weights (3 depth taps x 3 steps x 3 terms) stay in 108 VGPRs for the whole kernel: the main loop has no weight
//     loads, so the only global loads in flight are next plane's staging and the skip tensor, issued BEFORE the MFMA phase of the
//     current plane and consumed after it (two LDS buffers, one barrier per plane);
//   * a lane ends up with 4 channels of ONE output pixel: BatchNorm, ReLU, skip add and the 1x1x1 conv are 4 in-lane FMAs + one
//     exchange with the lane holding the other 4 channels (xor 16), the two column parities are paired with one more exchange (xor 32)
//     and leave as 8-byte stores, 128 contiguous bytes per instruction.
// A block owns 8 x 16 input pixels (16 x 32 logits) through all depth planes (or a depth segment), sweeping the INPUT planes like
// conv3d_x3.hip: input plane p feeds output planes p-1, p, p+1 (kd = 0, 1, 2), three accumulator sets rotate.
#include <stdlib.h>

#include "conv_common.h"
#include "split3.h"

#ifndef X3_STAGE_AUX
#define X3_STAGE_AUX 0      // cache policy of the staging / skip-tensor loads (aux of raw_buffer_load: 2 = nt)
#endif

// the ablation switches of the timeline experiments exist only in experiment builds: the shipped kernel has no environment-dependent path
#ifdef X3_ABLATE
#define TAIL_ABLATE(a, bit) (((a).ablate & (bit)) != 0)
#else
#define TAIL_ABLATE(a, bit) false
#endif

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;

__device__ __forceinline__ float stage_load(rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, X3_STAGE_AUX));
}
using mvsx3::mfma6;
using mvsx3::Split3;
using mvsx3::split3;

constexpr int CIN = 16, COUT = 8;
constexpr int R = 2;                                       // input rows per wavefront
constexpr int TIH = 4 * R, TIW = 16, BH = TIH + 1, BW = TIW + 1, NPIX = BH * BW, PB = 32, TERM_BYTES = NPIX * PB, BUF_BYTES = 3 * TERM_BYTES;
constexpr int NW = 3 * 3 * 3;                              // weight fragments: [kd][step][term]
constexpr int NI = (2 * NPIX + 255) / 256;                 // staging items (channel octet, box pixel) per thread

// packed[((kd*3 + step)*3 + term)*64 + lane][8]: A[m = lane & 15][K block kb = lane >> 4] of the step; m = (pw = m >> 3, co = m & 7),
// kb = (dj = kb >> 1, oct = kb & 1); step 0: row parity 0 (kh = 1); step 1: row parity 1, input row i (kh = 2); step 2: row parity 1,
// input row i + 1 (kh = 0).  Column: pw = 0: dj = 0 -> kw = 1, dj = 1 -> no tap; pw = 1: dj = 0 -> kw = 2, dj = 1 -> kw = 0.
// w = ConvTranspose3d weight [16][8][3][3][3].
__global__ void tail_x3_pack_kernel(const float* __restrict__ w, bf16x8* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NW * 64) return;
    const int lane = idx & 63, term = (idx >> 6) % 3, step = (idx / 192) % 3, kd = idx / 576;
    const int m = lane & 15, kb = lane >> 4, pw = m >> 3, co = m & 7, dj = kb >> 1, oct = kb & 1;
    const int kh = step == 0 ? 1 : (step == 1 ? 2 : 0);
    const int kw = pw == 0 ? (dj == 0 ? 1 : -1) : (dj == 0 ? 2 : 0);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = kw < 0 ? 0.0f : w[((size_t)(oct * 8 + e) * COUT + co) * 27 + kd * 9 + kh * 3 + kw];
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

struct TailArgs {
    const float* x;            // [B,16,D,H,W]   conv9's output
    const bf16x8* wp;
    const float* scale;        // [8] folded BatchNorm (NULL: 1)
    const float* shift;        // [8] (NULL: 0)
    const float* residual;     // [B,8,D,2H,2W] or NULL
    const float* prob_w;       // [8]
    const float* prob_b;       // [1] or NULL
    float* y;                  // [B,D,2H,2W]
    int B, D, H, W, relu, tiles_x, tiles, seg_planes, nseg, ablate;
};

#ifdef X3_TIMELINE
// experiment build only (make exp NAME=tl EXPSRC="conv3d_x3 tail_x3" EXPFLAGS=-DX3_TIMELINE, tools/x3_timeline.py): wavefront 0 of every block
// sums the clock ticks of each phase of a pass
__device__ unsigned long long tail_phase_ticks[8];
#define TAIL_STAMP(slot)                                                    \
    do {                                                                    \
        const unsigned long long now_ = __builtin_readcyclecounter();       \
        tl_acc_[slot] += now_ - t_prev_;                                    \
        t_prev_ = now_;                                                     \
    } while (0)
#else
#define TAIL_STAMP(slot)
#endif

// PERSISTENT: a block walks the work items (batch, depth segment, tile) item = blockIdx.x, + gridDim.x, ... with the weights loaded once;
// the staging pipeline runs ACROSS items (the last plane of a tile prefetches the first plane of the block's next tile), so the only
// exposed load latency is the block's very first plane.  (The first version launched one block per tile: with D = 4 planes a block lived
// for four passes, reloaded 27 KB of weight fragments per wavefront and started cold - 0.39 ms at stage 4 against 0.24 for the fp32 tail.)
__global__ __launch_bounds__(256, 2) void tail_x3_kernel(const TailArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;
    const int D = a.D, H = a.H, W = a.W, Ho = 2 * H, Wo = 2 * W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW, HWo = (size_t)Ho * Wo;
    const int nitems = a.tiles * a.nseg * a.B;
    int item = blockIdx.x;
    if (item >= nitems) return;

    // every weight fragment of the layer, for the whole kernel
    bf16x8 wf[3][3][3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t) wf[kd][s][t] = a.wp[((kd * 3 + s) * 3 + t) * 64 + lane];

    // this lane's 4 output channels (kb & 1) * 4 + e of column parity pw = kb >> 1; their BatchNorm scale / shift and 1x1x1 weight sit in a
    // 96-byte LDS table and are read where they are used (12 registers that would otherwise be live through the MFMA phase)
    const int pw = kb >> 1, cb = (kb & 1) * 4;
    __shared__ __attribute__((aligned(16))) float s_par[3][8];
    if (tid < 8) {
        s_par[0][tid] = a.scale ? a.scale[tid] : 1.0f;
        s_par[1][tid] = a.shift ? a.shift[tid] : 0.0f;
        s_par[2][tid] = a.prob_w[tid];
    }
    const float pb = a.prob_b ? a.prob_b[0] : 0.0f;

    // B fragment of box row `row` (0 .. TIH): pixel (row, n + dj), octet (kb & 1); + term * TERM_BYTES + buffer
    const unsigned foff = (unsigned)((n + (kb >> 1)) * PB + (kb & 1) * 16);

    // ---- a work item: tile origin (input coordinates), batch index, depth segment ----
    struct Item { int x0, y0, b, d_lo, d_hi, p_first, p_last; };
    auto decode = [&](int it) {
        Item I;
        const int tile = it % a.tiles, rest = it / a.tiles, seg = rest % a.nseg;
        I.b = rest / a.nseg;
        I.x0 = (tile % a.tiles_x) * TIW;
        I.y0 = (tile / a.tiles_x) * TIH;
        I.d_lo = seg * a.seg_planes;
        I.d_hi = min(D, I.d_lo + a.seg_planes);
        I.p_first = max(0, I.d_lo - 1);
        I.p_last = min(D - 1, I.d_hi);
        return I;
    };

    // ---- staging: item (octet, box pixel): the pixel's 8 channel planes are 8 coalesced dword BUFFER loads - one 32-bit lane offset per
    //      staging item and tile, the channel plane, the depth plane and the batch go into the scalar offset, and pixels outside the
    //      volume get the out-of-range offset (the load returns 0: zero padding without a select) ----
    const rsrc_t xin = make_rsrc(a.x, (unsigned)min((size_t)a.B * CIN * DHW * 4, (size_t)0x7FFFFFFF));
    unsigned voff[NI];
    auto staging_offsets = [&](const Item& I) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * 256;
            const int oct = i / NPIX, v = i % NPIX;
            const int gy = I.y0 + v / BW, gx = I.x0 + v % BW;
            voff[it] = (i < 2 * NPIX && gy < H && gx < W) ? (unsigned)(((size_t)(oct * 8) * DHW + (size_t)gy * W + gx) * 4) : OOB;
        }
    };
    float pre[NI][8];
    auto issue = [&](const Item& I, int p) {
        const size_t base = (size_t)I.b * CIN * DHW + (size_t)p * HW;
#pragma unroll
        for (int it = 0; it < NI; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) pre[it][e] = TAIL_ABLATE(a, 1) ? 1.0f : stage_load(xin, voff[it], (unsigned)((base + (size_t)e * DHW) * 4));
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * 256;
            if (i < 2 * NPIX) {
                const int oct = i / NPIX, v = i % NPIX;
                const Split3 sp = split3(pre[it]);
                unsigned char* dst = buf + v * PB + oct * 16;
                *reinterpret_cast<bf16x8*>(dst) = sp.h;
                *reinterpret_cast<bf16x8*>(dst + TERM_BYTES) = sp.m;
                *reinterpret_cast<bf16x8*>(dst + 2 * TERM_BYTES) = sp.l;
            }
        }
    };

    // ---- the skip tensor enters the logit only through sum_c w_c x_c (prob is linear): the loads of plane od go out at the start of the
    //      pass that stages input plane od, are reduced to ONE partial sum per output pixel at its end (4 registers instead of 16 held
    //      for another pass) and join plane od's logits one pass later, when its convolution is complete ----
    const rsrc_t rin = make_rsrc(a.residual ? a.residual : a.x, a.residual ? (unsigned)min((size_t)a.B * COUT * D * HWo * 4, (size_t)0x7FFFFFFF) : 0u);
    unsigned roff[R][2];
    auto residual_offsets = [&](const Item& I) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int oy = 2 * (I.y0 + wave * R + r) + ph, ox = 2 * (I.x0 + n) + pw;
                roff[r][ph] = (oy < Ho && ox < Wo) ? (unsigned)(((size_t)cb * D * HWo + (size_t)oy * Wo + ox) * 4) : OOB;
            }
    };
    float rs[R][2][4];
    auto issue_residual = [&](const Item& I, int od) {
        const size_t base = ((size_t)I.b * COUT * D + od) * HWo;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int e = 0; e < 4; ++e) rs[r][ph][e] = stage_load(rin, roff[r][ph], (unsigned)((base + (size_t)e * D * HWo) * 4));
    };
    // finished output plane od: BatchNorm, ReLU, the 1x1x1 conv over this lane's 4 channels (+ the skip tensor's share), the other 4
    // channels from lane ^ 16, the other column parity from lane ^ 32, 8-byte stores
    auto finish_plane = [&](const Item& I, int od, const f32x4 (&c)[R][2], const float (&rp)[R][2]) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(&s_par[0][cb]), sh = *reinterpret_cast<const f32x4*>(&s_par[1][cb]),
                    wl = *reinterpret_cast<const f32x4*>(&s_par[2][cb]);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                float part = rp[r][ph];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(c[r][ph][e], sc[e], sh[e]);
                    if (a.relu) v = fmaxf(v, 0.0f);
                    part = fmaf(wl[e], v, part);
                }
                part += __shfl_xor(part, 16, 64);
                const float other = __shfl_xor(part, 32, 64);
                const int oy = 2 * (I.y0 + wave * R + r) + ph, ox = 2 * (I.x0 + n);
                if (kb == 0 && oy < Ho && ox < Wo)
                    *reinterpret_cast<float2*>(a.y + ((size_t)I.b * D + od) * HWo + (size_t)oy * Wo + ox) = make_float2(part + pb, other + pb);
            }
    };

    // ---- prologue: the block's first plane, not overlapped ----
    Item I = decode(item);
    staging_offsets(I);
    issue(I, I.p_first);
    commit(lds);
    __syncthreads();
    int cur = 0;
#ifdef X3_TIMELINE
    unsigned long long tl_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev_ = __builtin_readcyclecounter();
#endif

    for (;;) {
        residual_offsets(I);
        f32x4 acc[3][R][2];                                // [output plane p-1 | p | p+1][row][row parity]
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[s][r][0] = acc[s][r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        float rprev[R][2];                                 // sum_c w_c x_c of the output plane that completes in the current pass
#pragma unroll
        for (int r = 0; r < R; ++r) rprev[r][0] = rprev[r][1] = 0.0f;
        const int next_item = item + gridDim.x;
        const bool has_next = next_item < nitems;          // block-uniform
        Item J = I;

        for (int p = I.p_first; p <= I.p_last; ++p) {
            const unsigned char* buf = lds + cur * BUF_BYTES;
            // depth taps whose output plane od = p - 1 + kd lies in [d_lo, d_hi)
            const int kd_lo = max(0, I.d_lo + 1 - p), kd_hi = min(2, I.d_hi - p);
            const bool fin = p - 1 >= I.d_lo;              // output plane p-1 completes with this input plane
            const bool res_now = a.residual != nullptr && p >= I.d_lo && p < I.d_hi;
#ifndef TAIL_INTERLEAVE
#define TAIL_INTERLEAVE 1
#endif
            // the next pass's plane: the next plane of this tile, or the first plane of the block's next tile
            const bool stage_next = p < I.p_last || has_next;
            int np = p + 1;
            if (p == I.p_last && has_next) {
                J = decode(next_item);
                staging_offsets(J);
                np = J.p_first;
            }
            const Item& S = (p < I.p_last) ? I : J;
#if TAIL_INTERLEAVE
            // The pass's 32 dword loads per lane (16 of the skip tensor, 16 of the next plane) go out BETWEEN the MFMA groups, a few per group:
            // as one burst at the top of the pass they took half of it (tools/x3_timeline.py: ~100 clocks per load instruction with the vector
            // memory queue full, the matrix pipe idle meanwhile); the weights sit in registers, so nothing in the MFMA phase waits for vmcnt.
            const size_t rbase = ((size_t)I.b * COUT * D + p) * HWo, sbase = (size_t)S.b * CIN * DHW + (size_t)np * HW;
            auto load_slot = [&](int k) {                  // k = 0 .. 31, compile-time at every call site
                if (k < 16) {
                    const int r = k >> 3, ph = (k >> 2) & 1, e = k & 3;
                    if (res_now && r < R) rs[r][ph][e] = stage_load(rin, roff[r][ph], (unsigned)((rbase + (size_t)e * D * HWo) * 4));
                } else {
                    const int it = (k - 16) >> 3, e = (k - 16) & 7;
                    if (stage_next && it < NI) pre[it][e] = stage_load(xin, voff[it], (unsigned)((sbase + (size_t)e * DHW) * 4));
                }
            };
            static_assert(R == 2 && NI == 2, "load_slot's schedule is written for two rows per wavefront and two staging items per thread");
#else
            if (res_now) issue_residual(I, p);
            if (stage_next) issue(S, np);
#endif
            TAIL_STAMP(0);                                 // loads issued (TAIL_INTERLEAVE: nothing yet)
            if (!TAIL_ABLATE(a, 4)) {
                // row fragments roll: F(row) serves tile P and step 0 of tile Q of its own row, and step 1 of tile Q of the row above
                bf16x8 f0[3], f1[3];
                const unsigned char* fp = buf + (wave * R) * (BW * PB) + foff;
#pragma unroll
                for (int t = 0; t < 3; ++t) f0[t] = *reinterpret_cast<const bf16x8*>(fp + t * TERM_BYTES);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bf16x8(&fa)[3] = (r & 1) ? f1 : f0;
                    bf16x8(&fb)[3] = (r & 1) ? f0 : f1;
#pragma unroll
                    for (int t = 0; t < 3; ++t) fb[t] = *reinterpret_cast<const bf16x8*>(fp + (r + 1) * (BW * PB) + t * TERM_BYTES);
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
#if TAIL_INTERLEAVE
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            if ((r * 3 + kd) * 6 + k < 32) load_slot((r * 3 + kd) * 6 + k);
                        __builtin_amdgcn_sched_barrier(0);
#endif
                        if (kd < kd_lo || kd > kd_hi) continue;             // block-uniform
                        acc[kd][r][0] = mfma6(wf[kd][0][0], wf[kd][0][1], wf[kd][0][2], fa[0], fa[1], fa[2], acc[kd][r][0]);
                        f32x4 q = acc[kd][r][1];
                        q = mfma6(wf[kd][1][0], wf[kd][1][1], wf[kd][1][2], fa[0], fa[1], fa[2], q);
                        q = mfma6(wf[kd][2][0], wf[kd][2][1], wf[kd][2][2], fb[0], fb[1], fb[2], q);
                        acc[kd][r][1] = q;
                    }
                }
            }
#if TAIL_INTERLEAVE
            else {
#pragma unroll
                for (int k = 0; k < 32; ++k) load_slot(k);
            }
#endif
            TAIL_STAMP(1);                                 // MFMA phase
            if (fin && !TAIL_ABLATE(a, 8)) finish_plane(I, p - 1, acc[0], rprev);
            TAIL_STAMP(2);                                 // epilogue + stores of the finished plane
#ifdef X3_TIMELINE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TAIL_STAMP(3);                                 // waiting for the skip tensor's and the next plane's loads
#endif
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    acc[0][r][ph] = acc[1][r][ph];
                    acc[1][r][ph] = acc[2][r][ph];
                    acc[2][r][ph] = f32x4{0.f, 0.f, 0.f, 0.f};
                    float t = 0.0f;
                    if (res_now) {
                        const f32x4 wl = *reinterpret_cast<const f32x4*>(&s_par[2][cb]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) t = fmaf(wl[e], rs[r][ph][e], t);
                    }
                    rprev[r][ph] = t;
                }
            TAIL_STAMP(4);                                 // skip tensor reduced
            if (stage_next) {
                commit(lds + (cur ^ 1) * BUF_BYTES);
                TAIL_STAMP(5);                             // split + LDS stores
                __syncthreads();                           // this plane's fragment reads are done, the next plane is visible
                cur ^= 1;
                TAIL_STAMP(6);                             // barrier
            }
        }
        if (I.p_last == D - 1 && I.d_hi == D && !TAIL_ABLATE(a, 8)) finish_plane(I, D - 1, acc[0], rprev);
        if (!has_next) break;
        I = J;
        item = next_item;
    }
#ifdef X3_TIMELINE
    if (tid == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(&tail_phase_ticks[i], tl_acc_[i]);
#endif
}

}  // namespace

extern "C" int64_t mvs_tail_x3_packed_bytes(void) { return (int64_t)NW * 64 * 16; }

extern "C" int mvs_tail_x3_pack_weights(const float* w, void* wpacked, mvs_stream_t stream) {
    MVS_REQUIRE(w && wpacked, "mvs_tail_x3_pack_weights: null pointer");
    hipLaunchKernelGGL(tail_x3_pack_kernel, dim3(mvs::ceil_div(NW * 64, 256)), dim3(256), 0, MVS_STREAM(stream), w, static_cast<bf16x8*>(wpacked));
    return mvs::finish_launch("mvs_tail_x3_pack_weights");
}

extern "C" int mvs_tail_x3_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual,
                               const float* prob_w, const float* prob_b, float* logits, int B, int D, int H, int W, int relu,
                               mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && prob_w && logits, "mvs_tail_x3_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W >= 1, "mvs_tail_x3_fwd: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
    MVS_REQUIRE(!scale || shift, "mvs_tail_x3_fwd: scale without shift");
    MVS_REQUIRE((int64_t)B * 32 * D * H * W * 4 < ((int64_t)1 << 31), "mvs_tail_x3_fwd: the skip volume exceeds the 2 GiB buffer window");
    TailArgs a;
    a.x = x; a.wp = static_cast<const bf16x8*>(wpacked); a.scale = scale; a.shift = shift; a.residual = residual; a.prob_w = prob_w;
    a.prob_b = prob_b; a.y = logits; a.B = B; a.D = D; a.H = H; a.W = W; a.relu = relu;
    a.tiles_x = mvs::ceil_div(W, TIW);
    a.tiles = a.tiles_x * mvs::ceil_div(H, TIH);
    const int64_t blocks = (int64_t)a.tiles * B;
    MVS_REQUIRE(blocks * D < ((int64_t)1 << 31), "mvs_tail_x3_fwd: too many tiles");
    // depth segments (each re-stages one halo plane per cut): only while there are fewer work items than ~6 per CU
    int nseg = 1;
    while (nseg * 2 <= D / 2 && blocks * nseg < 1536) nseg *= 2;
    a.seg_planes = mvs::ceil_div(D, nseg);
    nseg = mvs::ceil_div(D, a.seg_planes);
    a.nseg = nseg;
#ifdef X3_ABLATE            // experiment builds only (make exp EXPFLAGS=-DX3_ABLATE): bit 0 constant inputs, bit 2 no MFMAs, bit 3 no stores
    {
        const char* e = getenv("MVS_X3_ABLATE");
        a.ablate = e ? atoi(e) : 0;
    }
#else
    a.ablate = 0;
#endif
    const int64_t nitems = blocks * nseg;
    const int resident = 2 * mvs::device_cus();            // two blocks per CU (launch bounds), persistent over the work items
    hipLaunchKernelGGL(tail_x3_kernel, dim3((unsigned)(nitems < resident ? nitems : resident)), dim3(256), 0, MVS_STREAM(stream), a);
    return mvs::finish_launch("mvs_tail_x3_fwd");
}

#ifdef X3_TIMELINE
extern "C" int mvs_tail_timeline(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(tail_phase_ticks), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(tail_phase_ticks), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif
