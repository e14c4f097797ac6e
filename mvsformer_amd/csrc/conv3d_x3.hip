// 3x3x3 convolutions of the eval regularizer (models/module.py:83-123 Conv3d = conv -> BatchNorm -> ReLU, used by CostRegNet /
// CostRegNet3D, module.py:469-505,550-594) on the BF16 matrix cores in THREE-TERM SPLIT form - fp32 in, fp32 out, fp32-equivalent.
//
// Why.  v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (32 MAC/clk/SIMD) and shares the issue pipe with every other vector
// instruction (profiles/r03_ubench.txt: next to back-to-back fp32 MFMAs a second wavefront gets 2.4 vector instructions per 32
// clocks), so the fp32 convolutions sit at 0.3-0.6 of a peak that is itself 16x below the bf16 matrix rate.  Every fp32 value is
// EXACTLY h + m + l with h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) (3 x 8 significand bits = fp32's 24), so
//     x*w = xh*wh + (xh*wm + xm*wh) + (xh*wl + xl*wh + xm*wm)  + terms <= 2^-24 |x*w|
// and six v_mfma_f32_16x16x32_bf16 (512 MAC/clk/SIMD, fp32 accumulation, every partial product exact) replace eight fp32 MFMAs of
// the same K at 6/16 of their matrix time.  The dropped terms are at the level of ONE fp32 rounding of the product; measured
// against fp64 the result is as close as (closer than) an fp32 fma chain (tests/test_hip_x3.py, tools/sim_x3.py).  This is not a
// reduced-precision path: the 2-term form (error 2^-16) is NOT used anywhere.
//
// Structure (stride (1,1,1) and (1,2,2) layers: conv1 ... conv6 of CostRegNet3D, conv2 / conv4 / conv6 of CostRegNet):
//   * a block owns a (4*NT) x 16 column of output pixels through ALL depth planes and 16*MTB output channels, and sweeps the INPUT planes:
//     input plane p feeds output planes p+1, p, p-1 (kd = 0, 1, 2), whose accumulators live in registers (3 sets), so every input
//     plane is staged ONCE (no depth halo) and the depth taps that only see padding are simply not issued;
//   * staging: fp32 NCDHW -> (h, m, l) bf16 CHANNEL-LAST in LDS, the tile's input box (18 x 18 pixels at stride 1, (8NT+1) x 33 with the
//     even columns first at stride 2) x CK = 8 | 16 channels per pass (27-54 KB): a thread loads 8
//     channels of one pixel (8 coalesced dword loads), splits them (5.5 vector ops per value) and stores three 16-byte rows, so the
//     MFMA B operand (8 consecutive channels of one tap of one pixel) is ONE ds_read_b128 per term;
//   * K = (spatial tap, channel octet) in blocks of 8 channels, 4 blocks per MFMA: 9 taps x CK/8 octets = 18 (9) blocks = 5 (3) steps per
//     depth tap (the last step is partly zero weights); A operands (weights, pre-split and pre-packed per lane) come straight from
//     L1/L2, prefetched one step ahead into ping-pong registers, and are reused over the wavefront's NT pixel rows; M = 16 pixels of a
//     row (so that a lane ends up with 4 consecutive pixels of one channel: 16-byte stores), N = 16 output channels;
//   * volumes with few tiles are cut into depth segments (one halo plane re-staged per cut) so that every CU gets several blocks.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "conv_common.h"
#include "split3.h"

#ifndef X3_STAGE_AUX
#define X3_STAGE_AUX 0      // cache policy of the staging / skip-tensor loads (aux of raw_buffer_load; 2 = nt measured 20-30 % SLOWER: the eight
#endif                      // dword loads of a staging item share cache lines with its neighbours' and need the L1)

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;

__device__ __forceinline__ float stage_load(rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, X3_STAGE_AUX));
}
using mvsx3::Split3;
using mvsx3::split3;

// Compile-time geometry of one kernel instance.  CK = input channels per staging pass (8 | 16), SHW = H/W stride (1 | 2), NT = pixel rows
// per wavefront (the block's tile is 4*NT rows x 16 columns of OUTPUT pixels), MTB = 16-channel output tiles per block.
template <int CK_, int SHW_, int NT_, int MTB_>
struct X3Cfg {
    static constexpr int CK = CK_, SHW = SHW_, NT = NT_, MTB = MTB_;
    static constexpr int KQ = CK / 8;                          // channel octets per tap
    static constexpr int NKB = 9 * KQ;                         // K blocks (tap, octet) per depth tap
    static constexpr int STEPS = (NKB + 3) / 4;                // MFMAs of K = 32 per depth tap: 5 for CK = 16, 3 for CK = 8
    static_assert(STEPS % 2 == 1, "the ping-pong parity bookkeeping of the kernel assumes an odd step count per depth tap");
    static constexpr int TH = 4 * NT, TW = 16;
    static constexpr int BH = SHW * TH + (3 - SHW), BWC = SHW * TW + (3 - SHW);   // staged box (halo 1): (TH+2) x 18 | (2TH+1) x 33
    static constexpr int EV = (BWC + 1) / 2;                   // SHW = 2: even columns first (EV of them), then the odd ones
    static constexpr int PB = CK * 2;                          // bytes per pixel per term
    static constexpr int TERM_BYTES = BH * BWC * PB;
    static constexpr int LDS_BYTES = 3 * TERM_BYTES;
    static constexpr int FRAGS_PER_KD = STEPS * 3 * 64;        // bf16x8 units of one (cout tile, chunk, kd)
#ifndef X3_MB_MTB2
#define X3_MB_MTB2 2
#endif
#ifndef X3_MB_CK8
#define X3_MB_CK8 3      // conv1's instance (8 -> 16, stride 2) is HBM-bound (340 MB at stage 4): a third block per CU (164 registers, 3 x 52 KB of
#endif                   // LDS) keeps 50 % more loads in flight - measured 0.140 -> 0.130 ms at stage 4, 0.088 -> 0.076 at stage 3 (round 4)
    static constexpr int MIN_BLOCKS = (MTB == 1 && CK == 16) ? 3 : ((SHW == 1 && MTB == 2) ? X3_MB_MTB2 : (CK == 8 ? X3_MB_CK8 : 2));   // blocks per CU the register budget is set for
    // B fragments of the next (step, row) item read under the current item's MFMAs (+2...6 % measured); not where the 168-register budget
    // of three blocks per CU has no room for the second fragment set (<16,1,4,1>: 16 spills, -5 %)
    static constexpr bool BPIPE = MIN_BLOCKS < 3;
    __host__ __device__ static constexpr int col_index(int c) { return SHW == 1 ? c : (c & 1) * EV + (c >> 1); }
};

// packed[(((ct*NCH + chunk)*3 + kd)*STEPS + step)*3 + term][lane][8]:  A[m = ct*16 + (lane & 15)][K block q = 4*step + (lane >> 4)],
// q -> (tap9 = q / KQ -> (kh, kw), octet = q % KQ), channel = chunk*CK + octet*8 + e; zero for q >= 9*KQ, for rows beyond Cout and for
// the one extra unit at the end (the kernel prefetches one step ahead).  w = [Cout][Cin][27] fp32.
__global__ void x3_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int CK, bf16x8* __restrict__ out, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int KQ = CK / 8, NKB = 9 * KQ, STEPS = (NKB + 3) / 4, NCH = Cin / CK;
    const int lane = idx & 63, term = (idx >> 6) % 3, step = (idx / 192) % STEPS, kd = (idx / (192 * STEPS)) % 3;
    const int chunk = (idx / (192 * STEPS * 3)) % NCH, ct = idx / (192 * STEPS * 3 * NCH);
    const int m = ct * 16 + (lane & 15), q = 4 * step + (lane >> 4);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = 0.0f;
        if (q < NKB && m < Cout) {
            const int tap9 = q / KQ, c = chunk * CK + (q % KQ) * 8 + e;
            f = w[((size_t)m * Cin + c) * 27 + kd * 9 + tap9];
        }
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

struct X3Args {
    const float* x;
    const bf16x8* wp;
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int Cin, Cout, D, H, W, Ho, Wo, relu, tiles_x;
    int seg_planes, nseg;           // depth segments: block z = batch * nseg + segment
#ifdef X3_PRESPLIT                  // experiment build: activations kept in memory ALREADY SPLIT, channel-last:
    const void* xpre;               //   [B][D][H][W][h|m|l][16] bf16 (96 bytes per voxel) read by the 16 -> 16 stride-1 instance through LDS-DMA
    void* ypre;                     //   the same form written by the 8 -> 16 stride-2 instance's epilogue
    int fp32_out;                   //   0: the producer writes ONLY the pre-split form
#endif
};

#ifdef X3_PRESPLIT
typedef __attribute__((address_space(3))) void* x3_lds_ptr_t;
// LDS-DMA: 16 bytes per lane from src + voff (+ soff) to lds_dst + lane * 16.  In a __device__ helper on purpose (tools/probe/gather_probe.hip:
// with the builtin directly in a __global__ template body hipcc drops the kernel's host stub).
__device__ __forceinline__ void x3_dma16(rsrc_t src, unsigned char* lds_dst, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (x3_lds_ptr_t)lds_dst, 16, voff_bytes, soff_bytes, 0, 0);
}
#endif

template <int V>
using ic = std::integral_constant<int, V>;

#ifdef X3_TIMELINE
// experiment build only (make exp EXPFLAGS=-DX3_TIMELINE): wavefront 0 of every block adds the s_memtime ticks of each phase of a pass
__device__ unsigned long long x3_phase_ticks[8];
#define X3_STAMP(slot)                                                              \
    do {                                                                            \
        const unsigned long long now_ = __builtin_readcyclecounter();               \
        tl_acc_[slot] += now_ - t_prev_;                                            \
        t_prev_ = now_;                                                             \
    } while (0)
#else
#define X3_STAMP(slot)
#endif

#ifdef X3_PRESPLIT
template <class Cfg, bool PRE_IN = false, bool PRE_OUT = false>
#else
template <class Cfg>
#endif
__global__ __launch_bounds__(256, Cfg::MIN_BLOCKS) void x3_conv_kernel(const X3Args a) {
#ifndef X3_PRESPLIT
    constexpr bool PRE_IN = false, PRE_OUT = false;
#endif
#ifndef X3_XCD_ORDER
#define X3_XCD_ORDER 1
#endif
    // Block order (round 4): every XCD works on one contiguous slab of the logical order (output-channel block fastest, then depth segment,
    // then the tile in raster order, then the batch), so that the blocks sharing input - the channel blocks and depth segments of a tile,
    // the tiles next to it - find it in ONE L2 (counters before: 2-9x the layer's input fetched, profiles/traffic_by_kernel.json)
#if X3_XCD_ORDER
    unsigned nid_ = mvsconv::xcd_linear_block_id();
    const int ctb = (int)(nid_ % gridDim.y);
    nid_ /= gridDim.y;
    const int seg = (int)(nid_ % (unsigned)a.nseg);
    nid_ /= (unsigned)a.nseg;
    const int tile = (int)(nid_ % gridDim.x), b = (int)(nid_ / gridDim.x);
#else
    const int seg = blockIdx.z % a.nseg;
    const int tile = blockIdx.x, ctb = blockIdx.y, b = blockIdx.z / a.nseg;
#endif
    constexpr int CK = Cfg::CK, SHW = Cfg::SHW, NT = Cfg::NT, MTB = Cfg::MTB, KQ = Cfg::KQ, STEPS = Cfg::STEPS, BWC = Cfg::BWC, PB = Cfg::PB,
                  TERM_BYTES = Cfg::TERM_BYTES, NPIX = Cfg::BH * Cfg::BWC;
    __shared__ __attribute__((aligned(16))) unsigned char lds[Cfg::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;
    const int x0 = (tile % a.tiles_x) * Cfg::TW, y0 = (tile / a.tiles_x) * Cfg::TH;       // output coordinates
    const int Cin = a.Cin, Cout = a.Cout, D = a.D, H = a.H, W = a.W;
    const int NCH = Cin / CK;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW, HWo = (size_t)a.Ho * a.Wo;
    const float* xb = a.x + (size_t)b * Cin * DHW;

    // B-operand byte offsets of this lane's K block in each step (without the row / term parts)
    unsigned boff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int q = min(4 * s + kb, Cfg::NKB - 1), tap9 = q / KQ, kh = tap9 / 3, kw = tap9 % 3;
        boff[s] = (unsigned)((kh * BWC + Cfg::col_index(SHW * n + kw)) * PB + (q % KQ) * 16);
    }

    f32x4 acc[3][MTB][NT];                                 // [output plane p-1 | p | p+1][M tile][pixel row]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int mt = 0; mt < MTB; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[s][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // D[m = pixel][n = output channel] (the activations are the MFMA's A operand): a lane holds 4 consecutive pixels (4*kb .. 4*kb+3) of
    // output channel n of each (M tile, row) -> one 16-byte store, and one scale / shift pair per lane
    const bool vec_ok = (a.Wo & 3) == 0;
    float ep_sc[MTB], ep_sh[MTB];                          // read once: inside store_plane the loads would sit behind a full L2 round trip per plane
#pragma unroll
    for (int mt = 0; mt < MTB; ++mt) {
        const int co = min((ctb * MTB + mt) * 16 + n, Cout - 1);
        ep_sc[mt] = a.scale ? a.scale[co] : 1.0f;
        ep_sh[mt] = a.shift ? a.shift[co] : 0.0f;
    }
    auto store_plane = [&](int od, const f32x4 (&c)[MTB][NT]) {
#pragma unroll
        for (int mt = 0; mt < MTB; ++mt) {
            const int co = (ctb * MTB + mt) * 16 + n;
            if (co >= Cout) continue;
            const float sc = ep_sc[mt], sh = ep_sh[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int gy = y0 + wave * NT + nt, gx = x0 + kb * 4;
                if (gy >= a.Ho || gx >= a.Wo) continue;
                const size_t o = ((size_t)(b * Cout + co) * D + od) * HWo + (size_t)gy * a.Wo + gx;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = a.scale ? fmaf(c[mt][nt][r], sc, sh) : c[mt][nt][r] + sh;
                    if (a.relu) v[r] = fmaxf(v[r], 0.0f);
                }
#ifdef X3_PRESPLIT
                if (PRE_OUT) {                             // [b][od][gy][gx][term][16 channels] bf16: lanes n = 0..15 write 32 contiguous bytes per (pixel, term)
                    __bf16* pp = reinterpret_cast<__bf16*>(a.ypre) + ((((size_t)b * D + od) * a.Ho + gy) * a.Wo + gx) * 48 + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        __bf16 h, m, l;
                        mvsx3::split3(v[r], h, m, l);
                        pp[r * 48] = h;
                        pp[r * 48 + 16] = m;
                        pp[r * 48 + 32] = l;
                    }
                    if (!a.fp32_out) continue;
                }
#endif
                if (vec_ok) {
                    if (a.residual) {
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(a.residual + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] + rs[r];
                    }
                    *reinterpret_cast<f32x4*>(a.y + o) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gx + r < a.Wo) a.y[o + r] = a.residual ? v[r] + a.residual[o + r] : v[r];
                }
            }
        }
    };

    // ping-pong weight fragments [buffer][M tile][h | m | l].  A ring of three (two steps of weights in flight) was measured in round 4
    // (profiles/r04_bench_x3_wring.txt): 4-29 spilled registers in four of the five instances and 2-15 % slower everywhere - the MFMA
    // phase is not waiting for weights (tools/x3_timeline.py: its length follows the LDS fragment reads and the other wavefronts' MFMAs)
    constexpr int WR = 2, WAHEAD = 1;
    bf16x8 wbuf[WR][MTB][3];
    auto load_w = [&](const bf16x8* wk, bf16x8 (&aw)[MTB][3]) {
#pragma unroll
        for (int mt = 0; mt < MTB; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t) aw[mt][t] = wk[(size_t)mt * NCH * 3 * Cfg::FRAGS_PER_KD + t * 64];
    };
    // The STEPS steps of depth tap KD, starting with the weights of its first step already in wbuf[P]; every step prefetches the next
    // step's weights (contiguous in memory, running on into the next depth tap) into the other buffer (measured: 10-20 % over loading
    // each step's weights right before its MFMAs).  KD, P and s are compile-time, so every register index is static.
    // staging: item i = tid + it * 256 -> (channel octet, box pixel); the 8 channel planes of the pixel are 8 coalesced dword loads
    // staging through BUFFER loads (round 4): one 32-bit lane offset per item for the whole kernel, channel chunk / depth plane in the
    // scalar offset, out-of-volume pixels read 0 through the descriptor's range check - no 64-bit address arithmetic, no select
    constexpr int NI = (KQ * NPIX + 255) / 256;
    const rsrc_t xin = make_rsrc(xb, (unsigned)((size_t)Cin * DHW * 4));
    unsigned voff[NI];
    int ldst[NI];                                          // LDS byte offset of the item's h row (-1: no item)
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int i = tid + it * 256;
        const int oct = i / NPIX, v = i % NPIX;
        const int gy = y0 * SHW - 1 + v / BWC, gx = x0 * SHW - 1 + v % BWC;
        const bool item = i < KQ * NPIX;
        voff[it] = (item && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)(((size_t)(oct * 8) * DHW + (size_t)gy * W + gx) * 4) : OOB;
        ldst[it] = item ? ((v / BWC) * BWC + Cfg::col_index(v % BWC)) * PB + oct * 16 : -1;
    }
    float pre[NI][8];
    auto issue = [&](int it, int pp, int cc) {
        const size_t base = (size_t)(cc * CK) * DHW + (size_t)pp * HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[it][e] = stage_load(xin, voff[it], (unsigned)((base + (size_t)e * DHW) * 4));
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NI; ++it)
            if (ldst[it] >= 0) {
                const Split3 sp = split3(pre[it]);
                unsigned char* dst = lds + ldst[it];
                *reinterpret_cast<bf16x8*>(dst) = sp.h;
                *reinterpret_cast<bf16x8*>(dst + TERM_BYTES) = sp.m;
                *reinterpret_cast<bf16x8*>(dst + 2 * TERM_BYTES) = sp.l;
            }
    };
    auto load_b = [&](int s, int nt, bf16x8 (&bf)[3]) {
        const unsigned char* bp = lds + (SHW * (wave * NT + nt)) * (BWC * PB) + boff[s];
#pragma unroll
        for (int t = 0; t < 3; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(bp + t * TERM_BYTES);
    };
    auto kd_steps = [&](auto kd_tag, auto p_tag, const bf16x8* wk, bool more) {
        constexpr int KD = decltype(kd_tag)::value, P = decltype(p_tag)::value, SET = 2 - KD;
        bf16x8 bf[2][3];                                   // B fragments of (step, row) item j and j + 1: the next item's reads run under this item's MFMAs
        if (Cfg::BPIPE) load_b(0, 0, bf[0]);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int cur = (P + s) % WR;
            if (s + WAHEAD < STEPS || more) load_w(wk + (size_t)(s + WAHEAD) * 192, wbuf[(P + s + WAHEAD) % WR]);   // no load left in flight at the end of the pass
            // one step's (BPIPE: one item's) operands + the prefetches in flight at a time: left alone, the scheduler hoists every step's loads
            if (!Cfg::BPIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int j = s * NT + nt;
                if (Cfg::BPIPE) {
                    if (j + 1 < STEPS * NT) load_b((j + 1) / NT, (j + 1) % NT, bf[(j + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    load_b(s, nt, bf[0]);
                }
                const bf16x8 xh = bf[Cfg::BPIPE ? (j & 1) : 0][0], xm = bf[Cfg::BPIPE ? (j & 1) : 0][1], xl = bf[Cfg::BPIPE ? (j & 1) : 0][2];
#pragma unroll
                for (int mt = 0; mt < MTB; ++mt) {
                    f32x4 c = acc[SET][mt][nt];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, wbuf[cur][mt][1], c, 0, 0, 0);     // smallest products first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wbuf[cur][mt][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wbuf[cur][mt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wbuf[cur][mt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, wbuf[cur][mt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wbuf[cur][mt][0], c, 0, 0, 0);
                    acc[SET][mt][nt] = c;
                }
                if (Cfg::BPIPE) __builtin_amdgcn_sched_barrier(0);
            }
            if (!Cfg::BPIPE) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // this block's depth segment: output planes [d_lo, d_hi); input planes d_lo-1 .. d_hi (clipped)
    const int d_lo = seg * a.seg_planes, d_hi = min(D, d_lo + a.seg_planes);
    if (d_lo >= D) return;                                 // an empty segment owns no output plane (block-uniform; the launchers never create one)
    const int p_first = max(0, d_lo - 1), p_last = min(D - 1, d_hi);
    const int NP = (p_last - p_first + 1) * NCH;           // passes: (input plane, channel chunk)
#ifdef X3_TIMELINE
    unsigned long long tl_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int pass = 0; pass < NP; ++pass) {
        const int p = p_first + pass / NCH, chunk = pass % NCH;
        // depth taps of input plane p whose output plane od = p + 1 - kd lies in [d_lo, d_hi): a contiguous, block-uniform range
        const int kd_lo = max(0, p + 2 - d_hi), kd_hi = min(2, p + 1 - d_lo);
        const bf16x8* wk = a.wp + ((size_t)((ctb * MTB) * NCH + chunk) * 3 + kd_lo) * Cfg::FRAGS_PER_KD + lane;
#pragma unroll
        for (int k = 0; k < WAHEAD; ++k) load_w(wk + (size_t)k * 192, wbuf[k]);      // the first steps' weights travel while the plane is staged
        // ---- stage plane p, channels [CK*chunk, CK*chunk + CK): fp32 -> (h, m, l) bf16 channel-last: all of a thread's loads first (NI
        //      items x 8 channel planes in flight), then the split and the LDS stores.  Measured and dropped: the NEXT pass's loads in
        //      FRONT of this pass's MFMA phase (-10...25 %: the weight fragments are global loads too, vmcnt retires in order, so the
        //      first weight wait drains the whole prefetch), and INSIDE it, a few per step behind each step's weight prefetch, held in
        //      registers until the pass ends (+2 % for the 32/64-channel stride-1 layers, -12...-40 % elsewhere: 24-40 more live
        //      registers through the MFMA phase) ----
#ifdef X3_TIMELINE
        unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
#ifdef X3_PRESPLIT
        if (PRE_IN) {
            // the box of plane p straight into LDS: item i = 2*pixel + octet -> 16 bytes at lds + term*TERM_BYTES + i*16, three DMA instructions
            // (h, m, l) per 256 items; no staging registers, no split, no LDS stores.  (Not issued a pass ahead here: one buffer.)
            __syncthreads();                               // the previous pass's fragment reads are done
            const rsrc_t pin = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.xpre)) +
                                                                     (size_t)b * DHW * 96, 0, (unsigned)(DHW * 96), 0x00020000);
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int i = tid + it * 256;
                const int v = i >> 1, oct = i & 1;
                const int gy = y0 - 1 + v / BWC, gx = x0 - 1 + v % BWC;
                const unsigned vo = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)(((size_t)gy * W + gx) * 96 + oct * 16) : OOB;
                if (i < KQ * NPIX) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        x3_dma16(pin, lds + t * TERM_BYTES + (it * 256 + wave * 64) * 16, vo, (unsigned)((size_t)p * HW * 96 + t * 32));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#endif
#pragma unroll
        for (int it = 0; it < NI; ++it) issue(it, p, chunk);
        X3_STAMP(0);                                       // loads issued
        __syncthreads();                                   // the previous pass's fragment reads are done
        X3_STAMP(1);
#ifdef X3_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X3_STAMP(2);                                       // staging loads (and the first weights) have arrived
#endif
        commit();
#ifdef X3_PRESPLIT
        }
#endif
        X3_STAMP(3);
        __syncthreads();
        X3_STAMP(4);
#ifndef X3_SETPRIO
#define X3_SETPRIO 1
#endif
        if (X3_SETPRIO) __builtin_amdgcn_s_setprio(1);     // co-resident blocks are in other phases: the MFMA phase wins the issue arbitration
        // ---- the depth taps of this plane; the buffer parity flips after each one (STEPS is odd) ----
        // ring position of each depth tap's first step: STEPS % WR further on per executed tap (compile-time per branch)
        constexpr int ADV = STEPS % WR, P1 = ADV, P2 = (2 * ADV) % WR;
        if (kd_lo == 0) {
            kd_steps(ic<0>{}, ic<0>{}, wk, kd_hi >= 1);
            if (kd_hi >= 1) kd_steps(ic<1>{}, ic<P1>{}, wk + Cfg::FRAGS_PER_KD, kd_hi >= 2);
            if (kd_hi >= 2) kd_steps(ic<2>{}, ic<P2>{}, wk + 2 * Cfg::FRAGS_PER_KD, false);
        } else if (kd_lo == 1) {
            kd_steps(ic<1>{}, ic<0>{}, wk, kd_hi >= 2);
            if (kd_hi >= 2) kd_steps(ic<2>{}, ic<P1>{}, wk + Cfg::FRAGS_PER_KD, false);
        } else {
            kd_steps(ic<2>{}, ic<0>{}, wk, false);
        }
        if (X3_SETPRIO) __builtin_amdgcn_s_setprio(0);
        X3_STAMP(5);                                       // MFMA phase
        if (chunk == NCH - 1) {
            // output plane p-1 has seen its three input planes
            if (p - 1 >= d_lo) store_plane(p - 1, acc[0]);
            X3_STAMP(6);
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][mt][nt] = acc[1][mt][nt];
                    acc[1][mt][nt] = acc[2][mt][nt];
                    acc[2][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
    }
    if (p_last == D - 1 && d_hi == D) store_plane(D - 1, acc[0]);
#ifdef X3_TIMELINE
    if (tid == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(&x3_phase_ticks[i], tl_acc_[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4, built / measured / removed: `x3_conv_db_kernel`, the same convolution with two activation buffers and the
// NEXT pass's staging loads issued behind step 0's weight prefetch, split + stored after the last step, one barrier per pass.  It passed
// tests/test_hip_x3.py and was SLOWER (stage 4: conv2 0.145 vs 0.131 ms, conv4 0.155 vs 0.132, conv6 0.157 vs 0.136).  Why, from its ISA:
// vmcnt retires IN ORDER and the weight fragments are global loads too, so the wait for step 2's weights (issued after the staging
// loads) drains the staging loads - they get two steps (~800 clk) of cover, not the pass; and two 31 KB buffers cost the NT = 4
// instance its third block per CU.  Long-latency prefetch and a weight stream cannot share one wavefront's vmcnt queue.
// ---------------------------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------------------
// Transposed convolution, stride (1,2,2), kernel 3, padding 1, output_padding (0,1,1) (CostRegNet3D's conv7 / conv9 / conv11,
// models/module.py:562-575), same split form.  out[od, oh, ow] gathers in[od + 1 - kd, (oh + 1 - kh) / 2, (ow + 1 - kw) / 2] where
// divisible, so an output pixel of row parity ph and column parity pw sees only the taps kh in KH(ph), kw in KW(pw):
//     parity 0: k = 1 (input offset 0);   parity 1: k = 0 (input offset +1), k = 2 (input offset 0)
// i.e. 1, 2, 2 and 4 spatial taps for the four parity classes - 9 in total, no wasted work.  A block owns 8 x 16 INPUT pixels
// (16 x 32 output pixels) through all depth planes and 16 output channels, sweeps the input planes like the convolution above
// (input plane p feeds output planes p-1, p, p+1 for kd = 0, 1, 2), and a wavefront owns two input rows: for each class its N tiles
// are (row, 16 output pixels of that class), so the weight fragments of a class step are reused over two tiles.  K blocks of a
// class = (tap, channel octet): 2 / 4 / 4 / 8 blocks -> 1 / 1 / 1 / 2 steps per depth tap.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace dcv {
constexpr int TIH = 8, TIW = 16, BH = TIH + 1, BW = TIW + 1, NPIX = BH * BW, PB = 32, TERM_BYTES = NPIX * PB, LDS_BYTES = 3 * TERM_BYTES;
constexpr int CSTEPS = 5;                                  // class steps per depth tap
constexpr int FRAGS_PER_KD = CSTEPS * 3 * 64;
// class step -> (row parity, column parity, step inside the class)
__host__ __device__ constexpr int cs_ph(int cs) { return cs >= 2 ? 1 : 0; }
__host__ __device__ constexpr int cs_pw(int cs) { return (cs == 1 || cs >= 3) ? 1 : 0; }
__host__ __device__ constexpr int cs_step(int cs) { return cs == 4 ? 1 : 0; }
__host__ __device__ constexpr int cs_class(int cs) { return cs_ph(cs) * 2 + cs_pw(cs); }
// K block q of a class -> tap (kh, kw, input row offset, input column offset) and channel octet; false beyond the class's taps
__host__ __device__ inline bool tap_of(int ph, int pw, int q, int* kh, int* kw, int* di, int* dj, int* oct) {
    const int nkh = ph ? 2 : 1, nkw = pw ? 2 : 1, t = q >> 1;
    *oct = q & 1;
    if (t >= nkh * nkw) return false;
    const int th = t / nkw, tw = t % nkw;
    *kh = ph ? (th == 0 ? 0 : 2) : 1;
    *di = ph ? (th == 0 ? 1 : 0) : 0;
    *kw = pw ? (tw == 0 ? 0 : 2) : 1;
    *dj = pw ? (tw == 0 ? 1 : 0) : 0;
    return true;
}
}  // namespace dcv

// packed[(((ct*NCH + chunk)*3 + kd)*CSTEPS + cs)*3 + term][lane][8]: B[n = ct*16 + (lane & 15)][K block q = 4*step(cs) + (lane >> 4)] of
// class(cs); w = ConvTranspose3d weight [Cin][Cout][27]; one spare zero unit at the end (running prefetch)
__global__ void x3_deconv_pack_kernel(const float* __restrict__ w, int Cin, int Cout, bf16x8* __restrict__ out, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int NCH = Cin / 16;
    const int lane = idx & 63, term = (idx >> 6) % 3, cs = (idx / 192) % dcv::CSTEPS, kd = (idx / (192 * dcv::CSTEPS)) % 3;
    const int chunk = (idx / (192 * dcv::CSTEPS * 3)) % NCH, ct = idx / (192 * dcv::CSTEPS * 3 * NCH);
    const int n = ct * 16 + (lane & 15), q = 4 * dcv::cs_step(cs) + (lane >> 4);
    int kh, kw, di, dj, oct;
    const bool ok = dcv::tap_of(dcv::cs_ph(cs), dcv::cs_pw(cs), q, &kh, &kw, &di, &dj, &oct) && n < Cout;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = 0.0f;
        if (ok) f = w[((size_t)(chunk * 16 + oct * 8 + e) * Cout + n) * 27 + kd * 9 + kh * 3 + kw];
        v[e] = mvsx3::split3_term(f, term);
    }
    out[idx] = v;
}

#ifndef X3_DECONV_PREFETCH
#define X3_DECONV_PREFETCH 1
#endif
// H, W = INPUT size; output [B,Cout,D,2H,2W]
__global__ __launch_bounds__(256, 2) void x3_deconv_kernel(const X3Args a) {
    using namespace dcv;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;               // as the A (activation) operand: pixel column j = n; as the accumulator: channel n
#if X3_XCD_ORDER
    unsigned nid_ = mvsconv::xcd_linear_block_id();            // (the block order of x3_conv_kernel)
    const int ctb = (int)(nid_ % gridDim.y);
    nid_ /= gridDim.y;
    const int seg = (int)(nid_ % (unsigned)a.nseg);
    nid_ /= (unsigned)a.nseg;
    const int tile = (int)(nid_ % gridDim.x), b = (int)(nid_ / gridDim.x);
#else
    const int tile = blockIdx.x, ctb = blockIdx.y, b = blockIdx.z / a.nseg, seg = blockIdx.z % a.nseg;
#endif
    const int x0 = (tile % a.tiles_x) * TIW, y0 = (tile / a.tiles_x) * TIH;              // input coordinates
    const int Cin = a.Cin, Cout = a.Cout, D = a.D, H = a.H, W = a.W, Ho = 2 * H, Wo = 2 * W;
    const int NCH = Cin / 16;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW, HWo = (size_t)Ho * Wo;
    const float* xb = a.x + (size_t)b * Cin * DHW;

    // activation-operand byte offsets of this lane's K block in each class step (row / term parts added later)
    unsigned boff[CSTEPS];
#pragma unroll
    for (int cs = 0; cs < CSTEPS; ++cs) {
        int kh, kw, di, dj, oct;
        if (!tap_of(cs_ph(cs), cs_pw(cs), 4 * cs_step(cs) + kb, &kh, &kw, &di, &dj, &oct)) { di = 0; dj = 0; }      // zero weights: any valid address
        boff[cs] = (unsigned)((di * BW + n + dj) * PB + oct * 16);
    }

    f32x4 acc[3][4][2];                                    // [output plane p-1 | p | p+1][parity class][input row of the wavefront]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[s][c][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    // accumulator [pixel quad 4kb..4kb+3 of the class][channel n]: with both column classes of a row in one lane, the 8 output pixels
    // 8kb .. 8kb+7 of the row are two 16-byte stores
    const int co = ctb * 16 + n;
    const float sc = (a.scale && co < Cout) ? a.scale[co] : 1.0f, sh = (a.shift && co < Cout) ? a.shift[co] : 0.0f;
    // The skip tensor of a finished plane (round 4): its 16-byte loads go out TOGETHER with the staging loads of the plane's last channel chunk,
    // so the one wait that staging needs anyway covers them too (vmcnt retires in order; issued in the epilogue they cost a second
    // exposed HBM round trip per plane); buffer loads: rows / columns outside the output read 0 and are never stored.
    const rsrc_t rin = make_rsrc(a.residual ? a.residual + (size_t)b * Cout * D * HWo : a.x, a.residual ? (unsigned)((size_t)Cout * D * HWo * 4) : 0u);
    unsigned roff[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int oy = 2 * (y0 + wave * 2 + r) + ph, ox = 2 * (x0 + kb * 4);
            roff[r][ph] = (co < Cout && oy < Ho && ox < Wo) ? (unsigned)(((size_t)co * D * HWo + (size_t)oy * Wo + ox) * 4) : OOB;
        }
    f32x4 rs[2][2][2];
    auto issue_residual = [&](int od) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    rs[r][ph][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, roff[r][ph] + 16u * h, (unsigned)((size_t)od * HWo * 4), X3_STAGE_AUX));
    };
    auto store_plane = [&](int od, const f32x4 (&c)[4][2]) {
        if (co >= Cout) return;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int oy = 2 * (y0 + wave * 2 + r) + ph, ox = 2 * (x0 + kb * 4);
                if (oy >= Ho || ox >= Wo) continue;
                float v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = c[ph * 2 + 0][r][q];
                    v[2 * q + 1] = c[ph * 2 + 1][r][q];
                }
                const size_t o = ((size_t)(b * Cout + co) * D + od) * HWo + (size_t)oy * Wo + ox;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    v[q] = a.scale ? fmaf(v[q], sc, sh) : v[q] + sh;
                    if (a.relu) v[q] = fmaxf(v[q], 0.0f);
                }
                // Wo = 2W with W even: a multiple of 4, so each 16-byte half is either fully inside the row or fully outside
                const bool second = ox + 4 < Wo;
                if (a.residual) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q] += rs[r][ph][0][q];
                        v[4 + q] += rs[r][ph][1][q];       // (second half beyond the row: the load returned 0, the value is not stored)
                    }
                }
                *reinterpret_cast<f32x4*>(a.y + o) = f32x4{v[0], v[1], v[2], v[3]};
                if (second) *reinterpret_cast<f32x4*>(a.y + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
    };

    bf16x8 wbuf[2][3];
    auto load_w = [&](const bf16x8* wk, bf16x8 (&aw)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t) aw[t] = wk[t * 64];
    };
    auto kd_steps = [&](auto kd_tag, auto p_tag, const bf16x8* wk, bool more) {
        constexpr int KD = decltype(kd_tag)::value, P = decltype(p_tag)::value, SET = KD;      // od = p - 1 + kd
#pragma unroll
        for (int cs = 0; cs < CSTEPS; ++cs) {
            const int cur = (P + cs) & 1;
            if (cs + 1 < CSTEPS || more) load_w(wk + (size_t)(cs + 1) * 192, wbuf[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned char* bp = lds + (wave * 2 + r) * (BW * PB) + boff[cs];
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(bp);
                const bf16x8 xm = *reinterpret_cast<const bf16x8*>(bp + TERM_BYTES);
                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(bp + 2 * TERM_BYTES);
                f32x4 c = acc[SET][cs_class(cs)][r];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, wbuf[cur][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wbuf[cur][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wbuf[cur][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wbuf[cur][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, wbuf[cur][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wbuf[cur][0], c, 0, 0, 0);
                acc[SET][cs_class(cs)][r] = c;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // staging through BUFFER loads (round 4, as x3_conv_kernel): one lane offset per item for the whole kernel, chunk / plane in the scalar
    // offset, pixels outside the volume read 0 through the descriptor's range check
    constexpr int NI = (2 * NPIX + 255) / 256;
    const rsrc_t xin = make_rsrc(xb, (unsigned)((size_t)Cin * DHW * 4));
    unsigned voff[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int i = tid + it * 256;
        const int oct = i / NPIX, v = i % NPIX;
        const int gy = y0 + v / BW, gx = x0 + v % BW;
        voff[it] = (i < 2 * NPIX && gy < H && gx < W) ? (unsigned)(((size_t)(oct * 8) * DHW + (size_t)gy * W + gx) * 4) : OOB;
    }

    const int d_lo = seg * a.seg_planes, d_hi = min(D, d_lo + a.seg_planes);
    if (d_lo >= D) return;                                 // empty segment (block-uniform)
    const int p_first = max(0, d_lo - 1), p_last = min(D - 1, d_hi);
    for (int p = p_first; p <= p_last; ++p) {
        // depth taps whose output plane od = p - 1 + kd lies in [d_lo, d_hi)
        const int kd_lo = max(0, d_lo + 1 - p), kd_hi = min(2, d_hi - p);
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const bf16x8* wk = a.wp + ((size_t)(ctb * NCH + chunk) * 3 + kd_lo) * FRAGS_PER_KD + lane;
            load_w(wk, wbuf[0]);
#if X3_DECONV_PREFETCH
            if (chunk == NCH - 1 && a.residual && p - 1 >= d_lo) issue_residual(p - 1);
#endif
            // both of a thread's items' loads first (16 in flight), then the barrier, the split and the LDS stores
            float pre[NI][8];
            {
                const size_t base = (size_t)(chunk * 16) * DHW + (size_t)p * HW;
#pragma unroll
                for (int it = 0; it < NI; ++it)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pre[it][e] = stage_load(xin, voff[it], (unsigned)((base + (size_t)e * DHW) * 4));
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int i = tid + it * 256;
                if (i < 2 * NPIX) {
                    const Split3 sp = split3(pre[it]);
                    unsigned char* dst = lds + (i % NPIX) * PB + (i / NPIX) * 16;
                    *reinterpret_cast<bf16x8*>(dst) = sp.h;
                    *reinterpret_cast<bf16x8*>(dst + TERM_BYTES) = sp.m;
                    *reinterpret_cast<bf16x8*>(dst + 2 * TERM_BYTES) = sp.l;
                }
            }
            __syncthreads();
            if (X3_SETPRIO) __builtin_amdgcn_s_setprio(1);
            int pos = 0;
            if (kd_lo == 0 && kd_hi >= 0) {
                kd_steps(ic<0>{}, ic<0>{}, wk, kd_hi >= 1);
                wk += FRAGS_PER_KD;
                pos = 1;
            }
            if (kd_lo <= 1 && kd_hi >= 1) {
                if (pos == 0) kd_steps(ic<1>{}, ic<0>{}, wk, kd_hi >= 2); else kd_steps(ic<1>{}, ic<1>{}, wk, kd_hi >= 2);
                wk += FRAGS_PER_KD;
                pos ^= 1;
            }
            if (kd_lo <= 2 && kd_hi >= 2) {
                if (pos == 0) kd_steps(ic<2>{}, ic<0>{}, wk, false); else kd_steps(ic<2>{}, ic<1>{}, wk, false);
            }
            if (X3_SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
        // output plane p-1 has seen input planes p-2, p-1, p
#if !X3_DECONV_PREFETCH
        if (a.residual && p - 1 >= d_lo) issue_residual(p - 1);
#endif
        if (p - 1 >= d_lo) store_plane(p - 1, acc[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                acc[0][c][r] = acc[1][c][r];
                acc[1][c][r] = acc[2][c][r];
                acc[2][c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    if (p_last == D - 1 && d_hi == D) {
        if (a.residual) issue_residual(D - 1);
        store_plane(D - 1, acc[0]);
    }
}

// which instance serves a layer: CK, rows per wavefront, M tiles per block
struct X3Plan { int ck, nt, mtb; };
bool x3_plan(int Cin, int Cout, int sd, int shw, X3Plan* pl) {
    if (sd != 1 || (shw != 1 && shw != 2)) return false;
    if (Cout != 16 && Cout != 32 && Cout != 64) return false;
    if (shw == 1) {
        if (Cin != 16 && Cin != 32 && Cin != 64) return false;
        *pl = Cout >= 32 ? X3Plan{16, 2, 2} : X3Plan{16, 4, 1};
        return true;
    }
    if (Cin == 8) { *pl = X3Plan{8, 4, 1}; return Cout == 16; }
    if (Cin != 16 && Cin != 32) return false;
    *pl = X3Plan{16, 2, Cout >= 32 ? 2 : 1};
    return true;
}

// Depth segments (each re-stages one halo plane per cut) are added while the grid has fewer blocks than this.  Measured per layer at config-2 stages
// 3-4 (profiles/r04_bench_x3_segments.txt): the two-blocks-per-CU instances want a grid of ~1 block per slot and no more (conv5 / conv6 / conv7 at
// stage 4: 432 blocks unsplit 0.077 / 0.118 / 0.091 ms against 0.090 / 0.130 / 0.102 split in two - with D = 4 a cut stages 3 planes for 2), the
// three-blocks-per-CU instances (conv1, conv2) keep the old bound (conv2 at stage 3: 0.081 split in two against 0.086 unsplit).
// MVS_X3_SEG_BLOCKS overrides both (diagnostics).
int seg_blocks(int blocks_per_cu) {
    static const int env = [] {
        const char* e = getenv("MVS_X3_SEG_BLOCKS");
        return e ? std::max(1, atoi(e)) : 0;
    }();
    return env ? env : (blocks_per_cu >= 3 ? 1536 : 384);
}

template <class Cfg>
int launch_x3(X3Args a, int B, hipStream_t s) {
    const int ty = mvs::ceil_div(a.Ho, Cfg::TH);
    // depth segments: every segment re-stages one halo plane on each side, so split only while the grid is short of ~6 blocks per CU
    const int64_t blocks = (int64_t)a.tiles_x * ty * mvs::ceil_div(a.Cout, 16 * Cfg::MTB) * B;
    int nseg = 1;
    while (nseg * 2 <= a.D / 2 && blocks * nseg < seg_blocks(Cfg::MIN_BLOCKS)) nseg *= 2;
    a.seg_planes = mvs::ceil_div(a.D, nseg);
    nseg = mvs::ceil_div(a.D, a.seg_planes);               // no empty segments (D = 9: 4 x 3 planes would leave segment 3 = [9, 9) writing plane D-1)
    a.nseg = nseg;
    hipLaunchKernelGGL((x3_conv_kernel<Cfg>), dim3(a.tiles_x * ty, mvs::ceil_div(a.Cout, 16 * Cfg::MTB), B * nseg), dim3(256), 0, s, a);
    return mvs::finish_launch("mvs_conv3d_x3_fwd");
}

}  // namespace

extern "C" int mvs_conv3d_x3_supported(int Cin, int Cout, int sd, int shw) {
    X3Plan pl;
    return x3_plan(Cin, Cout, sd, shw, &pl) ? 1 : 0;
}

extern "C" int64_t mvs_conv3d_x3_packed_bytes(int Cin, int Cout, int sd, int shw) {
    X3Plan pl;
    if (!x3_plan(Cin, Cout, sd, shw, &pl)) return 0;
    const int steps = (9 * (pl.ck / 8) + 3) / 4;
    return ((int64_t)(Cout / 16) * (Cin / pl.ck) * 3 + 1) * steps * 3 * 64 * 16;        // + one zero unit for the running prefetch
}

extern "C" int mvs_conv3d_x3_pack_weights(const float* w, int Cin, int Cout, int sd, int shw, void* wpacked, mvs_stream_t stream) {
    MVS_REQUIRE(w && wpacked, "mvs_conv3d_x3_pack_weights: null pointer");
    X3Plan pl;
    MVS_REQUIRE(x3_plan(Cin, Cout, sd, shw, &pl), "mvs_conv3d_x3_pack_weights: Cin=%d Cout=%d stride (%d,%d,%d) is not built", Cin, Cout, sd, shw, shw);
    const int total = (int)(mvs_conv3d_x3_packed_bytes(Cin, Cout, sd, shw) / 16);
    hipLaunchKernelGGL(x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, MVS_STREAM(stream), w, Cin, Cout, pl.ck, static_cast<bf16x8*>(wpacked), total);
    return mvs::finish_launch("mvs_conv3d_x3_pack_weights");
}

extern "C" int mvs_conv3d_x3_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual, float* y,
                                 int B, int Cin, int Cout, int D, int H, int W, int sd, int shw, int relu, mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && y, "mvs_conv3d_x3_fwd: null pointer");
    X3Plan pl;
    MVS_REQUIRE(x3_plan(Cin, Cout, sd, shw, &pl), "mvs_conv3d_x3_fwd: Cin=%d Cout=%d stride (%d,%d,%d) is not built", Cin, Cout, sd, shw, shw);
    MVS_REQUIRE(B >= 1 && B <= 65535 && D >= 1 && H >= 1 && W >= 1, "mvs_conv3d_x3_fwd: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
    MVS_REQUIRE(!scale || shift, "mvs_conv3d_x3_fwd: scale without shift");
    MVS_REQUIRE((int64_t)Cin * D * H * W * 4 < ((int64_t)1 << 31), "mvs_conv3d_x3_fwd: one sample's input exceeds the 2 GiB buffer window");
    X3Args a;
    a.x = x; a.wp = static_cast<const bf16x8*>(wpacked); a.scale = scale; a.shift = shift; a.residual = residual; a.y = y;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = relu;
    a.Ho = (H - 1) / shw + 1;
    a.Wo = (W - 1) / shw + 1;
    a.tiles_x = mvs::ceil_div(a.Wo, 16);
    hipStream_t s = MVS_STREAM(stream);
    if (shw == 1) return pl.mtb == 2 ? launch_x3<X3Cfg<16, 1, 2, 2>>(a, B, s) : launch_x3<X3Cfg<16, 1, 4, 1>>(a, B, s);
    if (pl.ck == 8) return launch_x3<X3Cfg<8, 2, 4, 1>>(a, B, s);
    return pl.mtb == 2 ? launch_x3<X3Cfg<16, 2, 2, 2>>(a, B, s) : launch_x3<X3Cfg<16, 2, 2, 1>>(a, B, s);
}

#ifdef X3_PRESPLIT
// experiment entry (not in the header, experiment builds only; tools/exp_presplit.py): mode 0 = conv1 (8 -> 16, stride (1,2,2)) as shipped,
// 1 = conv1 writing ONLY the pre-split form, 2 = conv1 writing both, 3 = conv2 (16 -> 16, stride 1) as shipped, 4 = conv2 staged from the pre-split form
extern "C" int mvs_x3_presplit_exp(int mode, const float* x, const void* xpre, const void* wpacked, const float* scale, const float* shift, float* y,
                                   void* ypre, int B, int D, int H, int W, int relu, mvs_stream_t stream) {
    X3Args a{};
    a.x = x; a.wp = static_cast<const bf16x8*>(wpacked); a.scale = scale; a.shift = shift; a.residual = nullptr; a.y = y;
    a.xpre = xpre; a.ypre = ypre; a.fp32_out = mode != 1;
    a.D = D; a.H = H; a.W = W; a.relu = relu; a.Cout = 16;
    const int shw = mode <= 2 ? 2 : 1;
    a.Cin = mode <= 2 ? 8 : 16;
    a.Ho = (H - 1) / shw + 1;
    a.Wo = (W - 1) / shw + 1;
    a.tiles_x = mvs::ceil_div(a.Wo, 16);
    hipStream_t s = MVS_STREAM(stream);
    using C1 = X3Cfg<8, 2, 4, 1>;
    using C2 = X3Cfg<16, 1, 4, 1>;
    const int ty = mvs::ceil_div(a.Ho, 16);
    a.seg_planes = D, a.nseg = 1;
    const dim3 grid(a.tiles_x * ty, 1, B);
    if (mode == 0) hipLaunchKernelGGL((x3_conv_kernel<C1, false, false>), grid, dim3(256), 0, s, a);
    else if (mode <= 2) hipLaunchKernelGGL((x3_conv_kernel<C1, false, true>), grid, dim3(256), 0, s, a);
    else if (mode == 3) hipLaunchKernelGGL((x3_conv_kernel<C2, false, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((x3_conv_kernel<C2, true, false>), grid, dim3(256), 0, s, a);
    return mvs::finish_launch("mvs_x3_presplit_exp");
}
#endif

extern "C" int mvs_deconv3d_x3_supported(int Cin, int Cout, int sd) {
    return sd == 1 && (Cin == 16 || Cin == 32 || Cin == 64) && (Cout == 8 || Cout == 16 || Cout == 32);
}

extern "C" int64_t mvs_deconv3d_x3_packed_bytes(int Cin, int Cout, int sd) {
    if (!mvs_deconv3d_x3_supported(Cin, Cout, sd)) return 0;
    return ((int64_t)((Cout + 15) / 16) * (Cin / 16) * 3 + 1) * dcv::FRAGS_PER_KD * 16;
}

extern "C" int mvs_deconv3d_x3_pack_weights(const float* w, int Cin, int Cout, int sd, void* wpacked, mvs_stream_t stream) {
    MVS_REQUIRE(w && wpacked, "mvs_deconv3d_x3_pack_weights: null pointer");
    MVS_REQUIRE(mvs_deconv3d_x3_supported(Cin, Cout, sd), "mvs_deconv3d_x3_pack_weights: Cin=%d Cout=%d sd=%d is not built", Cin, Cout, sd);
    const int total = (int)(mvs_deconv3d_x3_packed_bytes(Cin, Cout, sd) / 16);
    hipLaunchKernelGGL(x3_deconv_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, MVS_STREAM(stream), w, Cin, Cout, static_cast<bf16x8*>(wpacked), total);
    return mvs::finish_launch("mvs_deconv3d_x3_pack_weights");
}

/* x [B,Cin,D,H,W] -> y [B,Cout,D,2H,2W] = [relu](conv_transpose3d(x, w, stride (1,2,2), padding 1, output_padding (0,1,1)) * scale + shift)
 * [+ residual] */
extern "C" int mvs_deconv3d_x3_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual, float* y,
                                   int B, int Cin, int Cout, int D, int H, int W, int sd, int relu, mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && y, "mvs_deconv3d_x3_fwd: null pointer");
    MVS_REQUIRE(mvs_deconv3d_x3_supported(Cin, Cout, sd), "mvs_deconv3d_x3_fwd: Cin=%d Cout=%d sd=%d is not built", Cin, Cout, sd);
    MVS_REQUIRE(B >= 1 && B <= 65535 && D >= 1 && H >= 1 && W >= 1 && (W % 2) == 0, "mvs_deconv3d_x3_fwd: bad shape B=%d D=%d H=%d W=%d (W even)", B, D, H, W);
    MVS_REQUIRE(!scale || shift, "mvs_deconv3d_x3_fwd: scale without shift");
    MVS_REQUIRE((int64_t)Cin * D * H * W * 4 < ((int64_t)1 << 31), "mvs_deconv3d_x3_fwd: one sample's input exceeds the 2 GiB buffer window");
    // the skip tensor is read through a buffer descriptor too (32-bit size and offsets): one sample's OUTPUT [Cout, D, 2H, 2W] must fit
    MVS_REQUIRE(!residual || (int64_t)Cout * D * 4 * H * W * 4 < ((int64_t)1 << 32),
                "mvs_deconv3d_x3_fwd: one sample's skip tensor exceeds the 4 GiB buffer window");
    X3Args a;
    a.x = x; a.wp = static_cast<const bf16x8*>(wpacked); a.scale = scale; a.shift = shift; a.residual = residual; a.y = y;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = relu; a.Ho = 2 * H; a.Wo = 2 * W;
    a.tiles_x = mvs::ceil_div(W, dcv::TIW);
    const int ty = mvs::ceil_div(H, dcv::TIH), cts = (Cout + 15) / 16;
    const int64_t blocks = (int64_t)a.tiles_x * ty * cts * B;
    int nseg = 1;
    while (nseg * 2 <= D / 2 && blocks * nseg < seg_blocks(2)) nseg *= 2;
    a.seg_planes = mvs::ceil_div(D, nseg);
    nseg = mvs::ceil_div(D, a.seg_planes);                 // no empty segments
    a.nseg = nseg;
    hipLaunchKernelGGL(x3_deconv_kernel, dim3(a.tiles_x * ty, cts, B * nseg), dim3(256), 0, MVS_STREAM(stream), a);
    return mvs::finish_launch("mvs_deconv3d_x3_fwd");
}

#ifdef X3_TIMELINE
extern "C" int mvs_x3_timeline(unsigned long long* out8, int reset) {
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(x3_phase_ticks), 64);
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(x3_phase_ticks), z, 64);
    }
    return 0;
}
#endif
