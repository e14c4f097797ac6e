// Visibility-weight CNN of StageNet, third generation: the two 3x3 layers that carry 94 % of its FLOPs (ConvBnReLU(16,16),
// ConvBnReLU(16,8); models/mvsformer_model.py:37,91, ConvBnReLU = module.py:168-197) as DIRECT convolutions on the BF16 matrix cores in
// the three-term split form of conv3d_x3.hip (every fp32 value = h + m + l exactly, six v_mfma_f32_16x16x32_bf16 per fp32-equivalent
// K = 32 step, fp32 accumulation: fp32 in, fp32 out, fp32-equivalent), fused with layer 1 (fp32 MFMA: its input is the raw entropy), the
// 1x1 conv and the sigmoid in ONE launch with every intermediate in LDS.
//
// Why a third form.  vis_net_wino.hip's Winograd GEMMs run on v_mfma_f32_16x16x4_f32, which shares the issue port with every vector
// instruction (profiles/r03_ubench.txt): its 0.24 ms of matrix time and its transforms ADD (0.58 ms at stage 4, matrix pipe 46 % busy).
// The split form needs about the same matrix time (6 MFMAs of 16 clk per K = 32 against 8 of 32 clk, but no 2.25x Winograd saving), on a
// pipe that leaves the vector ALU free, and no transforms at all: what remains on the vector side is the split itself (9 issue slots per
// intermediate value) and the BatchNorm / ReLU epilogues.
//
//   * block tile = 16 x 16 outputs; layer 2 is needed on 18 x 18, layer 1 on 20 x 20, the entropy on 22 x 22 (each conv zero-pads ITS
//     OWN input: activations at positions outside the image are stored as 0);
//   * LDS activations: [term h|m|l][channel octet][pixel][8 bf16] - 16 consecutive pixels of one octet are 256 contiguous bytes, so the
//     MFMA B operand (N = 16 pixels, K block = 8 channels of one tap) is one conflict-free ds_read_b128 per term; a layer's output tile
//     D[m = 4 channels][n = pixel] is split in the lane and leaves as three ds_write_b64;
//   * layer 2: M = 16 output channels, N = 16 consecutive pixels of the FLATTENED 18 x 18 region (no ragged rows), K = 9 taps x 16
//     channels = 4.5 steps of 32 (the last half step multiplies zero weights);
//   * layer 3 (8 output channels would fill half of M): M = (output row parity, channel) - two output rows share the FOUR input rows
//     they see, K = 4 rows x 3 columns x 16 channels = exactly 6 steps, a quarter of the A operand is structural zeros instead of half;
//   * weights of both layers, pre-split and laid out per lane by mvs_vis_x3_prepare, live in 132 VGPRs for the whole (persistent) kernel.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "split3.h"

// the ablation switches of the timeline experiments exist only in experiment builds (make exp EXPFLAGS=-DX3_ABLATE): the shipped kernel has no
// environment-dependent path (as tail_x3.hip)
#ifdef X3_ABLATE
#define VIS_ABLATE(a, bit) (((a) & (bit)) != 0)
static int vis_ablate_bits() { const char* e = getenv("MVS_VIS_ABLATE"); return e ? atoi(e) : 0; }
#else
#define VIS_ABLATE(a, bit) false
static constexpr int vis_ablate_bits() { return 0; }
#endif

namespace {
using namespace mvsconv;
using mvsx3::bf16x4;
using mvsx3::bf16x8;
using mvsx3::split3;

constexpr int T = 16;                                      // output tile edge
constexpr int INW = T + 6, A1W = T + 4, A2W = T + 2;       // entropy (halo 3), layer-1 output (halo 2), layer-2 output (halo 1)
constexpr int A1N = A1W * A1W, A2N = A2W * A2W;            // 400, 324 pixels
constexpr int A1_OCT = A1N * 16, A1_TERM = 2 * A1_OCT;     // bytes
constexpr int A2_OCT = A2N * 16, A2_TERM = 2 * A2_OCT;
constexpr int IN_BYTES = ((INW * INW * 4 + 255) / 256) * 256;
constexpr int IDX_BYTES = A1N * 4;                          // layer 1's walk: per flattened pixel {byte offset in the entropy tile, row, column}
constexpr int LDS_BYTES = IN_BYTES + 3 * A1_TERM + 3 * A2_TERM + IDX_BYTES;
constexpr int L2_STEPS = 5, L3_STEPS = 6;
// Layer 2's N tiles walk its 18 x 18 output region with the PITCH OF ITS INPUT (20): output index o = oy * 20 + ox, so the B fragment of tap
// (kh, kw) is pixels o + kh * 20 + kw ... + 15 of layer 1's region - 256 contiguous bytes for every tile, where the 18-pitch walk of round
// 3 wrapped around a row end inside most tiles (288-byte span: SQ_LDS_BANK_CONFLICT 45 % of SQ_LDS_IDX_ACTIVE).  The two junk columns per
// row (ox = 18, 19) make 23 tiles instead of 21 - the same six tiles on the busiest wavefront - and are dropped at the store.
constexpr int L2_FLAT = A2W * A1W;                         // 360
constexpr int L2_TILES = (L2_FLAT + 15) / 16, L1_TILES = A1N / 16;   // 23, 25
static_assert(A1N % 16 == 0, "layer 1 tiles are full");
// offsets inside the MVS_VIS_PARAM_FLOATS block (vis_net.hip)
constexpr int OFF_W0 = 0, OFF_S0 = 144, OFF_B0 = 160, OFF_W1 = 176, OFF_S1 = 2480, OFF_B1 = 2496, OFF_W2 = 2512, OFF_S2 = 3664,
              OFF_B2 = 3672, OFF_W3 = 3680, OFF_B3 = 3688;

// prepared[(layer 2: step 0..4 | layer 3: step 5..10)][term][lane][8]:  the MFMA A operand, lane = kb * 16 + m.  The folded BatchNorm SCALE of
// the output channel is multiplied into the weight before the split and the SHIFT is the accumulator's start value, so the epilogue of a
// layer is one v_max / v_med3 per value (the sums then round as shift + products instead of (products) * scale + shift: fp32-equivalent).
//   layer 2: m = output channel, K block 2*step + (kb >> 1) = tap (9 = zero), channels (kb & 1) * 8 + e
//   layer 3: m = (dy = m >> 3, co = m & 7), K block 2*step + (kb >> 1) = (input row t, kw), weight of kh = t - dy (outside 0..2: zero)
__global__ void vis_x3_prepare_kernel(const float* __restrict__ prm, bf16x8* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (L2_STEPS + L3_STEPS) * 3 * 64) return;
    const int lane = idx & 63, term = (idx >> 6) % 3, step = idx / 192;
    const int m = lane & 15, kb = lane >> 4;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = (kb & 1) * 8 + e;
        float f = 0.0f;
        if (step < L2_STEPS) {
            const int tap = 2 * step + (kb >> 1);
            if (tap < 9) f = prm[OFF_W1 + (ci * 9 + tap) * 16 + m] * prm[OFF_S1 + m];
        } else {
            const int kblk = 2 * (step - L2_STEPS) + (kb >> 1), t = kblk / 3, kw = kblk % 3, dy = m >> 3, co = m & 7, kh = t - dy;
            if (kh >= 0 && kh <= 2) f = prm[OFF_W2 + (ci * 9 + kh * 3 + kw) * 8 + co] * prm[OFF_S2 + co];
        }
        __bf16 h, mm, l;
        split3(f, h, mm, l);
        v[e] = term == 0 ? h : (term == 1 ? mm : l);
    }
    out[idx] = v;
}

// six MFMAs of one fp32-equivalent K = 32 step, smallest products first (the order of conv3d_x3.hip)
__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], c, 0, 0, 0);
    return c;
}

// the lane's 4 channels (4*kb .. 4*kb+3) of pixel `pix`, split and stored: [term][octet = kb >> 1][pix][(kb & 1) * 8 bytes]
template <int OCT, int TERM>
__device__ __forceinline__ void store_split(unsigned char* base, int pix, int kb, const float (&v)[4]) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 h, m, l;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        unsigned a, b, c;
#ifdef VIS_SPLIT_CLAMPED
        mvsx3::split3_pair<true>(v[2 * r], v[2 * r + 1], a, b, c);
#else
        mvsx3::split3_pair<false>(v[2 * r], v[2 * r + 1], a, b, c);    // bounded by construction (split3_bounded): no clamp
#endif
        h[r] = a; m[r] = b; l[r] = c;
    }
    unsigned char* dst = base + (kb >> 1) * OCT + pix * 16 + (kb & 1) * 8;
    *reinterpret_cast<u32x2*>(dst) = h;
    *reinterpret_cast<u32x2*>(dst + TERM) = m;
    *reinterpret_cast<u32x2*>(dst + 2 * TERM) = l;
}

__global__ __launch_bounds__(256, 2) void vis_x3_kernel(const float* __restrict__ entropy, const float* __restrict__ prm,
                                                        const bf16x8* __restrict__ prep, int N, int H, int W, int ntx, int nty,
                                                        float* __restrict__ weight, int ablate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_in = reinterpret_cast<float*>(smem);          // [INW][INW]
    unsigned char* s_a1 = smem + IN_BYTES;                 // [3][2][A1N][16 B]
    unsigned char* s_a2 = s_a1 + 3 * A1_TERM;              // [3][2][A2N][16 B]
    unsigned* s_idx = reinterpret_cast<unsigned*>(s_a2 + 3 * A2_TERM);   // [A1N]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;

    // both layers' weights (A operands) for the whole kernel
    bf16x8 w2[L2_STEPS][3], w3[L3_STEPS][3];
#pragma unroll
    for (int s = 0; s < L2_STEPS; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) w2[s][t] = prep[(s * 3 + t) * 64 + lane];
#pragma unroll
    for (int s = 0; s < L3_STEPS; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) w3[s][t] = prep[((L2_STEPS + s) * 3 + t) * 64 + lane];

    // layer 1 as fp32 MFMAs: A = w0[tap = 4t + kb][co = n] (taps 9-11: zero), B = the entropy at the tap's offset
    float W1A[3];
    int tap_off[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = 4 * t + kb;
        W1A[t] = tap < 9 ? prm[OFF_W0 + tap * 16 + n] * prm[OFF_S0 + n] : 0.0f;
        const int tc = tap < 9 ? tap : 8;
        tap_off[t] = (tc / 3) * INW + tc % 3;
    }
    // this lane's 4 output channels of layers 1 and 2 (4*kb + r), of layer 3 ((kb & 1) * 4 + r)
    f32x4 sh0, sh1, sh2;
    float wl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sh0[r] = prm[OFF_B0 + 4 * kb + r];
        sh1[r] = prm[OFF_B1 + 4 * kb + r];
        const int co = (kb & 1) * 4 + r;
        sh2[r] = prm[OFF_B2 + co];
        wl[r] = prm[OFF_W3 + co];
    }
    const float b3 = prm[OFF_B3];
    // B-operand byte offsets of this lane's K block per step (tap and octet parts)
    unsigned off2[L2_STEPS], off3[L3_STEPS];
#pragma unroll
    for (int s = 0; s < L2_STEPS; ++s) {
        const int tap = min(2 * s + (kb >> 1), 8);         // (tap 9 multiplies zero weights)
        off2[s] = (unsigned)((kb & 1) * A1_OCT + ((tap / 3) * A1W + tap % 3) * 16);
    }
#pragma unroll
    for (int s = 0; s < L3_STEPS; ++s) {
        const int kblk = 2 * s + (kb >> 1);
        off3[s] = (unsigned)((kb & 1) * A2_OCT + ((kblk / 3) * A2W + kblk % 3) * 16);
    }

    const int ntiles = N * ntx * nty;
    constexpr int EPT = (INW * INW + 255) / 256;           // entropy values per thread (2)
    float pre[EPT];
    auto tile_origin = [&](int tile, int& img, int& x0, int& y0) {
        img = tile / (ntx * nty);
        y0 = ((tile / ntx) % nty) * T;
        x0 = (tile % ntx) * T;
    };
    auto fetch_entropy = [&](int tile) {
        int img, x0, y0;
        tile_origin(tile, img, x0, y0);
        const float* src = entropy + (size_t)img * H * W;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * 256;
            const int gy = y0 - 3 + i / INW, gx = x0 - 3 + i % INW;
            pre[e] = (i < INW * INW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : 0.0f;
        }
    };
    auto commit_entropy = [&]() {
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (tid + e * 256 < INW * INW) s_in[tid + e * 256] = pre[e];
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    fetch_entropy(tile);
    commit_entropy();
    for (int p = tid; p < A1N; p += 256) {
        const int py = p / A1W, px = p % A1W;
        s_idx[p] = (unsigned)((py * INW + px) * 4) | (unsigned)py << 16 | (unsigned)px << 24;
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        int img, x0, y0;
        tile_origin(tile, img, x0, y0);
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;               // block-uniform
        if (has_next) fetch_entropy(next);

        // A tile whose 20 x 20 layer-1 region lies inside the image (95 % of them at stage 4) needs no padding mask in either epilogue:
        // block-uniform, so both forms of the two loops exist and a tile takes one branch
        const bool interior = x0 >= 2 && y0 >= 2 && x0 + T + 2 <= W && y0 + T + 2 <= H;

        // ---- layer 1: 1 -> 16 on the 20 x 20 region, N = 16 consecutive pixels of the flattened region; the walk's divisions by 20 come
        //      from a table built once per block ----
        auto layer1 = [&](auto inner) {
            constexpr bool INNER = decltype(inner)::value;
#pragma unroll 1
            for (int t = wave; t < L1_TILES && !VIS_ABLATE(ablate, 1); t += 4) {
                const int p = t * 16 + n;
                const unsigned ix = s_idx[p];
                const float* src = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(s_in) + (ix & 0xffffu));
                f32x4 z = sh0;
#pragma unroll
                for (int k = 0; k < 3; ++k) z = mfma4(W1A[k], src[tap_off[k]], z);
                // ReLU and the zero padding outside the image in one v_med3_f32: med3(x, 0, +inf) = max(x, 0), med3(x, 0, 0) = 0
                float cap = __builtin_inff();
                if (!INNER) {
                    const int gy = y0 - 2 + (int)((ix >> 16) & 0xffu), gx = x0 - 2 + (int)(ix >> 24);
                    cap = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? __builtin_inff() : 0.0f;
                }
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = INNER ? fmaxf(z[r], 0.0f) : __builtin_amdgcn_fmed3f(z[r], 0.0f, cap);
                store_split<A1_OCT, A1_TERM>(s_a1, p, kb, v);
            }
        };
        if (interior) layer1(std::true_type{}); else layer1(std::false_type{});
        __syncthreads();

        // ---- layer 2: 16 -> 16 on the 18 x 18 region; two pixel tiles per pass (independent MFMA chains) ----
#pragma unroll 1
        for (int t = wave; t < L2_TILES; t += 8) {
            const int o0 = t * 16 + n, o1 = min(t + 4, L2_TILES - 1) * 16 + n;        // (reads of junk pixels stay inside the LDS block)
            const bool two = t + 4 < L2_TILES;             // wave-uniform
            const unsigned char* b0 = s_a1 + o0 * 16;
            const unsigned char* b1 = s_a1 + o1 * 16;
            f32x4 c0 = sh1, c1 = sh1;
            bf16x8 x0f[2][3], x1f[2][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                x0f[0][k] = *reinterpret_cast<const bf16x8*>(b0 + off2[0] + k * A1_TERM);
                x1f[0][k] = *reinterpret_cast<const bf16x8*>(b1 + off2[0] + k * A1_TERM);
            }
#ifndef VIS_SETPRIO
#define VIS_SETPRIO 1
#endif
            if (VIS_SETPRIO) __builtin_amdgcn_s_setprio(1);    // the other block of the CU is in another phase: the MFMA steps win the issue arbitration
#pragma unroll
            for (int s = 0; s < L2_STEPS; ++s) {
                if (s + 1 < L2_STEPS) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        x0f[(s + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(b0 + off2[s + 1] + k * A1_TERM);
                        x1f[(s + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(b1 + off2[s + 1] + k * A1_TERM);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!VIS_ABLATE(ablate, 2)) {
                    c0 = mfma6(w2[s], x0f[s & 1], c0);
                    c1 = mfma6(w2[s], x1f[s & 1], c1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (VIS_SETPRIO) __builtin_amdgcn_s_setprio(0);
            auto finish = [&](int tt, const f32x4& c) {
                const int o = tt * 16 + n, oy = o / A1W, ox = o % A1W;
                if (oy < A2W && ox < A2W) {
                    float v[4];
                    if (interior) {                        // block-uniform
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(c[r], 0.0f);
                    } else {
                        const int gy = y0 - 1 + oy, gx = x0 - 1 + ox;
                        const float cap = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? __builtin_inff() : 0.0f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(c[r], 0.0f, cap);
                    }
                    store_split<A2_OCT, A2_TERM>(s_a2, oy * A2W + ox, kb, v);
                }
            };
            if (!VIS_ABLATE(ablate, 4)) {
                finish(t, c0);
                if (two) finish(t + 4, c1);
            }
        }
        if (has_next) commit_entropy();
        __syncthreads();

        // ---- layer 3 (16 -> 8) on output row pairs + the 1x1 conv + sigmoid; wavefront w owns row pairs w and w + 4 ----
        {
            const unsigned char* b0 = s_a2 + ((2 * wave) * A2W + n) * 16;
            const unsigned char* b1 = s_a2 + ((2 * (wave + 4)) * A2W + n) * 16;
            f32x4 c0 = sh2, c1 = sh2;
            bf16x8 x0f[2][3], x1f[2][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                x0f[0][k] = *reinterpret_cast<const bf16x8*>(b0 + off3[0] + k * A2_TERM);
                x1f[0][k] = *reinterpret_cast<const bf16x8*>(b1 + off3[0] + k * A2_TERM);
            }
            if (VIS_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < L3_STEPS; ++s) {
                if (s + 1 < L3_STEPS) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        x0f[(s + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(b0 + off3[s + 1] + k * A2_TERM);
                        x1f[(s + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(b1 + off3[s + 1] + k * A2_TERM);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!VIS_ABLATE(ablate, 8)) {
                    c0 = mfma6(w3[s], x0f[s & 1], c0);
                    c1 = mfma6(w3[s], x1f[s & 1], c1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // D[m = (dy, co)][n = x]: this lane holds dy = kb >> 1, channels (kb & 1) * 4 + r; the other 4 channels sit 16 lanes away
            if (VIS_SETPRIO) __builtin_amdgcn_s_setprio(0);
            auto reduce = [&](const f32x4& c) {
                float part = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(wl[r], fmaxf(c[r], 0.0f), part);
                return part + __shfl_xor(part, 16, 64);      // both channel halves now hold the pixel's sum
            };
            if (!VIS_ABLATE(ablate, 16)) {
                // the even channel-half lanes finish row pair `wave`, the odd ones row pair `wave + 4`: one sigmoid per lane instead of two
                const float p0 = reduce(c0), p1 = reduce(c1);
                const bool second = (kb & 1) != 0;
                const float part = second ? p1 : p0;
                const int gy = y0 + 2 * (second ? wave + 4 : wave) + (kb >> 1), gx = x0 + n;
                if (gy < H && gx < W) weight[(size_t)img * H * W + (size_t)gy * W + gx] = 1.0f / (1.0f + expf(-(part + b3)));
            }
        }
        // (no barrier here: the next tile's layer 1 reads s_in / writes s_a1, both released by the barrier above; its barrier then
        // orders this tile's s_a2 reads before the next layer 2's writes)
    }
}

}  // namespace

extern "C" int mvs_vis_x3_prepare(const float* params, void* prepared, mvs_stream_t stream) {
    MVS_REQUIRE(params && prepared, "mvs_vis_x3_prepare: null pointer");
    constexpr int total = (L2_STEPS + L3_STEPS) * 3 * 64;
    static_assert(total * 16 == MVS_VIS_X3_BYTES, "MVS_VIS_X3_BYTES");
    hipLaunchKernelGGL(vis_x3_prepare_kernel, dim3(mvs::ceil_div(total, 256)), dim3(256), 0, MVS_STREAM(stream), params, static_cast<bf16x8*>(prepared));
    return mvs::finish_launch("mvs_vis_x3_prepare");
}

extern "C" int mvs_vis_x3_fwd(const float* entropy, const float* params, const void* prepared, int N, int H, int W, float* weight,
                              mvs_stream_t stream) {
    MVS_REQUIRE(entropy && params && prepared && weight, "mvs_vis_x3_fwd: null pointer");
    MVS_REQUIRE(N >= 1 && H >= 1 && W >= 1, "mvs_vis_x3_fwd: bad shape N=%d H=%d W=%d", N, H, W);
    const int ntx = mvs::ceil_div(W, T), nty = mvs::ceil_div(H, T);
    MVS_REQUIRE((int64_t)N * ntx * nty < ((int64_t)1 << 31), "mvs_vis_x3_fwd: too many tiles");
    const int ncu = mvs::device_cus();                         // of the CURRENT device (cached per device id)
    const int ntiles = N * ntx * nty;
    const int blocks = ntiles < 2 * ncu ? ntiles : 2 * ncu;       // persistent: two resident blocks per CU
    {   // 71.5 KB of dynamic LDS per block: more than the 64 KB of earlier CDNA parts - gfx950 only, asked for once per device
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(&vis_x3_kernel), LDS_BYTES, "mvs_vis_x3_fwd");
        if (rc != MVS_OK) return rc;
    }
    hipLaunchKernelGGL(vis_x3_kernel, dim3(blocks), dim3(256), LDS_BYTES, MVS_STREAM(stream), entropy, params, static_cast<const bf16x8*>(prepared), N,
                       H, W, ntx, nty, weight, vis_ablate_bits());
    return mvs::finish_launch("mvs_vis_x3_fwd");
}
