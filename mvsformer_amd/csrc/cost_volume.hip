// Fused plane-sweep cost-volume build for fusion_type='cnn' — the inline block of StageNet.forward,
// models/mvsformer_model.py:62-105 — without ever materializing the warped volume [B,C,D,H,W], the repeated
// reference volume, their product or the normalized copies the reference's eval branch makes.
//
// What bounds it: not HBM but the CU's vector-memory front end.  rocprofv3 on the first (NCHW, one dword per
// lane) version showed TCP_TOTAL_CACHE_ACCESSES = (lane loads)/4: the texture addresser retires one 4-lane
// quad per clock whatever the access width, i.e. 16 B/clk/CU for dword gathers but 64 B/clk/CU when every lane
// brings 16 bytes and the 4 lanes of a quad are contiguous.  Hence the layout:
//
//   * features are gathered CHANNEL-LAST ([B,V,H,W,C], one mvs_nchw_to_nhwc pass per stage): a bilinear tap of
//     one pixel is C contiguous floats, fetched as C/4 dwordx4 loads by LPP = C/4 adjacent lanes
//     (lane = pixel*LPP + channel_quad) -> every quad reads 64 contiguous bytes, independent of how coherent
//     neighbouring pixels' sampling positions are;
//   * a wavefront owns PPW = 64/LPP consecutive pixels of one row and ALL their depth hypotheses;
//   * the projective geometry (homography, perspective divide, grid_sample un-normalization, zero-padding tap
//     masks) is evaluated once per (pixel, depth) with lanes = (pixel, depth) pairs — 64 samples per pass — and
//     handed to the gather phase through LDS (2 x ds_read_b128 per lane per sample, broadcast within a pixel);
//   * per-voxel reductions over channels (group sums, the cross-group similarity norms, sum over groups) are
//     DPP/shuffle butterflies over the LPP lanes of a pixel — no LDS, no atomics;
//   * the volume is written once, NCDHW, as the 3-D regularizer reads it.
//
// Sweep A (cv_entropy): per source view, sim[d] = sum_g in_prod[g,d] -> entropy of softmax_d (mvsformer_model.py:88-90).
// Sweep B (cv_aggregate): recomputes the correlation for all source views (cheaper than storing (V-1)
//   [B,G,D,H,W] volumes for the fine stages), accumulates sum_v w_v*in_prod_v in registers, writes
//   volume_mean once; also the eval-only similarity arg-max depth (mvsformer_model.py:81-85,151-158).
//
// Algorithmic HBM bytes per stage: 4*H*W*(V*C + D + G*D)  (features once, hypotheses once, volume once).
//
// Round 4, built / measured / removed: a 4-lanes-per-sample "pair" layout for C = 8 (the two horizontal taps of a row as
// ONE 64-byte quad access; 0.238 -> 0.177 ms of pure gather in tools/probe/gather_probe.hip) lost in the real kernels - sweep A 0.231 ->
// 0.277 ms, sweep B 0.291 -> 0.333 ms at stage 4 - even with sweep A's vector work cut below the generic kernel's (dot before blend).
#include "common.h"
#include "geometry.h"

namespace {

constexpr int G = 8;
constexpr int NW = 4;                         // wavefronts per block
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ f32x4 buf_load4(mvs::rsrc_t r, unsigned voff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, 0, 0));
}

// Cross-lane sums over the LPP lanes of a pixel as DPP row operations (full-rate VALU, no LDS crossbar traffic; the
// ds_bpermute form of __shfl_xor was ~25 LDS-path instructions per gather step in the similarity branch).
//   quad_perm xor 1 / xor 2, then row_half_mirror (i <-> 7-i) and row_mirror (i <-> 15-i): valid as "xor 4 / xor 8"
//   partners because after the quad steps all 4 lanes of a quad already hold the same partial sum.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, o);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;

// sum over the LPP lanes of a pixel (lanes pixel*LPP .. pixel*LPP+LPP-1); every lane gets the total
template <int LPP>
__device__ __forceinline__ float pixel_sum(float v) {
    v = dpp_add<DPP_XOR1>(v);
    if (LPP >= 4) v = dpp_add<DPP_XOR2>(v);
    if (LPP >= 8) v = dpp_add<DPP_HALF_MIRROR>(v);
    if (LPP >= 16) v = dpp_add<DPP_MIRROR>(v);
    return v;
}
// LPP == 16 only: sum over the 8 lanes of a pixel with the same parity (lanes 2k or 2k+1) - row rotations by 4 and 8
// keep the parity, the quad step pairs i with i^2
__device__ __forceinline__ float parity_sum16(float v) {
    v = dpp_add<DPP_XOR2>(v);
    v = dpp_add<DPP_ROR4>(v);
    return dpp_add<DPP_ROR8>(v);
}

// XCD-aware block order.  Blocks are dispatched round-robin over the 8 XCDs (block i -> XCD i % 8) and every XCD has its own
// L2, so the natural (x fastest) order makes each XCD pull EVERY feature map through its own L2 (measured 7.6x the
// algorithmic bytes at stage 1).  The grid is launched 1-D; XCD k takes the k-th contiguous eighth of the logical
// (z, y, x) order - a band of rows of one (batch, view) - so what neighbouring blocks share stays in one L2.
struct BlockId { int x, y, z; bool valid; };
__device__ __forceinline__ BlockId xcd_block(int gx, int gy, int total) {
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    BlockId b;
    b.valid = logical < total;
    b.x = logical % gx;
    b.y = (logical / gx) % gy;
    b.z = logical / (gx * gy);
    return b;
}

// ---------------------------------------------------------------------------------------------------------
// geometry pass: lane l = (p = l % PPW, dd = l / PPW) evaluates sample (pixel x0+p, depth c0+dd) and leaves
// {tap pixel indices, tap weights} in the wavefront's LDS slab at slot l = dd*PPW + p.
// ---------------------------------------------------------------------------------------------------------
template <int PPW, bool FAST>
__device__ __forceinline__ void geometry_pass(const float* __restrict__ rt, const float* __restrict__ depth_row /* + d*HW */,
                                              size_t HW, int c0, int D, int x0, int y, int H, int W, float half_w, float half_h,
                                              int lane, u32x4* taps_o, f32x4* taps_w) {
    const int p = lane % PPW, dd = lane / PPW;
    const int d = min(c0 + dd, D - 1);                    // clamped duplicates are never consumed
    const int x = min(x0 + p, W - 1);
    const float dv = depth_row[(size_t)d * HW + x];
    mvs::Taps t;
    if constexpr (FAST) {
        const float xf = (float)x, yf = (float)y;
        const float rx = fmaf(rt[2], 1.0f, fmaf(rt[1], yf, rt[0] * xf));
        const float ry = fmaf(rt[5], 1.0f, fmaf(rt[4], yf, rt[3] * xf));
        const float rz = fmaf(rt[8], 1.0f, fmaf(rt[7], yf, rt[6] * xf));
        t = mvs::sweep_taps_fast(rt, rx, ry, rz, dv, H, W);
    } else {
        float un, vn, z;
        mvs::sweep_project(rt, (float)x, (float)y, dv, half_w, half_h, &un, &vn, &z);
        t = mvs::sweep_taps(un, vn, H, W, half_w, half_h);
    }
    taps_o[lane] = u32x4{(unsigned)t.o00, (unsigned)t.o01, (unsigned)t.o10, (unsigned)t.o11};
    taps_w[lane] = f32x4{t.w00, t.w01, t.w10, t.w11};
}

// gather phase for one sample: 4 dwordx4 loads (this lane's 4 channels at the 4 taps) + bilinear blend
__device__ __forceinline__ f32x4 gather4(mvs::rsrc_t src, unsigned pix_bytes, unsigned cq_bytes, u32x4 o, f32x4 w) {
    const f32x4 a = buf_load4(src, o[0] * pix_bytes + cq_bytes);
    const f32x4 b = buf_load4(src, o[1] * pix_bytes + cq_bytes);
    const f32x4 c = buf_load4(src, o[2] * pix_bytes + cq_bytes);
    const f32x4 d = buf_load4(src, o[3] * pix_bytes + cq_bytes);
    f32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = a[i] * w[0];
        acc = fmaf(b[i], w[1], acc);
        acc = fmaf(c[i], w[2], acc);
        out[i] = fmaf(d[i], w[3], acc);
    }
    return out;
}

// split form for the software-pipelined sweeps: issue the 4 tap loads now, blend later
__device__ __forceinline__ void load_taps4(mvs::rsrc_t src, unsigned pix_bytes, unsigned cq_bytes, u32x4 o, f32x4 (&t)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = buf_load4(src, o[k] * pix_bytes + cq_bytes);
}
__device__ __forceinline__ f32x4 blend4(const f32x4 (&t)[4], f32x4 w) {
    f32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = t[0][i] * w[0];
        acc = fmaf(t[1][i], w[1], acc);
        acc = fmaf(t[2][i], w[2], acc);
        out[i] = fmaf(t[3][i], w[3], acc);
    }
    return out;
}

// Fine stages (LPP <= 4: few channels, few depths, many pixels) are software-pipelined inside the wavefront: a pass
// issues the tap loads of all its LPP steps, THEN evaluates the next pass's geometry (~130 VALU ops that need no
// memory) while those loads are in flight, then blends/accumulates.  Coarse stages (LPP >= 8) keep one step's loads
// in flight per wavefront and rely on 4 wavefronts per SIMD instead (hoisting all steps there costs 2x the registers
// and measured 1.6x slower).
constexpr bool pipelined(int LPP) { return LPP <= 4; }

// ---------------------------------------------------------------------------------------------------------
// sweep A
// ---------------------------------------------------------------------------------------------------------
template <int LPP, bool FAST>
__global__ __launch_bounds__(64 * NW) void cv_entropy_kernel(const float* __restrict__ feat /*[B,V,H,W,C]*/,
                                                             const float* __restrict__ rt_all, const float* __restrict__ depth,
                                                             int V, int D, int H, int W, float* __restrict__ entropy, int gx, int total) {
    constexpr int C = 4 * LPP, CPG = C / G, PPW = 64 / LPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per wavefront: 2 x 64 tap slots (double buffer for the pipelined schedule), then the sims[D][PPW] slab
    u32x4* taps_o = reinterpret_cast<u32x4*>(smem) + wave * 128;
    f32x4* taps_w = reinterpret_cast<f32x4*>(smem + NW * 128 * 16) + wave * 128;
    float* sims = reinterpret_cast<float*>(smem + NW * 128 * 32) + (size_t)wave * PPW * D;     // [D][PPW]

    const BlockId bid = xcd_block(gx, H, total);
    if (!bid.valid) return;
    const int x0 = (bid.x * NW + wave) * PPW, y = bid.y;
    const int b = bid.z / (V - 1), sv = bid.z % (V - 1);
    if (x0 >= W) return;                                   // wave-uniform; no block-wide barrier is used below
    const size_t HW = (size_t)H * W;
    const unsigned pix_bytes = C * 4u;
    const int pg = lane / LPP, cq = lane % LPP;
    const int xg = min(x0 + pg, W - 1);
    const f32x4 r = *reinterpret_cast<const f32x4*>(feat + ((size_t)(b * V) * HW + (size_t)y * W + xg) * C + cq * 4);
    const mvs::rsrc_t src = mvs::make_rsrc(feat + (size_t)(b * V + sv + 1) * HW * C, (unsigned)(HW * pix_bytes));
    const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
    const float* depth_row = depth + (size_t)b * D * HW + (size_t)y * W;
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);

    // one gather step; branch-free so that the unrolled steps of a chunk form ONE basic block and the compiler can
    // keep several steps' buffer loads in flight (with a per-step `if` every step waited for its own 4 loads)
    auto step = [&](int c0, int dd, bool valid) {
        const u32x4 o = taps_o[dd * PPW + pg];
        const f32x4 w = taps_w[dd * PPW + pg];
        const f32x4 g4 = gather4(src, pix_bytes, cq * 16u, o, w);
        float s = r[0] * g4[0];
        s = s + r[1] * g4[1];
        s = s + r[2] * g4[2];
        s = s + r[3] * g4[3];
        s = pixel_sum<LPP>(s) * (1.0f / CPG);
        if (valid && cq == 0) sims[(c0 + dd) * PPW + pg] = s;
    };
    if constexpr (pipelined(LPP)) {
        geometry_pass<PPW, FAST>(rt, depth_row, HW, 0, D, x0, y, H, W, half_w, half_h, lane, taps_o, taps_w);
        int buf = 0;
        for (int c0 = 0; c0 < D; c0 += LPP, buf ^= 1) {
            __builtin_amdgcn_wave_barrier();
            f32x4 w[LPP], t[LPP][4];
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) {
                w[dd] = taps_w[buf * 64 + dd * PPW + pg];
                load_taps4(src, pix_bytes, cq * 16u, taps_o[buf * 64 + dd * PPW + pg], t[dd]);
            }
            if (c0 + LPP < D)
                geometry_pass<PPW, FAST>(rt, depth_row, HW, c0 + LPP, D, x0, y, H, W, half_w, half_h, lane, taps_o + (buf ^ 1) * 64,
                                   taps_w + (buf ^ 1) * 64);
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) {
                const f32x4 g4 = blend4(t[dd], w[dd]);
                float s = r[0] * g4[0];
                s = s + r[1] * g4[1];
                s = s + r[2] * g4[2];
                s = s + r[3] * g4[3];
                s = pixel_sum<LPP>(s) * (1.0f / CPG);
                if (c0 + dd < D && cq == 0) sims[(c0 + dd) * PPW + pg] = s;
            }
        }
        __builtin_amdgcn_wave_barrier();
    } else
    for (int c0 = 0; c0 < D; c0 += LPP) {
        geometry_pass<PPW, FAST>(rt, depth_row, HW, c0, D, x0, y, H, W, half_w, half_h, lane, taps_o, taps_w);
        __builtin_amdgcn_wave_barrier();
        if (c0 + LPP <= D) {
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) step(c0, dd, true);
        } else {
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) step(c0, dd, c0 + dd < D);   // clamped duplicate samples, results dropped
        }
        __builtin_amdgcn_wave_barrier();
    }
    // entropy of softmax_d: lane l = (p = l % PPW, k = l / PPW) takes depths k, k+LPP, ...; reductions over k are
    // butterflies over lane strides PPW, 2*PPW, ... 32
    {
        const int p = lane % PPW, k = lane / PPW;
        float m = -INFINITY;
        for (int d = k; d < D; d += LPP) m = fmaxf(m, sims[d * PPW + p]);
#pragma unroll
        for (int s = PPW; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
        // FAST: hardware exp2 / log2 / reciprocal (1 ulp each) instead of the library sequences - the entropy moves by ~1e-6
        float sum = 0.0f;
        for (int d = k; d < D; d += LPP) sum += FAST ? __expf(sims[d * PPW + p] - m) : expf(sims[d * PPW + p] - m);
#pragma unroll
        for (int s = PPW; s < 64; s <<= 1) sum += __shfl_xor(sum, s, 64);
        const float inv_sum = __builtin_amdgcn_rcpf(sum);
        float ent = 0.0f;
        for (int d = k; d < D; d += LPP) {
            const float pr = FAST ? __expf(sims[d * PPW + p] - m) * inv_sum : expf(sims[d * PPW + p] - m) / sum;
            ent = ent + (-pr) * (FAST ? __logf(pr + 1e-7f) : logf(pr + 1e-7f));
        }
#pragma unroll
        for (int s = PPW; s < 64; s <<= 1) ent += __shfl_xor(ent, s, 64);
        if (k == 0 && x0 + p < W) entropy[((size_t)(b * (V - 1) + sv) * H + y) * W + x0 + p] = ent;
    }
}

// ---------------------------------------------------------------------------------------------------------
// sweep B
// ---------------------------------------------------------------------------------------------------------
template <int LPP, bool SIM, bool FAST>
__global__ __launch_bounds__(64 * NW) void cv_aggregate_kernel(const float* __restrict__ feat, const float* __restrict__ rt_all,
                                                               const float* __restrict__ depth, const float* __restrict__ weight,
                                                               int V, int D, int H, int W, float* __restrict__ volume,
                                                               float* __restrict__ sim_depth, int gx, int total, __bf16* __restrict__ vol16) {
    // vol16 (optional): the same volume ALSO as bf16 channel-last [B,D,H,W,G] - what the bf16 regularizer of the training path reads (the
    // separate fp32 NCDHW -> bf16 NDHWC pass then never runs)
    constexpr int C = 4 * LPP, CPG = C / G, PPW = 64 / LPP;
    constexpr int NG = (CPG >= 4) ? 1 : 4 / CPG;          // correlation groups whose sums live in this lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4* taps_o = reinterpret_cast<u32x4*>(smem) + wave * 128;
    f32x4* taps_w = reinterpret_cast<f32x4*>(smem + NW * 128 * 16) + wave * 128;

    const BlockId bid = xcd_block(gx, H, total);
    if (!bid.valid) return;
    const int x0 = (bid.x * NW + wave) * PPW, y = bid.y, b = bid.z;
    if (x0 >= W) return;
    const size_t HW = (size_t)H * W;
    const unsigned pix_bytes = C * 4u;
    const int pg = lane / LPP, cq = lane % LPP;
    const bool active = x0 + pg < W;
    const int xg = min(x0 + pg, W - 1);
    const size_t pix = (size_t)y * W + xg;
    const f32x4 r = *reinterpret_cast<const f32x4*>(feat + ((size_t)(b * V) * HW + pix) * C + cq * 4);
    const float* depth_row = depth + (size_t)b * D * HW + (size_t)y * W;
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);

    // F.normalize(ref_volume, dim=1): per channel-in-group index j, L2 norm over the 8 groups
    f32x4 rn = {0.f, 0.f, 0.f, 0.f};                      // this lane's 4 reference channels, normalized
    if (SIM) {
        const f32x4 sq = {r[0] * r[0], r[1] * r[1], r[2] * r[2], r[3] * r[3]};
        f32x4 n2 = sq;
        if (CPG < 4) {                                    // several groups inside the lane share j = i % CPG
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = 0.0f;
#pragma unroll
                for (int k = i % CPG; k < 4; k += CPG) t += sq[k];
                n2[i] = t;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) n2[i] = (CPG == 8) ? parity_sum16(n2[i]) : pixel_sum<LPP>(n2[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) rn[i] = r[i] / fmaxf(sqrtf(n2[i]), 1e-12f);
    }
    const float* wp = weight + (size_t)(b * (V - 1)) * HW + pix;
    float wsum = 0.0f;
    for (int sv = 0; sv < V - 1; ++sv) wsum = wsum + wp[(size_t)sv * HW];
    const float denom = wsum + 1e-6f;

    float best = -INFINITY;
    int besti = 0;
    if constexpr (pipelined(LPP)) {
        // flattened (depth chunk, source view) passes, geometry of pass i+1 overlapped with the loads of pass i
        const int nviews = V - 1, npass = ((D + LPP - 1) / LPP) * nviews;
        float acc[LPP][NG];
        float simtot[LPP];
        geometry_pass<PPW, FAST>(rt_all + (size_t)(b * nviews) * 12, depth_row, HW, 0, D, x0, y, H, W, half_w, half_h, lane, taps_o, taps_w);
        int c0 = 0, sv = 0, buf = 0;
        for (int i = 0; i < npass; ++i, buf ^= 1) {
            if (sv == 0) {
#pragma unroll
                for (int dd = 0; dd < LPP; ++dd) {
                    simtot[dd] = 0.0f;
#pragma unroll
                    for (int k = 0; k < NG; ++k) acc[dd][k] = 0.0f;
                }
            }
            const mvs::rsrc_t src = mvs::make_rsrc(feat + (size_t)(b * V + sv + 1) * HW * C, (unsigned)(HW * pix_bytes));
            const float wv = wp[(size_t)sv * HW];
            __builtin_amdgcn_wave_barrier();
            f32x4 w[LPP], t[LPP][4];
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) {
                w[dd] = taps_w[buf * 64 + dd * PPW + pg];
                load_taps4(src, pix_bytes, cq * 16u, taps_o[buf * 64 + dd * PPW + pg], t[dd]);
            }
            const int nsv = (sv + 1 == nviews) ? 0 : sv + 1, nc0 = (sv + 1 == nviews) ? c0 + LPP : c0;
            if (i + 1 < npass)
                geometry_pass<PPW, FAST>(rt_all + (size_t)(b * nviews + nsv) * 12, depth_row, HW, nc0, D, x0, y, H, W, half_w, half_h, lane,
                                   taps_o + (buf ^ 1) * 64, taps_w + (buf ^ 1) * 64);
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) {
                const f32x4 g4 = blend4(t[dd], w[dd]);
                const f32x4 p = {r[0] * g4[0], r[1] * g4[1], r[2] * g4[2], r[3] * g4[3]};
                if (CPG == 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[dd][k] = acc[dd][k] + p[k] * wv;
                } else {                                      // CPG == 2 (LPP == 4)
                    acc[dd][0] = acc[dd][0] + ((p[0] + p[1]) * 0.5f) * wv;
                    acc[dd][1] = acc[dd][1] + ((p[2] + p[3]) * 0.5f) * wv;
                }
                if (SIM) {
                    f32x4 q = {rn[0] * g4[0], rn[1] * g4[1], rn[2] * g4[2], rn[3] * g4[3]};
                    f32x4 n2 = {g4[0] * g4[0], g4[1] * g4[1], g4[2] * g4[2], g4[3] * g4[3]};
#pragma unroll
                    for (int ii = 0; ii < CPG; ++ii) {
#pragma unroll
                        for (int k = ii + CPG; k < 4; k += CPG) { q[ii] += q[k]; n2[ii] += n2[k]; }
                    }
#pragma unroll
                    for (int ii = 0; ii < CPG; ++ii) {
                        q[ii] = pixel_sum<LPP>(q[ii]);
                        n2[ii] = pixel_sum<LPP>(n2[ii]);
                    }
                    float ssum = 0.0f;
#pragma unroll
                    for (int ii = 0; ii < CPG; ++ii) ssum += q[ii] * __builtin_amdgcn_rsqf(fmaxf(n2[ii], 1e-24f));
                    simtot[dd] = simtot[dd] + ssum * (1.0f / CPG);
                }
            }
            if (sv + 1 == nviews) {
#pragma unroll
                for (int dd = 0; dd < LPP; ++dd) {
                    const int d = c0 + dd;
                    if (d < D) {
                        if (active) {
#pragma unroll
                            for (int k = 0; k < NG; ++k) {
                                const float v = acc[dd][k] / denom;
                                volume[((size_t)(b * G + cq * NG + k) * D + d) * HW + pix] = v;
                                if (vol16) vol16[(((size_t)b * D + d) * HW + pix) * G + cq * NG + k] = (__bf16)v;
                            }
                        }
                        if (SIM && simtot[dd] > best) { best = simtot[dd]; besti = d; }
                    }
                }
            }
            sv = nsv;
            c0 = nc0;
        }
    } else
    for (int c0 = 0; c0 < D; c0 += LPP) {
        float acc[LPP][NG];
        float simtot[LPP];
#pragma unroll
        for (int dd = 0; dd < LPP; ++dd) {
            simtot[dd] = 0.0f;
#pragma unroll
            for (int k = 0; k < NG; ++k) acc[dd][k] = 0.0f;
        }
        for (int sv = 0; sv < V - 1; ++sv) {
            const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
            const mvs::rsrc_t src = mvs::make_rsrc(feat + (size_t)(b * V + sv + 1) * HW * C, (unsigned)(HW * pix_bytes));
            const float wv = wp[(size_t)sv * HW];
            __builtin_amdgcn_wave_barrier();
            geometry_pass<PPW, FAST>(rt, depth_row, HW, c0, D, x0, y, H, W, half_w, half_h, lane, taps_o, taps_w);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) {
                // per-step branch on purpose: a branch-free body lets hipcc hoist all 16 steps' loads (208 VGPRs at
                // LPP=16, 2 waves/SIMD) and measured 1.6x SLOWER than 4 waves/SIMD with one step's loads in flight each
                if (c0 + dd < D) {
                    const u32x4 o = taps_o[dd * PPW + pg];
                    const f32x4 w = taps_w[dd * PPW + pg];
                    const f32x4 g4 = gather4(src, pix_bytes, cq * 16u, o, w);
                    const f32x4 p = {r[0] * g4[0], r[1] * g4[1], r[2] * g4[2], r[3] * g4[3]};
                    // group means held by this lane
                    if (CPG == 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[dd][k] = acc[dd][k] + p[k] * wv;
                    } else if (CPG == 2) {
                        acc[dd][0] = acc[dd][0] + ((p[0] + p[1]) * 0.5f) * wv;
                        acc[dd][1] = acc[dd][1] + ((p[2] + p[3]) * 0.5f) * wv;
                    } else if (CPG == 4) {
                        acc[dd][0] = acc[dd][0] + ((((p[0] + p[1]) + p[2]) + p[3]) * 0.25f) * wv;
                    } else {
                        float h = ((p[0] + p[1]) + p[2]) + p[3];
                        h = dpp_add<DPP_XOR1>(h);             // the other half of the group lives in the neighbour lane
                        acc[dd][0] = acc[dd][0] + (h * 0.125f) * wv;
                    }
                    if (SIM) {
                        // similarity: sum_j (sum_g refn[g,j]*warp[g,j]) / max(||warp[:,j]||, eps), mean over j
                        f32x4 q = {rn[0] * g4[0], rn[1] * g4[1], rn[2] * g4[2], rn[3] * g4[3]};
                        f32x4 n2 = {g4[0] * g4[0], g4[1] * g4[1], g4[2] * g4[2], g4[3] * g4[3]};
                        if (CPG < 4) {
#pragma unroll
                            for (int i = 0; i < CPG; ++i) {
#pragma unroll
                                for (int k = i + CPG; k < 4; k += CPG) { q[i] += q[k]; n2[i] += n2[k]; }
                            }
                        }
                        constexpr int NJ = (CPG < 4) ? CPG : 4;      // distinct j held by this lane
#pragma unroll
                        for (int i = 0; i < NJ; ++i) {
                            q[i] = (CPG == 8) ? parity_sum16(q[i]) : pixel_sum<LPP>(q[i]);
                            n2[i] = (CPG == 8) ? parity_sum16(n2[i]) : pixel_sum<LPP>(n2[i]);
                        }
                        float s = 0.0f;
#pragma unroll
                        for (int i = 0; i < NJ; ++i) s += q[i] * __builtin_amdgcn_rsqf(fmaxf(n2[i], 1e-24f));   // q / max(||w||, 1e-12)
                        if (CPG == 8) s = dpp_add<DPP_XOR1>(s);       // the other 4 j's live in the neighbour lane
                        simtot[dd] = simtot[dd] + s * (1.0f / CPG);
                    }
                }
            }
        }
        // write volume_mean for this depth chunk, track the similarity arg-max
#pragma unroll
        for (int dd = 0; dd < LPP; ++dd) {
            const int d = c0 + dd;
            if (d < D) {
                if (active && (CPG < 8 || (cq & 1) == 0)) {
#pragma unroll
                    for (int k = 0; k < NG; ++k) {
                        const int g = (CPG == 8) ? (cq >> 1) : cq * NG + k;
                        const float v = acc[dd][k] / denom;
                        volume[((size_t)(b * G + g) * D + d) * HW + pix] = v;
                        if (vol16) vol16[(((size_t)b * D + d) * HW + pix) * G + g] = (__bf16)v;
                    }
                }
                if (SIM && simtot[dd] > best) { best = simtot[dd]; besti = d; }
            }
        }
    }
    if (SIM && active && cq == 0) sim_depth[(size_t)b * HW + pix] = depth_row[(size_t)besti * HW + xg];
}

// ---------------------------------------------------------------------------------------------------------
// Stored-correlation form of the two sweeps for the COARSE stages (C >= 32: few pixels, many channels and planes).
//
// There the (V-1) per-view correlation volumes [G,D,H,W] are small enough to stay in the 256 MB Infinity Cache (113 MB at config-2
// stage 1, 226 MB at stage 2), so sweep A' keeps them (and the per-view similarity it computes on the way) and sweep B' becomes a pure
// stream: no second gather sweep, no second geometry pass.  cv_corr_kernel = cv_entropy_kernel + the group means and the eval
// similarity of cv_aggregate_kernel, operation for operation, so entropy and volume come out BIT-IDENTICAL to the recomputing pair.
//
// Store layout (private to this pair): pixel-group tiles, one per wavefront of sweep A' (TPX = 256/C consecutive pixels of a row):
//   corr [B*(V-1)][H][XG][D][G][TPX]    simv [B*(V-1)][H][XG][D][TPX]      XG = ceil(W / TPX)
// a wavefront's output for a chunk of LPP planes is one contiguous 2 KB run (staged through LDS, written as 16-byte stores).
//
// The similarity's sums over the 8 groups (an 8-lane all-reduce of 8 values per lane: 24 DPP additions, half-rate instructions, in
// cv_aggregate_kernel) go through LDS here: every lane drops its 8 partial values, lane k of a pixel picks up the 8 partials of
// value k (two ds_read_b128) and adds them - 7 full-rate additions instead of 24 half-rate ones.
// ---------------------------------------------------------------------------------------------------------
template <int LPP>
struct CorrCfg {
    static constexpr int C = 4 * LPP, CPG = C / G, PPW = 64 / LPP;
    static constexpr int UNITS = PPW * (LPP / 8);          // all-reduce units per wavefront: (pixel) or (pixel, lane parity)
    static constexpr int RED_FLOATS = UNITS * 72;          // unit stride 72 floats: 64 used, padded so a 32-lane store group hits 32 banks
    static constexpr int STAGE_FLOATS = LPP * G * PPW;     // one chunk of group means (512 floats)
    static constexpr int SSTAGE_FLOATS = LPP * PPW;        // one chunk of similarities (64 floats)
    static constexpr size_t lds_bytes(int D) { return (size_t)NW * (64 * 32 + (PPW * D + RED_FLOATS + STAGE_FLOATS + SSTAGE_FLOATS) * sizeof(float)); }
};

template <int LPP, bool FAST>
__global__ __launch_bounds__(64 * NW) void cv_corr_kernel(const float* __restrict__ feat /*[B,V,H,W,C]*/, const float* __restrict__ rt_all,
                                                          const float* __restrict__ depth, int V, int D, int H, int W, float* __restrict__ entropy,
                                                          float* __restrict__ corr, float* __restrict__ simv, int gx, int total, int y0, int Hs) {
    // (y0, Hs): the band of reference rows this launch covers - image rows y0 .. y0 + Hs - 1; entropy and store are band-local ([..][Hs][..]),
    // features / hypotheses are the whole image.  y0 = 0, Hs = H is the whole image.
    using Cfg = CorrCfg<LPP>;
    constexpr int C = Cfg::C, CPG = Cfg::CPG, PPW = Cfg::PPW;
    static_assert(LPP == 8 || LPP == 16, "stored-correlation sweeps are built for C = 32 and C = 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4* taps_o = reinterpret_cast<u32x4*>(smem) + wave * 64;
    f32x4* taps_w = reinterpret_cast<f32x4*>(smem + NW * 64 * 16) + wave * 64;
    float* fbase = reinterpret_cast<float*>(smem + NW * 64 * 32);
    float* sims = fbase + (size_t)wave * PPW * D;                                            // [D][PPW]
    float* red = fbase + (size_t)NW * PPW * D + wave * Cfg::RED_FLOATS;
    float* stage = fbase + (size_t)NW * (PPW * D + Cfg::RED_FLOATS) + wave * Cfg::STAGE_FLOATS;   // [LPP][G][PPW]
    float* sstage = fbase + (size_t)NW * (PPW * D + Cfg::RED_FLOATS + Cfg::STAGE_FLOATS) + wave * Cfg::SSTAGE_FLOATS;   // [LPP][PPW]

    const BlockId bid = xcd_block(gx, Hs, total);
    if (!bid.valid) return;
    const int xgi = bid.x * NW + wave;                      // pixel group of this wavefront
    const int x0 = xgi * PPW, ys = bid.y, y = y0 + ys;
    const int b = bid.z / (V - 1), sv = bid.z % (V - 1);
    if (x0 >= W) return;                                   // wave-uniform; only wavefront-level barriers below
    const int XG = (W + PPW - 1) / PPW;
    const size_t HW = (size_t)H * W;
    const unsigned pix_bytes = C * 4u;
    const int pg = lane / LPP, cq = lane % LPP;
    const int xg = min(x0 + pg, W - 1);
    const f32x4 r = *reinterpret_cast<const f32x4*>(feat + ((size_t)(b * V) * HW + (size_t)y * W + xg) * C + cq * 4);
    const mvs::rsrc_t src = mvs::make_rsrc(feat + (size_t)(b * V + sv + 1) * HW * C, (unsigned)(HW * pix_bytes));
    const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
    const float* depth_row = depth + (size_t)b * D * HW + (size_t)y * W;
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
    const size_t tile = ((size_t)(b * (V - 1) + sv) * Hs + ys) * XG + xgi;
    float* ctile = corr + tile * ((size_t)D * G * PPW);
    float* stile = simv + tile * ((size_t)D * PPW);

    // F.normalize(ref_volume, dim=1), exactly as in cv_aggregate_kernel
    f32x4 rn;
    {
        f32x4 n2 = {r[0] * r[0], r[1] * r[1], r[2] * r[2], r[3] * r[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) n2[i] = (CPG == 8) ? parity_sum16(n2[i]) : pixel_sum<LPP>(n2[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) rn[i] = r[i] / fmaxf(sqrtf(n2[i]), 1e-12f);
    }
    // all-reduce addressing.  Writer: this lane's values for j = 4*par + i go to unit (pixel, par), slot k' = 2*i + t (t = 0: q, 1: ||w||^2),
    // position gi (its group).  Reader: lane (pixel, cq) owns unit par_r = cq / 8 (C = 64) and slot k' = cq % 8.
    const int par_w = (LPP == 16) ? (cq & 1) : 0, gi = (LPP == 16) ? (cq >> 1) : cq;
    float* red_w = red + (pg * (LPP / 8) + par_w) * 72 + gi;
    const float* red_r = red + (pg * (LPP / 8) + (cq >> 3)) * 72 + (cq & 7) * 8;

    for (int c0 = 0; c0 < D; c0 += LPP) {
        geometry_pass<PPW, FAST>(rt, depth_row, HW, c0, D, x0, y, H, W, half_w, half_h, lane, taps_o, taps_w);
        __builtin_amdgcn_wave_barrier();
        // Software pipeline of depth 2 over the chunk's steps: step dd+1's four tap loads are issued before step dd's blend, reductions
        // and LDS all-reduce (a chain of ~300 dependent cycles that would otherwise sit between two gathers of the same wavefront).
        // Straight-line code on purpose: with wave-uniform branches around the steps hipcc's wait-count pass merges the two paths
        // conservatively and waits for the prefetch it has just issued.  The slots of planes >= D hold clamped duplicates (valid
        // addresses, geometry_pass), so every step runs; only the stores are predicated.  sched_barrier pins the issue order (left
        // alone, the scheduler hoists all 16 steps' loads: 2.5x the registers).  Ping-pong register sets by the step's parity.
        f32x4 tp[2][4], wp2[2];
        wp2[0] = taps_w[pg];
        load_taps4(src, pix_bytes, cq * 16u, taps_o[pg], tp[0]);
#pragma unroll
        for (int dd = 0; dd < LPP; ++dd) {
            if (dd + 1 < LPP) {
                wp2[(dd + 1) & 1] = taps_w[(dd + 1) * PPW + pg];
                load_taps4(src, pix_bytes, cq * 16u, taps_o[(dd + 1) * PPW + pg], tp[(dd + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const bool valid = c0 + dd < D;
                const f32x4 g4 = blend4(tp[dd & 1], wp2[dd & 1]);
                const f32x4 p = {r[0] * g4[0], r[1] * g4[1], r[2] * g4[2], r[3] * g4[3]};
                const float h = ((p[0] + p[1]) + p[2]) + p[3];
                // sim_vol = sum over groups (cv_entropy_kernel's expression) and this lane's group mean (cv_aggregate_kernel's)
                const float s = pixel_sum<LPP>(h) * (1.0f / CPG);
                if (valid && cq == 0) sims[(c0 + dd) * PPW + pg] = s;
                if (CPG == 4) {
                    stage[(dd * G + cq) * PPW + pg] = h * 0.25f;
                } else {
                    const float h2 = dpp_add<DPP_XOR1>(h);
                    if ((cq & 1) == 0) stage[(dd * G + (cq >> 1)) * PPW + pg] = h2 * 0.125f;
                }
                // eval similarity: sum_j (sum_g refn[g,j]*warp[g,j]) / max(||warp[:,j]||, 1e-12), mean over j
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    red_w[(2 * i) * 8] = rn[i] * g4[i];
                    red_w[(2 * i + 1) * 8] = g4[i] * g4[i];
                }
                __builtin_amdgcn_wave_barrier();
                const f32x4 ra = *reinterpret_cast<const f32x4*>(red_r), rb = *reinterpret_cast<const f32x4*>(red_r + 4);
                __builtin_amdgcn_wave_barrier();
                const float tot = ((((((ra[0] + ra[1]) + ra[2]) + ra[3]) + rb[0]) + rb[1]) + rb[2]) + rb[3];
                // even lanes hold q_j, their odd neighbours ||w_j||^2
                const float nrm = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tot), DPP_XOR1, 0xF, 0xF, true));
                float term = (cq & 1) ? 0.0f : tot * __builtin_amdgcn_rsqf(fmaxf(nrm, 1e-24f));
                term = pixel_sum<LPP>(term);
                if (cq == 0) sstage[dd * PPW + pg] = term * (1.0f / CPG);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();
        // this chunk's 2 KB of group means and 256 B of similarities: contiguous in the tile
        {
            const int nd = min(LPP, D - c0);
#pragma unroll
            for (int it = 0; it < Cfg::STAGE_FLOATS / 256; ++it) {
                const int f = it * 64 + lane;                                   // float4 index inside the chunk
                if (f * 4 < nd * G * PPW)
                    *reinterpret_cast<f32x4*>(ctile + (size_t)c0 * G * PPW + f * 4) = *reinterpret_cast<const f32x4*>(stage + f * 4);
            }
            if (lane * 4 < nd * PPW) *reinterpret_cast<f32x4*>(stile + (size_t)c0 * PPW + lane * 4) = *reinterpret_cast<const f32x4*>(sstage + lane * 4);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // entropy of softmax_d - cv_entropy_kernel's epilogue
    {
        const int p = lane % PPW, k = lane / PPW;
        float m = -INFINITY;
        for (int d = k; d < D; d += LPP) m = fmaxf(m, sims[d * PPW + p]);
#pragma unroll
        for (int s = PPW; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
        float sum = 0.0f;
        for (int d = k; d < D; d += LPP) sum += FAST ? __expf(sims[d * PPW + p] - m) : expf(sims[d * PPW + p] - m);
#pragma unroll
        for (int s = PPW; s < 64; s <<= 1) sum += __shfl_xor(sum, s, 64);
        const float inv_sum = __builtin_amdgcn_rcpf(sum);
        float ent = 0.0f;
        for (int d = k; d < D; d += LPP) {
            const float pr = FAST ? __expf(sims[d * PPW + p] - m) * inv_sum : expf(sims[d * PPW + p] - m) / sum;
            ent = ent + (-pr) * (FAST ? __logf(pr + 1e-7f) : logf(pr + 1e-7f));
        }
#pragma unroll
        for (int s = PPW; s < 64; s <<= 1) ent += __shfl_xor(ent, s, 64);
        if (k == 0 && x0 + p < W) entropy[((size_t)(b * (V - 1) + sv) * Hs + ys) * W + x0 + p] = ent;
    }
}

// sweep B' over the stored correlation: volume_mean = sum_v w_v*corr_v / (sum_v w_v + 1e-6) (mvsformer_model.py:101-105) in
// cv_aggregate_kernel's order of operations, and sim_depth = depth[argmax_d sum_v similarity_v] (mvsformer_model.py:151-158).
// One wavefront per pixel group; every load is a contiguous 1 KB run.
template <int TPX>
__global__ __launch_bounds__(64 * NW) void cv_merge_kernel(const float* __restrict__ corr, const float* __restrict__ simv, const float* __restrict__ depth,
                                                           const float* __restrict__ weight, int V, int D, int H, int W, float* __restrict__ volume,
                                                           float* __restrict__ sim_depth, int gx, int total, int y0, int Hs, int r_lo, int nrows) {
    // store and weight are band-local ([..][Hs][..], band row 0 = image row y0); this launch merges band rows r_lo .. r_lo + nrows - 1 (the
    // band without its halo) into the whole-image volume / sim_depth.  y0 = 0, Hs = H, r_lo = 0, nrows = H is the whole image.
    constexpr int LPR = TPX / 4;                          // lanes (float4s) per (plane, group) row
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const BlockId bid = xcd_block(gx, nrows, total);
    if (!bid.valid) return;
    const int XG = (W + TPX - 1) / TPX;
    const int xgi = bid.x * NW + wave, ys = r_lo + bid.y, y = y0 + ys, b = bid.z;
    if (xgi >= XG) return;
    const int x0 = xgi * TPX;
    const size_t HW = (size_t)H * W;
    const int nv = V - 1;
    const size_t vstride_c = (size_t)Hs * XG * D * G * TPX, vstride_s = (size_t)Hs * XG * D * TPX, HWs = (size_t)Hs * W;
    const float* ct = corr + (size_t)(b * nv) * vstride_c + ((size_t)ys * XG + xgi) * ((size_t)D * G * TPX);
    const float* stl = simv + (size_t)(b * nv) * vstride_s + ((size_t)ys * XG + xgi) * ((size_t)D * TPX);
    const float* wp = weight + (size_t)(b * nv) * HWs + (size_t)ys * W;
    {
        const int part = lane % LPR, xs = x0 + part * 4;
        f32x4 wsum = {0.f, 0.f, 0.f, 0.f};
        for (int v = 0; v < nv; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) wsum[i] = wsum[i] + wp[(size_t)v * HWs + min(xs + i, W - 1)];
        f32x4 denom;
#pragma unroll
        for (int i = 0; i < 4; ++i) denom[i] = wsum[i] + 1e-6f;
        const bool vec = ((W & 3) == 0) && xs < W;          // whole float4 inside the row, 16-byte aligned
        const int rows = D * G;
        for (int row = lane / LPR; row < rows; row += 64 / LPR) {
            const int f = row * LPR + part;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int v = 0; v < nv; ++v) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(ct + (size_t)v * vstride_c + (size_t)f * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = acc[i] + c4[i] * wp[(size_t)v * HWs + min(xs + i, W - 1)];
            }
            const int d = row / G, g = row % G;
            float* o = volume + ((size_t)(b * G + g) * D + d) * HW + (size_t)y * W + xs;
            if (vec) {
                *reinterpret_cast<f32x4*>(o) = f32x4{acc[0] / denom[0], acc[1] / denom[1], acc[2] / denom[2], acc[3] / denom[3]};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xs + i < W) o[i] = acc[i] / denom[i];
            }
        }
    }
    if (sim_depth) {
        const int px = lane % TPX, ds = lane / TPX;
        float best = -INFINITY;
        int besti = 0;
        for (int d = ds; d < D; d += 64 / TPX) {
            float s = 0.0f;
            for (int v = 0; v < nv; ++v) s = s + stl[(size_t)v * vstride_s + (size_t)d * TPX + px];
            if (s > best) { best = s; besti = d; }
        }
#pragma unroll
        for (int m = TPX; m < 64; m <<= 1) {                // first maximum wins, as a sequential scan over d would have it
            const float ob = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(besti, m, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (ds == 0 && x0 + px < W) sim_depth[(size_t)b * HW + (size_t)y * W + x0 + px] = depth[((size_t)b * D + besti) * HW + (size_t)y * W + x0 + px];
    }
}

// ---------------------------------------------------------------------------------------------------------
// NCHW -> NHWC feature transpose ([N,C,HW] -> [N,HW,C]) through an LDS tile of TP pixels: reads are 16 bytes per lane along a
// channel row (a wavefront reads 1 KB contiguous), writes are one contiguous TP*C-float run per block (dwordx4 per lane).  VEC = false
// (HW not a multiple of 4, or a ragged last block) falls back to dword reads.
// ---------------------------------------------------------------------------------------------------------
template <int C, int TP>
__device__ __forceinline__ void nchw_to_nhwc_tile(const float* __restrict__ in, float* __restrict__ out, size_t HW, int vec_ok, size_t p0,
                                                  size_t n, float* __restrict__ tile) {
    constexpr int LD = TP + 4;                               // row stride: 16-byte aligned rows, rows 4 banks apart
    const int tid = threadIdx.x;
    const int npix = (int)min((size_t)TP, HW - p0);
    if (vec_ok && npix == TP) {
        for (int i = tid; i < C * (TP / 4); i += 256) {
            const int c = i / (TP / 4), q = i % (TP / 4);
            *reinterpret_cast<f32x4*>(tile + c * LD + 4 * q) = *reinterpret_cast<const f32x4*>(in + (n * C + c) * HW + p0 + 4 * q);
        }
    } else {
        for (int i = tid; i < C * TP; i += 256) {
            const int c = i / TP, p = i % TP;
            tile[c * LD + p] = (p < npix) ? in[(n * C + c) * HW + p0 + p] : 0.0f;
        }
    }
    __syncthreads();
    float* o = out + (n * HW + p0) * C;
    for (int i = tid; i < npix * C / 4; i += 256) {
        const int p = (i * 4) / C, c = (i * 4) % C;
        const f32x4 v = {tile[c * LD + p], tile[(c + 1) * LD + p], tile[(c + 2) * LD + p], tile[(c + 3) * LD + p]};
        *reinterpret_cast<f32x4*>(o + (size_t)i * 4) = v;
    }
}

template <int C, int TP>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, size_t HW, int vec_ok) {
    __shared__ __attribute__((aligned(16))) float tile[C * (TP + 4)];
    nchw_to_nhwc_tile<C, TP>(in, out, HW, vec_ok, (size_t)blockIdx.x * TP, blockIdx.y, tile);
}

// up to four transposes (the four stages' feature maps of one cascade) in ONE launch: job j owns blocks [start[j], start[j+1])
struct TransposeJobs {
    const float* in[4];
    float* out[4];
    long long HW[4];
    int C[4], vec[4], start[5], n;
};
__device__ __forceinline__ int tp_of(int C) { return C <= 16 ? 256 : (C == 32 ? 128 : 64); }
__global__ __launch_bounds__(256) void nchw_to_nhwc_multi_kernel(const TransposeJobs jobs) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 68];          // the largest of the four tile shapes (17 KB)
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.start[j + 1]) ++j;
    const int local = (int)blockIdx.x - jobs.start[j], C = jobs.C[j], TP = tp_of(C);
    const size_t HW = (size_t)jobs.HW[j];
    const int bps = (int)((HW + TP - 1) / TP);
    const size_t n = local / bps, p0 = (size_t)(local % bps) * TP;
    switch (C) {
        case 8: nchw_to_nhwc_tile<8, 256>(jobs.in[j], jobs.out[j], HW, jobs.vec[j], p0, n, tile); break;
        case 16: nchw_to_nhwc_tile<16, 256>(jobs.in[j], jobs.out[j], HW, jobs.vec[j], p0, n, tile); break;
        case 32: nchw_to_nhwc_tile<32, 128>(jobs.in[j], jobs.out[j], HW, jobs.vec[j], p0, n, tile); break;
        default: nchw_to_nhwc_tile<64, 64>(jobs.in[j], jobs.out[j], HW, jobs.vec[j], p0, n, tile); break;
    }
}

int check_shapes(const char* who, int B, int V, int C, int Gin, int D, int H, int W) {
    MVS_REQUIRE(B >= 1 && V >= 2 && D >= 1 && H >= 1 && W >= 1, "%s: bad shape B=%d V=%d D=%d H=%d W=%d", who, B, V, D, H, W);
    MVS_REQUIRE(Gin == G, "%s: only G=8 correlation groups are built (got %d)", who, Gin);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "%s: C must be 8, 16, 32 or 64 (got %d)", who, C);
    MVS_REQUIRE((int64_t)B * (V - 1) <= 65535 && H <= 65535, "%s: grid limits exceeded", who);
    MVS_REQUIRE((int64_t)C * H * W * 4 < ((int64_t)1 << 32), "%s: one view's feature block exceeds the 4 GiB buffer range", who);
    return MVS_OK;
}

}  // namespace

extern "C" int mvs_nchw_to_nhwc(const float* in, float* out, int N, int C, int64_t HW, mvs_stream_t stream) {
    MVS_REQUIRE(in && out, "mvs_nchw_to_nhwc: null pointer");
    MVS_REQUIRE(N >= 1 && N <= 65535 && HW >= 1, "mvs_nchw_to_nhwc: bad shape N=%d HW=%lld", N, (long long)HW);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "mvs_nchw_to_nhwc: C must be 8, 16, 32 or 64 (got %d)", C);
    hipStream_t s = MVS_STREAM(stream);
    const int vec_ok = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
#define MVS_LAUNCH_T(CC, TP) \
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<CC, TP>), dim3((unsigned)((HW + TP - 1) / TP), N), dim3(256), 0, s, in, out, (size_t)HW, vec_ok)
    switch (C) {                                             // 8-16 KB of LDS per block whatever C
        case 8: MVS_LAUNCH_T(8, 256); break;
        case 16: MVS_LAUNCH_T(16, 256); break;
        case 32: MVS_LAUNCH_T(32, 128); break;
        default: MVS_LAUNCH_T(64, 64); break;
    }
#undef MVS_LAUNCH_T
    return mvs::finish_launch("mvs_nchw_to_nhwc");
}

// njobs <= 4 transposes [N_j, C_j, HW_j] -> [N_j, HW_j, C_j] in one launch (the cascade's four stages: three launch gaps less per depth map)
extern "C" int mvs_nchw_to_nhwc_multi(const float* const* in, float* const* out, const int* N, const int* C, const int64_t* HW, int njobs,
                                      mvs_stream_t stream) {
    MVS_REQUIRE(in && out && N && C && HW && njobs >= 1 && njobs <= 4, "mvs_nchw_to_nhwc_multi: 1..4 jobs");
    TransposeJobs jobs{};
    jobs.n = njobs;
    jobs.start[0] = 0;
    for (int j = 0; j < njobs; ++j) {
        MVS_REQUIRE(in[j] && out[j] && N[j] >= 1 && HW[j] >= 1, "mvs_nchw_to_nhwc_multi: job %d: bad shape", j);
        MVS_REQUIRE(C[j] == 8 || C[j] == 16 || C[j] == 32 || C[j] == 64, "mvs_nchw_to_nhwc_multi: C must be 8, 16, 32 or 64 (got %d)", C[j]);
        const int tp = C[j] <= 16 ? 256 : (C[j] == 32 ? 128 : 64);
        const int64_t blocks = (int64_t)N[j] * ((HW[j] + tp - 1) / tp);
        MVS_REQUIRE(jobs.start[j] + blocks < ((int64_t)1 << 31), "mvs_nchw_to_nhwc_multi: too many tiles");
        jobs.in[j] = in[j], jobs.out[j] = out[j], jobs.HW[j] = HW[j], jobs.C[j] = C[j];
        jobs.vec[j] = (HW[j] % 4 == 0) && ((reinterpret_cast<uintptr_t>(in[j]) & 15) == 0);
        jobs.start[j + 1] = jobs.start[j] + (int)blocks;
    }
    hipLaunchKernelGGL(nchw_to_nhwc_multi_kernel, dim3(jobs.start[njobs]), dim3(256), 0, MVS_STREAM(stream), jobs);
    return mvs::finish_launch("mvs_nchw_to_nhwc_multi");
}

extern "C" int mvs_cv_entropy_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int Gin, int D,
                                  int H, int W, float* entropy, int flags, mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && entropy, "mvs_cv_entropy_fwd: null pointer");
    if (int rc = check_shapes("mvs_cv_entropy_fwd", B, V, C, Gin, D, H, W)) return rc;
    const int LPP = C / 4, PPW = 64 / LPP;
    const size_t lds = (size_t)NW * 128 * 32 + (size_t)NW * PPW * D * sizeof(float);
    MVS_REQUIRE(lds <= 64 * 1024, "mvs_cv_entropy_fwd: D=%d with C=%d needs %zu bytes of LDS (> 64 KiB)", D, C, lds);
    const int gx = mvs::ceil_div(W, NW * PPW);
    const int64_t total64 = (int64_t)gx * H * B * (V - 1);
    MVS_REQUIRE(total64 < ((int64_t)1 << 30), "mvs_cv_entropy_fwd: too many blocks");
    const int total = (int)total64;
    dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(64 * NW);
    hipStream_t s = MVS_STREAM(stream);
    const bool fast = !(flags & 1);
#define MVS_LAUNCH_ENT(L)                                                                                                       \
    if (fast) hipLaunchKernelGGL((cv_entropy_kernel<L, true>), grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy, gx, total); \
    else hipLaunchKernelGGL((cv_entropy_kernel<L, false>), grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy, gx, total)
    switch (LPP) {
        case 2: MVS_LAUNCH_ENT(2); break;
        case 4: MVS_LAUNCH_ENT(4); break;
        case 8: MVS_LAUNCH_ENT(8); break;
        default: MVS_LAUNCH_ENT(16); break;
    }
#undef MVS_LAUNCH_ENT
    return mvs::finish_launch("mvs_cv_entropy_fwd");
}

static int cv_aggregate_impl(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V, int C, int Gin, int D, int H,
                             int W, float* volume, void* volume16, float* sim_depth, int flags, mvs_stream_t stream);

extern "C" int mvs_cv_aggregate_fwd(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V,
                                    int C, int Gin, int D, int H, int W, float* volume, float* sim_depth, int flags,
                                    mvs_stream_t stream) {
    return cv_aggregate_impl(feat, rt, depth, weight, B, V, C, Gin, D, H, W, volume, nullptr, sim_depth, flags, stream);
}

// the same + volume16: the volume ALSO as bf16 channel-last [B,D,H,W,G] (device bf16), the layout the bf16 training regularizer reads
extern "C" int mvs_cv_aggregate_fwd_bf16(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V, int C, int Gin,
                                         int D, int H, int W, float* volume, void* volume16, float* sim_depth, int flags, mvs_stream_t stream) {
    MVS_REQUIRE(volume16, "mvs_cv_aggregate_fwd_bf16: null volume16");
    return cv_aggregate_impl(feat, rt, depth, weight, B, V, C, Gin, D, H, W, volume, volume16, sim_depth, flags, stream);
}

static int cv_aggregate_impl(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V, int C, int Gin, int D, int H,
                             int W, float* volume, void* volume16, float* sim_depth, int flags, mvs_stream_t stream) {
    __bf16* vol16 = reinterpret_cast<__bf16*>(volume16);
    MVS_REQUIRE(feat && rt && depth && weight && volume, "mvs_cv_aggregate_fwd: null pointer");
    if (int rc = check_shapes("mvs_cv_aggregate_fwd", B, V, C, Gin, D, H, W)) return rc;
    const int LPP = C / 4, PPW = 64 / LPP;
    const int gx = mvs::ceil_div(W, NW * PPW);
    const int64_t total64 = (int64_t)gx * H * B;
    MVS_REQUIRE(total64 < ((int64_t)1 << 30), "mvs_cv_aggregate_fwd: too many blocks");
    const int total = (int)total64;
    dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(64 * NW);
    hipStream_t s = MVS_STREAM(stream);
    const size_t lds = (size_t)NW * 128 * 32;
    const bool fast = !(flags & 1);
#define MVS_LAUNCH_AGG2(L, SIMV, FASTV)                                                                                  \
    hipLaunchKernelGGL((cv_aggregate_kernel<L, SIMV, FASTV>), grid, block, lds, s, feat, rt, depth, weight, V, D, H, W, volume, \
                       sim_depth, gx, total, vol16)
#define MVS_LAUNCH_AGG(L)                                                              \
    if (sim_depth) {                                                                   \
        if (fast) MVS_LAUNCH_AGG2(L, true, true); else MVS_LAUNCH_AGG2(L, true, false);   \
    } else {                                                                           \
        if (fast) MVS_LAUNCH_AGG2(L, false, true); else MVS_LAUNCH_AGG2(L, false, false); \
    }
    switch (LPP) {
        case 2: MVS_LAUNCH_AGG(2); break;
        case 4: MVS_LAUNCH_AGG(4); break;
        case 8: MVS_LAUNCH_AGG(8); break;
        default: MVS_LAUNCH_AGG(16); break;
    }
#undef MVS_LAUNCH_AGG
#undef MVS_LAUNCH_AGG2
    return mvs::finish_launch("mvs_cv_aggregate_fwd");
}

// ---------------------------------------------------------------------------------------------------------
// stored-correlation sweeps (coarse stages)
// ---------------------------------------------------------------------------------------------------------
namespace {
struct StoreLayout { int64_t corr_floats, sim_floats; int TPX, XG; };
StoreLayout store_layout(int B, int V, int C, int D, int H, int W) {
    StoreLayout l;
    l.TPX = 256 / C;
    l.XG = mvs::ceil_div(W, l.TPX);
    l.corr_floats = (int64_t)B * (V - 1) * H * l.XG * D * G * l.TPX;
    l.sim_floats = (int64_t)B * (V - 1) * H * l.XG * D * l.TPX;
    return l;
}
}  // namespace

extern "C" int64_t mvs_cv_corr_store_bytes(int B, int V, int C, int Gin, int D, int H, int W) {
    if (B < 1 || V < 2 || D < 1 || H < 1 || W < 1 || Gin != G || (C != 32 && C != 64)) return -1;
    // mvs_cv_corr_fwd keeps D values per pixel in LDS: beyond 64 KiB the path is not built for the shape either (-1: the caller recomputes)
    if ((C == 64 ? CorrCfg<16>::lds_bytes(D) : CorrCfg<8>::lds_bytes(D)) > 64 * 1024) return -1;
    const StoreLayout l = store_layout(B, V, C, D, H, W);
    return (l.corr_floats + l.sim_floats) * (int64_t)sizeof(float);
}

namespace {
int corr_launch(const char* who, const float* feat, const float* rt, const float* depth, int B, int V, int C, int Gin, int D, int H, int W, int y0, int Hs,
                float* entropy, void* store, int flags, mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && entropy && store, "%s: null pointer", who);
    if (int rc = check_shapes(who, B, V, C, Gin, D, H, W)) return rc;
    MVS_REQUIRE(C == 32 || C == 64, "%s: the stored-correlation sweeps are built for C = 32 and C = 64 (got %d)", who, C);
    MVS_REQUIRE(y0 >= 0 && Hs >= 1 && y0 + Hs <= H, "%s: rows [%d, %d) outside the image (H = %d)", who, y0, y0 + Hs, H);
    MVS_REQUIRE((reinterpret_cast<uintptr_t>(store) & 15) == 0, "%s: store must be 16-byte aligned", who);
    const int LPP = C / 4, PPW = 64 / LPP;
    const size_t lds = LPP == 16 ? CorrCfg<16>::lds_bytes(D) : CorrCfg<8>::lds_bytes(D);
    MVS_REQUIRE(lds <= 64 * 1024, "%s: D=%d with C=%d needs %zu bytes of LDS (> 64 KiB)", who, D, C, lds);
    const StoreLayout l = store_layout(B, V, C, D, Hs, W);
    float* corr = static_cast<float*>(store);
    float* simv = corr + l.corr_floats;
    const int gx = mvs::ceil_div(W, NW * PPW);
    const int64_t total64 = (int64_t)gx * Hs * B * (V - 1);
    MVS_REQUIRE(total64 < ((int64_t)1 << 30), "%s: too many blocks", who);
    const int total = (int)total64;
    dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(64 * NW);
    hipStream_t s = MVS_STREAM(stream);
    const bool fast = !(flags & 1);
#define MVS_LAUNCH_CORR(L)                                                                                                                          \
    if (fast) hipLaunchKernelGGL((cv_corr_kernel<L, true>), grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy, corr, simv, gx, total, y0, Hs); \
    else hipLaunchKernelGGL((cv_corr_kernel<L, false>), grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy, corr, simv, gx, total, y0, Hs)
    if (LPP == 16) { MVS_LAUNCH_CORR(16); } else { MVS_LAUNCH_CORR(8); }
#undef MVS_LAUNCH_CORR
    return mvs::finish_launch(who);
}

int merge_launch(const char* who, const void* store, const float* depth, const float* weight, int B, int V, int C, int Gin, int D, int H, int W, int y0, int Hs,
                 int r_lo, int nrows, float* volume, float* sim_depth, mvs_stream_t stream) {
    MVS_REQUIRE(store && depth && weight && volume, "%s: null pointer", who);
    if (int rc = check_shapes(who, B, V, C, Gin, D, H, W)) return rc;
    MVS_REQUIRE(C == 32 || C == 64, "%s: the stored-correlation sweeps are built for C = 32 and C = 64 (got %d)", who, C);
    MVS_REQUIRE(y0 >= 0 && Hs >= 1 && y0 + Hs <= H && r_lo >= 0 && nrows >= 1 && r_lo + nrows <= Hs, "%s: band rows [%d, %d) of a band [%d, %d) of H = %d", who, r_lo,
                r_lo + nrows, y0, y0 + Hs, H);
    const StoreLayout l = store_layout(B, V, C, D, Hs, W);
    const float* corr = static_cast<const float*>(store);
    const float* simv = corr + l.corr_floats;
    const int gx = mvs::ceil_div(l.XG, NW);
    const int64_t total64 = (int64_t)gx * nrows * B;
    MVS_REQUIRE(total64 < ((int64_t)1 << 30), "%s: too many blocks", who);
    const int total = (int)total64;
    dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(64 * NW);
    hipStream_t s = MVS_STREAM(stream);
    if (l.TPX == 4) hipLaunchKernelGGL((cv_merge_kernel<4>), grid, block, 0, s, corr, simv, depth, weight, V, D, H, W, volume, sim_depth, gx, total, y0, Hs, r_lo, nrows);
    else hipLaunchKernelGGL((cv_merge_kernel<8>), grid, block, 0, s, corr, simv, depth, weight, V, D, H, W, volume, sim_depth, gx, total, y0, Hs, r_lo, nrows);
    return mvs::finish_launch(who);
}
}  // namespace

extern "C" int mvs_cv_corr_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int Gin, int D, int H, int W,
                               float* entropy, void* store, int flags, mvs_stream_t stream) {
    return corr_launch("mvs_cv_corr_fwd", feat, rt, depth, B, V, C, Gin, D, H, W, 0, H, entropy, store, flags, stream);
}

extern "C" int mvs_cv_corr_rows_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int Gin, int D, int H, int W, int y0,
                                    int rows, float* entropy, void* store, int flags, mvs_stream_t stream) {
    return corr_launch("mvs_cv_corr_rows_fwd", feat, rt, depth, B, V, C, Gin, D, H, W, y0, rows, entropy, store, flags, stream);
}

extern "C" int mvs_cv_merge_rows_fwd(const void* store, const float* depth, const float* weight, int B, int V, int C, int Gin, int D, int H, int W, int y0,
                                     int rows, int r_lo, int nrows, float* volume, float* sim_depth, mvs_stream_t stream) {
    return merge_launch("mvs_cv_merge_rows_fwd", store, depth, weight, B, V, C, Gin, D, H, W, y0, rows, r_lo, nrows, volume, sim_depth, stream);
}

extern "C" int mvs_cv_merge_fwd(const void* store, const float* depth, const float* weight, int B, int V, int C, int Gin, int D, int H, int W,
                                float* volume, float* sim_depth, mvs_stream_t stream) {
    return merge_launch("mvs_cv_merge_fwd", store, depth, weight, B, V, C, Gin, D, H, W, 0, H, 0, H, volume, sim_depth, stream);
}
