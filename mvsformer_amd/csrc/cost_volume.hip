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
                                                               float* __restrict__ sim_depth, int gx, int total) {
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
                            for (int k = 0; k < NG; ++k)
                                volume[((size_t)(b * G + cq * NG + k) * D + d) * HW + pix] = acc[dd][k] / denom;
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
                        volume[((size_t)(b * G + g) * D + d) * HW + pix] = acc[dd][k] / denom;
                    }
                }
                if (SIM && simtot[dd] > best) { best = simtot[dd]; besti = d; }
            }
        }
    }
    if (SIM && active && cq == 0) sim_depth[(size_t)b * HW + pix] = depth_row[(size_t)besti * HW + xg];
}

// ---------------------------------------------------------------------------------------------------------
// NCHW -> NHWC feature transpose ([N,C,HW] -> [N,HW,C]) through an LDS tile of TP pixels: reads are 16 bytes per lane along a
// channel row (a wavefront reads 1 KB contiguous), writes are one contiguous TP*C-float run per block (dwordx4 per lane).  VEC = false
// (HW not a multiple of 4, or a ragged last block) falls back to dword reads.
// ---------------------------------------------------------------------------------------------------------
template <int C, int TP>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, size_t HW, int vec_ok) {
    constexpr int LD = TP + 4;                               // row stride: 16-byte aligned rows, rows 4 banks apart
    __shared__ __attribute__((aligned(16))) float tile[C * LD];
    const int tid = threadIdx.x;
    const size_t p0 = (size_t)blockIdx.x * TP;
    const size_t n = blockIdx.y;
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

extern "C" int mvs_cv_aggregate_fwd(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V,
                                    int C, int Gin, int D, int H, int W, float* volume, float* sim_depth, int flags,
                                    mvs_stream_t stream) {
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
                       sim_depth, gx, total)
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
