// Backward of the fused cost-volume build (forward = cost_volume.hip sweep B, reference
// models/mvsformer_model.py:73-79,101-105).  The sampling grid is built under no_grad in the reference (warping.py:79-97)
// and the hypotheses enter detached, so gradients flow only to the feature maps and to the visibility weights:
//     vm[g,d]   = sum_v ip_v[g,d] * w_v / S,  S = sum_v w_v + 1e-6,  ip_v[g,d] = mean_j ref[g,j] * warp_v[g,j,d]
//     dL/dref[c]        = sum_{v,d} G[g(c),d] * w_v / (S*CPG) * warp_v[c,d]
//     dL/dwarp_v[c,d]   = G[g(c),d] * w_v / (S*CPG) * ref[c]     -> scattered to the 4 bilinear taps of source view v
//     dL/dw_v           = ( sum_{g,d} G[g,d] * ip_v[g,d]  -  sum_{g,d} G[g,d] * vm[g,d] ) / S
// Same thread mapping as the forward sweep (channel-last, LPP lanes per pixel, geometry pass through LDS); the warp is
// recomputed, nothing but the forward inputs and outputs is saved.  The bilinear scatter uses fp32 global atomics —
// the same non-determinism class as ATen's grid_sampler_2d_backward.
#include "common.h"
#include "geometry.h"

namespace {

constexpr int G = 8;
constexpr int NW = 4;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ f32x4 buf_load4(mvs::rsrc_t r, unsigned voff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, 0, 0));
}
template <int LPP>
__device__ __forceinline__ float pixel_sum(float v) {
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int LPP>
__global__ __launch_bounds__(64 * NW) void cv_aggregate_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ rt_all,
                                                                   const float* __restrict__ depth, const float* __restrict__ weight,
                                                                   const float* __restrict__ volume, const float* __restrict__ gvol,
                                                                   int V, int D, int H, int W, float* __restrict__ dfeat,
                                                                   float* __restrict__ dweight) {
    constexpr int C = 4 * LPP, CPG = C / G, PPW = 64 / LPP;
    constexpr int NG = (CPG >= 4) ? 1 : 4 / CPG;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NW * 64 * 32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4* taps_o = reinterpret_cast<u32x4*>(smem) + wave * 64;
    f32x4* taps_w = reinterpret_cast<f32x4*>(smem + NW * 64 * 16) + wave * 64;

    const int x0 = (blockIdx.x * NW + wave) * PPW, y = blockIdx.y, b = blockIdx.z;
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
    const float* wp = weight + (size_t)(b * (V - 1)) * HW + pix;
    float wsum = 0.0f;
    for (int sv = 0; sv < V - 1; ++sv) wsum = wsum + wp[(size_t)sv * HW];
    const float inv_s = 1.0f / (wsum + 1e-6f);
    // this lane's groups: g = cq*NG + k (CPG < 8) or cq/2 (CPG == 8); channel i of the lane belongs to group slot i / CPG (CPG < 4) else 0
    const int g0 = (CPG == 8) ? (cq >> 1) : cq * NG;
    const float* gp = gvol + ((size_t)(b * G + g0) * D) * HW + pix;      // + k*D*HW + d*HW
    const float* vp = volume + ((size_t)(b * G + g0) * D) * HW + pix;

    // T = sum_{g,d} G*vm over the pixel (each group counted once)
    float tsum = 0.0f;
    if (CPG < 8 || (cq & 1) == 0) {
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < NG; ++k) tsum = fmaf(gp[((size_t)k * D + d) * HW], vp[((size_t)k * D + d) * HW], tsum);
    }
    tsum = pixel_sum<LPP>(tsum);

    f32x4 dref = {0.f, 0.f, 0.f, 0.f};
    for (int sv = 0; sv < V - 1; ++sv) {
        const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
        const mvs::rsrc_t src = mvs::make_rsrc(feat + (size_t)(b * V + sv + 1) * HW * C, (unsigned)(HW * pix_bytes));
        float* dsrc = dfeat + (size_t)(b * V + sv + 1) * HW * C + cq * 4;
        const float wv = wp[(size_t)sv * HW];
        float gip = 0.0f;                                    // sum_{d, own channels} coef * ref * warp
        // Scatter with run merging: fp32 atomics are what this kernel costs (7.8 of 8.0 ms at the finest stage), and
        // neighbouring depth planes of one pixel usually land on the same 2x2 texel block (plane spacing < 1 px at the fine
        // stages).  The lane keeps the block's 4 taps x 4 channels in registers and only issues atomics when the block
        // changes (and once at the end of the view).
        u32x4 run_o = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        float run[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) run[k][i] = 0.0f;
        auto flush = [&]() {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float* dst = dsrc + (size_t)run_o[k] * C;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (run[k][i] != 0.0f) atomicAdd(dst + i, run[k][i]);
                    run[k][i] = 0.0f;
                }
            }
        };
        for (int c0 = 0; c0 < D; c0 += LPP) {
            __builtin_amdgcn_wave_barrier();
            {
                const int p = lane % PPW, dd = lane / PPW;
                const int d = min(c0 + dd, D - 1);
                const int x = min(x0 + p, W - 1);
                float un, vn, z;
                mvs::sweep_project(rt, (float)x, (float)y, depth_row[(size_t)d * HW + x], half_w, half_h, &un, &vn, &z);
                const mvs::Taps t = mvs::sweep_taps(un, vn, H, W, half_w, half_h);
                taps_o[lane] = u32x4{(unsigned)t.o00, (unsigned)t.o01, (unsigned)t.o10, (unsigned)t.o11};
                taps_w[lane] = f32x4{t.w00, t.w01, t.w10, t.w11};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int dd = 0; dd < LPP; ++dd) {
                const int d = c0 + dd;
                if (d < D) {
                    const u32x4 o = taps_o[dd * PPW + pg];
                    const f32x4 w = taps_w[dd * PPW + pg];
                    f32x4 t4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) t4[k] = buf_load4(src, o[k] * pix_bytes + cq * 16u);
                    float coef[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = (CPG < 4) ? i / CPG : 0;
                        coef[i] = gp[((size_t)k * D + d) * HW] * inv_s * (1.0f / CPG);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float g4 = t4[0][i] * w[0];
                        g4 = fmaf(t4[1][i], w[1], g4);
                        g4 = fmaf(t4[2][i], w[2], g4);
                        g4 = fmaf(t4[3][i], w[3], g4);
                        dref[i] = fmaf(coef[i] * wv, g4, dref[i]);
                        gip = fmaf(coef[i] * r[i], g4, gip);
                    }
                    if (active) {
                        const bool same = o[0] == run_o[0] && o[1] == run_o[1] && o[2] == run_o[2] && o[3] == run_o[3];
                        if (!same) {
                            // the block usually moves by ONE texel along the epipolar line: keep the half that stays
                            // (taps are ordered 00, 01 (x+1), 10 (y+1), 11), flush only the half that leaves
                            const bool xp = o[0] == run_o[1] && o[2] == run_o[3], xm = o[1] == run_o[0] && o[3] == run_o[2];
                            const bool yp = o[0] == run_o[2] && o[1] == run_o[3], ym = o[2] == run_o[0] && o[3] == run_o[1];
                            if (run_o[0] == 0xFFFFFFFFu) {
                            } else if (xp || xm || yp || ym) {
                                // a = leaving pair, b = staying pair (indices into the old block)
                                const int a0 = xp ? 0 : (xm ? 1 : (yp ? 0 : 2)), a1 = xp ? 2 : (xm ? 3 : (yp ? 1 : 3));
                                float keep[2][4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const bool leaving = (k == a0) || (k == a1);
                                    if (leaving) {
                                        float* dst = dsrc + (size_t)run_o[k] * C;
#pragma unroll
                                        for (int i = 0; i < 4; ++i)
                                            if (run[k][i] != 0.0f) atomicAdd(dst + i, run[k][i]);
                                    }
                                }
                                // staying pair in old indexing: xp -> (1,3), xm -> (0,2), yp -> (2,3), ym -> (0,1)
                                const int b0i = xp ? 1 : (xm ? 0 : (yp ? 2 : 0)), b1i = xp ? 3 : (xm ? 2 : (yp ? 3 : 1));
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        v0 = (k == b0i) ? run[k][i] : v0;
                                        v1 = (k == b1i) ? run[k][i] : v1;
                                    }
                                    keep[0][i] = v0;
                                    keep[1][i] = v1;
                                }
                                // new positions of the staying pair: xp -> (0,2), xm -> (1,3), yp -> (0,1), ym -> (2,3)
                                const int n0 = xp ? 0 : (xm ? 1 : (yp ? 0 : 2)), n1 = xp ? 2 : (xm ? 3 : (yp ? 1 : 3));
#pragma unroll
                                for (int k = 0; k < 4; ++k)
#pragma unroll
                                    for (int i = 0; i < 4; ++i) run[k][i] = (k == n0) ? keep[0][i] : ((k == n1) ? keep[1][i] : 0.0f);
                            } else {
                                flush();
                            }
                            run_o = o;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int i = 0; i < 4; ++i) run[k][i] = fmaf(w[k], coef[i] * wv * r[i], run[k][i]);
                    }
                }
            }
        }
        if (active && run_o[0] != 0xFFFFFFFFu) flush();
        gip = pixel_sum<LPP>(gip);
        if (active && cq == 0) dweight[(size_t)(b * (V - 1) + sv) * HW + pix] = gip - tsum * inv_s;
    }
    if (active) *reinterpret_cast<f32x4*>(dfeat + ((size_t)(b * V) * HW + pix) * C + cq * 4) = dref;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Second form: the bilinear scatter accumulated in LDS.  The kernel above spends ~90 % of its time in global fp32 atomics (one per
// tap and channel whenever a lane's 2x2 texel block changes - plane spacing is 1-1.7 px, so nearly every sample).  But the samples of a
// 16 x 8 tile of reference pixels over all D planes land on a few hundred source texels, each hit ~4*D times.  Here a block owns such
// a tile and ONE channel octet, keeps a WX x WY texel window (8 channels, fp32) of the source view's gradient in LDS, adds every tap
// that falls inside the window with ds_add_f32, and flushes the non-zero texels with one global atomic per value at the end of the
// view; taps outside the window (rare for coherent hypotheses) go straight to global atomics, so the result does not depend on the
// window at all.  C = 16/32/64 run as 2/4/8 independent octets (blockIdx.z) - more blocks on the small coarse stages, the same LDS
// footprint - whose visibility-weight partial sums land in gip_part[octet] and are added by the caller.
// ---------------------------------------------------------------------------------------------------------------------------
#ifndef MVS_BWD_EXP
#define MVS_BWD_EXP 0                                       // experiment switches (tools/exp_cv_bwd.py --build): 1 no LDS adds, 2 no outlier atomics,
#endif                                                      // 4 constant coefficients instead of the gvol loads, 8 no tap gathers,
                                                            // 16 ds_add_u32 (fixed point) / 32 ds_add_u64 (two channels per op) for timing
constexpr int LT_W = 16, LT_H = 8, LT_NW = 4;               // tile of reference pixels; wavefront = 2 rows x 16 pixels x 2 channel quads

template <int C>
__global__ __launch_bounds__(64 * LT_NW) void cv_aggregate_bwd_lds_kernel(const float* __restrict__ feat, const float* __restrict__ rt_all,
                                                                          const float* __restrict__ depth, const float* __restrict__ weight,
                                                                          const float* __restrict__ volume, const float* __restrict__ gvol,
                                                                          int V, int D, int H, int W, float* __restrict__ dfeat,
                                                                          float* __restrict__ gip_part, int wx_log2, int WY,
                                                                          unsigned* __restrict__ stats) {
    constexpr int CPG = C / G, NOCT = C / 8;                 // channels per correlation group, channel octets
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int WX = 1 << wx_log2;
    // window = 8 channel planes of WY x WX texels.  Planar, not texel-major: for one tap and channel the 64 lanes of a ds_add_f32 then
    // fall on neighbouring texels = neighbouring banks (texel-major put all of them on 8 of the 32 banks: 8-way conflicts on an
    // instruction that already costs several cycles per access - measured 1.44 of 1.74 ms at stage 4).  Plane stride == 16 (mod 32)
    // so that the two channel quads of a pixel (cq = 0 / 1, 4 planes apart) land on different bank halves.
    const int WPL = WX * WY + 4;                             // 4 planes apart = 16 banks apart
    float* win = reinterpret_cast<float*>(smem_raw);                                      // [8][WPL]
    unsigned char* hand = smem_raw + (size_t)WPL * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4* taps_o = reinterpret_cast<u32x4*>(hand) + wave * 64;
    f32x4* taps_w = reinterpret_cast<f32x4*>(hand + LT_NW * 64 * 16) + wave * 64;
    int* taps_xy = reinterpret_cast<int*>(hand + LT_NW * 64 * 32) + wave * 64;
    int* origin = reinterpret_cast<int*>(hand + LT_NW * 64 * 36);                        // [2]

    const int oct = blockIdx.z % NOCT, b = blockIdx.z / NOCT;
    const int tx0 = blockIdx.x * LT_W, ty0 = blockIdx.y * LT_H;
    const size_t HW = (size_t)H * W;
    const unsigned pix_bytes = C * 4u;
    const int pg = lane >> 1, cq = lane & 1;
    const int xi = tx0 + (pg & 15), yi = ty0 + wave * 2 + (pg >> 4);
    const bool active = xi < W && yi < H;
    const int xg = min(xi, W - 1), yg = min(yi, H - 1);
    const size_t pix = (size_t)yg * W + xg;
    const int cb = oct * 8 + cq * 4;                          // first of the lane's 4 channels
    const f32x4 r = *reinterpret_cast<const f32x4*>(feat + ((size_t)(b * V) * HW + pix) * C + cb);
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
    const float* wp = weight + (size_t)(b * (V - 1)) * HW + pix;
    float wsum = 0.0f;
    for (int sv = 0; sv < V - 1; ++sv) wsum = wsum + wp[(size_t)sv * HW];
    const float inv_s = 1.0f / (wsum + 1e-6f);
    const int g0 = cb / CPG;                                  // CPG = 1: 4 groups cb..cb+3; 2: 2 groups; >= 4: the one group
    const float* gp = gvol + ((size_t)(b * G + g0) * D) * HW + pix;

    // T = sum_{g,d} G*vm over the pixel: once per pixel, by the octet-0 block (its two lanes take four groups each)
    float tsum = 0.0f;
    if (oct == 0) {
        const float* ga = gvol + ((size_t)(b * G + cq * 4) * D) * HW + pix;
        const float* va = volume + ((size_t)(b * G + cq * 4) * D) * HW + pix;
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < 4; ++k) tsum = fmaf(ga[((size_t)k * D + d) * HW], va[((size_t)k * D + d) * HW], tsum);
        tsum += __shfl_xor(tsum, 1, 64);
    }
    for (int i = threadIdx.x; i < WPL * 2; i += 64 * LT_NW) reinterpret_cast<f32x4*>(win)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 dref = {0.f, 0.f, 0.f, 0.f};
    for (int sv = 0; sv < V - 1; ++sv) {
        const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
        const mvs::rsrc_t src = mvs::make_rsrc(feat + (size_t)(b * V + sv + 1) * HW * C, (unsigned)(HW * pix_bytes));
        float* dsrc = dfeat + (size_t)(b * V + sv + 1) * HW * C + oct * 8;
        const float wv = wp[(size_t)sv * HW];
        // window origin: around the projection of the tile centre on its middle plane
        if (threadIdx.x == 0) {
            const int xc = min(tx0 + LT_W / 2, W - 1), yc = min(ty0 + LT_H / 2, H - 1);
            float un, vn, z;
            mvs::sweep_project(rt, (float)xc, (float)yc, depth[((size_t)b * D + D / 2) * HW + (size_t)yc * W + xc], half_w, half_h, &un, &vn, &z);
            const float ix = (un + 1.0f) * half_w, iy = (vn + 1.0f) * half_h;
            const bool ok = fabsf(ix) < 1e6f && fabsf(iy) < 1e6f;       // also false for NaN
            // clamped to where a window can still hold an in-image tap: the packed 16-bit window coordinates below must not wrap
            origin[0] = ok ? min(max((int)floorf(ix) - WX / 2, -WX), W + 1) : 0;
            origin[1] = ok ? min(max((int)floorf(iy) - WY / 2, -WY), H + 1) : 0;
        }
        __syncthreads();                                      // origin visible; window zeroed (first view) / flushed (later views)
        const int ox = origin[0], oy = origin[1];
        float gip = 0.0f;                                     // sum_{d, own channels} coef * ref * warp
        for (int c0 = 0; c0 < D; c0 += 2) {
            __builtin_amdgcn_wave_barrier();
            {
                const int p = lane & 31, dd = lane >> 5;
                const int d = min(c0 + dd, D - 1);
                const int x = min(tx0 + (p & 15), W - 1), y = min(ty0 + wave * 2 + (p >> 4), H - 1);
                float un, vn, z;
                mvs::sweep_project(rt, (float)x, (float)y, depth[((size_t)b * D + d) * HW + (size_t)y * W + x], half_w, half_h, &un, &vn, &z);
                int x0, y0;
                const mvs::Taps t = mvs::sweep_taps_xy(un, vn, H, W, half_w, half_h, &x0, &y0);
                taps_o[lane] = u32x4{(unsigned)t.o00, (unsigned)t.o01, (unsigned)t.o10, (unsigned)t.o11};
                taps_w[lane] = f32x4{t.w00, t.w01, t.w10, t.w11};
                taps_xy[lane] = ((y0 - oy) << 16) | ((x0 - ox) & 0xFFFF);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                const int d = c0 + dd;
                if (d >= D) break;
                const u32x4 o = taps_o[dd * 32 + pg];
                const f32x4 w = taps_w[dd * 32 + pg];
                const int xy = taps_xy[dd * 32 + pg];
                f32x4 t4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = (MVS_BWD_EXP & 8) ? f32x4{w[k], w[0], w[1], w[2]} : buf_load4(src, o[k] * pix_bytes + (unsigned)cb * 4u);
                float coef[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = (CPG < 4) ? i / CPG : 0;
                    coef[i] = ((MVS_BWD_EXP & 4) ? w[k] : gp[((size_t)k * D + d) * HW]) * inv_s * (1.0f / CPG);
                }
                float cr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float g4 = t4[0][i] * w[0];
                    g4 = fmaf(t4[1][i], w[1], g4);
                    g4 = fmaf(t4[2][i], w[2], g4);
                    g4 = fmaf(t4[3][i], w[3], g4);
                    dref[i] = fmaf(coef[i] * wv, g4, dref[i]);
                    gip = fmaf(coef[i] * r[i], g4, gip);
                    cr[i] = coef[i] * wv * r[i];
                }
                if (active) {
                    const int rx = (int)(short)(xy & 0xFFFF), ry = xy >> 16;      // tap (0,0) relative to the window
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (w[k] == 0.0f) continue;           // taps outside the image carry weight 0 (and a clamped offset)
                        const int txx = rx + (k & 1), tyy = ry + (k >> 1);
                        if ((unsigned)txx < (unsigned)WX && (unsigned)tyy < (unsigned)WY) {
                            float* cell = win + (cq * 4) * WPL + (tyy << wx_log2) + txx;
                            if (MVS_BWD_EXP & 32) {          // timing experiment: two channels per 64-bit integer LDS atomic
                                unsigned long long* c64 = reinterpret_cast<unsigned long long*>(win) + ((cq * 2) * (WPL / 2) + (tyy << wx_log2) + txx);
#pragma unroll
                                for (int i = 0; i < 2; ++i) {
                                    const long long lo = (long long)(int)(w[k] * cr[2 * i] * 1048576.0f), hi = (long long)(int)(w[k] * cr[2 * i + 1] * 1048576.0f);
                                    atomicAdd(c64 + i * (WPL / 2), (unsigned long long)((hi << 32) + lo));
                                }
                            } else
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (MVS_BWD_EXP & 16) atomicAdd(reinterpret_cast<int*>(cell + i * WPL), (int)(w[k] * cr[i] * 1048576.0f));
                                else if (!(MVS_BWD_EXP & 1)) atomicAdd(cell + i * WPL, w[k] * cr[i]);
                                else dref[i] += w[k] * cr[i] * (float)txx;
                        } else {
                            float* dst = dsrc + (size_t)o[k] * C + cq * 4;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (!(MVS_BWD_EXP & 2)) atomicAdd(dst + i, w[k] * cr[i]);
                                else dref[i] += w[k] * cr[i] * (float)tyy;
                            if (stats) atomicAdd(stats + 1, 1u);           // diagnostics only (tools/exp_cv_bwd.py)
                        }
                        if (stats) atomicAdd(stats, 1u);
                    }
                }
            }
        }
        gip += __shfl_xor(gip, 1, 64);
        if (active && cq == 0)
            gip_part[((size_t)(oct * gridDim.z / NOCT + b) * (V - 1) + sv) * HW + pix] = oct == 0 ? gip - tsum * inv_s : gip;
        __syncthreads();                                      // every tap of this view is in the window
        // flush: a thread takes 4 consecutive texels of one channel plane (one ds_read_b128), touched values go out as global atomics
        for (int i = threadIdx.x; i < WX * WY * 2; i += 64 * LT_NW) {
            const int c = i / (WX * WY / 4), q = i % (WX * WY / 4);
            f32x4* cellp = reinterpret_cast<f32x4*>(win + c * WPL) + q;
            const f32x4 v = *cellp;
            if (v[0] != 0.0f || v[1] != 0.0f || v[2] != 0.0f || v[3] != 0.0f) {
                const int tex = q * 4, gx = ox + (tex & (WX - 1)), gy = oy + (tex >> wx_log2);
                float* dst = dsrc + ((size_t)gy * W + gx) * C + c;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] != 0.0f) atomicAdd(dst + (size_t)e * C, v[e]);
                *cellp = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    if (active) *reinterpret_cast<f32x4*>(dfeat + ((size_t)(b * V) * HW + pix) * C + cb) = dref;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Third form: the scatter WITHOUT atomics in the common case.  Float atomics retire one lane-operation every ~2.7 clocks per CU in
// LDS and in the L2 alike (profiles/r02_cv_bwd_lds_ablation.txt), so the only way below the direct kernel is not to issue them.
// A WAVEFRONT owns an 8 x 4 tile of reference pixels (32 pixels x 2 channel quads), one channel octet and a private window of the
// source view's gradient in LDS.  Nobody else touches that window, and a wavefront's LDS operations execute in program order, so a tap
// is added with a plain read-modify-write (ds_read_b128 / ds_write_b128 of the lane's four channels) once the lanes that target the
// SAME cell in the same instruction have been serialized: every pending lane writes its lane id into a per-cell tag byte, reads it
// back, and the lane that finds its own id owns the cell for this round; the others go round again (collisions need two pixels on
// one texel for the same tap - rare).  Planes are walked in chunks of 8 with the window re-centred per chunk (a D = 32 sweep spans
// 50+ texels along the epipolar line), the touched cells are flushed with global atomics per (view, chunk), taps outside the window
// go straight to global atomics - so the result does not depend on the window.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int OW_TW = 8, OW_TH = 4, OW_NW = 2, OW_DCH = 8;   // wave tile, wavefronts per block (side by side in x), planes per chunk
constexpr int OW_QCAP = 128;                                 // out-of-window taps queued per wavefront before they are issued

template <int C>
__global__ __launch_bounds__(64 * OW_NW) void cv_aggregate_bwd_own_kernel(const float* __restrict__ feat, const float* __restrict__ rt_all,
                                                                          const float* __restrict__ depth, const float* __restrict__ weight,
                                                                          const float* __restrict__ volume, const float* __restrict__ gvol,
                                                                          int V, int D, int H, int W, float* __restrict__ dfeat,
                                                                          float* __restrict__ gip_part, int wx_log2, int WY,
                                                                          unsigned* __restrict__ stats) {
    constexpr int CPG = C / G, NOCT = C / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int WX = 1 << wx_log2, NCELL = WX * WY;
    const unsigned omask = (1u << (31 - __builtin_clz((unsigned)NCELL))) - 1u;      // tag slots of the footprint origins: a power of two <= NCELL
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per_wave = (size_t)NCELL * 32 + 64 * 36 + (size_t)OW_QCAP * 20 + (size_t)NCELL * 2;
    unsigned char* base = smem_raw + wave * ((per_wave + 15) & ~(size_t)15);
    f32x4* win = reinterpret_cast<f32x4*>(base);                                          // [NCELL][2 quads]
    u32x4* taps_o = reinterpret_cast<u32x4*>(base + (size_t)NCELL * 32);
    f32x4* taps_w = reinterpret_cast<f32x4*>(base + (size_t)NCELL * 32 + 64 * 16);
    int* taps_xy = reinterpret_cast<int*>(base + (size_t)NCELL * 32 + 64 * 32);
    // [2 quads][NCELL] owner tags; volatile (no store-to-load forwarding: the read must see the last writer) and explicitly in the LDS
    // address space (a generic volatile pointer compiles to flat_store/flat_load with full memory waits)
    typedef volatile __attribute__((address_space(3))) unsigned char lds_tag_t;
    // Taps that miss the window are not issued one instruction per (plane, tap) - a global atomic instruction with one live lane
    // costs what one with 64 does - but queued (slot = running count + rank among the missing lanes, no atomics) and issued 64 at a time.
    f32x4* qval = reinterpret_cast<f32x4*>(base + (size_t)NCELL * 32 + 64 * 36);
    unsigned* qoff = reinterpret_cast<unsigned*>(base + (size_t)NCELL * 32 + 64 * 36 + (size_t)OW_QCAP * 16);
    lds_tag_t* tags = (lds_tag_t*)(base + (size_t)NCELL * 32 + 64 * 36 + (size_t)OW_QCAP * 20);
    int qn = 0;                                               // queued taps (wave-uniform)

    const int oct = blockIdx.z % NOCT, b = blockIdx.z / NOCT;
    const int tx0 = (blockIdx.x * OW_NW + wave) * OW_TW, ty0 = blockIdx.y * OW_TH;
    const size_t HW = (size_t)H * W;
    const unsigned pix_bytes = C * 4u;
    const int pg = lane >> 1, cq = lane & 1;
    const int xi = tx0 + (pg & 7), yi = ty0 + (pg >> 3);
    const bool active = xi < W && yi < H;
    const int xg = min(xi, W - 1), yg = min(yi, H - 1);
    const size_t pix = (size_t)yg * W + xg;
    const int cb = oct * 8 + cq * 4;
    const f32x4 r = *reinterpret_cast<const f32x4*>(feat + ((size_t)(b * V) * HW + pix) * C + cb);
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
    const float* wp = weight + (size_t)(b * (V - 1)) * HW + pix;
    float wsum = 0.0f;
    for (int sv = 0; sv < V - 1; ++sv) wsum = wsum + wp[(size_t)sv * HW];
    const float inv_s = 1.0f / (wsum + 1e-6f);
    const int g0 = cb / CPG;
    const float* gp = gvol + ((size_t)(b * G + g0) * D) * HW + pix;

    float tsum = 0.0f;
    if (oct == 0) {
        const float* ga = gvol + ((size_t)(b * G + cq * 4) * D) * HW + pix;
        const float* va = volume + ((size_t)(b * G + cq * 4) * D) * HW + pix;
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < 4; ++k) tsum = fmaf(ga[((size_t)k * D + d) * HW], va[((size_t)k * D + d) * HW], tsum);
        tsum += __shfl_xor(tsum, 1, 64);
    }
    for (int i = lane; i < NCELL * 2; i += 64) win[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 dref = {0.f, 0.f, 0.f, 0.f};
    for (int sv = 0; sv < V - 1; ++sv) {
        const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
        const mvs::rsrc_t src = mvs::make_rsrc(feat + (size_t)(b * V + sv + 1) * HW * C, (unsigned)(HW * pix_bytes));
        float* dsrc = dfeat + (size_t)(b * V + sv + 1) * HW * C + oct * 8;
        const float wv = wp[(size_t)sv * HW];
        float gip = 0.0f;
        auto drain = [&]() {
            for (int e = lane; e < qn; e += 64) {
                const f32x4 v = qval[e];
                float* dst = dsrc + qoff[e];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (v[i] != 0.0f) atomicAdd(dst + i, v[i]);
            }
            qn = 0;
        };
        for (int dc = 0; dc < D; dc += OW_DCH) {
            const int dend = min(dc + OW_DCH, D);
            // window origin: around the MEAN projection of the tile's pixels on the chunk's middle plane (a single pixel - say the tile
            // centre - can carry an outlier hypothesis and drag the whole window away: 8.8 % of the taps missed at stage 4 with the
            // centre pixel, 6.3 % with the mean of the 32; a second, trimmed pass changed nothing)
            int ox, oy;
            {
                float un, vn, z;
                mvs::sweep_project(rt, (float)xg, (float)yg, depth[((size_t)b * D + (dc + dend) / 2) * HW + pix], half_w, half_h, &un, &vn, &z);
                float ix = (un + 1.0f) * half_w, iy = (vn + 1.0f) * half_h;
                float ok = (fabsf(ix) < 1e6f && fabsf(iy) < 1e6f) ? 1.0f : 0.0f;     // also 0 for NaN
                ix = ok != 0.0f ? ix : 0.0f;
                iy = ok != 0.0f ? iy : 0.0f;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    ix += __shfl_xor(ix, m, 64);
                    iy += __shfl_xor(iy, m, 64);
                    ok += __shfl_xor(ok, m, 64);
                }
                // the mean of projections that are each only bounded by 1e6 px can land anywhere: clamp the origin to where a window can
                // still hold an in-image tap, so that (x0 - ox, y0 - oy) with x0 in [-2, W+1] stays far inside the signed 16-bit fields
                // packed below (a wrapped coordinate would alias into the window and be flushed OUTSIDE the image)
                ox = ok > 0.0f ? min(max((int)floorf(ix / ok) - WX / 2, -WX), W + 1) : 0;
                oy = ok > 0.0f ? min(max((int)floorf(iy / ok) - WY / 2, -WY), H + 1) : 0;
            }
            int clo = NCELL, chi = -1;                        // cells this lane touched in the chunk (the flush scans only the wave's range)
            for (int c0 = dc; c0 < dend; c0 += 2) {
                __builtin_amdgcn_wave_barrier();
                {
                    const int p = lane & 31, dd = lane >> 5;
                    const int d = min(c0 + dd, D - 1);
                    const int x = min(tx0 + (p & 7), W - 1), y = min(ty0 + (p >> 3), H - 1);
                    float un, vn, z;
                    mvs::sweep_project(rt, (float)x, (float)y, depth[((size_t)b * D + d) * HW + (size_t)y * W + x], half_w, half_h, &un, &vn, &z);
                    int x0, y0;
                    const mvs::Taps t = mvs::sweep_taps_xy(un, vn, H, W, half_w, half_h, &x0, &y0);
                    taps_o[lane] = u32x4{(unsigned)t.o00, (unsigned)t.o01, (unsigned)t.o10, (unsigned)t.o11};
                    taps_w[lane] = f32x4{t.w00, t.w01, t.w10, t.w11};
                    taps_xy[lane] = ((y0 - oy) << 16) | ((x0 - ox) & 0xFFFF);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    const int d = c0 + dd;
                    if (d >= dend) break;
                    const u32x4 o = taps_o[dd * 32 + pg];
                    const f32x4 w = taps_w[dd * 32 + pg];
                    const int xy = taps_xy[dd * 32 + pg];
                    f32x4 t4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) t4[k] = buf_load4(src, o[k] * pix_bytes + (unsigned)cb * 4u);
                    float coef[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = (CPG < 4) ? i / CPG : 0;
                        coef[i] = gp[((size_t)k * D + d) * HW] * inv_s * (1.0f / CPG);
                    }
                    f32x4 cr;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float g4 = t4[0][i] * w[0];
                        g4 = fmaf(t4[1][i], w[1], g4);
                        g4 = fmaf(t4[2][i], w[2], g4);
                        g4 = fmaf(t4[3][i], w[3], g4);
                        dref[i] = fmaf(coef[i] * wv, g4, dref[i]);
                        gip = fmaf(coef[i] * r[i], g4, gip);
                        cr[i] = coef[i] * wv * r[i];
                    }
                    const int rx = (int)(short)(xy & 0xFFFF), ry = xy >> 16;
                    // The four taps of a pixel are the 2 x 2 cells at its footprint origin (rx, ry), and all lanes handle tap k in the same
                    // instruction: two lanes meet in a cell of tap k exactly when their ORIGINS coincide - whatever k.  So ONE election per
                    // (pixel, plane) on a tag slot of the origin serializes such lanes for all four taps (round 5: it was one election per
                    // tap - four tag write / read-back round trips instead of one on a kernel that is bound by exactly those).  Different
                    // origins that share a slot (the origin may lie one cell outside the window: the slot is its hash) only wait a round.
                    int cellk[4];
                    bool inw[4];
                    bool any_in = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool want = active && w[k] != 0.0f;             // taps outside the image carry weight 0
                        const int txx = rx + (k & 1), tyy = ry + (k >> 1);
                        const bool inwin = want && (unsigned)txx < (unsigned)WX && (unsigned)tyy < (unsigned)WY;
                        const bool miss = want && !inwin;
                        const unsigned long long mm = __builtin_amdgcn_ballot_w64(miss);
                        if (mm != 0) {                                        // wave-uniform
                            const int nm = __builtin_popcountll(mm);
                            if (qn + nm > OW_QCAP) drain();
                            if (miss) {
                                const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u));
                                qoff[slot] = o[k] * (unsigned)C + (unsigned)cq * 4u;
                                qval[slot] = f32x4{w[k] * cr[0], w[k] * cr[1], w[k] * cr[2], w[k] * cr[3]};
                            }
                            qn += nm;
                        }
                        if (stats) {
                            if (want) atomicAdd(stats, 1u);
                            if (miss) atomicAdd(stats + 1, 1u);
                        }
                        cellk[k] = inwin ? (tyy << wx_log2) + txx : 0;
                        inw[k] = inwin;
                        any_in = any_in || inwin;
                        if (inwin) {
                            clo = min(clo, cellk[k]);
                            chi = max(chi, cellk[k]);
                        }
                    }
                    const int oslot = cq * NCELL + (int)(((unsigned)(ry + 1) * (unsigned)(WX + 1) + (unsigned)(rx + 1)) & omask);
                    bool pending = any_in;
                    while (__builtin_amdgcn_ballot_w64(pending) != 0) {        // wave-uniform; one round unless two lanes share an origin slot
                        if (pending) tags[oslot] = (unsigned char)lane;
                        const bool mine = pending && tags[oslot] == (unsigned char)lane;
                        if (mine) {
                            // tap by tap, each a read-modify-write of its own: lane A's tap 1 cell IS lane B's tap 0 cell when their origins are
                            // neighbours - reading all four cells first and writing them afterwards loses one of the two updates (tried: wrong)
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (inw[k]) {
                                    f32x4 v = win[cellk[k] * 2 + cq];
#pragma unroll
                                    for (int i = 0; i < 4; ++i) v[i] += w[k] * cr[i];
                                    win[cellk[k] * 2 + cq] = v;
                                }
                            pending = false;
                        }
                    }
                }
            }
            // flush this chunk's window: touched values -> global atomics, cells back to zero (wave-private: no barrier)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                clo = min(clo, __shfl_xor(clo, m, 64));
                chi = max(chi, __shfl_xor(chi, m, 64));
            }
            for (int i = 2 * clo + lane; i <= 2 * chi + 1; i += 64) {
                const f32x4 v = win[i];
                if (v[0] != 0.0f || v[1] != 0.0f || v[2] != 0.0f || v[3] != 0.0f) {
                    const int tex = i >> 1, gx = ox + (tex & (WX - 1)), gy = oy + (tex >> wx_log2);
                    float* dst = dsrc + ((size_t)gy * W + gx) * C + (i & 1) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[e] != 0.0f) atomicAdd(dst + e, v[e]);
                    win[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (qn) drain();                                      // the queue addresses this view's gradient
        gip += __shfl_xor(gip, 1, 64);
        if (active && cq == 0)
            gip_part[((size_t)(oct * gridDim.z / NOCT + b) * (V - 1) + sv) * HW + pix] = oct == 0 ? gip - tsum * inv_s : gip;
    }
    if (active) *reinterpret_cast<f32x4*>(dfeat + ((size_t)(b * V) * HW + pix) * C + cb) = dref;
}

}  // namespace

extern "C" int mvs_cv_aggregate_bwd(const float* feat, const float* rt, const float* depth, const float* weight, const float* volume,
                                    const float* gvolume, int B, int V, int C, int Gin, int D, int H, int W, float* dfeat, float* dweight,
                                    mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && weight && volume && gvolume && dfeat && dweight, "mvs_cv_aggregate_bwd: null pointer");
    MVS_REQUIRE(B >= 1 && V >= 2 && D >= 1 && H >= 1 && W >= 1 && B <= 65535 && H <= 65535, "mvs_cv_aggregate_bwd: bad shape");
    MVS_REQUIRE(Gin == G, "mvs_cv_aggregate_bwd: only G=8 correlation groups are built (got %d)", Gin);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "mvs_cv_aggregate_bwd: C must be 8, 16, 32 or 64 (got %d)", C);
    MVS_REQUIRE((int64_t)C * H * W * 4 < ((int64_t)1 << 32), "mvs_cv_aggregate_bwd: one view's feature block exceeds 4 GiB");
    const int LPP = C / 4, PPW = 64 / LPP;
    dim3 grid(mvs::ceil_div(W, NW * PPW), H, B), block(64 * NW);
    hipStream_t s = MVS_STREAM(stream);
    switch (LPP) {
        case 2: hipLaunchKernelGGL(cv_aggregate_bwd_kernel<2>, grid, block, 0, s, feat, rt, depth, weight, volume, gvolume, V, D, H, W, dfeat, dweight); break;
        case 4: hipLaunchKernelGGL(cv_aggregate_bwd_kernel<4>, grid, block, 0, s, feat, rt, depth, weight, volume, gvolume, V, D, H, W, dfeat, dweight); break;
        case 8: hipLaunchKernelGGL(cv_aggregate_bwd_kernel<8>, grid, block, 0, s, feat, rt, depth, weight, volume, gvolume, V, D, H, W, dfeat, dweight); break;
        default: hipLaunchKernelGGL(cv_aggregate_bwd_kernel<16>, grid, block, 0, s, feat, rt, depth, weight, volume, gvolume, V, D, H, W, dfeat, dweight); break;
    }
    return mvs::finish_launch("mvs_cv_aggregate_bwd");
}

// LDS-window form (see cv_aggregate_bwd_lds_kernel).  gip_part [C/8][B][V-1][H][W]: per channel octet partial d(loss)/d(vis weight);
// the caller adds the octets (octet 0 already carries the -T/S term).  window = (log2 WX, WY) texels, WX*WY*32 bytes of LDS.
// stats (optional, diagnostics): [0] += taps scattered, [1] += taps that missed the window.
extern "C" int mvs_cv_aggregate_bwd_lds(const float* feat, const float* rt, const float* depth, const float* weight, const float* volume,
                                        const float* gvolume, int B, int V, int C, int Gin, int D, int H, int W, float* dfeat,
                                        float* gip_part, int wx_log2, int wy, unsigned* stats, mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && weight && volume && gvolume && dfeat && gip_part, "mvs_cv_aggregate_bwd_lds: null pointer");
    MVS_REQUIRE(B >= 1 && V >= 2 && D >= 1 && H >= 1 && W >= 1, "mvs_cv_aggregate_bwd_lds: bad shape");
    MVS_REQUIRE(Gin == G, "mvs_cv_aggregate_bwd_lds: only G=8 correlation groups are built (got %d)", Gin);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "mvs_cv_aggregate_bwd_lds: C must be 8, 16, 32 or 64 (got %d)", C);
    MVS_REQUIRE((int64_t)C * H * W * 4 < ((int64_t)1 << 32), "mvs_cv_aggregate_bwd_lds: one view's feature block exceeds 4 GiB");
    MVS_REQUIRE(wx_log2 >= 4 && wx_log2 <= 7 && wy >= 8 && wy <= 64 && wy % 4 == 0, "mvs_cv_aggregate_bwd_lds: window %d x %d out of range",
                1 << wx_log2, wy);
    const size_t lds = (((size_t)wy << wx_log2) + 4) * 32 + (size_t)LT_NW * 64 * 36 + 16;
    MVS_REQUIRE(lds <= 64 * 1024, "mvs_cv_aggregate_bwd_lds: window needs %zu bytes of LDS (> 64 KiB)", lds);
    const int noct = C / 8;
    MVS_REQUIRE((int64_t)B * noct <= 65535 && mvs::ceil_div(H, LT_H) <= 65535, "mvs_cv_aggregate_bwd_lds: grid limits exceeded");
    dim3 grid(mvs::ceil_div(W, LT_W), mvs::ceil_div(H, LT_H), B * noct), block(64 * LT_NW);
    hipStream_t s = MVS_STREAM(stream);
#define MVS_LAUNCH_BL(CC) \
    hipLaunchKernelGGL(cv_aggregate_bwd_lds_kernel<CC>, grid, block, lds, s, feat, rt, depth, weight, volume, gvolume, V, D, H, W, dfeat, gip_part, wx_log2, wy, stats)
    switch (C) {
        case 8: MVS_LAUNCH_BL(8); break;
        case 16: MVS_LAUNCH_BL(16); break;
        case 32: MVS_LAUNCH_BL(32); break;
        default: MVS_LAUNCH_BL(64); break;
    }
#undef MVS_LAUNCH_BL
    return mvs::finish_launch("mvs_cv_aggregate_bwd_lds");
}

// Owner-election form (see cv_aggregate_bwd_own_kernel): same contract as mvs_cv_aggregate_bwd_lds; the window is per WAVEFRONT
// (8 x 4 pixels, 8 planes at a time), (1 << wx_log2) x wy texels, wx*wy*34 + 2304 bytes of LDS per wavefront.
extern "C" int mvs_cv_aggregate_bwd_own(const float* feat, const float* rt, const float* depth, const float* weight, const float* volume,
                                        const float* gvolume, int B, int V, int C, int Gin, int D, int H, int W, float* dfeat,
                                        float* gip_part, int wx_log2, int wy, unsigned* stats, mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && weight && volume && gvolume && dfeat && gip_part, "mvs_cv_aggregate_bwd_own: null pointer");
    MVS_REQUIRE(B >= 1 && V >= 2 && D >= 1 && H >= 1 && W >= 1, "mvs_cv_aggregate_bwd_own: bad shape");
    MVS_REQUIRE(Gin == G, "mvs_cv_aggregate_bwd_own: only G=8 correlation groups are built (got %d)", Gin);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "mvs_cv_aggregate_bwd_own: C must be 8, 16, 32 or 64 (got %d)", C);
    MVS_REQUIRE((int64_t)C * H * W * 4 < ((int64_t)1 << 32), "mvs_cv_aggregate_bwd_own: one view's feature block exceeds 4 GiB");
    MVS_REQUIRE(wx_log2 >= 3 && wx_log2 <= 7 && wy >= 4 && wy <= 64, "mvs_cv_aggregate_bwd_own: window %d x %d out of range", 1 << wx_log2, wy);
    const size_t per_wave = ((((size_t)wy << wx_log2) * 34 + 64 * 36 + (size_t)OW_QCAP * 20) + 15) & ~(size_t)15;
    const size_t lds = per_wave * OW_NW;
    MVS_REQUIRE(lds <= 64 * 1024, "mvs_cv_aggregate_bwd_own: window needs %zu bytes of LDS (> 64 KiB)", lds);
    const int noct = C / 8;
    MVS_REQUIRE((int64_t)B * noct <= 65535 && mvs::ceil_div(H, OW_TH) <= 65535, "mvs_cv_aggregate_bwd_own: grid limits exceeded");
    dim3 grid(mvs::ceil_div(W, OW_TW * OW_NW), mvs::ceil_div(H, OW_TH), B * noct), block(64 * OW_NW);
    hipStream_t s = MVS_STREAM(stream);
#define MVS_LAUNCH_BO(CC) \
    hipLaunchKernelGGL(cv_aggregate_bwd_own_kernel<CC>, grid, block, lds, s, feat, rt, depth, weight, volume, gvolume, V, D, H, W, dfeat, gip_part, wx_log2, wy, stats)
    switch (C) {
        case 8: MVS_LAUNCH_BO(8); break;
        case 16: MVS_LAUNCH_BO(16); break;
        case 32: MVS_LAUNCH_BO(32); break;
        default: MVS_LAUNCH_BO(64); break;
    }
#undef MVS_LAUNCH_BO
    return mvs::finish_launch("mvs_cv_aggregate_bwd_own");
}
