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
