// Fused plane-sweep cost-volume build for fusion_type='cnn' — the inline block of StageNet.forward,
// models/mvsformer_model.py:62-105 — without ever materializing the warped volume [B,C,D,H,W], the repeated
// reference volume, their product or the normalized copies the reference's eval branch makes.
//
// Data layout: everything stays NCHW / NCDHW exactly as the FPN decoder hands it over and as the 3-D
// regularizer consumes it, so there is no transpose pass.  One lane owns one reference pixel; the 64 lanes of
// a wavefront are 64 consecutive pixels of one image row, so for a fixed channel the four bilinear taps of
// the wavefront are four (nearly) contiguous row segments of the source plane — coalesced gathers that hit
// L1/L2 (a source map is 7-57 MB; neighbouring depths/rows re-touch the same lines), and the volume store
// is one contiguous 256-B row segment per (group, depth).  threadIdx.y splits the depth hypotheses of the same
// 64 pixels across the wavefronts of a block (coarse stages have few pixels but many depths).
//
// The reference feature vector of the pixel (C <= 64 floats) lives in registers for the whole sweep; the
// per-depth tap offsets/weights are computed once and reused over the C channels.  Group sums over the C/G
// channels of a group are in-lane (no cross-lane traffic at all).
//
// Sweep A (cv_entropy): per source view, correlation -> sim[d] = sum_g in_prod[g,d] -> entropy of softmax_d.
// Sweep B (cv_aggregate): recomputes the correlation for all source views (cheaper than storing (V-1)
//   [B,G,D,H,W] volumes for the fine stages), accumulates sum_v w_v*in_prod_v in registers, writes
//   volume_mean once; also the eval-only similarity arg-max depth.
//
// Algorithmic HBM bytes per stage: 4*H*W*(V*C + D + G*D)  (features once, hypotheses once, volume once).
#include "common.h"
#include "geometry.h"

namespace {

constexpr int G = 8;

template <int CPG>
__global__ __launch_bounds__(512) void cv_entropy_kernel(const float* __restrict__ feat, const float* __restrict__ rt_all,
                                                         const float* __restrict__ depth, int V, int D, int H, int W,
                                                         float* __restrict__ entropy) {
    constexpr int C = G * CPG;
    extern __shared__ float sims[];                      // [D][64]
    const int tx = threadIdx.x, ty = threadIdx.y, DS = blockDim.y;
    const int x = blockIdx.x * 64 + tx, y = blockIdx.y;
    const int b = blockIdx.z / (V - 1), sv = blockIdx.z % (V - 1);
    const bool active = x < W;
    const int xc = active ? x : W - 1;
    const size_t HW = (size_t)H * W;
    const float* ref = feat + (size_t)(b * V) * C * HW + (size_t)y * W + xc;
    const float* src = feat + (size_t)(b * V + sv + 1) * C * HW;
    const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);

    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = ref[(size_t)c * HW];

    for (int d = ty; d < D; d += DS) {
        const float dv = depth[((size_t)(b * D + d) * H + y) * W + xc];
        float un, vn, z;
        mvs::sweep_project(rt, (float)xc, (float)y, dv, half_w, half_h, &un, &vn, &z);
        const mvs::Taps t = mvs::sweep_taps(un, vn, H, W, half_w, half_h);
        float sim = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < CPG; ++j) {
                const int c = g * CPG + j;
                s = s + r[c] * mvs::bilinear(src + (size_t)c * HW, t);
            }
            sim = sim + s * (1.0f / CPG);
        }
        sims[d * 64 + tx] = sim;
    }
    __syncthreads();
    if (ty == 0 && active) {
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) m = fmaxf(m, sims[d * 64 + tx]);
        float sum = 0.0f;
        for (int d = 0; d < D; ++d) sum += expf(sims[d * 64 + tx] - m);
        float ent = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float p = expf(sims[d * 64 + tx] - m) / sum;
            ent = ent + (-p) * logf(p + 1e-7f);
        }
        entropy[((size_t)(b * (V - 1) + sv) * H + y) * W + x] = ent;
    }
}

template <int CPG, bool SIM>
__global__ __launch_bounds__(512) void cv_aggregate_kernel(const float* __restrict__ feat, const float* __restrict__ rt_all,
                                                           const float* __restrict__ depth, const float* __restrict__ weight,
                                                           int V, int D, int H, int W,
                                                           float* __restrict__ volume, float* __restrict__ sim_depth) {
    constexpr int C = G * CPG;
    extern __shared__ float red[];                       // SIM: [DS][64] best value, [DS][64] best index
    const int tx = threadIdx.x, ty = threadIdx.y, DS = blockDim.y;
    const int x = blockIdx.x * 64 + tx, y = blockIdx.y, b = blockIdx.z;
    const bool active = x < W;
    const int xc = active ? x : W - 1;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)y * W + xc;
    const float* ref = feat + (size_t)(b * V) * C * HW + pix;
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);

    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = ref[(size_t)c * HW];
    float rinv[CPG];
    if (SIM) {
#pragma unroll
        for (int j = 0; j < CPG; ++j) {
            float n2 = 0.0f;
#pragma unroll
            for (int g = 0; g < G; ++g) n2 = fmaf(r[g * CPG + j], r[g * CPG + j], n2);
            rinv[j] = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        }
    }
    const float* wp = weight + (size_t)(b * (V - 1)) * HW + pix;
    float wsum = 0.0f;
    for (int sv = 0; sv < V - 1; ++sv) wsum = wsum + wp[(size_t)sv * HW];
    const float denom = wsum + 1e-6f;

    float best = -INFINITY;
    int besti = 0;
    for (int d = ty; d < D; d += DS) {
        const float dv = depth[((size_t)(b * D + d) * H + y) * W + xc];
        float acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = 0.0f;
        float simtot = 0.0f;
        for (int sv = 0; sv < V - 1; ++sv) {
            const float* rt = rt_all + (size_t)(b * (V - 1) + sv) * 12;
            const float* src = feat + (size_t)(b * V + sv + 1) * C * HW;
            const float wv = wp[(size_t)sv * HW];
            float un, vn, z;
            mvs::sweep_project(rt, (float)xc, (float)y, dv, half_w, half_h, &un, &vn, &z);
            const mvs::Taps t = mvs::sweep_taps(un, vn, H, W, half_w, half_h);
            float dotj[CPG], nrm[CPG];
#pragma unroll
            for (int j = 0; j < CPG; ++j) { dotj[j] = 0.0f; nrm[j] = 0.0f; }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < CPG; ++j) {
                    const int c = g * CPG + j;
                    const float w = mvs::bilinear(src + (size_t)c * HW, t);
                    const float p = r[c] * w;
                    s = s + p;
                    if (SIM) {
                        dotj[j] = dotj[j] + p;
                        nrm[j] = fmaf(w, w, nrm[j]);
                    }
                }
                acc[g] = acc[g] + (s * (1.0f / CPG)) * wv;
            }
            if (SIM) {
                float ssum = 0.0f;
#pragma unroll
                for (int j = 0; j < CPG; ++j) ssum = ssum + (dotj[j] * rinv[j]) / fmaxf(sqrtf(nrm[j]), 1e-12f);
                simtot = simtot + ssum * (1.0f / CPG);
            }
        }
        if (active) {
            float* vp = volume + ((size_t)(b * G) * D + d) * HW + pix;
#pragma unroll
            for (int g = 0; g < G; ++g) vp[(size_t)g * D * HW] = acc[g] / denom;
        }
        if (SIM && simtot > best) { best = simtot; besti = d; }
    }
    if (SIM) {
        red[ty * 64 + tx] = best;
        red[(DS + ty) * 64 + tx] = __int_as_float(besti);
        __syncthreads();
        if (ty == 0 && active) {
            for (int s = 1; s < DS; ++s) {
                const float v = red[s * 64 + tx];
                const int i = __float_as_int(red[(DS + s) * 64 + tx]);
                if (v > best || (v == best && i < besti)) { best = v; besti = i; }
            }
            sim_depth[(size_t)b * HW + pix] = depth[((size_t)(b * D + besti) * H + y) * W + x];
        }
    }
}

int pick_depth_slices(int D) {
    int ds = D / 2;
    if (ds < 1) ds = 1;
    if (ds > 8) ds = 8;
    return ds;
}

int check_shapes(const char* who, int B, int V, int C, int Gin, int D, int H, int W) {
    MVS_REQUIRE(B >= 1 && V >= 2 && D >= 1 && H >= 1 && W >= 1, "%s: bad shape B=%d V=%d D=%d H=%d W=%d", who, B, V, D, H, W);
    MVS_REQUIRE(Gin == G, "%s: only G=8 correlation groups are built (got %d)", who, Gin);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "%s: C must be 8, 16, 32 or 64 (got %d)", who, C);
    MVS_REQUIRE((int64_t)B * (V - 1) <= 65535 && H <= 65535, "%s: grid limits exceeded", who);
    MVS_REQUIRE((int64_t)D * 64 * 4 <= 64 * 1024, "%s: D=%d needs more than 64 KiB of LDS", who, D);
    return MVS_OK;
}

}  // namespace

extern "C" int mvs_cv_entropy_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int Gin, int D,
                                  int H, int W, float* entropy, mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && entropy, "mvs_cv_entropy_fwd: null pointer");
    if (int rc = check_shapes("mvs_cv_entropy_fwd", B, V, C, Gin, D, H, W)) return rc;
    const int DS = pick_depth_slices(D);
    dim3 grid(mvs::ceil_div(W, 64), H, B * (V - 1)), block(64, DS);
    const size_t lds = (size_t)D * 64 * sizeof(float);
    hipStream_t s = MVS_STREAM(stream);
    switch (C / G) {
        case 1: hipLaunchKernelGGL(cv_entropy_kernel<1>, grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy); break;
        case 2: hipLaunchKernelGGL(cv_entropy_kernel<2>, grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy); break;
        case 4: hipLaunchKernelGGL(cv_entropy_kernel<4>, grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy); break;
        default: hipLaunchKernelGGL(cv_entropy_kernel<8>, grid, block, lds, s, feat, rt, depth, V, D, H, W, entropy); break;
    }
    return mvs::finish_launch("mvs_cv_entropy_fwd");
}

extern "C" int mvs_cv_aggregate_fwd(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V,
                                    int C, int Gin, int D, int H, int W, float* volume, float* sim_depth,
                                    mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && weight && volume, "mvs_cv_aggregate_fwd: null pointer");
    if (int rc = check_shapes("mvs_cv_aggregate_fwd", B, V, C, Gin, D, H, W)) return rc;
    const int DS = pick_depth_slices(D);
    dim3 grid(mvs::ceil_div(W, 64), H, B), block(64, DS);
    hipStream_t s = MVS_STREAM(stream);
    const size_t lds = sim_depth ? (size_t)2 * DS * 64 * sizeof(float) : 0;
#define MVS_LAUNCH_AGG(CPG)                                                                                              \
    if (sim_depth)                                                                                                       \
        hipLaunchKernelGGL((cv_aggregate_kernel<CPG, true>), grid, block, lds, s, feat, rt, depth, weight, V, D, H, W,  \
                           volume, sim_depth);                                                                           \
    else                                                                                                                 \
        hipLaunchKernelGGL((cv_aggregate_kernel<CPG, false>), grid, block, lds, s, feat, rt, depth, weight, V, D, H, W, \
                           volume, sim_depth)
    switch (C / G) {
        case 1: MVS_LAUNCH_AGG(1); break;
        case 2: MVS_LAUNCH_AGG(2); break;
        case 4: MVS_LAUNCH_AGG(4); break;
        default: MVS_LAUNCH_AGG(8); break;
    }
#undef MVS_LAUNCH_AGG
    return mvs::finish_launch("mvs_cv_aggregate_fwd");
}
