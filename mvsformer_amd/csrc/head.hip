// Per-pixel depth-axis sweeps around the regularizer: the softmax head (models/mvsformer_model.py:110-125 with
// depth_regression, models/module.py:597-603), the inverse-depth hypothesis schedulers (module.py:633-653) and
// the confidence accumulation of the cascade loop (mvsformer_model.py:297-301).  All pure bandwidth: one lane
// per pixel along W, every [B,D,H,W] plane access is a coalesced 256-B row segment per wavefront.
#include "common.h"

namespace {

// logits come either from memory (CostRegNet: 3x3x3 prob conv ran before) or from the fused 1x1x1 conv of
// CostRegNet3D.prob over the C-channel regularizer output x8 (weights via the scalar cache).
template <bool FUSED_CONV>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ logits, const float* __restrict__ x8,
                                                   const float* __restrict__ w1, const float* __restrict__ b1, int C,
                                                   const float* __restrict__ depth_values, float tmp, int training, int D, int H, int W,
                                                   float* __restrict__ pre_out, float* __restrict__ prob, float* __restrict__ depth,
                                                   float* __restrict__ conf) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const size_t base = (size_t)b * D * plane + pix;
    const float* lp = FUSED_CONV ? pre_out + base : logits + base;

    // pass 1: logits (optionally computed + stored), running maxima of l and l*tmp
    float m = -INFINITY, mt = -INFINITY;
    for (int d = 0; d < D; ++d) {
        float l;
        if (FUSED_CONV) {
            const float* xp = x8 + (size_t)b * C * D * plane + (size_t)d * plane + pix;
            float acc = 0.0f;
            for (int c = 0; c < C; ++c) acc = fmaf(w1[c], xp[(size_t)c * D * plane], acc);
            l = acc + b1[0];
            pre_out[base + (size_t)d * plane] = l;
        } else {
            l = lp[(size_t)d * plane];
        }
        m = fmaxf(m, l);
        mt = fmaxf(mt, l * tmp);
    }
    // pass 2: partition sums
    float s = 0.0f, st = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float l = lp[(size_t)d * plane];
        s += expf(l - m);
        st += expf(l * tmp - mt);
    }
    // pass 3: outputs
    float pmax = -INFINITY, reg = 0.0f;
    int arg = 0;
    for (int d = 0; d < D; ++d) {
        const float l = lp[(size_t)d * plane];
        const float p = expf(l - m) / s;
        prob[base + (size_t)d * plane] = p;
        if (p > pmax) { pmax = p; arg = d; }
        if (!training) {
            const float pt = expf(l * tmp - mt) / st;
            reg = reg + pt * depth_values[base + (size_t)d * plane];
        }
    }
    depth[(size_t)b * plane + pix] = training ? depth_values[base + (size_t)arg * plane] : reg;
    conf[(size_t)b * plane + pix] = pmax;
}

// The same head for the depth counts the cascade uses (D = 4 | 8 | 16 | 32) with the logits in memory: ALL of a pixel's logits and
// hypotheses are requested before the first is consumed, then the three sweeps run from registers.  The generic kernel above walks the
// depth axis three times with one dependent load per step - at stage 1 (27 648 pixels = 432 wavefronts, D = 32) that is 96 memory
// latencies in a row: 34 us for 3.5 MB.  Same operations in the same order on the same values: bit-identical outputs.
template <int DT>
__global__ __launch_bounds__(256) void head_reg_kernel(const float* __restrict__ logits, const float* __restrict__ depth_values, float tmp, int training,
                                                       int H, int W, float* __restrict__ prob, float* __restrict__ depth, float* __restrict__ conf) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, b = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const size_t base = (size_t)b * DT * plane + pix;
    float l[DT], dv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) l[d] = logits[base + (size_t)d * plane];
#pragma unroll
    for (int d = 0; d < DT; ++d) dv[d] = depth_values[base + (size_t)d * plane];
    float m = -INFINITY, mt = -INFINITY;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        m = fmaxf(m, l[d]);
        mt = fmaxf(mt, l[d] * tmp);
    }
    float e[DT], et[DT];
    float s = 0.0f, st = 0.0f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        e[d] = expf(l[d] - m);
        et[d] = expf(l[d] * tmp - mt);
        s += e[d];
        st += et[d];
    }
    float pmax = -INFINITY, reg = 0.0f;
    int arg = 0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const float p = e[d] / s;
        prob[base + (size_t)d * plane] = p;
        if (p > pmax) { pmax = p; arg = d; }
        if (!training) reg = reg + (et[d] / st) * dv[d];
    }
    float darg = dv[0];
#pragma unroll
    for (int d = 1; d < DT; ++d) darg = arg == d ? dv[d] : darg;
    depth[(size_t)b * plane + pix] = training ? darg : reg;
    conf[(size_t)b * plane + pix] = pmax;
}

__global__ void init_inverse_kernel(const float* __restrict__ range, int N, int D, int H, int W, float* __restrict__ hyp) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / D, d = blockIdx.z % D;
    if (x >= W || y >= H) return;
    // module.py:634-639: 1/(1/far + (1/near - 1/far) * d/(D-1)), near = range[:,0], far = range[:,-1]
    const float inv_near = 1.0f / range[(size_t)b * N];
    const float inv_far = 1.0f / range[(size_t)b * N + N - 1];
    const float itv = (float)d / (float)(D - 1);
    const float inv = inv_far + (inv_near - inv_far) * itv;
    hyp[((size_t)(b * D + d) * H + y) * W + x] = 1.0f / inv;
}

// module.py:642-653.  The low-resolution inverse-depth samples are generated on the fly from the previous
// stage's depth map and its hypothesis planes 1 and 2, then trilinearly upsampled (align_corners=True; the
// depth axis has equal input and output size, so its interpolation weight is exactly 0) and inverted.
__global__ void schedule_inverse_kernel(const float* __restrict__ prev_depth, const float* __restrict__ prev_hyp, int Dp,
                                        float split_itv, int D, int H, int W, float* __restrict__ hyp) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    if (x >= W || y >= H) return;
    const int Hl = H / 2, Wl = W / 2;
    const float sy = (H > 1) ? (float)(Hl - 1) / (float)(H - 1) : 0.0f;
    const float sx = (W > 1) ? (float)(Wl - 1) / (float)(W - 1) : 0.0f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < Hl - 1) ? 1 : 0), x1 = x0 + ((x0 < Wl - 1) ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const size_t lplane = (size_t)Hl * Wl;
    const float* pd = prev_depth + (size_t)b * lplane;
    const float* h1 = prev_hyp + ((size_t)b * Dp + 1) * lplane;
    const float* h2 = prev_hyp + ((size_t)b * Dp + 2) * lplane;
    // the four low-resolution neighbours' (inv_max, inv_min - inv_max): the 12 divisions happen once per pixel, not per plane
    float lo[4], span[4];
    const int ys[4] = {y0, y0, y1, y1}, xs[4] = {x0, x1, x0, x1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t o = (size_t)ys[k] * Wl + xs[k];
        const float last = 1.0f / h2[o] - 1.0f / h1[o];
        const float inv_d = 1.0f / pd[o];
        const float inv_min = inv_d + split_itv * last;
        lo[k] = inv_d - split_itv * last;
        span[k] = inv_min - lo[k];
    }
    const size_t plane = (size_t)H * W;
    float* out = hyp + (size_t)b * D * plane + (size_t)y * W + x;
    for (int d = 0; d < D; ++d) {
        const float itv = (float)d / (float)(D - 1);
        const float top = lx0 * (lo[0] + span[0] * itv) + lx1 * (lo[1] + span[1] * itv);
        const float bot = lx0 * (lo[2] + span[2] * itv) + lx1 * (lo[3] + span[3] * itv);
        const float inv = ly0 * top + ly1 * bot;
        out[(size_t)d * plane] = 1.0f / inv;
    }
}

__global__ void conf_accumulate_kernel(const float* __restrict__ conf, int H, int W, float* __restrict__ acc, int Hf, int Wf,
                                       float weight) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= Wf || y >= Hf) return;
    // F.interpolate(mode='nearest'): src = floor(dst * in/out)
    const int sy = min((int)floorf((float)y * ((float)H / (float)Hf)), H - 1);
    const int sx = min((int)floorf((float)x * ((float)W / (float)Wf)), W - 1);
    const size_t o = ((size_t)b * Hf + y) * Wf + x;
    acc[o] = acc[o] + conf[((size_t)b * H + sy) * W + sx] * weight;
}

// stand-alone depth_regression (module.py:597-603): sum_d p*depth_values, depth_values [B,D,H,W] or [B,D]
__global__ void depth_regression_kernel(const float* __restrict__ p, const float* __restrict__ dv, int per_pixel, int D, int H, int W,
                                        float* __restrict__ out) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x, base = (size_t)b * D * plane + pix;
    float acc = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float v = per_pixel ? dv[base + (size_t)d * plane] : dv[(size_t)b * D + d];
        acc = acc + p[base + (size_t)d * plane] * v;
    }
    out[(size_t)b * plane + pix] = acc;
}

// conf_regression (module.py:606-619): sum of the n probabilities around floor(sum_d p*d)
__global__ void conf_regression_kernel(const float* __restrict__ p, int n, int D, int H, int W, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x, base = (size_t)b * D * plane + pix;
    float e = 0.0f;
    for (int d = 0; d < D; ++d) e = e + p[base + (size_t)d * plane] * (float)d;
    int idx = (int)e;                                       // .long() truncates toward zero
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    const int lo = (n % 2 == 1) ? n / 2 : n / 2 - 1;        // zero padding: lo planes in front, n/2 behind
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) {
        const int d = idx - lo + k;
        if (d >= 0 && d < D) acc += p[base + (size_t)d * plane];
    }
    // the reference computes n * avg_pool(window): (sum / n) * n
    out[(size_t)b * plane + pix] = (float)n * (acc / (float)n);
}

// depth_type == 'mixup_ce' head (mvsformer_model.py:126-136): over adjacent hypothesis pairs (d, d+1) take the pair with the largest
// p[d]+p[d+1] (first maximum, torch.max), confidence = that sum, depth = the pair's hypotheses mixed by the renormalised pair
// probabilities p/(p[d]+p[d+1]+1e-7) -- same operation order as the reference (two divisions, two products, one sum).
__global__ void mixup_head_kernel(const float* __restrict__ p, const float* __restrict__ dv, int D, int H, int W,
                                  float* __restrict__ depth, float* __restrict__ conf) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x, base = (size_t)b * D * plane + pix;
    float left = p[base], best = -1.0f, bl = 0.0f, br = 0.0f;
    int idx = 0;
    for (int d = 0; d + 1 < D; ++d) {
        const float right = p[base + (size_t)(d + 1) * plane];
        const float s = left + right;
        if (s > best) { best = s; idx = d; bl = left; br = right; }
        left = right;
    }
    const float norm = (bl + br) + 1e-7f;
    const float dl = dv[base + (size_t)idx * plane], dr = dv[base + (size_t)(idx + 1) * plane];
    depth[(size_t)b * plane + pix] = dl * (bl / norm) + dr * (br / norm);
    conf[(size_t)b * plane + pix] = best;
}

// CostRegNet3D.prob on its own: 1x1x1 conv C -> 1 with bias over [B,C,N] voxels
__global__ void prob1_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int C, size_t N,
                             float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const float* xp = x + (size_t)b * C * N + i;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) acc = fmaf(w[c], xp[(size_t)c * N], acc);
    out[(size_t)b * N + i] = acc + (bias ? bias[0] : 0.0f);
}

}  // namespace

extern "C" int mvs_depth_regression(const float* p, const float* depth_values, int depth_per_pixel, int B, int D, int H, int W,
                                    float* depth, mvs_stream_t stream) {
    MVS_REQUIRE(p && depth_values && depth, "mvs_depth_regression: null pointer");
    MVS_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W >= 1 && B <= 65535, "mvs_depth_regression: bad shape");
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B), block(64, 4);
    hipLaunchKernelGGL(depth_regression_kernel, grid, block, 0, MVS_STREAM(stream), p, depth_values, depth_per_pixel, D, H, W, depth);
    return mvs::finish_launch("mvs_depth_regression");
}

extern "C" int mvs_conf_regression(const float* p, int n, int B, int D, int H, int W, float* conf, mvs_stream_t stream) {
    MVS_REQUIRE(p && conf, "mvs_conf_regression: null pointer");
    MVS_REQUIRE(n >= 1 && B >= 1 && D >= 1 && H >= 1 && W >= 1 && B <= 65535, "mvs_conf_regression: bad shape");
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B), block(64, 4);
    hipLaunchKernelGGL(conf_regression_kernel, grid, block, 0, MVS_STREAM(stream), p, n, D, H, W, conf);
    return mvs::finish_launch("mvs_conf_regression");
}

extern "C" int mvs_mixup_head(const float* p, const float* depth_values, int B, int D, int H, int W, float* depth, float* conf,
                             mvs_stream_t stream) {
    MVS_REQUIRE(p && depth_values && depth && conf, "mvs_mixup_head: null pointer");
    MVS_REQUIRE(B >= 1 && D >= 2 && H >= 1 && W >= 1 && B <= 65535, "mvs_mixup_head: bad shape (needs D >= 2)");
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B), block(64, 4);
    hipLaunchKernelGGL(mixup_head_kernel, grid, block, 0, MVS_STREAM(stream), p, depth_values, D, H, W, depth, conf);
    return mvs::finish_launch("mvs_mixup_head");
}

extern "C" int mvs_prob1_fwd(const float* x, const float* w, const float* bias, int B, int C, int64_t N, float* logits,
                             mvs_stream_t stream) {
    MVS_REQUIRE(x && w && logits, "mvs_prob1_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && C >= 1 && N >= 1 && B <= 65535, "mvs_prob1_fwd: bad shape");
    dim3 grid((unsigned)((N + 255) / 256), B);
    hipLaunchKernelGGL(prob1_kernel, grid, dim3(256), 0, MVS_STREAM(stream), x, w, bias, C, (size_t)N, logits);
    return mvs::finish_launch("mvs_prob1_fwd");
}

extern "C" int mvs_head_fwd(const float* logits, const float* x8, const float* w1, const float* b1, int x8_channels,
                            const float* depth_values, float tmp, int training, int B, int D, int H, int W, float* prob_volume_pre,
                            float* prob_volume, float* depth, float* conf, mvs_stream_t stream) {
    MVS_REQUIRE(depth_values && prob_volume && depth && conf, "mvs_head_fwd: null output/depth pointer");
    MVS_REQUIRE((logits != nullptr) != (x8 != nullptr), "mvs_head_fwd: give exactly one of logits / x8");
    MVS_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W >= 1 && B <= 65535, "mvs_head_fwd: bad shape");
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B), block(64, 4);
    hipStream_t s = MVS_STREAM(stream);
    if (x8) {
        MVS_REQUIRE(w1 && b1 && prob_volume_pre && x8_channels >= 1, "mvs_head_fwd: fused 1x1x1 conv needs w1, b1, prob_volume_pre");
        hipLaunchKernelGGL(head_kernel<true>, grid, block, 0, s, logits, x8, w1, b1, x8_channels, depth_values, tmp, training, D, H, W,
                           prob_volume_pre, prob_volume, depth, conf);
    } else if (D == 4 || D == 8 || D == 16 || D == 32) {
        // small maps: one wavefront per block so that a few hundred wavefronts spread over all CUs
        const bool small = (int64_t)H * W * B < 256 * 1024;
        dim3 g(mvs::ceil_div(W, 64), small ? H : mvs::ceil_div(H, 4), B), bl(64, small ? 1 : 4);
#define MVS_HEAD_REG(DT) hipLaunchKernelGGL(head_reg_kernel<DT>, g, bl, 0, s, logits, depth_values, tmp, training, H, W, prob_volume, depth, conf)
        if (D == 4) MVS_HEAD_REG(4); else if (D == 8) MVS_HEAD_REG(8); else if (D == 16) MVS_HEAD_REG(16); else MVS_HEAD_REG(32);
#undef MVS_HEAD_REG
    } else {
        hipLaunchKernelGGL(head_kernel<false>, grid, block, 0, s, logits, x8, w1, b1, x8_channels, depth_values, tmp, training, D, H, W,
                           prob_volume_pre, prob_volume, depth, conf);
    }
    return mvs::finish_launch("mvs_head_fwd");
}

extern "C" int mvs_init_inverse_range(const float* depth_range, int N, int B, int D, int H, int W, float* hyp, mvs_stream_t stream) {
    MVS_REQUIRE(depth_range && hyp, "mvs_init_inverse_range: null pointer");
    MVS_REQUIRE(N >= 1 && B >= 1 && D >= 2 && H >= 1 && W >= 1 && (int64_t)B * D <= 65535, "mvs_init_inverse_range: bad shape");
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B * D), block(64, 4);
    hipLaunchKernelGGL(init_inverse_kernel, grid, block, 0, MVS_STREAM(stream), depth_range, N, D, H, W, hyp);
    return mvs::finish_launch("mvs_init_inverse_range");
}

extern "C" int mvs_schedule_inverse_range(const float* prev_depth, const float* prev_hyp, int Dp, float split_itv, int B, int D,
                                          int H, int W, float* hyp, mvs_stream_t stream) {
    MVS_REQUIRE(prev_depth && prev_hyp && hyp, "mvs_schedule_inverse_range: null pointer");
    MVS_REQUIRE(Dp >= 3, "mvs_schedule_inverse_range: previous stage needs >= 3 hypotheses (got %d)", Dp);
    MVS_REQUIRE(B >= 1 && D >= 2 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && (int64_t)B * D <= 65535,
                "mvs_schedule_inverse_range: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B), block(64, 4);
    hipLaunchKernelGGL(schedule_inverse_kernel, grid, block, 0, MVS_STREAM(stream), prev_depth, prev_hyp, Dp, split_itv, D, H, W, hyp);
    return mvs::finish_launch("mvs_schedule_inverse_range");
}

extern "C" int mvs_conf_accumulate(const float* conf, int B, int H, int W, float* acc, int Hf, int Wf, float weight,
                                   mvs_stream_t stream) {
    MVS_REQUIRE(conf && acc, "mvs_conf_accumulate: null pointer");
    MVS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Hf >= 1 && Wf >= 1 && B <= 65535, "mvs_conf_accumulate: bad shape");
    dim3 grid(mvs::ceil_div(Wf, 64), mvs::ceil_div(Hf, 4), B), block(64, 4);
    hipLaunchKernelGGL(conf_accumulate_kernel, grid, block, 0, MVS_STREAM(stream), conf, H, W, acc, Hf, Wf, weight);
    return mvs::finish_launch("mvs_conf_accumulate");
}
