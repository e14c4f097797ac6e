// Cross-entropy depth loss fused with the classification head's gradient — SURVEY.md §8(f3), reference
// models/losses.py:304-350 (ce_loss_stage4, focal=False), one stage.
//
// The reference flips hypotheses and logits along depth (inverse-depth sampling runs far -> near), builds the
// half-interval bin edges as [B,D,H,W] tensors, gathers the masked pixels into an [N,D] matrix and calls
// F.cross_entropy.  Here one lane owns one pixel: a first walk over depth (in flipped order) finds the ground-truth bin,
// the range test and an online log-sum-exp; a second walk (logits still in L2) writes (softmax - onehot) * valid as the
// unnormalized gradient.  Two floats per launch (sum of -log p[gt], number of valid pixels) are reduced with
// wave shuffles + two atomics per wave; the mean and the 1/N scaling happen in mvs_ce_loss_finalize / _bwd_scale.
#include "common.h"

namespace {

template <bool INVERSE>
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, const float* __restrict__ hyp,
                                                      const float* __restrict__ gt, const float* __restrict__ mask, int D, size_t HW,
                                                      float* __restrict__ grad, float* __restrict__ acc /*[2]: loss sum, count*/,
                                                      uint8_t* __restrict__ valid_out, int* __restrict__ index_out) {
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    float lsum = 0.0f, cnt = 0.0f;
    if (pix < HW) {
        const float* lg = logits + (size_t)b * D * HW + pix;
        const float* hv = hyp + (size_t)b * D * HW + pix;
        const float g = gt[(size_t)b * HW + pix];
        auto at = [&](const float* p, int j) { return p[(size_t)(INVERSE ? D - 1 - j : j) * HW]; };
        // walk 1: bin index = #(right edges <= gt), range limits, online max / sum of exp
        float prev = at(hv, 0), cur = at(hv, 1);
        float itv = fabsf(cur - prev) / 2.0f;                                // intervals[0]
        const float lo = prev - itv;
        int idx = 0;
        float m = -INFINITY, s = 0.0f, hi = 0.0f;
        for (int j = 0; j < D; ++j) {
            // intervals[j] = |dv[j+1]-dv[j]|/2, last one repeated
            if (j < D - 1) {
                cur = at(hv, j + 1);
                itv = fabsf(cur - prev) / 2.0f;
            }
            idx += (prev + itv <= g) ? 1 : 0;
            if (j == D - 1) hi = prev + itv;
            prev = cur;
            const float l = at(lg, j);
            const float mn = fmaxf(m, l);
            s = s * expf(m - mn) + expf(l - mn);
            m = mn;
        }
        idx = min(idx, D - 1);
        const bool in_range = !(g < lo) && !(g > hi);
        const bool valid = in_range && (mask[(size_t)b * HW + pix] > 0.5f);
        const float lse = m + logf(s);
        if (valid) {
            lsum = lse - at(lg, idx);
            cnt = 1.0f;
        }
        if (valid_out) valid_out[(size_t)b * HW + pix] = valid ? 1 : 0;
        if (index_out) index_out[(size_t)b * HW + pix] = idx;
        // walk 2: unnormalized gradient, in the ORIGINAL depth order of `logits`
        if (grad) {
            float* gr = grad + (size_t)b * D * HW + pix;
            const int hot = INVERSE ? D - 1 - idx : idx;
            for (int d = 0; d < D; ++d) {
                const float p = expf(lg[(size_t)d * HW] - lse);
                gr[(size_t)d * HW] = valid ? (p - (d == hot ? 1.0f : 0.0f)) : 0.0f;
            }
        }
    }
    // block reduction -> 2 atomics per wave
    for (int o = 32; o > 0; o >>= 1) {
        lsum += __shfl_down(lsum, o, 64);
        cnt += __shfl_down(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt > 0.0f) {
        atomicAdd(acc, lsum);
        atomicAdd(acc + 1, cnt);
    }
}

__global__ void ce_finalize_kernel(const float* __restrict__ acc, float weight, float* __restrict__ loss) {
    loss[0] = weight * (acc[0] / acc[1]);             // N = 0 -> 0/0 = NaN, as F.cross_entropy on an empty selection
}

__global__ __launch_bounds__(256) void ce_scale_kernel(float* __restrict__ grad, size_t n, const float* __restrict__ acc,
                                                       const float* __restrict__ gout, float weight) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) grad[i] *= weight * gout[0] / acc[1];
}

}  // namespace

extern "C" int mvs_ce_loss_fwd(const float* logits, const float* depth_values, const float* depth_gt, const float* mask, int B, int D,
                               int64_t HW, int inverse_depth, float weight, float* grad_unscaled, float* acc2, float* loss,
                               uint8_t* valid, int* gt_index, mvs_stream_t stream) {
    MVS_REQUIRE(logits && depth_values && depth_gt && mask && acc2 && loss, "mvs_ce_loss_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && B <= 65535 && D >= 2 && HW >= 1, "mvs_ce_loss_fwd: bad shape B=%d D=%d (>= 2: bins need an interval) HW=%lld", B, D,
                (long long)HW);
    hipStream_t s = MVS_STREAM(stream);
    if (hipMemsetAsync(acc2, 0, 2 * sizeof(float), s) != hipSuccess) return mvs::finish_launch("mvs_ce_loss_fwd(memset)");
    dim3 grid((unsigned)mvs::ceil_div((long long)HW, 256LL), B);
    if (inverse_depth)
        hipLaunchKernelGGL(ce_loss_kernel<true>, grid, dim3(256), 0, s, logits, depth_values, depth_gt, mask, D, (size_t)HW, grad_unscaled,
                           acc2, valid, gt_index);
    else
        hipLaunchKernelGGL(ce_loss_kernel<false>, grid, dim3(256), 0, s, logits, depth_values, depth_gt, mask, D, (size_t)HW,
                           grad_unscaled, acc2, valid, gt_index);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1), 0, s, acc2, weight, loss);
    return mvs::finish_launch("mvs_ce_loss_fwd");
}

extern "C" int mvs_ce_loss_bwd_scale(float* grad_inplace, int64_t numel, const float* acc2, const float* grad_out, float weight,
                                     mvs_stream_t stream) {
    MVS_REQUIRE(grad_inplace && acc2 && grad_out && numel >= 1, "mvs_ce_loss_bwd_scale: bad arguments");
    hipLaunchKernelGGL(ce_scale_kernel, dim3((unsigned)mvs::ceil_div((long long)numel, 256LL)), dim3(256), 0, MVS_STREAM(stream),
                       grad_inplace, (size_t)numel, acc2, grad_out, weight);
    return mvs::finish_launch("mvs_ce_loss_bwd_scale");
}
