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
                                                      float* __restrict__ grad, float* __restrict__ rows /*[blocks][2]: loss sum, count*/,
                                                      uint8_t* __restrict__ valid_out, int* __restrict__ index_out) {
    __shared__ float red[2][4];
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
    // block reduction -> one row per block
    for (int o = 32; o > 0; o >>= 1) {
        lsum += __shfl_down(lsum, o, 64);
        cnt += __shfl_down(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = lsum;
        red[1][threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        rows[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// acc[0..1] = the rows (acc + 2) added in a fixed order; loss = weight * acc[0] / acc[1]
__global__ __launch_bounds__(256) void ce_finalize_kernel(float* __restrict__ acc, int nrows, float weight, float* __restrict__ loss) {
    __shared__ float red[2][4];
    const float* rows = acc + 2;
    float s = 0.0f, c = 0.0f;
    for (int i = threadIdx.x; i < nrows; i += 256) {
        s += rows[2 * i];
        c += rows[2 * i + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        c += __shfl_down(c, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        c = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        acc[0] = s;
        acc[1] = c;
        loss[0] = weight * (s / c);                   // N = 0 -> 0/0 = NaN, as F.cross_entropy on an empty selection
    }
}

__global__ __launch_bounds__(256) void ce_scale_kernel(const float* __restrict__ grad, float* __restrict__ out, size_t n,
                                                       const float* __restrict__ acc, const float* __restrict__ gout, float weight) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = grad[i] * (weight * gout[0] / acc[1]);
}

}  // namespace

extern "C" int64_t mvs_ce_loss_acc_floats(int B, int64_t HW) {
    return (B < 1 || HW < 1) ? -1 : 2 + 2 * (int64_t)B * ((HW + 255) / 256);
}

extern "C" int mvs_ce_loss_fwd(const float* logits, const float* depth_values, const float* depth_gt, const float* mask, int B, int D,
                               int64_t HW, int inverse_depth, float weight, float* grad_unscaled, float* acc, float* loss,
                               uint8_t* valid, int* gt_index, mvs_stream_t stream) {
    float* acc2 = acc ? acc + 2 : nullptr;                   // the block rows; acc[0..1] are written by the finalize kernel
    MVS_REQUIRE(logits && depth_values && depth_gt && mask && acc && loss, "mvs_ce_loss_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && B <= 65535 && D >= 2 && HW >= 1, "mvs_ce_loss_fwd: bad shape B=%d D=%d (>= 2: bins need an interval) HW=%lld", B, D,
                (long long)HW);
    hipStream_t s = MVS_STREAM(stream);
    dim3 grid((unsigned)mvs::ceil_div((long long)HW, 256LL), B);
    if (inverse_depth)
        hipLaunchKernelGGL(ce_loss_kernel<true>, grid, dim3(256), 0, s, logits, depth_values, depth_gt, mask, D, (size_t)HW, grad_unscaled,
                           acc2, valid, gt_index);
    else
        hipLaunchKernelGGL(ce_loss_kernel<false>, grid, dim3(256), 0, s, logits, depth_values, depth_gt, mask, D, (size_t)HW,
                           grad_unscaled, acc2, valid, gt_index);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, acc, (int)(grid.x * grid.y), weight, loss);
    return mvs::finish_launch("mvs_ce_loss_fwd");
}

extern "C" int mvs_ce_loss_bwd_scale(const float* grad_unscaled, float* grad, int64_t numel, const float* acc, const float* grad_out,
                                     float weight, mvs_stream_t stream) {
    MVS_REQUIRE(grad_unscaled && grad && acc && grad_out && numel >= 1, "mvs_ce_loss_bwd_scale: bad arguments");
    hipLaunchKernelGGL(ce_scale_kernel, dim3((unsigned)mvs::ceil_div((long long)numel, 256LL)), dim3(256), 0, MVS_STREAM(stream),
                       grad_unscaled, grad, (size_t)numel, acc, grad_out, weight);
    return mvs::finish_launch("mvs_ce_loss_bwd_scale");
}
