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

// models/losses.py:353-408 (mixup_ce_loss_stage4), one stage: the ground truth lies between hypotheses idx and idx + 1 (flipped order);
// loss = w_l * CE(logits[:-1], idx) + w_r * CE(logits[1:], idx) with w_l = clamp(|gt - dv[idx]| / |dv[idx+1] - dv[idx]|, 0, 1), w_r = 1 - w_l,
// summed over pixels TIMES the float mask (in range [dv[0], dv[D-1]] and mask > 0.5) and divided by (sum(mask) + 1e-6).  One lane = one
// pixel: a walk finds idx and the two log-sum-exps (over D-1 logits each), a second writes the unnormalized gradient
// mask * (w_l * (softmax_left - onehot_idx) + w_r * (softmax_right - onehot_{idx+1})) in the ORIGINAL depth order.
template <bool INVERSE>
__global__ __launch_bounds__(256) void mixup_ce_loss_kernel(const float* __restrict__ logits, const float* __restrict__ hyp,
                                                            const float* __restrict__ gt, const float* __restrict__ mask, int D, size_t HW,
                                                            float* __restrict__ grad, float* __restrict__ rows) {
    __shared__ float red[2][4];
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    float lsum = 0.0f, cnt = 0.0f;
    if (pix < HW) {
        const float* lg = logits + (size_t)b * D * HW + pix;
        const float* hv = hyp + (size_t)b * D * HW + pix;
        const float g = gt[(size_t)b * HW + pix];
        auto at = [&](const float* p, int j) { return p[(size_t)(INVERSE ? D - 1 - j : j) * HW]; };
        const float lo = at(hv, 0), hi = at(hv, D - 1);
        int idx = 0;
        float ml = -INFINITY, sl = 0.0f, mr = -INFINITY, sr = 0.0f;
        for (int j = 0; j < D; ++j) {
            if (j >= 1) idx += (at(hv, j) <= g) ? 1 : 0;
            const float l = at(lg, j);
            if (j < D - 1) {
                const float mn = fmaxf(ml, l);
                sl = sl * expf(ml - mn) + expf(l - mn);
                ml = mn;
            }
            if (j >= 1) {
                const float mn = fmaxf(mr, l);
                sr = sr * expf(mr - mn) + expf(l - mn);
                mr = mn;
            }
        }
        idx = min(idx, D - 2);
        const float left = at(hv, idx), itv = fabsf(at(hv, idx + 1) - left);
        const float x = fabsf(g - left) / itv;
        const float wl = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x), wr = 1.0f - wl;      // torch.clamp: a NaN stays a NaN
        const float outl = g < lo ? 1.0f : 0.0f, outr = g > hi ? 1.0f : 0.0f;
        const float fm = (1.0f - fminf(outl + outr, 1.0f)) * (mask[(size_t)b * HW + pix] > 0.5f ? 1.0f : 0.0f);
        const float lsel = ml + logf(sl), lser = mr + logf(sr);
        lsum = (lsel - at(lg, idx)) * wl * fm + (lser - at(lg, idx + 1)) * wr * fm;
        cnt = fm;
        if (grad) {
            float* gr = grad + (size_t)b * D * HW + pix;
            for (int d = 0; d < D; ++d) {
                const int j = INVERSE ? D - 1 - d : d;       // position in the flipped column
                const float l = lg[(size_t)d * HW];
                float v = 0.0f;
                if (j < D - 1) v += wl * fm * (expf(l - lsel) - (j == idx ? 1.0f : 0.0f));
                if (j >= 1) v += wr * fm * (expf(l - lser) - (j == idx + 1 ? 1.0f : 0.0f));
                gr[(size_t)d * HW] = v;
            }
        }
    }
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

// models/losses.py:51-85 (reg_loss_stage4), one stage: smooth-L1 (beta = 1) of depth / interval against gt / interval, mean over the pixels
// with mask > 0.5 [and, mask_out_range: gt inside the hypothesis column widened by half an interval at both ends, losses.py:62-74].
// grad = d(sum of the selected terms) / d depth; the 1 / N of the mean is applied by mvs_ce_loss_bwd_scale.
template <bool INVERSE>
__global__ __launch_bounds__(256) void reg_loss_kernel(const float* __restrict__ depth, const float* __restrict__ gt, const float* __restrict__ mask,
                                                       const float* __restrict__ hyp /*null: no range mask*/, const float* __restrict__ interval,
                                                       int D, size_t HW, float* __restrict__ grad, float* __restrict__ rows) {
    __shared__ float red[2][4];
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    float lsum = 0.0f, cnt = 0.0f;
    if (pix < HW) {
        const float itv = interval[b];
        const float gv = gt[(size_t)b * HW + pix];
        bool valid = mask[(size_t)b * HW + pix] > 0.5f;
        if (hyp) {
            const float* hv = hyp + (size_t)b * D * HW + pix;
            auto at = [&](int j) { return hv[(size_t)(INVERSE ? D - 1 - j : j) * HW]; };
            const float lo = at(0) - fabsf(at(1) - at(0)) / 2.0f, hi = at(D - 1) + fabsf(at(D - 1) - at(D - 2)) / 2.0f;
            valid = valid && !(gv < lo) && !(gv > hi);
        }
        const float x = depth[(size_t)b * HW + pix] / itv - gv / itv, ax = fabsf(x);
        if (valid) {
            lsum = ax < 1.0f ? 0.5f * x * x : ax - 0.5f;
            cnt = 1.0f;
        }
        if (grad) grad[(size_t)b * HW + pix] = valid ? (ax < 1.0f ? x : (x > 0.0f ? 1.0f : -1.0f)) / itv : 0.0f;
    }
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

// models/losses.py:88-162 (wasserstein_loss -> sinkhorn, continuous = False: what trainer/mvsformer_trainer.py:114-117 passes), one stage.
// Per pixel: nu = prob_volume column (D), mu = one-hot at the hypothesis nearest to the ground truth, cost M_ij = |i - j| / eps;
//   for k = 1..iters:  b_k[j] = log(mu_j + 1e-12) - LSE_i(M_ij + a_{k-1}[i]);   a_k[i] = log(nu_i + 1e-12) - LSE_j(M_ij + b_k[j])      (a_0 = 0)
//   T_ij = exp(M_ij + a_K[i] + b_K[j]);  pixel loss = sum_ij T_ij |i - j|  (the reference's signs: + M in the exponent), mean over mask > 0.5.
// The gradient with respect to prob_volume runs through all iterations (the reference's autograd does): reverse sweep
//   ga[i] = dL/da_K[i] = sum_j T_ij |i-j|, gb[j] likewise;  for k = K..1:  g_lognu += ga;  gb[j] -= sum_i ga[i] P^k_ij;  ga[i] = - sum_j gb[j] Q^k_ij
//   with P^k_ij = softmax_j(M_ij + b_k[j]) = exp(M_ij + b_k[j] + a_k[i] - lognu_i), Q^k_ij = softmax_i(M_ij + a_{k-1}[i]) = exp(M_ij + a_{k-1}[i] + b_k[j] - logmu_j);
//   d loss / d nu_i = g_lognu[i] / (nu_i + 1e-12).
// One wavefront per pixel (lane = (index l & 31, half l >> 5): a lane sums 16 terms of its row / column, the halves meet through one xor-32
// shuffle), a_k / b_k of all iterations in LDS (the backward needs them), D <= 32, iters <= 16.  grad = unnormalized (mvs_ce_loss_bwd_scale).
constexpr int WAS_MAXD = 32, WAS_MAXIT = 16;
__global__ __launch_bounds__(256) void was_loss_kernel(const float* __restrict__ prob, const float* __restrict__ hyp, const float* __restrict__ gt,
                                                       const float* __restrict__ mask, int D, size_t HW, int iters, float inv_eps,
                                                       float* __restrict__ grad, float* __restrict__ rows) {
    __shared__ float A[4][WAS_MAXIT + 1][WAS_MAXD], Bv[4][WAS_MAXIT + 1][WAS_MAXD], V[4][4][WAS_MAXD];      // V: lognu | ga | gb | g_lognu
    __shared__ float red[2][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, idx = lane & 31, half = lane >> 5;
    const size_t pix = (size_t)blockIdx.x * 4 + wave;
    const int b = blockIdx.y;
    float lsum = 0.0f, cnt = 0.0f;
    const bool inside = pix < HW;
    const bool valid = inside && mask[(size_t)b * HW + pix] > 0.5f;                 // wave-uniform
    if (inside && !valid && grad && lane < D) grad[((size_t)b * D + lane) * HW + pix] = 0.0f;
    if (valid) {
        float (*a)[WAS_MAXD] = A[wave];
        float (*bb)[WAS_MAXD] = Bv[wave];
        float* lognu = V[wave][0];
        float* ga = V[wave][1];
        float* gb = V[wave][2];
        float* gl = V[wave][3];
        const float gv = gt[(size_t)b * HW + pix];
        // nearest hypothesis (first index on ties, as torch.min): lanes 0..D-1 hold |hyp - gt|, butterfly on (distance, index)
        float dist = INFINITY;
        int gi = idx;
        float p = 0.0f;
        if (lane < D) {
            dist = fabsf(hyp[((size_t)b * D + lane) * HW + pix] - gv);
            p = prob[((size_t)b * D + lane) * HW + pix];
        }
        for (int m = 16; m >= 1; m >>= 1) {
            const float od = __shfl_xor(dist, m, 64);
            const int oi = __shfl_xor(gi, m, 64);
            if (od < dist || (od == dist && oi < gi)) dist = od, gi = oi;
        }
        gi = __shfl(gi, 0, 64);                              // (lanes 32..63 ran the same butterfly on +inf: take the low half's answer)
        const float logmu_hit = logf(1.0f + 1e-12f), logmu_miss = logf(1e-12f);
        if (lane < D) {
            lognu[lane] = logf(p + 1e-12f);
            a[0][lane] = 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        const int lo = half * 16, hi = min(D, lo + 16);       // the half of the other index this lane sums over
        auto lse = [&](const float* vec) {                    // log sum_t exp(|idx - t| * inv_eps + vec[t]) over t < D (both halves combined)
            float mx = -INFINITY;
            for (int t = lo; t < hi; ++t) mx = fmaxf(mx, fabsf((float)(idx - t)) * inv_eps + vec[t]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sm = 0.0f;
            for (int t = lo; t < hi; ++t) sm += expf(fabsf((float)(idx - t)) * inv_eps + vec[t] - mx);
            sm += __shfl_xor(sm, 32, 64);
            return mx + logf(sm);
        };
        for (int k = 1; k <= iters; ++k) {
            const float vb = (idx == gi ? logmu_hit : logmu_miss) - lse(a[k - 1]);
            if (lane < D) bb[k][lane] = vb;
            __builtin_amdgcn_wave_barrier();
            const float va = (idx < D ? lognu[idx] : 0.0f) - lse(bb[k]);
            if (lane < D) a[k][lane] = va;
            __builtin_amdgcn_wave_barrier();
        }
        // T_ij |i-j| summed over the lane's half row (ga) and half column (gb)
        float sa = 0.0f, sb = 0.0f;
        if (idx < D) {
            for (int t = lo; t < hi; ++t) {
                const float d = fabsf((float)(idx - t));
                sa += expf(d * inv_eps + a[iters][idx] + bb[iters][t]) * d;       // row idx, column t
                sb += expf(d * inv_eps + a[iters][t] + bb[iters][idx]) * d;       // row t, column idx
            }
        }
        sa += __shfl_xor(sa, 32, 64);
        sb += __shfl_xor(sb, 32, 64);
        float tot = lane < D ? sa : 0.0f;
        for (int m = 16; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, 64);
        lsum = tot, cnt = 1.0f;                               // (every lane of the low half holds the pixel's loss; lane 0 reports it)
        if (grad) {
            if (lane < D) ga[lane] = sa, gb[lane] = sb, gl[lane] = 0.0f;
            __builtin_amdgcn_wave_barrier();
            for (int k = iters; k >= 1; --k) {
                if (lane < D) gl[lane] += ga[lane];
                // gb[j] -= sum_i ga[i] * exp(M_ij + b_k[j] + a_k[i] - lognu[i])        (this lane: column j = idx, rows i = t)
                float acc = 0.0f;
                if (idx < D)
                    for (int t = lo; t < hi; ++t) acc += ga[t] * expf(fabsf((float)(idx - t)) * inv_eps + bb[k][idx] + a[k][t] - lognu[t]);
                acc += __shfl_xor(acc, 32, 64);
                __builtin_amdgcn_wave_barrier();
                if (lane < D) gb[lane] -= acc;
                __builtin_amdgcn_wave_barrier();
                // ga_prev[i] = - sum_j gb[j] * exp(M_ij + a_{k-1}[i] + b_k[j] - logmu[j])   (this lane: row i = idx, columns j = t)
                float acc2 = 0.0f;
                if (idx < D)
                    for (int t = lo; t < hi; ++t)
                        acc2 += gb[t] * expf(fabsf((float)(idx - t)) * inv_eps + a[k - 1][idx] + bb[k][t] - (t == gi ? logmu_hit : logmu_miss));
                acc2 += __shfl_xor(acc2, 32, 64);
                __builtin_amdgcn_wave_barrier();
                if (lane < D) ga[lane] = -acc2;
                // d b_k / d a_{k-1} only: b_k's own gradient restarts from zero for iteration k-1 (gb accumulates into b_{k-1} below)
                if (lane < D) gb[lane] = 0.0f;
                __builtin_amdgcn_wave_barrier();
            }
            if (lane < D) grad[((size_t)b * D + lane) * HW + pix] = gl[lane] / (p + 1e-12f);
        }
    }
    if (lane == 0) {
        red[0][wave] = lsum;
        red[1][wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        rows[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// acc[0..1] = the rows (acc + 2) added in a fixed order, acc[1] + eps = the denominator; loss = weight * acc[0] / acc[1]
__global__ __launch_bounds__(256) void ce_finalize_kernel(float* __restrict__ acc, int nrows, float weight, float eps, float* __restrict__ loss) {
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
        c += eps;                                     // 0 for the means over a selection, 1e-6 for mixup_ce's sum(mask) + 1e-6
        acc[0] = s;
        acc[1] = c;
        loss[0] = weight * (s / c);                   // N = 0, eps = 0 -> 0/0 = NaN, as F.cross_entropy on an empty selection
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
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, acc, (int)(grid.x * grid.y), weight, 0.0f, loss);
    return mvs::finish_launch("mvs_ce_loss_fwd");
}

extern "C" int mvs_mixup_ce_loss_fwd(const float* logits, const float* depth_values, const float* depth_gt, const float* mask, int B, int D,
                                     int64_t HW, int inverse_depth, float weight, float* grad_unscaled, float* acc, float* loss, mvs_stream_t stream) {
    MVS_REQUIRE(logits && depth_values && depth_gt && mask && acc && loss, "mvs_mixup_ce_loss_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && B <= 65535 && D >= 2 && HW >= 1, "mvs_mixup_ce_loss_fwd: bad shape B=%d D=%d (>= 2) HW=%lld", B, D, (long long)HW);
    hipStream_t s = MVS_STREAM(stream);
    dim3 grid((unsigned)mvs::ceil_div((long long)HW, 256LL), B);
    if (inverse_depth)
        hipLaunchKernelGGL(mixup_ce_loss_kernel<true>, grid, dim3(256), 0, s, logits, depth_values, depth_gt, mask, D, (size_t)HW, grad_unscaled, acc + 2);
    else
        hipLaunchKernelGGL(mixup_ce_loss_kernel<false>, grid, dim3(256), 0, s, logits, depth_values, depth_gt, mask, D, (size_t)HW, grad_unscaled, acc + 2);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, acc, (int)(grid.x * grid.y), weight, 1e-6f, loss);
    return mvs::finish_launch("mvs_mixup_ce_loss_fwd");
}

extern "C" int mvs_reg_loss_fwd(const float* depth, const float* depth_gt, const float* mask, const float* depth_values, const float* interval,
                                int B, int D, int64_t HW, int inverse_depth, float weight, float* grad_unscaled, float* acc, float* loss,
                                mvs_stream_t stream) {
    MVS_REQUIRE(depth && depth_gt && mask && interval && acc && loss, "mvs_reg_loss_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && B <= 65535 && HW >= 1 && (!depth_values || D >= 2), "mvs_reg_loss_fwd: bad shape B=%d D=%d HW=%lld", B, D, (long long)HW);
    hipStream_t s = MVS_STREAM(stream);
    dim3 grid((unsigned)mvs::ceil_div((long long)HW, 256LL), B);
    if (inverse_depth)
        hipLaunchKernelGGL(reg_loss_kernel<true>, grid, dim3(256), 0, s, depth, depth_gt, mask, depth_values, interval, D, (size_t)HW, grad_unscaled, acc + 2);
    else
        hipLaunchKernelGGL(reg_loss_kernel<false>, grid, dim3(256), 0, s, depth, depth_gt, mask, depth_values, interval, D, (size_t)HW, grad_unscaled, acc + 2);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, acc, (int)(grid.x * grid.y), weight, 0.0f, loss);
    return mvs::finish_launch("mvs_reg_loss_fwd");
}

extern "C" int64_t mvs_was_loss_acc_floats(int B, int64_t HW) {
    return (B < 1 || HW < 1) ? -1 : 2 + 2 * (int64_t)B * ((HW + 3) / 4);
}

extern "C" int mvs_was_loss_fwd(const float* prob_volume, const float* depth_values, const float* depth_gt, const float* mask, int B, int D, int64_t HW,
                                int ot_iter, float ot_eps, float weight, float* grad_unscaled, float* acc, float* loss, mvs_stream_t stream) {
    MVS_REQUIRE(prob_volume && depth_values && depth_gt && mask && acc && loss, "mvs_was_loss_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && B <= 65535 && D >= 1 && D <= WAS_MAXD && HW >= 1 && ot_iter >= 1 && ot_iter <= WAS_MAXIT && ot_eps > 0.0f,
                "mvs_was_loss_fwd: D <= %d, 1 <= ot_iter <= %d (got D=%d iters=%d)", WAS_MAXD, WAS_MAXIT, D, ot_iter);
    hipStream_t s = MVS_STREAM(stream);
    dim3 grid((unsigned)mvs::ceil_div((long long)HW, 4LL), B);
    hipLaunchKernelGGL(was_loss_kernel, grid, dim3(256), 0, s, prob_volume, depth_values, depth_gt, mask, D, (size_t)HW, ot_iter, 1.0f / ot_eps, grad_unscaled, acc + 2);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, acc, (int)(grid.x * grid.y), weight, 0.0f, loss);
    return mvs::finish_launch("mvs_was_loss_fwd");
}

extern "C" int mvs_ce_loss_bwd_scale(const float* grad_unscaled, float* grad, int64_t numel, const float* acc, const float* grad_out,
                                     float weight, mvs_stream_t stream) {
    MVS_REQUIRE(grad_unscaled && grad && acc && grad_out && numel >= 1, "mvs_ce_loss_bwd_scale: bad arguments");
    hipLaunchKernelGGL(ce_scale_kernel, dim3((unsigned)mvs::ceil_div((long long)numel, 256LL)), dim3(256), 0, MVS_STREAM(stream),
                       grad_unscaled, grad, (size_t)numel, acc, grad_out, weight);
    return mvs::finish_launch("mvs_ce_loss_bwd_scale");
}
