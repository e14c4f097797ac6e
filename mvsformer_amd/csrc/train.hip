// Training-mode pieces of the path: batch-statistics BatchNorm forward/backward around the raw
// MFMA convolutions (reference Conv3d/Deconv3d/ConvBnReLU in train mode, models/module.py:111-117,153-159,195-197;
// SyncBatchNorm hooks in here: the per-channel sums are what ranks all-reduce), softmax / 1x1x1-prob / sigmoid
// backward, and the NHWC->NCHW transpose of the feature gradient.  All bandwidth-bound elementwise / reduction kernels:
// tensors are [B, C, N] with N = D*H*W (or H*W), one block per (chunk of N, channel, batch) so every access is a
// coalesced 16-byte-per-lane stream; per-channel sums are block-reduced in LDS, written as one partial per block and
// combined in a fixed order by a second tiny kernel (no float atomics: BatchNorm statistics repeat bit-exactly run to run,
// which matters because the training head is an arg-max - a one-ulp change in a logit can move a whole hypothesis window).
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int CHUNK = 4096;            // elements of one (b, c) row handled by a block (256 threads x 4 x 4)

// Activation codes of the BatchNorm kernels' `relu` argument (0 / 1 keep their old meaning): 0 none, 1 ReLU (the 3-D regularizer and the
// visibility CNN), 2 leaky ReLU with slope 0.1 (FPNEncoder's Conv2d, models/module.py:66-67), 3 Swish x*sigmoid(x) (FPNDecoder, :200-206),
// 4 GELU in its erf form (VITDecoderStage4Single's nn.GELU(), models/module.py:361-364).
__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == 1) return fmaxf(z, 0.0f);
    if (act == 2) return z > 0.0f ? z : 0.1f * z;
    if (act == 3) return z / (1.0f + __expf(-z));
    if (act == 4) return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));          // GELU (erf form, nn.GELU()'s default)
    return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {          // d act(z) / dz
    if (act == 1) return z > 0.0f ? 1.0f : 0.0f;
    if (act == 2) return z > 0.0f ? 1.0f : 0.1f;
    if (act == 3) {
        const float sg = 1.0f / (1.0f + __expf(-z));
        return sg * (1.0f + z * (1.0f - sg));
    }
    if (act == 4) return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);   // Phi(z) + z phi(z)
    return 1.0f;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// part[p][c] = sum x, part[p][C + c] = sum x^2 over this block's share of channel c; p = b * gridDim.x + blockIdx.x
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int C, size_t N, float* __restrict__ part) {
    __shared__ float red[8];
    const int c = blockIdx.y, b = blockIdx.z;
    const float* row = x + ((size_t)b * C + c) * N;
    float s = 0.0f, q = 0.0f;
    for (size_t i0 = (size_t)blockIdx.x * CHUNK; i0 < N; i0 += (size_t)gridDim.x * CHUNK) {
        const size_t i1 = min(i0 + CHUNK, N);
        for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const float v = row[i];
            s += v;
            q = fmaf(v, v, q);
        }
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)b * gridDim.x + blockIdx.x) * 2 * C;
        o[c] = s;
        o[C + c] = q;
    }
}

// Element count per channel: a host value, or (SyncBatchNorm) two floats {n / 4096, n % 4096} that rode through the same
// all-reduce as the sums - each stays exactly representable in fp32 up to 2^36 elements, so the total is exact and the
// host never has to read it back.
__device__ __forceinline__ double resolve_count(double count_host, const float* __restrict__ count_dev) {
    return count_dev ? (double)count_dev[0] * 4096.0 + (double)count_dev[1] : count_host;
}

// mean/var from the (possibly all-reduced) sums; eval-style scale/shift for the apply kernel; running-stat update
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                   double count_host, const float* __restrict__ count_dev, int C, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double count = resolve_count(count_host, count_dev);
    const double mean = (double)sums[c] / count;
    double var = (double)sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    scale[c] = g * invstd;
    shift[c] = bt - (float)mean * g * invstd;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// Grouped form: the tensor holds `groups` independent BatchNorm calls of the SAME module side by side as channels
// g*C + c (the visibility CNN is applied once per source view in the reference, mvsformer_model.py:91: statistics per view,
// one set of affine parameters, running statistics updated once per call IN ORDER).  One thread per base channel walks the groups.
__global__ void bn_finalize_grouped_kernel(const float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                           float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                           double count_host, const float* __restrict__ count_dev, int C, int groups,
                                           float* __restrict__ scale, float* __restrict__ shift,
                                           float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double count = resolve_count(count_host, count_dev);
    const int CT = C * groups;
    const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    float rm = running_mean ? running_mean[c] : 0.0f, rv = running_var ? running_var[c] : 0.0f;
    for (int q = 0; q < groups; ++q) {
        const int cc = q * C + c;
        const double mean = (double)sums[cc] / count;
        double var = (double)sums[CT + cc] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        scale[cc] = g * invstd;
        shift[cc] = bt - (float)mean * g * invstd;
        mean_out[cc] = (float)mean;
        invstd_out[cc] = invstd;
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rm = (1.0f - momentum) * rm + momentum * (float)mean;
        rv = (1.0f - momentum) * rv + momentum * (float)unbiased;
    }
    if (running_mean) {
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// y = [relu](x*scale[c] + shift[c]) [+ residual]
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ res, int relu,
                                                         int C, size_t N, float* __restrict__ y) {
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t base = ((size_t)b * C + c) * N;
    const float sc = scale[c], sh = shift[c];
    const size_t i0 = (size_t)blockIdx.x * CHUNK, i1 = min(i0 + CHUNK, N);
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
        float v = act_fwd(fmaf(x[base + i], sc, sh), relu);
        if (res) v += res[base + i];
        y[base + i] = v;
    }
}

// backward of y = relu(bn(x)): g = dy * [x*scale+shift > 0];  sums[c] += sum g, sums[C+c] += sum g * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                            int C, size_t N, float* __restrict__ part) {
    __shared__ float red[8];
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t base = ((size_t)b * C + c) * N;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    float s1 = 0.0f, s2 = 0.0f;
    for (size_t i0 = (size_t)blockIdx.x * CHUNK; i0 < N; i0 += (size_t)gridDim.x * CHUNK) {
        const size_t i1 = min(i0 + CHUNK, N);
        for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const float xv = x[base + i];
            float g = dy[base + i];
            if (relu) g *= act_grad(fmaf(xv, sc, sh), relu);
            s1 += g;
            s2 = fmaf(g, (xv - mu) * is, s2);
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)b * gridDim.x + blockIdx.x) * 2 * C;
        o[c] = s1;
        o[C + c] = s2;
    }
}

// dx = gamma*invstd*(g - s1/n - xhat*s2/n)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           double count_host, const float* __restrict__ count_dev, int relu, int C,
                                                           size_t N, float* __restrict__ dx) {
    const double count = resolve_count(count_host, count_dev);
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t base = ((size_t)b * C + c) * N;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const float gi = (gamma ? gamma[c] : 1.0f) * is;
    const float m1 = (float)((double)sums[c] / count), m2 = (float)((double)sums[C + c] / count);
    const size_t i0 = (size_t)blockIdx.x * CHUNK, i1 = min(i0 + CHUNK, N);
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float xv = x[base + i];
        float g = dy[base + i];
        if (relu) g *= act_grad(fmaf(xv, sc, sh), relu);
        dx[base + i] = gi * (g - m1 - (xv - mu) * is * m2);
    }
}

// dpre = p * (dp - sum_d dp*p)   over [B,D,HW]
__global__ void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, int D, size_t HW, float* __restrict__ dpre) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= HW) return;
    const size_t base = (size_t)b * D * HW + i;
    float dot = 0.0f;
    for (int d = 0; d < D; ++d) dot = fmaf(dp[base + d * HW], p[base + d * HW], dot);
    for (int d = 0; d < D; ++d) dpre[base + d * HW] = p[base + d * HW] * (dp[base + d * HW] - dot);
}

// backward of logits[b,v] = sum_c w[c]*x[b,c,v] + bias:  dx[b,c,v] = w[c]*dl[b,v];  dwb[c] += sum dl*x[c], dwb[C] += sum dl
__global__ __launch_bounds__(256) void prob1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dl,
                                                        int C, size_t N, float* __restrict__ dx, float* __restrict__ dwb) {
    // grid-stride over chunks with the per-channel partial sums in registers: one block reduction + one atomic per
    // channel per BLOCK (not per chunk) — dwb has C+1 addresses and same-address atomics serialize
    constexpr int CMAX = 16;
    __shared__ float red[8];
    const int b = blockIdx.y;
    float s[CMAX], sb = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) s[c] = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
        const float g = dl[(size_t)b * N + i];
        sb += g;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) {
                const size_t o = ((size_t)b * C + c) * N + i;
                dx[o] = w[c] * g;
                s[c] = fmaf(g, x[o], s[c]);
            }
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            const float t = block_sum(s[c], red);
            if (threadIdx.x == 0) atomicAdd(&dwb[c], t);
        }
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) atomicAdd(&dwb[C], sb);
}

__global__ void sigmoid_kernel(const float* __restrict__ x, size_t n, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.0f / (1.0f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, size_t n, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dy[i] * y[i] * (1.0f - y[i]);
}

// [N,HW,C] -> [N,C,HW]
template <int C>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, size_t HW) {
    __shared__ float tile[C][65];
    const int tid = threadIdx.x;
    const size_t p0 = (size_t)blockIdx.x * 64, n = blockIdx.y;
    const int npix = (int)min((size_t)64, HW - p0);
    const float* src = in + (n * HW + p0) * C;
    for (int i = tid; i < npix * C; i += 256) tile[i % C][i / C] = src[i];
    __syncthreads();
    for (int i = tid; i < C * 64; i += 256) {
        const int c = i >> 6, p = i & 63;
        if (p < npix) out[(n * C + c) * HW + p0 + p] = tile[c][p];
    }
}

dim3 row_grid(int B, int C, size_t N) { return dim3((unsigned)((N + CHUNK - 1) / CHUNK), C, B); }
// reductions end in one fp32 atomic per block on 2C addresses, and same-address atomics serialize in L2: cap the blocks per
// (b, c) row (grid-stride over chunks) so that a channel has at most ~RED_ADDERS partial sums instead of N/4096
constexpr int RED_ADDERS = 48;
dim3 reduce_grid(int B, int C, size_t N) {
    size_t nb = (N + CHUNK - 1) / CHUNK;
    const size_t cap = (size_t)((RED_ADDERS + B - 1) / B);
    if (nb > cap) nb = cap;
    return dim3((unsigned)nb, C, B);
}

}  // namespace

extern "C" int64_t mvs_bn_reduce_workspace_bytes(int B, int C, int64_t N) {
    if (B < 1 || C < 1 || N < 1) return -1;
    const dim3 g = reduce_grid(B, C, (size_t)N);
    return (int64_t)g.x * B * 2 * C * (int64_t)sizeof(float);
}

extern "C" int mvs_bn_stats(const float* x, int B, int C, int64_t N, float* sums, void* workspace, mvs_stream_t stream) {
    MVS_REQUIRE(x && sums && workspace && B >= 1 && C >= 1 && C <= 65535 && B <= 65535 && N >= 1, "mvs_bn_stats: bad arguments");
    const dim3 g = reduce_grid(B, C, (size_t)N);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(bn_stats_kernel, g, dim3(256), 0, MVS_STREAM(stream), x, C, (size_t)N, part);
    mvs::launch_partials_reduce(part, (int)(g.x * B), 2 * C, sums, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_bn_stats");
}

extern "C" int mvs_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, double count, const float* count_dev, int C, float* scale, float* shift,
                               float* mean, float* invstd, mvs_stream_t stream) {
    MVS_REQUIRE(sums && scale && shift && mean && invstd && C >= 1 && (count_dev || count >= 1.0), "mvs_bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, MVS_STREAM(stream), sums, gamma, beta, running_mean,
                       running_var, momentum, eps, count, count_dev, C, scale, shift, mean, invstd);
    return mvs::finish_launch("mvs_bn_finalize");
}

extern "C" int mvs_bn_finalize_grouped(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                       float momentum, float eps, double count, const float* count_dev, int C, int groups, float* scale,
                                       float* shift, float* mean, float* invstd, mvs_stream_t stream) {
    MVS_REQUIRE(sums && scale && shift && mean && invstd && C >= 1 && groups >= 1 && (count_dev || count >= 1.0),
                "mvs_bn_finalize_grouped: bad arguments");
    hipLaunchKernelGGL(bn_finalize_grouped_kernel, dim3((C + 63) / 64), dim3(64), 0, MVS_STREAM(stream), sums, gamma, beta, running_mean,
                       running_var, momentum, eps, count, count_dev, C, groups, scale, shift, mean, invstd);
    return mvs::finish_launch("mvs_bn_finalize_grouped");
}

extern "C" int mvs_affine_act(const float* x, const float* scale, const float* shift, const float* residual, int relu, int B, int C,
                              int64_t N, float* y, mvs_stream_t stream) {
    MVS_REQUIRE(x && scale && shift && y && B >= 1 && C >= 1 && C <= 65535 && B <= 65535 && N >= 1, "mvs_affine_act: bad arguments");
    hipLaunchKernelGGL(affine_act_kernel, row_grid(B, C, N), dim3(256), 0, MVS_STREAM(stream), x, scale, shift, residual, relu, C,
                       (size_t)N, y);
    return mvs::finish_launch("mvs_affine_act");
}

extern "C" int mvs_bn_bwd_reduce(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, int relu, int B, int C, int64_t N, float* sums, void* workspace,
                                mvs_stream_t stream) {
    MVS_REQUIRE(dy && x && scale && shift && mean && invstd && sums && workspace && B >= 1 && C >= 1 && C <= 65535 && B <= 65535 && N >= 1,
                "mvs_bn_bwd_reduce: bad arguments");
    const dim3 g = reduce_grid(B, C, (size_t)N);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, g, dim3(256), 0, MVS_STREAM(stream), dy, x, scale, shift, mean, invstd, relu, C, (size_t)N, part);
    mvs::launch_partials_reduce(part, (int)(g.x * B), 2 * C, sums, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_bn_bwd_reduce");
}

extern "C" int mvs_bn_bwd_apply(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const float* gamma, const float* sums, double count, const float* count_dev,
                                int relu, int B, int C, int64_t N, float* dx, mvs_stream_t stream) {
    MVS_REQUIRE(dy && x && scale && shift && mean && invstd && sums && dx && B >= 1 && C >= 1 && C <= 65535 && B <= 65535 && N >= 1,
                "mvs_bn_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, row_grid(B, C, N), dim3(256), 0, MVS_STREAM(stream), dy, x, scale, shift, mean, invstd, gamma,
                       sums, count, count_dev, relu, C, (size_t)N, dx);
    return mvs::finish_launch("mvs_bn_bwd_apply");
}

extern "C" int mvs_softmax_bwd(const float* p, const float* dp, int B, int D, int64_t HW, float* dpre, mvs_stream_t stream) {
    MVS_REQUIRE(p && dp && dpre && B >= 1 && B <= 65535 && D >= 1 && HW >= 1, "mvs_softmax_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, MVS_STREAM(stream), p, dp, D, (size_t)HW,
                       dpre);
    return mvs::finish_launch("mvs_softmax_bwd");
}

extern "C" int mvs_prob1_bwd(const float* x, const float* w, const float* dlogits, int B, int C, int64_t N, float* dx, float* dwb,
                             mvs_stream_t stream) {
    MVS_REQUIRE(x && w && dlogits && dx && dwb && B >= 1 && B <= 65535 && C >= 1 && C <= 16 && N >= 1,
                "mvs_prob1_bwd: bad arguments (C <= 16)");
    const size_t nb_all = (size_t)((N + 255) / 256), nb_cap = (size_t)((512 + B - 1) / B);
    hipLaunchKernelGGL(prob1_bwd_kernel, dim3((unsigned)(nb_all < nb_cap ? nb_all : nb_cap), B), dim3(256), 0, MVS_STREAM(stream), x, w, dlogits, C,
                       (size_t)N, dx, dwb);
    return mvs::finish_launch("mvs_prob1_bwd");
}

extern "C" int mvs_sigmoid_fwd(const float* x, int64_t n, float* y, mvs_stream_t stream) {
    MVS_REQUIRE(x && y && n >= 1, "mvs_sigmoid_fwd: bad arguments");
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), x, (size_t)n, y);
    return mvs::finish_launch("mvs_sigmoid_fwd");
}

extern "C" int mvs_sigmoid_bwd(const float* y, const float* dy, int64_t n, float* dx, mvs_stream_t stream) {
    MVS_REQUIRE(y && dy && dx && n >= 1, "mvs_sigmoid_bwd: bad arguments");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), y, dy, (size_t)n, dx);
    return mvs::finish_launch("mvs_sigmoid_bwd");
}

// out = a * b elementwise (AttentionFusionSimple's x1 * x2, models/module.py:464; its backward is the same kernel twice: da = dy * b, db = dy * a)
__global__ __launch_bounds__(256) void ewise_mul_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + i), y = *reinterpret_cast<const f32x4*>(b + i);
        *reinterpret_cast<f32x4*>(out + i) = f32x4{x[0] * y[0], x[1] * y[1], x[2] * y[2], x[3] * y[3]};
    } else {
        for (size_t k = i; k < n; ++k) out[k] = a[k] * b[k];
    }
}

extern "C" int mvs_ewise_mul(const float* a, const float* b, int64_t n, float* out, mvs_stream_t stream) {
    MVS_REQUIRE(a && b && out && n >= 1 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "mvs_ewise_mul: 16-byte aligned tensors");
    hipLaunchKernelGGL(ewise_mul_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, MVS_STREAM(stream), a, b, (size_t)n, out);
    return mvs::finish_launch("mvs_ewise_mul");
}

extern "C" int mvs_nhwc_to_nchw(const float* in, float* out, int N, int C, int64_t HW, mvs_stream_t stream) {
    MVS_REQUIRE(in && out && N >= 1 && N <= 65535 && HW >= 1, "mvs_nhwc_to_nchw: bad shape");
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "mvs_nhwc_to_nchw: C must be 8, 16, 32 or 64 (got %d)", C);
    dim3 grid((unsigned)((HW + 63) / 64), N);
    hipStream_t s = MVS_STREAM(stream);
    switch (C) {
        case 8: hipLaunchKernelGGL(nhwc_to_nchw_kernel<8>, grid, dim3(256), 0, s, in, out, (size_t)HW); break;
        case 16: hipLaunchKernelGGL(nhwc_to_nchw_kernel<16>, grid, dim3(256), 0, s, in, out, (size_t)HW); break;
        case 32: hipLaunchKernelGGL(nhwc_to_nchw_kernel<32>, grid, dim3(256), 0, s, in, out, (size_t)HW); break;
        default: hipLaunchKernelGGL(nhwc_to_nchw_kernel<64>, grid, dim3(256), 0, s, in, out, (size_t)HW); break;
    }
    return mvs::finish_launch("mvs_nhwc_to_nchw");
}
