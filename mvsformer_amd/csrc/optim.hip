// AdamW over MANY parameter tensors in a few launches (the reference's optimizer, train.py:98 torch.optim.AdamW; decoupled weight
// decay, Loshchilov & Hutter):
//     p <- p * (1 - lr * wd);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2;
//     p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),        t = step + 1
// The cascade's hot path has ~190 parameter tensors with 1.2 M values: ATen's multi-tensor kernel gives every tensor (chunk of 64 K values)
// to ONE block, so a launch lasts as long as its largest tensor takes a single block - five launches of 40 us per step.  Here a block owns
// 2048 consecutive values of one tensor (block -> tensor through a table in the kernel arguments: by value, so a hipGraph captures it with
// the launch and nothing has to stay alive), ~650 blocks for the whole path; the step count lives on the device and is advanced by a
// one-thread launch after the updates (all of them read the old value).
#include <math.h>

#include "common.h"

namespace {
constexpr int AD_GROUP = 88;                                  // tensors per launch (kernel arguments are limited to 4 KB)
constexpr int AD_BLOCK = 2048;                                // values per block

struct AdamGroup {
    int n;
    int start[AD_GROUP + 1];                                  // first block of tensor i
    MvsAdamTensor t[AD_GROUP];
};
static_assert(sizeof(AdamGroup) <= 3900, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(256) void adamw_kernel(const AdamGroup g, float lr, float beta1, float beta2, float eps, float wd, int maximize,
                                                    const float* __restrict__ step) {
    int lo = 0, hi = g.n - 1;                                 // the tensor of this block: binary search over the (block-uniform) table
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= g.start[mid]) lo = mid;
        else hi = mid - 1;
    }
    const MvsAdamTensor t = g.t[lo];
    const long long base = (long long)((int)blockIdx.x - g.start[lo]) * AD_BLOCK;
    const double tt = (double)step[0] + 1.0;
    const float bc1 = (float)(1.0 - pow((double)beta1, tt)), bc2s = sqrtf((float)(1.0 - pow((double)beta2, tt)));
    const float step_size = lr / bc1, decay = 1.0f - lr * wd;
#pragma unroll
    for (int k = 0; k < AD_BLOCK / 256; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i >= t.n) break;
        const float gr = maximize ? -t.g[i] : t.g[i];
        const float m = beta1 * t.m[i] + (1.0f - beta1) * gr;
        const float v = beta2 * t.v[i] + (1.0f - beta2) * gr * gr;
        t.m[i] = m;
        t.v[i] = v;
        t.p[i] = t.p[i] * decay - step_size * (m / (sqrtf(v) / bc2s + eps));
    }
}

__global__ void adamw_advance_kernel(float* step) { step[0] += 1.0f; }
}  // namespace

extern "C" int mvs_adamw_step(const MvsAdamTensor* tensors, int ntensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int maximize, float* step, mvs_stream_t stream) {
    MVS_REQUIRE(tensors && ntensors >= 1 && ntensors <= 1 << 20 && step, "mvs_adamw_step: bad arguments (ntensors=%d)", ntensors);
    MVS_REQUIRE(lr >= 0.0f && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f && weight_decay >= 0.0f,
                "mvs_adamw_step: lr %g, betas (%g, %g), eps %g, weight_decay %g out of range", lr, beta1, beta2, eps, weight_decay);
    hipStream_t s = MVS_STREAM(stream);
    for (int first = 0; first < ntensors; first += AD_GROUP) {
        AdamGroup g{};
        g.n = ntensors - first < AD_GROUP ? ntensors - first : AD_GROUP;
        for (int i = 0; i < g.n; ++i) {
            const MvsAdamTensor& t = tensors[first + i];
            MVS_REQUIRE(t.p && t.g && t.m && t.v && t.n >= 1 && t.n < ((int64_t)1 << 40), "mvs_adamw_step: tensor %d: null pointer or bad size", first + i);
            g.t[i] = t;
            const int64_t blocks = (t.n + AD_BLOCK - 1) / AD_BLOCK;
            MVS_REQUIRE(g.start[i] + blocks < ((int64_t)1 << 31), "mvs_adamw_step: too many blocks");
            g.start[i + 1] = g.start[i] + (int)blocks;
        }
        hipLaunchKernelGGL(adamw_kernel, dim3(g.start[g.n]), dim3(256), 0, s, g, lr, beta1, beta2, eps, weight_decay, maximize, step);
        if (int rc = mvs::finish_launch("mvs_adamw_step")) return rc;
    }
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, s, step);
    return mvs::finish_launch("mvs_adamw_step");
}
