// 1x1x1 heads of the bf16 training path on channel-last activations (round 5): the layers between the half-precision regularizer /
// visibility CNN and the fp32 world, forward and backward in ONE launch each way + one fixed-order row reduce for the weight gradient.
//   out[v] = act( sum_c w[c] * x[v][c] + bias )      x = bf16 [N voxels][8], out fp32 [N]; act = identity or sigmoid
// Replaces, per stage and step: bf16 -> fp32 NCDHW transpose + mvs_prob1_fwd (+ mvs_sigmoid_fwd) forward, and mvs_sigmoid_bwd +
// mvs_prob1_bwd (float atomics on C + 1 addresses) + fp32 -> bf16 transpose backward.  References: CostRegNet3D.prob (models/module.py:577,
// a Conv3d(8, 1, 1) with bias), StageNet.vis[3:5] (Conv2d(8, 1, 1) + Sigmoid, models/mvsformer_model.py:37); `select` (w = e0, no bias, no
// parameter gradient) picks channel 0 of CostRegNet's 8 -> 1 `prob` convolution, which runs zero-padded to 8 output channels.
#include "common.h"

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int HEAD_ROWS = 1024;                              // voxels per block of the backward (4 per thread)

__global__ __launch_bounds__(256) void bf16_head_fwd_kernel(const __bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            int sigmoid, size_t N, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    float acc = bias ? bias[0] : 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc = fmaf(w ? w[c] : (c == 0 ? 1.0f : 0.0f), (float)v[c], acc);
    out[i] = sigmoid ? 1.0f / (1.0f + __expf(-acc)) : acc;
}

// dl = dout * (sigmoid ? y*(1-y) : 1);  dx[v][c] = w[c]*dl;  part[j][block] = sum over the block's voxels of dl*x[c] (j < 8) / dl (j = 8)
__global__ __launch_bounds__(256) void bf16_head_bwd_kernel(const __bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ y,
                                                            const float* __restrict__ dout, size_t N, __bf16* __restrict__ dx,
                                                            float* __restrict__ part) {
    __shared__ float red[4][9];
    float wv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) wv[c] = w ? w[c] : (c == 0 ? 1.0f : 0.0f);
    float s[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) s[j] = 0.0f;
    const size_t base = (size_t)blockIdx.x * HEAD_ROWS;
#pragma unroll
    for (int k = 0; k < HEAD_ROWS / 256; ++k) {
        const size_t i = base + k * 256 + threadIdx.x;
        if (i >= N) break;
        float dl = dout[i];
        if (y) {
            const float yy = y[i];
            dl *= yy * (1.0f - yy);
        }
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (__bf16)(wv[c] * dl);
        reinterpret_cast<bf16x8*>(dx)[i] = o;
        if (part) {
            const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
#pragma unroll
            for (int c = 0; c < 8; ++c) s[c] = fmaf(dl, (float)v[c], s[c]);
            s[8] += dl;
        }
    }
    if (!part) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s[j] += __shfl_xor(s[j], m, 64);
        if (lane == 0) red[wave][j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < 9) part[(size_t)blockIdx.x * 9 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void bf16_embed_ch0_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    bf16x8 v;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (__bf16)0.0f;
    v[0] = (__bf16)in[i];
    reinterpret_cast<bf16x8*>(out)[i] = v;
}
}  // namespace

extern "C" int mvs_bf16_embed_ch0(const float* in, void* out, int64_t N, mvs_stream_t stream) {
    MVS_REQUIRE(in && out && N >= 1, "mvs_bf16_embed_ch0: bad arguments");
    hipLaunchKernelGGL(bf16_embed_ch0_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), in,
                       reinterpret_cast<__bf16*>(out), (size_t)N);
    return mvs::finish_launch("mvs_bf16_embed_ch0");
}

extern "C" int mvs_bf16_head_fwd(const void* x, const float* w, const float* bias, int sigmoid, int64_t N, float* out, mvs_stream_t stream) {
    MVS_REQUIRE(x && out && N >= 1 && N < ((int64_t)1 << 40), "mvs_bf16_head_fwd: bad arguments");
    hipLaunchKernelGGL(bf16_head_fwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), reinterpret_cast<const __bf16*>(x),
                       w, bias, sigmoid, (size_t)N, out);
    return mvs::finish_launch("mvs_bf16_head_fwd");
}

extern "C" int64_t mvs_bf16_head_bwd_workspace_bytes(int64_t N) {
    return N < 1 ? -1 : ((N + HEAD_ROWS - 1) / HEAD_ROWS) * 9 * (int64_t)sizeof(float);
}

// y: the forward's output if it applied the sigmoid, else NULL.  dwb [9] = [dw (8) | dbias], or NULL (with w NULL: the channel-0 select,
// which has no parameters) - then no workspace is needed either.
extern "C" int mvs_bf16_head_bwd(const void* x, const float* w, const float* y, const float* dout, int64_t N, void* dx, float* dwb,
                                 void* workspace, mvs_stream_t stream) {
    MVS_REQUIRE(x && dout && dx && N >= 1 && (!dwb || workspace), "mvs_bf16_head_bwd: bad arguments");
    const unsigned nb = (unsigned)((N + HEAD_ROWS - 1) / HEAD_ROWS);
    float* part = dwb ? reinterpret_cast<float*>(workspace) : nullptr;
    hipLaunchKernelGGL(bf16_head_bwd_kernel, dim3(nb), dim3(256), 0, MVS_STREAM(stream), reinterpret_cast<const __bf16*>(x), w, y, dout, (size_t)N,
                       reinterpret_cast<__bf16*>(dx), part);
    if (dwb) mvs::launch_partials_reduce(part, (int)nb, 9, dwb, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_bf16_head_bwd");
}
