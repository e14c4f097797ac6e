// Visibility-weight CNN of StageNet (models/mvsformer_model.py:37,91; ConvBnReLU = models/module.py:168-197):
//   ConvBnReLU(1,16) -> ConvBnReLU(16,16) -> ConvBnReLU(16,8) -> Conv2d(8,1,1) -> Sigmoid
// applied to the [B*(V-1),1,H,W] entropy maps.  3.6 kMAC per pixel per source view: 68 GFLOP per config-2
// depth map, i.e. NOT negligible next to the 3-D regularizer — so the four layers run as ONE launch with all
// intermediates in LDS (the unfused form would stream 2x 16-channel full-resolution fp32 maps through HBM).
//
// A block produces a 32x8 output tile.  LDS holds the entropy tile with a 3-pixel halo, the 16-channel
// layer-1 activations (halo 2) and the 16-channel layer-2 activations (halo 1).  Activations at positions
// outside the image are stored as 0 — each conv zero-pads ITS OWN input.  Weights are wave-uniform, so they
// are fetched through the scalar cache (s_load) and feed v_fma as SGPR operands; each LDS read of an input
// value is reused for 8-16 output channels.
//
// params (MVS_VIS_PARAM_FLOATS floats), eval-mode BatchNorm folded to y = conv*scale + shift:
//   [0    ] w0[tap 9][cout 16]          [144 ] scale0[16]  [160 ] shift0[16]
//   [176  ] w1[cin 16][tap 9][cout 16]  [2480] scale1[16]  [2496] shift1[16]
//   [2512 ] w2[cin 16][tap 9][cout 8]   [3664] scale2[8]   [3672] shift2[8]
//   [3680 ] w3[8]                       [3688] b3
#include "common.h"

namespace {

constexpr int TW = 32, TH = 8;
constexpr int IN_W = TW + 6, IN_H = TH + 6;          // entropy tile, halo 3
constexpr int A1_W = TW + 4, A1_H = TH + 4;          // layer-1 output, halo 2
constexpr int A2_W = TW + 2, A2_H = TH + 2;          // layer-2 output, halo 1
constexpr int A1_PLANE = A1_W * A1_H, A2_PLANE = A2_W * A2_H;
constexpr int OFF_W0 = 0, OFF_S0 = 144, OFF_B0 = 160, OFF_W1 = 176, OFF_S1 = 2480, OFF_B1 = 2496, OFF_W2 = 2512,
              OFF_S2 = 3664, OFF_B2 = 3672, OFF_W3 = 3680, OFF_B3 = 3688;
static_assert(OFF_B3 + 1 == MVS_VIS_PARAM_FLOATS, "param layout");

__global__ __launch_bounds__(256) void vis_kernel(const float* __restrict__ entropy, const float* __restrict__ prm, int H, int W,
                                                  float* __restrict__ weight) {
    __shared__ float s_in[IN_H * IN_W];
    __shared__ float s_a1[16 * A1_PLANE];
    __shared__ float s_a2[16 * A2_PLANE];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z;
    const float* src = entropy + (size_t)n * H * W;

    for (int i = tid; i < IN_H * IN_W; i += 256) {
        const int py = i / IN_W, px = i % IN_W;
        const int gy = y0 - 3 + py, gx = x0 - 3 + px;
        s_in[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();

    // layer 1: 1 -> 16
    for (int i = tid; i < A1_PLANE; i += 256) {
        const int py = i / A1_W, px = i % A1_W;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float v = s_in[(py + ky) * IN_W + px + kx];
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(prm[OFF_W0 + (ky * 3 + kx) * 16 + c], v, acc[c]);
            }
        const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float o = fmaxf(fmaf(acc[c], prm[OFF_S0 + c], prm[OFF_B0 + c]), 0.0f);
            s_a1[c * A1_PLANE + i] = inside ? o : 0.0f;
        }
    }
    __syncthreads();

    // layer 2: 16 -> 16, a work item = 64 positions x 8 output channels (the channel half is wave-uniform)
    {
        constexpr int WPH = (A2_PLANE + 63) / 64;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        for (int wi = wave; wi < 2 * WPH; wi += 4) {
            const int half = wi / WPH;
            const int i = (wi % WPH) * 64 + lane;
            if (i < A2_PLANE) {
                const int py = i / A2_W, px = i % A2_W;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                float acc[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
                const float* w1 = prm + OFF_W1 + half * 8;
                for (int ci = 0; ci < 16; ++ci) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float v = s_a1[ci * A1_PLANE + (py + ky) * A1_W + px + kx];
#pragma unroll
                            for (int c = 0; c < 8; ++c) acc[c] = fmaf(w1[(ci * 9 + ky * 3 + kx) * 16 + c], v, acc[c]);
                        }
                }
                const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int co = half * 8 + c;
                    const float o = fmaxf(fmaf(acc[c], prm[OFF_S1 + co], prm[OFF_B1 + co]), 0.0f);
                    s_a2[co * A2_PLANE + i] = inside ? o : 0.0f;
                }
            }
        }
    }
    __syncthreads();

    // layer 3 (16 -> 8) + 1x1 conv + sigmoid: one output pixel per thread
    {
        const int py = tid / TW, px = tid % TW;
        const int gy = y0 + py, gx = x0 + px;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
        for (int ci = 0; ci < 16; ++ci) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = s_a2[ci * A2_PLANE + (py + ky) * A2_W + px + kx];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = fmaf(prm[OFF_W2 + (ci * 9 + ky * 3 + kx) * 8 + c], v, acc[c]);
                }
        }
        float o = prm[OFF_B3];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float a = fmaxf(fmaf(acc[c], prm[OFF_S2 + c], prm[OFF_B2 + c]), 0.0f);
            o = fmaf(prm[OFF_W3 + c], a, o);
        }
        if (gy < H && gx < W) weight[(size_t)n * H * W + (size_t)gy * W + gx] = 1.0f / (1.0f + expf(-o));
    }
}

}  // namespace

extern "C" int mvs_vis_fwd(const float* entropy, const float* params, int N, int H, int W, float* weight, mvs_stream_t stream) {
    MVS_REQUIRE(entropy && params && weight, "mvs_vis_fwd: null pointer");
    MVS_REQUIRE(N >= 1 && H >= 1 && W >= 1 && N <= 65535, "mvs_vis_fwd: bad shape N=%d H=%d W=%d", N, H, W);
    dim3 grid(mvs::ceil_div(W, TW), mvs::ceil_div(H, TH), N);
    hipLaunchKernelGGL(vis_kernel, grid, dim3(256), 0, MVS_STREAM(stream), entropy, params, H, W, weight);
    return mvs::finish_launch("mvs_vis_fwd");
}
