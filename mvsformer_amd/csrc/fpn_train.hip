// Training-mode pieces of the FPN decoder (models/module.py:257-270 under train()): the x2 bilinear upsampling with
// align_corners = True plus the lateral add (`F.interpolate(intra, scale_factor=2, mode="bilinear", align_corners=True) + inner(conv)`)
// as one kernel, and its adjoint as a GATHER (each input pixel collects the <= 3 x 3 output pixels whose 2 x 2 footprint contains it: no
// atomics, bit-reproducible).  fp32 NCHW like the reference; the convolutions around it run through mvs_conv2d_gemm_x3 (csrc/vit.hip),
// BatchNorm / activations through train.hip.
#include "common.h"

namespace {
// ATen's align_corners = True source index: src = dst * (in - 1) / (out - 1); i0 = (int)src, i1 = i0 + (i0 < in - 1), lambda = src - i0
__device__ __forceinline__ void src_of(int dst, int in, int out, int* i0, int* i1, float* l1) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    const float s = scale * (float)dst;
    *i0 = (int)s;
    *i1 = *i0 + (*i0 < in - 1 ? 1 : 0);
    *l1 = s - (float)*i0;
}

__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float* __restrict__ x, const float* __restrict__ lat, float* __restrict__ y, int h,
                                                             int w) {
    const int H = 2 * h, W = 2 * w;
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= W) return;
    const float* xp = x + (size_t)blockIdx.z * h * w;
    int y0, y1, x0, x1;
    float ly, lx;
    src_of(oy, h, H, &y0, &y1, &ly);
    src_of(ox, w, W, &x0, &x1, &lx);
    const float top = (1.0f - lx) * xp[(size_t)y0 * w + x0] + lx * xp[(size_t)y0 * w + x1];
    const float bot = (1.0f - lx) * xp[(size_t)y1 * w + x0] + lx * xp[(size_t)y1 * w + x1];
    const size_t o = ((size_t)blockIdx.z * H + oy) * W + ox;
    const float v = (1.0f - ly) * top + ly * bot;
    y[o] = lat ? v + lat[o] : v;
}

// dx[iy][ix] = sum over output pixels (oy, ox) of dy * (weight of (iy, ix) in the output pixel's footprint)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int h, int w) {
    const int H = 2 * h, W = 2 * w;
    const int ix = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y;
    if (ix >= w) return;
    const float* gp = dy + (size_t)blockIdx.z * H * W;
    // candidate outputs: src = o * (in - 1)/(out - 1) in (i - 1, i + 1)  ->  o in ((i - 1) * (out - 1)/(in - 1), (i + 1) * (out - 1)/(in - 1))
    const int oy_lo = h > 1 ? max(0, (int)floorf((float)(iy - 1) * (float)(H - 1) / (float)(h - 1))) : 0;
    const int oy_hi = h > 1 ? min(H - 1, (int)ceilf((float)(iy + 1) * (float)(H - 1) / (float)(h - 1))) : H - 1;
    const int ox_lo = w > 1 ? max(0, (int)floorf((float)(ix - 1) * (float)(W - 1) / (float)(w - 1))) : 0;
    const int ox_hi = w > 1 ? min(W - 1, (int)ceilf((float)(ix + 1) * (float)(W - 1) / (float)(w - 1))) : W - 1;
    float acc = 0.0f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float ly;
        src_of(oy, h, H, &y0, &y1, &ly);
        const float wy = (y0 == iy ? 1.0f - ly : 0.0f) + (y1 == iy ? ly : 0.0f);
        if (wy == 0.0f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            int x0, x1;
            float lx;
            src_of(ox, w, W, &x0, &x1, &lx);
            const float wx = (x0 == ix ? 1.0f - lx : 0.0f) + (x1 == ix ? lx : 0.0f);
            if (wx != 0.0f) acc = fmaf(wy * wx, gp[(size_t)oy * W + ox], acc);
        }
    }
    dx[((size_t)blockIdx.z * h + iy) * w + ix] = acc;
}
}  // namespace

// y [planes][2h][2w] = bilinear_x2(x [planes][h][w], align_corners = True) (+ lateral)
extern "C" int mvs_upsample2x_add(const float* x, const float* lateral, float* y, int planes, int h, int w, mvs_stream_t stream) {
    MVS_REQUIRE(x && y && planes >= 1 && planes <= 65535 && h >= 1 && h <= 32767 && w >= 1, "mvs_upsample2x_add: bad shape");
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3((2 * w + 255) / 256, 2 * h, planes), dim3(256), 0, MVS_STREAM(stream), x, lateral, y, h, w);
    return mvs::finish_launch("mvs_upsample2x_add");
}

// dx [planes][h][w] = adjoint of the upsampling applied to dy [planes][2h][2w] (the lateral branch's gradient is dy itself)
extern "C" int mvs_upsample2x_bwd(const float* dy, float* dx, int planes, int h, int w, mvs_stream_t stream) {
    MVS_REQUIRE(dy && dx && planes >= 1 && planes <= 65535 && h >= 1 && h <= 65535 && w >= 1, "mvs_upsample2x_bwd: bad shape");
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((w + 255) / 256, h, planes), dim3(256), 0, MVS_STREAM(stream), dy, dx, h, w);
    return mvs::finish_launch("mvs_upsample2x_bwd");
}
