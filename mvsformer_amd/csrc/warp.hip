// Unfused plane-sweep warp: homo_warping_3D_with_mask / homo_warping_3D (models/warping.py:69-109,155-189).
// Materializes warped[B,C,D,H,W] + mask[B,D,H,W] for callers that use the op on its own; the fused
// cost-volume sweeps (cost_volume.hip) never do.  Pure bandwidth: one lane per reference pixel along W
// (64 consecutive pixels per wavefront => both the source taps and the C stores are coalesced), the 4 tap
// offsets/weights are computed once per (d, pixel) and reused over the C channels.
#include "common.h"
#include "geometry.h"

namespace {

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ rt_all,
                                                       const float* __restrict__ depth, int depth_per_pixel,
                                                       int C, int D, int H, int W,
                                                       float* __restrict__ warped, uint8_t* __restrict__ mask) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / D, d = blockIdx.z % D;
    if (x >= W || y >= H) return;
    const float* rt = rt_all + b * 12;
    const size_t HW = (size_t)H * W;
    const float dv = depth_per_pixel ? depth[((size_t)(b * D + d) * H + y) * W + x] : depth[b * D + d];
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
    float un, vn, z;
    mvs::sweep_project(rt, (float)x, (float)y, dv, half_w, half_h, &un, &vn, &z);
    const mvs::Taps t = mvs::sweep_taps(un, vn, H, W, half_w, half_h);
    if (mask) mask[((size_t)(b * D + d) * H + y) * W + x] = mvs::sweep_outside(un, vn, z) ? 1 : 0;
    const float* sp = src + (size_t)b * C * HW;
    float* wp = warped + (((size_t)b * C * D + d) * H + y) * W + x;
    for (int c = 0; c < C; ++c) wp[(size_t)c * D * HW] = mvs::bilinear(sp + (size_t)c * HW, t);
}

}  // namespace

extern "C" int mvs_warp_fwd(const float* src, const float* rt, const float* depth, int depth_per_pixel,
                            int B, int C, int D, int H, int W, float* warped, uint8_t* mask, mvs_stream_t stream) {
    MVS_REQUIRE(src && rt && depth && warped, "mvs_warp_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && C >= 1 && D >= 1 && H >= 1 && W >= 1, "mvs_warp_fwd: bad shape B=%d C=%d D=%d H=%d W=%d", B, C, D, H, W);
    MVS_REQUIRE((int64_t)B * D <= 65535, "mvs_warp_fwd: B*D=%lld exceeds grid.z", (long long)B * D);
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B * D), block(64, 4);
    hipLaunchKernelGGL(warp_fwd_kernel, grid, block, 0, MVS_STREAM(stream), src, rt, depth, depth_per_pixel, C, D, H, W,
                       warped, mask);
    return mvs::finish_launch("mvs_warp_fwd");
}
