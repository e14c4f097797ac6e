// Plane-sweep sampling geometry shared by the unfused warp and the fused cost-volume sweeps.
//
// Semantics restated from the reference call chain (models/warping.py:84-106 feeding
// F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True)); the checklist is SURVEY.md
// Appendix B.  The file is compiled with -ffp-contract=off, so every rounding below is explicit: products
// and sums that the reference performs as separate tensor ops stay separate, fmaf() is used only where the
// reference's matmul accumulates.
#pragma once
#include <hip/hip_runtime.h>

namespace mvs {

struct Taps {
    int o00, o01, o10, o11;      // element offsets (y*W + x) of the 4 taps inside one H*W plane, clamped in-bounds
    float w00, w01, w10, w11;    // bilinear weights, already 0 for taps outside the source image
};

// rt: 12 floats (3x3 row-major, then translation).  x, y: integer pixel lattice of the reference view.
// Returns X2 (camera-space z in the source view) through *z, normalized coordinates through *un, *vn.
__device__ __forceinline__ void sweep_project(const float* __restrict__ rt, float x, float y, float d,
                                              float half_w, float half_h, float* un, float* vn, float* z) {
    // rot @ (x, y, 1): k-ordered fma chain like a k=3 sgemm
    const float rx = fmaf(rt[2], 1.0f, fmaf(rt[1], y, rt[0] * x));
    const float ry = fmaf(rt[5], 1.0f, fmaf(rt[4], y, rt[3] * x));
    const float rz = fmaf(rt[8], 1.0f, fmaf(rt[7], y, rt[6] * x));
    const float X0 = rx * d + rt[9];
    const float X1 = ry * d + rt[10];
    const float X2 = rz * d + rt[11];
    const float zz = X2 + 1e-6f;
    const float u = X0 / zz;
    const float v = X1 / zz;
    *un = u / half_w - 1.0f;
    *vn = v / half_h - 1.0f;
    *z = X2;
}

// grid_sample un-normalization (align_corners=True) + the 4 zero-padded taps.
__device__ __forceinline__ Taps sweep_taps(float un, float vn, int H, int W, float half_w, float half_h) {
    const float ix = (un + 1.0f) * half_w;     // ((u_n + 1) / 2) * (W - 1)
    const float iy = (vn + 1.0f) * half_h;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx = ix - x0f, wy = iy - y0f;
    const float ex = 1.0f - wx, ey = 1.0f - wy;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    // float comparisons: NaN / +-inf coordinates fail every test and contribute zero
    const bool vx0 = (x0f >= 0.0f) && (x0f <= wm1);
    const bool vx1 = (x0f >= -1.0f) && (x0f <= wm1 - 1.0f);
    const bool vy0 = (y0f >= 0.0f) && (y0f <= hm1);
    const bool vy1 = (y0f >= -1.0f) && (y0f <= hm1 - 1.0f);
    const int x0 = vx0 ? (int)x0f : 0;
    const int x1 = vx1 ? (int)x0f + 1 : 0;
    const int y0 = vy0 ? (int)y0f : 0;
    const int y1 = vy1 ? (int)y0f + 1 : 0;
    Taps t;
    t.o00 = y0 * W + x0;
    t.o01 = y0 * W + x1;
    t.o10 = y1 * W + x0;
    t.o11 = y1 * W + x1;
    t.w00 = (vx0 && vy0) ? ey * ex : 0.0f;
    t.w01 = (vx1 && vy0) ? ey * wx : 0.0f;
    t.w10 = (vx0 && vy1) ? wy * ex : 0.0f;
    t.w11 = (vx1 && vy1) ? wy * wx : 0.0f;
    return t;
}

// sweep_taps plus the integer position of tap (0,0), clamped to [-2, W+1] x [-2, H+1] (for callers that bin taps spatially)
__device__ __forceinline__ Taps sweep_taps_xy(float un, float vn, int H, int W, float half_w, float half_h, int* x0, int* y0) {
    const float ix = (un + 1.0f) * half_w, iy = (vn + 1.0f) * half_h;
    *x0 = (int)fminf(fmaxf(floorf(ix), -2.0f), (float)W + 1.0f);       // fmaxf / fminf drop NaN
    *y0 = (int)fminf(fmaxf(floorf(iy), -2.0f), (float)H + 1.0f);
    return sweep_taps(un, vn, H, W, half_w, half_h);
}

// Fast form of sweep_project + sweep_taps for the fused sweeps (one reciprocal + Newton step instead of four IEEE divisions, no
// normalize / un-normalize round trip, tap validity as four unsigned compares): the sampling position differs from the
// reference's by ~1e-4 px (the reference's own coordinates carry that much rounding from the round trip through [-1, 1]), which
// moves a bilinear sample by ~1e-4 of the local feature gradient - far inside the 1e-3 depth tolerance.  Taps outside the image
// keep offset 0 and weight 0 exactly as in sweep_taps.  rx, ry, rz = rot @ (x, y, 1) are passed in (computed once per pixel).
__device__ __forceinline__ Taps sweep_taps_fast(const float* __restrict__ rt, float rx, float ry, float rz, float d, int H, int W) {
    const float X0 = fmaf(rx, d, rt[9]), X1 = fmaf(ry, d, rt[10]), X2 = fmaf(rz, d, rt[11]);
    const float zz = X2 + 1e-6f;
    float rc = __builtin_amdgcn_rcpf(zz);
    rc = fmaf(rc, fmaf(-zz, rc, 1.0f), rc);
    // clamp to one texel outside the image: beyond that every tap is invalid anyway; fmaxf/fminf also drop NaN
    const float ix = fminf(fmaxf(X0 * rc, -2.0f), (float)W + 1.0f);
    const float iy = fminf(fmaxf(X1 * rc, -2.0f), (float)H + 1.0f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx = ix - x0f, wy = iy - y0f;
    const float ex = 1.0f - wx, ey = 1.0f - wy;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    const int r0 = vy0 ? y0 * W : 0, r1 = vy1 ? (y0 + 1) * W : 0;
    const int c0 = vx0 ? x0 : 0, c1 = vx1 ? x0 + 1 : 0;
    Taps t;
    t.o00 = r0 + c0;
    t.o01 = r0 + c1;
    t.o10 = r1 + c0;
    t.o11 = r1 + c1;
    t.w00 = (vx0 && vy0) ? ey * ex : 0.0f;
    t.w01 = (vx1 && vy0) ? ey * wx : 0.0f;
    t.w10 = (vx0 && vy1) ? wy * ex : 0.0f;
    t.w11 = (vx1 && vy1) ? wy * wx : 0.0f;
    return t;
}

// ---- buffer-descriptor gathers -----------------------------------------------------------------------------
// A source view's [C,H,W] block is addressed through one wave-uniform buffer descriptor: the per-lane tap
// offset goes in the 32-bit voffset, the channel plane (c*H*W*4 bytes) in the scalar soffset, so a gather
// costs no 64-bit address arithmetic and the compiler issues a whole batch of loads back to back behind
// counted vmcnt waits (with flat addressing hipcc serialized them: 2-4 loads per s_waitcnt vmcnt(0)).
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
}

// N consecutive channel planes starting at plane c0: issue all 4*N tap loads, then interpolate.
template <int N>
__device__ __forceinline__ void gather_bilinear(rsrc_t src, unsigned plane_bytes, int c0, const Taps& t, float (&out)[N]) {
    float v00[N], v01[N], v10[N], v11[N];
    const unsigned b00 = (unsigned)t.o00 * 4u, b01 = (unsigned)t.o01 * 4u, b10 = (unsigned)t.o10 * 4u, b11 = (unsigned)t.o11 * 4u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const unsigned so = (unsigned)(c0 + i) * plane_bytes;
        v00[i] = buf_load(src, b00, so);
        v01[i] = buf_load(src, b01, so);
        v10[i] = buf_load(src, b10, so);
        v11[i] = buf_load(src, b11, so);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float acc = v00[i] * t.w00;
        acc = fmaf(v01[i], t.w01, acc);
        acc = fmaf(v10[i], t.w10, acc);
        out[i] = fmaf(v11[i], t.w11, acc);
    }
}

__device__ __forceinline__ float bilinear(const float* __restrict__ plane, const Taps& t) {
    float acc = plane[t.o00] * t.w00;
    acc = fmaf(plane[t.o01], t.w01, acc);
    acc = fmaf(plane[t.o10], t.w10, acc);
    acc = fmaf(plane[t.o11], t.w11, acc);
    return acc;
}

__device__ __forceinline__ bool sweep_outside(float un, float vn, float z) {
    return (un > 1.0f) || (un < -1.0f) || (vn > 1.0f) || (vn < -1.0f) || (z <= 0.0f);
}

}  // namespace mvs
