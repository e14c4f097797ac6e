// Projection prep: M = P_src * inverse(P_ref) for every (batch, source view) in ONE tiny launch.
// Stands in for models/mvsformer_model.py:69-72 (K*E composition, twice per source view) and
// models/warping.py:80-82 (torch.inverse, a host-synchronizing LU, + matmul per source view).
//
// One thread per (b, v).  The composition K[:3,:3] @ E[:3,:4] is done in fp32 like the reference's matmul;
// the 4x4 inverse and the product run in fp64 (Gauss-Jordan with partial pivoting) and are rounded once,
// so the result is at least as close to the exact M as the reference's fp32 LU.
#include "common.h"

namespace {

__device__ void compose_f32(const float* pair, double* P) {
    const float* E = pair;
    const float* K = pair + 16;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = K[i * 4 + 0] * E[0 * 4 + j];
            acc = fmaf(K[i * 4 + 1], E[1 * 4 + j], acc);
            acc = fmaf(K[i * 4 + 2], E[2 * 4 + j], acc);
            P[i * 4 + j] = (double)acc;
        }
    for (int j = 0; j < 4; ++j) P[12 + j] = (double)E[12 + j];
}

__device__ void invert4(const double* A, double* inv) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = A[i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); p = r; }
        if (p != c)
            for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const double piv = 1.0 / a[c][c];            // singular input -> inf/nan propagate, as LAPACK would
        for (int j = 0; j < 8; ++j) a[c][j] *= piv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
}

__device__ void relative_rt(const double* Ps, const double* Pr, float* rt) {
    double inv[16];
    invert4(Pr, inv);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += Ps[i * 4 + k] * inv[k * 4 + j];
            if (j < 3) rt[i * 3 + j] = (float)acc;
            else rt[9 + i] = (float)acc;
        }
    }
}

__global__ void proj_prepare_kernel(const float* __restrict__ proj, int B, int V, float* __restrict__ rt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * (V - 1)) return;
    const int b = idx / (V - 1), v = idx % (V - 1) + 1;
    double Pr[16], Ps[16];
    compose_f32(proj + (size_t)(b * V) * 32, Pr);
    compose_f32(proj + (size_t)(b * V + v) * 32, Ps);
    relative_rt(Ps, Pr, rt + (size_t)idx * 12);
}

__global__ void proj_relative_kernel(const float* __restrict__ src, const float* __restrict__ ref, int B,
                                     float* __restrict__ rt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double Pr[16], Ps[16];
    for (int i = 0; i < 16; ++i) { Pr[i] = (double)ref[b * 16 + i]; Ps[i] = (double)src[b * 16 + i]; }
    relative_rt(Ps, Pr, rt + (size_t)b * 12);
}

}  // namespace

extern "C" int mvs_proj_prepare(const float* proj, int B, int V, float* rt, mvs_stream_t stream) {
    MVS_REQUIRE(proj && rt, "mvs_proj_prepare: null pointer");
    MVS_REQUIRE(B >= 1 && V >= 2, "mvs_proj_prepare: need B>=1, V>=2 (got B=%d V=%d)", B, V);
    const int n = B * (V - 1);
    hipLaunchKernelGGL(proj_prepare_kernel, dim3(mvs::ceil_div(n, 64)), dim3(64), 0, MVS_STREAM(stream), proj, B, V, rt);
    return mvs::finish_launch("mvs_proj_prepare");
}

extern "C" int mvs_proj_relative(const float* src_proj, const float* ref_proj, int B, float* rt, mvs_stream_t stream) {
    MVS_REQUIRE(src_proj && ref_proj && rt, "mvs_proj_relative: null pointer");
    MVS_REQUIRE(B >= 1, "mvs_proj_relative: B=%d", B);
    hipLaunchKernelGGL(proj_relative_kernel, dim3(mvs::ceil_div(B, 64)), dim3(64), 0, MVS_STREAM(stream), src_proj,
                       ref_proj, B, rt);
    return mvs::finish_launch("mvs_proj_relative");
}
