// Weight gradient of the 3-D (transposed) convolutions on the fp32 matrix cores:
//     dW[a][b][k] = sum over batch, small-grid voxels p of  A[a, p] * Bt[b, p*s - 1 + k],   k = (kd,kh,kw) in 3x3x3
// Conv3d (reference module.py:83-123):          A = dY [Cout grid], Bt = X,  dW layout [Cout,Cin,27] = nn.Conv3d.weight
// ConvTranspose3d (module.py:126-165,562-575):   A = X  [Cin grid],  Bt = dY, dW layout [Cin,Cout,27] = its weight
// i.e. one kernel, the "small" grid is always the one the stride divides.
//
// GEMM view: M = CA rows (16 per tile), N = (b, k) columns — a block owns 4 channels of Bt = 108 columns = 7 N tiles —
// K = voxels, 4 consecutive voxels along W per v_mfma_f32_16x16x4_f32.  Fragments are fetched straight from global
// memory with buffer loads (per step: MT loads of A, 7 loads of Bt feeding 7*MT MFMAs = 224*MT matrix-pipe cycles per
// 11 loads, rows are re-touched across kh taps and stay in L1/L2); zero padding comes from the descriptor bounds check.
// A wavefront walks many rows and keeps the whole 16*MT x 112 slab of dW in registers, then adds it to global dW with
// fp32 atomics once (a few thousand atomics per wavefront instead of one per voxel).
#include "conv_common.h"

namespace {
using namespace mvsconv;

template <int MT>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ dW,
                                                    int CA, int CB, int Dp, int Hp, int Wp, int Db, int Hb, int Wb, int sd, int shw,
                                                    int nbatch) {
    constexpr int NTL = 7;                                   // 4 channels x 27 taps = 108 columns -> 7 tiles of 16
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int b0 = blockIdx.x * 4;

    // per-lane column decode for each N tile: column n = t*16 + i16 -> (channel b0 + n/27, tap n%27)
    int colbase[NTL], ckd[NTL], ckh[NTL], ckw[NTL];
    bool colok[NTL];
    const size_t planeB = (size_t)Hb * Wb;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = t * 16 + i16, bl = n / 27, k = n % 27;
        ckd[t] = k / 9;
        ckh[t] = (k / 3) % 3;
        ckw[t] = k % 3;
        colok[t] = n < 108 && b0 + bl < CB;
        colbase[t] = (int)((size_t)(b0 + bl) * Db * planeB);
    }
    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t planeA = (size_t)Hp * Wp;
    const int rows = nbatch * Dp * Hp;
    const int stride_rows = gridDim.y * NWAVES;
    for (int row = blockIdx.y * NWAVES + wave; row < rows; row += stride_rows) {
        const int n = row / (Dp * Hp), d = (row / Hp) % Dp, h = row % Hp;
        const rsrc_t ra = make_rsrc(A + (size_t)n * CA * Dp * planeA, (unsigned)((size_t)CA * Dp * planeA * 4));
        const rsrc_t rb = make_rsrc(Bt + (size_t)n * CB * Db * planeB, (unsigned)((size_t)CB * Db * planeB * 4));
        // row-dependent parts
        int aoff[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int a = m * 16 + i16;
            aoff[m] = (a < CA) ? (int)((((size_t)a * Dp + d) * Hp + h) * Wp) : -1;
        }
        int boff[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const int zb = d * sd - 1 + ckd[t], yb = h * shw - 1 + ckh[t];
            boff[t] = (colok[t] && zb >= 0 && zb < Db && yb >= 0 && yb < Hb) ? colbase[t] + (int)((size_t)zb * planeB + (size_t)yb * Wb) : -1;
        }
        for (int w0 = 0; w0 < Wp; w0 += 4) {
            const int wv = w0 + kk;
            float af[MT], bf[NTL];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = buf_load(ra, (aoff[m] >= 0 && wv < Wp) ? (unsigned)(aoff[m] + wv) * 4u : OOB, 0);
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const int xb = wv * shw - 1 + ckw[t];
                bf[t] = buf_load(rb, (boff[t] >= 0 && wv < Wp && xb >= 0 && xb < Wb) ? (unsigned)(boff[t] + xb) * 4u : OOB, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[m][t] = mfma4(af[m], bf[t], acc[m][t]);
        }
    }
    // dW[(a*CB + b)*27 + k] += acc   (D layout: column = lane&15, row = (lane>>4)*4 + reg)
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = t * 16 + i16, bl = n / 27, k = n % 27;
        if (!(n < 108 && b0 + bl < CB)) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = m * 16 + kk * 4 + r;
                if (a < CA) atomicAdd(&dW[((size_t)a * CB + b0 + bl) * 27 + k], acc[m][t][r]);
            }
    }
}

}  // namespace

extern "C" int mvs_conv3d_wgrad(const float* A, const float* Bt, float* dW, int nbatch, int CA, int CB, int Dp, int Hp, int Wp, int Db,
                                int Hb, int Wb, int sd, int shw, mvs_stream_t stream) {
    MVS_REQUIRE(A && Bt && dW, "mvs_conv3d_wgrad: null pointer");
    MVS_REQUIRE(nbatch >= 1 && CA >= 1 && CA <= 64 && CB >= 1 && Dp >= 1 && Hp >= 1 && Wp >= 1, "mvs_conv3d_wgrad: bad shape");
    MVS_REQUIRE((sd == 1 || sd == 2) && (shw == 1 || shw == 2), "mvs_conv3d_wgrad: stride (%d,%d,%d) not built", sd, shw, shw);
    MVS_REQUIRE((int64_t)CA * Dp * Hp * Wp * 4 < ((int64_t)1 << 31) && (int64_t)CB * Db * Hb * Wb * 4 < ((int64_t)1 << 31),
                "mvs_conv3d_wgrad: one batch item exceeds the 2 GiB buffer window");
    const int nchunk = mvs::ceil_div(CB, 4);
    const int rows = nbatch * Dp * Hp;
    int gy = mvs::ceil_div(1024, nchunk);
    if (gy > mvs::ceil_div(rows, mvsconv::NWAVES)) gy = mvs::ceil_div(rows, mvsconv::NWAVES);
    if (gy < 1) gy = 1;
    dim3 grid(nchunk, gy);
    hipStream_t s = MVS_STREAM(stream);
    const int mt = mvs::ceil_div(CA, 16);
#define MVS_WG(M) hipLaunchKernelGGL(wgrad_kernel<M>, grid, dim3(256), 0, s, A, Bt, dW, CA, CB, Dp, Hp, Wp, Db, Hb, Wb, sd, shw, nbatch)
    switch (mt) {
        case 1: MVS_WG(1); break;
        case 2: MVS_WG(2); break;
        case 3: MVS_WG(3); break;
        default: MVS_WG(4); break;
    }
#undef MVS_WG
    return mvs::finish_launch("mvs_conv3d_wgrad");
}
