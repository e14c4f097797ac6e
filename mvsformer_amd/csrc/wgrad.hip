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
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------
// Tiled form (default).  The kernel above spends one vector-memory instruction per MFMA operand (~9 TFLOP/s).  Here a
// block owns 4 Bt channels (108 columns) and walks output tiles of TH rows x SEG voxels of one (batch, depth): the Bt
// tile — 4 channels x 3 depth taps x (TH*SHW+2) rows x (SEG*SHW+2) columns, every staged row serving the three kh taps
// of up to three output rows — and the A tile (16*MT channels x TH x SEG) go through LDS, the next tile is prefetched
// into registers under the MFMAs, and each wavefront reads its fragments with one ds_read_b32 per operand.
//   SHW = 1: TH x SEG = 8x64 (MT=1), 4x64 (MT=2), 2x64 (MT=3,4)        SHW = 2: 4x32
// ---------------------------------------------------------------------------------------------------------------
template <int MT, int SHW>
struct WgTile {
    static constexpr int SEG = SHW == 1 ? 64 : 32;
    static constexpr int TH = SHW == 2 ? 4 : (MT == 1 ? 8 : (MT == 2 ? 4 : 2));
    static constexpr int TBR = TH * SHW + 2, TBC = SEG * SHW + 2;     // Bt tile rows / columns (10 x 66 at most)
    static constexpr int TBCP = 68;                                   // Bt row stride (== 4 mod 32)
    static constexpr int BPS = TBR * TBCP + ((12 - (TBR * TBCP) % 32) + 32) % 32;   // (channel, kd) plane stride == 12 (mod 32)
    static constexpr int AS = TH * SEG + 1;                           // A channel stride (odd)
    static constexpr int NA = 16 * MT * AS, NBF = 12 * BPS;
    static constexpr int BPT = (TBC + 1) / 2;                         // Bt columns per staging thread (2 threads per row)
    static constexpr int ACH = SEG / 4;                               // float4 chunks per A row
    static constexpr int APASS = (16 * MT * TH * ACH + 255) / 256;    // A float4 per thread
};

template <int MT, int SHW>
__global__ __launch_bounds__(256) void wgrad_tiled_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                          float* __restrict__ dW, int CA, int CB, int Dp, int Hp, int Wp, int Db, int Hb,
                                                          int Wb, int sd, int nbatch) {
    using T = WgTile<MT, SHW>;
    constexpr int NTL = 7, SEG = T::SEG, TH = T::TH;
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];
    float* sA = wg_smem;
    float* sB = wg_smem + T::NA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int b0 = blockIdx.x * 4;
    const size_t planeA = (size_t)Hp * Wp, planeB = (size_t)Hb * Wb;

    // fragment read offsets of this lane's 7 columns: n = t*16 + i16 -> (bl, kd, kh, kw)
    int bcol[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = t * 16 + i16;
        const int pl = n / 9, kh = (n / 3) % 3, kw = n % 3;          // pl = bl*3 + kd
        bcol[t] = n < 108 ? pl * T::BPS + kh * T::TBCP + kw : 0;
    }
    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging roles.  Bt: 12 planes x TBR rows, two threads per row (left / right half of the columns)
    const int brow = tid >> 1, bhalf = tid & 1;
    const int bpl = brow / T::TBR, brr = brow % T::TBR;             // plane = bl*3 + kd, row inside the tile
    const bool brow_ok = brow < 12 * T::TBR && b0 + bpl / 3 < CB;
    float pb[T::BPT];
    f32x4 pa[T::APASS];

    const int hblocks = (Hp + TH - 1) / TH, wblocks = (Wp + SEG - 1) / SEG;
    const long long ntiles = (long long)nbatch * Dp * hblocks * wblocks;
    auto fetch = [&](long long tl) {
        const int w0 = (int)(tl % wblocks) * SEG;
        const int h0 = (int)((tl / wblocks) % hblocks) * TH;
        const int d = (int)((tl / ((long long)wblocks * hblocks)) % Dp);
        const int n = (int)(tl / ((long long)wblocks * hblocks * Dp));
        const rsrc_t ra = make_rsrc(A + (size_t)n * CA * Dp * planeA, (unsigned)((size_t)CA * Dp * planeA * 4));
        const rsrc_t rb = make_rsrc(Bt + (size_t)n * CB * Db * planeB, (unsigned)((size_t)CB * Db * planeB * 4));
        const int zb = d * sd - 1 + bpl % 3, yb = h0 * SHW - 1 + brr;
        const bool rok = brow_ok && zb >= 0 && zb < Db && yb >= 0 && yb < Hb;
        const int rbase = rok ? (int)(((size_t)(b0 + bpl / 3) * Db + zb) * planeB + (size_t)yb * Wb) : 0;
        const int x0 = w0 * SHW - 1 + bhalf * T::BPT;
#pragma unroll
        for (int j = 0; j < T::BPT; ++j) {
            const int xb = x0 + j;
            pb[j] = buf_load(rb, (rok && bhalf * T::BPT + j < T::TBC && xb >= 0 && xb < Wb) ? (unsigned)(rbase + xb) * 4u : OOB, 0);
        }
#pragma unroll
        for (int j = 0; j < T::APASS; ++j) {
            const int q = tid + j * 256;                              // float4 index: (a, r, chunk)
            const int a = q / (TH * T::ACH), r = (q / T::ACH) % TH, c4 = (q % T::ACH) * 4;
            const int h = h0 + r, w = w0 + c4;
            const bool ok = q < 16 * MT * TH * T::ACH && a < CA && h < Hp && w < Wp;
            const unsigned off = ok ? (unsigned)((((size_t)a * Dp + d) * Hp + h) * Wp + w) * 4u : OOB;
            if ((Wp & 3) == 0) {
                pa[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) pa[j][e] = buf_load(ra, (ok && w + e < Wp) ? off + 4u * e : OOB, 0);
            }
        }
    };
    auto commit = [&]() {
        if (brow < 12 * T::TBR) {
#pragma unroll
            for (int j = 0; j < T::BPT; ++j)
                if (bhalf * T::BPT + j < T::TBCP) sB[bpl * T::BPS + brr * T::TBCP + bhalf * T::BPT + j] = pb[j];
        }
#pragma unroll
        for (int j = 0; j < T::APASS; ++j) {
            const int q = tid + j * 256;
            if (q < 16 * MT * TH * T::ACH) {
                const int a = q / (TH * T::ACH), r = (q / T::ACH) % TH, c4 = (q % T::ACH) * 4;
                float* dst = sA + a * T::AS + r * SEG + c4;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = pa[j][e];
            }
        }
    };

    long long tl = blockIdx.y;
    if (tl >= ntiles) return;
    fetch(tl);
    commit();
    __syncthreads();
    constexpr int KSTEPS = TH * SEG / 4;                              // K steps per tile, KSTEPS/4 per wavefront
    for (;;) {
        const long long nxt = tl + gridDim.y;
        const bool has_next = nxt < ntiles;                           // block-uniform
        if (has_next) fetch(nxt);
        for (int s = wave; s < KSTEPS; s += 4) {
            const int r = s / (SEG / 4), c = (s % (SEG / 4)) * 4 + kk;   // voxel (row r, column c) of the tile
            float af[MT], bf[NTL];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = sA[(m * 16 + i16) * T::AS + r * SEG + c];
            const int bo = r * SHW * T::TBCP + c * SHW;
#pragma unroll
            for (int t = 0; t < NTL; ++t) bf[t] = sB[bcol[t] + bo];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[m][t] = mfma4(af[m], bf[t], acc[m][t]);
        }
        __syncthreads();
        if (!has_next) break;
        commit();
        __syncthreads();
        tl = nxt;
    }
    // Sum the four wavefronts' slabs in LDS (the tile buffers are free now), then ONE wavefront's worth of atomics per block:
    // dW has only CA*CB*27 addresses, and same-address fp32 atomics serialize in L2 — with every wavefront of ~800 blocks
    // adding its own slab the tail of atomics, not the MFMA loop, set the kernel time.
    // dW[(a*CB + b)*27 + k] += acc   (D layout: column = lane&15, row = (lane>>4)*4 + reg)
    float* red = wg_smem;                                            // [MT][NTL][4][64] floats = 7168*MT <= NA + NBF
    static_assert(MT * NTL * 256 <= T::NA + T::NBF, "reduction buffer must fit in the tile buffers");
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NTL; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* slot = red + ((m * NTL + t) * 4 + r) * 64 + lane;
                        *slot = (w == 1) ? acc[m][t][r] : *slot + acc[m][t][r];
                    }
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = t * 16 + i16, bl = n / 27, k = n % 27;
        if (!(n < 108 && b0 + bl < CB)) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = m * 16 + kk * 4 + r;
                if (a < CA) atomicAdd(&dW[((size_t)a * CB + b0 + bl) * 27 + k], acc[m][t][r] + red[((m * NTL + t) * 4 + r) * 64 + lane]);
            }
    }
}

template <int MT, int SHW>
int launch_wgrad_tiled(const float* A, const float* Bt, float* dW, int nbatch, int CA, int CB, int Dp, int Hp, int Wp, int Db, int Hb,
                        int Wb, int sd, hipStream_t s) {
    using T = WgTile<MT, SHW>;
    const int nchunk = mvs::ceil_div(CB, 4);
    const long long ntiles = (long long)nbatch * Dp * mvs::ceil_div(Hp, T::TH) * mvs::ceil_div(Wp, T::SEG);
    long long gy = mvs::ceil_div(512, nchunk);                        // ~2 blocks per CU in total: few adders per dW address (256: slower)
    if (gy > ntiles) gy = ntiles;
    if (gy < 1) gy = 1;
    constexpr size_t lds = (size_t)(T::NA + T::NBF) * sizeof(float);
    if (lds > 48 * 1024) {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad_tiled_kernel<MT, SHW>), (int)lds, "mvs_conv3d_wgrad");
        if (rc != MVS_OK) return rc;
    }
    hipLaunchKernelGGL((wgrad_tiled_kernel<MT, SHW>), dim3(nchunk, (unsigned)gy), dim3(256), lds, s, A, Bt, dW, CA, CB, Dp, Hp, Wp, Db, Hb,
                       Wb, sd, nbatch);
    return MVS_OK;
}

}  // namespace

extern "C" int mvs_conv3d_wgrad(const float* A, const float* Bt, float* dW, int nbatch, int CA, int CB, int Dp, int Hp, int Wp, int Db,
                                int Hb, int Wb, int sd, int shw, mvs_stream_t stream) {
    MVS_REQUIRE(A && Bt && dW, "mvs_conv3d_wgrad: null pointer");
    MVS_REQUIRE(nbatch >= 1 && CA >= 1 && CA <= 64 && CB >= 1 && Dp >= 1 && Hp >= 1 && Wp >= 1, "mvs_conv3d_wgrad: bad shape");
    MVS_REQUIRE((sd == 1 || sd == 2) && (shw == 1 || shw == 2), "mvs_conv3d_wgrad: stride (%d,%d,%d) not built", sd, shw, shw);
    MVS_REQUIRE((int64_t)CA * Dp * Hp * Wp * 4 < ((int64_t)1 << 31) && (int64_t)CB * Db * Hb * Wb * 4 < ((int64_t)1 << 31),
                "mvs_conv3d_wgrad: one batch item exceeds the 2 GiB buffer window");
    if (!(getenv("MVS_WGRAD_V1") && atoi(getenv("MVS_WGRAD_V1")))) {
        hipStream_t s2 = MVS_STREAM(stream);
        const int mt2 = mvs::ceil_div(CA, 16);
        int rc2 = MVS_OK;
#define MVS_WGT(M)                                                                                                       \
    if (shw == 1) rc2 = launch_wgrad_tiled<M, 1>(A, Bt, dW, nbatch, CA, CB, Dp, Hp, Wp, Db, Hb, Wb, sd, s2);              \
    else rc2 = launch_wgrad_tiled<M, 2>(A, Bt, dW, nbatch, CA, CB, Dp, Hp, Wp, Db, Hb, Wb, sd, s2)
        switch (mt2) {
            case 1: MVS_WGT(1); break;
            case 2: MVS_WGT(2); break;
            case 3: MVS_WGT(3); break;
            default: MVS_WGT(4); break;
        }
#undef MVS_WGT
        if (rc2 != MVS_OK) return rc2;
        return mvs::finish_launch("mvs_conv3d_wgrad");
    }
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
