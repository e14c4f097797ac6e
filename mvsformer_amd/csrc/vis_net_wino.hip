// Visibility-weight CNN of StageNet, second generation: the two 3x3 layers that carry 94 % of its FLOPs
// (ConvBnReLU(16,16), ConvBnReLU(16,8); models/mvsformer_model.py:37,91, ConvBnReLU = module.py:168-197) run as
// Winograd F(2x2,3x3) GEMMs on the fp32 matrix cores, still fused with layer 1 (VALU), the 1x1 conv and the sigmoid in
// ONE launch with every intermediate in LDS.  vis_net.hip (all-VALU, 58 TFLOP/s) stays as the reference implementation.
//
//   per 2x2 output tile and input channel: 4x4 patch -> B^T d B (32 adds, in the lane's registers) = the B operands of
//   16 transform-point GEMMs  Z_xi[cout, tile] += U_xi[cout, cin] X_xi[cin, tile]   (v_mfma_f32_16x16x4_f32:
//   M = 16 output channels, N = 16 tiles of one tile row, K = 4 input channels), then A^T Z A, BatchNorm, ReLU.
//   2.25x fewer MACs than the direct form; the transformed weights U (16 x 16 x 16 per layer) live in 2 x 64 VGPRs of
//   every lane for the whole kernel, which is why the kernel is persistent (blocks loop over output tiles).
//
// Block tile = 30 x 14 outputs: layer 2 is needed on 32 x 16 = 16 x 8 tiles (two tile rows per wavefront), layer 3 on
// 15 x 7 tiles.  LDS activations are stored per row as even columns then odd columns, so the 16 tile lanes of a patch
// read are unit-stride, and channel planes are padded to == 16 (mod 32) words: no bank conflicts on the operand path.
// Each conv zero-pads ITS OWN input: activations at positions outside the image are stored as 0.
#include <stdlib.h>

#include "conv_common.h"

namespace {
using namespace mvsconv;

constexpr int TW = 30, TH = 14;
constexpr int INW = TW + 6, INH = TH + 6;                  // entropy tile, halo 3
constexpr int A1W = TW + 4, A1H = TH + 4, A1O = A1W / 2;   // layer-1 output, halo 2; row = E[17] O[17]
constexpr int A2W = TW + 2, A2H = TH + 2, A2O = A2W / 2;   // layer-2 output, halo 1; row = E[16] O[16]
constexpr int A1PL = 624, A2PL = 528;                      // channel plane strides, == 16 (mod 32)
static_assert(A1PL >= A1W * A1H && A2PL >= A2W * A2H && A1PL % 32 == 16 && A2PL % 32 == 16, "plane padding");
constexpr int T2Y = A2H / 2, T3X = TW / 2, T3Y = TH / 2;   // 8 layer-2 tile rows (16 tiles each); 15 x 7 layer-3 tiles
static_assert(A2W / 2 == 16, "one layer-2 tile row = one MFMA N tile");
// offsets inside the MVS_VIS_PARAM_FLOATS block (vis_net.hip)
constexpr int OFF_W0 = 0, OFF_S0 = 144, OFF_B0 = 160, OFF_W1 = 176, OFF_S1 = 2480, OFF_B1 = 2496, OFF_W2 = 2512, OFF_S2 = 3664,
              OFF_B2 = 3672, OFF_W3 = 3680, OFF_B3 = 3688;

// prepared[layer][xi = a*4+b][j][lane]: lane = kk*16 + co holds U_xi[cin = 4j+kk][co] = sum G[a][kh] G[b][kw] w[co][cin][kh][kw]
__global__ void vis_wino_prepare_kernel(const float* __restrict__ prm, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MVS_VIS_WINO_FLOATS) return;
    const float G[4][3] = {{1.0f, 0.0f, 0.0f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.0f, 0.0f, 1.0f}};
    const int layer = idx / 4096, e = idx % 4096;
    const int lane = e % 64, j = (e / 64) % 4, xi = e / 256;
    const int co = lane & 15, ci = 4 * j + (lane >> 4), a = xi >> 2, b = xi & 3;
    float v = 0.0f;
    if (layer == 0 || co < 8)
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const int tap = kh * 3 + kw;
                const float w = layer == 0 ? prm[OFF_W1 + (ci * 9 + tap) * 16 + co] : prm[OFF_W2 + (ci * 9 + tap) * 8 + co];
                v += G[a][kh] * G[b][kw] * w;
            }
    out[idx] = v;
}

using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
// a - b in one packed instruction.  The backend has no packed fsub (a <2 x float> fsub is split into two v_sub_f32), so the
// subtraction is spelled b * -1 + a: exact product, one rounding - bit-identical to a - b.
// (The -1 comes out of an opaque asm so that the optimizer cannot fold the fma back into the fsub it would then split.)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    float m1;
    asm("s_mov_b32 %0, -1.0" : "=s"(m1));
    return pk_fma(b, f32x2{m1, m1}, a);
}

// X = B^T d B with the packed fp32 adds of gfx950 (v_pk_add_f32: two lanes' worth of adds per issue slot - this kernel is bound by
// vector-instruction issue, not by the matrix cores).  The patch arrives as E[q] = (d[q][0], d[q][2]) and O[q] = (d[q][1], d[q][3]),
// exactly the register pairs the ds_read2_b32 of the even / odd column halves deliver, so the row transform is 8 packed ops with no
// shuffles; the column transform packs (X1, X2) of each row (op_sel/neg modifiers) and leaves X0, X3 scalar: 20 instead of 32 ops.
__device__ __forceinline__ void xform(const f32x2 (&E)[4], const f32x2 (&O)[4], float (&X)[16]) {
    f32x2 T[4], U[4];                                       // T[a] = (t[a][0], t[a][2]), U[a] = (t[a][1], t[a][3])
    T[0] = pk_sub(E[0], E[2]);  U[0] = pk_sub(O[0], O[2]);
    T[1] = E[1] + E[2];         U[1] = O[1] + O[2];
    T[2] = pk_sub(E[2], E[1]);  U[2] = pk_sub(O[2], O[1]);
    T[3] = pk_sub(E[1], E[3]);  U[3] = pk_sub(O[1], O[3]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x2 m = pk_fma(f32x2{U[a].x, U[a].x}, f32x2{1.0f, -1.0f}, f32x2{T[a].y, T[a].y});     // (t1 + t2, t2 - t1)
        X[a * 4 + 0] = T[a].x - T[a].y;
        X[a * 4 + 1] = m.x;
        X[a * 4 + 2] = m.y;
        X[a * 4 + 3] = U[a].x - U[a].y;
    }
}

// one tile row: Z_xi = sum_j U[xi][j] x X_xi(patch of channel 4j+kk);  RS/OO = row stride / odd-column offset of the layout
template <int PL, int RS, int OO>
__device__ __forceinline__ void tile_row_gemm(const float* __restrict__ act, int ty, int i16, int kk, const float (&U)[16][4],
                                              f32x4 (&Z)[16]) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) Z[xi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float* p = act + kk * PL + (2 * ty) * RS + i16;
    f32x2 E[4], O[4];
    auto load_patch = [&](int j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            E[q] = f32x2{p[4 * j * PL + q * RS], p[4 * j * PL + q * RS + 1]};
            O[q] = f32x2{p[4 * j * PL + q * RS + OO], p[4 * j * PL + q * RS + OO + 1]};
        }
    };
    load_patch(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float X[16];
        xform(E, O, X);
        // software pipeline: the next channel group's patch is requested BEFORE this group's 16 MFMAs are issued, so the LDS latency
        // runs under ~500 cycles of matrix work instead of in front of it (the sched_barriers keep the compiler from sinking it back)
        __builtin_amdgcn_sched_barrier(0);
        if (j < 3) load_patch(j + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) Z[xi] = mfma4(U[xi][j], X[xi], Z[xi]);
        __builtin_amdgcn_sched_barrier(0);                  // the next transform (and its wait on the LDS) stays behind these MFMAs
    }
}

// A^T Z A for the accumulator row pair (2h, 2h+1): o[a][b] = (row 2h, row 2h+1) of output (a, b) inside the 2x2 tile; packed adds
// over the two rows (they sit in adjacent registers of every Z[xi])
__device__ __forceinline__ void out_xform(const f32x4 (&Z)[16], int h, f32x2 (&o)[2][2]) {
    auto z = [&](int xi) { return h ? f32x2{Z[xi][2], Z[xi][3]} : f32x2{Z[xi][0], Z[xi][1]}; };
    f32x2 s[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        s[0][b] = z(b) + z(4 + b) + z(8 + b);
        s[1][b] = pk_sub(pk_sub(z(4 + b), z(8 + b)), z(12 + b));
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        o[a][0] = s[a][0] + s[a][1] + s[a][2];
        o[a][1] = pk_sub(pk_sub(s[a][1], s[a][2]), s[a][3]);
    }
}

__global__ __launch_bounds__(256, 2) void vis_wino_kernel(const float* __restrict__ entropy, const float* __restrict__ prm,
                                                          const float* __restrict__ prep, int N, int H, int W, int ntx, int nty,
                                                          float* __restrict__ weight) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a1 = smem;                                    // [16][A1PL]
    float* s_a2 = smem + 16 * A1PL;                        // [16][A2PL]
    float* s_in = smem + 16 * (A1PL + A2PL);               // [INH][INW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;

    // Transform-domain weights of ONE layer at a time in registers (64 per lane), re-read from the 32 KB `prep` block (L1/L2 resident)
    // at the start of each phase: holding both layers' (128 registers) left the GEMM loops no room to keep LDS loads in flight.
    float U[16][4];
    auto load_weights = [&](int layer) {
        const float* src = prep + layer * 4096 + lane;
        asm volatile("" : "+v"(src));                      // opaque per call: keeps LICM from hoisting both layers' loads out of the tile loop
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int j = 0; j < 4; ++j) U[xi][j] = src[(xi * 4 + j) * 64];
    };
    float sc1[4], sh1[4], sc2[4], sh2[4], w3[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 4 * kk + r;
        sc1[r] = prm[OFF_S1 + co];
        sh1[r] = prm[OFF_B1 + co];
        sc2[r] = co < 8 ? prm[OFF_S2 + co] : 0.0f;
        sh2[r] = co < 8 ? prm[OFF_B2 + co] : 0.0f;
        w3[r] = co < 8 ? prm[OFF_W3 + co] : 0.0f;
    }
    const float b3 = prm[OFF_B3];
    // layer 1 as an MFMA A operand: lane (kk, co = i16) holds w0[tap = 4t + kk][co]; the matching B operand is the entropy at the
    // tap's offset inside the halo tile
    float W1A[3], sc0[4], sh0[4];
    int tap_off[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = 4 * t + kk;
        W1A[t] = tap < 9 ? prm[OFF_W0 + tap * 16 + i16] : 0.0f;
        const int tc = tap < 9 ? tap : 8;
        tap_off[t] = (tc / 3) * INW + tc % 3;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc0[r] = prm[OFF_S0 + 4 * kk + r];
        sh0[r] = prm[OFF_B0 + 4 * kk + r];
    }

    // Software pipeline over this block's tiles, two barriers per tile:
    //   phase A: layer 2 of tile t (s_a1 -> s_a2); the entropy of tile t+1, fetched into registers before, lands in s_in
    //   phase B: layer 3 of tile t (s_a2 -> global) and layer 1 of tile t+1 (s_in -> s_a1): matrix and vector pipes overlap
    const int ntiles = N * ntx * nty;
    constexpr int EPT = (INH * INW + 255) / 256;           // entropy values per thread (3)
    float pre[EPT];
    auto tile_origin = [&](int tile, int& n, int& x0, int& y0) {
        n = tile / (ntx * nty);
        y0 = ((tile / ntx) % nty) * TH;
        x0 = (tile % ntx) * TW;
    };
    auto fetch_entropy = [&](int tile) {
        int n, x0, y0;
        tile_origin(tile, n, x0, y0);
        const float* src = entropy + (size_t)n * H * W;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * 256;
            const int gy = y0 - 3 + i / INW, gx = x0 - 3 + i % INW;
            pre[e] = (i < INH * INW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : 0.0f;
        }
    };
    auto commit_entropy = [&]() {
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (tid + e * 256 < INH * INW) s_in[tid + e * 256] = pre[e];
    };
    // Layer 1 (1 -> 16 channels, 3x3) on the matrix cores as well: M = 16 output channels, N = 16 consecutive pixels of the
    // flattened 34 x 18 region, K = the 9 taps padded to 12 (three K = 4 steps; lane kk supplies tap 4t + kk, taps 9-11 carry zero
    // weights).  In the VALU form its 144 multiply-adds per pixel, with the weights in (spilled) scalar registers, cost as many
    // vector instructions as both Winograd layers' transforms together.
    auto layer1 = [&](int tile) {
        int n, x0, y0;
        tile_origin(tile, n, x0, y0);
        constexpr int NPIX = A1W * A1H, NT1 = (NPIX + 15) / 16;
#pragma unroll 1
        for (int nt = wave; nt < NT1; nt += 4) {
            const int pidx = min(nt * 16 + i16, NPIX - 1);
            const int py = pidx / A1W, px = pidx % A1W;
            const float* src = s_in + py * INW + px;
            f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 3; ++t) z = mfma4(W1A[t], src[tap_off[t]], z);
            const int gy = y0 - 2 + py, gx = x0 - 2 + px;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (nt * 16 + i16 < NPIX) {
                float* dst = s_a1 + (4 * kk) * A1PL + py * A1W + (px & 1) * A1O + (px >> 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * A1PL] = inside ? fmaxf(fmaf(z[r], sc0[r], sh0[r]), 0.0f) : 0.0f;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    fetch_entropy(tile);
    commit_entropy();
    __syncthreads();
    layer1(tile);
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        int n, x0, y0;
        tile_origin(tile, n, x0, y0);
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;                // block-uniform
        if (has_next) fetch_entropy(next);

        // ---- phase A: layer 2, 16 -> 16 on 16 x 8 tiles, two tile rows per wavefront ----
        load_weights(0);
#pragma unroll 1
        for (int ty = wave; ty < T2Y; ty += 4) {
            f32x4 Z[16];
            tile_row_gemm<A1PL, A1W, A1O>(s_a1, ty, i16, kk, U, Z);
            bool in_img[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int gy = y0 - 1 + 2 * ty + a, gx = x0 - 1 + 2 * i16 + b;
                    in_img[a][b] = gy >= 0 && gy < H && gx >= 0 && gx < W;
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 o[2][2];
                out_xform(Z, h, o);
                const f32x2 sc = {sc1[2 * h], sc1[2 * h + 1]}, sh = {sh1[2 * h], sh1[2 * h + 1]};
                float* dst = s_a2 + (4 * kk + 2 * h) * A2PL + (2 * ty) * A2W + i16;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f32x2 v = pk_fma(o[a][b], sc, sh);
                        dst[a * A2W + b * A2O] = in_img[a][b] ? fmaxf(v.x, 0.0f) : 0.0f;
                        dst[A2PL + a * A2W + b * A2O] = in_img[a][b] ? fmaxf(v.y, 0.0f) : 0.0f;
                    }
            }
        }
        if (has_next) commit_entropy();
        __syncthreads();

        // ---- phase B: layer 3, 16 -> 8 on 15 x 7 tiles (+ 1x1 conv + sigmoid), then layer 1 of the next tile ----
        load_weights(1);
#pragma unroll 1
        for (int ty = wave; ty < T3Y; ty += 4) {
            f32x4 Z[16];
            tile_row_gemm<A2PL, A2W, A2O>(s_a2, ty, i16, kk, U, Z);
            float part[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 o[2][2];
                out_xform(Z, h, o);
                const f32x2 sc = {sc2[2 * h], sc2[2 * h + 1]}, sh = {sh2[2 * h], sh2[2 * h + 1]};
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f32x2 v = pk_fma(o[a][b], sc, sh);
                        part[a][b] = fmaf(w3[2 * h], fmaxf(v.x, 0.0f), part[a][b]);
                        part[a][b] = fmaf(w3[2 * h + 1], fmaxf(v.y, 0.0f), part[a][b]);
                    }
            }
            // channels 0-3 live in lanes kk = 0, channels 4-7 in kk = 1 (kk = 2, 3 hold the zero padding rows)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) part[a][b] += __shfl_xor(part[a][b], 16, 64);
            if (kk == 0 && i16 < T3X) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int gy = y0 + 2 * ty + a;
                    if (gy >= H) continue;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int gx = x0 + 2 * i16 + b;
                        if (gx < W) weight[(size_t)n * H * W + (size_t)gy * W + gx] = 1.0f / (1.0f + expf(-(part[a][b] + b3)));
                    }
                }
            }
        }
        if (has_next) layer1(next);
        __syncthreads();
    }
}

}  // namespace

extern "C" int mvs_vis_wino_prepare(const float* params, float* prepared, mvs_stream_t stream) {
    MVS_REQUIRE(params && prepared, "mvs_vis_wino_prepare: null pointer");
    hipLaunchKernelGGL(vis_wino_prepare_kernel, dim3(mvs::ceil_div(MVS_VIS_WINO_FLOATS, 256)), dim3(256), 0, MVS_STREAM(stream), params,
                       prepared);
    return mvs::finish_launch("mvs_vis_wino_prepare");
}

extern "C" int mvs_vis_wino_fwd(const float* entropy, const float* params, const float* prepared, int N, int H, int W, float* weight,
                                mvs_stream_t stream) {
    MVS_REQUIRE(entropy && params && prepared && weight, "mvs_vis_wino_fwd: null pointer");
    MVS_REQUIRE(N >= 1 && H >= 1 && W >= 1, "mvs_vis_wino_fwd: bad shape N=%d H=%d W=%d", N, H, W);
    const int ntx = mvs::ceil_div(W, TW), nty = mvs::ceil_div(H, TH);
    MVS_REQUIRE((int64_t)N * ntx * nty < ((int64_t)1 << 31), "mvs_vis_wino_fwd: too many tiles");
    const int ncu = mvs::device_cus();
    const int ntiles = N * ntx * nty;
    const int blocks = ntiles < 2 * ncu ? ntiles : 2 * ncu;       // persistent: two resident blocks per CU
    constexpr size_t lds = (size_t)(16 * (A1PL + A2PL) + INH * INW) * sizeof(float);
    {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(vis_wino_kernel), (int)lds, "mvs_vis_wino_fwd");
        if (rc != MVS_OK) return rc;
    }
    hipLaunchKernelGGL(vis_wino_kernel, dim3(blocks), dim3(256), lds, MVS_STREAM(stream), entropy, params, prepared, N, H, W, ntx, nty,
                       weight);
    return mvs::finish_launch("mvs_vis_wino_fwd");
}
