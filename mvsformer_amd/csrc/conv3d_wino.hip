// Stride-1 3x3x3 convolution (+ folded BatchNorm + ReLU + residual) with the (H, W) taps in Winograd F(2x2, 3x3) form
// on the fp32 matrix cores — the conv2 / conv4 / conv6 layers of CostRegNet / CostRegNet3D (reference
// models/module.py:475-481, 554-560; Conv3d = conv -> BatchNorm3d -> ReLU, module.py:83-123).
//
// Those three layers are half of the regularizer's FLOPs.  Per 2x2 output tile and depth tap the 9 (kh, kw) products
// collapse to 16 transform-domain products for 4 outputs: 2.25x fewer MACs than the implicit GEMM of conv3d_fwd.hip.
// The depth taps stay direct (they become part of the GEMM's K together with the input channels), so there is no
// transform along D, no constraint on D, and the output transform runs once per tile after the whole K loop.
//
//   out tile (2x2)  = A^T [ sum_{kd, cin}  U[kd][cin->cout]  (.)  B^T d[kd][cin] B ] A          (.) = per transform point xi
//   GEMM per xi (16 of them):  Z_xi[cout, tile] += U_xi[cout, (kd,cin)] * X_xi[(kd,cin), tile]
//   v_mfma_f32_16x16x4_f32:    M = 16 output channels, N = 16 tiles along W, K = 4 input channels
//
// A wavefront owns 16 consecutive tiles of one tile row (32 output columns x 2 rows) of one output depth plane and
// NT*16 output channels: 16*NT accumulator tiles (64*NT VGPRs).  A lane is (tile = lane&15, channel-in-chunk = lane>>4):
// it reads its own 4x4 input patch from LDS, transforms it in registers (32 adds) and the 16 results ARE the B operands
// of the 16 xi-GEMMs — no cross-lane traffic, no transformed data in LDS or HBM.  A block = 4 wavefronts = 4 tile rows
// (8 x 32 outputs); per chunk of 4 input channels it stages the raw input (3 depth planes x 10 rows x 34 columns) and the
// transformed-weight slab U[kd][xi][4][NT*16] in LDS, the next chunk being prefetched into registers under the MFMAs.
//
// Numerics: F(2x2,3x3) uses only +-1 and 1/2 coefficients; fp32 results differ from the direct sum by a few ulp of the
// accumulated magnitude (tests/test_hip_parity.py::test_wino_conv_matches_direct).
#include <stdlib.h>

#include "conv_common.h"

namespace {
using namespace mvsconv;

using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
// a - b as ONE packed instruction: b * -1 + a (exact product, one rounding = a - b); the -1 comes out of an opaque asm so that the
// optimizer cannot fold the fma back into the <2 x float> fsub the backend would split in two
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    float m1;
    asm("s_mov_b32 %0, -1.0" : "=s"(m1));
    return pk_fma(b, f32x2{m1, m1}, a);
}

constexpr int WTY = 4;                       // tile rows per block = wavefronts
constexpr int RROWS = 2 * WTY + 2;           // staged input rows (halo 1 each side)
// LDS row = the 34 staged columns c = x - (x0-1) split by parity: E[j] = column 2j (j = 0..16), O[j] = column 2j+1.
// Tile i needs columns 2i..2i+3 = E[i], O[i], E[i+1], O[i+1]: unit stride over the 16 tile lanes, and with a channel
// stride == 16 (mod 32) words the 32 lanes of a half-wave (2 channels x 16 tiles) hit 32 distinct banks.
constexpr int ROFF_O = 18;                   // O[] starts here inside a row
constexpr int RCOLS = 36;
constexpr int RPLANE = RROWS * RCOLS;        // one depth plane of one channel
constexpr int RCH = 3 * RPLANE + 24;         // 1104 == 16 (mod 32)
constexpr int RAW_FLOATS = 4 * RCH;
constexpr int NQUAD = 4 * 3 * RROWS * 8;     // aligned float4 loads per chunk (960)
constexpr int NSINGLE = 4 * 3 * RROWS * 2;   // the x0-1 / x0+32 columns (240)
constexpr int QPT = (NQUAD + 255) / 256;     // 4

// ---- weight transform: U[c4][kd][xi = a*4+b][k][n] = sum_{kh,kw} G[a][kh] G[b][kw] w[n][4*c4+k][kd][kh][kw] ----
__global__ void wino_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ out) {
    const int64_t total = (int64_t)(Cin / 4) * 3 * 16 * 4 * Cout;
    const float G[4][3] = {{1.0f, 0.0f, 0.0f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.0f, 0.0f, 1.0f}};
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % Cout);
        const int k = (int)((idx / Cout) % 4);
        const int xi = (int)((idx / ((int64_t)Cout * 4)) % 16);
        const int kd = (int)((idx / ((int64_t)Cout * 64)) % 3);
        const int c4 = (int)(idx / ((int64_t)Cout * 192));
        const float* g = w + (((size_t)n * Cin + c4 * 4 + k) * 3 + kd) * 9;
        const int a = xi >> 2, b = xi & 3;
        float v = 0.0f;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) v += G[a][kh] * G[b][kw] * g[kh * 3 + kw];
        out[idx] = v;
    }
}

template <int NT>
__global__ __launch_bounds__(256) void wino_conv3d_kernel(const float* __restrict__ x, const float* __restrict__ U,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ res, float* __restrict__ y, int CIN, int COUT,
                                                          int D, int H, int W, int relu) {
    constexpr int NB = 16 * NT;                              // output channels of this block
    constexpr int USLAB = 3 * 16 * 4 * NB;                   // staged weights per chunk
    constexpr int UQ = USLAB / 4 / 256;                      // float4 per thread (3*NT)
    static_assert(USLAB % 1024 == 0, "weight slab must split evenly");
    __shared__ __attribute__((aligned(16))) float s_raw[RAW_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_u[USLAB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int ngroups = COUT / NB;
    // Block order: the launch grid is (D * ngroups, tiles along W, tiles along H * B) so that, after the XCD remap (each
    // XCD takes one contiguous slab of block ids), the blocks that share input — the D output planes over the same (x, y)
    // tile read each other's depth planes, the channel groups read the same planes — run back to back on ONE XCD and find
    // those planes in its L2 instead of fetching them once per XCD.
    unsigned bxi, byi, bzi;
    xcd_block_coords(bxi, byi, bzi);
    const int hblocks = (H + 2 * WTY - 1) / (2 * WTY);
    const int ng = bxi % ngroups, z = bxi / ngroups;
    const int b = bzi / hblocks;
    const int x0 = byi * 32, y0 = (bzi % hblocks) * (2 * WTY);
    const size_t plane = (size_t)H * W, vol = plane * D;

    // ---- chunk-invariant staging maps -------------------------------------------------------------------------
    // byte offsets relative to the first channel of the chunk; OOB => buffer load returns 0 (zero padding and halo)
    unsigned qoff[QPT];
    int qlds[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const int q = tid + j * 256;
        const int ch = q / (3 * RROWS * 8), rem = q % (3 * RROWS * 8);
        const int pl = rem / (RROWS * 8), row = (rem / 8) % RROWS, qx = rem % 8;
        const int gz = z - 1 + pl, gy = y0 - 1 + row, gx = x0 + 4 * qx;
        const bool ok = q < NQUAD && gz >= 0 && gz < D && gy >= 0 && gy < H && gx < W;
        qoff[j] = ok ? (unsigned)((((size_t)ch * D + gz) * plane + (size_t)gy * W + gx) * 4) : OOB;
        // quad = columns c = 4qx+1 .. 4qx+4  ->  O[2qx], E[2qx+1], O[2qx+1], E[2qx+2]
        qlds[j] = q < NQUAD ? ch * RCH + pl * RPLANE + row * RCOLS + 2 * qx : -1;
    }
    unsigned soff1;
    int slds;
    {
        const int s = tid;
        const int ch = s / (3 * RROWS * 2), rem = s % (3 * RROWS * 2);
        const int pl = rem / (RROWS * 2), row = (rem / 2) % RROWS, side = rem % 2;
        const int gz = z - 1 + pl, gy = y0 - 1 + row, gx = side ? x0 + 32 : x0 - 1;
        const bool ok = s < NSINGLE && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff1 = ok ? (unsigned)((((size_t)ch * D + gz) * plane + (size_t)gy * W + gx) * 4) : OOB;
        slds = s < NSINGLE ? ch * RCH + pl * RPLANE + row * RCOLS + (side ? ROFF_O + 16 : 0) : -1;      // c = 33 -> O[16], c = 0 -> E[0]
    }
    const float* xb = x + (size_t)b * CIN * vol;
    const unsigned chunk_bytes = (unsigned)(4 * vol * 4);

    f32x4 pq[QPT];
    float ps;
    f32x4 pu[UQ];
    auto fetch = [&](int c4) {
        const rsrc_t r = make_rsrc(xb + (size_t)c4 * 4 * vol, chunk_bytes);
#pragma unroll
        for (int j = 0; j < QPT; ++j) pq[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, qoff[j], 0, 0));
        ps = buf_load(r, soff1, 0);
        // weight slab rows of the chunk: [kd*16+xi][k] -> COUT floats each, this block's NB of them
        const float* ub = U + (size_t)c4 * 192 * COUT + ng * NB;
#pragma unroll
        for (int j = 0; j < UQ; ++j) {
            const int e = (tid + j * 256) * 4;               // float index inside the [192][NB] slab
            pu[j] = *reinterpret_cast<const f32x4*>(ub + (size_t)(e / NB) * COUT + (e % NB));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < QPT; ++j)
            if (qlds[j] >= 0) {
                float* e = s_raw + qlds[j];
                e[ROFF_O] = pq[j][0];
                e[1] = pq[j][1];
                e[ROFF_O + 1] = pq[j][2];
                e[2] = pq[j][3];
            }
        if (slds >= 0) s_raw[slds] = ps;
#pragma unroll
        for (int j = 0; j < UQ; ++j) *reinterpret_cast<f32x4*>(s_u + (tid + j * 256) * 4) = pu[j];
    };

    f32x4 Z[16][NT];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Z[xi][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int nchunks = CIN / 4;
    fetch(0);
    commit();
    __syncthreads();
    const float* patch0 = s_raw + kk * RCH + (2 * wave) * RCOLS + i16;
    const float* u0 = s_u + kk * NB + i16;
    for (int c4 = 0; c4 < nchunks; ++c4) {
        if (c4 + 1 < nchunks) fetch(c4 + 1);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            // a depth tap that falls outside the volume multiplies a plane of zero padding: skip it (block-uniform; the sums are
            // unchanged).  With D = 4 planes (stage 4) that is a sixth of all MFMAs, with D = 1 (the 2-D visibility CNN in training) 2/3.
            if ((unsigned)(z - 1 + kd) >= (unsigned)D) continue;
            const float* p = patch0 + kd * RPLANE;
            // the patch as the even / odd column pairs the LDS rows deliver: E[q] = (d[q][0], d[q][2]), O[q] = (d[q][1], d[q][3])
            f32x2 E[4], O[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                E[q] = f32x2{p[q * RCOLS], p[q * RCOLS + 1]};
                O[q] = f32x2{p[q * RCOLS + ROFF_O], p[q * RCOLS + ROFF_O + 1]};
            }
            // X = B^T d B with packed fp32 adds (row transform: 8 packed ops on the load pairs; column transform: (X1, X2) of each row
            // packed, X0 / X3 scalar): 20 instead of 32 vector instructions per patch - same scheme as vis_net_wino.hip
            float X[16];
            {
                f32x2 T[4], U2[4];
                T[0] = pk_sub(E[0], E[2]);  U2[0] = pk_sub(O[0], O[2]);
                T[1] = E[1] + E[2];         U2[1] = O[1] + O[2];
                T[2] = pk_sub(E[2], E[1]);  U2[2] = pk_sub(O[2], O[1]);
                T[3] = pk_sub(E[1], E[3]);  U2[3] = pk_sub(O[1], O[3]);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x2 m = pk_fma(f32x2{U2[a].x, U2[a].x}, f32x2{1.0f, -1.0f}, f32x2{T[a].y, T[a].y});
                    X[a * 4 + 0] = T[a].x - T[a].y;
                    X[a * 4 + 1] = m.x;
                    X[a * 4 + 2] = m.y;
                    X[a * 4 + 3] = U2[a].x - U2[a].y;
                }
            }
            const float* u = u0 + kd * 64 * NB;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Z[xi][nt] = mfma4(u[xi * 4 * NB + nt * 16], X[xi], Z[xi][nt]);
        }
        __syncthreads();
        if (c4 + 1 < nchunks) {
            commit();
            __syncthreads();
        }
    }

    // ---- output transform + epilogue: lane holds channels 16*nt + 4*kk + r (r = 0..3) of tile i16 -------------------
    const int oy = y0 + 2 * wave, ox = x0 + 2 * i16;
    if (ox >= W) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s[2][4];
#pragma unroll
            for (int bcol = 0; bcol < 4; ++bcol) {
                s[0][bcol] = Z[0 + bcol][nt][r] + Z[4 + bcol][nt][r] + Z[8 + bcol][nt][r];
                s[1][bcol] = Z[4 + bcol][nt][r] - Z[8 + bcol][nt][r] - Z[12 + bcol][nt][r];
            }
            const int co = ng * NB + nt * 16 + kk * 4 + r;
            const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (oy + a >= H) continue;
                float o0 = s[a][0] + s[a][1] + s[a][2];
                float o1 = s[a][1] - s[a][2] - s[a][3];
                o0 = fmaf(o0, sc, sh);
                o1 = fmaf(o1, sc, sh);
                if (relu) {
                    o0 = fmaxf(o0, 0.0f);
                    o1 = fmaxf(o1, 0.0f);
                }
                const size_t off = (((size_t)b * COUT + co) * D + z) * plane + (size_t)(oy + a) * W + ox;
                if (res) {
                    const float2 rv = *reinterpret_cast<const float2*>(res + off);
                    o0 += rv.x;
                    o1 += rv.y;
                }
                *reinterpret_cast<float2*>(y + off) = make_float2(o0, o1);
            }
        }
}

template <int NT>
int launch_wino(const float* x, const float* U, const float* scale, const float* shift, const float* res, float* y, int B, int Cin,
                int Cout, int D, int H, int W, int relu, hipStream_t s) {
    const int ngroups = Cout / (16 * NT);
    dim3 grid(D * ngroups, mvs::ceil_div(W, 32), mvs::ceil_div(H, 2 * WTY) * B);
    hipLaunchKernelGGL(wino_conv3d_kernel<NT>, grid, dim3(256), 0, s, x, U, scale, shift, res, y, Cin, Cout, D, H, W, relu);
    return mvs::finish_launch("mvs_conv3d_wino_fwd");
}

}  // namespace

extern "C" int mvs_conv3d_wino_supported(int Cin, int Cout, int D, int H, int W) {
    return Cin >= 4 && Cin % 4 == 0 && Cout >= 16 && Cout % 16 == 0 && Cout <= 64 && D >= 1 && H >= 1 && W >= 4 && W % 4 == 0 &&
           (int64_t)4 * D * H * W * 4 < ((int64_t)1 << 31);
}

extern "C" int64_t mvs_conv3d_wino_packed_floats(int Cin, int Cout) {
    if (Cin < 4 || Cin % 4 || Cout < 16 || Cout % 16 || Cout > 64) return 0;
    return (int64_t)(Cin / 4) * 192 * Cout;
}

extern "C" int mvs_conv3d_wino_pack_weights(const float* w, int Cin, int Cout, float* wpacked, mvs_stream_t stream) {
    MVS_REQUIRE(w && wpacked, "mvs_conv3d_wino_pack_weights: null pointer");
    MVS_REQUIRE(mvs_conv3d_wino_packed_floats(Cin, Cout) > 0, "mvs_conv3d_wino_pack_weights: Cin=%d (mult. of 4) Cout=%d (16/32/48/64)", Cin,
                Cout);
    const int64_t total = mvs_conv3d_wino_packed_floats(Cin, Cout);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), w, Cin, Cout, wpacked);
    return mvs::finish_launch("mvs_conv3d_wino_pack_weights");
}

extern "C" int mvs_conv3d_wino_fwd(const float* x, const float* wpacked, const float* scale, const float* shift, const float* residual,
                                   float* y, int B, int Cin, int Cout, int D, int H, int W, int relu, mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && y, "mvs_conv3d_wino_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && mvs_conv3d_wino_supported(Cin, Cout, D, H, W),
                "mvs_conv3d_wino_fwd: unsupported shape B=%d Cin=%d Cout=%d D=%d H=%d W=%d (Cin%%4, Cout%%16, Cout<=64, W%%4)", B, Cin, Cout, D,
                H, W);
    hipStream_t s = MVS_STREAM(stream);
    // all output channels in one block up to 32 (128 accumulator registers); 48/64 channels split over blocks
    const char* env = getenv("MVS_WINO_NT");
    const int nt = (env && atoi(env) == 2 && Cout % 32 == 0) ? 2 : 1;
    MVS_REQUIRE((int64_t)B * mvs::ceil_div(H, 8) <= 65535 && mvs::ceil_div(W, 32) <= 65535, "mvs_conv3d_wino_fwd: grid limit");
    if (nt == 2) return launch_wino<2>(x, wpacked, scale, shift, residual, y, B, Cin, Cout, D, H, W, relu, s);
    return launch_wino<1>(x, wpacked, scale, shift, residual, y, B, Cin, Cout, D, H, W, relu, s);
}
