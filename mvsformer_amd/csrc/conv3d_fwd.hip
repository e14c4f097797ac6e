// Forward 3-D convolution (kernel 3, padding 1, stride (SD,SHW,SHW)) + folded BatchNorm + ReLU + residual on the
// fp32 matrix cores: reference Conv3d = conv -> BatchNorm3d -> ReLU, models/module.py:83-123, as used by
// CostRegNet / CostRegNet3D (module.py:474-481,553-560).
//
// Implicit GEMM on v_mfma_f32_16x16x4_f32: M = 16 consecutive output voxels along W, N = 16 output channels,
// K = 4 input channels (x 27 taps).  The design point is ONE 4-wavefront block per CU with a large output tile:
//
//   block tile  = TD x TH output rows x 64 voxels x all Cout;  wavefront = RW = TD*TH/4 rows -> RW*4*NT accumulator
//                 tiles (64..128 registers), every B (weight) fragment is reused by RW*4 MFMAs;
//   per chunk of CC input channels the input tile + halo ([CC][ID][IH][IW]) and the packed weight slab are staged
//   in LDS.  Staging is software-pipelined inside the block: the NEXT chunk is fetched into registers with
//   buffer_load_dword (flat element index -> precomputed byte offset, OOB marker => 0 for padding/halo) right before
//   the current chunk's 27*(CC/4) MFMA steps and written to LDS after them, so HBM/L2 latency hides under
//   27*(CC/4)*RW*4*NT*32 matrix-pipe cycles (7k..28k cycles) and no second block is needed for overlap.
//   A big tile also cuts the halo re-read factor from 4.1x (2x2 rows) to 1.5-2.3x (4x4 rows, D padding free).
//
// LDS channel strides are padded to == 16 (mod 32) words (unit-stride fragments) or odd (stride-2 fragments), packed
// weight rows to == 16 (mod 32), so the two k-halves of every 32-lane ds_read_b32 group hit disjoint banks
// (SQ_LDS_BANK_CONFLICT = 0 in profiles/r01_pmc_reg_stage4_v2.txt).
#include <stdlib.h>
#include <string.h>

#include "conv_common.h"

namespace {
using namespace mvsconv;

// input channels per chunk: 8 if tile + weights fit in ~96 KiB of LDS, else 4
constexpr int tile_cs(int SD, int SHW, int TD, int TH, int MT) {
    return pad_cs(((TD - 1) * SD + 3) * ((TH - 1) * SHW + 3) * ((16 * MT - 1) * SHW + 3), SHW);
}
constexpr int pick_cc(int NP, int SD, int SHW, int TD, int TH, int MT) {
    return (8 * tile_cs(SD, SHW, TD, TH, MT) + 2 * 27 * 4 * NP) * 4 <= 96 * 1024 ? 8 : 4;
}
constexpr size_t lds_bytes(int NP, int SD, int SHW, int TD, int TH, int MT) {
    const int cc = pick_cc(NP, SD, SHW, TD, TH, MT);
    return (size_t)(cc * tile_cs(SD, SHW, TD, TH, MT) + (cc / 4) * 27 * 4 * NP) * 4;
}

// NT = 16-channel N tiles computed by one block (all of Cout, or 1 when the output channels are split over blocks
// for small volumes), NP = packed weight row length (depends on the layer's full Cout), MT = 16-voxel M tiles per row.
template <int NT, int NP, int SD, int SHW, int TD, int TH, int MT>
__global__ __launch_bounds__(256) void conv3d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const float* __restrict__ res, float* __restrict__ y, int CIN, int COUT,
                                                     int Di, int Hi, int Wi, int Do, int Ho, int Wo, int relu) {
    constexpr int RW = TD * TH / NWAVES;                     // output rows per wavefront
    static_assert(TD * TH % NWAVES == 0, "rows must split evenly over the wavefronts");
    constexpr int TW = 16 * MT;                              // output voxels along W per block
    constexpr int CC = pick_cc(NP, SD, SHW, TD, TH, MT);
    constexpr int ID = (TD - 1) * SD + 3, IH = (TH - 1) * SHW + 3, IW = (TW - 1) * SHW + 3;
    constexpr int RAW = ID * IH * IW;
    constexpr int CS = pad_cs(RAW, SHW);
    constexpr int WSLAB = 27 * 4 * NP;                       // one packed cin/4 slab
    constexpr int NEL = CC * RAW;                            // staged input elements per chunk
    constexpr int EPT = (NEL + 255) / 256;                   // ... per thread
    constexpr int NWV = ((CC / 4) * WSLAB / 4 + 255) / 256;  // weight float4s per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                                      // [CC][CS]
    float* s_w = smem + CC * CS;                             // [CC/4][27][4][NP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int ndt = (Do + TD - 1) / TD;
    const int wblocks = (Wo + TW - 1) / TW;
    unsigned bxi, byi, bzi;
    xcd_block_coords(bxi, byi, bzi);
    const int nbase = (bxi / wblocks) * NT * 16;             // first output channel of this block (N split)
    const int b = bzi / ndt, d0 = (bzi % ndt) * TD, h0 = byi * TH, w0 = (bxi % wblocks) * TW;
    const size_t plane = (size_t)Hi * Wi;

    // ---- per-thread staging map (chunk-invariant): element e = tid + 256*i of the [CC][ID][IH][IW] tile ----
    unsigned voff[EPT];                                      // byte offset inside the chunk's channel block, or OOB
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * 256;
        const int c = e / RAW, rem = e % RAW;
        const int dz = rem / (IH * IW), hy = (rem / IW) % IH, wx = rem % IW;
        const int gd = d0 * SD - 1 + dz, gh = h0 * SHW - 1 + hy, gw = w0 * SHW - 1 + wx;
        const bool ok = e < NEL && gd >= 0 && gd < Di && gh >= 0 && gh < Hi && gw >= 0 && gw < Wi;
        voff[i] = ok ? (unsigned)((((size_t)c * Di + gd) * Hi + gh) * Wi + gw) * 4u : OOB;
    }

    float sreg[EPT];
    f32x4 wreg[NWV];
    auto prefetch = [&](int ch) {
        const int cleft = min(CC, CIN - ch * CC);            // channels beyond CIN read as 0 (descriptor range)
        const rsrc_t xin = make_rsrc(x + (size_t)(b * CIN + ch * CC) * Di * plane, (unsigned)((size_t)cleft * Di * plane * 4));
#pragma unroll
        for (int i = 0; i < EPT; ++i) sreg[i] = buf_load(xin, voff[i], 0);
        const f32x4* src = reinterpret_cast<const f32x4*>(wp + (size_t)ch * (CC / 4) * WSLAB);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            wreg[i] = (idx < (CC / 4) * WSLAB / 4) ? src[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;                     // LDS word = e + (e / RAW) * (CS - RAW): recomputed, not kept in
            if (e < NEL) s_in[e + (e / RAW) * (CS - RAW)] = sreg[i];   // registers (a third of the staging registers otherwise)
        }
        f32x4* dst = reinterpret_cast<f32x4*>(s_w);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            if (idx < (CC / 4) * WSLAB / 4) dst[idx] = wreg[i];
        }
    };

    f32x4 acc[RW][MT][NT];
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A-fragment base of each output row this wavefront owns: row rr = wave*RW + j -> (dl, hl)
    const float* abase[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int rr = wave * RW + j, dl = rr / TH, hl = rr % TH;
        abase[j] = s_in + kk * CS + ((dl * SD) * IH + hl * SHW) * IW + i16 * SHW;
    }
    const float* bbase = s_w + kk * NP + nbase + i16;

    const int nchunks = (CIN + CC - 1) / CC;
    prefetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                                     // everyone is done reading the previous chunk
        commit();
        __syncthreads();
        if (ch + 1 < nchunks) prefetch(ch + 1);              // in flight during the MFMA steps below
#pragma unroll
        for (int ks = 0; ks < CC / 4; ++ks) {
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int tap = (kd * 3 + kh) * 3 + kw;
                        float bf[NT];
#pragma unroll
                        for (int n = 0; n < NT; ++n) bf[n] = bbase[ks * WSLAB + tap * 4 * NP + n * 16];
#pragma unroll
                        for (int j = 0; j < RW; ++j) {
                            float a[MT];
#pragma unroll
                            for (int m = 0; m < MT; ++m) a[m] = abase[j][ks * 4 * CS + (kd * IH + kh) * IW + kw + m * 16 * SHW];
#pragma unroll
                            for (int m = 0; m < MT; ++m)
#pragma unroll
                                for (int n = 0; n < NT; ++n) acc[j][m][n] = mfma4(a[m], bf[n], acc[j][m][n]);
                        }
                    }
        }
    }

    // ---- epilogue: y = relu(acc*scale + shift) + residual, 16-byte row segments ----
    const bool vec_ok = (Wo % 4) == 0;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int rr = wave * RW + j;
        const int od = d0 + rr / TH, oh = h0 + rr % TH;
        if (od >= Do || oh >= Ho) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = nbase + n * 16 + i16;
            if (co >= COUT) continue;
            const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
            const size_t rowoff = (((size_t)(b * COUT + co) * Do + od) * Ho + oh) * Wo;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int ow = w0 + m * 16 + kk * 4;
                if (ow >= Wo) continue;
                f32x4 o = bn_act(acc[j][m][n], sc, sh, relu);
                if (vec_ok) {
                    if (res) o += *reinterpret_cast<const f32x4*>(res + rowoff + ow);
                    *reinterpret_cast<f32x4*>(y + rowoff + ow) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ow + r < Wo) y[rowoff + ow + r] = o[r] + (res ? res[rowoff + ow + r] : 0.0f);
                }
            }
        }
    }
}

struct ConvArgs {
    const float *x, *wp, *scale, *shift, *res;
    float* y;
    int B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, relu;
    hipStream_t s;
};

template <int NT, int NP, int SD, int SHW, int TD, int TH, int MT>
int launch(const ConvArgs& a) {
    constexpr size_t lds = lds_bytes(NP, SD, SHW, TD, TH, MT);
    static_assert(lds <= 160 * 1024, "tile does not fit in LDS");
    if (lds > 48 * 1024) {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(conv3d_kernel<NT, NP, SD, SHW, TD, TH, MT>), (int)lds, "mvs_conv3d_fwd");
        if (rc != MVS_OK) return rc;
    }
    const int nsplit = mvs::ceil_div(nt_of(a.Cout), NT);
    dim3 grid(mvs::ceil_div(a.Wo, 16 * MT) * nsplit, mvs::ceil_div(a.Ho, TH), a.B * mvs::ceil_div(a.Do, TD));
    hipLaunchKernelGGL((conv3d_kernel<NT, NP, SD, SHW, TD, TH, MT>), grid, dim3(256), lds, a.s, a.x, a.wp, a.scale, a.shift, a.res,
                       a.y, a.Cin, a.Cout, a.Di, a.Hi, a.Wi, a.Do, a.Ho, a.Wo, a.relu);
    return mvs::finish_launch("mvs_conv3d_fwd");
}

// Tile choice (measured with tools/bench_conv.py on config-2 shapes, profiles/r01_conv_layers_*.txt):
//   * NTF == 1 (Cout 8/16) on large volumes: 4x2-row tiles, one block per CU — the weight fragment is reused by 8 MFMAs
//     and the halo re-read drops from 4.1x to 2.3x (2x faster for the stride-(1,2,2) conv1);
//   * NTF >= 2: 2x2-row tiles at 2 blocks per CU are as fast or faster;
//   * small volumes (deep U-Net levels of the coarse stages, < 256 blocks): 32-voxel tiles and one 16-channel N tile per
//     block, so a 64->64 layer on 4x18x24 voxels becomes 72 short blocks instead of 18 long ones.
template <int NTF, int SD, int SHW>
int dispatch_tile(const ConvArgs& a) {
    constexpr int NP = np_of(NTF);
    auto blocks = [&](int td, int th, int tw) {
        return (long)mvs::ceil_div(a.Wo, tw) * mvs::ceil_div(a.Ho, th) * mvs::ceil_div(a.Do, td) * a.B;
    };
    static const char* force = getenv("MVS_CONV_TILE");     // tuning knob for tools/bench_conv.py: 42 | 22 | 22s
    if (force) {
        if constexpr (SD == 1) { if (!strcmp(force, "42") && a.Do >= 3) return launch<NTF, NP, SD, SHW, 4, 2, 4>(a); }
        if (!strcmp(force, "22")) return launch<NTF, NP, SD, SHW, 2, 2, 4>(a);
        if (!strcmp(force, "22s")) return launch<1, NP, SD, SHW, 2, 2, 2>(a);
    }
    // cost model: the matrix pipe of a CU is the shared resource, so time ~ (blocks per CU, rounded up) x (MFMA cycles
    // per block + per-chunk staging overhead, which is exposed when only one block fits on the CU)
    auto model = [&](int td, int th, int mt, int ntb, size_t lds, int cc) {
        const double chunks = mvs::ceil_div(a.Cin, cc);
        const double mfma = 27.0 * (cc / 4) * (td * th / 4) * mt * ntb * 32.0;
        const double ovh = (lds > 80 * 1024) ? 3500.0 : 1200.0;
        const long nb = blocks(td, th, 16 * mt) * mvs::ceil_div(NTF, ntb);
        return (double)((nb + 255) / 256) * chunks * (mfma + ovh);
    };
    double best = 1e300;
    int pick = 1;
    {   // 1: 2x2 rows, 64 voxels, all channels
        const double c = model(2, 2, 4, NTF, lds_bytes(NP, SD, SHW, 2, 2, 4), pick_cc(NP, SD, SHW, 2, 2, 4));
        if (c < best) { best = c; pick = 1; }
    }
    if constexpr (NTF == 1 && SD == 1) {
        if (a.Do >= 3) {   // 2: 4x2 rows
            const double c = model(4, 2, 4, NTF, lds_bytes(NP, SD, SHW, 4, 2, 4), pick_cc(NP, SD, SHW, 4, 2, 4)) * (SHW == 2 ? 0.6 : 0.95);
            if (c < best) { best = c; pick = 2; }
        }
    }
    if constexpr (NTF > 1) {   // 3: channels split over blocks, 64 voxels
        const double c = model(2, 2, 4, 1, lds_bytes(NP, SD, SHW, 2, 2, 4), pick_cc(NP, SD, SHW, 2, 2, 4)) * 1.1;
        if (c < best) { best = c; pick = 3; }
    }
    {   // 4: channels split, 32 voxels
        const double c = model(2, 2, 2, 1, lds_bytes(NP, SD, SHW, 2, 2, 2), pick_cc(NP, SD, SHW, 2, 2, 2)) * 1.15;
        if (c < best || a.Wo <= 32) { best = c; pick = 4; }
    }
    if (pick == 4) return launch<1, NP, SD, SHW, 2, 2, 2>(a);
    if constexpr (NTF > 1) { if (pick == 3) return launch<1, NP, SD, SHW, 2, 2, 4>(a); }
    if constexpr (NTF == 1 && SD == 1) { if (pick == 2) return launch<NTF, NP, SD, SHW, 4, 2, 4>(a); }
    return launch<NTF, NP, SD, SHW, 2, 2, 4>(a);
}

}  // namespace

extern "C" int mvs_conv3d_fwd(const float* x, const float* wpacked, const float* scale, const float* shift, const float* residual,
                              float* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int shw, int relu,
                              mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && y, "mvs_conv3d_fwd: null pointer");
    if (int rc = check_conv_args("mvs_conv3d_fwd", B, Cin, Cout, Di, Hi, Wi)) return rc;
    MVS_REQUIRE((sd == 1 && shw == 1) || (sd == 2 && shw == 2) || (sd == 1 && shw == 2),
                "mvs_conv3d_fwd: stride (%d,%d,%d) not built", sd, shw, shw);
    ConvArgs a{x, wpacked, scale, shift, residual, y, B, Cin, Cout, Di, Hi, Wi, (Di - 1) / sd + 1, (Hi - 1) / shw + 1,
               (Wi - 1) / shw + 1, relu, MVS_STREAM(stream)};
    MVS_REQUIRE((int64_t)B * mvs::ceil_div(a.Do, 2) <= 65535, "mvs_conv3d_fwd: grid.z limit");
    const int nt = nt_of(Cout);
#define MVS_CONV(NTV)                                              \
    if (sd == 1 && shw == 1) return dispatch_tile<NTV, 1, 1>(a);   \
    if (sd == 2) return dispatch_tile<NTV, 2, 2>(a);               \
    return dispatch_tile<NTV, 1, 2>(a)
    if (nt == 1) { MVS_CONV(1); }
    if (nt == 2) { MVS_CONV(2); }
    MVS_CONV(4);
#undef MVS_CONV
}
