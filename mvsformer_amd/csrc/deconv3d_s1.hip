// Transposed 3-D convolution of CostRegNet3D's decoder: ConvTranspose3d(k=3, stride (1,2,2), padding 1,
// output_padding (0,1,1), bias=False) -> BatchNorm3d -> ReLU (+ skip), reference models/module.py:562-575,587-591,
// on the fp32 matrix cores, for Cout in {8, 16, 32}.  These three layers are 37 of the 147 GFLOP of a stage-4
// regularizer, and conv11 (16 -> 8) was the slowest kernel of the whole path in the gather-by-parity form
// (30 TFLOP/s: a 16-wide MFMA N tile half empty, one weight fragment per MFMA).
//
// Formulation.  out[co, do, 2hi+hh, 2wi+pw] = sum_{ci,kd} sum over taps (kh,kw) with
//     even output index: (k=1, input offset 0)      odd: (k=2, offset 0), (k=0, offset +1)
// Group the 9 (kh,kw) taps by the INPUT offset (oy,ox) they read instead of by the output class (hh,pw) they feed:
//     (0,0): classes 00,01,10,11   (0,1): classes 01,11   (1,0): classes 10,11   (1,1): class 11
// For one (ci, kd, offset) the A operand (16 consecutive input columns) is shared by all classes of the group, so the
// classes are concatenated along the GEMM N axis: 9*Cout weight columns per (ci, kd), laid out
//     [ c00 | c01 | c10 | c11 ][ c01 | c11 ][ c10 | c11 ][ c11 ]       (each block Cout wide)
// and cut into 16-column MFMA tiles.  Cout = 8 fills 72 of 80 columns (90 % useful) instead of 50 %; every A fragment
// feeds 4/2/2/1 x Cout/16 MFMAs and every B fragment 4 M tiles.  For Cout >= 16 a tile is exactly one class, so tiles
// accumulate straight into per-class accumulators; for Cout = 8 the five tiles are kept separately and the classes are
// recombined in the epilogue with one cross-lane exchange (lanes j and j^8 hold the same output channel).
//
// wavefront = 1 output depth x 1 input row (2 output rows) x 64 input columns (128 output columns)
// block     = 2 depths x 2 input rows;  LDS: input tile [CC][4][3][65] + weights [CC/4][3 kd][4][NPD]
#include "conv_common.h"

namespace {
using namespace mvsconv;

// sum over the 8 lanes of an aligned group (every lane gets the total): quad_perm xor 1, xor 2, then row_half_mirror
// (i <-> 7-i), which pairs the two quads once each quad holds its own sum
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, o);
}
__device__ __forceinline__ float group8_sum(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    return dpp_add<0x141>(v);
}

#ifndef MVS_DS1_EXP
#define MVS_DS1_EXP 0          // timing experiments (tools/exp_tail.py --build): 1 = no MFMA loop, 2 = no input staging loads, 4 = one chunk only, 8 = no epilogue
#endif
constexpr int npd_of(int NC) { return NC == 8 ? 80 : (NC == 16 ? 144 : 304); }       // 9*NC padded to == 16 (mod 32)
constexpr int ntiles_of(int NC) { return NC == 8 ? 5 : 9 * NC / 16; }

// column n of the packed row -> (class hh*2+pw, kh, kw, cout); cout = -1 for padding
__host__ __device__ inline void decode_column(int n, int NC, int* kh, int* kw, int* co) {
    const int blk = n / NC;
    *co = (blk < 9) ? n % NC : -1;
    // blocks: 0..3 = group (0,0) classes 00,01,10,11; 4,5 = group (0,1) classes 01,11; 6,7 = group (1,0) classes 10,11; 8 = (1,1)
    const int khs[9] = {1, 1, 2, 2, 1, 2, 0, 0, 0};
    const int kws[9] = {1, 2, 1, 2, 0, 0, 1, 2, 0};
    *kh = khs[blk < 9 ? blk : 0];
    *kw = kws[blk < 9 ? blk : 0];
}

__global__ void pack_deconv_s1_kernel(const float* __restrict__ w /*[Cin,Cout,3,3,3]*/, int Cin, int Cout, int NPD, int n4,
                                      float* __restrict__ out /*[n4][3][4][NPD]*/) {
    const int64_t total = (int64_t)n4 * 3 * 4 * NPD;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % NPD);
        const int c = (int)((idx / NPD) % 4);
        const int kd = (int)((idx / (NPD * 4)) % 3);
        const int cin = (int)(idx / ((int64_t)NPD * 12)) * 4 + c;
        int kh, kw, co;
        decode_column(n, Cout, &kh, &kw, &co);
        float v = 0.0f;
        if (cin < Cin && co >= 0 && n < 9 * Cout) v = w[((size_t)cin * Cout + co) * 27 + (kd * 3 + kh) * 3 + kw];
        out[idx] = v;
    }
}

// PROB (NC == 8 only): the layer is CostRegNet3D.conv11 and is followed by `prob` = Conv3d(8, 1, 1) (module.py:582,592): the
// epilogue multiplies each finished channel by its 1x1x1 weight, adds the 8 channel lanes with three DPP steps and stores
// ONE logit plane instead of eight feature planes (y = logits [B,1,D,2H,2W]; 8x less written here, 8x less read by the head).
template <int NC, bool PROB>
__global__ __launch_bounds__(256) void deconv3d_s1_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ res, float* __restrict__ y, int CIN, int Di, int Hi,
                                                          int Wi, int relu, const float* __restrict__ prob_w,
                                                          const float* __restrict__ prob_b) {
    static_assert(!PROB || NC == 8, "the fused 1x1x1 head needs all channels of a voxel inside one 8-lane group");
    constexpr int MT = 4;                                    // 16-column M tiles per wavefront (64 input columns)
    constexpr int NPD = npd_of(NC), NTL = ntiles_of(NC);
    constexpr int K16 = (NC >= 16) ? NC / 16 : 1;            // 16-channel tiles per class
    constexpr int NACC = (NC == 8) ? 5 : 4 * K16;            // accumulator tiles per M tile
    constexpr int CC = 8;                                    // input channels per chunk (2 blocks per CU: 75 staging regs, 33 KB LDS)
    constexpr int ID = 4, IH = 3, IW = 65;
    constexpr int RAW = ID * IH * IW;
    constexpr int CS = pad_cs(RAW, 1);
    constexpr int WSLAB = 3 * 4 * NPD;
    constexpr int NEL = CC * RAW, EPT = (NEL + 255) / 256;
    constexpr int NWV = ((CC / 4) * WSLAB / 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + CC * CS;

    const int Do = Di, Ho = Hi * 2, Wo = Wi * 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int ndt = (Di + 1) / 2;
    const int b = blockIdx.z / ndt, d0 = (blockIdx.z % ndt) * 2, hi0 = blockIdx.y * 2, wi0 = blockIdx.x * 64;
    const int dl = wave >> 1, hp = wave & 1;
    const size_t plane = (size_t)Hi * Wi;

    unsigned voff[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * 256;
        const int c = e / RAW, rem = e % RAW;
        const int dz = rem / (IH * IW), hy = (rem / IW) % IH, wx = rem % IW;
        const int gd = d0 - 1 + dz, gh = hi0 + hy, gw = wi0 + wx;
        const bool ok = e < NEL && gd >= 0 && gd < Di && gh < Hi && gw < Wi;
        voff[i] = ok ? (unsigned)((((size_t)c * Di + gd) * Hi + gh) * Wi + gw) * 4u : OOB;
    }
    float sreg[EPT];
    f32x4 wreg[NWV];
    auto prefetch = [&](int ch) {
        const int cleft = min(CC, CIN - ch * CC);
        const rsrc_t xin = make_rsrc(x + (size_t)(b * CIN + ch * CC) * Di * plane, (unsigned)((size_t)cleft * Di * plane * 4));
#pragma unroll
        for (int i = 0; i < EPT; ++i) sreg[i] = (MVS_DS1_EXP & 2) ? (float)voff[i] : buf_load(xin, voff[i], 0);
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
            const int e = tid + i * 256;
            if (e < NEL) s_in[e + (e / RAW) * (CS - RAW)] = sreg[i];
        }
        f32x4* dst = reinterpret_cast<f32x4*>(s_w);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            if (idx < (CC / 4) * WSLAB / 4) dst[idx] = wreg[i];
        }
    };

    f32x4 acc[MT][NACC];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* abase = s_in + kk * CS + hp * IW + i16;     // + ks*4*CS + (dz*IH + oy)*IW + m*16 + ox
    const float* bbase = s_w + kk * NPD + i16;               // + ks*WSLAB + kd*4*NPD + tile*16

    const int nchunks = (MVS_DS1_EXP & 4) ? 1 : (CIN + CC - 1) / CC;
    prefetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
        for (int ks = 0; ks < ((MVS_DS1_EXP & 1) ? 0 : CC / 4); ++ks) {
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const int dz = dl + 2 - kd;                  // input depth do + 1 - kd, tile origin d0 - 1
                float bf[NTL];
#pragma unroll
                for (int t = 0; t < NTL; ++t) bf[t] = bbase[ks * WSLAB + kd * 4 * NPD + t * 16];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float* ap = abase + ks * 4 * CS + dz * IH * IW + m * 16;
                    const float a00 = ap[0], a01 = ap[1], a10 = ap[IW], a11 = ap[IW + 1];
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        // packed tile t -> input offset group and accumulator
                        int grp, ai;
                        if (NC == 8) {
                            grp = (t < 2) ? 0 : t - 1;       // tiles: [c00|c01] [c10|c11] | [c01|c11] | [c10|c11] | [c11|pad]
                            ai = t;
                        } else {
                            const int blk = t / K16, k = t % K16;        // class blocks in packed order
                            const int cls[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};
                            grp = (blk < 4) ? 0 : (blk < 6 ? 1 : (blk < 8 ? 2 : 3));
                            ai = cls[blk] * K16 + k;
                        }
                        const float a = (grp == 0) ? a00 : (grp == 1 ? a01 : (grp == 2 ? a10 : a11));
                        acc[m][ai] = mfma4(a, bf[t], acc[m][ai]);
                    }
                }
            }
        }
    }

    // ---- epilogue ----
    const int od = d0 + dl, hi = hi0 + hp;
    if (od >= Do || hi >= Hi) return;
    if (MVS_DS1_EXP & 8) {                                   // timing experiment: no epilogue (one store keeps the accumulators alive)
        float t = 0.0f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < NACC; ++q) t += acc[m][q][0] + acc[m][q][1] + acc[m][q][2] + acc[m][q][3];
        if (t == 123.456f) y[0] = t;
        return;
    }
    const bool vec_ok = (Wi % 4) == 0;
    auto store_row = [&](int co, int oh, f32x4 e, f32x4 o, int wi) {
        const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
        e = bn_act(e, sc, sh, relu);
        o = bn_act(o, sc, sh, relu);
        const size_t off = (((size_t)(b * NC + co) * Do + od) * Ho + oh) * Wo + (size_t)wi * 2;
        if (vec_ok) {
            f32x4 v0 = {e[0], o[0], e[1], o[1]}, v1 = {e[2], o[2], e[3], o[3]};
            if (res) {
                v0 += *reinterpret_cast<const f32x4*>(res + off);
                v1 += *reinterpret_cast<const f32x4*>(res + off + 4);
            }
            *reinterpret_cast<f32x4*>(y + off) = v0;
            *reinterpret_cast<f32x4*>(y + off + 4) = v1;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (wi + r < Wi) {
                    y[off + 2 * r] = e[r] + (res ? res[off + 2 * r] : 0.0f);
                    y[off + 2 * r + 1] = o[r] + (res ? res[off + 2 * r + 1] : 0.0f);
                }
        }
    };
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int wi = wi0 + m * 16 + kk * 4;
        if (NC == 8) {
            // lanes j<8 hold [c00, c10, c01', c10', c11'''] for channel j; lanes j>=8 hold [c01, c11, c11', c11'', 0] for channel j-8
            f32x4 x0, x1, x3, x4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0[r] = __shfl_xor(acc[m][0][r], 8, 64);
                x1[r] = __shfl_xor(acc[m][1][r], 8, 64);
                x3[r] = __shfl_xor(acc[m][3][r], 8, 64);
                x4[r] = __shfl_xor(acc[m][4][r], 8, 64);
            }
            const bool low = i16 < 8;
            // low lanes write output row 2hi (classes 00 even / 01 odd), high lanes row 2hi+1 (classes 10 / 11)
            const f32x4 e = low ? acc[m][0] : x1 + x3;
            const f32x4 o = low ? x0 + acc[m][2] : ((acc[m][1] + acc[m][2]) + acc[m][3]) + x4;
            if (!PROB) {
                if (wi < Wi) store_row(i16 & 7, hi * 2 + (low ? 0 : 1), e, o, wi);
            } else {
                // v = relu(bn(deconv)) + skip for this lane's channel, then sum_c w_c v_c over the 8 lanes of the group
                const int co = i16 & 7, oh = hi * 2 + (low ? 0 : 1);
                const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f, pw = prob_w[co];
                f32x4 ee = bn_act(e, sc, sh, relu), oo = bn_act(o, sc, sh, relu);
                const bool inb = wi < Wi;                         // Wi % 4 == 0 is required by the launcher
                if (res && inb) {
                    const size_t roff = (((size_t)(b * NC + co) * Do + od) * Ho + oh) * Wo + (size_t)wi * 2;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(res + roff), r1 = *reinterpret_cast<const f32x4*>(res + roff + 4);
                    ee += f32x4{r0[0], r0[2], r1[0], r1[2]};
                    oo += f32x4{r0[1], r0[3], r1[1], r1[3]};
                }
                f32x4 v0 = {ee[0] * pw, oo[0] * pw, ee[1] * pw, oo[1] * pw}, v1 = {ee[2] * pw, oo[2] * pw, ee[3] * pw, oo[3] * pw};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = group8_sum(v0[r]);
                    v1[r] = group8_sum(v1[r]);
                }
                if (co == 0 && inb) {
                    const float pb = prob_b ? prob_b[0] : 0.0f;
                    const size_t off = (((size_t)b * Do + od) * Ho + oh) * Wo + (size_t)wi * 2;
                    *reinterpret_cast<f32x4*>(y + off) = v0 + pb;
                    *reinterpret_cast<f32x4*>(y + off + 4) = v1 + pb;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < K16; ++k) {
                const int co = k * 16 + i16;
                if (wi < Wi) {
                    store_row(co, hi * 2, acc[m][0 * K16 + k], acc[m][1 * K16 + k], wi);
                    store_row(co, hi * 2 + 1, acc[m][2 * K16 + k], acc[m][3 * K16 + k], wi);
                }
            }
        }
    }
}

template <int NC, bool PROB = false>
int launch_s1(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B, int Cin,
              int Di, int Hi, int Wi, int relu, hipStream_t s, const float* prob_w = nullptr, const float* prob_b = nullptr) {
    constexpr int CC = 8;
    constexpr size_t lds = (size_t)(CC * pad_cs(4 * 3 * 65, 1) + (CC / 4) * 3 * 4 * npd_of(NC)) * 4;
    if (lds > 48 * 1024) {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(deconv3d_s1_kernel<NC, PROB>), (int)lds, "mvs_deconv3d_fwd");
        if (rc != MVS_OK) return rc;
    }
    dim3 grid(mvs::ceil_div(Wi, 64), mvs::ceil_div(Hi, 2), B * mvs::ceil_div(Di, 2));
    hipLaunchKernelGGL((deconv3d_s1_kernel<NC, PROB>), grid, dim3(256), lds, s, x, wp, scale, shift, res, y, Cin, Di, Hi, Wi, relu, prob_w,
                       prob_b);
    return mvs::finish_launch("mvs_deconv3d_fwd");
}

}  // namespace

namespace mvsconv {

// measured (tools/bench_conv.py): the grouped form wins only where the plain form wastes half of every N tile
bool deconv_s1_supported(int Cout) { return Cout == 8 || Cout == 16; }

int64_t deconv_s1_packed_floats(int Cin, int Cout) {
    const int n4 = 2 * ((Cin + 7) / 8);                      // whole 8-channel chunks
    return (int64_t)n4 * 3 * 4 * npd_of(Cout);
}

int deconv_s1_pack(const float* w, int Cin, int Cout, float* out, hipStream_t s) {
    const int n4 = 2 * ((Cin + 7) / 8), NPD = npd_of(Cout);
    const int64_t total = (int64_t)n4 * 3 * 4 * NPD;
    hipLaunchKernelGGL(pack_deconv_s1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, Cin, Cout, NPD, n4, out);
    return mvs::finish_launch("mvs_conv3d_pack_weights");
}

int deconv_s1_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B,
                     int Cin, int Cout, int Di, int Hi, int Wi, int relu, hipStream_t s) {
    if (Cout == 8) return launch_s1<8>(x, wp, scale, shift, res, y, B, Cin, Di, Hi, Wi, relu, s);
    if (Cout == 16) return launch_s1<16>(x, wp, scale, shift, res, y, B, Cin, Di, Hi, Wi, relu, s);
    return launch_s1<32>(x, wp, scale, shift, res, y, B, Cin, Di, Hi, Wi, relu, s);
}

int deconv_s1_prob_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                          const float* prob_w, const float* prob_b, float* logits, int B, int Cin, int Di, int Hi, int Wi, int relu,
                          hipStream_t s) {
    return launch_s1<8, true>(x, wp, scale, shift, res, logits, B, Cin, Di, Hi, Wi, relu, s, prob_w, prob_b);
}

}  // namespace mvsconv
