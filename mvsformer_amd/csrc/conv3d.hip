// 3-D regularizer layers (models/module.py:83-165 Conv3d/Deconv3d = conv -> BatchNorm3d -> ReLU, and the
// residual adds of CostRegNet.forward / CostRegNet3D.forward, module.py:495-505,584-594) as implicit GEMMs on
// the fp32 matrix cores of gfx950: v_mfma_f32_16x16x4_f32 (exact fp32 = an fmaf chain, 64 FLOP/clk/SIMD).
//
// GEMM view: M = output voxels (16 consecutive voxels along W per MFMA tile), N = output channels (16 per
// tile), K = 27 taps x Cin (4 input channels per MFMA).  Layout stays NCDHW: for one (tap, cin) the A operand
// of a tile is 16 consecutive floats of an input row, so both the global->LDS staging and the LDS->register
// fragment reads are unit-stride.  Per input-channel chunk a block stages (a) the input tile with its halo and
// (b) the matching slice of the pre-packed weights into LDS; channel strides are padded so the two k-halves
// of a 32-lane LDS access group fall on disjoint banks.  Epilogue (fused): y = relu(acc*scale + shift) +
// residual, written as 16-byte row segments.
//
// Transposed convolutions are computed as gathers over the stride-2 output parities (no zero-stuffing, no
// scatter): an even output index has one contributing tap per strided dimension, an odd one two, and a
// wavefront always owns an even+odd pair in each strided dimension so that all wavefronts do equal work.
//
// Algorithmic FLOPs per layer: 2*27*Cin*Cout per output voxel (conv) / per input voxel (deconv).
#include "conv_common.h"

namespace {
using namespace mvsconv;
using mvs_rsrc_t = rsrc_t;
__device__ __forceinline__ mvs_rsrc_t mvs_make_rsrc(const float* b, unsigned n) { return make_rsrc(b, n); }
__device__ __forceinline__ float mvs_buf_load(mvs_rsrc_t r, unsigned v, unsigned s) { return buf_load(r, v, s); }
constexpr int CC_DECONV = 8;                                                  // input channels per LDS chunk

// --------------------------------------------------------------------------------------------------------
// weight packing: one zero-padded image per layer, [cin/4 slabs][tap 27][c 4][NP].  A kernel that stages CC
// input channels per chunk copies CC/4 consecutive slabs; the slab count is padded to an even number so CC=8
// kernels always read whole slabs.
// --------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, int Cin, int Cout, int transposed, int NP, int n4,
                                    float* __restrict__ out) {
    const int64_t total = (int64_t)n4 * 27 * 4 * NP;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % NP);
        const int c = (int)((idx / NP) % 4);
        const int tap = (int)((idx / (NP * 4)) % 27);
        const int ch = (int)(idx / ((int64_t)NP * 4 * 27));
        const int cin = ch * 4 + c;
        float v = 0.0f;
        // transposed == 2: data gradient of a stride-1 Conv3d = conv with swapped channels and the taps mirrored
        if (cin < Cin && n < Cout)
            v = transposed == 2 ? w[((size_t)cin * Cout + n) * 27 + (26 - tap)]
                                : (transposed ? w[((size_t)cin * Cout + n) * 27 + tap] : w[((size_t)n * Cin + cin) * 27 + tap]);
        out[idx] = v;
    }
}

// --------------------------------------------------------------------------------------------------------
// transposed convolution, stride (SD,2,2), kernel 3, padding 1, output_padding (SD-1,1,1): out = 2x in along
// every strided dim.  Gather form: out[o] = sum over taps k with o = 2*i - 1 + k:
//   o even: (k=1, i=o/2)          o odd: (k=2, i=(o-1)/2), (k=0, i=(o+1)/2)        (stride-1 dim: i = o+1-k)
// SD==1: wavefront = 1 output depth x 2 output rows (even+odd) x 64 output columns  (8 M-tiles)
//        block     = 2 depths x 2 row pairs x 64 columns
// SD==2: wavefront = 2 depths (even+odd) x 2 rows x 32 columns                      (8 M-tiles)
//        block     = 2 depths x 2 row pairs x 64 columns
// --------------------------------------------------------------------------------------------------------
template <int NT, int SD>
__global__ __launch_bounds__(256) void deconv3d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ res, float* __restrict__ y, int CIN, int COUT,
                                                       int Di, int Hi, int Wi, int relu) {
    constexpr int CC = CC_DECONV;
    constexpr int NP = np_of(NT);
    constexpr int ID = (SD == 1) ? 4 : 2;                    // input depths staged per block
    constexpr int IH = 3;                                    // input rows: 2 row pairs + 1
    constexpr int IWI = 32;                                  // input columns per block (excluding the +1 halo)
    constexpr int IW = IWI + 1;
    constexpr int CS = pad_cs(ID * IH * IW, 1);
    constexpr int WSLAB = 27 * 4 * NP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + CC * CS;

    const int Do = Di * SD, Ho = Hi * 2, Wo = Wi * 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    // block origin in INPUT coordinates
    const int ndt = (SD == 1) ? (Di + 1) / 2 : Di;
    const int b = blockIdx.z / ndt;
    const int di0 = (SD == 1) ? (blockIdx.z % ndt) * 2 : (blockIdx.z % ndt);   // SD==1: first output depth; SD==2: input depth
    const int hi0 = blockIdx.y * 2, wi0 = blockIdx.x * IWI;
    // wavefront role
    const int hp = wave & 1;                                 // row pair within the block
    const int wsel = wave >> 1;                              // SD==1: output depth within the block; SD==2: 16-column half
    const int dl = (SD == 1) ? wsel : 0;
    const int wq0 = (SD == 1) ? 0 : wsel * 16;               // first input column (relative) of this wavefront

    // 8 M-tiles of 16 input columns: SD==1: t = (hh, pw, q), q = column half;  SD==2: t = (dd, hh, pw)
    f32x4 acc[8][NT];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = (CIN + CC - 1) / CC;
    const size_t plane = (size_t)Hi * Wi;

    constexpr int RPW = (CC * ID * IH + NWAVES - 1) / NWAVES;
    constexpr int NWV = ((CC / 4) * WSLAB / 4 + 255) / 256;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(IW <= 64, "one load per row per lane");
    float sreg[RPW];
    f32x4 wreg[NWV];
    auto prefetch = [&](int ch) {
        const int cleft = min(CC, CIN - ch * CC);
        const mvs_rsrc_t xin = mvs_make_rsrc(x + (size_t)(b * CIN + ch * CC) * Di * plane, (unsigned)((size_t)cleft * Di * plane * 4));
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave + i * NWAVES;
            const int c = r / (ID * IH), rem = r % (ID * IH), dz = rem / IH, hy = rem % IH;
            const int gd = (SD == 1) ? di0 - 1 + dz : di0 + dz;
            const int gh = hi0 + hy;
            const bool rowok = (r < CC * ID * IH) && gd >= 0 && gd < Di && gh < Hi;
            const unsigned soff = rowok ? (unsigned)((((size_t)c * Di + gd) * Hi + gh) * Wi * 4) : 0u;
            const int gw = wi0 + lane;
            const unsigned voff = (rowok && lane < IW && gw < Wi) ? (unsigned)gw * 4u : OOB;
            sreg[i] = mvs_buf_load(xin, voff, soff);
        }
        const f32x4* src = reinterpret_cast<const f32x4*>(wp + (size_t)ch * (CC / 4) * WSLAB);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            wreg[i] = (idx < (CC / 4) * WSLAB / 4) ? src[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave + i * NWAVES;
            if (r < CC * ID * IH && lane < IW) {
                const int c = r / (ID * IH), rem = r % (ID * IH);
                s_in[c * CS + rem * IW + lane] = sreg[i];
            }
        }
        f32x4* dst = reinterpret_cast<f32x4*>(s_w);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            if (idx < (CC / 4) * WSLAB / 4) dst[idx] = wreg[i];
        }
    };

    prefetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch + 1 < nchunks) prefetch(ch + 1);

        const float* bbase = s_w + kk * NP + i16;
#pragma unroll
        for (int ks = 0; ks < CC / 4; ++ks) {
            const float* abase = s_in + (ks * 4 + kk) * CS + i16;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                // decode tile -> parities / offsets
                const int dd = (SD == 1) ? 0 : (t >> 2);
                const int hh = (SD == 1) ? (t >> 2) : ((t >> 1) & 1);
                const int pw = (SD == 1) ? ((t >> 1) & 1) : (t & 1);
                const int q = (SD == 1) ? (t & 1) : 0;
                const int col = wq0 + q * 16;
                const int ndtap = (SD == 1) ? 3 : (1 + dd);
#pragma unroll
                for (int td = 0; td < ndtap; ++td) {
                    // stride-1 depth: kd = td, dz = dl + 2 - kd.  stride-2 depth: even -> (k=1,off 0); odd -> (k=2,off 0),(k=0,off 1)
                    const int kd = (SD == 1) ? td : (dd == 0 ? 1 : (td == 0 ? 2 : 0));
                    const int dz = (SD == 1) ? (dl + 2 - td) : ((dd == 1 && td == 1) ? 1 : 0);
#pragma unroll
                    for (int th = 0; th < 1 + hh; ++th) {
                        const int kh = hh == 0 ? 1 : (th == 0 ? 2 : 0);
                        const int hy = hp + ((hh == 1 && th == 1) ? 1 : 0);
#pragma unroll
                        for (int tw = 0; tw < 1 + pw; ++tw) {
                            const int kw = pw == 0 ? 1 : (tw == 0 ? 2 : 0);
                            const int wx = col + ((pw == 1 && tw == 1) ? 1 : 0);
                            const int tap = (kd * 3 + kh) * 3 + kw;
                            const float a = abase[(dz * IH + hy) * IW + wx];
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[t][n] = mfma4(a, bbase[ks * WSLAB + tap * 4 * NP + n * 16], acc[t][n]);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: interleave even/odd columns into contiguous 32-byte runs ----
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + i16;
        if (co >= COUT) continue;
        const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {                     // tile pairs (pw = 0 / 1)
            int dd, hh, q;
            if (SD == 1) { dd = 0; hh = tp >> 1; q = tp & 1; } else { dd = tp >> 1; hh = tp & 1; q = 0; }
            const int te = (SD == 1) ? (hh * 4 + 0 * 2 + q) : (dd * 4 + hh * 2 + 0);
            const int to = (SD == 1) ? (hh * 4 + 1 * 2 + q) : (dd * 4 + hh * 2 + 1);
            const int od = (SD == 1) ? di0 + dl : di0 * 2 + dd;
            const int oh = (hi0 + hp) * 2 + hh;
            const int wi = wi0 + wq0 + q * 16 + kk * 4;      // first of this lane's 4 input columns
            if (od >= Do || oh >= Ho || wi >= Wi) continue;
            const f32x4 e = bn_act(acc[te][n], sc, sh, relu);
            const f32x4 o = bn_act(acc[to][n], sc, sh, relu);
            const size_t off = (((size_t)(b * COUT + co) * Do + od) * Ho + oh) * Wo + (size_t)wi * 2;
            if ((Wi % 4) == 0) {
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
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// CostRegNet.prob: Conv3d(C -> 1, k=3, padding=1, bias=False).  N=1 is not matrix-core work: one lane per
// output voxel, 27*C coalesced row reads served by L1, weights through the scalar cache.
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prob3_kernel(const float* __restrict__ x, const float* __restrict__ w, int C, int D, int H,
                                                    int W, float* __restrict__ out) {
    const int xw = blockIdx.x * 64 + threadIdx.x;
    const int yh = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / D, d = blockIdx.z % D;
    if (xw >= W || yh >= H) return;
    const size_t plane = (size_t)H * W;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float* xc = x + (size_t)(b * C + c) * D * plane;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int zd = d + kd - 1;
            if (zd < 0 || zd >= D) continue;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int zh = yh + kh - 1;
                if (zh < 0 || zh >= H) continue;
                const float* row = xc + (size_t)zd * plane + (size_t)zh * W;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int zw = xw + kw - 1;
                    const float v = (zw >= 0 && zw < W) ? row[zw] : 0.0f;
                    acc = fmaf(w[c * 27 + (kd * 3 + kh) * 3 + kw], v, acc);
                }
            }
        }
    }
    out[((size_t)(b * D + d) * H + yh) * W + xw] = acc;
}

// Register-blocked form for W % 4 == 0: a lane owns 4 consecutive voxels along W on PD consecutive depth planes.  Each input row segment
// (1 float4 + its two neighbours) is loaded once and feeds up to min(3, PD) planes x 4 voxels x 3 taps FMAs, 8x fewer vector-memory lane
// requests per output than the kernel above at PD = 4.  All of a channel's loads are issued before its FMAs (the first form waited on every
// row segment: 18 memory latencies per channel with 2 wavefronts per SIMD - 67 us for a 57 MB volume).  PD is chosen by the host so that
// the volume yields several wavefronts per SIMD: one wavefront issues at most one vector instruction per ~4.5 clocks, and the 216 FMAs
// per output are what this kernel consists of.  Same accumulation order per output (c, kd, kh, kw ascending) for every PD and for
// prob3_kernel: bit-identical results.
template <int PD>
__global__ __launch_bounds__(256) void prob3_blocked_kernel(const float* __restrict__ x, const float* __restrict__ w, int C, int D, int H,
                                                            int W, float* __restrict__ out) {
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int yh = blockIdx.y * blockDim.y + threadIdx.y;
    const int ndb = (D + PD - 1) / PD;
    const int b = blockIdx.z / ndb, d0 = (blockIdx.z % ndb) * PD;
    if (x4 >= W || yh >= H) return;
    const size_t plane = (size_t)H * W;
    const mvsconv::rsrc_t r = mvsconv::make_rsrc(x + (size_t)b * C * D * plane, (unsigned)((size_t)C * D * plane * 4));
    float acc[PD][4];
#pragma unroll
    for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    // per-lane byte offsets of the row segments of a channel (depth planes d0-1 .. d0+PD x rows yh-1 .. yh+1), OOB where padded
    unsigned roff[PD + 2][3];
#pragma unroll
    for (int pz = 0; pz < PD + 2; ++pz)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int zd = d0 - 1 + pz, zh = yh + kh - 1;
            const bool ok = zd >= 0 && zd < D && zh >= 0 && zh < H;
            roff[pz][kh] = ok ? (unsigned)((((size_t)zd) * plane + (size_t)zh * W + x4) * 4) : mvsconv::OOB;
        }
    const bool has_l = x4 > 0, has_r = x4 + 4 < W;
    const int lane = threadIdx.x;
    const unsigned cstride = (unsigned)((size_t)D * plane * 4);
    for (int c = 0; c < C; ++c) {
        const float* wc = w + c * 27;
        mvsconv::f32x4 q[PD + 2][3];
        float lft[PD + 2][3], rgt[PD + 2][3];
        const unsigned cbase = (unsigned)c * cstride;
#pragma unroll
        for (int pz = 0; pz < PD + 2; ++pz)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)      // (a padded segment's offset stays beyond the descriptor's range after the channel offset is added)
                q[pz][kh] = __builtin_bit_cast(mvsconv::f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, roff[pz][kh] + cbase, 0, 0));
        // The two halo values of a segment come from the neighbouring lanes' segments; only the wavefront's first / last lane load theirs
        // (a 64-lane dword load costs the address path as much as the 16-byte one: with three loads per segment the kernel was bound by
        // exactly that), in one two-lane load per segment.
        float edge[PD + 2][3];
#pragma unroll
        for (int pz = 0; pz < PD + 2; ++pz)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) edge[pz][kh] = 0.0f;
        if ((lane == 0 && has_l) || (lane == 63 && has_r)) {
            const unsigned eo = cbase + (lane == 0 ? 0u - 4u : 16u);
#pragma unroll
            for (int pz = 0; pz < PD + 2; ++pz)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) edge[pz][kh] = mvsconv::buf_load(r, roff[pz][kh] + eo, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pz = 0; pz < PD + 2; ++pz)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float up = __shfl_up(q[pz][kh][3], 1, 64), dn = __shfl_down(q[pz][kh][0], 1, 64);
                lft[pz][kh] = lane == 0 ? edge[pz][kh] : up;                              // (x4 == 0: zero padding)
                rgt[pz][kh] = !has_r ? 0.0f : (lane == 63 ? edge[pz][kh] : dn);          // (the lane behind the row's last one has exited)
            }
#pragma unroll
        for (int pz = 0; pz < PD + 2; ++pz) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float v[6] = {lft[pz][kh], q[pz][kh][0], q[pz][kh][1], q[pz][kh][2], q[pz][kh][3], rgt[pz][kh]};
#pragma unroll
                for (int od = 0; od < PD; ++od) {
                    const int kd = pz - od;
                    if (kd < 0 || kd > 2) continue;
#pragma unroll
                    for (int ox = 0; ox < 4; ++ox)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc[od][ox] = fmaf(wc[(kd * 3 + kh) * 3 + kw], v[ox + kw], acc[od][ox]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int od = 0; od < PD; ++od)
        if (d0 + od < D) {
            mvsconv::f32x4 o = {acc[od][0], acc[od][1], acc[od][2], acc[od][3]};
            *reinterpret_cast<mvsconv::f32x4*>(out + ((size_t)(b * D + d0 + od) * H + yh) * W + x4) = o;
        }
}

template <int NT, int SD>
size_t deconv_lds_bytes() {
    constexpr int ID = (SD == 1) ? 4 : 2, IW = 32 + 1;
    return (size_t)(CC_DECONV * pad_cs(ID * 3 * IW, 1) + (CC_DECONV / 4) * 27 * 4 * np_of(NT)) * sizeof(float);
}

template <int NT, int SD>
int launch_deconv(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B,
                  int Cin, int Cout, int Di, int Hi, int Wi, int relu, hipStream_t s) {
    const size_t lds = deconv_lds_bytes<NT, SD>();
    if (lds > 48 * 1024) {
        const int rc = mvs::ensure_dynamic_lds(reinterpret_cast<const void*>(deconv3d_kernel<NT, SD>), (int)lds, "mvs_deconv3d_fwd");
        if (rc != MVS_OK) return rc;
    }
    dim3 grid(mvs::ceil_div(Wi, 32), mvs::ceil_div(Hi, 2), B * ((SD == 1) ? mvs::ceil_div(Di, 2) : Di));
    hipLaunchKernelGGL((deconv3d_kernel<NT, SD>), grid, dim3(256), lds, s, x, wp, scale, shift, res, y, Cin, Cout, Di, Hi, Wi, relu);
    return mvs::finish_launch("mvs_deconv3d_fwd");
}

}  // namespace

extern "C" int64_t mvs_conv3d_packed_floats(int Cin, int Cout, int mode) {
    if (Cin < 1 || Cout < 1 || Cout > 64 || mode < 0 || mode > 3) return 0;
    if (mode == 2 && deconv_s1_supported(Cout)) return deconv_s1_packed_floats(Cin, Cout);
    // padded to a multiple of 8 input channels so CC=8 kernels can always stage two full slabs
    const int n4 = 2 * ((Cin + 7) / 8);
    return (int64_t)n4 * 27 * 4 * np_of(nt_of(Cout));
}

extern "C" int mvs_conv3d_pack_weights(const float* w, int Cin, int Cout, int mode, float* wpacked, mvs_stream_t stream) {
    MVS_REQUIRE(w && wpacked, "mvs_conv3d_pack_weights: null pointer");
    MVS_REQUIRE(mode >= 0 && mode <= 3, "mvs_conv3d_pack_weights: mode %d", mode);
    if (int rc = check_conv_args("mvs_conv3d_pack_weights", 1, Cin, Cout, 1, 1, 1)) return rc;
    if (mode == 2 && deconv_s1_supported(Cout)) return deconv_s1_pack(w, Cin, Cout, wpacked, MVS_STREAM(stream));
    const int transposed = mode == 3 ? 2 : (mode != 0);
    const int n4 = 2 * ((Cin + 7) / 8), NP = np_of(nt_of(Cout));
    const int64_t total = (int64_t)n4 * 27 * 4 * NP;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, MVS_STREAM(stream), w, Cin, Cout,
                       transposed, NP, n4, wpacked);
    return mvs::finish_launch("mvs_conv3d_pack_weights");
}

extern "C" int mvs_deconv3d_fwd(const float* x, const float* wpacked, const float* scale, const float* shift, const float* residual,
                                float* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int relu, mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && y, "mvs_deconv3d_fwd: null pointer");
    if (int rc = check_conv_args("mvs_deconv3d_fwd", B, Cin, Cout, Di, Hi, Wi)) return rc;
    MVS_REQUIRE(sd == 1 || sd == 2, "mvs_deconv3d_fwd: depth stride %d not built", sd);
    MVS_REQUIRE((int64_t)B * Di <= 65535, "mvs_deconv3d_fwd: grid.z limit");
    hipStream_t s = MVS_STREAM(stream);
    if (sd == 1 && deconv_s1_supported(Cout))
        return deconv_s1_launch(x, wpacked, scale, shift, residual, y, B, Cin, Cout, Di, Hi, Wi, relu, s);
    const int nt = nt_of(Cout);
#define MVS_DECONV(NTV)                                                                                                     \
    if (sd == 1) return launch_deconv<NTV, 1>(x, wpacked, scale, shift, residual, y, B, Cin, Cout, Di, Hi, Wi, relu, s);   \
    return launch_deconv<NTV, 2>(x, wpacked, scale, shift, residual, y, B, Cin, Cout, Di, Hi, Wi, relu, s)
    if (nt == 1) { MVS_DECONV(1); }
    if (nt == 2) { MVS_DECONV(2); }
    MVS_DECONV(4);
#undef MVS_DECONV
}

extern "C" int mvs_deconv3d_prob1_fwd(const float* x, const float* wpacked, const float* scale, const float* shift, const float* residual,
                                      const float* prob_w, const float* prob_b, float* logits, int B, int Cin, int Di, int Hi, int Wi,
                                      int relu, mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && prob_w && logits, "mvs_deconv3d_prob1_fwd: null pointer");
    if (int rc = check_conv_args("mvs_deconv3d_prob1_fwd", B, Cin, 8, Di, Hi, Wi)) return rc;
    MVS_REQUIRE(Wi % 4 == 0, "mvs_deconv3d_prob1_fwd: input width must be a multiple of 4 (got %d)", Wi);
    MVS_REQUIRE((int64_t)B * Di <= 65535, "mvs_deconv3d_prob1_fwd: grid.z limit");
    return deconv_s1_prob_launch(x, wpacked, scale, shift, residual, prob_w, prob_b, logits, B, Cin, Di, Hi, Wi, relu, MVS_STREAM(stream));
}

extern "C" int mvs_prob3_fwd(const float* x, const float* w, int B, int C, int D, int H, int W, float* logits, mvs_stream_t stream) {
    MVS_REQUIRE(x && w && logits, "mvs_prob3_fwd: null pointer");
    MVS_REQUIRE(B >= 1 && C >= 1 && D >= 1 && H >= 1 && W >= 1 && (int64_t)B * D <= 65535, "mvs_prob3_fwd: bad shape");
    if (W % 4 == 0 && (int64_t)C * D * H * W * 4 < ((int64_t)1 << 31)) {
        // one wavefront per block, and as many depth planes per lane as still leave ~2 wavefronts per SIMD (1024 SIMDs)
        const int64_t waves1 = (int64_t)mvs::ceil_div(W / 4, 64) * H * B * D;
        const int pd = waves1 >= 4 * 2048 ? 4 : (waves1 >= 2 * 2048 ? 2 : 1);
        // four consecutive rows per block: their wavefronts share 2 of every 3 input rows through the CU's L1 (one row per block sent
        // every row to the L2 three times - the kernel ran at the L2's rate: 43 us for a 57 MB volume); MVS_PROB3_ROWS overrides
        const char* e = getenv("MVS_PROB3_ROWS");
        const int rows = e ? atoi(e) : 4;
        MVS_REQUIRE(rows == 1 || rows == 2 || rows == 4, "MVS_PROB3_ROWS must be 1, 2 or 4");
        dim3 grid(mvs::ceil_div(W / 4, 64), mvs::ceil_div(H, rows), B * mvs::ceil_div(D, pd)), block(64, rows);
        MVS_REQUIRE((int64_t)B * mvs::ceil_div(D, pd) <= 65535, "mvs_prob3_fwd: bad shape");
        if (pd == 4) hipLaunchKernelGGL(prob3_blocked_kernel<4>, grid, block, 0, MVS_STREAM(stream), x, w, C, D, H, W, logits);
        else if (pd == 2) hipLaunchKernelGGL(prob3_blocked_kernel<2>, grid, block, 0, MVS_STREAM(stream), x, w, C, D, H, W, logits);
        else hipLaunchKernelGGL(prob3_blocked_kernel<1>, grid, block, 0, MVS_STREAM(stream), x, w, C, D, H, W, logits);
        return mvs::finish_launch("mvs_prob3_fwd");
    }
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), B * D), block(64, 4);
    hipLaunchKernelGGL(prob3_kernel, grid, block, 0, MVS_STREAM(stream), x, w, C, D, H, W, logits);
    return mvs::finish_launch("mvs_prob3_fwd");
}
