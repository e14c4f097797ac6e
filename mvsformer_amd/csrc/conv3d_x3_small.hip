// Small-volume form of the regularizer's 3x3x3 convolutions and stride-2 transposed convolutions in three-term split form (split3.h):
// CostRegNet's inner layers at the coarse cascade stages (models/module.py:469-505: conv3 ... conv7 / conv9 at 1/8 and 1/4 resolution,
// volumes of 2 k ... 30 k voxels with 16 ... 64 channels; Conv3d / Deconv3d = conv -> BatchNorm3d -> ReLU [+ skip], module.py:83-165).
//
// Why another kernel.  On those volumes the tiled kernels (conv3d_fwd.hip on the fp32 matrix cores, conv3d_x3.hip's plane sweep) are a
// few dozen blocks that each run a CHAIN of staging rounds - load a chunk, barrier, split, barrier, multiply - 8 ... 16 dependent round
// trips: 25-40 us per layer whatever its size (profiles/r03_stage_breakdown.txt: ten such launches are 0.3 ms of a 0.5 ms stage).  Here
// nothing is staged and nothing is shared: ONE WAVEFRONT owns 16 output voxels of a row x 16 output channels and gathers its own MFMA
// operands straight from global memory (everything is L2 resident at these sizes):
//   * A operand (M = output voxel): lane (m, kb) loads the 8 channels of K block (tap, channel octet) = 4*step + kb at the input voxel its
//     tap reads - 8 dword buffer loads (channel plane in the scalar offset, out-of-volume taps read 0 through the descriptor's range
//     check) - and splits them in registers;
//   * B operand (N = output channel): pre-split, pre-packed weight fragments, three 16-byte loads per step;
//   * six v_mfma_f32_16x16x32_bf16 per step; no LDS, no barrier, hundreds to thousands of independent wavefronts, each with the next
//     step's loads in flight under the current step's MFMAs.
// The split costs ~52 vector operations per step against 96 clocks of MFMA, so the kernel is bound by the vector ALU - which is idle in
// the chained kernels - and finishes a layer in one short round: it is used where the whole layer is a few thousand wavefront-steps.
//
// Transposed convolution (kernel 3, stride 2, padding 1, output_padding 1 in every dimension): output index o = 2i + p gathers
//     p = 0: tap k = 1 at input i          p = 1: tap k = 2 at input i, tap k = 0 at input i + 1
// per dimension, so an output voxel of parity class (pd, ph, pw) has 1 / 2 / 4 / 8 taps - no structural zeros are multiplied.  A wavefront
// owns 16 consecutive INPUT columns of one output (depth, row): both column parities (two accumulators), which leave as 8 consecutive
// output voxels per lane (two 16-byte stores).
#include <stdlib.h>

#include "conv_common.h"
#include "split3.h"

namespace {
using namespace mvsconv;
using mvsx3::bf16x8;
using mvsx3::mfma6;
using mvsx3::Split3;
using mvsx3::split3;

#ifndef SMALL_PF
#define SMALL_PF 4
#endif
constexpr int PF = SMALL_PF;                               // steps of operands in flight per wavefront

struct SmallArgs {
    const float* x;
    const bf16x8* wp;
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, stride, relu;
    int tiles_w, items;             // 16-position tiles along W; wavefront work items = B * ceil(Cout/16) * Do * Ho * tiles_w
};

// K blocks of a transposed-conv class with nt taps: steps of 4 blocks
__host__ __device__ inline int steps_of(int nblocks) { return (nblocks + 3) >> 2; }
// taps per dimension of parity p
__host__ __device__ inline int ntap(int p) { return p ? 2 : 1; }
// tap j of parity p -> kernel index and input offset
__host__ __device__ inline void dtap(int p, int j, int* k, int* off) {
    if (!p) { *k = 1; *off = 0; }
    else if (j == 0) { *k = 2; *off = 0; }
    else { *k = 0; *off = 1; }
}
// first step of class (pd, ph, pw) inside a cout tile's fragment block; classes in order pd, ph, pw (pw fastest)
__host__ __device__ inline int deconv_class_step0(int KQ, int cls) {
    int s = 0;
    for (int c = 0; c < cls; ++c) s += steps_of(ntap(c >> 2) * ntap((c >> 1) & 1) * ntap(c & 1) * KQ);
    return s;
}
__host__ __device__ inline int deconv_total_steps(int KQ) { return deconv_class_step0(KQ, 8); }

// conv:   packed[((ct * STEPS + step) * 3 + term) * 64 + lane][8] = B[n = ct*16 + (lane & 15)][K block q = 4*step + (lane >> 4)], q -> (tap = q / KQ,
//         octet = q % KQ), w [Cout][Cin][27]
// deconv: packed[((ct * TOTAL + class_step0(cls) + step) * 3 + term) * 64 + lane][8], q -> (tap (td, th, tw) of the class, octet), w [Cin][Cout][27]
__global__ void small_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int transposed, bf16x8* __restrict__ out, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int KQ = Cin / 8;
    const int lane = idx & 63, term = (idx >> 6) % 3, gstep = idx / 192;
    const int n = lane & 15, kb = lane >> 4;
    bf16x8 v;
    if (!transposed) {
        const int STEPS = steps_of(27 * KQ), ct = gstep / STEPS, step = gstep % STEPS;
        const int q = 4 * step + kb, co = ct * 16 + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = 0.0f;
            if (q < 27 * KQ && co < Cout) f = w[((size_t)co * Cin + (q % KQ) * 8 + e) * 27 + q / KQ];
            v[e] = mvsx3::split3_term(f, term);
        }
    } else {
        const int TOTAL = deconv_total_steps(KQ), ct = gstep / TOTAL;
        int rem = gstep % TOTAL, cls = 0;
        while (cls < 7 && rem >= deconv_class_step0(KQ, cls + 1) - deconv_class_step0(KQ, cls)) {
            rem -= deconv_class_step0(KQ, cls + 1) - deconv_class_step0(KQ, cls);
            ++cls;
        }
        const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
        const int nh = ntap(ph), nw = ntap(pw), nt = ntap(pd) * nh * nw;
        const int q = 4 * rem + kb, co = ct * 16 + n;
        int kd = 0, kh = 0, kw = 0, o;
        if (q < nt * KQ) {
            const int t = q / KQ;
            dtap(pd, t / (nh * nw), &kd, &o);
            dtap(ph, (t / nw) % nh, &kh, &o);
            dtap(pw, t % nw, &kw, &o);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = 0.0f;
            if (q < nt * KQ && co < Cout) f = w[((size_t)((q % KQ) * 8 + e) * Cout + co) * 27 + (kd * 3 + kh) * 3 + kw];
            v[e] = mvsx3::split3_term(f, term);
        }
    }
    out[idx] = v;
}

// the three weight fragments of one step
struct WFrag { bf16x8 t[3]; };
__device__ __forceinline__ WFrag load_w(const bf16x8* wk) {
    WFrag f;
#pragma unroll
    for (int t = 0; t < 3; ++t) f.t[t] = wk[t * 64];
    return f;
}
// the 8 channels of this lane's K block at byte offset voff (OOB: zeros)
struct AFrag { float f[8]; };
__device__ __forceinline__ AFrag load_a(rsrc_t xin, unsigned voff, unsigned plane_bytes) {
    AFrag a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a.f[e] = buf_load(xin, voff, (unsigned)e * plane_bytes);
    return a;
}
__device__ __forceinline__ f32x4 step_mfma(const AFrag& a, const WFrag& w, f32x4 c) {
    const Split3 s = split3(a.f);
    return mfma6(s.h, s.m, s.l, w.t[0], w.t[1], w.t[2], c);
}

// KS = 1: a block is four independent work items (one per wavefront).  KS = 4: ONE work item per block, its K steps dealt round-robin to the
// four wavefronts and the partial accumulators added through LDS in a fixed order - a single wavefront issues one vector instruction per
// ~4.5 clocks, so a 54-step item (64 input channels) is an 11 us serial chain however many CUs are idle; split four ways it is 3 us.
template <bool TRANSPOSED, int KS>
__global__ __launch_bounds__(256) void x3_small_kernel(const SmallArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = KS == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    const int ks = KS == 1 ? 0 : wave;                     // this wavefront's share of the K steps: ks, ks + KS, ...
    if (item >= a.items) return;                           // wave-uniform for KS = 1, block-uniform for KS = 4 (the only barrier is below)
    __shared__ f32x4 red[KS == 1 ? 1 : 2 * 4 * 64];
    const int m = lane & 15, kb = lane >> 4;
    const int KQ = a.Cin >> 3, kq_shift = 31 - __builtin_clz(KQ);
    const int CT = (a.Cout + 15) >> 4;
    int r = item;
    const int wt = r % a.tiles_w; r /= a.tiles_w;
    const int oh = r % a.Ho; r /= a.Ho;
    const int od = r % a.Do; r /= a.Do;
    const int ct = r % CT, b = r / CT;
    const size_t HWi = (size_t)a.Hi * a.Wi, DHWi = (size_t)a.Di * HWi, HWo = (size_t)a.Ho * a.Wo;
    const rsrc_t xin = make_rsrc(a.x + (size_t)b * a.Cin * DHWi, (unsigned)((size_t)a.Cin * DHWi * 4));
    const unsigned plane_bytes = (unsigned)(DHWi * 4);
    const int co = ct * 16 + m;                            // as the accumulator's column: this lane's output channel
    const float sc = (a.scale && co < a.Cout) ? a.scale[co] : 1.0f, sh = (a.shift && co < a.Cout) ? a.shift[co] : 0.0f;
    auto act = [&](float v) {
        v = a.scale ? fmaf(v, sc, sh) : v + sh;
        return a.relu ? fmaxf(v, 0.0f) : v;
    };

    if constexpr (!TRANSPOSED) {
        const int S = a.stride, nblk = 27 * KQ, nsteps = steps_of(nblk);
        const bf16x8* wk = a.wp + (size_t)ct * nsteps * 192 + lane;
        const int ow = wt * 16 + m;                        // as the A operand's row: this lane's output voxel
        const int zb = od * S - 1, yb = oh * S - 1, xb = ow * S - 1;
        auto a_off = [&](int s) -> unsigned {
            const int q = 4 * s + kb, t = q >> kq_shift, oct = q & (KQ - 1);
            const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
            const int z = zb + kd, y = yb + kh, x = xb + kw;
            const bool ok = q < nblk && ow < a.Wo && (unsigned)z < (unsigned)a.Di && (unsigned)y < (unsigned)a.Hi && (unsigned)x < (unsigned)a.Wi;
            return ok ? ((((unsigned)(oct * 8) * (unsigned)a.Di + (unsigned)z) * (unsigned)a.Hi + (unsigned)y) * (unsigned)a.Wi + (unsigned)x) * 4u : OOB;
        };
        // PF steps of operands in flight: a step's split + MFMAs take ~300 clocks, an L2 round trip 1-2 k - with one step of prefetch the
        // first version ran at one round trip per step (14 us for 27 steps, whatever the layer)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        AFrag ar[PF];
        WFrag wr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (ks + i * KS < nsteps) {
                ar[i] = load_a(xin, a_off(ks + i * KS), plane_bytes);
                wr[i] = load_w(wk + (size_t)(ks + i * KS) * 192);
            }
        for (int s0 = ks; s0 < nsteps; s0 += PF * KS) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int s = s0 + i * KS;
                if (s < nsteps) {
                    acc = step_mfma(ar[i], wr[i], acc);
                    if (s + PF * KS < nsteps) {
                        ar[i] = load_a(xin, a_off(s + PF * KS), plane_bytes);
                        wr[i] = load_w(wk + (size_t)(s + PF * KS) * 192);
                    }
                }
            }
        }
        if constexpr (KS > 1) {
            red[wave * 64 + lane] = acc;
            __syncthreads();
            if (wave != 0) return;
            acc = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
        }
        // D[m = voxel][n = channel]: this lane holds voxels 4kb .. 4kb+3 of channel co
        const int ow0 = wt * 16 + kb * 4;
        if (co < a.Cout && ow0 < a.Wo) {
            const size_t o = ((size_t)(b * a.Cout + co) * a.Do + od) * HWo + (size_t)oh * a.Wo + ow0;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = act(acc[i]);
            if ((a.Wo & 3) == 0) {
                if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + o);
                *reinterpret_cast<f32x4*>(a.y + o) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ow0 + i < a.Wo) a.y[o + i] = a.residual ? v[i] + a.residual[o + i] : v[i];
            }
        }
    } else {
        // output (od, oh) of parities (pd, ph); 16 input columns iw = wt*16 + m, output columns 2 iw + pw
        const int pd = od & 1, ph = oh & 1, id0 = od >> 1, ih0 = oh >> 1;
        const int iw = wt * 16 + m;
        const int nd = ntap(pd), nh = ntap(ph);
        const int TOTAL = deconv_total_steps(KQ);
        f32x4 acc[2];
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
            const int nw = ntap(pw), nblk = nd * nh * nw * KQ, nsteps = steps_of(nblk);
            const int cls = pd * 4 + ph * 2 + pw;
            const bf16x8* wk = a.wp + ((size_t)ct * TOTAL + deconv_class_step0(KQ, cls)) * 192 + lane;
            auto a_off = [&](int s) -> unsigned {
                const int q = 4 * s + kb, t = q >> kq_shift, oct = q & (KQ - 1);
                const int td = t / (nh * nw), th = (t / nw) % nh, tw = t % nw;
                // parity 1: tap 0 reads input i (k = 2), tap 1 reads input i + 1 (k = 0); parity 0: its one tap reads input i
                const int z = id0 + (pd ? td : 0), y = ih0 + (ph ? th : 0), x = iw + (pw ? tw : 0);
                const bool ok = q < nblk && z < a.Di && y < a.Hi && x < a.Wi;
                return ok ? ((((unsigned)(oct * 8) * (unsigned)a.Di + (unsigned)z) * (unsigned)a.Hi + (unsigned)y) * (unsigned)a.Wi + (unsigned)x) * 4u : OOB;
            };
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            AFrag ar[PF];
            WFrag wr[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (ks + i * KS < nsteps) {
                    ar[i] = load_a(xin, a_off(ks + i * KS), plane_bytes);
                    wr[i] = load_w(wk + (size_t)(ks + i * KS) * 192);
                }
            for (int s0 = ks; s0 < nsteps; s0 += PF * KS) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int s = s0 + i * KS;
                    if (s < nsteps) {
                        c = step_mfma(ar[i], wr[i], c);
                        if (s + PF * KS < nsteps) {
                            ar[i] = load_a(xin, a_off(s + PF * KS), plane_bytes);
                            wr[i] = load_w(wk + (size_t)(s + PF * KS) * 192);
                        }
                    }
                }
            }
            acc[pw] = c;
        }
        if constexpr (KS > 1) {
            red[wave * 64 + lane] = acc[0];
            red[256 + wave * 64 + lane] = acc[1];
            __syncthreads();
            if (wave != 0) return;
            acc[0] = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
            acc[1] = ((red[256 + lane] + red[320 + lane]) + red[384 + lane]) + red[448 + lane];
        }
        // this lane: input columns 4kb .. 4kb+3 -> output columns 2(wt*16 + 4kb) .. + 7, even = pw 0, odd = pw 1
        const int ow0 = 2 * (wt * 16 + kb * 4);
        if (co < a.Cout && ow0 < a.Wo) {
            const size_t o = ((size_t)(b * a.Cout + co) * a.Do + od) * HWo + (size_t)oh * a.Wo + ow0;
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = act(acc[0][i]);
                v[2 * i + 1] = act(acc[1][i]);
            }
            // Wo = 2 Wi: with Wi % 2 == 0 both 16-byte halves are inside or outside the row together with their first element
            if ((a.Wi & 1) == 0) {
                const bool second = ow0 + 4 < a.Wo;
                if (a.residual) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(a.residual + o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += r0[i];
                    if (second) {
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(a.residual + o + 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[4 + i] += r1[i];
                    }
                }
                *reinterpret_cast<f32x4*>(a.y + o) = f32x4{v[0], v[1], v[2], v[3]};
                if (second) *reinterpret_cast<f32x4*>(a.y + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (ow0 + i < a.Wo) a.y[o + i] = a.residual ? v[i] + a.residual[o + i] : v[i];
            }
        }
    }
}

bool small_ok(int Cin, int Cout, int stride, int transposed) {
    if (Cin != 8 && Cin != 16 && Cin != 32 && Cin != 64) return false;
    if (Cout < 8 || Cout > 64 || (Cout & 7)) return false;
    return transposed ? stride == 2 : (stride == 1 || stride == 2);
}

}  // namespace

extern "C" int mvs_conv3d_small_supported(int Cin, int Cout, int stride, int transposed) { return small_ok(Cin, Cout, stride, transposed) ? 1 : 0; }

extern "C" int64_t mvs_conv3d_small_packed_bytes(int Cin, int Cout, int stride, int transposed) {
    if (!small_ok(Cin, Cout, stride, transposed)) return 0;
    const int KQ = Cin / 8, CT = (Cout + 15) / 16;
    return (int64_t)CT * (transposed ? deconv_total_steps(KQ) : steps_of(27 * KQ)) * 3 * 64 * 16;
}

extern "C" int mvs_conv3d_small_pack_weights(const float* w, int Cin, int Cout, int stride, int transposed, void* wpacked, mvs_stream_t stream) {
    MVS_REQUIRE(w && wpacked, "mvs_conv3d_small_pack_weights: null pointer");
    MVS_REQUIRE(small_ok(Cin, Cout, stride, transposed), "mvs_conv3d_small_pack_weights: Cin=%d Cout=%d stride=%d transposed=%d is not built", Cin, Cout,
                stride, transposed);
    const int total = (int)(mvs_conv3d_small_packed_bytes(Cin, Cout, stride, transposed) / 16);
    hipLaunchKernelGGL(small_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, MVS_STREAM(stream), w, Cin, Cout, transposed, static_cast<bf16x8*>(wpacked), total);
    return mvs::finish_launch("mvs_conv3d_small_pack_weights");
}

/* transposed = 0: y [B,Cout,Do,Ho,Wo] = [relu](conv3d(x [B,Cin,D,H,W], w, stride (s,s,s), padding 1) * scale + shift) [+ residual], Do = (D-1)/s + 1 ...
 * transposed = 1: y [B,Cout,2D,2H,2W] = [relu](conv_transpose3d(x, w, stride 2, padding 1, output_padding 1) * scale + shift) [+ residual] */
extern "C" int mvs_conv3d_small_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual, float* y,
                                    int B, int Cin, int Cout, int D, int H, int W, int stride, int transposed, int relu, mvs_stream_t stream) {
    MVS_REQUIRE(x && wpacked && y, "mvs_conv3d_small_fwd: null pointer");
    MVS_REQUIRE(small_ok(Cin, Cout, stride, transposed), "mvs_conv3d_small_fwd: Cin=%d Cout=%d stride=%d transposed=%d is not built", Cin, Cout, stride,
                transposed);
    MVS_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W >= 1, "mvs_conv3d_small_fwd: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
    MVS_REQUIRE(!scale || shift, "mvs_conv3d_small_fwd: scale without shift");
    MVS_REQUIRE((int64_t)Cin * D * H * W * 4 < ((int64_t)1 << 31), "mvs_conv3d_small_fwd: one sample's input exceeds the 2 GiB buffer window");
    SmallArgs a;
    a.x = x; a.wp = static_cast<const bf16x8*>(wpacked); a.scale = scale; a.shift = shift; a.residual = residual; a.y = y;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.Di = D; a.Hi = H; a.Wi = W; a.stride = stride; a.relu = relu;
    if (transposed) {
        a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W;
        a.tiles_w = mvs::ceil_div(W, 16);
    } else {
        a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
        a.tiles_w = mvs::ceil_div(a.Wo, 16);
    }
    const int64_t items = (int64_t)B * ((Cout + 15) / 16) * a.Do * a.Ho * a.tiles_w;
    MVS_REQUIRE(items < ((int64_t)1 << 31), "mvs_conv3d_small_fwd: too many work items");
    a.items = (int)items;
    // K split over a block's four wavefronts where an item is a long serial chain (>= 16 steps: 32 / 64 input channels) and the items
    // alone would not fill the chip anyway
    const int KQ = Cin / 8, nsteps = transposed ? steps_of(8 * KQ) : steps_of(27 * KQ);
    static const char* env = getenv("MVS_SMALL_KSPLIT");   // diagnostics: 0 = never, 1 = always
    const bool split = env ? atoi(env) != 0 : (nsteps >= 16 && items <= 16384);
    if (split) {
        const dim3 grid((unsigned)items);
        if (transposed) hipLaunchKernelGGL((x3_small_kernel<true, 4>), grid, dim3(256), 0, MVS_STREAM(stream), a);
        else hipLaunchKernelGGL((x3_small_kernel<false, 4>), grid, dim3(256), 0, MVS_STREAM(stream), a);
    } else {
        const dim3 grid((unsigned)((items + 3) / 4));
        if (transposed) hipLaunchKernelGGL((x3_small_kernel<true, 1>), grid, dim3(256), 0, MVS_STREAM(stream), a);
        else hipLaunchKernelGGL((x3_small_kernel<false, 1>), grid, dim3(256), 0, MVS_STREAM(stream), a);
    }
    return mvs::finish_launch("mvs_conv3d_small_fwd");
}
