// LDS-tiled plane sweep: the fused cost-volume build of StageNet.forward (models/mvsformer_model.py:62-105,
// models/warping.py:69-109) with the source texels of a reference-pixel TILE staged through LDS once per
// (source view, depth-plane chunk) and every bilinear tap served from LDS.
//
// Why: the direct sweeps (cost_volume.hip) issue 4 taps x C channels of 16-byte gathers per (pixel, plane, view) sample and
// are bound by the vector-memory front end (64 B/clk/CU whatever the hit rate).  Neighbouring pixels and
// planes hit the same source texels 4-10 times (tools/footprint_stats.py), so here
//   * a block owns TP = TW x TH reference pixels and S plane slots (256 threads = TP x S; thread = one pixel, DCL planes);
//   * per (pass of S*DCL planes, source view) every thread evaluates its samples' projective geometry in its own registers
//     (one reciprocal + Newton step instead of four IEEE divisions unless MVS_CV_EXACT is requested), the block reduces the
//     bounding box of all taps (DPP row butterflies + 4 LDS atomics per row), and stages that box STRAIGHT FROM THE
//     NCHW FEATURE MAPS - 16-byte loads along x for 4 channels, a register transpose, 16-byte LDS stores - as
//     [channel quad][box row][box column] float4 texels with a zero border for out-of-image texels.  The NCHW->NHWC
//     transpose kernels of the direct path disappear;
//   * a tap is ONE ds_read_b128 per channel quad (LDS: 256 B/clk/CU, 4x the gather rate) at {plane*q, +16 B} immediates from two
//     per-sample row addresses; the 4-tap blend, the group correlation, the similarity norms and the sums over channels are
//     all in the thread's registers: no cross-lane traffic, no per-sample address arithmetic beyond two adds;
//   * channels go through LDS in chunks of CC (whole correlation groups), so the LDS footprint is (CC/4)*CAP*16 bytes whatever
//     C is; a box that does not fit CAP texels (or is wider than 64) takes a direct-from-global gather path for that round -
//     correct for any hypothesis map, fast for the coherent ones a cascade produces.
//
// Sweep A (entropy): grid = tiles x source views; sim[d] per plane -> LDS -> softmax_d entropy per pixel.
// Sweep B (aggregate): grid = tiles x plane-pass groups; loops over ALL views, accumulates sum_v w_v*in_prod_v in registers,
//   writes volume_mean once; eval similarity arg-max (per block, merged across pass groups through a small workspace).
//
// Algorithmic HBM bytes per stage (what bench.py credits): 4*H*W*(V*C + D + G*D), as for the direct sweeps.
#include <limits.h>

#include <type_traits>

#include "common.h"
#include "geometry.h"

namespace {

constexpr int G = 8;
constexpr int NT = 256;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int C_, int TW_, int TH_, int S_, int DCL_, int CC_, int CAP_, int OCC_>
struct TileCfg {
    static constexpr int C = C_, TW = TW_, TH = TH_, S = S_, DCL = DCL_, CC = CC_, CAP = CAP_, OCC = OCC_;   // OCC: blocks per CU the register budget must allow
    static constexpr int TP = TW * TH;                 // pixels per tile
    static constexpr int CPG = C / G;                  // channels per correlation group
    static constexpr int NQ = CC / 4;                  // channel quads (LDS planes) per chunk
    static constexpr int NCH = C / CC;                 // channel chunks
    static constexpr int PPP = S * DCL;                // planes per pass
    static constexpr int PLANE_BYTES = CAP * 16;
    static constexpr int TILE_BYTES = NQ * PLANE_BYTES;
    static constexpr int QPIPE = 2;                    // channel quads whose tap loads may be in flight together
    static_assert(TP * S == NT, "256 threads = pixels x plane slots");
    static_assert(CC % 4 == 0 && C % CC == 0 && (CC % CPG == 0), "a chunk holds whole channel quads and whole groups");
    static_assert(TILE_BYTES <= 65536 - 32, "tap offsets must fit the DS instruction's 16-bit immediate");
};
// stage 1 / 2 (many planes, few pixels): 4x16 pixels x 4 plane slots; stage 3 / 4 (few planes, many pixels): 16x16 pixels, 4 planes each
#ifndef MVS_T8
#define MVS_T8 16, 16, 1, 4, 8, 1536, 3
#endif
#ifndef MVS_T16
#define MVS_T16 16, 16, 1, 4, 16, 768, 3
#endif
#ifndef MVS_T32
#define MVS_T32 16, 4, 4, 1, 16, 768, 3
#endif
#ifndef MVS_T64
#define MVS_T64 16, 4, 4, 1, 16, 768, 3
#endif
using Cfg64 = TileCfg<64, MVS_T64>;
using Cfg32 = TileCfg<32, MVS_T32>;
using Cfg16 = TileCfg<16, MVS_T16>;
using Cfg8 = TileCfg<8, MVS_T8>;

struct Args {
    const float* feat;     // [B,V,C,H,W]
    const float* rt;       // [B,V-1,12]
    const float* depth;    // [B,D,H,W]
    const float* weight;   // [B,V-1,H,W]   sweep B
    float* entropy;        // [B,V-1,H,W]   sweep A
    float* volume;         // [B,G,D,H,W]   sweep B
    float* sim_depth;      // [B,H,W]       sweep B, SIM, nz == 1
    float* sim_part;       // [nz][B][HW][2] sweep B, SIM, nz > 1: (best similarity, its depth)
    unsigned* stats;       // optional: [0] rounds, [1] rounds whose box did not fit (direct path)
    int B, V, D, H, W;
    int ntx, ntiles, nz, total;
    int passes_per_block;  // sweep B
    int allviews;          // sweep A: one block walks all source views of its tile (nz = 1)
};

__device__ __forceinline__ float buf_load1(mvs::rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ f32x4 buf_load4(mvs::rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float buf_load1s(mvs::rsrc_t r, unsigned voff, unsigned soff) {     // soff must be wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4s(mvs::rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// voffset of a masked-off load: with any soffset the address stays beyond every block (< 2 GiB each), so the buffer unit returns
// 0 without touching memory
constexpr unsigned OOB = 0x80000000u;      // beyond any view block (< 2 GiB each): the buffer unit returns 0, no memory access

// integer min / max across the 16 lanes of a DPP row; every lane of the row gets the result
template <bool MAX>
__device__ __forceinline__ int row_reduce(int v) {
    auto op = [](int a, int b) { return MAX ? max(a, b) : min(a, b); };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));    // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));    // row_mirror
    return v;
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>) - every index below is a constant, so every register
// array (accumulators, per-sample geometry) is addressed statically and stays in VGPRs
template <int I>
using Int = std::integral_constant<int, I>;
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(Int<I>{});
        static_for<N, I + 1>(f);
    }
}

// ---------------------------------------------------------------------------------------------------------
template <class T, bool SWEEP_B, bool SIM, bool FAST, bool VEC>
__global__ __launch_bounds__(NT, T::OCC) void cv_tiled_kernel(const Args a) {
    constexpr int C = T::C, TW = T::TW, TP = T::TP, S = T::S, DCL = T::DCL, CC = T::CC, CPG = T::CPG, NQ = T::NQ, NCH = T::NCH;
    constexpr int NJ = CPG;                                         // distinct channel-in-group indices
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* box = reinterpret_cast<int*>(smem + T::TILE_BYTES);        // [2][4]: xmin, ymin, xmax, ymax
    float* sims = reinterpret_cast<float*>(smem + T::TILE_BYTES + 64);   // sweep A: [D][TP]

    // ---- which tile: XCD-aware (block i runs on XCD i % 8; give every XCD one contiguous band of the logical order, so that
    //      neighbouring tiles - which share source texels, hypotheses and reference features - share an L2) ----
    const int per = (a.total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= a.total) return;
    const int z = logical % a.nz;
    const int tile = (logical / a.nz) % a.ntiles;
    const int b = logical / (a.nz * a.ntiles);
    const int H = a.H, W = a.W, D = a.D, V = a.V;
    const int tid = threadIdx.x;
    const int slot = tid / TP, pi = tid % TP;
    const int x = (tile % a.ntx) * TW + pi % TW, y = (tile / a.ntx) * T::TH + pi / TW;
    const bool inimg = x < W && y < H;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)yc * W + xc;
    const float xf = (float)xc, yf = (float)yc;
    // per-pixel streams go through wave-uniform buffer descriptors: one 32-bit offset register per stream instead of a 64-bit
    // address pair per access (this kernel lives on its register budget)
    const unsigned pix4 = (unsigned)(pix * 4), HW4 = (unsigned)(HW * 4);
    const mvs::rsrc_t ref_rs = mvs::make_rsrc(a.feat + (size_t)(b * V) * C * HW, (unsigned)(C * HW * 4));          // [C,H,W]
    const mvs::rsrc_t depth_rs = mvs::make_rsrc(a.depth + (size_t)b * D * HW, (unsigned)(D * HW * 4));            // [D,H,W]

    if (tid < 8) box[tid] = (tid & 2) ? INT_MIN : INT_MAX;
    int cur = 0;

    // reference features of this pixel: one chunk lives in registers (the whole vector when C == CC: loaded once per block)
    float rch[CC];
    auto load_ref = [&](int cbase) {
#pragma unroll
        for (int c = 0; c < CC; ++c) rch[c] = buf_load1s(ref_rs, pix4, (unsigned)(cbase + c) * HW4);
    };
    if constexpr (NCH == 1) load_ref(0);

    // ---- per-pixel constants of sweep B ----
    float denom = 1.0f;
    const mvs::rsrc_t weight_rs = mvs::make_rsrc(SWEEP_B ? a.weight + (size_t)(b * (V - 1)) * HW : a.depth, (unsigned)((V - 1) * HW * 4));
    float inv_ref[NJ];                                              // 1 / max(||ref[:, j]||_2 over groups, 1e-12)   (F.normalize)
    if constexpr (SWEEP_B) {
        float wsum = 0.0f;
        for (int sv = 0; sv < V - 1; ++sv) wsum = wsum + buf_load1s(weight_rs, pix4, (unsigned)sv * HW4);
        denom = wsum + 1e-6f;
        if constexpr (SIM) {
            float n2[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) n2[j] = 0.0f;
#pragma unroll 8
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float r = buf_load1s(ref_rs, pix4, (unsigned)(g * CPG + j) * HW4);
                    n2[j] += r * r;
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) inv_ref[j] = 1.0f / fmaxf(sqrtf(n2[j]), 1e-12f);
        }
    }

    // ---- one round = (pass of S*DCL planes, one source view): geometry -> box -> [stage chunk -> blend chunk]* ----
    // Every channel quad's products are handed to consume(Int<k>, Int<ch>, Int<q>, prod[4], wsq[4]).
    unsigned sxy[DCL];          // (y0 + 2) << 16 | (x0 + 2): tap origin (x0, y0 >= -1, so a valid sample is never 0); 0 = invalid sample
    float swx[DCL], swy[DCL];   // bilinear fractions

    auto round = [&](int d0, int sv, auto&& consume) {
        const float* rt = a.rt + (size_t)(b * (V - 1) + sv) * 12;
        const float* srcp = a.feat + (size_t)(b * V + sv + 1) * C * HW;
        const mvs::rsrc_t src = mvs::make_rsrc(srcp, (unsigned)(C * HW * 4));
        // rot @ (x, y, 1), the same fma chain as geometry.h:sweep_project
        const float rx = fmaf(rt[2], 1.0f, fmaf(rt[1], yf, rt[0] * xf));
        const float ry = fmaf(rt[5], 1.0f, fmaf(rt[4], yf, rt[3] * xf));
        const float rz = fmaf(rt[8], 1.0f, fmaf(rt[7], yf, rt[6] * xf));
        int lxmin = INT_MAX, lymin = INT_MAX, lxmax = INT_MIN, lymax = INT_MIN;
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            const int d = d0 + slot * DCL + k;
            const bool pv = inimg && d < D;
            const float dv = buf_load1(depth_rs, (unsigned)min(d, D - 1) * HW4 + pix4);
            float ix, iy;
            if constexpr (FAST) {
                const float X0 = fmaf(rx, dv, rt[9]), X1 = fmaf(ry, dv, rt[10]), X2 = fmaf(rz, dv, rt[11]);
                const float zz = X2 + 1e-6f;
                float rc = __builtin_amdgcn_rcpf(zz);
                rc = fmaf(rc, fmaf(-zz, rc, 1.0f), rc);               // one Newton step: ~0.5 ulp reciprocal
                ix = X0 * rc;                                          // == ((X0/zz)/half_w - 1 + 1) * half_w up to rounding
                iy = X1 * rc;
            } else {                                                   // the reference's op order, IEEE divisions (warping.py:90-96)
                const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
                const float X0 = rx * dv + rt[9], X1 = ry * dv + rt[10], X2 = rz * dv + rt[11];
                const float zz = X2 + 1e-6f;
                const float un = (X0 / zz) / half_w - 1.0f, vn = (X1 / zz) / half_h - 1.0f;
                ix = (un + 1.0f) * half_w;
                iy = (vn + 1.0f) * half_h;
            }
            // zero padding by construction: coordinates are clamped to one texel outside the image, where the staged box holds
            // zeros; beyond that every tap is zero in the reference too.  fmaxf/fminf drop NaN (-> border -> zero, as the
            // reference's failed comparisons give)
            ix = fminf(fmaxf(ix, -1.0f), (float)W);
            iy = fminf(fmaxf(iy, -1.0f), (float)H);
            const float x0f = floorf(ix), y0f = floorf(iy);
            swx[k] = ix - x0f;
            swy[k] = iy - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;
            sxy[k] = pv ? (unsigned)(((y0 + 2) << 16) | (x0 + 2)) : 0u;
            if (pv) {
                lxmin = min(lxmin, x0);
                lxmax = max(lxmax, x0);
                lymin = min(lymin, y0);
                lymax = max(lymax, y0);
            }
        }
        // ---- bounding box of the block's taps ----
        lxmin = row_reduce<false>(lxmin);
        lymin = row_reduce<false>(lymin);
        lxmax = row_reduce<true>(lxmax);
        lymax = row_reduce<true>(lymax);
        __syncthreads();                                 // [A] previous round's LDS reads done; box[cur] was reset after its last use
        if ((tid & 15) == 0 && lxmin <= lxmax) {
            // one DS atomic per value from the row leaders, written out: hipcc's atomic optimizer otherwise wraps each atomicMin/Max
            // in a scalar loop over the active lanes (~100 instructions per round).  The dynamic LDS block starts at LDS address 0
            // (this kernel has no static __shared__), and the wait is inside the asm because the compiler cannot see these DS ops
            // when it decides whether the barrier below needs one
            const unsigned baddr = (unsigned)(T::TILE_BYTES + cur * 16);
            asm volatile("ds_min_i32 %0, %1\n\tds_min_i32 %0, %2 offset:4\n\tds_max_i32 %0, %3 offset:8\n\tds_max_i32 %0, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                         :: "v"(baddr), "v"(lxmin), "v"(lymin), "v"(lxmax), "v"(lymax) : "memory");
        }
        __syncthreads();                                 // [B]
        const int xmin = box[cur * 4 + 0], ymin = box[cur * 4 + 1], xmax = box[cur * 4 + 2], ymax = box[cur * 4 + 3];
        if (tid < 4) box[(cur ^ 1) * 4 + tid] = (tid & 2) ? INT_MIN : INT_MAX;     // the set the NEXT round reduces into
        cur ^= 1;
        if (xmin > xmax) return;                         // no valid sample in this block for this round (uniform)
        const int bx0 = xmin & ~3;                       // 16-byte aligned columns (xmin >= -1 -> -4)
        const int BW = ((xmax + 1 - bx0 + 1) + 3) & ~3;  // taps reach x0 + 1
        const int BH = ymax + 1 - ymin + 1;
        const int BWP = BW + 2;                          // LDS row pitch: != 0 mod 4 keeps the staging stores to <= 2-way conflicts
        const bool fits = BW <= 64 && BH * BWP <= T::CAP;
        if (a.stats && tid == 0) {
            atomicAdd(&a.stats[0], 1u);
            if (!fits) atomicAdd(&a.stats[1], 1u);
        }
        const unsigned rowb = (unsigned)(BWP * 16);
        // origin of the box in the packed (y + 2, x + 2) coordinates of sxy
        const int oy = ymin + 2, ox = bx0 + 2;
        static_for<NCH>([&](auto chc) {
            constexpr int ch = decltype(chc)::value;
            constexpr int cbase = ch * CC;
            if (fits) {
                if (ch > 0) __syncthreads();             // [D] previous chunk's taps consumed
                // ---- stage the box: thread -> (row ty, float4 column tx4); a quad of lanes reads 64 contiguous bytes, an
                //      8-lane LDS store group covers 2 rows x 4 columns ----
                const int tx4 = (tid & 3) | (((tid >> 3) & 3) << 2);
                const int ty = ((tid >> 2) & 1) | ((tid >> 5) << 1);
                if (tx4 * 4 < BW) {
                    const int gx = bx0 + tx4 * 4;
#pragma unroll 1
                    for (int by = ty; by < BH; by += 16) {
                        const int gy = ymin + by;
                        const bool rowok = (unsigned)gy < (unsigned)H;
                        const unsigned rowoff = ((unsigned)gy * (unsigned)W + (unsigned)gx) * 4u;   // garbage when !rowok: replaced by OOB
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            f32x4 r[4];
                            if constexpr (VEC) {
                                const bool ok = rowok && (unsigned)gx < (unsigned)W;      // W % 4 == 0: all 4 texels in or out
#pragma unroll
                                for (int j = 0; j < 4; ++j) r[j] = buf_load4s(src, ok ? rowoff : OOB, (unsigned)(cbase + q * 4 + j) * HW4);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        const bool ok = rowok && (unsigned)(gx + i) < (unsigned)W;
                                        r[j][i] = buf_load1s(src, ok ? rowoff + 4u * i : OOB, (unsigned)(cbase + q * 4 + j) * HW4);
                                    }
                                }
                            }
                            unsigned char* dst = smem + q * T::PLANE_BYTES + (by * BWP + tx4 * 4) * 16;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                *reinterpret_cast<f32x4*>(dst + i * 16) = f32x4{r[0][i], r[1][i], r[2][i], r[3][i]};
                        }
                    }
                }
            }
            if constexpr (NCH > 1) load_ref(cbase);      // in flight across the barrier
            if (fits) __syncthreads();                   // [C]/[E] box staged
            // ---- blend + correlate: everything below is in this thread's registers.  Two straight-line copies (taps from LDS /
            //      taps gathered from global) selected by ONE uniform branch: with the test inside the unrolled steps the compiler
            //      sank every step's arithmetic behind the last branch and spilled all the taps ----
            auto blend = [&](auto fitsc) {
                constexpr bool FITS = decltype(fitsc)::value;
                static_for<DCL>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const bool pv = sxy[k] != 0u;
                    // tap origin inside the box.  Branch-free on purpose (control flow here lets the compiler sink every sample's
                    // arithmetic behind the last branch): an invalid sample has sxy = 0, the max() parks it on a texel near the
                    // box origin (always staged), its results are never stored
                    const int ry0 = max((int)(sxy[k] >> 16) - oy, 0), rx0 = max((int)(sxy[k] & 0xFFFFu) - ox, 0);
                    const unsigned o0 = (unsigned)((ry0 * BWP + rx0) * 16);
                    const unsigned o1 = o0 + rowb;
                    const float wx = swx[k], wy = swy[k];
                    const float ex = 1.0f - wx, ey = 1.0f - wy;
                    const float w00 = ey * ex, w01 = ey * wx, w10 = wy * ex, w11 = wy * wx;
                    static_for<NQ>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        f32x4 t00, t01, t10, t11;
                        if constexpr (FITS) {
                            const unsigned char* p0 = smem + q * T::PLANE_BYTES + o0;
                            const unsigned char* p1 = smem + q * T::PLANE_BYTES + o1;
                            t00 = *reinterpret_cast<const f32x4*>(p0);
                            t01 = *reinterpret_cast<const f32x4*>(p0 + 16);
                            t10 = *reinterpret_cast<const f32x4*>(p1);
                            t11 = *reinterpret_cast<const f32x4*>(p1 + 16);
                        } else {                         // box too large for LDS: gather this sample's taps from global (dword per channel)
                            const int x0 = (int)(sxy[k] & 0xFFFFu) - 2, y0 = (int)(sxy[k] >> 16) - 2;
                            const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
                            const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const unsigned plane = (unsigned)(cbase + q * 4 + i) * HW4;
                                const unsigned b00 = ((unsigned)y0 * (unsigned)W + (unsigned)x0) * 4u;     // wraps when invalid: unused then
                                t00[i] = buf_load1s(src, (pv && vx0 && vy0) ? b00 : OOB, plane);
                                t01[i] = buf_load1s(src, (pv && vx1 && vy0) ? b00 + 4u : OOB, plane);
                                t10[i] = buf_load1s(src, (pv && vx0 && vy1) ? b00 + (unsigned)W * 4u : OOB, plane);
                                t11[i] = buf_load1s(src, (pv && vx1 && vy1) ? b00 + (unsigned)W * 4u + 4u : OOB, plane);
                            }
                        }
                        float prod[4], wsq[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float g = t00[i] * w00;
                            g = fmaf(t01[i], w01, g);
                            g = fmaf(t10[i], w10, g);
                            g = fmaf(t11[i], w11, g);
                            prod[i] = rch[q * 4 + i] * g;
                            wsq[i] = g * g;
                        }
                        consume(kc, chc, qc, prod, wsq);
                        // bound how many (sample, quad) steps the scheduler may interleave (tap registers in flight)
                        if constexpr ((k * NQ + q) % T::QPIPE == T::QPIPE - 1) __builtin_amdgcn_sched_barrier(0);
                    });
                });
            };
            if (fits) blend(std::true_type{});
            else blend(std::false_type{});
        });
    };

    if constexpr (!SWEEP_B) {
        // =========================================== sweep A: z = source view ===========================================
        for (int sv = a.allviews ? 0 : z; sv < (a.allviews ? V - 1 : z + 1); ++sv) {
        for (int d0 = 0; d0 < D; d0 += T::PPP) {
            float sim[DCL];
#pragma unroll
            for (int k = 0; k < DCL; ++k) sim[k] = 0.0f;
            round(d0, sv, [&](auto kc, auto, auto, const float (&prod)[4], const float (&)[4]) {
                constexpr int k = decltype(kc)::value;
                sim[k] = (((sim[k] + prod[0]) + prod[1]) + prod[2]) + prod[3];
            });
#pragma unroll
            for (int k = 0; k < DCL; ++k) {
                const int d = d0 + slot * DCL + k;
                if (d < D) sims[d * TP + pi] = sim[k] * (1.0f / CPG);          // sum_g mean_j = (sum_c)/CPG
            }
        }
        __syncthreads();
        if (slot == 0 && inimg) {                        // softmax_d entropy (mvsformer_model.py:88-90), one thread per pixel
            float m = -INFINITY;
            for (int d = 0; d < D; ++d) m = fmaxf(m, sims[d * TP + pi]);
            float sum = 0.0f;                            // FAST: hardware exp2 / log2 / reciprocal, as the direct sweep
            for (int d = 0; d < D; ++d) sum += FAST ? __expf(sims[d * TP + pi] - m) : expf(sims[d * TP + pi] - m);
            const float inv_sum = __builtin_amdgcn_rcpf(sum);
            float ent = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float pr = FAST ? __expf(sims[d * TP + pi] - m) * inv_sum : expf(sims[d * TP + pi] - m) / sum;
                ent = ent + (-pr) * (FAST ? __logf(pr + 1e-7f) : logf(pr + 1e-7f));
            }
            a.entropy[(size_t)(b * (V - 1) + sv) * HW + pix] = ent;
        }
        if (a.allviews) __syncthreads();                 // sims is rewritten by the next view
        }
    } else {
        // ================================= sweep B: z = group of plane passes, all views =================================
        float best = -INFINITY, best_depth = 0.0f;
        int best_d = INT_MAX;
        const mvs::rsrc_t vol_rs = mvs::make_rsrc(a.volume + (size_t)b * G * D * HW, (unsigned)(G * D * HW * 4));   // [G,D,H,W]
        const int pass0 = z * a.passes_per_block;
        for (int p = pass0; p < pass0 + a.passes_per_block; ++p) {
            const int d0 = p * T::PPP;
            if (d0 >= D) break;
            float acc[DCL][G];
            float simtot[DCL];
#pragma unroll
            for (int k = 0; k < DCL; ++k) {
                simtot[k] = 0.0f;
#pragma unroll
                for (int g = 0; g < G; ++g) acc[k][g] = 0.0f;
            }
            for (int sv = 0; sv < V - 1; ++sv) {
                const float wv = buf_load1s(weight_rs, pix4, (unsigned)sv * HW4);
                float sq[DCL][NJ], sn[DCL][NJ];
                float carry[DCL];                        // CPG == 8: first half of a group's channel sum
                if constexpr (SIM) {
#pragma unroll
                    for (int k = 0; k < DCL; ++k)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) sq[k][j] = sn[k][j] = 0.0f;
                }
                round(d0, sv, [&](auto kc, auto chc, auto qc, const float (&prod)[4], const float (&wsq)[4]) {
                    constexpr int k = decltype(kc)::value, c0 = decltype(chc)::value * CC + decltype(qc)::value * 4;   // first channel
                    // in_prod[g] = mean_j ref*warp (mvsformer_model.py:77-79); a quad holds 4/CPG groups (or half of one)
                    if constexpr (CPG == 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[k][c0 + i] = acc[k][c0 + i] + prod[i] * wv;
                    } else if constexpr (CPG == 2) {
                        acc[k][c0 / 2] = acc[k][c0 / 2] + ((prod[0] + prod[1]) * 0.5f) * wv;
                        acc[k][c0 / 2 + 1] = acc[k][c0 / 2 + 1] + ((prod[2] + prod[3]) * 0.5f) * wv;
                    } else if constexpr (CPG == 4) {
                        acc[k][c0 / 4] = acc[k][c0 / 4] + ((((prod[0] + prod[1]) + prod[2]) + prod[3]) * 0.25f) * wv;
                    } else {                             // CPG == 8: (quad sum) + (quad sum), as the direct kernel
                        const float h = ((prod[0] + prod[1]) + prod[2]) + prod[3];
                        if constexpr ((c0 & 4) == 0) carry[k] = h;
                        else acc[k][c0 / 8] = acc[k][c0 / 8] + ((carry[k] + h) * 0.125f) * wv;
                    }
                    if constexpr (SIM) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            sq[k][(c0 + i) % CPG] += prod[i];
                            sn[k][(c0 + i) % CPG] += wsq[i];
                        }
                    }
                });
                if constexpr (SIM) {
                    // similarity (mvsformer_model.py:81-85): sum_j <refn[:, j], warp[:, j]> / max(||warp[:, j]||, 1e-12), mean over j
#pragma unroll
                    for (int k = 0; k < DCL; ++k) {
                        float s = 0.0f;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) s += (sq[k][j] * inv_ref[j]) * __builtin_amdgcn_rsqf(fmaxf(sn[k][j], 1e-24f));
                        simtot[k] = simtot[k] + s * (1.0f / CPG);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < DCL; ++k) {
                const int d = d0 + slot * DCL + k;
                if (d < D && inimg) {
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[k][g] / denom), vol_rs,
                                                              (unsigned)d * HW4 + pix4, (unsigned)(g * D) * HW4, 0);
                    if (SIM && (simtot[k] > best || (simtot[k] == best && d < best_d))) {
                        best = simtot[k];
                        best_d = d;
                    }
                }
            }
        }
        if constexpr (SIM) {
            if (best_d != INT_MAX) best_depth = buf_load1(depth_rs, (unsigned)best_d * HW4 + pix4);
            if constexpr (S > 1) {                       // the pixel's planes are spread over S threads: merge through LDS
                __syncthreads();
                float* mv = reinterpret_cast<float*>(smem);
                int* md = reinterpret_cast<int*>(smem) + NT;
                float* mz = reinterpret_cast<float*>(smem) + 2 * NT;
                mv[tid] = best;
                md[tid] = best_d;
                mz[tid] = best_depth;
                __syncthreads();
                if (slot == 0) {
#pragma unroll
                    for (int s = 1; s < S; ++s) {
                        const float v2 = mv[s * TP + pi];
                        const int d2 = md[s * TP + pi];
                        if (v2 > best || (v2 == best && d2 < best_d)) {
                            best = v2;
                            best_d = d2;
                            best_depth = mz[s * TP + pi];
                        }
                    }
                }
            }
            if (slot == 0 && inimg) {
                if (a.nz == 1) {
                    a.sim_depth[(size_t)b * HW + pix] = best_depth;
                } else {
                    float* o = a.sim_part + (((size_t)z * a.B + b) * HW + pix) * 2;
                    o[0] = best;
                    o[1] = best_depth;
                }
            }
        }
    }
}

// merge the per-pass-group similarity maxima: the first (lowest-plane) group wins ties, like torch.argmax
__global__ void sim_merge_kernel(const float* __restrict__ part, int nz, size_t n, float* __restrict__ sim_depth) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = part[i * 2], depth = part[i * 2 + 1];
    for (int zz = 1; zz < nz; ++zz) {
        const float v = part[((size_t)zz * n + i) * 2];
        if (v > best) {
            best = v;
            depth = part[((size_t)zz * n + i) * 2 + 1];
        }
    }
    sim_depth[i] = depth;
}

template <class T>
int passes_per_block(int D) {
    const int npass = (D + T::PPP - 1) / T::PPP;
    return T::S == 1 ? npass : 1;      // pixel-rich stages keep all planes in one block; plane-rich stages get their blocks from the planes
}

int check(const char* who, int B, int V, int C, int Gin, int D, int H, int W) {
    MVS_REQUIRE(B >= 1 && V >= 2 && D >= 1 && H >= 1 && W >= 1, "%s: bad shape B=%d V=%d D=%d H=%d W=%d", who, B, V, D, H, W);
    MVS_REQUIRE(Gin == G, "%s: only G=8 correlation groups are built (got %d)", who, Gin);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "%s: C must be 8, 16, 32 or 64 (got %d)", who, C);
    MVS_REQUIRE((int64_t)C * H * W * 4 < ((int64_t)1 << 31), "%s: one view's feature block exceeds the 2 GiB buffer range", who);
    MVS_REQUIRE((int64_t)G * D * H * W * 4 < ((int64_t)1 << 31) && (int64_t)(V - 1) * H * W * 4 < ((int64_t)1 << 31),
                "%s: one sample's volume / weight block exceeds the 2 GiB buffer range", who);
    MVS_REQUIRE(H <= 65532 && W <= 65532, "%s: image too large", who);
    return MVS_OK;
}

template <class T, bool SWEEP_B>
int launch(const char* who, Args a, bool sim, int flags, hipStream_t s) {
    const int ntx = (a.W + T::TW - 1) / T::TW, nty = (a.H + T::TH - 1) / T::TH;
    a.ntx = ntx;
    a.ntiles = ntx * nty;
    if (SWEEP_B) {
        a.passes_per_block = passes_per_block<T>(a.D);
        const int npass = (a.D + T::PPP - 1) / T::PPP;
        a.nz = (npass + a.passes_per_block - 1) / a.passes_per_block;
    } else {
        a.allviews = (flags >> 2) & 1;
        a.nz = a.allviews ? 1 : a.V - 1;
    }
    const int64_t total = (int64_t)a.B * a.ntiles * a.nz;
    MVS_REQUIRE(total < ((int64_t)1 << 30), "%s: too many blocks", who);
    a.total = (int)total;
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    size_t lds = (size_t)T::TILE_BYTES + 64 + (SWEEP_B ? 0 : (size_t)a.D * T::TP * sizeof(float));
    if (SWEEP_B && T::S > 1 && lds < 3 * NT * sizeof(float)) lds = 3 * NT * sizeof(float);
    MVS_REQUIRE(lds <= 160 * 1024, "%s: D=%d needs %zu bytes of LDS (> 160 KiB)", who, a.D, lds);
    const bool fast = !(flags & 1);
    const bool vec = (a.W % 4 == 0) && (((uintptr_t)a.feat & 15) == 0);
#define MVS_TILED_GO(SIMV, FASTV, VECV)                                                                                \
    do {                                                                                                               \
        auto kern = cv_tiled_kernel<T, SWEEP_B, SIMV, FASTV, VECV>;                                                    \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);                                                     \
    } while (0)
    if constexpr (SWEEP_B) {
        if (sim) {
            if (fast) { if (vec) MVS_TILED_GO(true, true, true); else MVS_TILED_GO(true, true, false); }
            else { if (vec) MVS_TILED_GO(true, false, true); else MVS_TILED_GO(true, false, false); }
        } else {
            if (fast) { if (vec) MVS_TILED_GO(false, true, true); else MVS_TILED_GO(false, true, false); }
            else { if (vec) MVS_TILED_GO(false, false, true); else MVS_TILED_GO(false, false, false); }
        }
    } else {
        if (fast) { if (vec) MVS_TILED_GO(false, true, true); else MVS_TILED_GO(false, true, false); }
        else { if (vec) MVS_TILED_GO(false, false, true); else MVS_TILED_GO(false, false, false); }
    }
#undef MVS_TILED_GO
    if (int rc = mvs::finish_launch(who)) return rc;
    if (SWEEP_B && sim && a.nz > 1) {
        const size_t n = (size_t)a.B * a.H * a.W;
        hipLaunchKernelGGL(sim_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.sim_part, a.nz, n, a.sim_depth);
        return mvs::finish_launch(who);
    }
    return MVS_OK;
}

template <class T>
int64_t workspace_bytes(int B, int D, int H, int W) {
    const int npass = (D + T::PPP - 1) / T::PPP;
    const int ppb = passes_per_block<T>(D);
    const int nz = (npass + ppb - 1) / ppb;
    return nz > 1 ? (int64_t)nz * B * H * W * 2 * (int64_t)sizeof(float) : 0;
}

}  // namespace

extern "C" int64_t mvs_cv_tiled_workspace_bytes(int B, int V, int C, int D, int H, int W) {
    (void)V;
    if (B < 1 || D < 1 || H < 1 || W < 1) return -1;
    switch (C) {
        case 8: return workspace_bytes<Cfg8>(B, D, H, W);
        case 16: return workspace_bytes<Cfg16>(B, D, H, W);
        case 32: return workspace_bytes<Cfg32>(B, D, H, W);
        case 64: return workspace_bytes<Cfg64>(B, D, H, W);
        default: return -1;
    }
}

extern "C" int mvs_cv_tiled_entropy_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int Gin, int D,
                                        int H, int W, float* entropy, int flags, uint32_t* stats, mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && entropy, "mvs_cv_tiled_entropy_fwd: null pointer");
    if (int rc = check("mvs_cv_tiled_entropy_fwd", B, V, C, Gin, D, H, W)) return rc;
    Args a{};
    a.feat = feat, a.rt = rt, a.depth = depth, a.entropy = entropy, a.stats = stats;
    a.B = B, a.V = V, a.D = D, a.H = H, a.W = W;
    hipStream_t s = MVS_STREAM(stream);
    switch (C) {
        case 8: return launch<Cfg8, false>("mvs_cv_tiled_entropy_fwd", a, false, flags, s);
        case 16: return launch<Cfg16, false>("mvs_cv_tiled_entropy_fwd", a, false, flags, s);
        case 32: return launch<Cfg32, false>("mvs_cv_tiled_entropy_fwd", a, false, flags, s);
        default: return launch<Cfg64, false>("mvs_cv_tiled_entropy_fwd", a, false, flags, s);
    }
}

extern "C" int mvs_cv_tiled_aggregate_fwd(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V,
                                          int C, int Gin, int D, int H, int W, float* volume, float* sim_depth, void* workspace,
                                          int flags, uint32_t* stats, mvs_stream_t stream) {
    MVS_REQUIRE(feat && rt && depth && weight && volume, "mvs_cv_tiled_aggregate_fwd: null pointer");
    if (int rc = check("mvs_cv_tiled_aggregate_fwd", B, V, C, Gin, D, H, W)) return rc;
    MVS_REQUIRE(!sim_depth || workspace || mvs_cv_tiled_workspace_bytes(B, V, C, D, H, W) == 0,
                "mvs_cv_tiled_aggregate_fwd: sim_depth at this shape needs the workspace of mvs_cv_tiled_workspace_bytes()");
    Args a{};
    a.feat = feat, a.rt = rt, a.depth = depth, a.weight = weight, a.volume = volume, a.sim_depth = sim_depth;
    a.sim_part = reinterpret_cast<float*>(workspace), a.stats = stats;
    a.B = B, a.V = V, a.D = D, a.H = H, a.W = W;
    hipStream_t s = MVS_STREAM(stream);
    const bool sim = sim_depth != nullptr;
    switch (C) {
        case 8: return launch<Cfg8, true>("mvs_cv_tiled_aggregate_fwd", a, sim, flags, s);
        case 16: return launch<Cfg16, true>("mvs_cv_tiled_aggregate_fwd", a, sim, flags, s);
        case 32: return launch<Cfg32, true>("mvs_cv_tiled_aggregate_fwd", a, sim, flags, s);
        default: return launch<Cfg64, true>("mvs_cv_tiled_aggregate_fwd", a, sim, flags, s);
    }
}
