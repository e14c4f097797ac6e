// bf16 regularizer for training under autocast (BASELINE configs[2]; reference trainer/mvsformer_trainer.py:43-45,104-106 wraps the
// model in torch.cuda.amp.autocast: Conv3d / ConvTranspose3d of models/module.py:83-165,469-594 then run in half precision with
// fp32 accumulation and half-precision activations, BatchNorm statistics in fp32; the cost volume itself stays fp32,
// models/mvsformer_model.py:65,68,78).
//
// MI355X form: activations are bf16 CHANNEL-LAST [B,D,H,W,C] (NDHWC) in HBM - half the bytes of the fp32 path, and the 8 input
// channels a v_mfma_f32_16x16x32_bf16 k-block needs are 16 contiguous bytes of one voxel, so both MFMA operands are plain 16-byte
// loads (weights: pre-packed per lane; inputs: straight from L1/L2, every voxel is re-read by its 27 taps) - no LDS staging, no
// barriers in the convolution.  D[co, voxel] = sum_k W[co, k] * X[k, voxel]: M = 16 output channels, N = 16 consecutive output
// voxels of a row, K = (tap, cin) flattened in blocks of 8 channels, 4 blocks per MFMA.  One kernel covers Conv3d stride
// (1,1,1)/(2,2,2)/(1,2,2) and, as a gather over output parities, ConvTranspose3d stride (2,2,2)/(1,2,2) - which is also every
// data gradient (stride-1 conv <-> flipped stride-1 conv, strided conv <-> transposed conv).  Weight gradients contract over voxels
// on the same MFMA with fragments gathered from LDS tiles.  BatchNorm / ReLU / skip kernels are the channel-last bf16 twins of
// train.hip (statistics, and all arithmetic, in fp32).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 ld8(rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void xcd_item(int total, int& logical, bool& valid) {
    const int per = (total + 7) >> 3;
    logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    valid = logical < total;
}

// ------------------------------------------------------------------------------------------------ weight packing
// out[((step * NT + nt) * 64 + lane) * 8 + e] = A[m = nt*16 + (lane & 15)][k-block t = 4*step + (lane >> 4)][e]  as bf16, where k-block
// t = (tap, channel octet cq) = (t / (KC/8), t % (KC/8)), channel c = cq*8 + e; zero beyond M rows / 27*KC/8 blocks.
// src: 0 = w[m][c][tap], 1 = w[c][m][tap], 2 = w[c][m][26 - tap]   (w = [d0][d1][27] fp32)
struct PackJob {
    int d0, d1, src, M, KC, NT, steps, nblocks, taps;   // taps: 27 (w = [d0][d1][27]) or 9 (a 2-D kernel [d0][d1][9] = the centre depth tap)
    const float* w;
    __bf16* out;
};
// Rows / channels beyond the weight's own extents pack as zeros, so a narrower parameter is padded on the fly: the 1 -> 16 first layer
// of the visibility CNN as 8 -> 16, CostRegNet's 8 -> 1 `prob` as 8 -> 8.
__device__ __forceinline__ void pack_one(const PackJob& j, int idx) {
    if (idx >= j.steps * j.NT * 64) return;
    const int lane = idx & 63, nt = (idx >> 6) % j.NT, step = idx / (64 * j.NT);
    const int m = nt * 16 + (lane & 15), t = 4 * step + (lane >> 4), KQ = j.KC / 8;
    const int tap = t / KQ, cq = t % KQ;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cq * 8 + e;
        float f = 0.0f;
        if (m < j.M && t < j.taps * KQ) {
            if (j.src == 0) {
                if (m < j.d0 && c < j.d1) f = j.w[((size_t)m * j.d1 + c) * j.taps + tap];
            } else if (c < j.d0 && m < j.d1) {
                f = j.w[((size_t)c * j.d1 + m) * j.taps + (j.src == 2 ? j.taps - 1 - tap : tap)];
            }
        }
        v[e] = (__bf16)f;
    }
    reinterpret_cast<bf16x8*>(j.out)[idx] = v;
}
// one launch packs up to two layouts of the same weight (the forward's and the data gradient's): blocks [0, a.nblocks) do job a
__global__ void bf16_pack_kernel(const PackJob a, const PackJob b) {
    const bool first = (int)blockIdx.x < a.nblocks;
    pack_one(first ? a : b, ((int)blockIdx.x - (first ? 0 : a.nblocks)) * blockDim.x + threadIdx.x);
}
// a whole table of jobs (every layer of a stage, both layouts) in ONE launch: job i owns blocks [start[i], start[i+1])
__global__ void bf16_pack_table_kernel(const PackJob* __restrict__ jobs, const int* __restrict__ start, int njobs) {
    int i = 0;
    while (i + 1 < njobs && (int)blockIdx.x >= start[i + 1]) ++i;                // uniform scalar scan (a few dozen jobs)
    pack_one(jobs[i], ((int)blockIdx.x - start[i]) * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------------------------------------------------ convolution (forward and data gradients)
struct ConvArgs {
    const __bf16* x;          // [B,Di,Hi,Wi,CIN]
    const __bf16* wp;         // packed, see bf16_pack_kernel
    __bf16* y;                // [B,Do,Ho,Wo,Cout]
    const float* scale;       // optional epilogue: y = [relu](acc*scale[co] + shift[co]) [+ residual]
    const float* shift;
    const __bf16* residual;
    int relu;
    int B, Di, Hi, Wi, Do, Ho, Wo, Cout;
    int nwchunks, items;
    float* stats_part;        // optional [items][2*Cout]: per work item, sum and sum of squares of the bf16-ROUNDED outputs per channel
                              // (the batch statistics of the BatchNorm that follows: no separate pass over y)
    int stats_rows;           // > 0: BLOCK rows instead, transposed: stats_part[j * stats_rows + block] for j < 2*Cout (the four waves of
                              // a block combined through LDS in a fixed order; mvs_bf16_conv3d_bn_fwd)
    // bn_y != null (with stats_rows > 0): the rows are NOT the output's statistics but the two sums of a BatchNorm(+ReLU) BACKWARD whose
    // incoming gradient this convolution produces (the data gradient of the NEXT layer = dz of the previous one): g = y_out * [relu'(bn_y *
    // scale + shift)], rows = [sum g | sum g * (bn_y - mean) * invstd] per channel; bn4 = [scale | shift | mean | invstd], each bn_groups*Cout
    const __bf16* bn_y;
    const float* bn4;
    int bn_relu, bn_groups;
};

// GATHER = 0: Conv3d, input voxel = out*stride - 1 + k.  GATHER = 1: ConvTranspose3d (k=3, padding 1, output_padding stride-1) as a
// gather: input voxel = (out + 1 - k) / stride where divisible.
// KSPLIT (32 / 64 input channels, small launches): the four wavefronts of a block share ONE work item and take every fourth K step
// each, their partial tiles are added through LDS in a fixed order and wavefront 0 runs the epilogue.  A wavefront's K loop is a chain
// of STEPS (27 / 54) load -> MFMA steps with ~3 steps of loads in flight: a launch with fewer wavefronts than the chip has SIMD slots
// is bound by the length of that chain (a 64 -> 64 layer took 50-60 us on 320 voxels as on 41 000), and this makes it 4x shorter.
template <int CIN, int NT, int GATHER, int SD, int SHW, int TAPS = 27, bool KSPLIT = false>     // TAPS = 9: a 2-D kernel (depth tap kd = 1 only)
__global__ __launch_bounds__(256) void bf16_conv_kernel(const ConvArgs a) {
    constexpr int KQ = CIN / 8, NKB = TAPS * KQ, STEPS = (NKB + 3) / 4, VT = 4;
    int logical;
    bool ok;
    xcd_item(KSPLIT ? a.items : (a.items + 3) / 4, logical, ok);
    if (!ok) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform FOR THE COMPILER: the buffer descriptor below depends on it (a divergent one is a waterfall loop per load)
    __shared__ float sred[4][2 * 64];                      // block rows of the statistics (stats_rows > 0 only)
    __shared__ f32x4 kred[KSPLIT ? 3 * NT * 4 * 64 : 1];   // KSPLIT: the partial tiles of wavefronts 1..3
    const int item = KSPLIT ? logical : logical * 4 + wave;
    if (item >= a.items) {                                 // no block-wide barrier below, except for block rows of statistics
        if (a.stats_rows > 0) {
            if (lane < 2 * a.Cout) sred[wave][lane] = 0.0f;
            if (lane + 64 < 2 * a.Cout) sred[wave][lane + 64] = 0.0f;
            __syncthreads();
        }
        return;
    }
    // Transposed convolution (GATHER = 1), column stride 2: a work item is the 64 output voxels of ONE column parity of a 128-column chunk
    // of a row (item's lowest bit = the parity), so that which taps exist is the same for all of its voxels - row / depth parities are a
    // row's anyway.  The K loop then walks only the existing taps (1 / 2 per strided axis: 6.75 of 27 on average at stride (1,2,2), 3.4 at
    // (2,2,2)) instead of multiplying structural zeros for the others.
    constexpr int WSTEP = (GATHER == 1 && SHW == 2) ? 2 : 1;                    // output-column step between a lane's neighbours
    const int pw = WSTEP == 2 ? (item & 1) : 0, item_ = WSTEP == 2 ? (item >> 1) : item;
    const int wchunk = item_ % a.nwchunks;
    const int oh = (item_ / a.nwchunks) % a.Ho;
    const int od = (item_ / (a.nwchunks * a.Ho)) % a.Do;
    const int b = item_ / (a.nwchunks * a.Ho * a.Do);
    const int j = lane & 15, kb = lane >> 4;
    const int ow0 = wchunk * 64 * WSTEP + pw;
    // existing taps per axis for this item's parities: stride 1 -> k = 0, 1, 2; stride 2 -> parity 0: k = 1, parity 1: k = 0, 2
    const int nkd = (GATHER == 1 && SD == 2) ? ((od & 1) ? 2 : 1) : 3, nkh = (GATHER == 1 && SHW == 2) ? ((oh & 1) ? 2 : 1) : 3,
              nkw = (GATHER == 1 && SHW == 2) ? (pw ? 2 : 1) : 3;
    const int nsteps = GATHER == 1 ? (nkd * nkh * nkw * KQ + 3) / 4 : STEPS;
    const rsrc_t xr = make_rsrc(a.x + (size_t)b * a.Di * a.Hi * a.Wi * CIN, (unsigned)((size_t)a.Di * a.Hi * a.Wi * CIN * 2));
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wp) + (GATHER == 1 ? j : lane);

    f32x4 acc[NT][VT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) acc[nt][vt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K loop, software-pipelined one step deep: the operands of step s + 1 (VT activation fragments straight from global memory through
    // the buffer descriptor + NT weight fragments) are in flight under the MFMAs of step s.  Two register sets with literal indices, the
    // load order pinned (sched_barrier: the scheduler otherwise sinks the loads below the MFMAs towards their use) and no exit between
    // the loop's halves, so the compiler's waits are counted (the younger step's loads stay in flight) - NOTEBOOK.md, late round 5.
    bf16x8 xs[2][VT], ws[2][NT];
    auto load_step = [&](int set, int step) {
        int t = 4 * step + kb;                              // K block of this lane group: (tap, channel octet)
        int tap = t / KQ;
        const int cq = t % KQ;
        bool rowok = t < NKB && step < nsteps;
        int kd, kh, kw;
        if (GATHER == 1) {
            // compact index over the EXISTING taps -> the tap itself; the weight fragment of K block (tap, cq) sits in the packed image at
            // step (tap*KQ + cq) / 4, lane group (tap*KQ + cq) % 4 - read from there whatever lane group multiplies it
            rowok = rowok && tap < nkd * nkh * nkw;
            const int ikd = tap / (nkh * nkw), ikh = (tap / nkw) % nkh, ikw = tap % nkw;
            kd = SD == 2 ? (nkd == 2 ? 2 * ikd : 1) : ikd;
            kh = SHW == 2 ? (nkh == 2 ? 2 * ikh : 1) : ikh;
            kw = SHW == 2 ? (nkw == 2 ? 2 * ikw : 1) : ikw;
            tap = min((kd * 3 + kh) * 3 + kw, 26);
            t = tap * KQ + cq;
        } else {
            kd = TAPS == 9 ? 1 : tap / 9, kh = TAPS == 9 ? tap / 3 : (tap / 3) % 3, kw = tap % 3;
        }
        int id, ih;
        if (GATHER == 0) {
            id = od * SD - 1 + kd;
            ih = oh * SHW - 1 + kh;
        } else {
            const int nd = od + 1 - kd, nh = oh + 1 - kh;
            rowok = rowok && nd >= 0 && nh >= 0 && (nd % SD) == 0 && (nh % SHW) == 0;
            id = nd / SD;
            ih = nh / SHW;
        }
        rowok = rowok && (unsigned)id < (unsigned)a.Di && (unsigned)ih < (unsigned)a.Hi;
        const unsigned rowbase = (unsigned)((id * a.Hi + ih) * a.Wi) * (unsigned)(CIN * 2) + (unsigned)(cq * 16);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            const int ow = ow0 + (vt * 16 + j) * WSTEP;
            int iw;
            bool v = rowok;
            if (GATHER == 0) {
                iw = ow * SHW - 1 + kw;
            } else {
                const int nw = ow + 1 - kw;
                v = v && nw >= 0 && (nw % SHW) == 0;
                iw = nw / SHW;
            }
            v = v && (unsigned)iw < (unsigned)a.Wi;
            xs[set][vt] = ld8(xr, v ? rowbase + (unsigned)iw * (unsigned)(CIN * 2) : OOB);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)                      // (a step beyond the last reads the last step's fragment: inside the packed image)
            ws[set][nt] = GATHER == 1 ? wp[(size_t)((min(t, NKB - 1) >> 2) * NT + nt) * 64 + (t & 3) * 16] : wp[(size_t)(min(step, STEPS - 1) * NT + nt) * 64];
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) acc[nt][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws[set][nt], xs[set][vt], acc[nt][vt], 0, 0, 0);
    };
    {
        constexpr int DS = KSPLIT ? 4 : 1;
        int step = KSPLIT ? wave : 0;
        load_step(0, step);
        __builtin_amdgcn_sched_barrier(0);
        for (; step + DS < nsteps; step += 2 * DS) {         // pairs of steps
            load_step(1, step + DS);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            load_step(0, step + 2 * DS);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
        }
        if (step < nsteps) mma(0);                           // an odd count: the last step sits in set 0
    }
    if (KSPLIT) {                                          // partial tiles of wavefronts 1..3 -> wavefront 0, added in that order
        if (wave > 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) kred[((wave - 1) * NT * VT + nt * VT + vt) * 64 + lane] = acc[nt][vt];
        }
        __syncthreads();
        if (wave > 0) {
            if (a.stats_rows > 0) {                        // only wavefront 0 has statistics: the others contribute zeros to the block row
                if (lane < 2 * a.Cout) sred[wave][lane] = 0.0f;
                if (lane + 64 < 2 * a.Cout) sred[wave][lane + 64] = 0.0f;
                __syncthreads();
                if ((int)threadIdx.x < 2 * a.Cout)         // ... and wavefront 1 writes the row's entries 64..127 (Cout = 64)
                    a.stats_part[(size_t)threadIdx.x * a.stats_rows + logical] =
                        (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
            }
            return;
        }
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) acc[nt][vt] += kred[(w * NT * VT + nt * VT + vt) * 64 + lane];
    }
    // D[i = co (4*kb + r inside the tile)][j = voxel]: a lane owns 4 consecutive output channels of one voxel -> one 8-byte store
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[nt][r] = ssq[nt][r] = 0.0f;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int ow = ow0 + (vt * 16 + j) * WSTEP;
        if (ow >= a.Wo) continue;
        const size_t vox = (((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co0 = nt * 16 + kb * 4;
            if (co0 >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[nt][vt][r];
                if (a.scale) v[r] = fmaf(v[r], a.scale[co0 + r], a.shift[co0 + r]);
                if (a.relu) v[r] = fmaxf(v[r], 0.0f);
            }
            if (a.residual) {
                const bf16x4 rs = *reinterpret_cast<const bf16x4*>(a.residual + vox * a.Cout + co0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rs[r];
            }
            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4*>(a.y + vox * a.Cout + co0) = o;
            if (a.bn_y) {                                  // BatchNorm backward sums of the layer this gradient belongs to
                const bf16x4 yv = *reinterpret_cast<const bf16x4*>(a.bn_y + vox * a.Cout + co0);
                const int CT = a.Cout * a.bn_groups;
                const float* p4 = a.bn4 + (b % a.bn_groups) * a.Cout + co0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float f = (float)yv[r];
                    float g = (float)o[r];                 // the ROUNDED gradient: what the apply kernel reads
                    if (a.bn_relu && !(fmaf(f, p4[r], p4[CT + r]) > 0.0f)) g = 0.0f;
                    ssum[nt][r] += g;
                    ssq[nt][r] = fmaf(g, (f - p4[2 * CT + r]) * p4[3 * CT + r], ssq[nt][r]);
                }
            } else if (a.stats_part) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float f = (float)o[r];          // what BatchNorm will see
                    ssum[nt][r] += f;
                    ssq[nt][r] = fmaf(f, f, ssq[nt][r]);
                }
            }
        }
    }
    if (a.stats_part) {
        // the 16 lanes of a kb group hold the same 4 channels of different voxels: butterfly over j, lane j = 0 writes
        float* row = a.stats_rows > 0 ? &sred[wave][0] : a.stats_part + (size_t)item * 2 * a.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = ssum[nt][r], s2 = ssq[nt][r];
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    s1 += __shfl_xor(s1, m, 64);
                    s2 += __shfl_xor(s2, m, 64);
                }
                const int co = nt * 16 + kb * 4 + r;
                if (j == 0 && co < a.Cout) {
                    row[co] = s1;
                    row[a.Cout + co] = s2;
                }
            }
        if (a.stats_rows > 0) {                            // the block's four wavefronts in a fixed order -> one column of the row table
            __syncthreads();
            if ((int)threadIdx.x < 2 * a.Cout)
                a.stats_part[(size_t)threadIdx.x * a.stats_rows + logical] =
                    (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[a][b][tap] = sum_{batch, p} A[p][a] * Bt[p*s - 1 + k][b]   (A on the grid the stride divides; both channel-last bf16, dW fp32).
// The contraction runs over VOXELS while memory is channel-contiguous, so both MFMA operands are "columns" of the stored tiles.
// gfx950's LDS transpose read does that turn for free: ds_read_b64_tr_b16 hands lane l the 4 keys (voxels) x channel (l & 15) of a
// [4 voxels][16 channels] block whose 16 lanes each supplied the address of 4 contiguous channels of one voxel (any row stride;
// tools/probe/tr16.hip pins the mapping) - two such reads are one v_mfma_f32_16x16x32_bf16 operand (K = 32 voxels, 8 per lane).
//
// Block = a patch of PH x 16 voxels of one (n, d) plane of A (PH = 2, 4 or 8 rows: what keeps the Bt plane ring under 56 KB of LDS)
// and TA x TB channel tiles (16 x 16 each; TA*TB <= 4).  A work item is a COLUMN of such patches over a depth segment: per depth the
// block stages the A tile [PH*16 voxels][CA block] and only the sd NEW Bt plane tiles [PH*s + 2][16*s + 2][CB] into a ring of four
// depth planes (the other two of the three planes a depth needs were staged by the previous depth) - 16-byte global loads -> registers
// -> LDS, the next depth's loads in flight under this depth's MFMAs; out-of-volume cells arrive as zeros from out-of-range buffer
// loads.  Every K step (2 rows x 16 columns of A) wave w runs taps w, w+4, ... for all its tiles: the A fragments are read once per K
// step, a Bt fragment once per (tap, tb).  A block keeps its partial sums in registers over all its work items and writes ONE slab
// at the end; bf16_wgrad_reduce_kernel adds the slabs in a fixed order (no float atomics: the weight gradient is bit-reproducible).
struct WgradArgs {
    const __bf16* A;
    const __bf16* Bt;
    float* part;              // [gridDim.x][CA][CB][27]
    int nb, CA, CB, Dp, Hp, Wp, Db, Hb, Wb, sd, shw;
    int PH, npr, npc, npatch; // patch rows, patches per column / row of a plane, work items (columns of patches) in total
    int dseg, nseg;           // depths per work item, depth segments per column
    int tap0, ntaps;          // taps [tap0, tap0 + ntaps) of the 27: all of them, or (9, 9) = the centre depth tap (a 2-D kernel); the
                              // slabs are [CA][CB][ntaps]
    int nbx;                  // blocks (= slabs) of this job along x; its channel groups (blockIdx.y of a launch of its own) follow
};

// Jobs of one launch: the weight gradients of SEVERAL layers (all those of a training step that share the <TA, TB> instance) run as one
// grid - block b belongs to job j with start[j] <= b < start[j+1], inside it (b - start[j]) % nbx is the slab and / nbx the channel
// group.  A layer alone is a chain of short round trips on a fraction of the chip (40 us against 4 us of matrix work); fifty of them
// in a row were a quarter of the training step.  Side by side they fill it.  The struct travels as the kernel argument (by value:
// captured with the launch by a hipGraph, no device table to keep alive).
constexpr int WG_GROUP = 24;
struct WgradGroup {
    int njobs;
    int start[WG_GROUP + 1];
    WgradArgs job[WG_GROUP];
};
static_assert(sizeof(WgradGroup) <= 3800, "kernel arguments are limited to 4 KB");

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr16(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 frag_tr(const unsigned short* p0, const unsigned short* p1) {
    const s16x4 lo = lds_tr16(p0), hi = lds_tr16(p1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

constexpr int WG_PW = 16;                                   // patch columns

template <int TA, int TB>                                   // channel tiles per block
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int bx, const int by) {
    constexpr int TT = TA * TB;
    constexpr int NIB = 4;                                  // 16-byte pieces of ONE Bt plane tile per thread, at most
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, kb = lane >> 4;
    const int s = a.shw;
    const int BH = a.PH * s + 2, BW = WG_PW * s + 2;
    constexpr int CAB = TA * 16, CBB = TB * 16;             // channels of the staged tiles (a CA / CB of 8 stages 8, see *_row)
    const int a_row = min(CAB, a.CA), b_row = min(CBB, a.CB);            // stored channels per voxel
    const int ca0 = by * CAB;                               // TB covers all of CB
    unsigned short* sA = smem;                              // [PH*16][a_row]
    unsigned short* sB = smem + a.PH * WG_PW * a_row + 64;  // ring of 4 depth planes [slot][BH][BW][b_row] (+ slack: 8-channel tiles)
    const int plane_elems = BH * BW * b_row;
    const int pcsB = b_row / 8, itemsB = BH * BW * pcsB;    // 16-byte pieces of one plane tile
    const int nvoxA = a.PH * WG_PW, pcsA = a_row / 8, itemsA = nvoxA * pcsA;

    // column-invariant staging maps: item -> (row, col, piece) relative to the tile origin
    int relB[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int it = tid + i * 256;
        const int pc = it % pcsB, v = it / pcsB;
        relB[i] = it < itemsB ? ((v / BW) << 16) | ((v % BW) << 8) | pc : -1;
    }
    int relA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int it = tid + i * 256;
        const int pc = it % pcsA, v = it / pcsA;
        relA[i] = it < itemsA ? ((v / WG_PW) << 16) | ((v % WG_PW) << 8) | pc : -1;
    }
    u32x4 pfB[2][NIB], pfA[2];

    constexpr int NTAP = 7;
    f32x4 acc[NTAP][TT];
#pragma unroll
    for (int q = 0; q < NTAP; ++q)
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int vsub = t16 >> 2, csub = (t16 & 3) * 4;        // transpose-read lane roles: voxel (8*kb + 4*r + t16/4), channels 4*(t16%4)..+3

    // Work item = a COLUMN of patches: (n, patch row, patch column, depth segment [d0, d1)).  Consecutive depths of A need the Bt
    // planes (dp*sd - 1 .. dp*sd + 1): all but sd of them are already in the ring (slot = plane & 3), so a step stages sd new
    // planes instead of three - the Bt halo tile was 3/4 of the kernel's staging traffic.
    for (int col = bx; col < a.npatch; col += a.nbx) {
        int r = col;
        const int seg = r % a.nseg;
        r /= a.nseg;
        const int pcx = r % a.npc;
        r /= a.npc;
        const int pry = r % a.npr, n = r / a.npr;
        const int h0 = pry * a.PH, w0 = pcx * WG_PW;
        const int d0 = seg * a.dseg, d1 = min(d0 + a.dseg, a.Dp);
        const rsrc_t ra = make_rsrc(a.A + (size_t)n * a.Dp * a.Hp * a.Wp * a.CA, (unsigned)((size_t)a.Dp * a.Hp * a.Wp * a.CA * 2));
        const rsrc_t rb = make_rsrc(a.Bt + (size_t)n * a.Db * a.Hb * a.Wb * a.CB, (unsigned)((size_t)a.Db * a.Hb * a.Wb * a.CB * 2));
        auto fetch_a = [&](int dp) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rel = relA[i];
                const int hh = h0 + ((rel >> 16) & 0xFF), ww = w0 + ((rel >> 8) & 0xFF), pc = rel & 0xFF;
                const bool ok = rel >= 0 && hh < a.Hp && ww < a.Wp;
                const unsigned off = (unsigned)((((dp * a.Hp + hh) * a.Wp + ww) * a.CA + ca0 + pc * 8) * 2);
                pfA[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? off : OOB, 0, 0));
            }
        };
        auto fetch_b = [&](int j, int db) {                 // Bt plane db (possibly outside the volume: zeros) -> register set j
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const int rel = relB[i];
                const int hb = h0 * s - 1 + ((rel >> 16) & 0xFF), wb = w0 * s - 1 + ((rel >> 8) & 0xFF);
                const bool ok = rel >= 0 && (unsigned)db < (unsigned)a.Db && (unsigned)hb < (unsigned)a.Hb && (unsigned)wb < (unsigned)a.Wb;
                const unsigned off = (unsigned)((((db * a.Hb + hb) * a.Wb + wb) * a.CB + (rel & 0xFF) * 8) * 2);
                if (i * 256 < itemsB) pfB[j][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? off : OOB, 0, 0));
            }
        };
        auto commit_a = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (relA[i] >= 0) *reinterpret_cast<u32x4*>(sA + (tid + i * 256) * 8) = pfA[i];
        };
        auto commit_b = [&](int j, int db) {
            unsigned short* dst = sB + ((db + 4) & 3) * plane_elems;
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                if (relB[i] >= 0) *reinterpret_cast<u32x4*>(dst + (tid + i * 256) * 8) = pfB[j][i];
        };
        // column prologue: the first depth's three planes (the last sd of them through the steady-state path below)
        __syncthreads();                                    // the previous column's fragments are consumed
        fetch_b(0, d0 * a.sd - 1);                          // one round trip for the prologue planes (one plane if sd == 2)
        if (a.sd == 1) fetch_b(1, d0);
        commit_b(0, d0 * a.sd - 1);
        if (a.sd == 1) commit_b(1, d0);
        fetch_a(d0);
        fetch_b(0, d0 * a.sd + 2 - a.sd);                   // literal register-set indices: the sets must stay in registers
        if (a.sd == 2) fetch_b(1, d0 * a.sd + 1);
        for (int dp = d0; dp < d1; ++dp) {
            __syncthreads();                                // previous depth's fragments consumed
            commit_a();
            commit_b(0, dp * a.sd + 2 - a.sd);
            if (a.sd == 2) commit_b(1, dp * a.sd + 1);
            __syncthreads();
            if (dp + 1 < d1) {                              // next depth's A tile and its sd new Bt planes, in flight under the MFMAs
                fetch_a(dp + 1);
                fetch_b(0, (dp + 1) * a.sd + 2 - a.sd);
                if (a.sd == 2) fetch_b(1, (dp + 1) * a.sd + 1);
            }
            const int dbase = dp * a.sd - 1;                // Bt plane of depth tap kd = dbase + kd
            for (int ks = 0; ks < a.PH / 2; ++ks) {
                int arow[2], acol[2];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    // K position (8*kb + 4*rr + e) -> voxel of the 2 x 16 patch rows: any bijection serves (A and Bt fragments share it).  This one
                    // lets the two lane groups that share an LDS pass (kb = 0, 1 / 2, 3) read NEIGHBOURING runs of 4 voxels - 8 voxels x 32
                    // bytes = every bank once for 16-channel tiles - where 8*kb + 4*rr + e put them 8 voxels = 256 bytes apart, on the same
                    // banks (SQ_LDS_BANK_CONFLICT 35 % of the LDS cycles of the single-tile instance, profiles/r05_pmc_train_final.txt)
                    const int v = 16 * (kb >> 1) + 8 * rr + 4 * (kb & 1) + vsub;
                    arow[rr] = 2 * ks + (v >> 4);
                    acol[rr] = v & 15;
                }
                bf16x8 fa[TA];
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    fa[t] = frag_tr(sA + (arow[0] * WG_PW + acol[0]) * a_row + t * 16 + csub, sA + (arow[1] * WG_PW + acol[1]) * a_row + t * 16 + csub);
#pragma unroll
                for (int q = 0; q < NTAP; ++q) {
                    if (wave + 4 * q >= a.ntaps) break;     // wave-uniform
                    const int tap = a.tap0 + wave + 4 * q;
                    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                    const unsigned short* pl = sB + ((dbase + kd + 4) & 3) * plane_elems;
                    const unsigned short* b0 = pl + (((arow[0] * s + kh) * BW + acol[0] * s + kw) * b_row) + csub;
                    const unsigned short* b1 = pl + (((arow[1] * s + kh) * BW + acol[1] * s + kw) * b_row) + csub;
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) {
                        const bf16x8 fb = frag_tr(b0 + tb * 16, b1 + tb * 16);
#pragma unroll
                        for (int ta = 0; ta < TA; ++ta)
                            acc[q][ta * TB + tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ta], fb, acc[q][ta * TB + tb], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[i = a (4*kb + r)][j = b] -> this block's slab
    float* slab = a.part + (size_t)bx * a.CA * a.CB * a.ntaps;
#pragma unroll
    for (int q = 0; q < NTAP; ++q) {
        const int tap = wave + 4 * q;                       // slab index: relative to tap0
        if (tap >= a.ntaps) break;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int ta = t / TB, tb = t % TB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ca = ca0 + ta * 16 + kb * 4 + r, cb = tb * 16 + t16;
                if (ca < a.CA && cb < a.CB) slab[((size_t)ca * a.CB + cb) * a.ntaps + tap] = acc[q][t][r];
            }
        }
    }
}

template <int TA, int TB>
__global__ __launch_bounds__(256, 2) void bf16_wgrad_kernel(const WgradGroup g) {
    int j = 0;
    while (j + 1 < g.njobs && (int)blockIdx.x >= g.start[j + 1]) ++j;       // block-uniform
    const int local = (int)blockIdx.x - g.start[j];
    const int nbx = g.job[j].nbx;
    wgrad_body<TA, TB>(g.job[j], local % nbx, local / nbx);
}

// dW[i] = sum over slabs, in a fixed order: 64 outputs per block (coalesced), the four waves take every fourth slab each
// row_in / row_out: elements per A channel in the slabs (CB*taps) and in dW (CBout*taps, CBout <= CB: a padded operand's extra
// channels are dropped here instead of by a strided view of the result)
struct WgradReduceJob {
    const float* part;
    float* dW;
    int chunks, n, row_in, row_out;
};
constexpr int WG_RGROUP = 64;
struct WgradReduceGroup {
    int njobs;
    int start[WG_RGROUP + 1];
    WgradReduceJob job[WG_RGROUP];
};
static_assert(sizeof(WgradReduceGroup) <= 3800, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(256) void bf16_wgrad_reduce_kernel(const WgradReduceGroup g) {
    __shared__ float red[4][64];
    int j = 0;
    while (j + 1 < g.njobs && (int)blockIdx.x >= g.start[j + 1]) ++j;       // block-uniform
    const float* __restrict__ part = g.job[j].part;
    const int chunks = g.job[j].chunks, n = g.job[j].n, row_in = g.job[j].row_in, row_out = g.job[j].row_out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = ((int)blockIdx.x - g.start[j]) * 64 + lane;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < n) {
        int c = wave;
        for (; c + 12 < chunks; c += 16) {
            s0 += part[(size_t)c * n + i];
            s1 += part[(size_t)(c + 4) * n + i];
            s2 += part[(size_t)(c + 8) * n + i];
            s3 += part[(size_t)(c + 12) * n + i];
        }
        for (; c < chunks; c += 4) s0 += part[(size_t)c * n + i];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && i < n && (i % row_in) < row_out)
        g.job[j].dW[(size_t)(i / row_in) * row_out + (i % row_in)] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ------------------------------------------------------------------------------------------------ layout / precision converters
// fp32 [B,C,N] (NCDHW) -> bf16 [B,N,C] (NDHWC) and back; C a multiple of 8
template <int C>
__global__ void f32_ncdhw_to_bf16_ndhwc_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= N) return;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 8) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)in[(b * C + c0 + e) * N + i];
        *reinterpret_cast<bf16x8*>(out + (b * N + i) * C + c0) = v;
    }
}
template <int C>
__global__ void bf16_ndhwc_to_f32_ncdhw_kernel(const __bf16* __restrict__ in, float* __restrict__ out, size_t N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= N) return;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(in + (b * N + i) * C + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) out[(b * C + c0 + e) * N + i] = (float)v[e];
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm / ReLU / skip, channel-last bf16
// x is [R rows (voxels), C]; a thread owns one channel octet of a strided set of rows.
constexpr int ROWS_PER_BLOCK = 2048;                        // at most; bn_shape picks fewer rows per block for small tensors

template <bool BWD>
__global__ __launch_bounds__(256) void bf16_bn_reduce_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                             int C, size_t RS, int groups, float* __restrict__ part, int rpb) {
    // grid = (blocks per sample, samples); RS = rows per sample; rpb = rows per block.  Plain BatchNorm: one "sample" of all R rows.  Grouped (the batch
    // holds `groups` independent calls, sample n belongs to group n % groups): parameters of group g live at [g*C, (g+1)*C).
    __shared__ float red[4][2 * 64];
    const int CQ = C / 8;
    const int cq = threadIdx.x % CQ, rsub = threadIdx.x / CQ, rstep = 256 / CQ;
    const int gofs = (int)(blockIdx.y % groups) * C;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.0f;
    float sc[8], sh[8], mu[8], is[8];
    if (BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = scale[gofs + cq * 8 + e];
            sh[e] = shift[gofs + cq * 8 + e];
            mu[e] = mean[gofs + cq * 8 + e];
            is[e] = invstd[gofs + cq * 8 + e];
        }
    }
    const size_t s0 = (size_t)blockIdx.y * RS;
    const size_t r0 = s0 + (size_t)blockIdx.x * rpb, r1 = min(r0 + (size_t)rpb, s0 + RS);
    for (size_t r = r0 + rsub; r < r1; r += rstep) {
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + r * C + cq * 8);
        if (!BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)xv[e];
                s[e] += f;
                q[e] = fmaf(f, f, q[e]);
            }
        } else {
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(dy + r * C + cq * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)xv[e];
                float g = (float)gv[e];
                if (relu && !(fmaf(f, sc[e], sh[e]) > 0.0f)) g = 0.0f;
                s[e] += g;
                q[e] = fmaf(g, (f - mu[e]) * is[e], q[e]);
            }
        }
    }
    // fixed-order reduction (no atomics, bit-reproducible): lanes of a wave that share a channel octet (lane % CQ) combine through
    // an xor-shuffle tree, the four waves through LDS, and the block writes ONE partial row; mvs::launch_partials_reduce adds the rows.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        for (int m = 32; m >= CQ; m >>= 1) {
            s[e] += __shfl_xor(s[e], m, 64);
            q[e] += __shfl_xor(q[e], m, 64);
        }
    }
    if (lane < CQ) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[wave][lane * 8 + e] = s[e];
            red[wave][C + lane * 8 + e] = q[e];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C)
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// y = [relu](x*scale[c] + shift[c]) [+ residual]
__global__ __launch_bounds__(256) void bf16_affine_act_kernel(const __bf16* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const __bf16* __restrict__ res, int relu, int C,
                                                              size_t total8, size_t RS, int groups, __bf16* __restrict__ y) {
    __shared__ float P[2][256];                              // scale | shift of the groups*C <= 256 channels, once per block
    if ((int)threadIdx.x < C * groups) {
        P[0][threadIdx.x] = scale[threadIdx.x];
        P[1][threadIdx.x] = shift[threadIdx.x];
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one channel octet
    if (i >= total8) return;
    const int c0 = (int)(i % (C / 8)) * 8 + (groups > 1 ? (int)(((i / (C / 8)) / RS) % groups) * C : 0);
    const bf16x8 xv = reinterpret_cast<const bf16x8*>(x)[i];
    bf16x8 rv;
    if (res) rv = reinterpret_cast<const bf16x8*>(res)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = fmaf((float)xv[e], P[0][c0 + e], P[1][c0 + e]);
        if (relu) v = fmaxf(v, 0.0f);
        if (res) v += (float)rv[e];
        o[e] = (__bf16)v;
    }
    reinterpret_cast<bf16x8*>(y)[i] = o;
}

// dx = gamma*invstd*(g - s1/n - xhat*s2/n), g = dy*[x*scale+shift > 0 or !relu]
__device__ __forceinline__ double resolve_count(double count_host, const float* __restrict__ count_dev) {
    return count_dev ? (double)count_dev[0] * 4096.0 + (double)count_dev[1] : count_host;
}
__global__ __launch_bounds__(256) void bf16_bn_bwd_apply_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ x,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ sums,
                                                                double count_host, const float* __restrict__ count_dev, int relu, int C,
                                                                size_t total8, size_t RS, int groups, __bf16* __restrict__ dx,
                                                                float* __restrict__ dgb) {
    // per-channel constants once per BLOCK into LDS (they used to be 56 global loads and 16 fp64 divisions per THREAD: the divisions alone
    // were most of the kernel): [scale | shift | mean | invstd | gamma*invstd | sum g / n | sum g*xhat / n], each groups*C <= 256 entries
    __shared__ float P[7][256];
    const int CT = C * groups;                              // sums = [sum g (groups*C)] [sum g*xhat (groups*C)]
    if ((int)threadIdx.x < CT) {
        const int c = threadIdx.x;
        const double count = resolve_count(count_host, count_dev);
        const float is = invstd[c];
        P[0][c] = scale[c], P[1][c] = shift[c], P[2][c] = mean[c], P[3][c] = is;
        P[4][c] = (gamma ? gamma[c] : 1.0f) * is;
        P[5][c] = (float)((double)sums[c] / count);
        P[6][c] = (float)((double)sums[CT + c] / count);
    }
    // the parameter gradients of a GROUPED BatchNorm (shared affine parameters: the groups' sums added, in group order) by block 0 - it
    // holds the sums anyway; saves the separate reduction launch: dgb = [dbeta (C) | dgamma (C)]
    if (dgb && blockIdx.x == 0 && (int)threadIdx.x < 2 * C) {
        const int kind = threadIdx.x / C, c = threadIdx.x % C;
        float acc = 0.0f;
        for (int g = 0; g < groups; ++g) acc += sums[kind * CT + g * C + c];
        dgb[threadIdx.x] = acc;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int c0 = (int)(i % (C / 8)) * 8 + (groups > 1 ? (int)(((i / (C / 8)) / RS) % groups) * C : 0);
    const bf16x8 xv = reinterpret_cast<const bf16x8*>(x)[i], gv = reinterpret_cast<const bf16x8*>(dy)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        const float f = (float)xv[e];
        float g = (float)gv[e];
        if (relu && !(fmaf(f, P[0][c], P[1][c]) > 0.0f)) g = 0.0f;
        o[e] = (__bf16)(P[4][c] * (g - P[5][c] - (f - P[2][c]) * P[3][c] * P[6][c]));
    }
    reinterpret_cast<bf16x8*>(dx)[i] = o;
}

bool chan_ok(int c) { return c == 8 || c == 16 || c == 32 || c == 64; }

// K split over a block's wavefronts (see bf16_conv_kernel): 32 / 64 input channels (27 / 54 steps per wavefront) while the launch has
// at most ~3 work items per CU.  Every block streams the whole packed weight image (221 KB at 64 -> 64) from L2, and with one 64-voxel item
// per block that traffic is what bounds a larger launch: measured per call (tools/exp_small_conv.py, 64 -> 64 stride 1), split / not split:
// 256 items 14 / 37 us, 512 items 28 / 39 us, 1024 items 53 / 41 us (32 -> 64 stride (1,2,2): 11 / 21, 21 / 22, 41 / 24 us).
bool conv_ksplit(int cin, int items, int taps) {
    static const int lim = [] { const char* e = getenv("MVS_BF16_KSPLIT_ITEMS"); return e ? atoi(e) : 768; }();
    return taps == 27 && cin >= 32 && items <= lim;
}

template <int GATHER, int SD, int SHW>
int launch_conv(const ConvArgs& a, int cin, int nt, hipStream_t s) {
    const bool ks = conv_ksplit(cin, a.items, 27);
    const unsigned grid = (unsigned)((((ks ? a.items : (a.items + 3) / 4) + 7) / 8) * 8);
#define MVS_BF16_GO(CINV, NTV, KSV) hipLaunchKernelGGL((bf16_conv_kernel<CINV, NTV, GATHER, SD, SHW, 27, KSV>), dim3(grid), dim3(256), 0, s, a)
#define MVS_BF16_NT(CINV, KSV)                      \
    switch (nt) {                                   \
        case 1: MVS_BF16_GO(CINV, 1, KSV); break;   \
        case 2: MVS_BF16_GO(CINV, 2, KSV); break;   \
        default: MVS_BF16_GO(CINV, 4, KSV); break;  \
    }
    switch (cin) {
        case 8: MVS_BF16_NT(8, false); break;
        case 16: MVS_BF16_NT(16, false); break;
        case 32: if (ks) { MVS_BF16_NT(32, true); } else { MVS_BF16_NT(32, false); } break;
        default: if (ks) { MVS_BF16_NT(64, true); } else { MVS_BF16_NT(64, false); } break;
    }
#undef MVS_BF16_NT
#undef MVS_BF16_GO
    return mvs::finish_launch("mvs_bf16_conv3d");
}

int nt_of(int cout) { return cout <= 16 ? 1 : (cout <= 32 ? 2 : 4); }

}  // namespace

extern "C" int64_t mvs_bf16_packed_elems(int Cin, int Cout) {
    if (!chan_ok(Cin) || !chan_ok(Cout)) return -1;
    const int steps = (27 * Cin / 8 + 3) / 4;
    return (int64_t)steps * nt_of(Cout) * 64 * 8;
}
extern "C" int64_t mvs_bf16_packed_elems_taps(int Cin, int Cout, int taps) {
    if (!chan_ok(Cin) || !chan_ok(Cout) || (taps != 27 && taps != 9)) return -1;
    const int steps = (taps * Cin / 8 + 3) / 4;
    return (int64_t)steps * nt_of(Cout) * 64 * 8;
}

namespace {
// w = [d0][d1][taps]; the packed map is Cin -> Cout with Cin/Cout >= the weight's own extents (missing rows / channels pack as zeros)
bool pack_job(const float* w, int d0, int d1, int src, int Cout, int Cin, int taps, void* out, PackJob* j) {
    if (!w || !chan_ok(Cin) || !chan_ok(Cout) || src < 0 || src > 2 || !out || (taps != 27 && taps != 9) || d0 < 1 || d1 < 1) return false;
    if (!((src == 0 && d0 <= Cout && d1 <= Cin) || (src != 0 && d0 <= Cin && d1 <= Cout))) return false;
    j->d0 = d0, j->d1 = d1, j->src = src, j->M = Cout, j->KC = Cin, j->NT = nt_of(Cout), j->steps = (taps * Cin / 8 + 3) / 4, j->taps = taps;
    j->nblocks = (j->steps * j->NT * 64 + 255) / 256;
    j->w = w;
    j->out = reinterpret_cast<__bf16*>(out);
    return true;
}
}  // namespace

extern "C" int mvs_bf16_pack_weights(const float* w, int d0, int d1, int src, int Cout, int Cin, void* wpacked, mvs_stream_t stream) {
    PackJob a{}, none{};
    MVS_REQUIRE(pack_job(w, d0, d1, src, Cout, Cin, 27, wpacked, &a), "mvs_bf16_pack_weights: weight [%d][%d][27] / src %d / Cin=%d Cout=%d", d0,
                d1, src, Cin, Cout);
    hipLaunchKernelGGL(bf16_pack_kernel, dim3(a.nblocks), dim3(256), 0, MVS_STREAM(stream), a, none);
    return mvs::finish_launch("mvs_bf16_pack_weights");
}

// the forward's and the data gradient's layout of one weight in ONE launch (a training step packs every layer twice)
extern "C" int mvs_bf16_pack_weights2(const float* w, int d0, int d1, int srcA, int CoutA, int CinA, void* packedA, int srcB, int CoutB,
                                      int CinB, void* packedB, mvs_stream_t stream) {
    PackJob a{}, b{};
    MVS_REQUIRE(pack_job(w, d0, d1, srcA, CoutA, CinA, 27, packedA, &a) && pack_job(w, d0, d1, srcB, CoutB, CinB, 27, packedB, &b),
                "mvs_bf16_pack_weights2: weight [%d][%d][27] does not match (src %d: %d->%d) / (src %d: %d->%d)", d0, d1, srcA, CinA, CoutA,
                srcB, CinB, CoutB);
    hipLaunchKernelGGL(bf16_pack_kernel, dim3(a.nblocks + b.nblocks), dim3(256), 0, MVS_STREAM(stream), a, b);
    return mvs::finish_launch("mvs_bf16_pack_weights2");
}

// ---- job table: every weight layout a training step of one stage needs, packed by ONE launch -------------------------------------
// The caller fills a HOST table entry by entry (mvs_bf16_pack_table_fill), copies its mvs_bf16_pack_table_bytes(njobs) bytes to the
// device once (the pointers stay valid while the parameters and the packed buffers live), and runs it every step.  Host layout:
// [njobs PackJob][njobs + 1 int block starts]; fill entries in order 0..njobs-1.
extern "C" int64_t mvs_bf16_pack_table_bytes(int njobs) {
    return njobs < 1 ? -1 : (int64_t)njobs * (int64_t)sizeof(PackJob) + (int64_t)(njobs + 1) * (int64_t)sizeof(int);
}
extern "C" int mvs_bf16_pack_table_fill(void* host_table, int njobs, int index, const float* w, int d0, int d1, int src, int Cout, int Cin,
                                        int taps, void* wpacked) {
    MVS_REQUIRE(host_table && njobs >= 1 && index >= 0 && index < njobs, "mvs_bf16_pack_table_fill: bad table / index");
    PackJob* jobs = reinterpret_cast<PackJob*>(host_table);
    int* start = reinterpret_cast<int*>(jobs + njobs);
    MVS_REQUIRE(pack_job(w, d0, d1, src, Cout, Cin, taps, wpacked, &jobs[index]),
                "mvs_bf16_pack_table_fill: weight [%d][%d][%d] / src %d does not fit a %d -> %d map", d0, d1, taps, src, Cin, Cout);
    if (index == 0) start[0] = 0;
    start[index + 1] = start[index] + jobs[index].nblocks;
    return MVS_OK;
}
extern "C" int mvs_bf16_pack_table_run(const void* dev_table, int njobs, int total_blocks, mvs_stream_t stream) {
    MVS_REQUIRE(dev_table && njobs >= 1 && total_blocks >= 1, "mvs_bf16_pack_table_run: bad arguments");
    const PackJob* jobs = reinterpret_cast<const PackJob*>(dev_table);
    const int* start = reinterpret_cast<const int*>(jobs + njobs);
    hipLaunchKernelGGL(bf16_pack_table_kernel, dim3(total_blocks), dim3(256), 0, MVS_STREAM(stream), jobs, start, njobs);
    return mvs::finish_launch("mvs_bf16_pack_table_run");
}

// gather: 0 = Conv3d (out = (in - 1)/stride + 1), 1 = ConvTranspose3d k3 p1 op(stride-1) (out = in*stride)
static int bf16_conv3d_impl(const void* x, const void* wpacked, const float* scale, const float* shift, const void* residual, void* y,
                            int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd, int shw, int relu, float* stats_part,
                            int groups, float* sums, mvs_stream_t stream, bool block_rows = false, int taps = 27, const void* bn_y = nullptr,
                            const float* bn4 = nullptr, int bn_relu = 0, int bn_groups = 1);

extern "C" int mvs_bf16_conv3d(const void* x, const void* wpacked, const float* scale, const float* shift, const void* residual, void* y,
                               int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd, int shw, int relu,
                               mvs_stream_t stream) {
    return bf16_conv3d_impl(x, wpacked, scale, shift, residual, y, B, Cin, Cout, Di, Hi, Wi, gather, sd, shw, relu, nullptr, 1, nullptr, stream);
}

// mvs_bf16_conv3d without an epilogue and with a tap count: taps = 9 runs a 2-D kernel (the centre depth tap of a [*,*,1,3,3] weight) on a
// D-agnostic volume - the visibility CNN's layers and their data gradients; weights packed with the same tap count
extern "C" int mvs_bf16_conv3d_taps(const void* x, const void* wpacked, void* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather,
                                    int sd, int shw, int taps, mvs_stream_t stream) {
    return bf16_conv3d_impl(x, wpacked, nullptr, nullptr, nullptr, y, B, Cin, Cout, Di, Hi, Wi, gather, sd, shw, 0, nullptr, 1, nullptr, stream,
                            false, taps);
}

// Raw convolution + the batch statistics of its (bf16-rounded) output in the same pass: sums [2*groups*Cout] = [sum | sum of squares]
// per (group, channel), sample b in group b % groups - exactly what mvs_bf16_bn_stats would return for y.  workspace:
// mvs_bf16_conv3d_stats_workspace_bytes (one partial row per work item, then the fixed-order reduce).
extern "C" int64_t mvs_bf16_conv3d_stats_workspace_bytes(int B, int Cout, int Do, int Ho, int Wo) {
    if (!chan_ok(Cout) || B < 1 || Do < 1 || Ho < 1 || Wo < 1) return -1;
    return (int64_t)B * Do * Ho * ((Wo + 127) / 128) * 2 * 2 * Cout * (int64_t)sizeof(float);      // >= the work items of either form
}

extern "C" int mvs_bf16_conv3d_stats(const void* x, const void* wpacked, void* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather,
                                     int sd, int shw, int groups, float* sums, void* workspace, mvs_stream_t stream) {
    MVS_REQUIRE(sums && workspace && groups >= 1 && B % groups == 0, "mvs_bf16_conv3d_stats: bad statistics arguments (B=%d groups=%d)", B, groups);
    return bf16_conv3d_impl(x, wpacked, nullptr, nullptr, nullptr, y, B, Cin, Cout, Di, Hi, Wi, gather, sd, shw, 0,
                            reinterpret_cast<float*>(workspace), groups, sums, stream);
}

static int bf16_conv3d_impl(const void* x, const void* wpacked, const float* scale, const float* shift, const void* residual, void* y,
                            int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd, int shw, int relu, float* stats_part,
                            int groups, float* sums, mvs_stream_t stream, bool block_rows, int taps, const void* bn_y, const float* bn4,
                            int bn_relu, int bn_groups) {
    MVS_REQUIRE(x && wpacked && y, "mvs_bf16_conv3d: null pointer");
    MVS_REQUIRE(taps == 27 || (taps == 9 && gather == 0 && sd == 1 && shw == 1 && Cin <= 16 && Cout <= 16),
                "mvs_bf16_conv3d: taps = 9 (2-D kernel) is built for stride-1 convolutions of up to 16 channels");
    MVS_REQUIRE(chan_ok(Cin) && chan_ok(Cout), "mvs_bf16_conv3d: channels must be 8/16/32/64 (Cin=%d Cout=%d)", Cin, Cout);
    MVS_REQUIRE(B >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && (gather == 0 || gather == 1), "mvs_bf16_conv3d: bad shape");
    MVS_REQUIRE((sd == 1 || sd == 2) && (shw == 1 || shw == 2) && !(sd == 2 && shw == 1), "mvs_bf16_conv3d: stride (%d,%d,%d) is not built", sd, shw, shw);
    MVS_REQUIRE((!scale) == (!shift), "mvs_bf16_conv3d: scale and shift come together");
    MVS_REQUIRE((int64_t)Di * Hi * Wi * Cin * 2 < ((int64_t)1 << 31), "mvs_bf16_conv3d: one sample's input exceeds the 2 GiB buffer range");
    ConvArgs a{};
    a.x = reinterpret_cast<const __bf16*>(x), a.wp = reinterpret_cast<const __bf16*>(wpacked), a.y = reinterpret_cast<__bf16*>(y);
    a.scale = scale, a.shift = shift, a.residual = reinterpret_cast<const __bf16*>(residual), a.relu = relu;
    a.B = B, a.Di = Di, a.Hi = Hi, a.Wi = Wi, a.Cout = Cout;
    if (gather == 0) a.Do = (Di - 1) / sd + 1, a.Ho = (Hi - 1) / shw + 1, a.Wo = (Wi - 1) / shw + 1;
    else a.Do = Di * sd, a.Ho = Hi * shw, a.Wo = Wi * shw;
    // work items: 64 output voxels of a row; the transposed form with column stride 2 takes the two column parities of a 128-voxel chunk as two items
    const bool wpar = gather == 1 && shw == 2;
    a.nwchunks = wpar ? (a.Wo + 127) / 128 : (a.Wo + 63) / 64;
    const int64_t items = (int64_t)B * a.Do * a.Ho * a.nwchunks * (wpar ? 2 : 1);
    MVS_REQUIRE(items < ((int64_t)1 << 30), "mvs_bf16_conv3d: too many rows");
    a.items = (int)items;
    a.bn_y = reinterpret_cast<const __bf16*>(bn_y), a.bn4 = bn4, a.bn_relu = bn_relu, a.bn_groups = bn_groups;
    a.stats_part = stats_part;
    a.stats_rows = block_rows ? (int)(conv_ksplit(Cin, (int)items, taps) ? items : (items + 3) / 4) : 0;      // one row per BLOCK
    hipStream_t s = MVS_STREAM(stream);
    const int nt = nt_of(Cout);
    int rc;
    if (taps == 9) {
        const unsigned grid = (unsigned)((((a.items + 3) / 4 + 7) / 8) * 8);
        if (Cin == 8) hipLaunchKernelGGL((bf16_conv_kernel<8, 1, 0, 1, 1, 9>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((bf16_conv_kernel<16, 1, 0, 1, 1, 9>), dim3(grid), dim3(256), 0, s, a);
        rc = mvs::finish_launch("mvs_bf16_conv3d");
    } else if (gather == 0) {
        if (sd == 1 && shw == 1) rc = launch_conv<0, 1, 1>(a, Cin, nt, s);
        else if (sd == 2) rc = launch_conv<0, 2, 2>(a, Cin, nt, s);
        else rc = launch_conv<0, 1, 2>(a, Cin, nt, s);
    } else {
        if (sd == 1 && shw == 1) rc = launch_conv<1, 1, 1>(a, Cin, nt, s);
        else if (sd == 2) rc = launch_conv<1, 2, 2>(a, Cin, nt, s);
        else rc = launch_conv<1, 1, 2>(a, Cin, nt, s);
    }
    if (rc != MVS_OK || !stats_part || block_rows) return rc;
    // work items are sample-major, so the rows of sample b are [b*bps, (b+1)*bps): the grouped fixed-order reduce applies as is
    mvs::launch_partials_reduce_grouped(stats_part, (int)(items / B), B, groups, Cout, sums, s);
    return mvs::finish_launch("mvs_bf16_conv3d_stats");
}

namespace {
struct WgradPlan { int PH, npr, npc, npatch, dseg, nseg, TA, TB, gy, blocks; size_t lds; };
constexpr size_t WG_RING_B = 56 * 1024;                      // budget of the 4-plane Bt ring (bytes)
int wgrad_nseg(int columns, int Dp, int target) {            // depth segments per column: enough work items, >= 2 depths each
    int nseg = (target + columns - 1) / columns;
    const int cap = Dp / 2 > 1 ? Dp / 2 : 1;
    if (nseg > cap) nseg = cap;
    return nseg < 1 ? 1 : nseg;
}
// grouped: a job of a GROUP launch - the jobs fill the chip together, so each takes fewer blocks (= slabs the reduce reads back) than a
// launch of its own needs: 256 for the single-tile instance, 128 for the others (measured on config 3's step, kernel time per step of
// the instances <1,1> / <2,2> / <1,4> / <2,1> at 512: 0.46 / 0.30 / 0.16 / 0.13 ms, at 256: 0.41 / 0.23 / 0.10 / 0.10, at 128: 0.42 / 0.15 /
// 0.11 / <0.09, at 64: 0.71 / 0.13 / 0.15 / 0.11; block counts proportional to the jobs' patch x depth steps were slower than either).
WgradPlan wgrad_plan(int nbatch, int CA, int CB, int Dp, int Hp, int Wp, int shw, bool grouped = false) {
    WgradPlan p;
    const int nA = (CA + 15) / 16, nB = (CB + 15) / 16;
    p.TB = nB;                                               // one block sees all of CB (nB <= 4)
    p.TA = nA < 4 / nB ? nA : 4 / nB;
    if (p.TA > 2) p.TA = 2;                                  // the A tile is staged as two 16-byte pieces per thread: 32 channels at most
    if (p.TA < 1) p.TA = 1;                                  // (64 x <=16 channels - no layer of the networks - runs as two channel groups)
    p.gy = (nA + p.TA - 1) / p.TA;
    const int b_row = CB < p.TB * 16 ? CB : p.TB * 16, a_row = CA < p.TA * 16 ? CA : p.TA * 16;
    p.PH = 8;
    while (p.PH > 2 && (size_t)4 * (p.PH * shw + 2) * (WG_PW * shw + 2) * b_row * 2 > WG_RING_B) p.PH >>= 1;
    p.npr = (Hp + p.PH - 1) / p.PH;
    p.npc = (Wp + WG_PW - 1) / WG_PW;
    // two resident blocks per CU over all channel groups (three for the single-tile instance was measured: no faster, and every block
    // writes a slab the reduce kernel reads back)
    static const int gblocks = [] { const char* e = getenv("MVS_WGRAD_GROUP_BLOCKS"); return e ? atoi(e) : 0; }();       // diagnostics
    const int target = (grouped ? (gblocks > 0 ? gblocks : p.TA * p.TB == 1 ? 256 : 128) : 512) / p.gy;
    p.nseg = wgrad_nseg(nbatch * p.npr * p.npc, Dp, target);
    p.dseg = (Dp + p.nseg - 1) / p.nseg;
    p.nseg = (Dp + p.dseg - 1) / p.dseg;
    p.npatch = nbatch * p.npr * p.npc * p.nseg;
    // the slab (= block) count must not depend on the stride (the workspace is sized without it): the count at PH = 8
    const int col8 = nbatch * ((Hp + 7) / 8) * p.npc;
    const int seg8 = wgrad_nseg(col8, Dp, target), d8 = (Dp + seg8 - 1) / seg8;
    const int least = col8 * ((Dp + d8 - 1) / d8);
    p.blocks = target < least ? target : least;
    if (p.blocks < 1) p.blocks = 1;
    p.lds = ((size_t)p.PH * WG_PW * a_row + 64 + (size_t)4 * (p.PH * shw + 2) * (WG_PW * shw + 2) * b_row + 64) * 2;
    return p;
}
}  // namespace

extern "C" int64_t mvs_bf16_conv3d_wgrad_workspace_bytes(int nbatch, int CA, int CB, int Dp, int Hp, int Wp) {
    if (!chan_ok(CA) || !chan_ok(CB) || nbatch < 1 || Dp < 1 || Hp < 1 || Wp < 1) return -1;
    // the stride only moves PH (not the block count), so the plan of stride 1 sizes the slabs for both
    return (int64_t)wgrad_plan(nbatch, CA, CB, Dp, Hp, Wp, 1).blocks * CA * CB * 27 * (int64_t)sizeof(float);
}

// ---- the weight gradients of several layers as one grid per kernel instance + one reduce (MvsWgradJob, include/mvs_hip.h) ----
namespace {
constexpr int64_t WG_ALIGN = 256;
int64_t wgrad_slab_bytes(const MvsWgradJob& j, const WgradPlan& p) {
    const int64_t b = (int64_t)p.blocks * j.CA * j.CB * j.taps * (int64_t)sizeof(float);
    return (b + WG_ALIGN - 1) / WG_ALIGN * WG_ALIGN;
}
int wgrad_job_check(const MvsWgradJob& j, WgradPlan& p, bool grouped) {
    MVS_REQUIRE(j.A && j.Bt && j.dW, "mvs_bf16_wgrad_group: null pointer");
    MVS_REQUIRE(chan_ok(j.CA) && chan_ok(j.CB) && j.nbatch >= 1 && j.CBout >= 1 && j.CBout <= j.CB,
                "mvs_bf16_wgrad_group: channels must be 8/16/32/64 (CA=%d CB=%d)", j.CA, j.CB);
    MVS_REQUIRE((j.sd == 1 || j.sd == 2) && (j.shw == 1 || j.shw == 2) && (j.taps == 27 || j.taps == 9), "mvs_bf16_wgrad_group: bad stride / taps");
    MVS_REQUIRE(j.Dp >= 1 && j.Hp >= 1 && j.Wp >= 1 && j.Db >= 1 && j.Hb >= 1 && j.Wb >= 1, "mvs_bf16_wgrad_group: bad grid");
    MVS_REQUIRE((int64_t)j.Dp * j.Hp * j.Wp * j.CA * 2 < ((int64_t)1 << 31) && (int64_t)j.Db * j.Hb * j.Wb * j.CB * 2 < ((int64_t)1 << 31),
                "mvs_bf16_wgrad_group: one sample exceeds the 2 GiB buffer range");
    p = wgrad_plan(j.nbatch, j.CA, j.CB, j.Dp, j.Hp, j.Wp, j.shw, grouped);
    const int b_row = j.CB < p.TB * 16 ? j.CB : p.TB * 16;
    const int itemsB = (p.PH * j.shw + 2) * (WG_PW * j.shw + 2) * (b_row / 8);      // one plane tile
    MVS_REQUIRE(itemsB <= 4 * 256 && p.lds <= 64 * 1024, "mvs_bf16_wgrad_group: Bt halo tile of %d pieces / %zu bytes does not fit (CB=%d stride %d)",
                itemsB, p.lds, j.CB, j.shw);
    return MVS_OK;
}
template <int TA, int TB>
void wgrad_launch(const WgradGroup& g, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((bf16_wgrad_kernel<TA, TB>), dim3(g.start[g.njobs]), dim3(256), lds, s, g);
}
}  // namespace

extern "C" int64_t mvs_bf16_wgrad_group_workspace_bytes(const MvsWgradJob* jobs, int njobs) {
    if (!jobs || njobs < 1) return -1;
    int64_t total = 0;
    for (int i = 0; i < njobs; ++i) {
        const MvsWgradJob& j = jobs[i];
        if (!chan_ok(j.CA) || !chan_ok(j.CB) || j.nbatch < 1 || j.Dp < 1 || j.Hp < 1 || j.Wp < 1 || (j.shw != 1 && j.shw != 2) || (j.taps != 27 && j.taps != 9))
            return -1;
        total += wgrad_slab_bytes(j, wgrad_plan(j.nbatch, j.CA, j.CB, j.Dp, j.Hp, j.Wp, j.shw, njobs > 1));
    }
    return total;
}

extern "C" int mvs_bf16_wgrad_group(const MvsWgradJob* jobs, int njobs, void* workspace, int64_t workspace_bytes, mvs_stream_t stream) {
    MVS_REQUIRE(jobs && njobs >= 1 && njobs <= 4096 && workspace, "mvs_bf16_wgrad_group: bad arguments (njobs=%d)", njobs);
    std::vector<WgradPlan> plan(njobs);
    std::vector<int64_t> off(njobs);
    int64_t total = 0;
    for (int i = 0; i < njobs; ++i) {
        if (int rc = wgrad_job_check(jobs[i], plan[i], njobs > 1)) return rc;
        off[i] = total;
        total += wgrad_slab_bytes(jobs[i], plan[i]);
    }
    MVS_REQUIRE(total <= workspace_bytes, "mvs_bf16_wgrad_group: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)total);
    hipStream_t s = MVS_STREAM(stream);
    // one launch per kernel instance (and per WG_GROUP jobs of it), the longest blocks first: a block's length ~ columns per block x depths
    static const int inst[5][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {1, 4}};
    for (int k = 0; k < 5; ++k) {
        std::vector<int> ids;
        for (int i = 0; i < njobs; ++i)
            if (plan[i].TA == inst[k][0] && plan[i].TB == inst[k][1]) ids.push_back(i);
        auto cost = [&](int i) { return (double)((plan[i].npatch + plan[i].blocks - 1) / plan[i].blocks) * plan[i].dseg * plan[i].PH * jobs[i].taps; };
        std::stable_sort(ids.begin(), ids.end(), [&](int x, int y) { return cost(x) > cost(y); });
        for (size_t first = 0; first < ids.size(); first += WG_GROUP) {
            WgradGroup g{};
            size_t lds = 0;
            g.njobs = (int)std::min<size_t>(WG_GROUP, ids.size() - first);
            for (int q = 0; q < g.njobs; ++q) {
                const int i = ids[first + q];
                const MvsWgradJob& j = jobs[i];
                const WgradPlan& p = plan[i];
                WgradArgs& a = g.job[q];
                a.A = reinterpret_cast<const __bf16*>(j.A), a.Bt = reinterpret_cast<const __bf16*>(j.Bt);
                a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + off[i]);
                a.nb = j.nbatch, a.CA = j.CA, a.CB = j.CB, a.Dp = j.Dp, a.Hp = j.Hp, a.Wp = j.Wp, a.Db = j.Db, a.Hb = j.Hb, a.Wb = j.Wb, a.sd = j.sd, a.shw = j.shw;
                a.PH = p.PH, a.npr = p.npr, a.npc = p.npc, a.npatch = p.npatch, a.dseg = p.dseg, a.nseg = p.nseg;
                a.tap0 = j.taps == 9 ? 9 : 0, a.ntaps = j.taps, a.nbx = p.blocks;
                g.start[q + 1] = g.start[q] + p.blocks * p.gy;
                lds = std::max(lds, p.lds);
            }
            if (k == 0) wgrad_launch<1, 1>(g, lds, s);
            else if (k == 1) wgrad_launch<1, 2>(g, lds, s);
            else if (k == 2) wgrad_launch<2, 1>(g, lds, s);
            else if (k == 3) wgrad_launch<2, 2>(g, lds, s);
            else wgrad_launch<1, 4>(g, lds, s);
            if (int rc = mvs::finish_launch("mvs_bf16_wgrad_group")) return rc;
        }
    }
    for (int first = 0; first < njobs; first += WG_RGROUP) {
        WgradReduceGroup g{};
        g.njobs = std::min(WG_RGROUP, njobs - first);
        for (int q = 0; q < g.njobs; ++q) {
            const int i = first + q;
            const MvsWgradJob& j = jobs[i];
            const int n = j.CA * j.CB * j.taps;
            g.job[q] = WgradReduceJob{reinterpret_cast<const float*>(static_cast<char*>(workspace) + off[i]), j.dW, plan[i].blocks, n, j.CB * j.taps,
                                      j.CBout * j.taps};
            g.start[q + 1] = g.start[q] + (n + 63) / 64;
        }
        hipLaunchKernelGGL(bf16_wgrad_reduce_kernel, dim3(g.start[g.njobs]), dim3(256), 0, s, g);
        if (int rc = mvs::finish_launch("mvs_bf16_wgrad_group")) return rc;
    }
    return MVS_OK;
}

// taps = 27: dW [CA][CBout][27]; taps = 9: the centre depth tap only, dW [CA][CBout][9] (a 2-D kernel's gradient).  CBout <= CB drops the
// padding channels of Bt (the visibility CNN's 1-channel input lives in an 8-channel tensor).  One layer = a group of one.
extern "C" int mvs_bf16_conv3d_wgrad_taps(const void* A, const void* Bt, float* dW, void* workspace, int nbatch, int CA, int CB, int CBout,
                                          int Dp, int Hp, int Wp, int Db, int Hb, int Wb, int sd, int shw, int taps, mvs_stream_t stream) {
    MVS_REQUIRE(A && Bt && dW && workspace, "mvs_bf16_conv3d_wgrad: null pointer");
    MVS_REQUIRE(chan_ok(CA) && chan_ok(CB) && nbatch >= 1, "mvs_bf16_conv3d_wgrad: channels must be 8/16/32/64 (CA=%d CB=%d)", CA, CB);
    const MvsWgradJob job{A, Bt, dW, nbatch, CA, CB, CBout, Dp, Hp, Wp, Db, Hb, Wb, sd, shw, taps, 0};
    // the caller's workspace is mvs_bf16_conv3d_wgrad_workspace_bytes (sized for 27 taps): at least what the one job needs
    return mvs_bf16_wgrad_group(&job, 1, workspace, mvs_bf16_conv3d_wgrad_workspace_bytes(nbatch, CA, CB, Dp, Hp, Wp), stream);
}

extern "C" int mvs_bf16_conv3d_wgrad(const void* A, const void* Bt, float* dW, void* workspace, int nbatch, int CA, int CB, int Dp, int Hp,
                                     int Wp, int Db, int Hb, int Wb, int sd, int shw, mvs_stream_t stream) {
    return mvs_bf16_conv3d_wgrad_taps(A, Bt, dW, workspace, nbatch, CA, CB, CB, Dp, Hp, Wp, Db, Hb, Wb, sd, shw, 27, stream);
}

extern "C" int mvs_bf16_from_f32_ncdhw(const float* in, void* out, int B, int C, int64_t N, mvs_stream_t stream) {
    MVS_REQUIRE(in && out && B >= 1 && B <= 65535 && N >= 1 && chan_ok(C), "mvs_bf16_from_f32_ncdhw: bad arguments (C=%d)", C);
    dim3 grid((unsigned)((N + 255) / 256), B);
    hipStream_t s = MVS_STREAM(stream);
    __bf16* o = reinterpret_cast<__bf16*>(out);
    switch (C) {
        case 8: hipLaunchKernelGGL(f32_ncdhw_to_bf16_ndhwc_kernel<8>, grid, dim3(256), 0, s, in, o, (size_t)N); break;
        case 16: hipLaunchKernelGGL(f32_ncdhw_to_bf16_ndhwc_kernel<16>, grid, dim3(256), 0, s, in, o, (size_t)N); break;
        case 32: hipLaunchKernelGGL(f32_ncdhw_to_bf16_ndhwc_kernel<32>, grid, dim3(256), 0, s, in, o, (size_t)N); break;
        default: hipLaunchKernelGGL(f32_ncdhw_to_bf16_ndhwc_kernel<64>, grid, dim3(256), 0, s, in, o, (size_t)N); break;
    }
    return mvs::finish_launch("mvs_bf16_from_f32_ncdhw");
}

extern "C" int mvs_bf16_to_f32_ncdhw(const void* in, float* out, int B, int C, int64_t N, mvs_stream_t stream) {
    MVS_REQUIRE(in && out && B >= 1 && B <= 65535 && N >= 1 && chan_ok(C), "mvs_bf16_to_f32_ncdhw: bad arguments (C=%d)", C);
    dim3 grid((unsigned)((N + 255) / 256), B);
    hipStream_t s = MVS_STREAM(stream);
    const __bf16* i = reinterpret_cast<const __bf16*>(in);
    switch (C) {
        case 8: hipLaunchKernelGGL(bf16_ndhwc_to_f32_ncdhw_kernel<8>, grid, dim3(256), 0, s, i, out, (size_t)N); break;
        case 16: hipLaunchKernelGGL(bf16_ndhwc_to_f32_ncdhw_kernel<16>, grid, dim3(256), 0, s, i, out, (size_t)N); break;
        case 32: hipLaunchKernelGGL(bf16_ndhwc_to_f32_ncdhw_kernel<32>, grid, dim3(256), 0, s, i, out, (size_t)N); break;
        default: hipLaunchKernelGGL(bf16_ndhwc_to_f32_ncdhw_kernel<64>, grid, dim3(256), 0, s, i, out, (size_t)N); break;
    }
    return mvs::finish_launch("mvs_bf16_to_f32_ncdhw");
}

namespace {
// Grouped BatchNorm over a channel-last batch (groups independent calls of the same module interleaved in the batch dimension, sample n
// -> group n % groups: the visibility CNN, applied once per source view by the reference): statistics per (group, channel); every
// per-channel array is [groups*C], sums are [sum (groups*C)][sum sq (groups*C)] - the layout mvs_bn_finalize_grouped consumes.
struct BnShape { int64_t RS; int nsamples; unsigned bps; int rpb; };
bool bn_shape(int C, int64_t R, int groups, int64_t rows_per_sample, BnShape* o) {
    if (!chan_ok(C) || R < 1 || groups < 1) return false;
    if (groups == 1) { o->RS = R; o->nsamples = 1; }
    else {
        if (rows_per_sample < 1 || R % rows_per_sample || (R / rows_per_sample) % groups || R / rows_per_sample > 65535) return false;
        o->RS = rows_per_sample; o->nsamples = (int)(R / rows_per_sample);
    }
    // rows per block: a multiple of the rows one sweep of the block's 256 threads covers (256 / (C/8)), sized for ~512 blocks in total so that
    // a 5 MB tensor is not reduced by 20 blocks (a 64-channel level-3 map: 47 us for 41 000 rows), at most ROWS_PER_BLOCK
    const int rstep = 256 / (C / 8);
    int64_t rpb = (o->RS * o->nsamples + 511) / 512;
    rpb = ((rpb + rstep - 1) / rstep) * rstep;
    if (rpb < 2 * rstep) rpb = 2 * rstep;
    if (rpb > ROWS_PER_BLOCK) rpb = ROWS_PER_BLOCK;
    o->rpb = (int)rpb;
    o->bps = (unsigned)((o->RS + rpb - 1) / rpb);
    return true;
}
}  // namespace

extern "C" int64_t mvs_bf16_bn_reduce_workspace_bytes(int C, int64_t R, int groups, int64_t rows_per_sample) {
    BnShape sh;
    if (!bn_shape(C, R, groups, rows_per_sample, &sh)) return -1;
    return (int64_t)sh.bps * sh.nsamples * 2 * C * (int64_t)sizeof(float);
}

extern "C" int mvs_bf16_bn_stats(const void* x, int C, int64_t R, int groups, int64_t rows_per_sample, float* sums, void* workspace,
                                 mvs_stream_t stream) {
    BnShape sh;
    MVS_REQUIRE(x && sums && workspace && bn_shape(C, R, groups, rows_per_sample, &sh), "mvs_bf16_bn_stats: bad arguments");
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(bf16_bn_reduce_kernel<false>, dim3(sh.bps, sh.nsamples), dim3(256), 0, MVS_STREAM(stream),
                       reinterpret_cast<const __bf16*>(x), (const __bf16*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0, C, (size_t)sh.RS, groups, part, sh.rpb);
    mvs::launch_partials_reduce_grouped(part, (int)sh.bps, sh.nsamples, groups, C, sums, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_bf16_bn_stats");
}

extern "C" int mvs_bf16_affine_act(const void* x, const float* scale, const float* shift, const void* residual, int relu, int C, int64_t R,
                                   int groups, int64_t rows_per_sample, void* y, mvs_stream_t stream) {
    BnShape sh;
    MVS_REQUIRE(x && scale && shift && y && bn_shape(C, R, groups, rows_per_sample, &sh) && (int64_t)C * groups <= 256,
                "mvs_bf16_affine_act: bad arguments (channels x groups <= 256)");
    const size_t total8 = (size_t)R * (C / 8);
    hipLaunchKernelGGL(bf16_affine_act_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, MVS_STREAM(stream),
                       reinterpret_cast<const __bf16*>(x), scale, shift, reinterpret_cast<const __bf16*>(residual), relu, C, total8,
                       (size_t)sh.RS, groups, reinterpret_cast<__bf16*>(y));
    return mvs::finish_launch("mvs_bf16_affine_act");
}

extern "C" int mvs_bf16_bn_bwd_reduce(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                                      const float* invstd, int relu, int C, int64_t R, int groups, int64_t rows_per_sample, float* sums,
                                      void* workspace, mvs_stream_t stream) {
    BnShape sh;
    MVS_REQUIRE(dy && x && scale && shift && mean && invstd && sums && workspace && bn_shape(C, R, groups, rows_per_sample, &sh),
                "mvs_bf16_bn_bwd_reduce: bad arguments");
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(bf16_bn_reduce_kernel<true>, dim3(sh.bps, sh.nsamples), dim3(256), 0, MVS_STREAM(stream),
                       reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(dy), scale, shift, mean, invstd, relu, C,
                       (size_t)sh.RS, groups, part, sh.rpb);
    mvs::launch_partials_reduce_grouped(part, (int)sh.bps, sh.nsamples, groups, C, sums, MVS_STREAM(stream));
    return mvs::finish_launch("mvs_bf16_bn_bwd_reduce");
}

namespace {
// partial rows -> per-(group, channel) sums -> mean / invstd / scale / shift and the running-statistics update, in ONE launch: the
// arithmetic of partials_reduce_grouped followed by bn_finalize(_grouped) of train.hip (same order, same double-precision steps).
// One wavefront per base channel walks the groups in order (the running statistics take the groups' updates in order).
__global__ __launch_bounds__(256) void bf16_bn_reduce_finalize_kernel(const float* __restrict__ part, int bps, int nsamples, int groups, int C,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                      float momentum, float eps, double count, float* __restrict__ out4,
                                                                      long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    if (num_batches_tracked && c == 0 && lane == 0) num_batches_tracked[0] += groups;       // nn.BatchNorm's counter: one call per group
    const int CT = C * groups, per = (nsamples / groups) * bps;
    const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    float rm = running_mean ? running_mean[c] : 0.0f, rv = running_var ? running_var[c] : 0.0f;
    for (int q = 0; q < groups; ++q) {
        float s1 = 0.0f, s2 = 0.0f;
        for (int i = lane; i < per; i += 64) {
            const size_t row = (size_t)(q + (i / bps) * groups) * bps + (i % bps);
            s1 += part[row * 2 * C + c];
            s2 += part[row * 2 * C + C + c];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
        }
        const int cc = q * C + c;
        const double mean = (double)s1 / count;
        double var = (double)s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (lane == 0) {
            out4[cc] = g * invstd;                           // scale
            out4[CT + cc] = bt - (float)mean * g * invstd;   // shift
            out4[2 * CT + cc] = (float)mean;
            out4[3 * CT + cc] = invstd;
            out4[4 * CT + cc] = g;                          // the affine weight per (group, channel): what the backward's apply kernel reads
        }
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rm = (1.0f - momentum) * rm + momentum * (float)mean;
        rv = (1.0f - momentum) * rv + momentum * (float)unbiased;
    }
    if (running_mean && lane == 0) {
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}
}  // namespace

// Training-mode BatchNorm forward of a channel-last bf16 tensor in one call and three launches (statistics partials; reduce + finalize;
// normalize + ReLU + skip) - what mvs_bf16_bn_stats + mvs_bn_finalize(_grouped) + mvs_bf16_affine_act do in four, for the case without a
// cross-rank reduction in between.  stats4 = [scale | shift | mean | invstd], each groups*C.
extern "C" int mvs_bf16_bn_train_fwd(const void* x, const void* residual, int relu, int C, int64_t R, int groups, int64_t rows_per_sample,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                     float eps, int64_t* num_batches_tracked, float* stats4, void* y, void* workspace, mvs_stream_t stream) {
    BnShape sh;
    MVS_REQUIRE(x && stats4 && y && workspace && bn_shape(C, R, groups, rows_per_sample, &sh) && (int64_t)C * groups <= 256,
                "mvs_bf16_bn_train_fwd: bad arguments (channels x groups <= 256)");
    MVS_REQUIRE((!running_mean) == (!running_var), "mvs_bf16_bn_train_fwd: running_mean and running_var come together");
    hipStream_t s = MVS_STREAM(stream);
    float* part = reinterpret_cast<float*>(workspace);
    const int CT = C * groups;
    hipLaunchKernelGGL(bf16_bn_reduce_kernel<false>, dim3(sh.bps, sh.nsamples), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x),
                       (const __bf16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, C,
                       (size_t)sh.RS, groups, part, sh.rpb);
    hipLaunchKernelGGL(bf16_bn_reduce_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, (int)sh.bps, sh.nsamples, groups, C, gamma, beta,
                       running_mean, running_var, momentum, eps, (double)(R / groups), stats4,
                       reinterpret_cast<long long*>(num_batches_tracked));
    const size_t total8 = (size_t)R * (C / 8);
    hipLaunchKernelGGL(bf16_affine_act_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x),
                       stats4, stats4 + CT, reinterpret_cast<const __bf16*>(residual), relu, C, total8, (size_t)sh.RS, groups,
                       reinterpret_cast<__bf16*>(y));
    return mvs::finish_launch("mvs_bf16_bn_train_fwd");
}

static int bn_bwd_apply_impl(const void* dy, const void* x, const float* scale, const float* shift, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, double count, const float* count_dev, int relu, int C, int64_t R, int groups,
                             int64_t rows_per_sample, void* dx, float* dgb, mvs_stream_t stream);

extern "C" int mvs_bf16_bn_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const float* gamma, const float* sums, double count, const float* count_dev,
                                     int relu, int C, int64_t R, int groups, int64_t rows_per_sample, void* dx, mvs_stream_t stream) {
    return bn_bwd_apply_impl(dy, x, scale, shift, mean, invstd, gamma, sums, count, count_dev, relu, C, R, groups, rows_per_sample, dx, nullptr, stream);
}

// the same + dgb [2*C] = [dbeta | dgamma] of the shared affine parameters (the groups' sums added in group order), written by the kernel's
// first block: no separate reduction launch for a grouped BatchNorm's parameter gradients
extern "C" int mvs_bf16_bn_bwd_apply_dgb(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                                         const float* invstd, const float* gamma, const float* sums, double count, const float* count_dev,
                                         int relu, int C, int64_t R, int groups, int64_t rows_per_sample, void* dx, float* dgb,
                                         mvs_stream_t stream) {
    MVS_REQUIRE(dgb, "mvs_bf16_bn_bwd_apply_dgb: null dgb");
    return bn_bwd_apply_impl(dy, x, scale, shift, mean, invstd, gamma, sums, count, count_dev, relu, C, R, groups, rows_per_sample, dx, dgb, stream);
}

static int bn_bwd_apply_impl(const void* dy, const void* x, const float* scale, const float* shift, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, double count, const float* count_dev, int relu, int C, int64_t R, int groups,
                             int64_t rows_per_sample, void* dx, float* dgb, mvs_stream_t stream) {
    BnShape sh;
    MVS_REQUIRE(dy && x && scale && shift && mean && invstd && sums && dx && bn_shape(C, R, groups, rows_per_sample, &sh),
                "mvs_bf16_bn_bwd_apply: bad arguments");
    MVS_REQUIRE((int64_t)C * groups <= 256, "mvs_bf16_bn_bwd_apply: %d channels x %d groups exceed the 256 per-channel constants kept in LDS", C, groups);
    const size_t total8 = (size_t)R * (C / 8);
    hipLaunchKernelGGL(bf16_bn_bwd_apply_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, MVS_STREAM(stream),
                       reinterpret_cast<const __bf16*>(dy), reinterpret_cast<const __bf16*>(x), scale, shift, mean, invstd, gamma, sums, count,
                       count_dev, relu, C, total8, (size_t)sh.RS, groups, reinterpret_cast<__bf16*>(dx), dgb);
    return mvs::finish_launch("mvs_bf16_bn_bwd_apply");
}


// =====================================================================================================================================
// Round 5: fewer, fatter launches for the training step (the step is a serial chain of ~1100 graph nodes; per-node cost, not bytes, is
// what a 640x512 sample pays).
//   mvs_bf16_conv3d_bn_fwd   raw conv (+ the batch statistics of its output as BLOCK rows in the epilogue) -> finalize (one block per
//                            channel: the rows in a fixed order, mean / invstd / scale / shift, running statistics) -> normalize + ReLU
//                            + skip: 3 launches for what conv + mvs_bf16_bn_train_fwd did in 4, and no separate pass over y.
// =====================================================================================================================================
namespace {
// part = [2C][nrows] (transposed rows: column r = block r of the convolution); rows of sample b = [b*rps, (b+1)*rps); sample b belongs to
// group b % groups.  One block per base channel; the groups in order (running statistics take their updates in order).
__global__ __launch_bounds__(256) void bf16_bn_rows_finalize_kernel(const float* __restrict__ part, int nrows, int rps, int nsamples, int groups,
                                                                    int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                    float momentum, float eps, double count, float* __restrict__ out4,
                                                                    long long* __restrict__ num_batches_tracked) {
    __shared__ float red[2][4];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (num_batches_tracked && c == 0 && tid == 0) num_batches_tracked[0] += groups;
    const int CT = C * groups, per = (nsamples / groups) * rps;
    const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    float rm = running_mean ? running_mean[c] : 0.0f, rv = running_var ? running_var[c] : 0.0f;
    const float* p1 = part + (size_t)c * nrows;
    const float* p2 = part + (size_t)(C + c) * nrows;
    for (int q = 0; q < groups; ++q) {
        float s1 = 0.0f, s2 = 0.0f;
        for (int i = tid; i < per; i += 256) {
            const int row = (q + (i / rps) * groups) * rps + (i % rps);
            s1 += p1[row];
            s2 += p2[row];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
        }
        __syncthreads();                                    // the previous group's red[] is consumed
        if (lane == 0) {
            red[0][wave] = s1;
            red[1][wave] = s2;
        }
        __syncthreads();
        s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const int cc = q * C + c;
        const double mean = (double)s1 / count;
        double var = (double)s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (tid == 0) {
            out4[cc] = g * invstd;
            out4[CT + cc] = bt - (float)mean * g * invstd;
            out4[2 * CT + cc] = (float)mean;
            out4[3 * CT + cc] = invstd;
            out4[4 * CT + cc] = g;                          // the affine weight per (group, channel): what the backward's apply kernel reads
        }
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rm = (1.0f - momentum) * rm + momentum * (float)mean;
        rv = (1.0f - momentum) * rv + momentum * (float)unbiased;
    }
    if (running_mean && tid == 0) {
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

}  // namespace

namespace {
// part = [2C][nrows] transposed block rows (rows of sample b = [b*rps, (b+1)*rps), sample b in group b % groups) -> sums
// [sum (groups*C) | second sum (groups*C)], fixed order: one block per (group, row kind j), its threads stride over the group's rows
__global__ __launch_bounds__(256) void bf16_rows_reduce_kernel(const float* __restrict__ part, int nrows, int rps, int nsamples, int groups, int C,
                                                               float* __restrict__ sums) {
    __shared__ float red[4];
    const int q = blockIdx.x / (2 * C), j = blockIdx.x % (2 * C);
    const int per = (nsamples / groups) * rps;
    const float* p = part + (size_t)j * nrows;
    float s = 0.0f;
    for (int i = threadIdx.x; i < per; i += 256) s += p[(q + (i / rps) * groups) * rps + (i % rps)];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[(j >= C ? groups * C : 0) + q * C + (j % C)] = (red[0] + red[1]) + (red[2] + red[3]);
}
}  // namespace

// Raw convolution (no epilogue arithmetic) whose OUTPUT is the gradient dz arriving at a BatchNorm(+ReLU) layer - the data gradient of the layer
// after it - with that BatchNorm's two backward sums taken in the convolution's epilogue: sums [2*groups*Cout] = [sum g | sum g*xhat],
// g = dz * [bn_y*scale + shift > 0 or !relu], exactly what mvs_bf16_bn_bwd_reduce(dz, bn_y, ...) returns, without the pass over dz and bn_y.
// bn4 = the forward's stats4 ([scale | shift | mean | invstd], each groups*Cout); workspace = mvs_bf16_conv3d_bn_fwd_workspace_bytes.
// addend (optional, y's shape): a second gradient of the same tensor (it also fed a skip connection), added before the rounding and the
// sums - the total gradient leaves this launch, no separate sum of the two and no separate reduce.
extern "C" int mvs_bf16_conv3d_bnbwd(const void* x, const void* wpacked, void* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd,
                                     int shw, int taps, const void* bn_y, const float* bn4, int relu, int groups, const void* addend, float* sums,
                                     void* workspace, mvs_stream_t stream) {
    MVS_REQUIRE(bn_y && bn4 && sums && workspace && groups >= 1 && B % groups == 0, "mvs_bf16_conv3d_bnbwd: bad arguments (B=%d groups=%d)", B, groups);
    MVS_REQUIRE((sd == 1 || sd == 2) && (shw == 1 || shw == 2) && (gather == 0 || gather == 1), "mvs_bf16_conv3d_bnbwd: bad stride / gather");
    int Do, Ho, Wo;
    if (gather == 0) Do = (Di - 1) / sd + 1, Ho = (Hi - 1) / shw + 1, Wo = (Wi - 1) / shw + 1;
    else Do = Di * sd, Ho = Hi * shw, Wo = Wi * shw;
    const int64_t ips = (gather == 1 && shw == 2) ? (int64_t)Do * Ho * ((Wo + 127) / 128) * 2 : (int64_t)Do * Ho * ((Wo + 63) / 64);
    const bool ks = conv_ksplit(Cin, (int)(ips * B), taps);
    MVS_REQUIRE(B == 1 || ks || ips % 4 == 0, "mvs_bf16_conv3d_bnbwd: %lld work items per sample are not a multiple of 4", (long long)ips);
    if (int rc = bf16_conv3d_impl(x, wpacked, nullptr, nullptr, addend, y, B, Cin, Cout, Di, Hi, Wi, gather, sd, shw, 0,
                                  reinterpret_cast<float*>(workspace), groups, nullptr, stream, true, taps, bn_y, bn4, relu, groups))
        return rc;
    const int nrows = ks ? (int)(ips * B) : (int)((ips * B + 3) / 4), rps = B == 1 ? nrows : (int)(ks ? ips : ips / 4);
    hipLaunchKernelGGL(bf16_rows_reduce_kernel, dim3(groups * 2 * Cout), dim3(256), 0, MVS_STREAM(stream), reinterpret_cast<const float*>(workspace),
                       nrows, rps, B, groups, Cout, sums);
    return mvs::finish_launch("mvs_bf16_conv3d_bnbwd");
}

extern "C" int64_t mvs_bf16_conv3d_bn_fwd_workspace_bytes(int B, int Cout, int Do, int Ho, int Wo) {
    if (!chan_ok(Cout) || B < 1 || Do < 1 || Ho < 1 || Wo < 1) return -1;
    const int64_t items = (int64_t)B * Do * Ho * ((Wo + 127) / 128) * 2;           // >= the work items of either form (see bf16_conv3d_impl)
    return items * 2 * Cout * (int64_t)sizeof(float);      // one row per block: up to one block per work item (the K-split form)
}

// y = raw conv(x) (kept: the BatchNorm backward needs it), z = [relu](BatchNorm_train(y)) [+ residual]; stats4 = [scale | shift | mean |
// invstd | gamma], FIVE rows of groups*Cout (the fifth = the affine weight replicated per group, 1 without one: the backward's operand).  groups > 1: sample b belongs to group b % groups, and the work items of one sample (Do*Ho*ceil(Wo/64))
// must be a multiple of 4 so that a block never straddles two samples.
extern "C" int mvs_bf16_conv3d_bn_fwd(const void* x, const void* wpacked, void* y, void* z, const void* residual, int relu, int B, int Cin,
                                      int Cout, int Di, int Hi, int Wi, int gather, int sd, int shw, int taps, int groups, const float* gamma,
                                      const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                      int64_t* num_batches_tracked, float* stats4, void* workspace, mvs_stream_t stream) {
    MVS_REQUIRE(z && stats4 && workspace && groups >= 1 && B % groups == 0 && (int64_t)Cout * groups <= 256,
                "mvs_bf16_conv3d_bn_fwd: bad arguments (B=%d groups=%d; channels x groups <= 256)", B, groups);
    MVS_REQUIRE((!running_mean) == (!running_var), "mvs_bf16_conv3d_bn_fwd: running_mean and running_var come together");
    MVS_REQUIRE((sd == 1 || sd == 2) && (shw == 1 || shw == 2) && (gather == 0 || gather == 1), "mvs_bf16_conv3d_bn_fwd: bad stride / gather");
    int Do, Ho, Wo;
    if (gather == 0) Do = (Di - 1) / sd + 1, Ho = (Hi - 1) / shw + 1, Wo = (Wi - 1) / shw + 1;
    else Do = Di * sd, Ho = Hi * shw, Wo = Wi * shw;
    const int64_t ips = (gather == 1 && shw == 2) ? (int64_t)Do * Ho * ((Wo + 127) / 128) * 2 : (int64_t)Do * Ho * ((Wo + 63) / 64);   // work items per sample
    const bool ks = conv_ksplit(Cin, (int)(ips * B), taps);                  // one work item per block then: four times the rows
    // (one statistics group: every row belongs to it, a block may straddle samples; the K-split form has one work item per row anyway)
    MVS_REQUIRE(B == 1 || groups == 1 || ks || ips % 4 == 0,
                "mvs_bf16_conv3d_bn_fwd: %lld work items per sample are not a multiple of 4 (block rows would straddle samples of different groups)", (long long)ips);
    if (int rc = bf16_conv3d_impl(x, wpacked, nullptr, nullptr, nullptr, y, B, Cin, Cout, Di, Hi, Wi, gather, sd, shw, 0,
                                  reinterpret_cast<float*>(workspace), groups, nullptr, stream, true, taps))
        return rc;
    hipStream_t s = MVS_STREAM(stream);
    const bool one = B == 1 || groups == 1;                                  // all rows are one group's
    const int nrows = ks ? (int)(ips * B) : (int)((ips * B + 3) / 4), rps = one ? nrows : (int)(ks ? ips : ips / 4);
    const int64_t R = (int64_t)B * Do * Ho * Wo;
    hipLaunchKernelGGL(bf16_bn_rows_finalize_kernel, dim3(Cout), dim3(256), 0, s, reinterpret_cast<const float*>(workspace), nrows, rps, one ? 1 : B, groups,
                       Cout, gamma, beta, running_mean, running_var, momentum, eps, (double)(R / groups), stats4,
                       reinterpret_cast<long long*>(num_batches_tracked));
    const size_t total8 = (size_t)R * (Cout / 8);
    const int CT = Cout * groups;
    hipLaunchKernelGGL(bf16_affine_act_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(y),
                       stats4, stats4 + CT, reinterpret_cast<const __bf16*>(residual), relu, Cout, total8, (size_t)(R / B), groups,
                       reinterpret_cast<__bf16*>(z));
    return mvs::finish_launch("mvs_bf16_conv3d_bn_fwd");
}
