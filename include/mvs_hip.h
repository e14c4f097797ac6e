/*
 * mvs_hip.h — C ABI of libmvs_hip.so, the MI355X (gfx950) plane-sweep cost-volume path.
 *
 * The reference (ewrfcas/MVSFormer) has no FFI: its "operator interface" for this path is the set of
 * Python symbols models/mvsformer_model.py looks up.  Each entry point below replaces the
 * ATen launches behind one of those symbols; the reference file:line it stands in for is cited on every
 * declaration, and INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions (all entry points):
 *   - return 0 on success, MVS_EINVAL (-22) for a shape/argument violation, -(1000+hipError_t) for a HIP
 *     runtime error; never throw, never exit.  mvs_last_error() returns a thread-local message.
 *   - every pointer is a DEVICE pointer owned by the caller (fp32 unless stated, dense, row-major, the
 *     layout given in brackets).  Nothing is allocated or freed; scratch comes in as a caller workspace.
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream); the
 *     device is the caller's current HIP device.  No global mutable state beyond two per-device caches (the CU count, "dynamic LDS
 *     limit already raised for this kernel"): re-entrant across streams, threads and devices.
 *   - Arithmetic.  Data is fp32 in, fp32 out (the reference forces fp32 for the cost volume, mvsformer_model.py:65-78).  Sweeps,
 *     heads, schedulers, filters and the `mvs_conv3d_fwd` / `mvs_deconv3d_fwd` family compute in IEEE fp32 (VALU / fp32 MFMA); the two
 *     entry points that also offer a shortcut form (reciprocal instead of IEEE division, hardware exp2/log2) say so at their `flags`
 *     argument - the Python layer asks for the IEEE form unless told otherwise.  The entry points with `x3` in their name
 *     (`mvs_conv3d_x3_*`, `mvs_deconv3d_x3_*`, `mvs_conv3d_small_*`, `mvs_tail_x3_*`, `mvs_vis_x3_*`: the DEFAULT regularizer / visibility-CNN path of the
 *     Python layer) are fp32-EQUIVALENT, not IEEE fp32 op for op: every operand is split exactly into three bf16 terms
 *     (v = h + m + l) and a product is the six bf16 matrix-core products xh*wh + xh*wm + xm*wh + xh*wl + xl*wh + xm*wm with fp32
 *     accumulation (dropped terms <= 2^-24 |x*w|).  Their contract, tested in tests/test_hip_x3.py against fp64:
 *       * error against the exact result <= 3x the fp32-MFMA kernel's own error on the same operands (+ 2e-7 of the output scale);
 *       * any finite fp32 input is accepted, including |v| above the largest finite bf16 (3.3895e38; h is clamped, m and l carry
 *         the remainder exactly) and inputs scaled by 2^+-100, as long as the exact products and sums stay inside the fp32 range.
 *         ONE exception: the INTERNAL activations of mvs_vis_x3_fwd are split without the clamp (csrc/split3.h split3_pair<false>):
 *         a layer-1/2 activation above 3.3895e38 - reachable only through BatchNorm statistics that scale the entropy map
 *         (range [0, ln D]) by ~1e37 - turns into NaN instead of being carried exactly; its fp32 INPUT obeys the rule above;
 *       * subnormal operands may be flushed to zero by the matrix cores: absolute error <= K * 2^-126 * max|other operand|;
 *       * a non-finite input (+-Inf, NaN) makes exactly the outputs whose receptive field contains it non-finite; the value is NaN
 *         where IEEE fp32 arithmetic would give +-Inf (Inf - Inf in the remainder terms).
 *     The x3 kernels need gfx950: v_mfma_f32_16x16x32_bf16 and up to 72 KB of LDS per block (`mvs_vis_x3_fwd`, raised per device).
 */
#ifndef MVS_HIP_H
#define MVS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVS_OK 0
#define MVS_EINVAL (-22)
#define MVS_ABI_VERSION 40

typedef void* mvs_stream_t;

/* ABI version (MVS_ABI_VERSION) and the last error message of the calling thread. */
int mvs_version(void);
const char* mvs_last_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * Projection prep.  Replaces the per-source-view clone + 2 matmul + torch.inverse + matmul of
 * models/mvsformer_model.py:69-72 and models/warping.py:80-82 (one launch per stage instead of 5 per view,
 * no host sync).
 *   proj [B,V,2,4,4]  proj[b,v,0] = extrinsic 4x4, proj[b,v,1,:3,:3] = stage intrinsic
 *   rt   [B,V-1,12]   for source view v>=1: rows of M[:3,:3] (9 floats) then M[:3,3] (3 floats),
 *                     M = P_v * inverse(P_0), P = [[K*E[:3,:4]],[E[3,:]]]
 * ------------------------------------------------------------------------------------------------------- */
int mvs_proj_prepare(const float* proj, int B, int V, float* rt, mvs_stream_t stream);

/* Same for already composed projections (the op-level warp API, models/warping.py:80-82):
 *   src_proj, ref_proj [B,4,4] -> rt [B,12] */
int mvs_proj_relative(const float* src_proj, const float* ref_proj, int B, float* rt, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Unfused plane-sweep warp = homo_warping_3D_with_mask / homo_warping_3D, models/warping.py:69-109,155-189.
 *   src    [B,C,H,W]
 *   rt     [B,12]            from mvs_proj_relative
 *   depth  [B,D,H,W] if depth_per_pixel else [B,D]
 *   warped [B,C,D,H,W]
 *   mask   [B,D,H,W] uint8 (1 = outside the source frustum), may be NULL
 * ------------------------------------------------------------------------------------------------------- */
int mvs_warp_fwd(const float* src, const float* rt, const float* depth, int depth_per_pixel,
                 int B, int C, int D, int H, int W, float* warped, uint8_t* mask, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused cost-volume build, fusion_type='cnn' (models/mvsformer_model.py:62-105 inline in StageNet.forward).
 * The warped volume, the repeated reference volume, the products and the normalized copies are never
 * materialized.  Two sweeps because the visibility weight needs the full-D entropy plus a 3x3 CNN halo:
 *
 * layout   mvs_nchw_to_nhwc: the sweeps gather CHANNEL-LAST features (one tap = C contiguous floats, fetched as
 *          16-byte loads by adjacent lanes), so the FPN decoder's [B,V,C,H,W] maps are transposed once per stage:
 *          in [N,C,HW] -> out [N,HW,C], C in {8,16,32,64}
 * sweep A  mvs_cv_entropy_fwd: per source view, warp + group correlation -> softmax_d entropy
 *          (mvsformer_model.py:73-79,88-90)
 *   feat    [B,V,H,W,C]  channel-last, view 0 = reference   rt [B,V-1,12]   depth [B,D,H,W]
 *   entropy [B,V-1,H,W]
 * vis CNN  mvs_vis_fwd: ConvBnReLU(1,16)->ConvBnReLU(16,16)->ConvBnReLU(16,8)->Conv2d(8,1,1)->Sigmoid in one
 *          LDS-tiled launch (mvsformer_model.py:37,91; module.py:168-197), eval-mode BN folded to scale/shift
 *   entropy [N,H,W] -> weight [N,H,W];  params: MVS_VIS_PARAM_FLOATS floats, layout in vis_net.hip
 * sweep B  mvs_cv_aggregate_fwd: recompute warp + correlation for all views, accumulate
 *          sum_v w_v*corr_v / (sum_v w_v + 1e-6)  (mvsformer_model.py:101-105) and, if sim_depth != NULL, the
 *          eval-only similarity arg-max depth (mvsformer_model.py:81-85,151-158)
 *   weight  [B,V-1,H,W]   volume [B,G,D,H,W]   sim_depth [B,H,W] or NULL
 * flags bit 0 (both sweeps): 1 = the reference's op order with IEEE divisions (warping.py:90-96) and libm exp/log - what
 * mvsformer_amd passes by default, and what the training forward needs so that it matches mvs_cv_aggregate_bwd's recomputed
 * geometry bit for bit; 0 = sampling coordinates from one reciprocal + Newton step and hardware exp2/log2 in the entropy
 * (~1e-4 px / ~1e-6 entropy away from the reference's arithmetic; measured no faster on MI355X, kept for experiments).
 * Constraints: G == 8, C in {8,16,32,64} (C/G channels per group); sweep A needs (1024/C)*D*4 + 8192 bytes of
 * LDS per block (<= 64 KiB).
 * ------------------------------------------------------------------------------------------------------- */
int mvs_nchw_to_nhwc(const float* in, float* out, int N, int C, int64_t HW, mvs_stream_t stream);
/* the same for up to four independent tensors in ONE launch (host arrays of njobs pointers / shapes: the four stages' feature maps) */
int mvs_nchw_to_nhwc_multi(const float* const* in, float* const* out, const int* N, const int* C, const int64_t* HW, int njobs,
                           mvs_stream_t stream);
int mvs_cv_entropy_fwd(const float* feat, const float* rt, const float* depth,
                       int B, int V, int C, int G, int D, int H, int W, float* entropy, int flags, mvs_stream_t stream);
#define MVS_VIS_PARAM_FLOATS 3689
/* Winograd/MFMA form of the same CNN (vis_net_wino.hip): `prepared` = MVS_VIS_WINO_FLOATS floats written once per
 * parameter set by mvs_vis_wino_prepare(params, prepared) (the two 3x3 layers' weights in F(2x2,3x3) transform domain,
 * laid out per MFMA lane); mvs_vis_wino_fwd computes what mvs_vis_fwd computes, to fp32 rounding. */
#define MVS_VIS_WINO_FLOATS 8192
int mvs_vis_wino_prepare(const float* params, float* prepared, mvs_stream_t stream);
int mvs_vis_wino_fwd(const float* entropy, const float* params, const float* prepared, int N, int H, int W, float* weight,
                     mvs_stream_t stream);
/* Split-form bf16-MFMA version of the same CNN (vis_net_x3.hip; the eval default since round 3): `prepared` = MVS_VIS_X3_BYTES bytes
 * written once per parameter set by mvs_vis_x3_prepare (the two 3x3 layers' weights as three-term bf16 splits, laid out per MFMA
 * lane); fp32 in, fp32 out, fp32-equivalent arithmetic (every fp32 operand = h + m + l exactly, six bf16 MFMAs with fp32 accumulation
 * per K = 32 step) - mvs_vis_x3_fwd computes what mvs_vis_fwd computes, to fp32 rounding. */
#define MVS_VIS_X3_BYTES 33792
int mvs_vis_x3_prepare(const float* params, void* prepared, mvs_stream_t stream);
int mvs_vis_x3_fwd(const float* entropy, const float* params, const void* prepared, int N, int H, int W, float* weight,
                   mvs_stream_t stream);
int mvs_vis_fwd(const float* entropy, const float* params, int N, int H, int W, float* weight, mvs_stream_t stream);
int mvs_cv_aggregate_fwd(const float* feat, const float* rt, const float* depth, const float* weight,
                         int B, int V, int C, int G, int D, int H, int W,
                         float* volume, float* sim_depth, int flags, mvs_stream_t stream);
/* the same + volume16: the volume ALSO as bf16 channel-last [B,D,H,W,G] - the layout the bf16 training regularizer reads (no separate
 * fp32 NCDHW -> bf16 NDHWC pass) */
int mvs_cv_aggregate_fwd_bf16(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V, int C, int G, int D, int H,
                              int W, float* volume, void* volume16, float* sim_depth, int flags, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stored-correlation form of the two sweeps for the coarse cascade stages (C = 32 | 64), same reference lines
 * (models/mvsformer_model.py:73-105,151-158).  There the V-1 per-view correlation volumes [G,D,H,W] fit the 256 MB
 * Infinity Cache (113 / 226 MB at stages 1 / 2 of BASELINE configs[1]), so sweep A' keeps them - and the per-view eval
 * similarity - in `store`, and sweep B' is a pure stream over the store: no second gather sweep.
 *   mvs_cv_corr_store_bytes  size of `store` for a shape (-1 if the shape is not built: C must be 32 or 64, G 8, and D small enough for
 *                     mvs_cv_corr_fwd's per-pixel LDS rows - about D <= 300 at C = 32; the caller then uses the recomputing sweeps)
 *   mvs_cv_corr_fwd   feat [B,V,H,W,C], rt, depth as in mvs_cv_entropy_fwd -> entropy [B,V-1,H,W] (bit-identical to
 *                     mvs_cv_entropy_fwd's) + store; flags as above
 *   mvs_cv_merge_fwd  store + weight [B,V-1,H,W] + depth -> volume [B,G,D,H,W] (bit-identical to mvs_cv_aggregate_fwd's on the
 *                     same flags) and, if sim_depth != NULL, the similarity arg-max depth [B,H,W] (the sums over groups are taken in
 *                     a different order than mvs_cv_aggregate_fwd's: equal up to arg-max ties)
 * `store` is caller-owned scratch, 16-byte aligned, layout private to the pair (pixel-group tiles, cost_volume.hip).
 * ------------------------------------------------------------------------------------------------------- */
int64_t mvs_cv_corr_store_bytes(int B, int V, int C, int G, int D, int H, int W);
int mvs_cv_corr_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int G, int D, int H, int W,
                    float* entropy, void* store, int flags, mvs_stream_t stream);
int mvs_cv_merge_fwd(const void* store, const float* depth, const float* weight, int B, int V, int C, int G, int D, int H, int W,
                     float* volume, float* sim_depth, mvs_stream_t stream);
/* The same pair on a BAND of reference rows, for stages whose whole store would not stay in the Infinity Cache (config-2 stage 2: 254 MB):
 *   mvs_cv_corr_rows_fwd   image rows y0 .. y0+rows-1 of the reference view (features, rt, depth: the whole image as above) -> BAND-LOCAL
 *                          entropy [B,V-1,rows,W] and store (mvs_cv_corr_store_bytes with H = rows)
 *   mvs_cv_merge_rows_fwd  band-local store + weight [B,V-1,rows,W]; band rows r_lo .. r_lo+nrows-1 (the band without the halo rows the
 *                          visibility CNN needed) -> rows y0+r_lo ... of the whole-image volume [B,G,D,H,W] / sim_depth [B,H,W]
 * Per row the arithmetic is that of the whole-image calls: a banded stage is bit-identical to an unbanded one. */
int mvs_cv_corr_rows_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int G, int D, int H, int W, int y0,
                         int rows, float* entropy, void* store, int flags, mvs_stream_t stream);
int mvs_cv_merge_rows_fwd(const void* store, const float* depth, const float* weight, int B, int V, int C, int G, int D, int H, int W,
                          int y0, int rows, int r_lo, int nrows, float* volume, float* sim_depth, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * LDS-tiled form of the same two sweeps (cost_volume_tiled.hip) - the default eval path of StageNet.  Same math as
 * mvs_cv_entropy_fwd / mvs_cv_aggregate_fwd (models/mvsformer_model.py:73-105,151-158 + models/warping.py:84-106), but
 *   feat [B,V,C,H,W]  the FPN decoder's NCHW maps as they are (no transpose launch): a block stages the source texels
 *                     its tile of reference pixels touches into LDS once per (view, plane chunk) and serves every
 *                     bilinear tap from LDS;
 *   flags bit 0       0 = sampling coordinates from one reciprocal + Newton step (deviates from the reference's
 *                     grid_sample coordinates by ~1e-4 px, far inside the 1e-3 depth tolerance), 1 = the reference's
 *                     op order with IEEE divisions (warping.py:90-96), as the direct sweeps compute them;
 *   workspace         sweep B with sim_depth at C >= 32 splits the planes over blocks and merges the per-block
 *                     similarity maxima through mvs_cv_tiled_workspace_bytes(...) bytes (0 for C <= 16); may be NULL
 *                     when that is 0 or sim_depth is NULL;
 *   stats             optional device uint32[2]: [0] += rounds (block x plane pass x view), [1] += rounds whose tap
 *                     bounding box did not fit the LDS tile and took the direct-gather path; NULL to skip.
 * Constraints: G == 8, C in {8,16,32,64}; a view's feature block, a sample's volume and weight block < 2 GiB.
 * ------------------------------------------------------------------------------------------------------- */
int64_t mvs_cv_tiled_workspace_bytes(int B, int V, int C, int D, int H, int W);
int mvs_cv_tiled_entropy_fwd(const float* feat, const float* rt, const float* depth, int B, int V, int C, int G, int D,
                             int H, int W, float* entropy, int flags, uint32_t* stats, mvs_stream_t stream);
int mvs_cv_tiled_aggregate_fwd(const float* feat, const float* rt, const float* depth, const float* weight, int B, int V,
                               int C, int G, int D, int H, int W, float* volume, float* sim_depth, void* workspace,
                               int flags, uint32_t* stats, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * 3-D regularizer layers on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), implicit GEMM with the input
 * tile + weights staged through LDS and a fused epilogue  y = [relu](acc*scale + shift) [+ residual].
 * Replaces Conv3d / Deconv3d (conv -> BatchNorm3d -> ReLU, models/module.py:83-165) and the residual adds
 * of CostRegNet.forward / CostRegNet3D.forward (module.py:495-505,584-594).  Kernel 3x3x3, padding 1.
 *   x [B,Cin,Di,Hi,Wi]   y [B,Cout,Do,Ho,Wo]   residual [B,Cout,Do,Ho,Wo] or NULL (added after the ReLU)
 *   scale, shift [Cout]  (eval BN folded; scale==NULL means 1, shift==NULL means 0)
 *   wpacked: weights re-laid by mvs_conv3d_pack_weights (size from mvs_conv3d_packed_floats)
 *   conv:   stride (sd,shw,shw) in {(1,1,1),(2,2,2),(1,2,2)};  Do = (Di-1)/sd+1 etc.
 *   deconv: ConvTranspose3d stride (sd,2,2), output_padding (sd-1,1,1):  Do = Di*sd, Ho = 2*Hi, Wo = 2*Wi
 * Constraints: Cin % 4 == 0, Cout % 8 == 0, Cout <= 64.
 * ------------------------------------------------------------------------------------------------------- */
/* mode 0: Conv3d weight [Cout,Cin,3,3,3];  mode 1 / 2: ConvTranspose3d weight [Cin,Cout,3,3,3] for depth stride
 * sd = 2 / sd = 1 (the two transposed kernels use different weight images; pass the same mode to both calls);
 * mode 3: w is a Conv3d weight [Cin,Cout,3,3,3] in the roles of the DATA-GRADIENT conv of a stride-1 layer
 * (channels swapped, taps mirrored) — feeds mvs_conv3d_fwd(dY) -> dX.  The other data gradients need no extra mode:
 * strided Conv3d -> mvs_deconv3d_fwd with mode 1/2 of the same weight, ConvTranspose3d -> mvs_conv3d_fwd with mode 0. */
int64_t mvs_conv3d_packed_floats(int Cin, int Cout, int mode);
int mvs_conv3d_pack_weights(const float* w, int Cin, int Cout, int mode, float* wpacked, mvs_stream_t stream);
int mvs_conv3d_fwd(const float* x, const float* wpacked, const float* scale, const float* shift,
                   const float* residual, float* y, int B, int Cin, int Cout, int Di, int Hi, int Wi,
                   int sd, int shw, int relu, mvs_stream_t stream);
int mvs_deconv3d_fwd(const float* x, const float* wpacked, const float* scale, const float* shift,
                     const float* residual, float* y, int B, int Cin, int Cout, int Di, int Hi, int Wi,
                     int sd, int relu, mvs_stream_t stream);

/* CostRegNet.prob: Conv3d(C -> 1, k=3, padding=1, bias=False), models/module.py:493,503.
 *   x [B,C,D,H,W], w [1,C,3,3,3] (PyTorch layout) -> logits [B,D,H,W] */
int mvs_prob3_fwd(const float* x, const float* w, int B, int C, int D, int H, int W, float* logits, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Depth head, depth_type 'ce' (models/mvsformer_model.py:110-125, module.py:597-603), one per-pixel D sweep:
 *   prob_volume_pre = logits                      (or the fused CostRegNet3D.prob 1x1x1 conv, module.py:581,592:
 *                                                  x8 [B,C,D,H,W], w1 [C], b1 [1] device pointers, C = x8_channels)
 *   prob_volume     = softmax_d(prob_volume_pre)
 *   eval:  depth = sum_d softmax_d(pre*tmp) * depth_values      train: depth = depth_values[argmax_d prob_volume]
 *   photometric_confidence = max_d prob_volume
 * Exactly one of (logits) / (x8,w1,b1) is given.  prob_volume_pre is written only in the fused-conv form
 * (may be NULL otherwise).  depth_values [B,D,H,W].
 * ------------------------------------------------------------------------------------------------------- */
int mvs_head_fwd(const float* logits, const float* x8, const float* w1, const float* b1, int x8_channels,
                 const float* depth_values, float tmp, int training, int B, int D, int H, int W,
                 float* prob_volume_pre, float* prob_volume, float* depth, float* conf, mvs_stream_t stream);

/* CostRegNet3D tail in one launch (module.py:573-575,582,591-592): logits = prob( residual + relu(bn(conv11(x))) ) with
 * conv11 = ConvTranspose3d(Cin, 8, stride (1,2,2)) packed with mode 2 and prob = Conv3d(8, 1, 1): prob_w [8], prob_b [1] or
 * NULL.  x [B,Cin,Di,Hi,Wi] (Wi % 4 == 0), residual [B,8,Di,2Hi,2Wi] or NULL, logits [B,1,Di,2Hi,2Wi].  The 8-channel
 * feature volume is never written. */
int mvs_deconv3d_prob1_fwd(const float* x, const float* wpacked, const float* scale, const float* shift, const float* residual,
                           const float* prob_w, const float* prob_b, float* logits, int B, int Cin, int Di, int Hi, int Wi, int relu,
                           mvs_stream_t stream);

/* Stride-1 Conv3d (kernel 3, padding 1) with the (H, W) taps in Winograd F(2x2,3x3) form: same contract as
 * mvs_conv3d_fwd with stride (1,1,1) — y = [relu](conv(x)*scale + shift) [+ residual] — at 2.25x fewer MACs; used for
 * CostRegNet/CostRegNet3D conv2/conv4/conv6 (module.py:475-481,554-560).  Supported: Cin % 4 == 0, Cout in
 * {16,32,48,64}, W % 4 == 0 (query with mvs_conv3d_wino_supported, 1 = yes).  wpacked: Conv3d weight [Cout,Cin,3,3,3]
 * transformed by mvs_conv3d_wino_pack_weights into mvs_conv3d_wino_packed_floats(Cin, Cout) floats. */
int mvs_conv3d_wino_supported(int Cin, int Cout, int D, int H, int W);
int64_t mvs_conv3d_wino_packed_floats(int Cin, int Cout);
int mvs_conv3d_wino_pack_weights(const float* w, int Cin, int Cout, float* wpacked, mvs_stream_t stream);
int mvs_conv3d_wino_fwd(const float* x, const float* wpacked, const float* scale, const float* shift, const float* residual,
                        float* y, int B, int Cin, int Cout, int D, int H, int W, int relu, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Training.  The reference trains StageNet with batch-statistics BatchNorm (module.py:111-117,
 * 153-159,195-197) and autograd through every op; here the forward is  raw conv (mvs_conv3d_fwd / mvs_deconv3d_fwd with
 * scale = shift = NULL, relu = 0) -> mvs_bn_stats -> [all-reduce of sums across ranks = SyncBatchNorm] -> mvs_bn_finalize
 * -> mvs_affine_act, and the backward is mvs_bn_bwd_reduce -> [all-reduce] -> mvs_bn_bwd_apply -> data gradient through
 * the same MFMA conv kernels with re-packed weights -> mvs_conv3d_wgrad.  Tensors are [B,C,N], N = D*H*W or H*W.
 *   mvs_bn_stats:      sums[c] = sum x, sums[C+c] = sum x^2.  Like every per-channel reduction here it runs in two launches -
 *                      one partial row per block into `workspace` (mvs_bn_reduce_workspace_bytes), then a fixed-order sum -
 *                      so the statistics repeat bit-exactly run to run (no float atomics) and `sums` needs no zeroing
 *   mvs_bn_finalize:   mean/var from sums and count -> scale = gamma*invstd, shift = beta - mean*scale, mean, invstd;
 *                      running stats updated with momentum (unbiased variance), as nn.BatchNorm does; NULL to skip.
 *                      `count` is the per-channel element count as a host value; SyncBatchNorm (train.py:138-139) passes
 *                      `count_dev` instead: two DEVICE floats {n / 4096, n % 4096} that were summed over the ranks by the same
 *                      all-reduce as `sums` (exact up to 2^36 elements; the host never reads the global count back)
 *   mvs_affine_act:    y = [relu](x*scale[c] + shift[c]) [+ residual]
 *   mvs_bn_bwd_reduce: g = dy*[x*scale+shift > 0 or !relu]; sums[c] = sum g, sums[C+c] = sum g*xhat (same workspace)
 *   mvs_bn_bwd_apply:  dx = gamma*invstd*(g - sums[c]/count - xhat*sums[C+c]/count);  dgamma = sums[C+c], dbeta = sums[c]
 *   mvs_conv3d_wgrad:  dW[a][b][k] += sum_{batch,p} A[a,p] * Bt[b, p*s - 1 + k]  (dW zeroed by the caller); Conv3d:
 *                      A = dY, Bt = X, dW = [Cout,Cin,27]; ConvTranspose3d: A = X, Bt = dY, dW = [Cin,Cout,27];
 *                      (Dp,Hp,Wp) is the grid the stride divides (conv output / deconv input), (Db,Hb,Wb) the other
 *   mvs_cv_aggregate_bwd: gradient of mvs_cv_aggregate_fwd w.r.t. the channel-last features (dfeat [B,V,H,W,C], zeroed by
 *                      the caller; source views are accumulated with fp32 atomics) and the visibility weights
 *   mvs_softmax_bwd, mvs_prob1_bwd (dwb[C+1] zeroed by the caller: dW then dbias), mvs_sigmoid_fwd/bwd, mvs_nhwc_to_nchw
 * ------------------------------------------------------------------------------------------------------- */
int64_t mvs_bn_reduce_workspace_bytes(int B, int C, int64_t N);
int mvs_bn_stats(const float* x, int B, int C, int64_t N, float* sums, void* workspace, mvs_stream_t stream);
int mvs_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, double count, const float* count_dev, int C, float* scale, float* shift, float* mean,
                    float* invstd, mvs_stream_t stream);
/* `groups` BatchNorm calls of one module laid side by side as channels g*C + c (sums [2*groups*C], gamma/beta/running [C],
 * scale/shift/mean/invstd [groups*C]); running statistics receive the groups' updates in order, as separate calls would. */
int mvs_bn_finalize_grouped(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, double count, const float* count_dev, int C, int groups, float* scale,
                            float* shift, float* mean, float* invstd, mvs_stream_t stream);
int mvs_affine_act(const float* x, const float* scale, const float* shift, const float* residual, int relu, int B, int C,
                   int64_t N, float* y, mvs_stream_t stream);
int mvs_bn_bwd_reduce(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                      const float* invstd, int relu, int B, int C, int64_t N, float* sums, void* workspace, mvs_stream_t stream);
int mvs_bn_bwd_apply(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                     const float* invstd, const float* gamma, const float* sums, double count, const float* count_dev, int relu,
                     int B, int C, int64_t N, float* dx, mvs_stream_t stream);
int mvs_conv3d_wgrad(const float* A, const float* Bt, float* dW, int nbatch, int CA, int CB, int Dp, int Hp, int Wp, int Db,
                     int Hb, int Wb, int sd, int shw, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * bf16 regularizer for training under autocast (BASELINE configs[2]; the reference wraps the model in torch.cuda.amp.autocast,
 * trainer/mvsformer_trainer.py:43-45,104-106, so Conv3d / ConvTranspose3d of models/module.py:83-165,469-594 run in half
 * precision with fp32 accumulation while the cost volume stays fp32, models/mvsformer_model.py:65,68,78).  Activations are bf16
 * CHANNEL-LAST [B,D,H,W,C] (void* = device bf16), parameters stay fp32 and are packed to bf16 per call; every kernel
 * accumulates in fp32 on v_mfma_f32_16x16x32_bf16.  Channels: 8, 16, 32 or 64.
 *   mvs_bf16_packed_elems / mvs_bf16_pack_weights: w = [d0][d1][27] fp32 -> per-lane MFMA A fragments of a Cin -> Cout map;
 *       src 0: w[cout][cin][tap] (Conv3d forward, ConvTranspose3d data gradient), 1: w[cin][cout][tap] (ConvTranspose3d forward,
 *       strided Conv3d data gradient), 2: w[cin][cout][26 - tap] (stride-1 Conv3d data gradient)
 *   mvs_bf16_conv3d: gather 0 = Conv3d k3 p1 stride (sd,shw,shw); gather 1 = ConvTranspose3d k3 p1 output_padding stride-1;
 *       optional fused epilogue y = [relu](acc*scale + shift) [+ residual] (NULL scale/shift: raw output for batch-stat BatchNorm)
 *   mvs_bf16_conv3d_wgrad: dW[a][b][tap] = sum A[p][a] * Bt[p*s - 1 + k][b] (dW fp32, overwritten; same operand roles as
 *       mvs_conv3d_wgrad); row chunks write partial slabs into `workspace` (mvs_bf16_conv3d_wgrad_workspace_bytes) that a second
 *       launch sums - no atomics, deterministic
 *   mvs_bf16_from_f32_ncdhw / mvs_bf16_to_f32_ncdhw: fp32 [B,C,N] <-> bf16 [B,N,C]
 *   mvs_bf16_bn_stats / mvs_bf16_affine_act / mvs_bf16_bn_bwd_reduce / mvs_bf16_bn_bwd_apply: channel-last bf16 twins of
 *       mvs_bn_stats / mvs_affine_act / mvs_bn_bwd_reduce / mvs_bn_bwd_apply over R = B*D*H*W rows (fp32 statistics;
 *       mvs_bn_finalize is shared; partial rows in a workspace of mvs_bf16_bn_reduce_workspace_bytes, fixed-order sums).
 *       groups > 1 = grouped BatchNorm over the batch dimension (the visibility CNN applied once per source view,
 *       mvsformer_model.py:91, as ONE batch): sample n = rows [n*rows_per_sample, (n+1)*rows_per_sample) belongs to group
 *       n % groups; statistics and every per-channel array are per (group, channel), laid out [groups*C] (sums: [sum | sum of
 *       squares], each groups*C) - what mvs_bn_finalize_grouped consumes and produces.  groups = 1: rows_per_sample is ignored.
 * ------------------------------------------------------------------------------------------------------- */
int64_t mvs_bf16_packed_elems(int Cin, int Cout);
int mvs_bf16_pack_weights(const float* w, int d0, int d1, int src, int Cout, int Cin, void* wpacked, mvs_stream_t stream);
/* two layouts of one weight (the forward's and the data gradient's) in one launch */
int mvs_bf16_pack_weights2(const float* w, int d0, int d1, int srcA, int CoutA, int CinA, void* packedA, int srcB, int CoutB, int CinB,
                           void* packedB, mvs_stream_t stream);
int mvs_bf16_conv3d(const void* x, const void* wpacked, const float* scale, const float* shift, const void* residual, void* y,
                    int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd, int shw, int relu, mvs_stream_t stream);
/* raw convolution + the batch statistics of its bf16-rounded output in the same pass (what mvs_bf16_bn_stats(y, groups) returns):
 * sums [2*groups*Cout]; workspace = mvs_bf16_conv3d_stats_workspace_bytes(B, Cout, Do, Ho, Wo) of the OUTPUT grid */
int64_t mvs_bf16_conv3d_stats_workspace_bytes(int B, int Cout, int Do, int Ho, int Wo);
int mvs_bf16_conv3d_stats(const void* x, const void* wpacked, void* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd,
                          int shw, int groups, float* sums, void* workspace, mvs_stream_t stream);
int64_t mvs_bf16_conv3d_wgrad_workspace_bytes(int nbatch, int CA, int CB, int Dp, int Hp, int Wp);
int mvs_bf16_conv3d_wgrad(const void* A, const void* Bt, float* dW, void* workspace, int nbatch, int CA, int CB, int Dp, int Hp, int Wp,
                          int Db, int Hb, int Wb, int sd, int shw, mvs_stream_t stream);
int mvs_bf16_from_f32_ncdhw(const float* in, void* out, int B, int C, int64_t N, mvs_stream_t stream);
int mvs_bf16_to_f32_ncdhw(const void* in, float* out, int B, int C, int64_t N, mvs_stream_t stream);
int64_t mvs_bf16_bn_reduce_workspace_bytes(int C, int64_t R, int groups, int64_t rows_per_sample);
int mvs_bf16_bn_stats(const void* x, int C, int64_t R, int groups, int64_t rows_per_sample, float* sums, void* workspace,
                      mvs_stream_t stream);
int mvs_bf16_affine_act(const void* x, const float* scale, const float* shift, const void* residual, int relu, int C, int64_t R,
                        int groups, int64_t rows_per_sample, void* y, mvs_stream_t stream);
/* training-mode BatchNorm forward in one call / three launches when no cross-rank reduction sits between statistics and finalize:
 * = mvs_bf16_bn_stats + mvs_bn_finalize(_grouped) + mvs_bf16_affine_act; stats4 = [scale | shift | mean | invstd | gamma]: FIVE rows of
 * groups*C floats (the fifth = the affine weight per (group, channel), 1 without one: ABI 25) */
int mvs_bf16_bn_train_fwd(const void* x, const void* residual, int relu, int C, int64_t R, int groups, int64_t rows_per_sample,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          int64_t* num_batches_tracked /* NULL, or nn.BatchNorm's counter: += groups */, float* stats4, void* y,
                          void* workspace, mvs_stream_t stream);
int mvs_bf16_bn_bwd_reduce(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                           const float* invstd, int relu, int C, int64_t R, int groups, int64_t rows_per_sample, float* sums,
                           void* workspace, mvs_stream_t stream);
int mvs_bf16_bn_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                          const float* invstd, const float* gamma, const float* sums, double count, const float* count_dev,
                          int relu, int C, int64_t R, int groups, int64_t rows_per_sample, void* dx, mvs_stream_t stream);
/* the same + dgb [2*C] = [dbeta | dgamma] of a GROUPED BatchNorm's shared parameters (the groups' sums added in group order) */
int mvs_bf16_bn_bwd_apply_dgb(const void* dy, const void* x, const float* scale, const float* shift, const float* mean, const float* invstd,
                              const float* gamma, const float* sums, double count, const float* count_dev, int relu, int C, int64_t R,
                              int groups, int64_t rows_per_sample, void* dx, float* dgb, mvs_stream_t stream);

/* Round-5 fused forms of the bf16 training layer (fewer graph nodes per step; reference layer: models/module.py:83-165 under
 * trainer/mvsformer_trainer.py:104-106 autocast).
 *   mvs_bf16_conv3d_bn_fwd: y = raw conv(x) (kept for the backward), z = [relu](BatchNorm_train(y)) [+ residual]; the batch statistics of
 *       the bf16-rounded y are taken in the convolution's epilogue as one row per BLOCK (workspace =
 *       mvs_bf16_conv3d_bn_fwd_workspace_bytes of the OUTPUT grid), a one-block-per-channel kernel adds the rows in a fixed order and
 *       writes stats4 = [scale | shift | mean | invstd | gamma] (FIVE rows of groups*Cout; the fifth is the affine weight replicated per
 *       group, 1 without one - the operand of the backward's apply kernel) + the running statistics, a third launch normalizes.
 *       groups > 1: sample b belongs to group b % groups (the visibility CNN, one call per source view in the reference). */
int64_t mvs_bf16_conv3d_bn_fwd_workspace_bytes(int B, int Cout, int Do, int Ho, int Wo);
int mvs_bf16_conv3d_bn_fwd(const void* x, const void* wpacked, void* y, void* z, const void* residual, int relu, int B, int Cin, int Cout,
                           int Di, int Hi, int Wi, int gather, int sd, int shw, int taps, int groups, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, int64_t* num_batches_tracked,
                           float* stats4, void* workspace, mvs_stream_t stream);
/* mvs_bf16_conv3d_bnbwd: a raw convolution whose output is the gradient dz arriving at a BatchNorm(+ReLU) layer (the data gradient of the
 * layer AFTER it), with that BatchNorm's backward sums [sum g | sum g*xhat] (g = dz * relu'(bn_y*scale + shift)) taken in the epilogue:
 * what mvs_bf16_bn_bwd_reduce(dz, bn_y, ...) returns, without its pass over dz and bn_y.  bn4 = the forward's stats4; workspace as
 * mvs_bf16_conv3d_bn_fwd_workspace_bytes of the output grid.  addend (optional, y's shape, bf16): a second gradient of the same tensor (the
 * tensor also fed a skip connection) added before the rounding and the sums: y = the TOTAL gradient. */
int mvs_bf16_conv3d_bnbwd(const void* x, const void* wpacked, void* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd,
                          int shw, int taps, const void* bn_y, const float* bn4, int relu, int groups, const void* addend, float* sums,
                          void* workspace, mvs_stream_t stream);
/* The weight gradients of SEVERAL layers in one call: one grid per kernel instance (all jobs of it side by side) + one fixed-order reduce for
 * all of them.  A layer alone is a chain of short round trips on a fraction of the chip; a training step's ~50 of them in a row were a
 * quarter of the step.  jobs = HOST array; per job dW [CA][CBout][taps] = sum A[p][a] * Bt[p*s-1+k][b] exactly as mvs_bf16_conv3d_wgrad_taps
 * (same kernels, same slab order: bit-identical to the layer-by-layer calls).  workspace >= mvs_bf16_wgrad_group_workspace_bytes(jobs). */
typedef struct MvsWgradJob {
    const void* A;            /* [nbatch,Dp,Hp,Wp,CA] bf16 channel-last: the operand on the grid the stride divides */
    const void* Bt;           /* [nbatch,Db,Hb,Wb,CB] bf16 channel-last */
    float* dW;                /* [CA][CBout][taps] fp32 */
    int nbatch, CA, CB, CBout, Dp, Hp, Wp, Db, Hb, Wb, sd, shw, taps, reserved;
} MvsWgradJob;
int64_t mvs_bf16_wgrad_group_workspace_bytes(const MvsWgradJob* jobs, int njobs);
int mvs_bf16_wgrad_group(const MvsWgradJob* jobs, int njobs, void* workspace, int64_t workspace_bytes, mvs_stream_t stream);
/* 2-D kernels on the same machinery (`taps` = 27 or 9; 9 = only the centre depth tap exists: the visibility CNN's Conv2d layers run as
 * D = 1 volumes at a third of the matrix work of a zero-embedded 3x3x3 kernel, forward, data gradient and weight gradient):
 *   mvs_bf16_packed_elems_taps / mvs_bf16_pack_table_*: w = [d0][d1][taps]; rows / channels of the Cin -> Cout map beyond d0 / d1 pack as
 *       zeros (the CNN's 1-channel input as 8, CostRegNet's 8 -> 1 `prob` as 8 -> 8).  A TABLE of jobs - every layout a stage's training
 *       step needs - is filled on the host entry by entry, copied to the device once, and packed by ONE launch per step.
 *   mvs_bf16_conv3d_taps: raw convolution (no epilogue); taps = 9 needs gather 0, stride 1, <= 16 channels.
 *   mvs_bf16_conv3d_wgrad_taps: dW [CA][CBout][taps], CBout <= CB drops Bt's padding channels. */
int64_t mvs_bf16_packed_elems_taps(int Cin, int Cout, int taps);
int64_t mvs_bf16_pack_table_bytes(int njobs);
int mvs_bf16_pack_table_fill(void* host_table, int njobs, int index, const float* w, int d0, int d1, int src, int Cout, int Cin, int taps,
                             void* wpacked);
int mvs_bf16_pack_table_run(const void* dev_table, int njobs, int total_blocks, mvs_stream_t stream);
int mvs_bf16_conv3d_taps(const void* x, const void* wpacked, void* y, int B, int Cin, int Cout, int Di, int Hi, int Wi, int gather, int sd,
                         int shw, int taps, mvs_stream_t stream);
int mvs_bf16_conv3d_wgrad_taps(const void* A, const void* Bt, float* dW, void* workspace, int nbatch, int CA, int CB, int CBout, int Dp, int Hp,
                               int Wp, int Db, int Hb, int Wb, int sd, int shw, int taps, mvs_stream_t stream);
/* 1x1x1 heads on bf16 channel-last activations (csrc/bf16_head.hip): out[v] = act(sum_c w[c]*x[v][c] + bias), x = bf16 [N][8], out fp32
 * [N], act = sigmoid or identity; w = NULL selects channel 0 (no parameters).  Backward: dx bf16 [N][8], dwb [9] = [dw | dbias] by block
 * rows + a fixed-order reduce (workspace = mvs_bf16_head_bwd_workspace_bytes(N)); y = the forward's output when it applied the sigmoid. */
/* fp32 [N] -> bf16 [N][8] with the value in channel 0 and zeros in channels 1..7 (the entropy map enters the visibility CNN) */
int mvs_bf16_embed_ch0(const float* in, void* out, int64_t N, mvs_stream_t stream);
int mvs_bf16_head_fwd(const void* x, const float* w, const float* bias, int sigmoid, int64_t N, float* out, mvs_stream_t stream);
int64_t mvs_bf16_head_bwd_workspace_bytes(int64_t N);
int mvs_bf16_head_bwd(const void* x, const float* w, const float* y, const float* dout, int64_t N, void* dx, float* dwb, void* workspace,
                      mvs_stream_t stream);
int mvs_cv_aggregate_bwd(const float* feat, const float* rt, const float* depth, const float* weight, const float* volume,
                         const float* gvolume, int B, int V, int C, int G, int D, int H, int W, float* dfeat, float* dweight,
                         mvs_stream_t stream);
/* The same gradients with the bilinear scatter accumulated in LDS: a block owns a 16 x 8 tile of reference pixels and one channel
 * octet, adds the taps of all D planes into a (1 << wx_log2) x wy texel window of the source view's gradient (ds_add_f32) and flushes
 * the touched texels with one global atomic per value; taps outside the window go straight to global atomics, so the result is
 * independent of the window.  gip_part [C/8][B][V-1][H][W]: d(loss)/d(vis weight) per channel octet - the caller adds the octets.
 * stats: NULL, or 2 device counters (diagnostics): [0] += taps scattered, [1] += taps that missed the window. */
int mvs_cv_aggregate_bwd_lds(const float* feat, const float* rt, const float* depth, const float* weight, const float* volume,
                             const float* gvolume, int B, int V, int C, int G, int D, int H, int W, float* dfeat, float* gip_part,
                             int wx_log2, int wy, unsigned* stats, mvs_stream_t stream);
/* Same contract, no atomics in the common case: a WAVEFRONT owns an 8 x 4 pixel tile and a private LDS window; lanes that target the
 * same cell in one instruction are serialized by a tag byte (owner election), the add is a plain LDS read-modify-write; planes in
 * chunks of 8, window re-centred and flushed (global atomics) per (view, chunk); taps outside the window use global atomics. */
int mvs_cv_aggregate_bwd_own(const float* feat, const float* rt, const float* depth, const float* weight, const float* volume,
                             const float* gvolume, int B, int V, int C, int G, int D, int H, int W, float* dfeat, float* gip_part,
                             int wx_log2, int wy, unsigned* stats, mvs_stream_t stream);
int mvs_softmax_bwd(const float* p, const float* dp, int B, int D, int64_t HW, float* dpre, mvs_stream_t stream);
int mvs_prob1_bwd(const float* x, const float* w, const float* dlogits, int B, int C, int64_t N, float* dx, float* dwb,
                  mvs_stream_t stream);
int mvs_sigmoid_fwd(const float* x, int64_t n, float* y, mvs_stream_t stream);
int mvs_sigmoid_bwd(const float* y, const float* dy, int64_t n, float* dx, mvs_stream_t stream);
/* out = a * b elementwise (models/module.py:464 x1 * x2 of AttentionFusionSimple in training; backward = the same call twice) */
int mvs_ewise_mul(const float* a, const float* b, int64_t n, float* out, mvs_stream_t stream);
int mvs_nhwc_to_nchw(const float* in, float* out, int N, int C, int64_t HW, mvs_stream_t stream);

/* Stand-alone heads for callers that use the ops directly.
 *   mvs_depth_regression: module.py:597-603, depth = sum_d p*depth_values; depth_values [B,D,H,W] or [B,D]
 *   mvs_conf_regression:  module.py:606-619, sum of the n probabilities around floor(sum_d p*d)
 *   mvs_mixup_head:       the depth_type 'mixup_ce' head, models/mvsformer_model.py:126-136: best adjacent pair of p [B,D,H,W]
 *                         (first maximum of p[d]+p[d+1]) -> conf = that sum, depth = the two hypotheses of depth_values
 *                         [B,D,H,W] mixed by the pair probabilities renormalised with +1e-7
 *   mvs_prob1_fwd:        CostRegNet3D.prob alone (module.py:581,592): 1x1x1 conv C->1, x [B,C,N] -> logits [B,N],
 *                         w [C], bias [1] or NULL */
int mvs_depth_regression(const float* p, const float* depth_values, int depth_per_pixel, int B, int D, int H, int W,
                         float* depth, mvs_stream_t stream);
int mvs_conf_regression(const float* p, int n, int B, int D, int H, int W, float* conf, mvs_stream_t stream);
int mvs_mixup_head(const float* p, const float* depth_values, int B, int D, int H, int W, float* depth, float* conf,
                   mvs_stream_t stream);
int mvs_prob1_fwd(const float* x, const float* w, const float* bias, int B, int C, int64_t N, float* logits,
                  mvs_stream_t stream);

/* Hypothesis schedulers, models/module.py:633-653.
 *   init:     depth_range [B,N] (only [:,0] and [:,N-1] are read) -> hyp [B,D,H,W], index 0 = far
 *   schedule: prev_depth [B,H/2,W/2], prev_hyp [B,Dp,H/2,W/2] (planes 1 and 2 are read) -> hyp [B,D,H,W]
 *             (trilinear x2 upsampling, align_corners=True, of the inverse-depth samples, then reciprocal) */
int mvs_init_inverse_range(const float* depth_range, int N, int B, int D, int H, int W, float* hyp, mvs_stream_t stream);
int mvs_schedule_inverse_range(const float* prev_depth, const float* prev_hyp, int Dp, float split_itv,
                               int B, int D, int H, int W, float* hyp, mvs_stream_t stream);

/* Nearest-neighbour upsample + accumulate of the per-stage confidences (mvsformer_model.py:297-301):
 *   acc [B,Hf,Wf] += nearest(conf [B,H,W]) * weight */
int mvs_conf_accumulate(const float* conf, int B, int H, int W, float* acc, int Hf, int Wf, float weight, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Row before the path: FPNDecoder.forward, models/module.py:242-270, eval-mode BatchNorm.  Inputs and the
 * top-down intermediates are NCHW like the reference's tensors; the four feature maps come out CHANNEL-LAST [N,H,W,C], the layout
 * the sweeps read (mvs_cv_*: feat [B,V,H,W,C] with N = B*V) - no nchw_to_nhwc pass.  scale/shift fold BatchNorm2d and the conv bias:
 * scale = gamma / sqrt(var + eps), shift = beta + (bias - mean) * scale; the activation is Swish (module.py:200-206).
 *   pack:   w [Cout,64,3,3] (out_k.0.weight, Cout = 8 | 16 | 32) -> packed, mvs_fpn_packed_floats(Cout) floats
 *   out0:   x = conv31 [N,64,h,w], w [64,64] (out0.0.weight) -> out [N,h,w,64]
 *   level:  intra_prev [N,64,h,w], lateral [N,Ck,2h,2w], w_inner_p [32,Ck,2] (inner_k.weight [64,Ck] regrouped by output-channel PAIR:
 *           w_inner_p[q][j][e] = weight[2q+e][j], one scalar load per pair for the packed fp32 pipe) + b_inner [64], packed out_k weights ->
 *           intra_out [N,64,2h,2w] (NULL for the last level: it is only ever consumed inside this kernel) and out [N,2h,2w,Ck] */
/* ---------------------------------------------------------------------------------------------------------
 * Regularizer convolutions in THREE-TERM BF16 SPLIT form (conv3d_x3.hip): same contract as mvs_conv3d_fwd (fp32 NCDHW in and out,
 * y = [relu](conv3d(x, w, padding 1) * scale[co] + shift[co]) [+ residual], models/module.py:83-123), but the contraction runs on the
 * bf16 matrix cores with every fp32 operand written EXACTLY as h + m + l (three bf16) and the six products xh*wh, xh*wm, xm*wh, xh*wl,
 * xl*wh, xm*wm accumulated in fp32: what is dropped is <= 2^-24 of a product, i.e. the result is as close to the exact convolution as an
 * fp32 fma chain (tests/test_hip_x3.py) at 6/16 of the fp32 MFMA's matrix time.
 *   supported:  1 if (Cin, Cout, depth stride sd, H/W stride shw) is built: sd = 1; shw = 1 with Cin, Cout in {16, 32, 64}; shw = 2 with
 *               (Cin, Cout) = (8, 16) or Cin in {16, 32}, Cout in {16, 32, 64}  - the six Conv3d layers of CostRegNet3D
 *   pack:       w [Cout,Cin,3,3,3] fp32 -> wpacked, mvs_conv3d_x3_packed_bytes(...) bytes (pre-split, per-lane fragments)
 *   fwd:        x [B,Cin,D,H,W] -> y [B,Cout,D,(H-1)/shw+1,(W-1)/shw+1]; residual (optional) has y's shape
 * ------------------------------------------------------------------------------------------------------- */
int mvs_conv3d_x3_supported(int Cin, int Cout, int sd, int shw);
int64_t mvs_conv3d_x3_packed_bytes(int Cin, int Cout, int sd, int shw);
int mvs_conv3d_x3_pack_weights(const float* w, int Cin, int Cout, int sd, int shw, void* wpacked, mvs_stream_t stream);
int mvs_conv3d_x3_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual, float* y,
                      int B, int Cin, int Cout, int D, int H, int W, int sd, int shw, int relu, mvs_stream_t stream);
/* Transposed twin (ConvTranspose3d k = 3, stride (1,2,2), padding 1, output_padding (0,1,1): conv7 / conv9 / conv11 of CostRegNet3D,
 * models/module.py:562-575), same split form: w [Cin,Cout,3,3,3], x [B,Cin,D,H,W] -> y [B,Cout,D,2H,2W]; built for sd = 1, Cin in
 * {16,32,64}, Cout in {8,16,32}; W even. */
int mvs_deconv3d_x3_supported(int Cin, int Cout, int sd);
int64_t mvs_deconv3d_x3_packed_bytes(int Cin, int Cout, int sd);
int mvs_deconv3d_x3_pack_weights(const float* w, int Cin, int Cout, int sd, void* wpacked, mvs_stream_t stream);
int mvs_deconv3d_x3_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual, float* y,
                        int B, int Cin, int Cout, int D, int H, int W, int sd, int relu, mvs_stream_t stream);

/* Small-volume form of the same layers, split form (csrc/conv3d_x3_small.hip): one wavefront per 16 output voxels x 16 output channels gathers its
 * operands straight from global memory - no staging rounds, no barriers - for CostRegNet's inner layers at the coarse cascade stages
 * (models/module.py:469-505: volumes of a few thousand voxels, where the tiled kernels are latency chains on a handful of CUs).
 *   transposed = 0: Conv3d(k 3, padding 1, stride (s,s,s)), s in {1, 2}; w [Cout,Cin,3,3,3]; x [B,Cin,D,H,W] -> y [B,Cout,(D-1)/s+1,(H-1)/s+1,(W-1)/s+1]
 *   transposed = 1: ConvTranspose3d(k 3, stride 2, padding 1, output_padding 1); w [Cin,Cout,3,3,3]; x [B,Cin,D,H,W] -> y [B,Cout,2D,2H,2W]
 *   y = [relu](conv * scale + shift) [+ residual]; Cin in {8,16,32,64}, Cout a multiple of 8 up to 64. */
int mvs_conv3d_small_supported(int Cin, int Cout, int stride, int transposed);
int64_t mvs_conv3d_small_packed_bytes(int Cin, int Cout, int stride, int transposed);
int mvs_conv3d_small_pack_weights(const float* w, int Cin, int Cout, int stride, int transposed, void* wpacked, mvs_stream_t stream);
int mvs_conv3d_small_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual, float* y, int B,
                         int Cin, int Cout, int D, int H, int W, int stride, int transposed, int relu, mvs_stream_t stream);

/* CostRegNet3D's tail in one launch, split form (csrc/tail_x3.hip): logits = prob(residual + relu(bn(conv11(x)))) with conv11 =
 * ConvTranspose3d(16, 8, 3, stride (1,2,2), padding 1, output_padding (0,1,1), bias=False) and prob = Conv3d(8, 1, 1) - models/module.py:
 * 575-576 (conv11), 582 (prob), 590-592 (`x = x + self.conv11(x)` / `self.prob(x)`).  Same contract as mvs_deconv3d_prob1_fwd (which
 * computes it on the fp32 matrix cores) for Cin = 16; the 8-channel feature volume is never written.
 *   pack: w [16,8,3,3,3] -> wpacked, mvs_tail_x3_packed_bytes() bytes
 *   x [B,16,D,H,W] (conv9's output), scale/shift [8] or NULL, residual [B,8,D,2H,2W] or NULL, prob_w [8], prob_b [1] or NULL
 *   -> logits [B,D,2H,2W] */
int64_t mvs_tail_x3_packed_bytes(void);
int mvs_tail_x3_pack_weights(const float* w, void* wpacked, mvs_stream_t stream);
int mvs_tail_x3_fwd(const float* x, const void* wpacked, const float* scale, const float* shift, const float* residual, const float* prob_w,
                    const float* prob_b, float* logits, int B, int D, int H, int W, int relu, mvs_stream_t stream);

/* FPNEncoder layers, models/module.py:40-73,208-240: y = leaky_relu(BatchNorm2d_eval(conv2d(x, w, stride, padding = K/2)), slope), NCHW.
 * Built for the encoder's eight layer shapes (Cin,Cout,K,stride) = (3,8,7,1) (8,8,5,1) (8,16,5,2) (16,16,3,1) (16,32,5,2) (32,32,3,1)
 * (32,64,3,2) (64,64,3,1); anything else returns MVS_EINVAL.  scale = gamma / sqrt(var + eps), shift = beta - mean * scale.
 *   pack: w [Cout,Cin,K,K] -> packed, mvs_conv2d_packed_floats(Cin, Cout, K) floats
 *   x [N,Cin,H,W] -> y [N,Cout,(H-1)/stride+1,(W-1)/stride+1] */
int64_t mvs_conv2d_packed_floats(int Cin, int Cout, int K);
int mvs_conv2d_pack_weights(const float* w, int Cin, int Cout, int K, float* packed, mvs_stream_t stream);
int mvs_conv2d_bn_lrelu(const float* x, const float* packed, const float* scale, const float* shift, int N, int Cin, int Cout, int K,
                        int stride, int H, int W, float slope, float* y, mvs_stream_t stream);
/* The same level with the channel contraction IN FRONT of the upsampling (csrc/fpn_cp.hip): P_tap[q] = w3[:, :, tap] . intra_prev[q] on the bf16
 * matrix cores at the coarse resolution (split form), blended at the fine level with ATen's per-pixel bilinear weights - the same linear map,
 * a quarter of the matrix work and no per-pixel split.  Same operands (w3, wc, scale -> prepared; shift, border) and contract as
 * mvs_fpn_level_x3; fp32-equivalent. */
int64_t mvs_fpn_level_cp_prepared_bytes(int Ck);
int mvs_fpn_level_cp_prepare(const float* w3, const float* wc, const float* scale, int Ck, void* prepared, mvs_stream_t stream);
int mvs_fpn_level_cp(const float* intra_prev, const float* lateral, const void* prepared, const float* shift, const float* border, int N,
                     int Ck, int h, int w, float* out, mvs_stream_t stream);
/* conv00 / conv01 (the two full-resolution layers: (Cin,Cout,K,stride) = (3,8,7,1), (8,8,5,1)) in three-term bf16 split form (csrc/conv2d_x3.hip):
 * same contract as mvs_conv2d_bn_lrelu - fp32 NCHW in and out, fp32-equivalent - with the BatchNorm scale folded into the pre-split weights.
 *   prepare: w [8,Cin,K,K], scale [8] -> prepared, mvs_conv2d_x3_prepared_bytes(Cin, 8, K) bytes
 *   x [N,Cin,H,W], shift [8] -> y [N,8,H,W] = leaky_relu(conv(x, w) * scale + shift, slope) */
int mvs_conv2d_x3_supported(int Cin, int Cout, int K, int stride);
int64_t mvs_conv2d_x3_prepared_bytes(int Cin, int Cout, int K);
int mvs_conv2d_x3_prepare(const float* w, const float* scale, int Cin, int Cout, int K, void* prepared, mvs_stream_t stream);
int mvs_conv2d_x3_bn_lrelu(const float* x, const void* prepared, const float* shift, int N, int Cin, int Cout, int K, int stride, int H, int W,
                           float slope, float* y, mvs_stream_t stream);
/* the same with the layouts chosen: x_nhwc = 1 reads x as [N,H,W,8] (Cin = 8: conv01 after a channel-last conv00); y [N,8,H,W] and / or
 * y_nhwc [N,H,W,8] are written (either may be NULL): conv00 -> conv01 hand over channel-last (16-byte stores and loads instead of scattered
 * dwords), conv01's y_nhwc is the lateral mvs_fpn_level_cp reads */
int mvs_conv2d_x3_bn_lrelu_layout(const float* x, int x_nhwc, const void* prepared, const float* shift, int N, int Cin, int Cout, int K,
                                  int stride, int H, int W, float slope, float* y, float* y_nhwc, mvs_stream_t stream);
/* The encoder's layers below full resolution in three-term bf16 split form (csrc/conv2d_x3s.hip): the six 3x3 stride-1 layers (Cin = Cout = 16 | 32 |
 * 64) and the three stride-2 layers ((Cin,Cout,K) = (8,16,5), (16,32,5), (32,64,3)); same contract as mvs_conv2d_bn_lrelu (fp32 NCHW in and out,
 * fp32-equivalent), the BatchNorm scale folded into the pre-split weights.  x_nhwc = 1: x is [N,H,W,8] (the 8-channel stride-2 layer reading
 * conv01's channel-last companion).
 *   prepare: w [Cout,Cin,K,K], scale [Cout] -> prepared, mvs_conv2d_x3s_prepared_bytes(Cin, Cout, K, stride) bytes */
int mvs_conv2d_x3s_supported(int Cin, int Cout, int K, int stride);
int64_t mvs_conv2d_x3s_prepared_bytes(int Cin, int Cout, int K, int stride);
int mvs_conv2d_x3s_prepare(const float* w, const float* scale, int Cin, int Cout, int K, int stride, void* prepared, mvs_stream_t stream);
int mvs_conv2d_x3s_bn_lrelu(const float* x, int x_nhwc, const void* prepared, const float* shift, int N, int Cin, int Cout, int K, int stride, int H,
                            int W, float slope, float* y, mvs_stream_t stream);
int64_t mvs_fpn_packed_floats(int Cout);
int mvs_fpn_pack_weights(const float* w, int Cout, float* packed, mvs_stream_t stream);
int mvs_fpn_out0(const float* x, const float* w, const float* scale, const float* shift, int N, int h, int wd, float* out,
                 mvs_stream_t stream);
int mvs_fpn_level(const float* intra_prev, const float* lateral, const float* w_inner_p, const float* b_inner, const float* w_packed,
                  const float* scale, const float* shift, int N, int Ck, int h, int w, float* intra_out, float* out,
                  mvs_stream_t stream);
/* the same with the layout of intra_out chosen: intra_nhwc = 1 writes [N,2h,2w,64] (what mvs_fpn_level_cp reads as intra_prev) */
int mvs_fpn_level_layout(const float* intra_prev, const float* lateral, const float* w_inner_p, const float* b_inner, const float* w_packed,
                         const float* scale, const float* shift, int N, int Ck, int h, int w, float* intra_out, int intra_nhwc, float* out,
                         mvs_stream_t stream);
/* Levels 1 and 2 (Ck = 32 | 16) with the 3x3 convolution in three-term bf16 split form (csrc/fpn_lvl_x3.hip): mvs_fpn_level_layout's contract
 * (same sources, same intra_out in either layout, same out), the convolution's weights pre-split with the BatchNorm scale folded in.
 *   prepare: w3 [Ck,64,3,3] (out_k.0.weight), scale [Ck] -> prepared, mvs_fpn_level_x3s_prepared_bytes(Ck) bytes */
int64_t mvs_fpn_level_x3s_prepared_bytes(int Ck);
int mvs_fpn_level_x3s_prepare(const float* w3, const float* scale, int Ck, void* prepared, mvs_stream_t stream);
int mvs_fpn_level_x3s(const float* intra_prev, const float* lateral, const float* w_inner_p, const float* b_inner, const void* prepared,
                      const float* shift, int N, int Ck, int h, int w, float* intra_out, int intra_nhwc, float* out, mvs_stream_t stream);
/* The full-resolution level (models/module.py:266-268, Ck = 8: out3 = Swish(BN(conv3x3(up2(intra2) + inner3(conv01))))) in three-term bf16
 * split form (csrc/fpn_x3.hip): fp32 in / out, fp32-equivalent.  The convolution is linear, so the lateral path runs as ONE composed 3x3
 * convolution: the caller passes wc [Ck,Ck,3,3] = sum_c w3[:,c] * w_inner[c,:] (composed in fp64), shift = the folded BatchNorm shift PLUS
 * scale * (the response of inner3's bias through all nine taps), and border [9][Ck] = scale * (that response per tap), which the kernel
 * subtracts for the taps that fall into intra3's zero padding at the image border.
 *   prepare: w3 [Ck,64,3,3] (out3.0.weight), wc, scale [Ck] -> prepared, mvs_fpn_level_x3_prepared_bytes(Ck) bytes
 *   level:   intra_prev [N,64,h,w], lateral [N,Ck,2h,2w] -> out [N,2h,2w,Ck] channel-last */
int64_t mvs_fpn_level_x3_prepared_bytes(int Ck);
int mvs_fpn_level_x3_prepare(const float* w3, const float* wc, const float* scale, int Ck, void* prepared, mvs_stream_t stream);
int mvs_fpn_level_x3(const float* intra_prev, const float* lateral, const void* prepared, const float* shift, const float* border, int N,
                     int Ck, int h, int w, float* out, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Next row after the path: geometric consistency filtering of the depth maps, misc/fusion.py:79-122
 * (get_reproj / project_img, vis_filter, ave_fusion) as driven by test.py:404-438, in one pass per reference pixel.
 *   ref_depth [n,1,H,W]   src_depths [n,v,1,H,W]   ref_cam [n,2,4,4]   src_cams [n,v,2,4,4]  (cam[0]=extrinsic, cam[1][:3,:3]=K)
 *   workspace: mvs_geo_filter_workspace_bytes(n, v) bytes of device scratch (per-view camera algebra)
 * outputs (each may be NULL): reproj_xyd [n,v,3,H,W], in_range [n,v,1,H,W], masks [n,v,1,H,W] (float 0/1),
 *   mask [n,1,H,W] uint8 (sum_v masks >= vthresh - 1.1), ref_depth_ave [n,1,H,W], points [n,3,H,W] (world xyz of ave depth)
 * ------------------------------------------------------------------------------------------------------- */
int64_t mvs_geo_filter_workspace_bytes(int n, int v);
int mvs_geo_filter_fwd(const float* ref_depth, const float* src_depths, const float* ref_cam, const float* src_cams, int n, int v,
                       int H, int W, float img_dist_thresh, float depth_thresh, float vthresh, void* workspace, float* reproj_xyd,
                       float* in_range, float* masks, uint8_t* mask, float* ref_depth_ave, float* points, mvs_stream_t stream);
/* op-level forms for callers that keep the reference's three calls: vis_filter (fusion.py:101-109) and ave_fusion
 * (fusion.py:112-114) on a materialized reproj_xyd.  masks_in != NULL: averaged depth from given masks (ave_fusion);
 * else masks are computed from in_range and the thresholds.  Outputs may be NULL. */
int mvs_vis_filter_fwd(const float* ref_depth, const float* reproj_xyd, const float* in_range, const float* masks_in, int n, int v,
                       int H, int W, float img_dist_thresh, float depth_thresh, float vthresh, float* masks, uint8_t* mask,
                       float* ref_depth_ave, mvs_stream_t stream);
/* Dynamic consistency check: get_reproj_dynamic + vis_filter_dynamic (fusion.py:116-165) + the reduction of
 * test.py:503-514, one pass.  2 <= v <= 16.  A view passes level k (k = 2..v) when dist < k/dist_base and
 * |d_ref - d|/d_ref < k/rel_diff_base.  Outputs (each may be NULL): reproj_xyd [n,v,3,H,W]; masks [n,v,v-1,H,W] uint8
 * (level k at index k-2); vis_mask [n,v,1,H,W] uint8 (= level v); geo_mask [n,1,H,W] uint8 (exists k: #views passing
 * level k >= k); ref_depth_ave [n,1,H,W] over the views in vis_mask; points [n,3,H,W].
 * mvs_vis_filter_dynamic_fwd is the op-level form on a materialized reproj_xyd. */
int mvs_geo_filter_dynamic_fwd(const float* ref_depth, const float* src_depths, const float* ref_cam, const float* src_cams, int n,
                               int v, int H, int W, float dist_base, float rel_diff_base, void* workspace, float* reproj_xyd,
                               uint8_t* masks, uint8_t* vis_mask, uint8_t* geo_mask, float* ref_depth_ave, float* points,
                               mvs_stream_t stream);
int mvs_vis_filter_dynamic_fwd(const float* ref_depth, const float* reproj_xyd, int n, int v, int H, int W, float dist_base,
                               float rel_diff_base, uint8_t* masks, uint8_t* vis_mask, uint8_t* geo_mask, float* ref_depth_ave,
                               mvs_stream_t stream);
/* prob_filter (fusion.py:69-77): mask = AND_c (conf[:, c] > thresh[c]), conf [n,C,HW], C <= 4, thresh in HOST memory;
 * depth_inplace (may be NULL) [n,HW] is multiplied by the mask (test.py:414-418). */
int mvs_prob_filter(const float* conf, int n, int C, int64_t HW, const float* thresh_host, uint8_t* mask, float* depth_inplace,
                    mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * the classification loss fused with the head's gradient, one stage of ce_loss_stage4
 * (models/losses.py:304-350, focal=False).
 *   logits [B,D,HW] = prob_volume_pre (D >= 2), depth_values [B,D,HW], depth_gt [B,HW], mask [B,HW] (valid where > 0.5)
 *   inverse_depth != 0: hypotheses run far -> near and are read in flipped order, as the reference flips them
 * fwd: acc = mvs_ce_loss_acc_floats(B, HW) floats: acc[0] = sum over valid pixels of -log softmax(logits)[gt bin], acc[1] = number of
 *      valid pixels (both written by the finalize launch from the per-block rows that follow them: fixed order, no atomics),
 *      loss[0] = weight * acc[0] / acc[1] (NaN when no pixel is valid, like F.cross_entropy on an empty selection);
 *      grad_unscaled [B,D,HW] (may be NULL) = (softmax - onehot) on valid pixels, 0 elsewhere;
 *      valid [B,HW] uint8 and gt_index [B,HW] int32 (index in FLIPPED order when inverse_depth) may be NULL.
 * bwd_scale: grad = grad_unscaled * weight * grad_out[0] / acc[1]   (grad_out: device scalar, dL/dloss; grad may alias grad_unscaled)
 * ------------------------------------------------------------------------------------------------------- */
int64_t mvs_ce_loss_acc_floats(int B, int64_t HW);
int mvs_ce_loss_fwd(const float* logits, const float* depth_values, const float* depth_gt, const float* mask, int B, int D, int64_t HW,
                    int inverse_depth, float weight, float* grad_unscaled, float* acc, float* loss, uint8_t* valid, int* gt_index,
                    mvs_stream_t stream);
/* models/losses.py:353-408 (mixup_ce_loss_stage4), one stage: loss = weight * sum(mask * (w_l*CE(logits[:-1], idx) + w_r*CE(logits[1:], idx)))
 * / (sum(mask) + 1e-6); acc / grad_unscaled / loss as mvs_ce_loss_fwd (acc[1] holds the denominator: mvs_ce_loss_bwd_scale applies it). */
int mvs_mixup_ce_loss_fwd(const float* logits, const float* depth_values, const float* depth_gt, const float* mask, int B, int D, int64_t HW,
                          int inverse_depth, float weight, float* grad_unscaled, float* acc, float* loss, mvs_stream_t stream);
/* models/losses.py:51-85 (reg_loss_stage4), one stage: weight * mean over {mask > 0.5 [and gt in range if depth_values != NULL]} of
 * smooth_l1(depth/interval[b] - gt/interval[b]); depth, gt, mask [B][HW], depth_values [B][D][HW] or NULL (mask_out_range=False),
 * grad_unscaled [B][HW] = d(sum)/d depth (may be NULL), acc >= mvs_ce_loss_acc_floats(B, HW) floats. */
int mvs_reg_loss_fwd(const float* depth, const float* depth_gt, const float* mask, const float* depth_values, const float* interval, int B, int D,
                     int64_t HW, int inverse_depth, float weight, float* grad_unscaled, float* acc, float* loss, mvs_stream_t stream);
/* models/losses.py:88-162 (wasserstein_loss -> sinkhorn with continuous = False, the form trainer/mvsformer_trainer.py:114-117 uses), one stage:
 * weight * mean over {mask > 0.5} of sum_ij T_ij |i - j|, T the Sinkhorn plan after ot_iter log-domain iterations between the prob_volume column
 * and the one-hot of the hypothesis nearest to depth_gt (cost |i - j| / ot_eps, the reference's signs).  prob_volume, depth_values [B][D][HW],
 * D <= 32, ot_iter <= 16; grad_unscaled [B][D][HW] = d(sum of the selected pixels' losses) / d prob_volume THROUGH all iterations (may be
 * NULL); acc >= mvs_was_loss_acc_floats(B, HW) floats; the 1 / N of the mean is applied by mvs_ce_loss_bwd_scale. */
int64_t mvs_was_loss_acc_floats(int B, int64_t HW);
int mvs_was_loss_fwd(const float* prob_volume, const float* depth_values, const float* depth_gt, const float* mask, int B, int D, int64_t HW,
                     int ot_iter, float ot_eps, float weight, float* grad_unscaled, float* acc, float* loss, mvs_stream_t stream);
int mvs_ce_loss_bwd_scale(const float* grad_unscaled, float* grad, int64_t numel, const float* acc, const float* grad_out, float weight,
                          mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * the DINO ViT-small feature branch of MVSFormer-P (csrc/vit.hip; models/vision_transformer.py:104-154,194-214,324-451,
 * models/module.py:353-368,450-466, models/mvsformer_model.py:243-262), eval mode.  fp32 in / fp32 out; every matrix product on the bf16
 * matrix cores in the three-term split form (fp32-equivalent, see "Arithmetic" above).
 *   mvs_gemm_x3: C[b1][b2] = epi(alpha * A[b1][b2] . B[b1][b2]^T) for nb1 x nb2 batches with element strides s?1 / s?2 per operand;
 *       A [M][K] rows lda apart; B [N][K] rows ldb apart (b_kn = 0) or [K][N] (b_kn = 1); C [M][N] rows ldc apart.
 *       a_mode 1: A is the implicit im2col of a 3x3 / pad-1 convolution over a channel-last map [H][W][Cp] (M = H*W, K = 9*Cp, k = tap*Cp + c);
 *       a_mode 2: the 2x2 taps of output-parity class b2 (ph = b2 / 2, pw = b2 % 2; nb2 = 4) of a ConvTranspose2d(kernel 4, stride 2,
 *       padding 1) over [H][W][Cp] (M = H*W input pixels, K = 4*Cp; class output (y, x) is output pixel (2y + ph, 2x + pw)).  Cp % 8 == 0.
 *       epi(v) = act(v*scale[n] + shift[n]) * mul + res   (scale / shift / mul / res may be NULL; act 0 none, 1 GELU(erf), 2 Swish, 3 ReLU;
 *       mul and res are C-shaped with C's strides).
 *   mvs_layernorm: y = (x - mean) * rsqrt(var + eps) * gamma + beta over rows of C <= 1024 features.
 *   mvs_softmax_rows: y = softmax(scale * x) over rows of N <= 8192 (in place allowed).
 *   mvs_bicubic_resize: ATen's upsample_bicubic2d, align_corners = False (A = -0.75, clamped taps), [planes][H][W] -> [planes][Ho][Wo];
 *       rscale = input / output size, or 1 / scale_factor when the caller resized by a scale factor (vision_transformer.py:407-411).
 * ------------------------------------------------------------------------------------------------------- */
int mvs_gemm_x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int nb1, int nb2, int64_t sA1,
                int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, int b_kn, int a_mode, int H, int W, int Cp, float alpha,
                const float* scale, const float* shift, int act, const float* mul, const float* res, mvs_stream_t stream);
/* out [B][N][heads*64] = softmax(scale * Q K^T) V per (image, head) in flash form (online softmax; the N x N matrix is never written), split-form
 * arithmetic; qkv = [B][N][3*heads*64] packed rows (q | k | v), vt = V transposed [B][heads][64][ldv] with row stride ldv >= N floats, a
 * multiple of 4 (16-byte aligned rows: the kernel reads them with 16-byte buffer loads, two key tiles ahead), padding finite; head_dim = 64 */
int mvs_attention_x3(const float* qkv, const float* vt, float* out, int B, int N, int heads, int head_dim, int ldv, float scale,
                     mvs_stream_t stream);
int mvs_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps, mvs_stream_t stream);
int mvs_softmax_rows(const float* x, float* y, int64_t rows, int N, float scale, mvs_stream_t stream);
int mvs_bicubic_resize(const float* in, float* out, int planes, int H, int W, int Ho, int Wo, float rscale_h, float rscale_w,
                       mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * The transformer blocks on PRE-SPLIT ("packed") operands (csrc/vit_packed.hip; models/vision_transformer.py:123-154,194-214): every matrix
 * operand is stored already split into the three bf16 terms of "Arithmetic" above and already in MFMA fragment order, so the main loops move
 * it with LDS-DMA and contain no conversion.  Same arithmetic as mvs_gemm_x3 / mvs_attention_x3 (six v_mfma_f32_16x16x32_bf16 per K = 32
 * step, fp32 accumulation), fp32-equivalent.
 *   packed X [R][K] (K % 32 == 0, rows_alloc % 16 == 0 rows allocated): piece (rt = r/16, ks = k/32, term t) = 1 KiB of 64 lanes x 8 bf16 at
 *   byte ((rt * K/32 + ks) * 3 + t) * 1024, lane = ((k%32)/8)*16 + r%16, element k%8; mvs_x3p_bytes(rows_alloc, K) bytes in all.
 *   mvs_x3p_pack / mvs_x3p_unpack: fp32 [R][K] (rows ld apart) <-> packed (rows >= R packed as zeros; unpack is exact: h + m + l).
 *   mvs_layernorm_x3p: LayerNorm (vision_transformer.py:199,207 norm1 / norm2) of rows laid out [images][Np] (Np % 16 == 0), written packed;
 *       rows t >= N of an image are padding: zeros.  32 <= C <= 512, C % 32 == 0.
 *   mvs_gemm_x3p: out = epi(A . B^T) with A [M][K] and B [N][K] packed (the nn.Linear layers: attn.proj, mlp.fc1, mlp.fc2);
 *       epi(v) = act(v*scale[n] + shift[n]) + res (act 0 none, 1 GELU(erf)); written as fp32 C [M][ldc] and / or packed Op [M][N]
 *       (the next GEMM's A operand; Op needs a_rows_alloc rows allocated).  N % 4 == 0.
 *   mvs_gemm_x3p_qkv: attn.qkv (vision_transformer.py:139): rows = [images][Np] tokens, columns q | k | v of `heads` heads of 64; writes
 *       Qp = (q + bias) * qscale (qscale = softmax scale * log2(e): the attention kernels exponentiate in base 2) and Kp packed [image][head][Np rows][64], and Vtp = V^T packed [image][head][64 rows][Np] with the 32 keys of
 *       a k step permuted (element e of chunk c <-> key (e < 4 ? 4c + e : 16 + 4c + e - 4)): the order mvs_attention_x3p's accumulators
 *       produce P in.  C = heads * 64, C % 128 == 0, Np % 32 == 0.
 *   mvs_attention_x3p: P V with P = 2^(Q K^T - reference) / row sum = softmax(scale * q k^T) for Qp as above, per (image, head), flash form,
 *       keys >= N masked; output packed [images * Np][heads * 64].
 *   mvs_cls_attention_x3p: att [image][head][N] = that softmax's CLS row: the one attention row the model reads (mvsformer_model.py:257).
 * ------------------------------------------------------------------------------------------------------- */
int64_t mvs_x3p_bytes(int64_t rows_alloc, int K);
int mvs_x3p_pack(const float* x, void* out, int64_t R, int K, int ld, int64_t rows_alloc, mvs_stream_t stream);
int mvs_x3p_unpack(const void* in, float* x, int64_t R, int K, int ld, int64_t rows_alloc, mvs_stream_t stream);
int mvs_layernorm_x3p(const float* x, const float* gamma, const float* beta, void* out, int64_t rows, int C, int Np, int N, float eps,
                      mvs_stream_t stream);
int mvs_gemm_x3p(const void* Ap, const void* Bp, int M, int N, int K, int64_t a_rows_alloc, int64_t b_rows_alloc, float* C, int ldc,
                 const float* scale, const float* shift, int act, const float* res, void* Op, mvs_stream_t stream);
/* Implicit convolutions of the ViT decoder (models/module.py:353-368 VITDecoderStage4Single, :450-466 AttentionFusionSimple) on the same kernel:
 * Xp = a packed channel-last map [images*H*W pixels][Cp] (Cp % 32 == 0) whose row `zero_row` is all zeros (taps outside the image read it).
 *   mode 1: 3x3, padding 1: out[pixel][n] = epi(sum_{tap,c} X[pixel + tap][c] * Wp[n][tap*Cp + c]), tap = ky*3 + kx.
 *   mode 2: ConvTranspose2d(kernel 4, stride 2, padding 1): Wp = four packed matrices [class][w_rows_alloc][4*Cp] (class = ph*2 + pw the output
 *           parity, k = (th*2 + tw)*Cp + c, taps ky = (1,3) / (0,2) for ph = 0 / 1, kx likewise); output row = image*4HW + (2y+ph)*2W + 2x+pw.
 * epi(v) = act(v*scale[n] + shift[n]) * mul (act 0 none, 1 GELU(erf), 2 Swish, 3 ReLU; mul [rows][ldc] or NULL); outputs as mvs_gemm_x3p. */
int mvs_conv_x3p(const void* Xp, int64_t x_rows_alloc, int zero_row, const void* Wp, int64_t w_rows_alloc, int mode, int images, int H, int W, int Cp,
                 int N, float* C, int ldc, const float* scale, const float* shift, int act, const float* mul, void* Op, mvs_stream_t stream);
int mvs_gemm_x3p_qkv(const void* Ap, const void* Bp, int images, int Np, int C, int heads, int64_t a_rows_alloc, int64_t b_rows_alloc,
                     const float* bias, float qscale, void* Qp, void* Kp, void* Vtp, mvs_stream_t stream);
int mvs_attention_x3p(const void* Qp, const void* Kp, const void* Vtp, void* Op, int images, int N, int Np, int heads, mvs_stream_t stream);
int mvs_cls_attention_x3p(const void* Qp, const void* Kp, float* att, int images, int N, int Np, int heads, mvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * training mode of the FPN encoder / decoder (models/module.py:208-270 under train(); csrc/vit.hip, csrc/fpn_train.hip) and of
 * the ViT decoder (models/module.py:353-368,450-466: a ConvTranspose2d's forward / data gradient / weight gradient are modes 2 / 1 / 3 with the
 * tensors' roles swapped).  fp32 NCHW like the reference; convolutions as split-form GEMMs with an implicit patch matrix, any kernel size, stride 1 / 2:
 *   mvs_conv2d_gemm_x3 mode 1: y [nb1][Cout][Ho*Wo] = w [Cout][Cin*KS*KS] . patches(x [nb1][Cin][H][W])          (A = w, Bmap = x)
 *                      mode 2: dx [nb1][Cin][H*W]   = wT [Cin][Cout*KS*KS] . gather(dy [nb1][Cout][Ho][Wo])     (A = w.permute(1,0,2,3), Bmap = dy)
 *                      mode 3: part [nb1][nsplit][Cout][Cin*KS*KS] = dy . patches(x)^T over ksplit output pixels per split (A = dy, Bmap = x;
 *                              ksplit % 32 == 0, nsplit = ceil(Ho*Wo / ksplit)); the caller adds the partial matrices in a fixed order.
 *   The BatchNorm kernels' `relu` argument (mvs_affine_act, mvs_bn_bwd_reduce, mvs_bn_bwd_apply) is an activation code: 0 none, 1 ReLU,
 *   2 leaky ReLU(0.1) (FPNEncoder), 3 Swish (FPNDecoder), 4 GELU(erf) (VITDecoderStage4Single).
 *   mvs_upsample2x_add: y = bilinear_x2(x, align_corners = True) (+ lateral); mvs_upsample2x_bwd: its adjoint (a gather: no atomics).
 * ------------------------------------------------------------------------------------------------------- */
int mvs_conv2d_gemm_x3(int mode, const float* A, const float* Bmap, float* C, int nb1, int Cin, int Cout, int H, int W, int Ho, int Wo, int KS,
                       int S, int P, int ksplit, mvs_stream_t stream);
int mvs_partials_reduce(const float* part, int nparts, int n, float* out, mvs_stream_t stream);   /* out[j] = sum_p part[p*n + j], fixed order */
int mvs_upsample2x_add(const float* x, const float* lateral, float* y, int planes, int h, int w, mvs_stream_t stream);
int mvs_upsample2x_bwd(const float* dy, float* dx, int planes, int h, int w, mvs_stream_t stream);

/* ---- AdamW over many parameter tensors in a few launches (the optimizer of the reference's trainer: train.py:98 torch.optim.AdamW) ----
 * p <- p*(1 - lr*wd);  m <- b1*m + (1-b1)*g;  v <- b2*v + (1-b2)*g*g;  p <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = step[0] + 1;
 * then step[0] += 1 (a device scalar: the call is capturable).  tensors = HOST array; all tensors fp32, contiguous, on the stream's device. */
typedef struct MvsAdamTensor {
    float* p;                 /* parameter, updated in place */
    const float* g;           /* its gradient */
    float* m;                 /* exp_avg */
    float* v;                 /* exp_avg_sq */
    int64_t n;                /* elements */
} MvsAdamTensor;
int mvs_adamw_step(const MvsAdamTensor* tensors, int ntensors, float lr, float beta1, float beta2, float eps, float weight_decay, int maximize,
                   float* step, mvs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVS_HIP_H */
