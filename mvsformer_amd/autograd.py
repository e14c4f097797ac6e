"""torch.autograd glue for training mode: every forward AND backward op is a hand-written HIP
kernel of libmvs_hip.so; PyTorch only sequences them (autograd graph, parameter .grad accumulation, DDP hooks,
SyncBatchNorm's all-reduce of the per-channel sums).

Gradient contract restated from the reference: the sampling grid is built under ``no_grad`` and hypotheses are detached
(models/warping.py:79-97, mvsformer_model.py:290,430), the entropy that feeds ``self.vis`` is detached
(mvsformer_model.py:89) — so gradients reach the feature maps, the ``vis`` CNN parameters and the regularizer
parameters; ``depth`` (arg-max gather) and ``photometric_confidence`` are not differentiable.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _sync_sums(sums: torch.Tensor, count: float, bn):
    """SyncBatchNorm (reference train.py:138-139): all-reduce ``[sum, sumsq]`` and the element count over the process group.

    Returns ``(sums, count_dev)``.  ``count_dev`` is None when nothing was reduced (plain BatchNorm, or world size 1: the
    caller keeps its host count); otherwise it is a 2-element DEVICE tensor ``[n / 4096, n % 4096]`` summed over the ranks in
    the SAME collective as the sums (one all-reduce of 2C+2 floats per layer, identical on every rank whatever the local
    batch shapes are; each half stays exact in fp32 up to 2^36 elements) - the BatchNorm kernels read the global count from
    device memory, so the step never synchronizes with the host and no rank has to guess whether counts are equal.
    ``count == 0`` (backward: the caller already holds the forward's ``count_dev``) reduces only the sums."""
    if not (isinstance(bn, nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized()):
        return sums, None
    world = dist.get_world_size(bn.process_group)
    if world <= 1:
        return sums, None
    if count == 0.0:
        red = sums.clone()                         # out of place: the caller's tensor keeps this rank's values (dgamma / dbeta)
        dist.all_reduce(red, group=bn.process_group)
        return red, None
    packed = torch.cat([sums, _count_halves(sums.device, int(count))])
    dist.all_reduce(packed, group=bn.process_group)
    return packed[:-2], packed[-2:]


_COUNT_HALVES = {}


def _capturing(device) -> bool:
    return device.type == "cuda" and torch.cuda.is_current_stream_capturing()


def _count_halves(device, n: int) -> torch.Tensor:
    """``[n / 4096, n % 4096]`` as a device tensor, cached per (device, n): the host-to-device copy happens once, OUTSIDE any hipGraph
    capture (a pageable copy inside a capture would either fail or bake a stale host pointer into the graph); ``CapturedStep`` warms
    every layer up eagerly first, so a capture only ever finds the cached tensor."""
    key = (str(device), n)
    t = _COUNT_HALVES.get(key)
    if t is None:
        if _capturing(device):
            raise ops._lib.MvsHipError("SyncBatchNorm element count %d first seen inside a hipGraph capture: run the step eagerly once" % n)
        t = torch.tensor([float(n // 4096), float(n % 4096)], dtype=torch.float32, device=device)
        _COUNT_HALVES[key] = t
    return t


def _momentum(bn) -> float:
    """``momentum=None`` is torch's cumulative moving average: factor 1/(num_batches_tracked + 1) for this update."""
    if bn.momentum is not None:
        return float(bn.momentum)
    if bn.num_batches_tracked is not None and _capturing(bn.num_batches_tracked.device):
        raise ops._lib.MvsHipError("BatchNorm(momentum=None) reads num_batches_tracked on the host and cannot be captured in a hipGraph")
    seen = int(bn.num_batches_tracked.item()) if bn.num_batches_tracked is not None else 0
    return 1.0 / float(seen + 1)


def _wino_ok(stride, cin, cout, x) -> bool:
    """TRAINING forward: stride-1 layers with 16/32/48/64 output channels take the Winograd F(2x2,3x3) fp32 kernel (MVS_CONV_WINO=0
    disables).  The eval path prefers the split-form bf16 kernels instead (module.Conv3d: Winograd there is opt-in, MVS_CONV_WINO=1);
    training keeps fp32 MFMA arithmetic end to end, and the Winograd kernel is bit-reproducible in the shipped build (no packed fp32
    instructions: DESIGN.md 4.7c)."""
    return tuple(stride) == (1, 1) and os.environ.get("MVS_CONV_WINO", "1") != "0" and \
        ops.conv3d_wino_supported(cin, cout, *x.shape[2:])


class ConvFn(torch.autograd.Function):
    """Raw 3x3x3 convolution, padding 1, stride (sd, shw, shw); weight is the layer's ``nn.Conv3d.weight``."""

    @staticmethod
    def forward(ctx, x, weight, stride):
        x = x.contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        cout, cin = w.shape[0], w.shape[1]
        if _wino_ok(stride, cin, cout, x):
            y = ops.conv3d_wino(x, ops.conv3d_wino_pack(w), cin, cout, None, None, None, relu=False)
        else:
            y = ops.conv3d(x, ops.conv3d_pack(w, False), cin, cout, stride, None, None, None, relu=False)
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = w.shape[0], w.shape[1]
        sd, shw = ctx.stride
        dx = None
        if ctx.needs_input_grad[0]:
            if shw == 1 and _wino_ok((1, 1), cout, cin, dy):
                # data gradient of a stride-1 conv = stride-1 conv of dY with channels swapped and taps mirrored: Winograd too
                wt = w.flip(2, 3, 4).transpose(0, 1).contiguous()
                dx = ops.conv3d_wino(dy, ops.conv3d_wino_pack(wt), cout, cin, None, None, None, relu=False)
            elif shw == 1:
                dx = ops.conv3d(dy, ops.conv3d_pack(w, "dgrad"), cout, cin, (1, 1), None, None, None, relu=False)
            else:       # strided conv: the data gradient is the transposed conv with the same weight
                dx = ops.deconv3d(dy, ops.conv3d_pack(w, True, sd), cout, cin, sd, None, None, None, relu=False)
            if dx.shape != x.shape:
                raise ops._lib.MvsHipError("conv backward: input %s is not 2x the output grid %s" % (tuple(x.shape), tuple(dy.shape)))
        dw = ops.conv3d_wgrad(dy, x, (sd, shw)) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class DeconvFn(torch.autograd.Function):
    """Raw ConvTranspose3d k=3, stride (sd,2,2), padding 1, output_padding (sd-1,1,1); weight ``[Cin,Cout,3,3,3]``."""

    @staticmethod
    def forward(ctx, x, weight, sd):
        x = x.contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        cin, cout = w.shape[0], w.shape[1]
        y = ops.deconv3d(x, ops.conv3d_pack(w, True, sd), cin, cout, sd, None, None, None, relu=False)
        ctx.save_for_backward(x, w)
        ctx.sd = sd
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        cin, cout = w.shape[0], w.shape[1]
        sd = ctx.sd
        dx = None
        if ctx.needs_input_grad[0]:
            # dX[ci] = sum_{co,k} dY[co, i*s-1+k] * W[ci,co,k]  ==  strided conv of dY with W read as [out=ci, in=co]
            dx = ops.conv3d(dy, ops.conv3d_pack(w, False), cout, cin, (sd, 2), None, None, None, relu=False)
        dw = ops.conv3d_wgrad(x, dy, (sd, 2)) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class BnActFn(torch.autograd.Function):
    """Training-mode BatchNorm (batch statistics, running-stat update) + optional ReLU + optional residual add.

    ``groups > 1``: the batch holds ``groups`` independent calls of the same module, batch index ``b*groups + g`` (the
    visibility CNN is applied once per source view, mvsformer_model.py:91).  Statistics are taken per (group, channel) —
    the tensor is simply viewed as ``[B, groups*C, ...]`` — the affine parameters are shared, running statistics receive the
    groups' momentum updates in order, exactly as ``groups`` separate calls would leave them."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu, groups=1):
        x = x.contiguous()
        shape = x.shape
        if groups > 1:
            if shape[0] % groups or residual is not None:
                raise ops._lib.MvsHipError("grouped BatchNorm: batch %d is not a multiple of %d groups (or has a residual)" % (shape[0], groups))
            x = x.view(shape[0] // groups, groups * shape[1], *shape[2:])
        B, C = x.shape[0], x.shape[1]
        count = float(x.numel() // C)
        sums = ops.bn_stats(x)
        sums, count_dev = _sync_sums(sums, count, bn)
        g = gamma.detach().to(torch.float32).contiguous() if gamma is not None else None
        b = beta.detach().to(torch.float32).contiguous() if beta is not None else None
        track = bn.track_running_stats and bn.running_mean is not None
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
        if groups > 1:
            if bn.momentum is None:
                raise ops._lib.MvsHipError("grouped BatchNorm needs a fixed momentum (cumulative averaging changes per group)")
            scale, shift, mean, invstd = ops.bn_finalize_grouped(sums, g, b, rm, rv, bn.momentum, bn.eps, count, groups, count_dev)
        else:
            scale, shift, mean, invstd = ops.bn_finalize(sums, g, b, rm, rv, _momentum(bn) if track else 0.0, bn.eps, count, count_dev)
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(groups)
        res = residual.contiguous() if residual is not None else None
        y = ops.affine_act(x, scale, shift, res, relu)
        gfull = (g if g is not None else scale.new_ones(C // groups))
        if groups > 1:
            gfull = gfull.repeat(groups)
        ctx.save_for_backward(x, scale, shift, mean, invstd, gfull, count_dev)
        ctx.relu, ctx.count, ctx.bn, ctx.has_res, ctx.groups, ctx.shape = relu, count, bn, residual is not None, groups, shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, invstd, g, count_dev = ctx.saved_tensors
        dy = dy.contiguous().view(x.shape)
        sums = ops.bn_bwd_reduce(dy, x, scale, shift, mean, invstd, ctx.relu)
        C = x.shape[1]
        local = sums                               # dgamma / dbeta stay per-rank (DDP averages parameter grads itself); _sync_sums
                                                   # all-reduces a packed COPY, so `sums` keeps this rank's values - no clone needed
        sums, _ = _sync_sums(sums, 0.0, ctx.bn)
        dx = ops.bn_bwd_apply(dy, x, scale, shift, mean, invstd, g, sums, ctx.count, ctx.relu, count_dev).view(ctx.shape)
        dgamma = local[C:] if ctx.needs_input_grad[1] else None          # views of the 2C-float reduction result (48 BN layers x 3
        dbeta = local[:C] if ctx.needs_input_grad[2] else None           # device copies per step otherwise)
        if ctx.groups > 1:                         # shared parameters: sum the groups' gradients
            dgamma = dgamma.view(ctx.groups, -1).sum(0) if dgamma is not None else None
            dbeta = dbeta.view(ctx.groups, -1).sum(0) if dbeta is not None else None
        dres = dy if ctx.has_res else None
        return dx, dgamma, dbeta, dres, None, None, None


class Prob1Fn(torch.autograd.Function):
    """1x1x1 (or 1x1) convolution C -> 1 with bias."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        w = weight.detach().to(torch.float32).reshape(-1).contiguous()
        b = bias.detach().to(torch.float32).reshape(-1).contiguous()
        B, C = x.shape[0], x.shape[1]
        x5 = x.reshape(B, C, 1, 1, -1)
        y = ops.prob1(x5, w, b).reshape(B, 1, *x.shape[2:])
        ctx.save_for_backward(x, w)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dwb = ops.prob1_bwd(x, w, dy.contiguous().reshape(x.shape[0], -1))
        C = x.shape[1]
        return dx, dwb[:C].reshape(ctx.wshape), dwb[C:]


class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.sigmoid(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.sigmoid_bwd(y, dy.contiguous())


class AggregateFn(torch.autograd.Function):
    """volume_mean = sum_v w_v * corr(ref, warp(src_v)) / (sum_v w_v + 1e-6); grads to features and to w."""

    @staticmethod
    def forward(ctx, feat, weight, rt, hyp, G, feat_cl=None, as_bf16=False):
        """``feat_cl``: the channel-last copy of ``feat`` if the caller already made it (the entropy sweep reads the same one).
        ``as_bf16``: return the volume as bf16 channel-last ``[B,D,H,W,G]`` (what the bf16 regularizer reads; written by the same launch,
        its gradient arrives in the same form and the backward kernel reads it as it is) - the fp32 volume is still kept for the backward."""
        if feat_cl is None:
            feat_cl = ops.to_channels_last(feat.detach().to(torch.float32).contiguous())
        weight = weight.contiguous()
        if as_bf16:
            vol, _, vol16 = ops.cv_aggregate(feat_cl, rt, hyp, weight, G, want_sim_depth=False, exact=True, want_bf16=True)
        else:
            vol, _ = ops.cv_aggregate(feat_cl, rt, hyp, weight, G, want_sim_depth=False, exact=True)   # same geometry as the backward kernel
        ctx.save_for_backward(feat_cl, rt, hyp, weight, vol)
        ctx.G, ctx.dtype = G, feat.dtype
        return vol16 if as_bf16 else vol

    @staticmethod
    def backward(ctx, gvol):
        feat_cl, rt, hyp, weight, vol = ctx.saved_tensors
        dfeat_cl, dw = ops.cv_aggregate_bwd(feat_cl, rt, hyp, weight, vol, gvol.contiguous(), ctx.G)
        dfeat = ops.to_channels_first(dfeat_cl).to(ctx.dtype) if ctx.needs_input_grad[0] else None
        return dfeat, (dw if ctx.needs_input_grad[1] else None), None, None, None, None, None


class HeadFn(torch.autograd.Function):
    """softmax over depth + arg-max depth + max-prob confidence (training branch of mvsformer_model.py:110-125)."""

    @staticmethod
    def forward(ctx, pre, hyp, tmp):
        pre = pre.contiguous()
        _, prob, depth, conf = ops.head(hyp, float(tmp), True, logits=pre)
        ctx.save_for_backward(prob)
        ctx.mark_non_differentiable(depth, conf)
        return prob, depth, conf

    @staticmethod
    def backward(ctx, dprob, _dd, _dc):
        (prob,) = ctx.saved_tensors
        return ops.softmax_bwd(prob, dprob.contiguous()), None, None


# ---------------------------------------------------------------------------------------------------------
class _EmbedConv2dWeightFn(torch.autograd.Function):
    """``[Cout,Cin,3,3]`` -> ``[Cout,cin_pad,3,3,3]`` with the 2-D kernel in the centre depth tap, zeros elsewhere: one fill + one strided
    copy forward, a view backward (two ``F.pad`` calls are four launches forward and two more backward, 12 layers per step)."""

    @staticmethod
    def forward(ctx, w2d, cin_pad):
        cout, cin = w2d.shape[0], w2d.shape[1]
        w = w2d.new_zeros(cout, max(cin_pad, cin), 3, 3, 3)
        w[:, :cin, 1] = w2d
        ctx.cin = cin
        return w

    @staticmethod
    def backward(ctx, dw):
        return dw[:, :ctx.cin, 1], None


def embed_conv2d_weight(w2d: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """``[Cout,Cin,3,3]`` -> ``[Cout,cin_pad,3,3,3]`` with the 2-D kernel in the centre depth tap (so the 2-D convs of
    ``StageNet.vis`` run, forward and backward, on the same MFMA conv3d kernels at D = 1).  Pure padding: differentiable."""
    return _EmbedConv2dWeightFn.apply(w2d, cin_pad)


def _fused_layers() -> bool:
    """MVS_TRAIN_FUSED=0 keeps the round-4 chain of separate conv / BatchNorm autograd nodes (diagnostics, A/B timing)."""
    return os.environ.get("MVS_TRAIN_FUSED", "1") != "0"


def vis_train(entropy: torch.Tensor, vis: nn.Sequential) -> torch.Tensor:
    """``self.vis(entropy)`` in training mode for ONE source view: ``entropy [B,1,H,W]`` -> ``[B,1,H,W]``.  The reference
    calls the CNN once per view (mvsformer_model.py:91), so batch statistics are per view; keep that."""
    B, _, H, W = entropy.shape
    x = torch.zeros(B, 4, 1, H, W, device=entropy.device, dtype=torch.float32)
    x[:, 0, 0] = entropy[:, 0]
    for i in range(3):
        blk = vis[i]
        cin_pad = 4 if i == 0 else blk.conv.in_channels
        x = ConvFn.apply(x, embed_conv2d_weight(blk.conv.weight, cin_pad), (1, 1))
        x = BnActFn.apply(x, blk.bn.weight, blk.bn.bias, None, blk.bn, True)
    y = Prob1Fn.apply(x, vis[3].weight, vis[3].bias)                    # [B,1,1,H,W]
    return SigmoidFn.apply(y).reshape(B, 1, H, W)


def vis_train_views(entropy: torch.Tensor, vis: nn.Sequential) -> torch.Tensor:
    """All source views of ``entropy [B,Vs,H,W]`` through ``self.vis`` in training mode in ONE batched pass, with the
    statistics, running-stat updates and gradients of ``Vs`` separate per-view calls (reference mvsformer_model.py:91 calls
    the CNN once per view): the convolutions see a batch of ``B*Vs`` maps, BatchNorm works per (view, channel) group."""
    B, Vs, H, W = entropy.shape
    if os.environ.get("MVS_VIS_PER_VIEW", "0") == "1" or Vs == 1:
        return torch.cat([vis_train(entropy[:, v:v + 1], vis) for v in range(Vs)], dim=1)
    from .module import autocast_bf16
    if autocast_bf16() and os.environ.get("MVS_VIS_BF16", "1") != "0":
        # under autocast the reference runs this CNN in half precision too (it is inside the autocast region,
        # trainer/mvsformer_trainer.py:104-106): bf16 channel-last kernels, fp32 statistics, the 1x1 conv + sigmoid in fp32
        # batch index b*Vs + v; entropy is detached (no gradient)
        x16 = ops.bf16_embed_ch0(entropy.detach().to(torch.float32).contiguous().reshape(B * Vs, 1, H, W))
        fused = _fused_layers() and not any(_bn_synced(vis[i].bn) for i in range(3))
        for i in range(3):
            blk = vis[i]
            pk = packed_of(blk.conv)
            if fused and pk is not None:
                # the 2-D parameter itself (9 taps of a D = 1 volume), packed this step by the stage's StagePack
                x16 = LayerBf16Fn.apply(x16, route_of(blk.conv), blk.bn.weight, blk.bn.bias, None, blk.bn, True, 0, (1, 1), Vs, pk)
                continue
            cin_pad = 8 if i == 0 else blk.conv.in_channels
            w3 = embed_conv2d_weight(blk.conv.weight, cin_pad)
            if fused:
                x16 = LayerBf16Fn.apply(x16, w3, blk.bn.weight, blk.bn.bias, None, blk.bn, True, 0, (1, 1), Vs)
                continue
            x16 = ConvBf16Fn.apply(x16, w3, (1, 1))
            x16 = BnActBf16Fn.apply(x16, blk.bn.weight, blk.bn.bias, None, blk.bn, True, Vs)
        if _fused_layers():
            return HeadBf16Fn.apply(x16, vis[3].weight, vis[3].bias, True).reshape(B, Vs, H, W)
        y = Prob1Fn.apply(FromBf16Fn.apply(x16), vis[3].weight, vis[3].bias)
        return SigmoidFn.apply(y).reshape(B, Vs, H, W)
    x = torch.zeros(B * Vs, 4, 1, H, W, device=entropy.device, dtype=torch.float32)
    x[:, 0, 0] = entropy.reshape(B * Vs, H, W)                          # batch index b*Vs + v
    for i in range(3):
        blk = vis[i]
        cin_pad = 4 if i == 0 else blk.conv.in_channels
        x = ConvFn.apply(x, embed_conv2d_weight(blk.conv.weight, cin_pad), (1, 1))
        x = BnActFn.apply(x, blk.bn.weight, blk.bn.bias, None, blk.bn, True, Vs)
    y = Prob1Fn.apply(x, vis[3].weight, vis[3].bias)                    # [B*Vs,1,1,H,W]
    return SigmoidFn.apply(y).reshape(B, Vs, H, W)


# =========================================================================================================
# bf16 channel-last regularizer (training under torch autocast; reference trainer/mvsformer_trainer.py:104-106)
# =========================================================================================================
class ToBf16Fn(torch.autograd.Function):
    """fp32 ``[B,C,D,H,W]`` -> bf16 channel-last ``[B,D,H,W,C]`` (the cost volume enters the half-precision regularizer)."""

    @staticmethod
    def forward(ctx, x):
        return ops.bf16_from_f32(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return ops.bf16_to_f32(dy.contiguous())


class FromBf16Fn(torch.autograd.Function):
    """bf16 channel-last -> fp32 ``[B,C,D,H,W]`` (the logits leave it)."""

    @staticmethod
    def forward(ctx, x):
        return ops.bf16_to_f32(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return ops.bf16_from_f32(dy.contiguous())


class WgradQueue:
    """Weight-gradient jobs of one stage's backward, run TOGETHER when the last of them has been queued (:class:`WgradFlushFn`): one
    grid per kernel instance with all its layers side by side + one reduce (ops.bf16_wgrad_group) instead of two launches per layer.
    ``push`` returns the (still unwritten) gradient tensor; the operands stay referenced until the flush."""

    def __init__(self):
        self.jobs = []

    def push(self, A, Bt, stride, taps: int = 27, cb_out=None):
        dW = torch.empty(ops.bf16_wgrad_shape(A, Bt, taps, cb_out), device=A.device, dtype=torch.float32)
        self.jobs.append((A, Bt, dW, (int(stride[0]), int(stride[1])), int(taps), cb_out))
        return dW

    def flush(self, grads=()):
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        # every queued gradient must arrive here as the tensor we handed out (not a copy autograd made on the way): the fill below
        # writes through the pointer
        seen = {g.data_ptr() for g in grads if g is not None}
        for job in jobs:
            if job[2].data_ptr() not in seen:
                raise ops._lib.MvsHipError("WgradQueue: a deferred weight gradient did not reach its flush node as handed out "
                                           "(MVS_TRAIN_WGRAD_GROUP=0 runs the weight gradients layer by layer)")
        ops.bf16_wgrad_group(jobs)


class WgradFlushFn(torch.autograd.Function):
    """Identity on a stage's weights that marks where their gradients are COMPLETED: the layers that consume the routed weights return
    unwritten gradient tensors and queue the work (``weight._mvs_wq``); this node's backward runs when all of them have arrived - the
    end of the stage's backward - fills them with grouped launches and hands them on to the parameters (AccumulateGrad, DDP's hooks)."""

    @staticmethod
    def forward(ctx, queue, *weights):
        ctx.queue = queue
        return tuple(w.detach() for w in weights)

    @staticmethod
    def backward(ctx, *grads):
        ctx.queue.flush(grads)
        return (None,) + tuple(grads)


def _wgrad_group_on() -> bool:
    return os.environ.get("MVS_TRAIN_WGRAD_GROUP", "1") != "0"


def route_of(conv):
    """The weight a training layer should consume: the routed twin a :class:`StagePack` made for this forward (its gradient is completed
    by the stage's :class:`WgradFlushFn`), else the parameter itself."""
    r = getattr(conv, "_mvs_wroute", None)
    return r if r is not None else conv.weight


def _wgrad(weight_in, A, Bt, stride, taps: int = 27, cb_out=None):
    q = getattr(weight_in, "_mvs_wq", None)
    if q is not None:
        return q.push(A, Bt, stride, taps, cb_out)
    return ops.bf16_conv3d_wgrad(A, Bt, stride, taps, cb_out)


class ConvBf16Fn(torch.autograd.Function):
    """Raw 3x3x3 convolution on bf16 channel-last activations, fp32 master weight ``[Cout,Cin,3,3,3]`` cast per call."""

    @staticmethod
    def forward(ctx, x, weight, stride, stats_groups=0, packed=None, cout_pad=None):
        """``stats_groups`` > 0: also return the batch statistics of the output (``sums [2*groups*Cout]``, non-differentiable) computed
        in the convolution's epilogue, for the BatchNorm that follows (:class:`BnActBf16Fn` ``sums=``).  ``packed``: the (forward,
        data-gradient) layouts if a :class:`StagePack` already made them this step.  ``cout_pad`` > Cout: run as a wider map whose extra
        output channels are zero (CostRegNet's 8 -> 1 ``prob`` as 8 -> 8; needs ``packed``)."""
        x = x.contiguous()
        cout, cin = weight.shape[0], weight.shape[1]
        cpad = cout_pad or cout
        if packed is not None:
            wf, wb = packed
        else:
            if cpad != cout:
                raise ops._lib.MvsHipError("a padded convolution needs its weights packed by a StagePack")
            w = weight.detach().to(torch.float32).contiguous()
            if ctx.needs_input_grad[0]:     # the data gradient's layout (stride-1: channels swapped + taps mirrored; strided: transposed
                wf, wb = ops.bf16_pack2(w, (0, cin, cout), (2 if stride[1] == 1 else 1, cout, cin))     # conv) in the same launch
            else:
                wf, wb = ops.bf16_pack(w, 0, cin, cout), None
        ctx.save_for_backward(x, wb)
        ctx.stride, ctx.wshape, ctx.cpad = stride, (cout, cin), cpad
        ctx.win = weight if getattr(weight, "_mvs_wq", None) is not None else None
        if stats_groups:
            y, sums = ops.bf16_conv3d_stats(x, wf, cin, cpad, 0, stride, stats_groups)
            ctx.mark_non_differentiable(sums)
            return y, sums
        return ops.bf16_conv3d(x, wf, cin, cpad, 0, stride)

    @staticmethod
    def backward(ctx, dy, _dsums=None):
        x, wb = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = ctx.wshape
        cpad = ctx.cpad
        sd, shw = ctx.stride
        dx = None
        if ctx.needs_input_grad[0]:
            if shw == 1:        # stride-1 conv: the data gradient is the same conv with channels swapped and taps mirrored
                dx = ops.bf16_conv3d(dy, wb, cpad, cin, 0, (1, 1))
            else:               # strided conv: the transposed conv of the same weight
                dx = ops.bf16_conv3d(dy, wb, cpad, cin, 1, (sd, shw))
            if dx.shape != x.shape:
                raise ops._lib.MvsHipError("conv backward: input %s is not 2x the output grid %s" % (tuple(x.shape), tuple(dy.shape)))
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad(ctx.win, dy, x, (sd, shw))
            if cpad != cout:
                dw = dw[:cout]
        return dx, dw, None, None, None, None


class DeconvBf16Fn(torch.autograd.Function):
    """Raw ConvTranspose3d k3, stride (sd,2,2), padding 1, output_padding (sd-1,1,1) on bf16 channel-last activations."""

    @staticmethod
    def forward(ctx, x, weight, sd, stats_groups=0, packed=None):
        x = x.contiguous()
        cin, cout = weight.shape[0], weight.shape[1]                # [Cin,Cout,3,3,3]
        if packed is not None:
            wf, wb = packed
        else:
            w = weight.detach().to(torch.float32).contiguous()
            if ctx.needs_input_grad[0]:     # data gradient = strided conv of dY with W read as [out = cin, in = cout]
                wf, wb = ops.bf16_pack2(w, (1, cin, cout), (0, cout, cin))
            else:
                wf, wb = ops.bf16_pack(w, 1, cin, cout), None
        ctx.save_for_backward(x, wb)
        ctx.sd, ctx.wshape = sd, (cin, cout)
        ctx.win = weight if getattr(weight, "_mvs_wq", None) is not None else None
        if stats_groups:
            y, sums = ops.bf16_conv3d_stats(x, wf, cin, cout, 1, (sd, 2), stats_groups)
            ctx.mark_non_differentiable(sums)
            return y, sums
        return ops.bf16_conv3d(x, wf, cin, cout, 1, (sd, 2))

    @staticmethod
    def backward(ctx, dy, _dsums=None):
        x, wb = ctx.saved_tensors
        dy = dy.contiguous()
        cin, cout = ctx.wshape
        dx = None
        if ctx.needs_input_grad[0]:   # strided conv of dY with W read as [out = cin, in = cout]
            dx = ops.bf16_conv3d(dy, wb, cout, cin, 0, (ctx.sd, 2))
        dw = _wgrad(ctx.win, x, dy, (ctx.sd, 2)) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


class BnActBf16Fn(torch.autograd.Function):
    """Training-mode BatchNorm (fp32 batch statistics of the bf16 conv output, running-stat update) + ReLU + optional skip,
    channel-last bf16 in and out; SyncBatchNorm statistics ride the same all-reduce as in :class:`BnActFn`.  ``groups`` > 1: the
    batch holds ``groups`` independent calls of the module (sample n -> group n % groups), statistics per (group, channel)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu, groups=1, sums=None):
        """``sums``: the statistics of ``x`` if the producing convolution already took them (``ConvBf16Fn(..., stats_groups)``)."""
        x = x.contiguous()
        C = x.shape[-1]
        if groups > 1 and residual is not None:
            raise ops._lib.MvsHipError("grouped BatchNorm has no residual form")
        count = float(x.numel() // (C * groups))
        g = gamma.detach().to(torch.float32).contiguous() if gamma is not None else None
        b = beta.detach().to(torch.float32).contiguous() if beta is not None else None
        track = bn.track_running_stats and bn.running_mean is not None
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
        if groups > 1 and bn.momentum is None:
            raise ops._lib.MvsHipError("grouped BatchNorm needs a fixed momentum (cumulative averaging changes per group)")
        res = residual.contiguous() if residual is not None else None
        synced = isinstance(bn, nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() and dist.get_world_size(bn.process_group) > 1
        if sums is None and not synced:
            # no collective between statistics and finalize: one call, three launches
            mom = bn.momentum if groups > 1 else (_momentum(bn) if track else 0.0)
            nbt = bn.num_batches_tracked if track and bn.num_batches_tracked is not None else None       # incremented by the kernel
            y, scale, shift, mean, invstd = ops.bf16_bn_train_fwd(x, res, relu, g, b, rm, rv, mom, bn.eps, groups, nbt)
            count_dev = None
            track = False
        else:
            if sums is None:
                sums = ops.bf16_bn_stats(x, groups)
            sums, count_dev = _sync_sums(sums, count, bn)
            if groups > 1:
                scale, shift, mean, invstd = ops.bn_finalize_grouped(sums, g, b, rm, rv, bn.momentum, bn.eps, count, groups, count_dev)
            else:
                scale, shift, mean, invstd = ops.bn_finalize(sums, g, b, rm, rv, _momentum(bn) if track else 0.0, bn.eps, count, count_dev)
            y = ops.bf16_affine_act(x, scale, shift, res, relu, groups)
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(groups)
        gfull = g if g is not None else scale.new_ones(C)
        if groups > 1:
            gfull = gfull.repeat(groups)
        ctx.save_for_backward(x, scale, shift, mean, invstd, gfull, count_dev)
        ctx.relu, ctx.count, ctx.bn, ctx.has_res, ctx.groups = relu, count, bn, residual is not None, groups
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, invstd, g, count_dev = ctx.saved_tensors
        dy = dy.contiguous()
        G = ctx.groups
        sums = ops.bf16_bn_bwd_reduce(dy, x, scale, shift, mean, invstd, ctx.relu, G)
        CT = x.shape[-1] * G
        local = sums
        sums, _ = _sync_sums(sums, 0.0, ctx.bn)
        dx = ops.bf16_bn_bwd_apply(dy, x, scale, shift, mean, invstd, g, sums, ctx.count, ctx.relu, count_dev, G)
        if G > 1:                                                        # shared parameters: sum the groups' gradients (one launch)
            both = local.view(2, G, -1).sum(1)
            dbeta, dgamma = both[0], both[1]
        else:
            dbeta, dgamma = local[:CT], local[CT:]                       # views of the reduction result (no device copies)
        return dx, (dgamma if ctx.needs_input_grad[1] else None), (dbeta if ctx.needs_input_grad[2] else None), \
            (dy if ctx.has_res else None), None, None, None, None


def _bn_synced(bn) -> bool:
    return isinstance(bn, nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() and dist.get_world_size(bn.process_group) > 1


class SkipLink:
    """Hand-over of a skip connection's gradient between two :class:`LayerBf16Fn` nodes of one network (CostRegNet's c2 / c4 / input
    volume: each feeds a strided convolution AND, later, the residual input of a transposed-convolution layer).  Autograd would add the
    tensor's two gradients with a kernel of its own and the sum would arrive without the BatchNorm sums attached; instead

      * the strided convolution's layer (``take``) ARMS the link in its forward - it promises to add whatever the link holds to its
        data gradient in the convolution's epilogue (ops.bf16_conv3d with ``residual`` / ops.bf16_conv3d_bnbwd with ``addend``);
      * the residual layer (``give``), whose backward always runs first (its output depends on the strided layer's), leaves its incoming
        gradient - which IS the residual's gradient - in the link and returns None for the residual.

    An armed link that is empty when the taker's backward runs is an error (a gradient would be lost), never a silent zero."""

    __slots__ = ("armed", "grad")

    def __init__(self):
        self.armed, self.grad = False, None


class LayerBf16Fn(torch.autograd.Function):
    """One whole training layer on bf16 channel-last activations as ONE autograd node: (transposed) 3x3x3 convolution -> batch-statistics
    BatchNorm (statistics in the convolution's epilogue) -> [ReLU] [+ skip]; backward = BatchNorm backward, data gradient, weight
    gradient.  For layers whose statistics stay on this rank (plain BatchNorm, or SyncBatchNorm with a process group of one);
    :func:`mvsformer_amd.module._train_conv_bn_act` keeps the ConvBf16Fn -> BnActBf16Fn pair for the synchronized case, where an
    all-reduce sits between the statistics and the finalize.

    ``gather``: 0 = Conv3d with ``stride = (sd, shw)``, 1 = ConvTranspose3d with stride ``(sd, 2, 2)``.  ``weight`` may be a 2-D
    ``[Cout,Cin,3,3]`` parameter (the visibility CNN): it runs as the centre depth tap of a D = 1 volume (9 taps, a third of the matrix
    work of a zero-embedded 3x3x3 kernel) with the input's channel count (>= Cin, a multiple of 8) as the map's width.  ``packed``: the
    (forward, data-gradient) weight layouts if the caller's :class:`StagePack` already made them for this step, else None (packed
    here; 3x3x3 only)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, bn, relu, gather, stride, groups=1, packed=None, take=None, give=None):
        x = x.contiguous()
        taps = ops._taps_of(weight)
        if gather == 0:
            cout, cin = weight.shape[0], weight.shape[1]
            cin_map = x.shape[-1]                                  # >= cin: padding channels of the input (their weights pack as zeros)
            fwd, bwd = (0, cin_map, cout), (2 if stride[1] == 1 else 1, cout, cin_map)
        else:
            cin, cout = weight.shape[0], weight.shape[1]
            cin_map = cin
            fwd, bwd = (1, cin, cout), (0, cout, cin)
        need_dx = ctx.needs_input_grad[0]
        if packed is not None:
            wf, wb = packed
        else:
            if taps != 27 or cin_map != cin:
                raise ops._lib.MvsHipError("a 2-D / padded layer needs its weights packed by a StagePack")
            w = weight.detach().to(torch.float32).contiguous()
            if need_dx:
                wf, wb = ops.bf16_pack2(w, fwd, bwd)
            else:
                wf, wb = ops.bf16_pack(w, *fwd), None
        if groups > 1 and (residual is not None or bn.momentum is None):
            raise ops._lib.MvsHipError("grouped BatchNorm has no residual form and needs a fixed momentum")
        g = gamma.detach().to(torch.float32).contiguous() if gamma is not None else None
        b = beta.detach().to(torch.float32).contiguous() if beta is not None else None
        track = bn.track_running_stats and bn.running_mean is not None
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
        mom = bn.momentum if groups > 1 else (_momentum(bn) if track else 0.0)
        nbt = bn.num_batches_tracked if track and bn.num_batches_tracked is not None else None       # incremented by the kernel
        res = residual.contiguous() if residual is not None else None
        y, z, st = ops.bf16_conv3d_bn_fwd(x, wf, cin_map, cout, gather, stride, res, relu, g, b, rm, rv, mom, bn.eps, groups, nbt, taps)
        ctx.save_for_backward(x, wb, y, st)                  # st[4] = the affine weight per (group, channel), written by the finalize kernel
        ctx.cfg = (relu, gather, tuple(stride), groups, cin, cin_map, cout, residual is not None, taps)
        # The producer of x, if it is a layer of this kind whose output feeds only this one: its BatchNorm's backward sums can be taken in
        # the epilogue of OUR data gradient (which IS its incoming gradient) instead of by a pass of their own.  The tag rides on the
        # tensor object; a tensor that also feeds a skip connection has it removed by the network (module.py), and a gradient that was
        # accumulated from two consumers arrives as a new tensor without the answer attached - the plain reduce runs then.
        ctx.prev_bn = getattr(x, "_mvs_bn", None) if need_dx and _fused_layers() else None
        # skip-connection gradient hand-over (SkipLink): we take x's other gradient into our data gradient / we give the residual's away
        ctx.take = take if take is not None and need_dx and _fused_layers() else None
        if ctx.take is not None:
            ctx.take.armed, ctx.take.grad = True, None
        ctx.give = give if give is not None and give.armed and residual is not None and ctx.needs_input_grad[4] else None
        ctx.win = weight if getattr(weight, "_mvs_wq", None) is not None else None
        z._mvs_bn = (y, st, relu, groups)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, wb, y, st = ctx.saved_tensors
        scale, shift, mean, invstd, gfull = st[0], st[1], st[2], st[3], st[4]
        relu, gather, (sd, shw), groups, cin, cin_map, cout, has_res, taps = ctx.cfg
        dz = dz.contiguous()
        # the two sums of the BatchNorm backward: taken by the consumer's data-gradient convolution if it knew us (see forward), else here.
        # (An in-launch completion of the reduce - last block to arrive adds the partial rows - was built and measured: a thousand
        #  same-address agent-scope atomics cost 15-25 us per call, three times the launch they replace; NOTEBOOK.md)
        tag = getattr(dz, "_mvs_bn_sums", None)
        if tag is not None and tag[0] == y.data_ptr() and tag[1] == dz.data_ptr():
            sums = tag[2]
        else:
            sums = ops.bf16_bn_bwd_reduce(dz, y, scale, shift, mean, invstd, relu, groups)
        CT = cout * groups
        if groups > 1:                                                        # shared parameters: the groups' gradients summed by the apply kernel
            dy, dgb = ops.bf16_bn_bwd_apply(dz, y, scale, shift, mean, invstd, gfull, sums, float(y.numel() // CT), relu, None, groups, True)
            dbeta, dgamma = dgb[:cout], dgb[cout:]
        else:
            dy = ops.bf16_bn_bwd_apply(dz, y, scale, shift, mean, invstd, gfull, sums, float(y.numel() // CT), relu, None, groups)
            dbeta, dgamma = sums[:CT], sums[CT:]
        dx = None
        if ctx.needs_input_grad[0]:
            # stride 1: the same conv with channels swapped and taps mirrored; strided: the transposed conv; transposed conv: the strided
            # conv of dY with W read as [out = cin, in = cout]
            g_, st_, cin_ = ((0 if shw == 1 else 1), ((1, 1) if shw == 1 else (sd, shw)), cin_map) if gather == 0 else (0, (sd, 2), cin)
            prev = ctx.prev_bn
            other = None
            if ctx.take is not None:
                other, ctx.take.grad, ctx.take.armed = ctx.take.grad, None, False
                if other is None:
                    raise ops._lib.MvsHipError("SkipLink: the residual layer's backward has not run before the strided layer's - a gradient "
                                               "would be lost (MVS_TRAIN_SKIPLINK=0 turns the hand-over off)")
                if tuple(other.shape) != tuple(x.shape):
                    raise ops._lib.MvsHipError("SkipLink: gradient %s for a tensor of shape %s" % (tuple(other.shape), tuple(x.shape)))
            if prev is not None and tuple(prev[0].shape) == tuple(x.shape):
                dx, psums = ops.bf16_conv3d_bnbwd(dy, wb, cout, cin_, g_, st_, prev[0], prev[1], prev[2], prev[3], taps, other)
                dx._mvs_bn_sums = (prev[0].data_ptr(), dx.data_ptr(), psums)
            elif other is not None and taps == 27:
                dx = ops.bf16_conv3d(dy, wb, cout, cin_, g_, st_, residual=other)
            else:
                dx = ops.bf16_conv3d(dy, wb, cout, cin_, g_, st_, taps=taps)
                if other is not None:
                    dx = dx + other
            if dx.shape != x.shape:
                raise ops._lib.MvsHipError("conv backward: input %s does not match the gradient grid %s" % (tuple(x.shape), tuple(dx.shape)))
        dw = None
        if ctx.needs_input_grad[1]:
            if gather == 0:
                dw = _wgrad(ctx.win, dy, x, (sd, shw), taps, cin)
            else:
                dw = _wgrad(ctx.win, x, dy, (sd, 2))
        dres = dz if has_res else None
        if ctx.give is not None:                            # the strided layer that also consumed the residual adds it in its epilogue
            ctx.give.grad, dres = dz, None
        return dx, dw, (dgamma if ctx.needs_input_grad[2] else None), (dbeta if ctx.needs_input_grad[3] else None), \
            dres, None, None, None, None, None, None, None, None


class HeadBf16Fn(torch.autograd.Function):
    """1x1(x1) convolution 8 -> 1 with bias [+ sigmoid] straight on bf16 channel-last activations: ``x [...,8]`` -> fp32 ``[...]``
    (CostRegNet3D.prob, models/module.py:577; StageNet.vis[3:5], mvsformer_model.py:37).  One launch forward, two backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, sigmoid):
        x = x.contiguous()
        w = weight.detach().to(torch.float32).reshape(-1).contiguous()
        b = bias.detach().to(torch.float32).reshape(-1).contiguous()
        out = ops.bf16_head_fwd(x, w, b, sigmoid)
        ctx.save_for_backward(x, w, out if sigmoid else None)
        ctx.wshape = weight.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, y = ctx.saved_tensors
        dx, dwb = ops.bf16_head_bwd(x, w, y, dout.contiguous())
        return dx, dwb[:8].reshape(ctx.wshape), dwb[8:], None


class Select0Bf16Fn(torch.autograd.Function):
    """Channel 0 of a bf16 channel-last tensor as fp32 (``[...,8]`` -> ``[...]``): the one real output of CostRegNet's 8 -> 1 ``prob``
    convolution, which runs zero-padded to 8 output channels; backward fills channel 0 and zeros the padding."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.bf16_head_fwd(x, None, None, False)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return ops.bf16_head_bwd(x, None, None, dout.contiguous())[0]


class StagePack:
    """Every bf16 weight layout one StageNet's training step needs (forward + data gradient of the regularizer's nine layers, its
    3x3x3 ``prob`` if it has one, the visibility CNN's three 2-D layers), packed by ONE launch per step instead of one per layer.
    ``run()`` stamps each conv module with ``_mvs_packed = (weight version, forward layout, data-gradient layout)``;
    :func:`packed_of` hands the pair to the layer if the weight has not changed since."""

    def __init__(self, stage):
        jobs, self.slots = [], []

        def add(holder, weight, fwd, bwd):
            w = weight.detach()
            self.slots.append((holder, weight, len(jobs), len(jobs) + 1 if bwd else None))
            jobs.append((w,) + fwd)
            if bwd:
                jobs.append((w,) + bwd)
        reg = stage.cost_reg
        for name in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6"):
            conv = getattr(reg, name).conv
            cout, cin = conv.weight.shape[:2]
            add(conv, conv.weight, (0, cin, cout), (2 if conv.stride[1] == 1 else 1, cout, cin))
        for name in ("conv7", "conv9", "conv11"):
            layer = getattr(reg, name)
            conv = layer.conv if hasattr(layer, "conv") else layer[0]
            cin, cout = conv.weight.shape[:2]
            add(conv, conv.weight, (1, cin, cout), (0, cout, cin))
        prob = getattr(reg, "prob", None)
        if prob is not None and tuple(prob.weight.shape[2:]) == (3, 3, 3):           # CostRegNet: 8 -> 1 run as 8 -> 8
            add(prob, prob.weight, (0, 8, 8), (2, 8, 8))
        for i in range(3):
            conv = stage.vis[i].conv
            cout, cin = conv.weight.shape[:2]
            cin_map = max(8, cin)
            add(conv, conv.weight, (0, cin_map, cout), (2, cout, cin_map) if i else None)
        for _, w, _, _ in self.slots:
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise ops._lib.MvsHipError("StagePack: fp32 contiguous master weights expected")
        self.table = ops.PackTable(jobs)

    def valid(self) -> bool:
        return self.table.valid()

    def run(self, route: bool = True) -> None:
        """``route``: also route this stage's weights through a flush node of its own (False: the caller - the cascade - has routed the
        weights of all its stages through one, :func:`route_weights`)."""
        outs = self.table.run()
        for holder, w, i, j in self.slots:
            holder._mvs_packed = (w._version, outs[i], outs[j] if j is not None else None)
        if route:
            self.unroute()
            route_weights([self])

    def live(self):
        return [(h, w) for h, w, _, _ in self.slots if w.requires_grad]

    def unroute(self) -> None:
        """Drop the routed weights (they belong to ONE forward's autograd graph); :func:`route_of` falls back to the parameters."""
        for holder, _, _, _ in self.slots:
            if getattr(holder, "_mvs_wroute", None) is not None:
                holder._mvs_wroute = None


def route_weights(packs) -> bool:
    """Route the weights of the given :class:`StagePack` s through ONE :class:`WgradFlushFn`: their gradients are computed together
    (ops.bf16_wgrad_group) when the last of them has been queued - the end of the stage's / the cascade's backward."""
    if not (_wgrad_group_on() and torch.is_grad_enabled()):
        return False
    live = [hw for p in packs for hw in p.live()]
    if not live:
        return False
    queue = WgradQueue()
    for (h, _), r in zip(live, WgradFlushFn.apply(queue, *[w for _, w in live])):
        r._mvs_wq = queue
        h._mvs_wroute = r
    return True


def packed_of(conv):
    p = getattr(conv, "_mvs_packed", None)
    return (p[1], p[2]) if p is not None and p[0] == conv.weight._version else None
