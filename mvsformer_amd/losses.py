"""Training losses of the cascade on the HIP path — the reference's ``models/losses.py`` interface: ``ce_loss_stage4`` (every shipped
config), ``mixup_ce_loss_stage4`` (``depth_type='mixup_ce'``), ``reg_loss_stage4`` (``depth_type='regression'``) and ``wasserstein_loss``
(``depth_type='was'``: Sinkhorn iterations on a D x D plan per pixel, losses.py:88-162, one wavefront per pixel, the gradient through all
iterations).

``ce_loss_stage4`` (losses.py:304-350) keeps its name, arguments and return value (dict stage -> weighted scalar loss).
Per stage ONE kernel finds every pixel's ground-truth depth bin in the (flipped) hypothesis column, applies the range
and validity masks, evaluates the cross entropy of ``prob_volume_pre`` and leaves ``softmax - onehot`` behind as the
gradient, so neither the flipped copies, the ``[B,D,H,W]`` interval tensors, the gathered ``[N,D]`` matrix nor a separate
log-softmax backward are materialized.
"""
from __future__ import annotations

import torch

from . import ops


class CeLossFn(torch.autograd.Function):
    """``(prob_volume_pre [B,D,H,W], depth_values [B,D,H,W], depth_gt [B,H,W], mask [B,H,W]) -> scalar loss``."""

    @staticmethod
    def forward(ctx, logits, depth_values, depth_gt, mask, inverse_depth, weight):
        need_grad = bool(ctx.needs_input_grad[0])
        loss, acc, grad = ops.ce_loss(logits.contiguous(), depth_values.contiguous(), depth_gt.contiguous(), mask.contiguous(),
                                      bool(inverse_depth), float(weight), want_grad=need_grad)
        ctx.weight, ctx.has_grad = float(weight), need_grad
        ctx.save_for_backward(acc, grad if need_grad else acc)
        return loss

    @staticmethod
    def backward(ctx, gout):
        acc, grad = ctx.saved_tensors
        if not ctx.has_grad:
            return None, None, None, None, None, None
        # out of place: the saved buffer stays unscaled, a second backward (retain_graph) scales it again
        return ops.ce_loss_bwd_scale(grad, acc, gout.contiguous().reshape(1), ctx.weight), None, None, None, None, None


def ce_loss_stage4(inputs, depth_gt_ms, mask_ms, dlossw, focal=False, gamma=0.0, inverse_depth=True):
    """losses.py:304-350.  ``inputs[stage]`` holds ``depth_values`` and ``prob_volume_pre`` (the StageNet outputs);
    ``mask_ms[stage]`` is thresholded at 0.5 inside the kernel."""
    if focal:
        raise NotImplementedError("focal=True is not on the HIP path (no shipped config enables it; the reference's focal "
                                  "branch also skips the validity mask and the reduction)")
    out = {}
    for key in ("stage1", "stage2", "stage3", "stage4"):
        st = inputs[key]
        w = 1.0 if dlossw is None else float(dlossw[int(key.replace("stage", "")) - 1])
        out[key] = CeLossFn.apply(st["prob_volume_pre"].to(torch.float32), st["depth_values"].to(torch.float32),
                                  depth_gt_ms[key].to(torch.float32), mask_ms[key].to(torch.float32), inverse_depth, w)
    return out


class MixupCeLossFn(torch.autograd.Function):
    """losses.py:353-408 for one stage: ``(prob_volume_pre [B,D,H,W], depth_values, depth_gt [B,H,W], mask) -> scalar``; one launch finds
    the ground-truth interval, the two (D-1)-way cross entropies and their mix weights and leaves the unnormalized gradient behind."""

    @staticmethod
    def forward(ctx, logits, depth_values, depth_gt, mask, inverse_depth, weight):
        need_grad = bool(ctx.needs_input_grad[0])
        loss, acc, grad = ops.mixup_ce_loss(logits.contiguous(), depth_values.contiguous(), depth_gt.contiguous(), mask.contiguous(),
                                            bool(inverse_depth), float(weight), want_grad=need_grad)
        ctx.weight, ctx.has_grad = float(weight), need_grad
        ctx.save_for_backward(acc, grad if need_grad else acc)
        return loss

    @staticmethod
    def backward(ctx, gout):
        acc, grad = ctx.saved_tensors
        if not ctx.has_grad:
            return None, None, None, None, None, None
        return ops.ce_loss_bwd_scale(grad, acc, gout.contiguous().reshape(1), ctx.weight), None, None, None, None, None


def mixup_ce_loss_stage4(inputs, depth_gt_ms, mask_ms, dlossw, inverse_depth=True):
    """losses.py:353-408; same dict-in / dict-out contract as :func:`ce_loss_stage4`."""
    out = {}
    for key in ("stage1", "stage2", "stage3", "stage4"):
        st = inputs[key]
        w = 1.0 if dlossw is None else float(dlossw[int(key.replace("stage", "")) - 1])
        out[key] = MixupCeLossFn.apply(st["prob_volume_pre"].to(torch.float32), st["depth_values"].to(torch.float32),
                                       depth_gt_ms[key].to(torch.float32), mask_ms[key].to(torch.float32), inverse_depth, w)
    return out


class RegLossFn(torch.autograd.Function):
    """losses.py:51-85 for one stage: ``(depth [B,H,W], depth_gt, mask, depth_interval [B], depth_values or None) -> scalar``."""

    @staticmethod
    def forward(ctx, depth, depth_gt, mask, interval, depth_values, inverse_depth, weight):
        need_grad = bool(ctx.needs_input_grad[0])
        loss, acc, grad = ops.reg_loss(depth.contiguous(), depth_gt.contiguous(), mask.contiguous(), interval.contiguous(),
                                       None if depth_values is None else depth_values.contiguous(), bool(inverse_depth), float(weight),
                                       want_grad=need_grad)
        ctx.weight, ctx.has_grad = float(weight), need_grad
        ctx.save_for_backward(acc, grad if need_grad else acc)
        return loss

    @staticmethod
    def backward(ctx, gout):
        acc, grad = ctx.saved_tensors
        if not ctx.has_grad:
            return (None,) * 7
        return (ops.ce_loss_bwd_scale(grad, acc, gout.contiguous().reshape(1), ctx.weight),) + (None,) * 6


def reg_loss_stage4(inputs, depth_gt_ms, mask_ms, dlossw, depth_interval, mask_out_range=False, inverse_depth=True):
    """losses.py:51-85: masked smooth-L1 of ``inputs[stage]['depth'] / depth_interval`` against the ground truth, per stage."""
    out = {}
    itv = depth_interval.detach().to(torch.float32).reshape(-1)
    for key in ("stage1", "stage2", "stage3", "stage4"):
        st = inputs[key]
        w = 1.0 if dlossw is None else float(dlossw[int(key.replace("stage", "")) - 1])
        dv = st["depth_values"].detach().to(torch.float32) if mask_out_range else None
        out[key] = RegLossFn.apply(st["depth"].to(torch.float32), depth_gt_ms[key].to(torch.float32), mask_ms[key].to(torch.float32), itv, dv,
                                   inverse_depth, w)
    return out


class WasLossFn(torch.autograd.Function):
    """losses.py:88-128 for one stage: ``(prob_volume [B,D,H,W], depth_values, depth_gt [B,H,W], mask) -> scalar`` (discrete Sinkhorn)."""

    @staticmethod
    def forward(ctx, prob, depth_values, depth_gt, mask, ot_iter, ot_eps, weight):
        need_grad = bool(ctx.needs_input_grad[0])
        loss, acc, grad = ops.was_loss(prob.contiguous(), depth_values.contiguous(), depth_gt.contiguous(), mask.contiguous(), int(ot_iter), float(ot_eps),
                                       float(weight), want_grad=need_grad)
        ctx.weight, ctx.has_grad = float(weight), need_grad
        ctx.save_for_backward(acc, grad if need_grad else acc)
        return loss

    @staticmethod
    def backward(ctx, gout):
        acc, grad = ctx.saved_tensors
        if not ctx.has_grad:
            return (None,) * 7
        return (ops.ce_loss_bwd_scale(grad, acc, gout.contiguous().reshape(1), ctx.weight),) + (None,) * 6


def wasserstein_loss(inputs, depth_gt_ms, mask_ms, dlossw, ot_iter=10, ot_eps=1, ot_continous=False, inverse=True):
    """losses.py:131-162: per stage ``dlossw[i] * sinkhorn(depth_gt, depth_values, prob_volume, mask > 0.5, ot_iter, ot_eps)[1]``.  Only the
    discrete plan (``ot_continous=False``, what the trainer passes, mvsformer_trainer.py:114-117) is built."""
    if ot_continous:
        raise ops._lib.MvsHipError("wasserstein_loss: ot_continous=True is not built (the reference's trainer always passes False)")
    out = {}
    for i, key in enumerate(k for k in inputs.keys() if "stage" in k):
        st = inputs[key]
        out[key] = WasLossFn.apply(st["prob_volume"].to(torch.float32), st["depth_values"].detach().to(torch.float32), depth_gt_ms[key].to(torch.float32),
                                   mask_ms[key].to(torch.float32), ot_iter, ot_eps, float(dlossw[i]))
    return out
