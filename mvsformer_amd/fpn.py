"""``FPNEncoder`` / ``FPNDecoder`` with the reference's interface (models/module.py:208-270) on the MI355X path — the steps that hand
the four feature maps to the plane sweeps.

Same constructor argument, same parameter / buffer names (``out0.0.weight``, ``out0.1.running_mean``, ``inner1.bias`` ...), so
``load_state_dict`` of a reference checkpoint's ``decoder.*`` keys works unchanged.  ``forward`` is different code: one HIP
kernel per level (``csrc/fpn.hip``) fuses the bilinear x2 upsampling, the lateral 1x1 convolution, the 3x3 output convolution,
eval-mode BatchNorm and Swish; the full-resolution 64-channel ``intra_feat`` of the reference is never written.

The returned maps have the reference's logical shape ``[N,C,H,W]`` but CHANNEL-LAST memory (they are ``permute`` views of
``[N,H,W,C]`` buffers), which is what the sweeps gather from: ``ops.to_channels_last`` passes such a tensor through without a
copy, so the reference's ``features['stageK'] = feat.reshape(B,V,C,H,W)`` hand-over costs nothing.

``FPNEncoder`` is eleven fused conv + BatchNorm + leaky-ReLU launches (``csrc/conv2d.hip``), NCHW like the reference, so its
outputs (and a ViT branch added to ``conv31``, mvsformer_model.py:229) feed the decoder unchanged.

Training mode (``module.train()``; what ``TwinMVSNet`` / ``DINOMVSNet`` do with their FPN, models/mvsformer_model.py:229-233): every
layer is raw convolution -> batch-statistics BatchNorm -> activation through autograd Functions whose forward AND backward are HIP
kernels - the convolutions (any of the encoder's 7x7 / 5x5 / 3x3, stride 1 / 2, and the decoder's 1x1 / 3x3) as split-form GEMMs with an
implicit patch matrix (``mvs_conv2d_gemm_x3``: forward, data gradient, split-K weight gradient), BatchNorm + leaky-ReLU / Swish through
the fp32 BatchNorm kernels of ``csrc/train.hip`` (SyncBatchNorm statistics ride the same all-reduce as in the regularizer), the decoder's
bilinear x2 upsampling + lateral add and its adjoint in ``csrc/fpn_train.hip``.  fp32 NCHW in and out, like the reference's modules.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib, ops
from .module import _publish_cache, _versions

ACT_LRELU, ACT_SWISH = 2, 3          # activation codes of the BatchNorm kernels (csrc/train.hip)


class Conv2dFn(torch.autograd.Function):
    """Raw 2-D convolution (no bias), fp32 NCHW; forward, data gradient and weight gradient are split-form GEMMs (csrc/vit.hip)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        x = x.to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        ctx.save_for_backward(x, w)
        ctx.cfg = (int(stride), int(pad))
        return ops.conv2d_fwd_x3(x, w, int(stride), int(pad))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.cfg
        dy = dy.contiguous()
        dx = ops.conv2d_dgrad_x3(dy, w, stride, pad, x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = ops.conv2d_wgrad_x3(dy, x, w.shape[2], stride, pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


class BiasFn(torch.autograd.Function):
    """``x + bias[c]`` (the decoder's convolutions have a bias in front of their BatchNorm); the bias gradient is the per-channel sum the
    BatchNorm statistics kernel already computes."""

    @staticmethod
    def forward(ctx, x, bias):
        b = bias.detach().to(torch.float32).contiguous()
        return ops.affine_act(x.contiguous(), torch.ones_like(b), b, None, 0)

    @staticmethod
    def backward(ctx, dy):
        C = dy.shape[1]
        return dy, (ops.bn_stats(dy.contiguous())[:C] if ctx.needs_input_grad[1] else None)


class UpsampleAddFn(torch.autograd.Function):
    """``F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True) + lateral`` (models/module.py:261,264,267)."""

    @staticmethod
    def forward(ctx, x, lateral):
        return ops.upsample2x_add(x.contiguous(), lateral.contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return (ops.upsample2x_bwd(dy) if ctx.needs_input_grad[0] else None), (dy if ctx.needs_input_grad[1] else None)


def _train_layer(x, conv: nn.Conv2d, bn, act: int):
    """conv (+ bias) -> batch-statistics BatchNorm -> activation, all autograd-tracked HIP ops."""
    from .autograd import BnActFn
    y = Conv2dFn.apply(x, conv.weight, conv.stride[0], conv.padding[0])
    if conv.bias is not None:
        y = BiasFn.apply(y, conv.bias)
    return y if bn is None else BnActFn.apply(y, bn.weight, bn.bias, None, bn, act)


ACT_SWISH_GEMM, ACT_RELU_GEMM = 2, 3      # activation codes of mvs_gemm_x3's epilogue (include/mvs_hip.h)


class Swish(nn.Module):
    """Parameterless placeholder (models/module.py:200-206) so that ``nn.Sequential`` indices match the reference's keys; the
    activation itself runs in the kernels' epilogues."""

    def forward(self, x):
        raise _lib.MvsHipError("Swish is a structural placeholder of FPNDecoder: call the decoder")


def _channels_last(t: torch.Tensor) -> torch.Tensor:
    """``[N,C,H,W]`` (logical) -> a contiguous ``[N,H,W,C]`` tensor: free when the producer already wrote channel-last memory (the level kernels'
    ``intra`` with ``intra_nhwc``, the encoder's ``conv01`` companion, any ``permute`` view of an NHWC buffer), one copy otherwise."""
    tag = getattr(t, "_mvs_nhwc", None)
    if tag is not None and tag[1] == t._version and tag[0].shape == (t.shape[0], t.shape[2], t.shape[3], t.shape[1]):
        return tag[0]                                        # (a companion is dropped as soon as its tensor was written to)
    return t.float().permute(0, 2, 3, 1).contiguous()


class FPNDecoder(nn.Module):
    def __init__(self, feat_chs):
        super().__init__()
        feat_chs = list(feat_chs)
        if feat_chs != [8, 16, 32, 64]:
            raise _lib.MvsHipError("FPNDecoder: the HIP path is built for feat_chs=[8,16,32,64] (every shipped config), got %s" % feat_chs)
        final_ch = feat_chs[-1]
        self.out0 = nn.Sequential(nn.Conv2d(final_ch, feat_chs[3], kernel_size=1), nn.BatchNorm2d(feat_chs[3]), Swish())
        self.inner1 = nn.Conv2d(feat_chs[2], final_ch, 1)
        self.out1 = nn.Sequential(nn.Conv2d(final_ch, feat_chs[2], kernel_size=3, padding=1), nn.BatchNorm2d(feat_chs[2]), Swish())
        self.inner2 = nn.Conv2d(feat_chs[1], final_ch, 1)
        self.out2 = nn.Sequential(nn.Conv2d(final_ch, feat_chs[1], kernel_size=3, padding=1), nn.BatchNorm2d(feat_chs[1]), Swish())
        self.inner3 = nn.Conv2d(feat_chs[0], final_ch, 1)
        self.out3 = nn.Sequential(nn.Conv2d(final_ch, feat_chs[0], kernel_size=3, padding=1), nn.BatchNorm2d(feat_chs[0]), Swish())
        self._cache = None

    @staticmethod
    def _fold(seq: nn.Sequential):
        """BatchNorm2d (eval) + conv bias -> per-channel (scale, shift), in float64 on the way."""
        conv, bn = seq[0], seq[1]
        scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
        shift = bn.bias.detach().double() + (conv.bias.detach().double() - bn.running_mean.double()) * scale
        return scale.float().contiguous(), shift.float().contiguous()

    def _prepared(self):
        key = _versions(self)
        if self._cache is None or self._cache[0] != key:
            levels = []
            for k in (1, 2, 3):
                inner, seq = getattr(self, "inner%d" % k), getattr(self, "out%d" % k)
                scale, shift = self._fold(seq)
                x3 = None
                if k == 3 and os.environ.get("MVS_FPN_X3", "1") != "0":      # the full-resolution level in split form (csrc/fpn_cp.hip, fpn_x3.hip)
                    x3 = ops.fpn_level_x3_prepare(seq[0].weight.detach().contiguous(), inner.weight.detach(), inner.bias.detach(), scale, shift)
                elif os.environ.get("MVS_FPN_X3", "1") != "0":               # levels 1-2: the 3x3 convolution in split form (csrc/fpn_lvl_x3.hip)
                    x3 = ops.fpn_level_x3s_prepare(seq[0].weight.detach().contiguous(), scale)
                levels.append((inner.weight.detach().reshape(ops.FPN_CH // 2, 2, -1).permute(0, 2, 1).contiguous(), inner.bias.detach().contiguous(),
                               ops.fpn_pack_weights(seq[0].weight.detach().contiguous()), scale, shift, x3))
            s0, h0 = self._fold(self.out0)
            _publish_cache()
            self._cache = (key, (self.out0[0].weight.detach().reshape(ops.FPN_CH, ops.FPN_CH).contiguous(), s0, h0), levels)
        return self._cache[1], self._cache[2]

    def _forward_train(self, conv01, conv11, conv21, conv31):
        """models/module.py:257-270 with batch statistics; NCHW fp32 outputs carrying the autograd graph."""
        intra = conv31.to(torch.float32)
        outs = [_train_layer(intra, self.out0[0], self.out0[1], ACT_SWISH)]
        for k, lateral in ((1, conv21), (2, conv11), (3, conv01)):
            inner, seq = getattr(self, "inner%d" % k), getattr(self, "out%d" % k)
            intra = UpsampleAddFn.apply(intra, _train_layer(lateral.to(torch.float32), inner, None, 0))
            outs.append(_train_layer(intra, seq[0], seq[1], ACT_SWISH))
        return outs

    def forward(self, conv01, conv11, conv21, conv31):
        if self.training:
            return self._forward_train(conv01, conv11, conv21, conv31)
        with torch.no_grad():
            (w0, s0, h0), levels = self._prepared()
            intra = conv31.float().contiguous()
            outs = [ops.fpn_out0(intra, w0, s0, h0)]
            for i, lateral in enumerate((conv21, conv11, conv01)):
                w_in, b_in, packed, scale, shift, x3 = levels[i]
                if i == 2 and x3 is not None:                # MVS_FPN_X3: 1 (default) = csrc/fpn_cp.hip, strip = csrc/fpn_x3.hip, 0 = csrc/fpn.hip
                    prepared, shift_x, border, prepared_cp = x3
                    if os.environ.get("MVS_FPN_X3", "1") == "strip":
                        out = ops.fpn_level_x3(intra, lateral.float().contiguous(), prepared, shift_x, border)
                    else:
                        out = ops.fpn_level_cp(_channels_last(intra), _channels_last(lateral), prepared_cp, shift_x, border)
                else:
                    # the level below the full-resolution one hands its intra map over channel-last when csrc/fpn_cp.hip will read it
                    nhwc = i == 1 and levels[2][5] is not None and os.environ.get("MVS_FPN_X3", "1") != "strip"
                    if x3 is not None:
                        intra, out = ops.fpn_level_x3s(intra, lateral.float().contiguous(), w_in, b_in, x3, shift, want_intra=True, intra_nhwc=nhwc)
                    else:
                        intra, out = ops.fpn_level(intra, lateral.float().contiguous(), w_in, b_in, packed, scale, shift, want_intra=(i < 2), intra_nhwc=nhwc)
                    if nhwc:                                 # logical NCHW view over the channel-last memory, the memory itself riding along
                        cl = intra
                        intra = cl.permute(0, 3, 1, 2)
                        intra._mvs_nhwc = (cl, intra._version)
                outs.append(out)
        return [o.permute(0, 3, 1, 2) for o in outs]


class FPNDecoderV2(nn.Module):
    """models/module.py:273-302 (the decoder of ``TwinMVSNet``): four 3x3 Conv + BatchNorm + Swish outputs, the upper three reading the
    concatenation [decoder map | transformer map], joined by ConvTranspose2d(4, 2, 1) + BatchNorm + ReLU upsamplings with the encoder
    maps added.  Eval mode on the split-form implicit GEMMs of csrc/vit.hip (``mvs_gemm_x3``: 3x3 convolution = A rows gathered from the
    channel-last map, the transposed convolution = four output-parity classes of 2x2 taps; BatchNorm folded, activation in the epilogue):
    fp32-equivalent.  ``forward(conv01, conv11, conv21, conv31, vit1, vit2, vit3) -> [out1 (1/8), out2 (1/4), out3 (1/2), out4 (full)]``,
    logical NCHW over channel-last memory.  Training mode (``_forward_train``): batch-statistics BatchNorm, every gradient, fp32 NCHW.  (Twins
    itself cannot be pinned here: models/gvt.py needs timm.)"""

    def __init__(self, feat_chs):
        super().__init__()
        c = list(feat_chs)
        if any(ch % 8 for ch in c) or len(c) != 4:
            raise _lib.MvsHipError("FPNDecoderV2: four channel counts, multiples of 8 (got %s)" % (c,))
        self.out1 = nn.Sequential(nn.Conv2d(c[3] * 2, c[3], kernel_size=3, padding=1), nn.BatchNorm2d(c[3]), Swish())
        self.upsample1 = nn.Sequential(nn.ConvTranspose2d(c[3], c[2], kernel_size=4, stride=2, padding=1), nn.BatchNorm2d(c[2]), nn.ReLU(True))
        self.out2 = nn.Sequential(nn.Conv2d(c[2] * 2, c[2], kernel_size=3, padding=1), nn.BatchNorm2d(c[2]), Swish())
        self.upsample2 = nn.Sequential(nn.ConvTranspose2d(c[2], c[1], kernel_size=4, stride=2, padding=1), nn.BatchNorm2d(c[1]), nn.ReLU(True))
        self.out3 = nn.Sequential(nn.Conv2d(c[1] * 2, c[1], kernel_size=3, padding=1), nn.BatchNorm2d(c[1]), Swish())
        self.upsample3 = nn.Sequential(nn.ConvTranspose2d(c[1], c[0], kernel_size=4, stride=2, padding=1), nn.BatchNorm2d(c[0]), nn.ReLU(True))
        self.out4 = nn.Sequential(nn.Conv2d(c[0], c[0], kernel_size=3, padding=1), nn.BatchNorm2d(c[0]), Swish())
        self._cache = None

    def _prepared(self):
        from .vit import _conv3_matrix, _convT_matrices, _fold
        key = _versions(self)
        if self._cache is None or self._cache[0] != key:
            prep = {}
            for k in (1, 2, 3, 4):
                seq = getattr(self, "out%d" % k)
                prep["out%d" % k] = (_conv3_matrix(seq[0].weight, seq[0].in_channels), _fold(seq[0], seq[1]))
            for k in (1, 2, 3):
                seq = getattr(self, "upsample%d" % k)
                prep["up%d" % k] = (_convT_matrices(seq[0].weight), _fold(seq[0], seq[1]))
            _publish_cache()
            self._cache = (key, prep)
        return self._cache[1]

    @staticmethod
    def _conv3(x_cl, wm, fold):
        """3x3 convolution (padding 1) + folded BatchNorm + Swish on a channel-last map ``[B,h,w,C]`` -> ``[B,h,w,Cout]``."""
        B, h, w, C = x_cl.shape
        cout = wm.shape[0]
        out = torch.empty(B, h, w, cout, device=x_cl.device, dtype=torch.float32)
        ops.gemm_x3(x_cl, wm, out, h * w, cout, 9 * C, 0, 9 * C, cout, nb1=B, sA=(h * w * C, 0), sC=(h * w * cout, 0), a_mode=1, H=h, W=w, Cp=C,
                    scale=fold[0], shift=fold[1], act=ACT_SWISH_GEMM)
        return out

    def _forward_train(self, conv01, conv11, conv21, conv31, vit1, vit2, vit3):
        """models/module.py:290-302 with batch statistics: autograd-tracked HIP ops, fp32 NCHW (as the FPNDecoder's training path): 3x3
        convolutions through ``Conv2dFn``, the transposed ones through ``vit.ConvT2dFn``, BatchNorm + Swish / ReLU (+ the encoder map as the
        residual) through ``BnActFn``; ``torch.cat`` only moves data."""
        from .autograd import BnActFn
        from .vit import ConvT2dFn

        def out(x, seq):
            return BnActFn.apply(BiasFn.apply(Conv2dFn.apply(x, seq[0].weight, 1, 1), seq[0].bias), seq[1].weight, seq[1].bias, None, seq[1], ACT_SWISH)

        def up(x, seq, skip):                                 # ReLU(BatchNorm(ConvTranspose2d(x))) + skip
            y = BiasFn.apply(ConvT2dFn.apply(x, seq[0].weight, 2, 1), seq[0].bias)
            return BnActFn.apply(y, seq[1].weight, seq[1].bias, skip.to(torch.float32), seq[1], 1)

        f = lambda t: t.to(torch.float32)
        out1 = out(torch.cat([f(conv31), f(vit1)], dim=1), self.out1)
        out2 = out(torch.cat([up(out1, self.upsample1, conv21), f(vit2)], dim=1), self.out2)
        out3 = out(torch.cat([up(out2, self.upsample2, conv11), f(vit3)], dim=1), self.out3)
        out4 = out(up(out3, self.upsample3, conv01), self.out4)
        return [out1, out2, out3, out4]

    def forward(self, conv01, conv11, conv21, conv31, vit1, vit2, vit3):
        if self.training:
            return self._forward_train(conv01, conv11, conv21, conv31, vit1, vit2, vit3)
        from .vit import VITDecoderStage4Single
        p = self._prepared()

        def cl(t):                                            # logical NCHW -> channel-last memory (free when it already is)
            return t.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()

        with torch.no_grad():
            outs = []
            x = torch.cat([cl(conv31), cl(vit1)], dim=-1)
            for k, (skip, vit) in enumerate(((conv21, vit2), (conv11, vit3), (conv01, None)), start=1):
                out = self._conv3(x, *p["out%d" % k])
                outs.append(out)
                up = VITDecoderStage4Single._up(out, p["up%d" % k][0], p["up%d" % k][1], ACT_RELU_GEMM)      # [B,2h,2w,C/2]
                s = cl(skip)
                if vit is None:
                    x = up.add_(s)
                else:                                         # [up + skip | vit] written straight into the next convolution's input
                    C = up.shape[-1]
                    x = torch.empty(up.shape[:-1] + (2 * C,), device=up.device, dtype=torch.float32)
                    torch.add(up, s, out=x[..., :C])
                    x[..., C:] = cl(vit)
            outs.append(self._conv3(x, *p["out4"]))
        return [o.permute(0, 3, 1, 2) for o in outs]


class Conv2d(nn.Module):
    """Parameter holder with the reference's attribute names (models/module.py:40-73: ``conv`` without bias, ``bn``); the
    arithmetic runs in ``FPNEncoder.forward``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, norm_type="BN"):
        super().__init__()
        if norm_type != "BN" or padding != kernel_size // 2:
            raise _lib.MvsHipError("Conv2d: the HIP path is built for norm_type='BN' and padding = kernel_size // 2 (FPNEncoder's layers)")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels, momentum=0.1)
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x):
        raise _lib.MvsHipError("Conv2d is a parameter holder of FPNEncoder: call the encoder")


class FPNEncoder(nn.Module):
    """models/module.py:208-240.  ``forward(x [N,3,H,W]) -> [conv01 (full res), conv11 (1/2), conv21 (1/4), conv31 (1/8)]``."""

    LAYERS = (("conv00", 7, 1), ("conv01", 5, 1), ("downsample1", 5, 2), ("conv10", 3, 1), ("conv11", 3, 1), ("downsample2", 5, 2),
              ("conv20", 3, 1), ("conv21", 3, 1), ("downsample3", 3, 2), ("conv30", 3, 1), ("conv31", 3, 1))

    def __init__(self, feat_chs, norm_type="BN"):
        super().__init__()
        feat_chs = list(feat_chs)
        if feat_chs != [8, 16, 32, 64]:
            raise _lib.MvsHipError("FPNEncoder: the HIP path is built for feat_chs=[8,16,32,64] (every shipped config), got %s" % feat_chs)
        c = feat_chs
        self.conv00 = Conv2d(3, c[0], 7, 1, padding=3, norm_type=norm_type)
        self.conv01 = Conv2d(c[0], c[0], 5, 1, padding=2, norm_type=norm_type)
        self.downsample1 = Conv2d(c[0], c[1], 5, stride=2, padding=2, norm_type=norm_type)
        self.conv10 = Conv2d(c[1], c[1], 3, 1, padding=1, norm_type=norm_type)
        self.conv11 = Conv2d(c[1], c[1], 3, 1, padding=1, norm_type=norm_type)
        self.downsample2 = Conv2d(c[1], c[2], 5, stride=2, padding=2, norm_type=norm_type)
        self.conv20 = Conv2d(c[2], c[2], 3, 1, padding=1, norm_type=norm_type)
        self.conv21 = Conv2d(c[2], c[2], 3, 1, padding=1, norm_type=norm_type)
        self.downsample3 = Conv2d(c[2], c[3], 3, stride=2, padding=1, norm_type=norm_type)
        self.conv30 = Conv2d(c[3], c[3], 3, 1, padding=1, norm_type=norm_type)
        self.conv31 = Conv2d(c[3], c[3], 3, 1, padding=1, norm_type=norm_type)
        self._cache = None

    def _prepared(self):
        key = _versions(self)
        if self._cache is None or self._cache[0] != key:
            layers = []
            for name, k, stride in self.LAYERS:
                m = getattr(self, name)
                scale = m.bn.weight.detach().double() / torch.sqrt(m.bn.running_var.double() + m.bn.eps)
                shift = m.bn.bias.detach().double() - m.bn.running_mean.double() * scale
                wt = m.conv.weight.detach().contiguous()
                x3 = None                                    # conv00 / conv01: the split form (csrc/conv2d_x3.hip)
                if os.environ.get("MVS_FPN_X3", "1") != "0" and ops.conv2d_x3_supported(wt.shape[1], wt.shape[0], k, stride):
                    x3 = ops.conv2d_x3_prepare(wt, scale.float().contiguous())
                elif os.environ.get("MVS_FPN_X3", "1") != "0" and ops.conv2d_x3s_supported(wt.shape[1], wt.shape[0], k, stride):
                    x3 = ops.conv2d_x3s_prepare(wt, scale.float().contiguous(), stride)      # the layers below full resolution (csrc/conv2d_x3s.hip)
                layers.append((ops.conv2d_pack_weights(wt), scale.float().contiguous(), shift.float().contiguous(), m.conv.out_channels, k, stride, x3))
            _publish_cache()
            self._cache = (key, layers)
        return self._cache[1]

    def forward(self, x, conv01_channels_last: bool = False):
        """``conv01_channels_last`` (eval): return ``conv01`` as a logical-NCHW VIEW of its channel-last buffer and skip writing the NCHW copy no kernel
        of the eval path reads (downsample1 and the decoder's last level take the channel-last map) - what ``DINOMVSNet`` asks for."""
        if self.training:                                    # models/module.py:226-240 with batch statistics
            outs = {}
            x = x.to(torch.float32)
            for name, _, _ in self.LAYERS:
                m = getattr(self, name)
                x = _train_layer(x, m.conv, m.bn, ACT_LRELU)
                outs[name] = x
            return [outs["conv01"], outs["conv11"], outs["conv21"], outs["conv31"]]
        with torch.no_grad():
            x = x.float().contiguous()
            outs = {}
            for (name, _, _), (packed, scale, shift, cout, k, stride, x3) in zip(self.LAYERS, self._prepared()):
                if x3 is not None and name == "conv00":     # conv00 -> conv01 channel-last: 16-byte stores / loads instead of scattered dwords
                    x00 = ops.conv2d_x3_bn_lrelu(x, x3, shift, cout, k, 0.1, out="nhwc")
                    outs[name] = x00.permute(0, 3, 1, 2)
                    continue
                if x3 is not None and name == "conv01":     # + the channel-last companion the decoder's last level stages with 16-byte loads
                    if conv01_channels_last:
                        cl = ops.conv2d_x3_bn_lrelu(x00, x3, shift, cout, k, 0.1, x_nhwc=True, out="nhwc")
                        x = cl.permute(0, 3, 1, 2)
                    else:
                        x, cl = ops.conv2d_x3_bn_lrelu(x00, x3, shift, cout, k, 0.1, x_nhwc=True, out="both")
                    x._mvs_nhwc = (cl, x._version)
                elif x3 is not None:                        # (downsample1 reads conv01's channel-last companion)
                    cl = getattr(x, "_mvs_nhwc", None) if name == "downsample1" else None
                    if cl is not None and cl[1] == x._version:
                        x = ops.conv2d_x3s_bn_lrelu(cl[0], x3, shift, cout, k, stride, 0.1, x_nhwc=True)
                    else:
                        x = ops.conv2d_x3s_bn_lrelu(x, x3, shift, cout, k, stride, 0.1)
                else:
                    x = ops.conv2d_bn_lrelu(x, packed, scale, shift, cout, k, stride, 0.1)
                outs[name] = x
        return [outs["conv01"], outs["conv11"], outs["conv21"], outs["conv31"]]
