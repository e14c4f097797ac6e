"""Host-side mirror of the hot-path symbols of the reference's ``models/module.py``.

Same class names, constructor signatures, forward signatures and ``state_dict`` keys as the reference
(models/module.py:83-165 ``Conv3d``/``Deconv3d``, :168-197 ``ConvBnReLU``, :469-505 ``CostRegNet``, :550-594
``CostRegNet3D``, :597-619 ``depth_regression``/``conf_regression``, :633-653 the inverse-depth schedulers), so a
reference checkpoint loads with ``strict=True`` and ``models/mvsformer_model.py`` can use them unchanged.  The
``nn.Conv3d`` / ``nn.BatchNorm3d`` children are parameter holders only (they keep ``.to()``, ``state_dict()``,
DDP and SyncBatchNorm conversion working); every forward runs the hand-written HIP kernels of
``libmvs_hip.so`` through :mod:`mvsformer_amd.ops`.  There is no PyTorch/CPU fallback.

Eval-mode BatchNorm is folded into a per-channel ``scale``/``shift`` pair applied in the conv epilogue;
folded parameters and the MFMA-friendly weight packing are cached and rebuilt when any parameter changes.
Training mode (``module.train()``) runs raw conv -> batch-statistics BatchNorm -> ReLU (+ skip) through the autograd
functions of :mod:`mvsformer_amd.autograd`, whose forward and backward are HIP kernels as well.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import MvsHipError


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _bn_fold(bn: nn.modules.batchnorm._BatchNorm) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval BatchNorm as y = x*scale + shift (fp32, on the module's device)."""
    var = bn.running_var.detach().to(torch.float32)
    mean = bn.running_mean.detach().to(torch.float32)
    g = bn.weight.detach().to(torch.float32) if bn.weight is not None else torch.ones_like(var)
    b = bn.bias.detach().to(torch.float32) if bn.bias is not None else torch.zeros_like(var)
    scale = g / torch.sqrt(var + bn.eps)
    return scale.contiguous(), (b - mean * scale).contiguous()


def _versions(mod: nn.Module) -> tuple:
    """Cache key of everything derived from a module's parameters / buffers: torch's (data_ptr, _version) per tensor + the epoch of
    raw-pointer writes (``ops.bump_weights_epoch``: FusedAdamW, training-mode BatchNorm kernels, replayed hipGraphs)."""
    return (ops.weights_epoch(),) + tuple((t.data_ptr(), t._version) for t in list(mod.parameters()) + list(mod.buffers()))


def _publish_cache() -> None:
    """Packed weights / folded BatchNorm are produced by kernels on the CURRENT stream and then reused by every later
    forward, possibly on other streams (bench.py walks reference views over several).  The cache is rebuilt only when a
    parameter changes, so one stream synchronization per rebuild makes the cached tensors safe to read from any stream."""
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


def autocast_bf16() -> bool:
    """True inside ``torch.autocast('cuda', dtype=torch.bfloat16)`` (how BASELINE configs[2] trains; the reference trainer wraps
    the model in ``torch.cuda.amp.autocast``, trainer/mvsformer_trainer.py:104-106) or with MVS_TRAIN_BF16=1: the regularizer
    then runs on bf16 channel-last activations and the bf16 matrix cores."""
    if os.environ.get("MVS_TRAIN_BF16", "") == "1":
        return True
    try:
        if hasattr(torch, "get_autocast_dtype"):
            return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        return torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16
    except Exception:                                                    # noqa: BLE001 - torch builds without these queries
        return False


def _multi_use(t, link=None):
    """A training-layer output that feeds more than one consumer (a skip connection) or a consumer of another kind (the head): its
    BatchNorm's backward sums cannot be taken by ONE consumer's data gradient (autograd.LayerBf16Fn) - drop the tag that offers it.
    With a ``link`` (autograd.SkipLink) the strided consumer's data gradient WILL be the tensor's total gradient: the tag stays."""
    if link is None and hasattr(t, "_mvs_bn"):
        del t._mvs_bn
    return t


def _skip_links(x, n=3):
    """``n`` SkipLinks for one forward of a U-Net on the fused bf16 training layers, else Nones (MVS_TRAIN_SKIPLINK=0: autograd adds)."""
    from . import autograd as ag
    if x.dtype == torch.bfloat16 and ag._fused_layers() and os.environ.get("MVS_TRAIN_SKIPLINK", "1") != "0":
        return tuple(ag.SkipLink() for _ in range(n))
    return (None,) * n


def _train_conv_bn_act(x, conv, bn, relu, residual, transposed_sd=None, take=None, give=None):
    """Training-mode layer: raw (transposed) conv -> batch-stat BN -> ReLU (+ residual), all autograd-tracked HIP ops.
    bf16 channel-last input (``[B,D,H,W,C]``) selects the bf16 kernels, fp32 ``[B,C,D,H,W]`` the fp32 ones."""
    from . import autograd as ag
    if x.dtype == torch.bfloat16:
        if conv.bias is not None or bn is None:
            raise MvsHipError("bf16 training layer without BatchNorm / with bias is not built")
        # MVS_BN_FUSED_STATS=1: the convolution's epilogue takes the batch statistics of its output instead of a separate pass over y.
        # Measured 13.5 vs 13.3 ms per step (one partial row per wavefront makes the fixed-order reduce long): off by default.
        fused = 1 if os.environ.get("MVS_BN_FUSED_STATS", "0") == "1" else 0
        pk = ag.packed_of(conv)                              # this step's layouts if the stage's StagePack made them (one launch per stage)
        if ag._fused_layers() and not ag._bn_synced(bn):
            # statistics stay on this rank: the whole layer is one autograd node (conv with the statistics in its epilogue -> finalize ->
            # normalize): 8 graph nodes per layer and step instead of 11
            if transposed_sd is None:
                s = tuple(conv.stride)
                return ag.LayerBf16Fn.apply(x, ag.route_of(conv), bn.weight, bn.bias, residual, bn, bool(relu), 0, (s[0], s[1]), 1, pk, take, give)
            return ag.LayerBf16Fn.apply(x, ag.route_of(conv), bn.weight, bn.bias, residual, bn, bool(relu), 1, (transposed_sd, 2), 1, pk, take, give)
        if take is not None:
            _multi_use(x)                                    # no hand-over on the unfused chain: the tag (if any) must not promise one
        if transposed_sd is None:
            s = tuple(conv.stride)
            out = ag.ConvBf16Fn.apply(x, ag.route_of(conv), (s[0], s[1]), fused, pk)
        else:
            out = ag.DeconvBf16Fn.apply(x, ag.route_of(conv), transposed_sd, fused, pk)
        y, sums = out if fused else (out, None)
        return ag.BnActBf16Fn.apply(y, bn.weight, bn.bias, residual, bn, bool(relu), 1, sums)
    if transposed_sd is None:
        s = tuple(conv.stride)
        y = ag.ConvFn.apply(x, conv.weight, (s[0], s[1]))
    else:
        y = ag.DeconvFn.apply(x, conv.weight, transposed_sd)
    if conv.bias is not None:
        raise MvsHipError("training-mode conv with bias (bn=False) is not built")
    if bn is None:
        raise MvsHipError("training-mode layer without BatchNorm is not built")
    return ag.BnActFn.apply(y, bn.weight, bn.bias, residual, bn, bool(relu))


# ---------------------------------------------------------------------------------------------------------
# layer holders (reference models/module.py:83-165, 168-197)
# ---------------------------------------------------------------------------------------------------------
X3_MIN_VOXELS = 40 * 1024      # output voxels from which the split-form conv is used (MVS_CONV_X3_MIN_VOXELS overrides)
# The small-volume split-form kernel (csrc/conv3d_x3_small.hip) serves a layer while voxels x Cin x Cout stays below these bounds (output voxels
# of a convolution, input voxels of a transposed convolution): measured per layer at the two coarse config-2 stages (profiles/r04_bench_small.txt),
# it wins 4-21 us per launch below them and loses above (its vector ALU bound grows with the volume, the tiled kernels' latency chains do not).
SMALL_MAX_WORK = (16 << 20, 8 << 20)      # (Conv3d, Deconv3d); MVS_CONV_SMALL_MAX_WORK="conv,deconv" overrides, "0,0" = never


def _parse_small_limit(text) -> Tuple[int, int]:
    """``"conv,deconv"`` or one value for both; anything else is a configuration error reported at import, not in the middle of a forward."""
    if not text:
        return SMALL_MAX_WORK
    try:
        v = [int(t) for t in text.split(",")]
    except ValueError:
        v = []
    if len(v) == 1:
        v = v * 2
    if len(v) != 2 or min(v) < 0:
        raise ValueError("MVS_CONV_SMALL_MAX_WORK must be 'conv,deconv' or one non-negative integer, got %r" % (text,))
    return v[0], v[1]


_SMALL_LIMITS = _parse_small_limit(os.environ.get("MVS_CONV_SMALL_MAX_WORK"))      # read once: not on every forward of every layer
_WINDOW = 1 << 31        # bytes a buffer descriptor of the split-form kernels addresses (per sample for conv / deconv, per call for the tail)


def _small_limit(transposed: bool) -> int:
    if os.environ.get("MVS_CONV_X3", "1") == "0":
        return 0
    return _SMALL_LIMITS[1 if transposed else 0]


class Conv3d(nn.Module):
    """conv(bias = not bn) -> BatchNorm3d -> ReLU, as reference ``Conv3d`` (module.py:83-123)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu
        self._cache = None

    def _prepared(self):
        key = _versions(self)
        if self._cache is None or self._cache[0] != key:
            conv = self.conv
            if tuple(conv.kernel_size) != (3, 3, 3) or tuple(conv.padding) != (1, 1, 1) or tuple(conv.dilation) != (1, 1, 1) \
                    or conv.groups != 1:
                raise MvsHipError("Conv3d: only kernel 3, padding 1, dilation 1, groups 1 is built (got %s)" % conv)
            s = tuple(conv.stride)
            if s not in ((1, 1, 1), (2, 2, 2), (1, 2, 2)):
                raise MvsHipError("Conv3d: stride %s is not built" % (s,))
            packed = ops.conv3d_pack(_f32c(conv.weight), transposed=False)
            # Winograd F(2x2,3x3) fp32-MFMA image of the stride-1 layers (conv2 / conv4 / conv6): opt-in for eval (MVS_CONV_WINO=1) because the
            # split form below is as fast at the sizes that matter.  (Round 3 saw its output move from run to run under concurrent streams;
            # round 4 traced that to packed-fp32 instructions reading an SGPR pair's high half - a gfx950 hazard reproduced stand-alone in
            # tools/probe/pk_mfma_race.hip, DESIGN.md 4.7c - and the library is built without packed fp32, so the kernel is reproducible:
            # tests/test_hip_multistream.py[conv_wino].)
            wino = None
            if s == (1, 1, 1) and conv.in_channels % 4 == 0 and conv.out_channels % 16 == 0 and conv.out_channels <= 64 \
                    and os.environ.get("MVS_CONV_WINO", "0") == "1":
                wino = ops.conv3d_wino_pack(_f32c(conv.weight))
            # 3-term bf16 split form (csrc/conv3d_x3.hip: fp32-equivalent, bf16 matrix cores) for every layer shape it is built for
            # (stride (1,1,1) and (1,2,2)); forward() uses it from X3_MIN_VOXELS output voxels up (below that the launch is too small to
            # fill the chip with its 16 x 16 x D tiles and the fp32-MFMA kernel wins).  MVS_CONV_X3=0 turns it off.
            x3 = None
            x3_mode = os.environ.get("MVS_CONV_X3", "1")        # "1" all built shapes, "strided" / "s1" one kind only (diagnostics), "0" off
            if x3_mode != "0" and ops.conv3d_x3_supported(conv.in_channels, conv.out_channels, (s[0], s[1])) \
                    and not (x3_mode == "strided" and s[1] == 1) and not (x3_mode == "s1" and s[1] == 2):
                x3 = ops.conv3d_x3_pack(_f32c(conv.weight), (s[0], s[1]))
            # small-volume split form (csrc/conv3d_x3_small.hip) for stride (1,1,1) / (2,2,2): CostRegNet's inner layers at the coarse stages
            small = None
            if s[0] == s[1] and _small_limit(False) > 0 and ops.conv3d_small_supported(conv.in_channels, conv.out_channels, s[0], False):
                small = ops.conv3d_small_pack(_f32c(conv.weight), s[0], False)
            if self.bn is not None:
                scale, shift = _bn_fold(self.bn)
            else:
                scale = None
                shift = _f32c(conv.bias) if conv.bias is not None else None
            _publish_cache()
            self._cache = (key, packed, scale, shift, (s[0], s[1]), wino, x3, small)
        return self._cache[1:]

    def forward(self, x, residual: Optional[torch.Tensor] = None, take=None):
        if self.training:
            return _train_conv_bn_act(x, self.conv, self.bn, self.relu, residual, take=take)
        packed, scale, shift, stride, wino, x3, small = self._prepared()
        # (a sample beyond the split-form kernel's 2 GiB buffer window falls through to the 64-bit-addressed fp32-MFMA kernel below)
        if x3 is not None and x.shape[2] * (x.shape[3] // stride[1]) * (x.shape[4] // stride[1]) >= int(os.environ.get("MVS_CONV_X3_MIN_VOXELS", X3_MIN_VOXELS)) \
                and x[0].numel() * 4 < _WINDOW:
            return ops.conv3d_x3(x, x3, self.conv.in_channels, self.conv.out_channels, stride, scale, shift, residual, relu=self.relu)
        if small is not None and ((x.shape[2] - 1) // stride[0] + 1) * ((x.shape[3] - 1) // stride[1] + 1) * ((x.shape[4] - 1) // stride[1] + 1) \
                * self.conv.in_channels * self.conv.out_channels <= _small_limit(False):
            return ops.conv3d_small(x, small, self.conv.in_channels, self.conv.out_channels, stride[0], False, scale, shift, residual, relu=self.relu)
        if wino is not None and ops.conv3d_wino_supported(self.conv.in_channels, self.conv.out_channels, *x.shape[2:]):
            return ops.conv3d_wino(x, wino, self.conv.in_channels, self.conv.out_channels, scale, shift, residual, relu=self.relu)
        return ops.conv3d(x, packed, self.conv.in_channels, self.conv.out_channels, stride, scale, shift, residual,
                          relu=self.relu, tag="conv3d_%dto%d_s%d%d" % (self.conv.in_channels, self.conv.out_channels, *stride))


class Deconv3d(nn.Module):
    """conv_transpose(bias = not bn) -> BatchNorm3d -> ReLU, as reference ``Deconv3d`` (module.py:126-165)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        self.out_channels = out_channels
        self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu
        self._cache = None

    def forward(self, x, residual: Optional[torch.Tensor] = None, give=None):
        if self.training:
            _prepare_deconv_check(self.conv)
            return _train_conv_bn_act(x, self.conv, self.bn, self.relu, residual, transposed_sd=self.conv.stride[0], give=give)
        key = _versions(self)
        if self._cache is None or self._cache[0] != key:
            prepared = _prepare_deconv(self.conv, self.bn)
            cin, cout = self.conv.in_channels, self.conv.out_channels
            small = None
            if tuple(self.conv.stride) == (2, 2, 2) and cout >= 16 and _small_limit(True) > 0 and ops.conv3d_small_supported(cin, cout, 2, True):
                small = ops.conv3d_small_pack(_f32c(self.conv.weight), 2, True)
            _publish_cache()
            self._cache = (key,) + prepared + (small,)
        _, packed, scale, shift, sd, small = self._cache
        if small is not None and x.shape[2] * x.shape[3] * x.shape[4] * self.conv.in_channels * self.conv.out_channels <= _small_limit(True):
            return ops.conv3d_small(x, small, self.conv.in_channels, self.conv.out_channels, 2, True, scale, shift, residual, relu=self.relu)
        return ops.deconv3d(x, packed, self.conv.in_channels, self.conv.out_channels, sd, scale, shift, residual,
                            relu=self.relu, tag="deconv3d_%dto%d_s%d" % (self.conv.in_channels, self.conv.out_channels, sd))


def _prepare_deconv_check(conv: nn.ConvTranspose3d):
    s, op = tuple(conv.stride), tuple(conv.output_padding)
    if tuple(conv.kernel_size) != (3, 3, 3) or tuple(conv.padding) != (1, 1, 1) or conv.groups != 1:
        raise MvsHipError("Deconv3d: only kernel 3, padding 1, groups 1 is built (got %s)" % conv)
    if not ((s == (2, 2, 2) and op == (1, 1, 1)) or (s == (1, 2, 2) and op == (0, 1, 1))):
        raise MvsHipError("Deconv3d: stride %s / output_padding %s is not built" % (s, op))
    return s


def _prepare_deconv(conv: nn.ConvTranspose3d, bn):
    s = _prepare_deconv_check(conv)
    packed = ops.conv3d_pack(_f32c(conv.weight), transposed=True, sd=s[0])
    if bn is not None:
        scale, shift = _bn_fold(bn)
    else:
        scale, shift = None, (_f32c(conv.bias) if conv.bias is not None else None)
    return packed, scale, shift, s[0]


class ConvBnReLU(nn.Module):
    """2-D conv(bias=False) -> BatchNorm2d -> ReLU holder (reference module.py:168-197).  Used only inside
    ``StageNet.vis``, whose four layers run as one fused HIP launch (:func:`pack_vis_params`)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, pad: int = 1, dilation: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        raise MvsHipError("ConvBnReLU is a parameter holder; the fused visibility CNN runs through StageNet (mvs_vis_fwd)")


def pack_vis_params(vis: nn.Sequential) -> torch.Tensor:
    """Flatten ``StageNet.vis`` (ConvBnReLU(1,16), ConvBnReLU(16,16), ConvBnReLU(16,8), Conv2d(8,1,1), Sigmoid) into the
    3689-float parameter block of ``mvs_vis_fwd`` (layout documented in csrc/vis_net.hip)."""
    c0, c1, c2, c3 = vis[0], vis[1], vis[2], vis[3]
    shapes = [tuple(c.conv.weight.shape) for c in (c0, c1, c2)] + [tuple(c3.weight.shape)]
    if shapes != [(16, 1, 3, 3), (16, 16, 3, 3), (8, 16, 3, 3), (1, 8, 1, 1)]:
        raise MvsHipError("vis CNN has unexpected shapes %s" % (shapes,))
    parts = []
    for c in (c0, c1, c2):
        w = _f32c(c.conv.weight)                               # [cout, cin, 3, 3] -> [cin, tap, cout]
        parts.append(w.permute(1, 2, 3, 0).reshape(-1))
        s, b = _bn_fold(c.bn)
        parts += [s, b]
    parts += [_f32c(c3.weight).reshape(-1), _f32c(c3.bias).reshape(-1)]
    out = torch.cat(parts).contiguous()
    assert out.numel() == ops.VIS_PARAM_FLOATS
    return out


# ---------------------------------------------------------------------------------------------------------
# regularizers (reference models/module.py:469-505, 550-594)
# ---------------------------------------------------------------------------------------------------------
class CostRegNet(nn.Module):
    """3-D U-Net with stride-2 down/up-sampling in D, H, W (reference ``CostRegNet``).  ``forward(x[B,Cin,D,H,W])``
    returns ``[B,1,D,H,W]`` logits (``[B,base,D,H,W]`` features if ``last_layer=False``)."""

    def __init__(self, in_channels, base_channels, last_layer=True):
        super().__init__()
        self.last_layer = last_layer
        b = base_channels
        self.conv1 = Conv3d(in_channels, b * 2, stride=2, padding=1)
        self.conv2 = Conv3d(b * 2, b * 2, padding=1)
        self.conv3 = Conv3d(b * 2, b * 4, stride=2, padding=1)
        self.conv4 = Conv3d(b * 4, b * 4, padding=1)
        self.conv5 = Conv3d(b * 4, b * 8, stride=2, padding=1)
        self.conv6 = Conv3d(b * 8, b * 8, padding=1)
        self.conv7 = Deconv3d(b * 8, b * 4, stride=2, padding=1, output_padding=1)
        self.conv9 = Deconv3d(b * 4, b * 2, stride=2, padding=1, output_padding=1)
        self.conv11 = Deconv3d(b * 2, b * 1, stride=2, padding=1, output_padding=1)
        if in_channels != base_channels:
            self.inner = nn.Conv3d(in_channels, base_channels, 1, 1)
        else:
            self.inner = nn.Identity()
        if self.last_layer:
            self.prob = nn.Conv3d(base_channels, 1, 3, stride=1, padding=1, bias=False)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """Everything up to (not including) ``prob``; residual adds are fused into the deconv epilogues."""
        if not isinstance(self.inner, nn.Identity):
            raise MvsHipError("CostRegNet: in_channels != base_channels (1x1x1 'inner' conv) is not built")
        pre16 = self.training and x.dtype == torch.bfloat16       # already bf16 channel-last [B,D,H,W,C] (autograd.AggregateFn as_bf16)
        if not pre16:
            x = x.to(torch.float32)
        x = x if x.is_contiguous() else x.contiguous()
        dhw = x.shape[1:4] if pre16 else x.shape[2:]
        if dhw[0] % 8 or dhw[1] % 8 or dhw[2] % 8:
            raise MvsHipError("CostRegNet needs D, H, W divisible by 8 (three stride-2 levels), got %s" % (tuple(dhw),))
        if self.training and autocast_bf16() and not pre16:
            from . import autograd as ag
            x = ag.ToBf16Fn.apply(x)                     # fp32 cost volume -> bf16 channel-last; every layer below follows the dtype
        if not self.training:
            c2 = self.conv2(self.conv1(x))
            c4 = self.conv4(self.conv3(c2))
            y = self.conv7(self.conv6(self.conv5(c4)), residual=c4)
            return self.conv11(self.conv9(y, residual=c2), residual=x)
        # training: each skip tensor's two gradients meet in the strided convolution's data-gradient epilogue (autograd.SkipLink)
        l0, l2, l4 = _skip_links(x)
        c2 = _multi_use(self.conv2(self.conv1(x, take=l0)), l2)
        c4 = _multi_use(self.conv4(self.conv3(c2, take=l2)), l4)
        y = self.conv6(self.conv5(c4, take=l4))
        y = self.conv7(y, residual=c4, give=l4)
        y = self.conv9(y, residual=c2, give=l2)
        return _multi_use(self.conv11(y, residual=x, give=l0))

    def forward(self, x):
        y = self.features(x)
        if self.last_layer:
            if self.training:
                from . import autograd as ag
                # N = 1 conv embedded in an 8-channel MFMA conv (zero rows) so forward, dgrad and wgrad reuse the conv kernels
                pk = ag.packed_of(self.prob) if y.dtype == torch.bfloat16 else None
                if pk is not None and ag._fused_layers():
                    # the 8 -> 1 parameter packed as an 8 -> 8 map by the stage's StagePack (no padded copy of the weight), channel 0 of
                    # the result straight to fp32
                    y = ag.Select0Bf16Fn.apply(ag.ConvBf16Fn.apply(y, ag.route_of(self.prob), (1, 1), 0, pk, 8)).unsqueeze(1)
                else:
                    w8 = torch.nn.functional.pad(self.prob.weight, (0, 0, 0, 0, 0, 0, 0, 0, 0, 7))
                    if y.dtype == torch.bfloat16:        # autocast: the logits come out of a half-precision conv, then fp32
                        y = ag.FromBf16Fn.apply(ag.ConvBf16Fn.apply(y, w8, (1, 1)))[:, :1]
                    else:
                        y = ag.ConvFn.apply(y, w8, (1, 1))[:, :1]
            else:
                y = ops.prob3(y, _f32c(self.prob.weight)).unsqueeze(1)
        return y


def _deconv_seq(cin, cout):
    return nn.Sequential(
        nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=(0, 1, 1), stride=(1, 2, 2), bias=False),
        nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


class CostRegNet3D(nn.Module):
    """3-D U-Net with stride (1,2,2): depth resolution kept (reference ``CostRegNet3D``).  The decoder layers are
    ``nn.Sequential(ConvTranspose3d, BatchNorm3d, ReLU)`` so the checkpoint keys are ``convN.0.weight`` / ``convN.1.*``."""

    def __init__(self, in_channels, base_channel=8):
        super().__init__()
        b = base_channel
        self.conv1 = Conv3d(in_channels, b * 2, kernel_size=3, stride=(1, 2, 2), padding=1)
        self.conv2 = Conv3d(b * 2, b * 2, padding=1)
        self.conv3 = Conv3d(b * 2, b * 4, kernel_size=3, stride=(1, 2, 2), padding=1)
        self.conv4 = Conv3d(b * 4, b * 4, padding=1)
        self.conv5 = Conv3d(b * 4, b * 8, kernel_size=3, stride=(1, 2, 2), padding=1)
        self.conv6 = Conv3d(b * 8, b * 8, padding=1)
        self.conv7 = _deconv_seq(b * 8, b * 4)
        self.conv9 = _deconv_seq(b * 4, b * 2)
        self.conv11 = _deconv_seq(b * 2, b)
        if in_channels != base_channel:
            self.inner = nn.Conv3d(in_channels, base_channel, 1, 1)
        else:
            self.inner = nn.Identity()
        self.prob = nn.Conv3d(base_channel, 1, 1, stride=1, padding=0)
        self._dcache: Dict[str, tuple] = {}

    def _up(self, name: str, x, residual, give=None):
        seq = getattr(self, name)
        if self.training:
            _prepare_deconv_check(seq[0])
            return _train_conv_bn_act(x, seq[0], seq[1], True, residual, transposed_sd=seq[0].stride[0], give=give)
        key = _versions(seq)
        c = self._dcache.get(name)
        if c is None or c[0] != key:
            c = (key,) + _prepare_deconv(seq[0], seq[1])
            _publish_cache()
            self._dcache[name] = c
        _, packed, scale, shift, sd = c
        cin, cout = seq[0].in_channels, seq[0].out_channels
        # split-form transposed conv (csrc/conv3d_x3.hip) where it beats the fp32-MFMA kernel: conv7 / conv9 at real sizes; conv11
        # (8 output channels fill half a matrix tile, and its fp32 kernel fuses the 1x1x1 prob) stays
        if cout >= 16 and os.environ.get("MVS_CONV_X3", "1") != "0" and ops.deconv3d_x3_supported(cin, cout, sd) and x.shape[4] % 2 == 0 \
                and 4 * x.shape[2] * x.shape[3] * x.shape[4] >= int(os.environ.get("MVS_CONV_X3_MIN_VOXELS", X3_MIN_VOXELS)) \
                and x[0].numel() * 4 < _WINDOW and (residual is None or residual[0].numel() * 4 < 2 * _WINDOW):
            px = self._dcache.get(name + ".x3")
            if px is None or px[0] != key:
                px = (key, ops.deconv3d_x3_pack(_f32c(seq[0].weight), sd))
                _publish_cache()
                self._dcache[name + ".x3"] = px
            return ops.deconv3d_x3(x, px[1], cin, cout, sd, scale, shift, residual, relu=True)
        return ops.deconv3d(x, packed, cin, cout, sd, scale, shift, residual, relu=True, tag="deconv3d_%dto%d_s%d" % (cin, cout, sd))

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        """Eval-mode ``forward`` without the channel axis, ``[B,D,H,W]``: conv11 and the 1x1x1 ``prob`` run as ONE launch
        (the 8-channel volume between them is never written) when the shape allows, else as two."""
        y, skip = self._trunk(x)
        seq = self.conv11
        w, b = self.prob_params()
        if not self.training and seq[0].out_channels == 8 and tuple(seq[0].stride) == (1, 2, 2) and y.shape[4] % 4 == 0 \
                and os.environ.get("MVS_FUSE_PROB", "1") != "0":
            key = _versions(seq)
            c = self._dcache.get("conv11")
            if c is None or c[0] != key:
                c = (key,) + _prepare_deconv(seq[0], seq[1])
                _publish_cache()
                self._dcache["conv11"] = c
            _, packed, scale, shift, _sd = c
            # split form (csrc/tail_x3.hip) for the shape it is built for (16 -> 8) from X3_MIN_VOXELS up; MVS_TAIL=fp32 keeps the fp32-MFMA tail
            # (the tail's window covers the whole batch of the skip volume: a larger call takes the fp32-MFMA tail below)
            if seq[0].in_channels == 16 and os.environ.get("MVS_CONV_X3", "1") != "0" and os.environ.get("MVS_TAIL", "x3") == "x3" \
                    and 4 * y.shape[2] * y.shape[3] * y.shape[4] >= int(os.environ.get("MVS_CONV_X3_MIN_VOXELS", X3_MIN_VOXELS)) \
                    and skip.numel() * 4 < _WINDOW:
                px = self._dcache.get("conv11.x3")
                if px is None or px[0] != key:
                    px = (key, ops.tail_x3_pack(_f32c(seq[0].weight)))
                    _publish_cache()
                    self._dcache["conv11.x3"] = px
                return ops.tail_x3(y, px[1], scale, shift, skip, w, b, relu=True)
            return ops.deconv3d_prob1(y, packed, seq[0].in_channels, scale, shift, skip, w, b, relu=True)
        return ops.prob1(self._up("conv11", y, skip), w, b).squeeze(1)

    def _trunk(self, x: torch.Tensor):
        """Everything up to conv9: returns (conv9 output, skip tensor of conv11 = the input volume)."""
        if not isinstance(self.inner, nn.Identity):
            raise MvsHipError("CostRegNet3D: in_channels != base_channel (1x1x1 'inner' conv) is not built")
        x = x.to(torch.float32)
        x = x if x.is_contiguous() else x.contiguous()
        if x.shape[3] % 8 or x.shape[4] % 8:
            raise MvsHipError("CostRegNet3D needs H, W divisible by 8 (three stride-2 levels), got %s" % (tuple(x.shape[3:]),))
        c2 = self.conv2(self.conv1(x))
        c4 = self.conv4(self.conv3(c2))
        y = self.conv6(self.conv5(c4))
        y = self._up("conv7", y, c4)
        return self._up("conv9", y, c2), x

    def features(self, x: torch.Tensor) -> torch.Tensor:
        if not isinstance(self.inner, nn.Identity):
            raise MvsHipError("CostRegNet3D: in_channels != base_channel (1x1x1 'inner' conv) is not built")
        pre16 = self.training and x.dtype == torch.bfloat16       # already bf16 channel-last [B,D,H,W,C] (autograd.AggregateFn as_bf16)
        if not pre16:
            x = x.to(torch.float32)
        x = x if x.is_contiguous() else x.contiguous()
        hw = x.shape[2:4] if pre16 else x.shape[3:]
        if hw[0] % 8 or hw[1] % 8:
            raise MvsHipError("CostRegNet3D needs H, W divisible by 8 (three stride-2 levels), got %s" % (tuple(hw),))
        if self.training and autocast_bf16() and not pre16:
            from . import autograd as ag
            x = ag.ToBf16Fn.apply(x)                     # fp32 cost volume -> bf16 channel-last; every layer below follows the dtype
        l0, l2, l4 = _skip_links(x) if self.training else (None,) * 3
        c2 = _multi_use(self.conv2(self.conv1(x, take=l0)), l2)
        c4 = _multi_use(self.conv4(self.conv3(c2, take=l2)), l4)
        y = self.conv6(self.conv5(c4, take=l4))
        y = self._up("conv7", y, c4, l4)
        y = self._up("conv9", y, c2, l2)
        return _multi_use(self._up("conv11", y, x, l0))

    def prob_params(self):
        return _f32c(self.prob.weight).reshape(-1), _f32c(self.prob.bias).reshape(-1)

    def forward(self, x):
        y = self.features(x)
        if self.training:
            from . import autograd as ag
            if y.dtype == torch.bfloat16:                # the 1x1x1 head and everything after it are fp32 again
                if ag._fused_layers():
                    return ag.HeadBf16Fn.apply(y, self.prob.weight, self.prob.bias, False).unsqueeze(1)
                y = ag.FromBf16Fn.apply(y)
            return ag.Prob1Fn.apply(y, self.prob.weight, self.prob.bias)
        w, b = self.prob_params()
        return ops.prob1(y, w, b)


# ---------------------------------------------------------------------------------------------------------
# heads and schedulers (reference models/module.py:597-619, 633-653)
# ---------------------------------------------------------------------------------------------------------
def depth_regression(p, depth_values):
    """``sum_d p * depth_values``; ``depth_values`` is ``[B,D,H,W]`` or ``[B,D]`` (reference module.py:597-603)."""
    if depth_values.dim() > 2 and depth_values.shape != p.shape:
        depth_values = depth_values.expand_as(p)
    return ops.depth_regression(p.to(torch.float32).contiguous(), depth_values.to(torch.float32).contiguous())


def conf_regression(p, n=4):
    """Windowed probability mass around the expected index (reference module.py:606-619)."""
    return ops.conf_regression(p.detach().to(torch.float32).contiguous(), n)


def init_inverse_range(cur_depth, ndepths, device, dtype, H, W):
    """Reference signature (module.py:633); ``device``/``dtype`` are accepted for compatibility, output is fp32 on
    ``cur_depth``'s device."""
    return ops.init_inverse_range(cur_depth.to(torch.float32).contiguous(), ndepths, H, W)


def schedule_inverse_range(depth, depth_hypo, ndepths, split_itv, H, W):
    """Reference signature (module.py:642)."""
    return ops.schedule_inverse_range(depth.to(torch.float32).contiguous(), depth_hypo.to(torch.float32).contiguous(),
                                      ndepths, float(split_itv), H, W)
