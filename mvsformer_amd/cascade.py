"""The coarse-to-fine loop the reference's top models run around ``StageNet`` — restated from
``DINOMVSNet.forward`` / ``TwinMVSNet.forward`` (models/mvsformer_model.py:273-306 / 410-449) for callers that
start from precomputed per-stage feature maps (the bench, the tests, inference sharding).  Feature extraction
(FPN + ViT) is outside the path and not part of this package.
"""
from __future__ import annotations

import os
from typing import Dict, Sequence

import torch
import torch.nn as nn

from . import ops
from .module import init_inverse_range, schedule_inverse_range
from .stagenet import StageNet

DEFAULT_ARGS = dict(base_ch=8, fusion_type="cnn", depth_type="ce", model_th=8, inverse_depth=True,
                    ndepths=[32, 16, 8, 4], depth_interals_ratio=[4.0, 2.67, 1.5, 1.0], feat_chs=[8, 16, 32, 64])


class CascadeMVS(nn.Module):
    """``fusions`` holds the StageNets under the same attribute name (and therefore the same checkpoint key
    prefix ``fusions.{i}.``) as the reference models."""

    def __init__(self, args: dict = None):
        super().__init__()
        self.args = dict(DEFAULT_ARGS, **(args or {}))
        self.ndepths = list(self.args["ndepths"])
        self.depth_interals_ratio = list(self.args["depth_interals_ratio"])
        if not self.args.get("inverse_depth", False):
            raise NotImplementedError("only inverse_depth=True (every shipped reference config) is built")
        self.fusions = nn.ModuleList([StageNet(self.args, self.ndepths[i], i) for i in range(len(self.ndepths))])

    def forward(self, features: Dict[str, torch.Tensor], proj_matrices: Dict[str, torch.Tensor], depth_values: torch.Tensor,
                tmp=2.0) -> Dict[str, object]:
        """Eval: no autograd graph (as ``test.py`` runs the reference under ``torch.no_grad``).  Train: every stage keeps
        its graph; hypotheses come from the previous stage's detached depth (mvsformer_model.py:290,430)."""
        if not self.training:
            with torch.no_grad():
                return self._forward(features, proj_matrices, depth_values, tmp)
        # bf16 training: the weights of ALL stages routed through one flush node - the ~50 weight gradients of the step run together at the
        # end of the backward (autograd.route_weights; MVS_TRAIN_WGRAD_SCOPE=stage keeps one group per stage)
        from . import autograd as ag
        from .module import autocast_bf16
        stages = list(self.fusions)
        shared = autocast_bf16() and ag._fused_layers() and os.environ.get("MVS_TRAIN_WGRAD_SCOPE", "cascade") != "stage" \
            and ag.route_weights([st.train_pack() for st in stages])
        if not shared:
            return self._forward(features, proj_matrices, depth_values, tmp)
        for st in stages:
            st._routed_by_cascade = True
        try:
            return self._forward(features, proj_matrices, depth_values, tmp)
        finally:
            for st in stages:
                st._routed_by_cascade = False
                st.train_pack().unroute()                       # the routed weights belong to this forward's graph only

    def _forward(self, features, proj_matrices, depth_values, tmp):
        n = len(self.ndepths)
        last = features["stage%d" % n]
        B, Hf, Wf = last.shape[0], last.shape[-2], last.shape[-1]
        prob_maps = torch.zeros(B, Hf, Wf, dtype=torch.float32, device=last.device)
        outputs: Dict[str, object] = {}
        stage_out = None
        if not self.training and n <= 4 and last.is_cuda and os.environ.get("MVS_TRANSPOSE_MULTI", "1") != "0":
            # eval: the four stages' NCHW -> NHWC transposes in ONE launch up front (each stage then finds its map channel-last already)
            keys = ["stage%d" % (i + 1) for i in range(n)]
            cl = ops.to_channels_last_multi([features[k].detach().to(torch.float32) for k in keys])
            features = dict(features, **dict(zip(keys, cl)))
        for i in range(n):
            f = features["stage%d" % (i + 1)]
            H, W = f.shape[-2:]
            if i == 0:
                hyp = init_inverse_range(depth_values, self.ndepths[0], f.device, torch.float32, H, W)
            else:
                hyp = schedule_inverse_range(stage_out["depth"].detach(), stage_out["depth_values"].detach(), self.ndepths[i],
                                             self.depth_interals_ratio[i], H, W)
            stage_out = self.fusions[i](f, proj_matrices["stage%d" % (i + 1)], hyp, tmp=tmp)
            # nearest-upsampled confidences averaged over the stages (mvsformer_model.py:297-301,305)
            ops.conf_accumulate(stage_out["photometric_confidence"].detach().contiguous(), prob_maps, 1.0 / n)
            outputs["stage%d" % (i + 1)] = stage_out
            outputs.update(stage_out)
        outputs["refined_depth"] = stage_out["depth"]
        outputs["photometric_confidence"] = prob_maps
        return outputs


def randomize_bn_(module: nn.Module, seed: int = 1) -> None:
    """Give every BatchNorm non-trivial running statistics / affine parameters (default init makes eval-mode BN the
    identity, which would make the bench skip the epilogue arithmetic a trained checkpoint has)."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            with torch.no_grad():
                m.running_mean.copy_(0.2 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
