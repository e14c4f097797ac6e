"""``DINOMVSNet`` - the whole MVSFormer-P model (models/mvsformer_model.py:163-308) on the MI355X path, images -> depth map, with the reference's
constructor arguments (``configs/config_mvsformer-p.json`` ``arch.args``), sub-module names and therefore ``state_dict`` keys
(``encoder.`` / ``decoder.`` / ``vit.`` / ``decoder_vit.`` / ``fusions.<i>.``: a reference checkpoint loads with ``strict=True``) and
``forward(imgs [B,V,3,H,W], proj_matrices, depth_values, tmp)`` -> the reference's output dict.

Composition only: every piece is a HIP-backed module of this package (``FPNEncoder`` / ``FPNDecoder`` csrc/conv2d.hip + fpn.hip, the DINO
ViT-small branch csrc/vit_packed.hip, the four ``StageNet``s).  Eval mode runs the V views of all B samples as ONE batch through the 2-D
networks (the reference loops over views, mvsformer_model.py:238-271; with eval BatchNorm the results per image are the same) and hands the
feature maps to the cascade channel-last.  Only what the shipped MVSFormer-P config builds is built: ``multi_scale=False``, ``att_fusion=True``,
``vit_arch='vit_small'``; Twins (``TwinMVSNet``) needs ``timm``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .cascade import CascadeMVS
from .fpn import FPNDecoder, FPNEncoder
from .vit import VITDecoderStage4Single, vit_branch, vit_small


class DINOMVSNet(CascadeMVS):
    def __init__(self, args: dict):
        super().__init__(args)                               # ndepths, depth_interals_ratio, fusions (mvsformer_model.py:166-170,203)
        a = self.args
        if a.get("multi_scale", False):
            raise _lib.MvsHipError("DINOMVSNet: multi_scale=True (VITDecoderStage4 + FPNDecoderV2 wiring) is not built; no shipped DINO config uses it")
        va = dict(a["vit_args"])
        if va.get("vit_arch", "vit_small") != "vit_small" or not va.get("att_fusion", True) or va.get("twin", False):
            raise _lib.MvsHipError("DINOMVSNet: only vit_arch='vit_small' with att_fusion=True is built (configs/config_mvsformer-p.json)")
        self.vit_args = va
        self.encoder = FPNEncoder(feat_chs=a["feat_chs"])
        self.decoder = FPNDecoder(feat_chs=a["feat_chs"])
        self.vit = vit_small(patch_size=va["patch_size"], qk_scale=va["qk_scale"])
        self.decoder_vit = VITDecoderStage4Single(va)
        fusions = self.fusions                               # registered last, as the reference does (state_dict key ORDER too)
        del self.fusions
        self.fusions = fusions

    def train(self, mode: bool = True):
        """``"fix": true`` (MVSFormer-P): the ViT runs under ``no_grad`` in the reference (mvsformer_model.py:216-218); it has no dropout and no
        BatchNorm, so keeping it in eval mode changes nothing and lets it stay on the eval-only HIP path."""
        super().train(mode)
        if self.args.get("fix", False):
            self.vit.eval()
        return self

    def extract_features(self, imgs: torch.Tensor):
        """mvsformer_model.py:209-271 -> ``{stageK: [B,V,C,H/s,W/s]}`` (logical NCHW, channel-last memory in eval)."""
        B, V, _, H, W = imgs.shape
        x = imgs.reshape(B * V, 3, H, W)
        conv01, conv11, conv21, conv31 = self.encoder(x) if self.training else self.encoder(x, conv01_channels_last=True)
        if self.training and not self.args.get("fix", False):
            raise _lib.MvsHipError("DINOMVSNet: training the ViT itself (fix=False) is not built; MVSFormer-P freezes it (\"fix\": true)")
        with torch.no_grad():                                # the frozen ViT (mvsformer_model.py:216-218,248-250)
            vb = vit_branch(self.vit, self.decoder_vit if not self.training else None, x, self.vit_args["rescale"])
        if self.training:
            P = self.vit.patch_size
            hp, wp = int(H * self.vit_args["rescale"]) // P, int(W * self.vit_args["rescale"]) // P
            feat = vb["vit_feat"][:, 1:].reshape(B * V, hp, wp, self.vit.embed_dim).permute(0, 3, 1, 2)
            vit_out = self.decoder_vit(feat, vb["att_cls"].reshape(B * V, -1, hp, wp))
        else:
            vit_out = vb["vit_out"]
        conv31 = conv31 + vit_out                            # mvsformer_model.py:229,263
        feats = self.decoder(conv01, conv11, conv21, conv31)
        return {"stage%d" % (i + 1): f.reshape(B, V, *f.shape[1:]) for i, f in enumerate(feats)}

    def forward(self, imgs, proj_matrices, depth_values, tmp=2.0):
        if not imgs.is_cuda:
            raise _lib.MvsHipError("DINOMVSNet: the MI355X HIP path is the only implementation (no CPU fallback)")
        if self.training:
            features = self.extract_features(imgs.to(torch.float32))
        else:
            with torch.no_grad():
                features = self.extract_features(imgs.to(torch.float32))
        return CascadeMVS.forward(self, features, proj_matrices, depth_values, tmp=tmp)
