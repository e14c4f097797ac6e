"""ORACLE - test infrastructure, not product code.

The whole MVSFormer-P model in eval mode, images -> depth map: a restatement of ``DINOMVSNet.forward`` (models/mvsformer_model.py:205-308, the
``else`` branch :236-271 + the cascade :273-306) as a composition of the per-module restatements (``ref_fpn``, ``ref_vit``, ``ref_torch``),
over a flat ``state_dict`` with the reference's key prefixes (``encoder.``, ``decoder.``, ``vit.``, ``decoder_vit.``, ``fusions.<i>.``).
Pinned to the REAL class's output by tests/test_oracle_vs_golden.py (tests/golden/dinomvsnet_e2e.npz, oracle/gen_golden.py::gen_end_to_end).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from . import ref_fpn, ref_torch, ref_vit


def _sub(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    n = len(prefix) + 1
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def extract_features(sd: Dict[str, torch.Tensor], imgs: torch.Tensor, rescale: float = 0.5) -> Dict[str, torch.Tensor]:
    """mvsformer_model.py:236-271: per view FPN encoder, ViT branch added to ``conv31``, FPN decoder -> ``{stageK: [B,V,C,H/s,W/s]}``."""
    enc, dec, vit, dvit = (_sub(sd, p) for p in ("encoder", "decoder", "vit", "decoder_vit"))
    per_stage = [[], [], [], []]
    for v in range(imgs.shape[1]):
        img = imgs[:, v]
        conv01, conv11, conv21, conv31 = ref_fpn.fpn_encoder_forward(enc, img)
        conv31 = conv31 + ref_vit.vit_branch(vit, dvit, img, rescale)["vit_out"]
        for lst, f in zip(per_stage, ref_fpn.fpn_decoder_forward(dec, conv01, conv11, conv21, conv31)):
            lst.append(f)
    return {"stage%d" % (i + 1): torch.stack(lst, dim=1) for i, lst in enumerate(per_stage)}


def dinomvsnet_forward(sd: Dict[str, torch.Tensor], imgs: torch.Tensor, proj: Dict[str, torch.Tensor], depth_values: torch.Tensor, *,
                       ndepths: Sequence[int] = (32, 16, 8, 4), depth_interals_ratio: Sequence[float] = (4.0, 2.67, 1.5, 1.0), tmp=2.0,
                       rescale: float = 0.5) -> Dict[str, object]:
    with torch.no_grad():
        feats = extract_features(sd, imgs, rescale)
        out = ref_torch.cascade_forward(feats, proj, depth_values, [_sub(sd, "fusions.%d" % i) for i in range(len(ndepths))], ndepths=list(ndepths),
                                        depth_interals_ratio=list(depth_interals_ratio), tmp=tmp)
    out["features"] = feats
    return out
