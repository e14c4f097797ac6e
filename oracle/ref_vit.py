"""ORACLE — test infrastructure, not product code.

CPU restatement (plain torch functional ops, fp32) of the DINO ViT-small feature branch of the reference's shipped
``configs/config_mvsformer-p.json`` (``DINOMVSNet``): what runs per view between the image and ``conv31 + vit_out``:

  * the bicubic half-size resize of the image                     models/mvsformer_model.py:246-247
  * ``VisionTransformer.forward_with_last_att``                   models/vision_transformer.py:442-451 (prepare_tokens :418-431,
    interpolate_pos_encoding :394-416, Block :194-214, Attention :122-150, Mlp :104-119, PatchEmbed :324-338)
  * the token / attention reshapes                                models/mvsformer_model.py:254-257
  * ``VITDecoderStage4Single`` + ``AttentionFusionSimple``        models/module.py:353-368, 450-466

Pinned against outputs of the real reference modules (``tests/golden/vit_small.npz``, made by ``oracle/gen_golden.py`` with the
weights of ``oracle/weights.make_vit_state_dict``) in ``tests/test_oracle_vs_golden.py``.  Twins (``models/gvt.py``) needs ``timm`` and
stays unpinned; this file needs nothing but torch.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

EMBED, HEADS, DEPTH, PATCH = 384, 6, 12, 16          # vits.vit_small(patch_size=16), vision_transformer.py:610-614
LN_EPS = 1e-6


def resize_for_vit(img: torch.Tensor, rescale: float = 0.5) -> torch.Tensor:
    """mvsformer_model.py:246-247: bicubic to (int(H*rescale), int(W*rescale)), align_corners=False."""
    H, W = img.shape[-2:]
    return F.interpolate(img, (int(H * rescale), int(W * rescale)), mode="bicubic", align_corners=False)


def pos_encoding(sd: Dict[str, torch.Tensor], h: int, w: int) -> torch.Tensor:
    """interpolate_pos_encoding (vision_transformer.py:394-416) as prepare_tokens calls it - with (h, w) bound to the parameters named
    (w, h): the 14x14 table of a 224 checkpoint is resized bicubically by the scale factors ((h/16 + 0.1)/14, (w/16 + 0.1)/14)."""
    pe = sd["pos_embed"]
    N = pe.shape[1] - 1
    n = int(math.sqrt(N))
    hp, wp = h // PATCH, w // PATCH
    if hp * wp == N and h == w:
        return pe
    grid = pe[:, 1:].reshape(1, n, n, EMBED).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((hp + 0.1) / n, (wp + 0.1) / n), mode="bicubic", align_corners=False)
    assert grid.shape[-2:] == (hp, wp), (tuple(grid.shape), hp, wp)
    return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, hp * wp, EMBED)], dim=1)


def vit_forward_with_last_att(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (tokens after the final LayerNorm ``[B, 1+hw, 384]``, softmax attention of the LAST block ``[B, 6, 1+hw, 1+hw]``)."""
    B, _, h, w = x.shape
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=PATCH).flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1) + pos_encoding(sd, h, w)
    N, hd = t.shape[1], EMBED // HEADS
    att = None
    for i in range(DEPTH):
        p = "blocks.%d." % i
        y = F.layer_norm(t, (EMBED,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
        att = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)            # qk_scale='default'
        y = (att @ qkv[2]).transpose(1, 2).reshape(B, N, EMBED)
        t = t + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        y = F.layer_norm(t, (EMBED,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
        y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        t = t + y
    return F.layer_norm(t, (EMBED,), sd["norm.weight"], sd["norm.bias"], LN_EPS), att


def _bn(x, sd, p, training=False):
    """eval: running statistics; training: batch statistics, the running statistics in ``sd`` updated in place (momentum 0.1), as nn.BatchNorm2d."""
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], training, 0.1 if training else 0.0, 1e-5)


def vit_decoder(sd: Dict[str, torch.Tensor], feat: torch.Tensor, att: torch.Tensor, training: bool = False) -> torch.Tensor:
    """VITDecoderStage4Single (models/module.py:353-368,450-466): ``feat [B,384,h,w]``, ``att [B,6,h,w]`` -> ``[B,64,4h,4w]`` (added to conv31)."""
    swish = lambda v: v * torch.sigmoid(v)                                                   # models/module.py Swish
    x1 = swish(_bn(F.conv2d(torch.cat([feat, att], 1), sd["attn.conv_l.0.weight"], sd["attn.conv_l.0.bias"], padding=1), sd, "attn.conv_l.1.", training))
    x2 = swish(_bn(F.conv2d(feat * att.mean(1, keepdim=True), sd["attn.conv_r.0.weight"], sd["attn.conv_r.0.bias"], padding=1), sd, "attn.conv_r.1.", training))
    x = F.conv2d(x1 * x2, sd["attn.proj.weight"], sd["attn.proj.bias"])
    x = F.gelu(_bn(F.conv_transpose2d(x, sd["decoder.0.weight"], sd["decoder.0.bias"], stride=2, padding=1), sd, "decoder.1.", training))
    return F.gelu(_bn(F.conv_transpose2d(x, sd["decoder.3.weight"], sd["decoder.3.bias"], stride=2, padding=1), sd, "decoder.4.", training))


def vit_branch(sd_vit, sd_dec, img: torch.Tensor, rescale: float = 0.5):
    """One view of mvsformer_model.py:243-262 up to ``vit_out``: -> dict(vit_imgs, vit_feat [B,1+hw,384], att_cls [B,6,hw], vit_out)."""
    vit_imgs = resize_for_vit(img, rescale)
    tok, att = vit_forward_with_last_att(sd_vit, vit_imgs)
    B = img.shape[0]
    hp, wp = vit_imgs.shape[-2] // PATCH, vit_imgs.shape[-1] // PATCH
    feat = tok[:, 1:].reshape(B, hp, wp, EMBED).permute(0, 3, 1, 2).contiguous()
    att_cls = att[:, :, 0, 1:]
    return {"vit_imgs": vit_imgs, "vit_feat": tok, "att_cls": att_cls,
            "vit_out": vit_decoder(sd_dec, feat, att_cls.reshape(B, HEADS, hp, wp))}
