"""The DINO ViT-small feature branch of MVSFormer-P on the MI355X path: ``vit_small`` / ``VisionTransformer`` with the
interface and the ``state_dict`` keys of the reference's ``models/vision_transformer.py`` (:340-451, ``vit_small`` :610-614) and
``VITDecoderStage4Single`` / ``AttentionFusionSimple`` of ``models/module.py`` (:353-368, :450-466), so the checkpoint of the shipped
``configs/config_mvsformer-p.json`` loads with ``strict=True``.  The ViT itself is eval-only (the reference freezes it: ``"fix": true``); the
decoder also runs in training mode (batch-statistics BatchNorm, every gradient; ``_forward_train``).  In eval every
matrix product - patch embedding, QKV, attention scores, attention x V, projections, MLP, the decoder's 3x3 convolutions and transposed
convolutions as implicit GEMMs - runs in ``csrc/vit.hip`` on the bf16 matrix cores in three-term split form (fp32-equivalent), LayerNorm,
softmax and the bicubic resizes are HIP kernels too; torch only reshapes / concatenates / slices (no arithmetic beyond one broadcast
multiply and one mean over 6 heads).  ``mvsformer_amd.install(features=True)`` rebinds ``models.vision_transformer.vit_small`` and
``VITDecoderStage4Single``.  Twins (``models/gvt.py``) needs ``timm`` and is not built.
"""
from __future__ import annotations

import math
import os
from functools import partial

import torch
import torch.nn as nn

from . import _lib, ops
from .module import _publish_cache, _versions


def _f(t):
    return t.detach().to(torch.float32).contiguous()


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    """models/vision_transformer.py:340-451 (``cross_att=False``, ``qk_scale='default'``: what the shipped configs build)."""

    def __init__(self, img_size=(224,), patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=False, qk_scale="default", norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        if qk_scale != "default" or kwargs.get("cross_att", False) or num_classes:
            raise _lib.MvsHipError("VisionTransformer: only qk_scale='default', no cross attention, no classifier head is built (the shipped configs)")
        self.embed_dim, self.num_heads, self.patch_size = embed_dim, num_heads, patch_size
        self.patch_embed = PatchEmbed(img_size[0], patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self._cache = None
        self._pos_cache = {}

    # ---- position table resized to the token grid (vision_transformer.py:394-416; prepare_tokens passes (h, w) as (w, h)) ----
    def _pos(self, hp: int, wp: int) -> torch.Tensor:
        key = (self.pos_embed.data_ptr(), self.pos_embed._version, hp, wp)
        if self._pos_cache.get("key") != key:
            pe = _f(self.pos_embed)
            N = pe.shape[1] - 1
            n = int(math.sqrt(N))
            if hp * wp == N and hp == wp:
                pos = pe
            else:
                grid = pe[0, 1:].reshape(n, n, self.embed_dim).permute(2, 0, 1).contiguous()              # [C, n, n]
                sh, sw = (hp + 0.1) / n, (wp + 0.1) / n                                                  # the reference's scale factors
                if int(n * sh) != hp or int(n * sw) != wp:
                    raise _lib.MvsHipError("position table resize: %dx%d does not come out of scale factors %.4f, %.4f" % (hp, wp, sh, sw))
                grid = ops.bicubic_resize(grid, hp, wp, 1.0 / sh, 1.0 / sw)
                pos = torch.cat([pe[:, :1], grid.permute(1, 2, 0).reshape(1, hp * wp, self.embed_dim)], dim=1).contiguous()
            _publish_cache()
            self._pos_cache = {"key": key, "pos": pos}
        return self._pos_cache["pos"]

    def _prepared(self):
        key = _versions(self)
        if self._cache is None or self._cache[0] != key:
            pw = _f(self.patch_embed.proj.weight).reshape(self.embed_dim, -1)
            blocks = []
            for b in self.blocks:
                blocks.append(tuple(_f(t) for t in (b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight,
                                                    b.attn.proj.bias, b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias,
                                                    b.mlp.fc2.weight, b.mlp.fc2.bias)))
            # the linear layers' weights split once into (h, m, l) bf16 MFMA fragments (csrc/vit_packed.hip): no block of any GEMM splits them again
            packed = None
            if self._packed_ok():
                packed = [tuple(ops.x3p_pack(w) for w in (b[2], b[4], b[8], b[10])) for b in blocks]
            _publish_cache()
            self._cache = (key, pw, _f(self.patch_embed.proj.bias), blocks, _f(self.norm.weight), _f(self.norm.bias), _f(self.cls_token), packed)
        return self._cache[1:]

    def _packed_ok(self) -> bool:
        """The pre-split path (csrc/vit_packed.hip) covers heads of 64 with C a multiple of 128 up to 512 (ViT-small)."""
        C, NH = self.embed_dim, self.num_heads
        return C == NH * 64 and C % 128 == 0 and C <= 512 and os.environ.get("MVS_VIT_PACKED", "1") != "0"

    def _run_packed(self, x: torch.Tensor, want_cls: bool):
        """The 12 blocks on pre-split operands: LayerNorm -> packed, qkv GEMM -> packed Q / K / V^T per head, flash attention -> packed,
        projection (+ residual) -> fp32 tokens, LayerNorm -> packed, fc1 (+ GELU) -> packed, fc2 (+ residual) -> fp32 tokens.  Token rows are
        laid out [B][Np] with Np = N rounded up to 32 (padding rows stay finite and are masked as keys).  ``want_cls``: also the CLS query's
        attention row of the last block, ``[B, heads, N]`` - the only part of the attention matrix the model reads (mvsformer_model.py:257)."""
        pw, pb, blocks, nw, nb, cls, packed = self._prepared()
        x = x.detach().to(torch.float32)
        B, nc, h, w = x.shape
        P, C, NH = self.patch_size, self.embed_dim, self.num_heads
        hp, wp = h // P, w // P
        n = hp * wp
        N = n + 1
        Np = (N + 31) // 32 * 32
        M = B * Np
        dev = x.device
        patches = x[:, :, :hp * P, :wp * P].reshape(B, nc, hp, P, wp, P).permute(0, 2, 4, 1, 3, 5).reshape(B * n, nc * P * P).contiguous()
        tok = torch.zeros(B, Np, C, device=dev, dtype=torch.float32)
        tok[:, 0] = cls[0, 0]
        ops.gemm_x3(patches, pw, tok, n, C, nc * P * P, nc * P * P, nc * P * P, C, nb1=B, sA=(n * nc * P * P, 0), sC=(Np * C, 0), shift=pb, c_off=C)
        tok[:, :N] += self._pos(hp, wp)
        t, t2 = tok.view(M, C), torch.empty(M, C, device=dev, dtype=torch.float32)
        hidden = blocks[0][8].shape[0]
        y_p, a_p, h_p = ops.Packed(M, C, dev), ops.Packed(M, C, dev), ops.Packed(M, hidden, dev)
        qkv_p = ops.QkvPacked(B, NH, Np, dev)
        eps = self.norm.eps
        att_cls = None
        for i, ((n1w, n1b, _, qb, _, prb, n2w, n2b, _, f1b, _, f2b), (wq, wpr, w1, w2)) in enumerate(zip(blocks, packed)):
            ops.layernorm_x3p(t, n1w, n1b, eps, Np, N, out=y_p)
            ops.gemm_x3p_qkv(y_p, wq, qb, B, Np, NH, (C // NH) ** -0.5, out=qkv_p)
            if want_cls and i == len(blocks) - 1:
                att_cls = ops.cls_attention_x3p(qkv_p, N)
            ops.attention_x3p(qkv_p, N, out=a_p)
            ops.gemm_x3p(a_p, wpr, C, C=t2, shift=prb, res=t)
            ops.layernorm_x3p(t2, n2w, n2b, eps, Np, N, out=y_p)
            ops.gemm_x3p(y_p, w1, hidden, shift=f1b, act=1, out=h_p)
            ops.gemm_x3p(h_p, w2, C, C=t, shift=f2b, res=t2)
        out = ops.layernorm(tok, nw, nb, eps)[:, :N].contiguous()
        return (out, att_cls) if want_cls else out

    def _run(self, x: torch.Tensor, want_att):
        """``want_att``: False, True (the last block's whole attention matrix, ``forward_with_last_att``'s contract) or ``"cls"`` (only the CLS
        query's row ``[B, heads, N]``: what ``vit_branch`` needs)."""
        if self.training:
            raise _lib.MvsHipError("VisionTransformer: only eval mode is built on the HIP path (the reference freezes the ViT: \"fix\": true)")
        if want_att is not True and self._packed_ok():
            return self._run_packed(x, want_att == "cls")
        pw, pb, blocks, nw, nb, cls = self._prepared()[:6]
        x = x.detach().to(torch.float32)
        B, nc, h, w = x.shape
        P, C, NH = self.patch_size, self.embed_dim, self.num_heads
        hp, wp = h // P, w // P
        n = hp * wp
        N = n + 1
        hd = C // NH
        # patch embedding = GEMM over non-overlapping patches: [B*n, nc*P*P] x [C, nc*P*P]^T
        patches = x[:, :, :hp * P, :wp * P].reshape(B, nc, hp, P, wp, P).permute(0, 2, 4, 1, 3, 5).reshape(B * n, nc * P * P).contiguous()
        tok = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        tok[:, 0] = cls[0, 0]
        ops.gemm_x3(patches, pw, tok, n, C, nc * P * P, nc * P * P, nc * P * P, C, nb1=B, sA=(n * nc * P * P, 0), sC=(N * C, 0), shift=pb, c_off=C)
        t = (tok + self._pos(hp, wp)).contiguous()
        eps = self.norm.eps
        scores, vt_pad = None, None
        for i, (n1w, n1b, qw, qb, prw, prb, n2w, n2b, f1w, f1b, f2w, f2b) in enumerate(blocks):
            y = ops.layernorm(t, n1w, n1b, eps)
            qkv = torch.empty(B, N, 3 * C, device=x.device, dtype=torch.float32)
            ops.gemm_x3(y, qw, qkv, B * N, 3 * C, C, C, C, 3 * C, shift=qb)
            last = want_att and i == len(blocks) - 1
            if hd == 64 and not last and os.environ.get("MVS_VIT_FLASH", "1") != "0":
                # flash form: softmax(Q K^T / sqrt(hd)) V without the N x N matrix (csrc/vit.hip attention_x3_kernel); V handed over
                # TRANSPOSED with 16-byte aligned rows ([B, heads, hd, N rounded up to 4], one strided copy into a buffer reused by all blocks)
                vt_pad = ops.attention_vt(qkv, NH, vt_pad)
                att_out = ops.attention_x3(qkv, vt_pad, NH, hd ** -0.5)
            else:
                # materialized form (the LAST block's attention matrix is an output: mvsformer_model.py:257 reads its CLS row): scores[b, h] =
                # Q . K^T (head slices of the packed qkv rows), softmax(scale * .), out[b, :, h] = P . V
                vt = qkv[:, :, 2 * C:].reshape(B, N, NH, hd).permute(0, 2, 3, 1).contiguous()     # the GEMM's B operand, read along K
                if scores is None:
                    scores = torch.empty(B, NH, N, N, device=x.device, dtype=torch.float32)
                ops.gemm_x3(qkv, qkv, scores, N, N, hd, 3 * C, 3 * C, N, nb1=B, nb2=NH, sA=(N * 3 * C, hd), sB=(N * 3 * C, hd), sC=(NH * N * N, N * N),
                            b_off=C)
                ops.softmax_rows_(scores, hd ** -0.5)
                att_out = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
                ops.gemm_x3(scores, vt, att_out, N, hd, N, N, N, C, nb1=B, nb2=NH, sA=(NH * N * N, N * N), sB=(NH * hd * N, hd * N), sC=(N * C, hd))
            t2 = torch.empty_like(t)
            ops.gemm_x3(att_out, prw, t2, B * N, C, C, C, C, C, shift=prb, res=t)
            y = ops.layernorm(t2, n2w, n2b, eps)
            hid = torch.empty(B, N, f1w.shape[0], device=x.device, dtype=torch.float32)
            ops.gemm_x3(y, f1w, hid, B * N, f1w.shape[0], C, C, C, f1w.shape[0], shift=f1b, act=1)
            ops.gemm_x3(hid, f2w, t, B * N, C, f1w.shape[0], f1w.shape[0], f1w.shape[0], C, shift=f2b, res=t2)
        out = ops.layernorm(t, nw, nb, eps)
        if want_att == "cls":
            return out, scores[:, :, 0].contiguous()
        return (out, scores) if want_att else out

    def forward(self, x, src_epipoles=None):
        return self._run(x, False)

    def forward_with_last_att(self, x):
        """-> (tokens after the final LayerNorm ``[B, 1+hw, C]``, attention of the last block ``[B, heads, 1+hw, 1+hw]``)."""
        return self._run(x, True)

    def forward_with_cls_att(self, x):
        """-> (tokens ``[B, 1+hw, C]``, the CLS query's attention row of the last block ``[B, heads, 1+hw]``) = what mvsformer_model.py:246-257
        uses of ``forward_with_last_att`` (``vit_att[:, :, 0, 1:]``) without writing the other 1+hw rows."""
        return self._run(x, "cls")


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


class _Act(nn.Module):
    """Parameterless placeholder keeping ``nn.Sequential`` indices (and therefore checkpoint keys) as in the reference."""

    def forward(self, x):
        raise _lib.MvsHipError("structural placeholder: the activation runs in the GEMM epilogue")


class AttentionFusionSimple(nn.Module):
    def __init__(self, vit_ch, out_ch, nhead):
        super().__init__()
        self.conv_l = nn.Sequential(nn.Conv2d(vit_ch + nhead, vit_ch, kernel_size=3, padding=1), nn.BatchNorm2d(vit_ch))
        self.conv_r = nn.Sequential(nn.Conv2d(vit_ch, vit_ch, kernel_size=3, padding=1), nn.BatchNorm2d(vit_ch))
        self.act = _Act()
        self.proj = nn.Conv2d(vit_ch, out_ch, kernel_size=1)


def _fold(conv, bn):
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
    shift = bn.bias.detach().double() + (conv.bias.detach().double() - bn.running_mean.double()) * scale
    return scale.float().contiguous(), shift.float().contiguous()


def _conv3_matrix(w: torch.Tensor, cp: int) -> torch.Tensor:
    """``[Cout,Cin,3,3]`` -> ``[Cout, 9*cp]`` with k = tap*cp + c (channels zero-padded to ``cp``)."""
    cout, cin = w.shape[:2]
    m = torch.zeros(cout, 9, cp, device=w.device, dtype=torch.float32)
    m[:, :, :cin] = _f(w).permute(0, 2, 3, 1).reshape(cout, 9, cin)
    return m.reshape(cout, 9 * cp).contiguous()


def _convT_matrices(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(k 4, s 2, p 1) weight ``[Cin,Cout,4,4]`` -> ``[4 classes, Cout, 4*Cin]``: class (ph, pw) = output parity, its 2x2 taps
    (th, tw) use ky = (1, 3) for ph = 0 and (0, 2) for ph = 1 (input rows y, y-1 and y+1, y), kx likewise; k = (th*2 + tw)*Cin + c."""
    cin, cout = w.shape[:2]
    wf = _f(w)
    ks = ((1, 3), (0, 2))
    out = torch.empty(4, cout, 4, cin, device=w.device, dtype=torch.float32)
    for ph in range(2):
        for pw in range(2):
            for th in range(2):
                for tw in range(2):
                    out[ph * 2 + pw, :, th * 2 + tw] = wf[:, :, ks[ph][th], ks[pw][tw]].t()
    return out.reshape(4, cout, 4 * cin).contiguous()


class ConvT2dFn(torch.autograd.Function):
    """Raw ``ConvTranspose2d`` (weight ``[Cin,Cout,KS,KS]``, no bias) in training, fp32 NCHW: the transposed convolution IS the data gradient of
    the convolution with the same weight read as ``[Cout_conv = Cin][Cin_conv = Cout]`` - forward = ``mvs_conv2d_gemm_x3`` mode 2, data
    gradient = mode 1, weight gradient = mode 3 with ``x`` in the role of the convolution's output gradient."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        x = x.to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        ctx.save_for_backward(x, w)
        ctx.cfg = (int(stride), int(pad))
        KS = w.shape[2]
        Ho, Wo = (x.shape[2] - 1) * stride - 2 * pad + KS, (x.shape[3] - 1) * stride - 2 * pad + KS
        return ops.conv2d_dgrad_x3(x, w, int(stride), int(pad), Ho, Wo)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.cfg
        dy = dy.contiguous()
        dx = ops.conv2d_fwd_x3(dy, w, stride, pad) if ctx.needs_input_grad[0] else None
        dw = ops.conv2d_wgrad_x3(x, dy, w.shape[2], stride, pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


class MulFn(torch.autograd.Function):
    """``a * b`` (models/module.py:464) with both gradients through ``mvs_ewise_mul``."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.ewise_mul(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        return (ops.ewise_mul(dy, b) if ctx.needs_input_grad[0] else None), (ops.ewise_mul(dy, a) if ctx.needs_input_grad[1] else None)


ACT_GELU_BN = 4                          # GELU(erf) in the fp32 BatchNorm kernels (csrc/train.hip); 3 = Swish


class VITDecoderStage4Single(nn.Module):
    """models/module.py:353-368: ``forward(x [B,vit_ch,h,w], att [B,nhead,h,w]) -> [B,out_ch,4h,4w]`` (added to ``conv31``)."""

    def __init__(self, args):
        super().__init__()
        ch, vit_ch = args["out_ch"], args["vit_ch"]
        assert args["att_fusion"] is True
        self.attn = AttentionFusionSimple(vit_ch, ch * 4, args["nhead"])
        self.decoder = nn.Sequential(nn.ConvTranspose2d(ch * 4, ch * 2, 4, stride=2, padding=1), nn.BatchNorm2d(ch * 2), nn.GELU(),
                                     nn.ConvTranspose2d(ch * 2, ch, 4, stride=2, padding=1), nn.BatchNorm2d(ch), nn.GELU())
        self._cache = None

    def _prepared(self):
        key = _versions(self)
        if self._cache is None or self._cache[0] != key:
            a = self.attn
            cl_in = a.conv_l[0].in_channels
            cp = (cl_in + 7) // 8 * 8
            prep = dict(cp=cp, wl=_conv3_matrix(a.conv_l[0].weight, cp), fl=_fold(a.conv_l[0], a.conv_l[1]),
                        wr=_conv3_matrix(a.conv_r[0].weight, a.conv_r[0].in_channels), fr=_fold(a.conv_r[0], a.conv_r[1]),
                        wp=_f(a.proj.weight).reshape(a.proj.out_channels, -1), bp=_f(a.proj.bias),
                        w1=_convT_matrices(self.decoder[0].weight), f1=_fold(self.decoder[0], self.decoder[1]),
                        w2=_convT_matrices(self.decoder[3].weight), f2=_fold(self.decoder[3], self.decoder[4]))
            if self._packed_ok():
                # the same matrices split once into MFMA fragments (csrc/vit_packed.hip; channels of the [x | att] map padded to a multiple of 32)
                cpp = (cl_in + 31) // 32 * 32
                prep.update(cpp=cpp, wl_p=ops.x3p_pack(_conv3_matrix(a.conv_l[0].weight, cpp)), wr_p=ops.x3p_pack(prep["wr"]), wp_p=ops.x3p_pack(prep["wp"]),
                            w1_p=ops.x3p_pack_classes(prep["w1"], 128), w2_p=ops.x3p_pack_classes(prep["w2"], 128))
            _publish_cache()
            self._cache = (key, prep)
        return self._cache[1]

    def _packed_ok(self) -> bool:
        a, d = self.attn, self.decoder
        chans = (a.conv_r[0].in_channels, a.proj.out_channels, d[0].out_channels)
        return all(c % 32 == 0 for c in chans) and d[3].out_channels % 4 == 0 and os.environ.get("MVS_VIT_PACKED", "1") != "0"

    def _forward_packed(self, p, xc, ac):
        """The decoder on pre-split operands: every convolution is an implicit GEMM whose A operand is gathered from a packed channel-last map
        by LDS-DMA (``mvs_conv_x3p``); intermediate maps are written packed by the producing epilogue."""
        B, h, w, C = xc.shape
        M, nh, dev = B * h * w, ac.shape[-1], xc.device
        ra = (M + 128) // 128 * 128                           # rows allocated: the pixels + at least one zero row (taps outside the image)
        cat = torch.zeros(M, p["cpp"], device=dev, dtype=torch.float32)
        cat[:, :C] = xc.reshape(M, C)
        cat[:, C:C + nh] = ac.reshape(M, nh)
        x1 = torch.empty(M, C, device=dev, dtype=torch.float32)
        ops.conv_x3p(ops.x3p_pack(cat, ra), p["wl_p"], 1, B, h, w, C, C=x1, scale=p["fl"][0], shift=p["fl"][1], act=2)
        xr = (xc * ac.mean(dim=-1, keepdim=True)).reshape(M, C).contiguous()
        x12 = ops.Packed(M, C, dev)
        ops.conv_x3p(ops.x3p_pack(xr, ra), p["wr_p"], 1, B, h, w, C, scale=p["fr"][0], shift=p["fr"][1], act=2, mul=x1, out=x12)
        co = p["wp"].shape[0]
        y = ops.Packed(M, co, dev, rows_alloc=ra, zero=True)
        ops.gemm_x3p(x12, p["wp_p"], co, shift=p["bp"], out=y)
        c1, c2 = p["w1"].shape[1], p["w2"].shape[1]
        y1 = ops.Packed(4 * M, c1, dev, rows_alloc=(4 * M + 128) // 128 * 128, zero=True)
        ops.conv_x3p(y, p["w1_p"], 2, B, h, w, c1, scale=p["f1"][0], shift=p["f1"][1], act=1, out=y1)
        out = torch.empty(B, 4 * h, 4 * w, c2, device=dev, dtype=torch.float32)
        ops.conv_x3p(y1, p["w2_p"], 2, B, 2 * h, 2 * w, c2, C=out.view(16 * M, c2), scale=p["f2"][0], shift=p["f2"][1], act=1)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def _up(x_cl, wm, fold, act):
        """One ConvTranspose2d(4, 2, 1) + folded BatchNorm + activation on a channel-last map ``[B,h,w,C]`` -> ``[B,2h,2w,Cout]``."""
        B, h, w, C = x_cl.shape
        cout = wm.shape[1]
        tmp = torch.empty(B, 4, h * w, cout, device=x_cl.device, dtype=torch.float32)
        ops.gemm_x3(x_cl, wm, tmp, h * w, cout, 4 * C, 0, 4 * C, cout, nb1=B, nb2=4, sA=(h * w * C, 0), sB=(0, cout * 4 * C), sC=(4 * h * w * cout, h * w * cout),
                    a_mode=2, H=h, W=w, Cp=C, scale=fold[0], shift=fold[1], act=act)
        return tmp.view(B, 2, 2, h, w, cout).permute(0, 3, 1, 4, 2, 5).reshape(B, 2 * h, 2 * w, cout).contiguous()

    def _forward_train(self, x, att):
        """models/module.py:365-368 + :459-466 with batch statistics, every op an autograd-tracked HIP kernel (fp32 NCHW like the FPN's training
        path): conv_l / conv_r / proj through ``Conv2dFn`` + ``BiasFn``, BatchNorm + Swish / GELU through ``BnActFn`` (SyncBatchNorm-aware), the
        gated product through ``MulFn``, the two ``ConvTranspose2d`` through ``ConvT2dFn``."""
        from .autograd import BnActFn
        from .fpn import ACT_SWISH, BiasFn, Conv2dFn
        a, d = self.attn, self.decoder
        x = x.to(torch.float32).contiguous()
        att = att.to(torch.float32).contiguous()

        def conv_bn(t, seq):
            y = BiasFn.apply(Conv2dFn.apply(t, seq[0].weight, 1, 1), seq[0].bias)
            return BnActFn.apply(y, seq[1].weight, seq[1].bias, None, seq[1], ACT_SWISH)

        x1 = conv_bn(torch.cat([x, att], dim=1), a.conv_l)
        gate = att.mean(dim=1, keepdim=True)                  # inputs come from the frozen ViT: no gradient flows through these two torch ops
        x2 = conv_bn(x * gate if not (x.requires_grad or att.requires_grad) else MulFn.apply(x, gate.expand_as(x)), a.conv_r)
        y = BiasFn.apply(Conv2dFn.apply(MulFn.apply(x1, x2), a.proj.weight, 1, 0), a.proj.bias)
        for conv, bn in ((d[0], d[1]), (d[3], d[4])):
            y = BiasFn.apply(ConvT2dFn.apply(y, conv.weight, conv.stride[0], conv.padding[0]), conv.bias)
            y = BnActFn.apply(y, bn.weight, bn.bias, None, bn, ACT_GELU_BN)
        return y

    def forward(self, x, att):
        if self.training:
            return self._forward_train(x, att)
        p = self._prepared()
        with torch.no_grad():
            B, C, h, w = x.shape
            # channel-last views ([B, vit_ch, h, w] made from tokens is a permuted view of [B, h*w, vit_ch]: contiguous() is free then)
            xc = x.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
            ac = att.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
            nh = ac.shape[-1]
            if "wl_p" in p:
                return self._forward_packed(p, xc, ac)
            cat = torch.zeros(B, h, w, p["cp"], device=x.device, dtype=torch.float32)
            cat[..., :C] = xc
            cat[..., C:C + nh] = ac
            x1 = torch.empty(B, h * w, C, device=x.device, dtype=torch.float32)
            ops.gemm_x3(cat, p["wl"], x1, h * w, C, 9 * p["cp"], 0, 9 * p["cp"], C, nb1=B, sA=(h * w * p["cp"], 0), sC=(h * w * C, 0), a_mode=1, H=h, W=w,
                        Cp=p["cp"], scale=p["fl"][0], shift=p["fl"][1], act=2)
            xr = (xc * ac.mean(dim=-1, keepdim=True)).contiguous()
            x12 = torch.empty_like(x1)
            ops.gemm_x3(xr, p["wr"], x12, h * w, C, 9 * C, 0, 9 * C, C, nb1=B, sA=(h * w * C, 0), sC=(h * w * C, 0), a_mode=1, H=h, W=w, Cp=C,
                        scale=p["fr"][0], shift=p["fr"][1], act=2, mul=x1)
            co = p["wp"].shape[0]
            y = torch.empty(B, h, w, co, device=x.device, dtype=torch.float32)
            ops.gemm_x3(x12, p["wp"], y, B * h * w, co, C, C, C, co, shift=p["bp"])
            y = self._up(y, p["w1"], p["f1"], 1)
            y = self._up(y, p["w2"], p["f2"], 1)
            return y.permute(0, 3, 1, 2)                      # logical NCHW (channel-last memory); `conv31 + vit_out` broadcasts layouts


def vit_branch(vit: VisionTransformer, dec: VITDecoderStage4Single, img: torch.Tensor, rescale: float = 0.5):
    """One view of mvsformer_model.py:243-262 up to ``vit_out``: bicubic resize, ViT with the last block's attention, reshapes, decoder."""
    B, _, H, W = img.shape
    vh, vw = int(H * rescale), int(W * rescale)
    x = ops.bicubic_resize(img.detach().to(torch.float32).contiguous(), vh, vw, H / vh, W / vw)
    tok, att = vit.forward_with_cls_att(x)
    P = vit.patch_size
    hp, wp = vh // P, vw // P
    feat = tok[:, 1:].reshape(B, hp, wp, vit.embed_dim).permute(0, 3, 1, 2)
    att_cls = att[:, :, 1:].reshape(B, -1, hp, wp)
    return {"vit_imgs": x, "vit_feat": tok, "att_cls": att[:, :, 1:], "vit_out": dec(feat, att_cls) if dec is not None else None}
