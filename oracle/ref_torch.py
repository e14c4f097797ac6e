"""ORACLE — test infrastructure, not product code.

CPU restatement (torch.float32 on the host, materializing formulation) of the
reference's plane-sweep cost-volume path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product
(``mvsformer_amd``) never imports it and has no CPU fallback.

The reference for this path is pure Python over ATen ops (no native code under
/root/reference), so the arithmetic lives in the third-party dependency
``torch`` (reference pins torch==1.9.0+cu111, requirements.txt:9; the oracle
runs on the image's torch 2.10 CPU kernels).  The reference ships no tests or
golden vectors for this path, so the oracle is pinned by
``tests/golden/*.npz``: outputs of the real reference functions imported from
/root/reference in the build container by ``oracle/gen_golden.py`` (committed
beside the vectors).  ``tests/test_oracle_vs_golden.py`` checks every function
here against those vectors.

Every function cites the reference lines it restates.  Weights come in as a
flat ``dict`` with the reference's ``state_dict`` key names, relative to one
``StageNet`` (e.g. ``cost_reg.conv1.conv.weight``, ``vis.0.bn.running_var``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

BN_EPS = 1e-5   # nn.BatchNorm2d/3d default, never overridden by the reference


# --------------------------------------------------------------------------------------
# a2: projection prep  (reference models/mvsformer_model.py:69-72)
# --------------------------------------------------------------------------------------
def compose_projection(pair: torch.Tensor) -> torch.Tensor:
    """``pair [B,2,4,4]`` (extrinsic, intrinsic) -> ``P [B,4,4]`` with ``P[:3,:4] = K[:3,:3] @ E[:3,:4]``
    and the extrinsic's last row kept."""
    E, K = pair[:, 0], pair[:, 1]
    P = E.clone()
    P[:, :3, :4] = torch.matmul(K[:, :3, :3], E[:, :3, :4])
    return P


# --------------------------------------------------------------------------------------
# a1: homography warp  (reference models/warping.py:69-109, 155-189)
# --------------------------------------------------------------------------------------
def sweep_coordinates(src_proj: torch.Tensor, ref_proj: torch.Tensor, depth: torch.Tensor, H: int, W: int):
    """Normalized sampling grid and camera-space z for every (depth, pixel).

    warping.py:80-96: ``M = src_proj @ inv(ref_proj)``; ``X = (M[:3,:3] @ (x,y,1)) * d + M[:3,3]``;
    ``u = X0/(X2+1e-6)``, ``v = X1/(X2+1e-6)``; ``u_n = u/((W-1)/2) - 1``.  Pixel grid is the integer
    lattice x in [0,W-1], y in [0,H-1].  Returns ``(u_n, v_n, z)`` each ``[B,D,H*W]``.
    """
    B, D = depth.shape[0], depth.shape[1]
    M = torch.matmul(src_proj, torch.inverse(ref_proj))
    R, t = M[:, :3, :3], M[:, :3, 3]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W)], dim=0)        # [3,HW]
    rays = torch.matmul(R, pix.unsqueeze(0).expand(B, -1, -1))                           # [B,3,HW]
    dep = depth.reshape(B, 1, D, -1)                                                     # [B,1,D,HW|1]
    X = rays.unsqueeze(2) * dep + t.reshape(B, 3, 1, 1)                                  # [B,3,D,HW]
    uv = X[:, :2] / (X[:, 2:3] + 1e-6)
    u_n = uv[:, 0] / ((W - 1) / 2) - 1
    v_n = uv[:, 1] / ((H - 1) / 2) - 1
    return u_n, v_n, X[:, 2]


def homo_warping_3D_with_mask(src_fea, src_proj, ref_proj, depth_values):
    """warping.py:69-109.  ``src_fea [B,C,H,W]`` -> ``warped [B,C,D,H,W]``, ``mask [B,D,H,W]`` bool
    (True = sample left the source frustum: |u_n|>1 or |v_n|>1 or z<=0)."""
    B, C, H, W = src_fea.shape
    D = depth_values.shape[1]
    with torch.no_grad():
        u_n, v_n, z = sweep_coordinates(src_proj, ref_proj, depth_values, H, W)
        grid = torch.stack([u_n, v_n], dim=3)                                            # [B,D,HW,2]
    outside = (u_n > 1) | (u_n < -1) | (v_n > 1) | (v_n < -1) | (z <= 0)
    warped = F.grid_sample(src_fea, grid.reshape(B, D * H, W, 2), mode="bilinear", padding_mode="zeros",
                           align_corners=True)
    return warped.reshape(B, C, D, H, W), outside.reshape(B, D, H, W)


def homo_warping_3D(src_fea, src_proj, ref_proj, depth_values):
    """warping.py:155-189 — same math without the mask."""
    return homo_warping_3D_with_mask(src_fea, src_proj, ref_proj, depth_values)[0]


# --------------------------------------------------------------------------------------
# a3: group-wise correlation and eval-only similarity  (mvsformer_model.py:75-85)
# --------------------------------------------------------------------------------------
def group_correlation(ref_feat: torch.Tensor, warped: torch.Tensor, G: int) -> torch.Tensor:
    """``in_prod[b,g,d,p] = mean_j ref[b,g*C/G+j,p] * warped[b,g*C/G+j,d,p]`` -> ``[B,G,D,H,W]``."""
    B, C, D, H, W = warped.shape
    wv = warped.reshape(B, G, C // G, D, H, W)
    rv = ref_feat.reshape(B, G, C // G, 1, H, W).to(torch.float32)
    return (rv * wv).mean(dim=2)


def group_similarity(ref_feat: torch.Tensor, warped: torch.Tensor, G: int) -> torch.Tensor:
    """mvsformer_model.py:81-84: L2-normalize both volumes along the *group* axis (F.normalize dim=1,
    eps 1e-12), multiply, mean over channels-in-group, sum over groups -> ``[B,D,H,W]``."""
    B, C, D, H, W = warped.shape
    wv = warped.reshape(B, G, C // G, D, H, W)
    rv = ref_feat.reshape(B, G, C // G, 1, H, W).to(torch.float32).expand(-1, -1, -1, D, -1, -1)
    s = F.normalize(rv, dim=1) * F.normalize(wv, dim=1)
    return s.mean(dim=2).sum(dim=1)


# --------------------------------------------------------------------------------------
# a4: entropy-driven visibility weight  (mvsformer_model.py:37,87-91; module.py:168-197)
# --------------------------------------------------------------------------------------
def view_entropy(in_prod: torch.Tensor) -> torch.Tensor:
    """``-sum_d s*log(s+1e-7)`` with ``s = softmax_d(sum_g in_prod)`` -> ``[B,1,H,W]``."""
    s = F.softmax(in_prod.sum(dim=1), dim=1)
    return (-s * torch.log(s + 1e-7)).sum(dim=1, keepdim=True)


def _bn(x, sd, prefix, training=False):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=training, momentum=0.1, eps=BN_EPS)


def vis_net(entropy: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "vis", training=False) -> torch.Tensor:
    """``ConvBnReLU(1,16) -> ConvBnReLU(16,16) -> ConvBnReLU(16,8) -> Conv2d(8,1,1) -> Sigmoid``."""
    x = entropy
    for i in range(3):
        x = F.conv2d(x, sd["%s.%d.conv.weight" % (prefix, i)], None, stride=1, padding=1)
        x = F.relu(_bn(x, sd, "%s.%d.bn" % (prefix, i), training))
    x = F.conv2d(x, sd[prefix + ".3.weight"], sd[prefix + ".3.bias"])
    return torch.sigmoid(x)


# --------------------------------------------------------------------------------------
# a5/a6: 3-D U-Net regularizers  (module.py:83-165, 469-505, 550-594)
# --------------------------------------------------------------------------------------
def _cbr3d(x, sd, prefix, stride, training=False):
    x = F.conv3d(x, sd[prefix + ".conv.weight"], None, stride=stride, padding=1)
    return F.relu(_bn(x, sd, prefix + ".bn", training))


def cost_reg_net(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "cost_reg", training=False,
                 last_layer: bool = True) -> torch.Tensor:
    """``CostRegNet`` (module.py:469-505): stride-2 encoder, ``Deconv3d`` (keys ``convN.conv/.bn``) decoder with
    ``output_padding=1``, skips added after the ReLU, ``prob`` = 3x3x3 conv without bias."""
    p = prefix + "."
    c0 = x
    c2 = _cbr3d(_cbr3d(c0, sd, p + "conv1", 2, training), sd, p + "conv2", 1, training)
    c4 = _cbr3d(_cbr3d(c2, sd, p + "conv3", 2, training), sd, p + "conv4", 1, training)
    y = _cbr3d(_cbr3d(c4, sd, p + "conv5", 2, training), sd, p + "conv6", 1, training)

    def up(t, name):
        t = F.conv_transpose3d(t, sd[p + name + ".conv.weight"], None, stride=2, padding=1, output_padding=1)
        return F.relu(_bn(t, sd, p + name + ".bn", training))

    y = c4 + up(y, "conv7")
    y = c2 + up(y, "conv9")
    inner = c0 if (p + "inner.weight") not in sd else F.conv3d(c0, sd[p + "inner.weight"], sd[p + "inner.bias"])
    y = inner + up(y, "conv11")
    if last_layer:
        y = F.conv3d(y, sd[p + "prob.weight"], None, stride=1, padding=1)
    return y


def cost_reg_net_3d(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "cost_reg", training=False) -> torch.Tensor:
    """``CostRegNet3D`` (module.py:550-594): stride (1,2,2), depth kept; decoder is
    ``Sequential(ConvTranspose3d, BN, ReLU)`` (keys ``convN.0/.1``) with ``output_padding=(0,1,1)``;
    ``prob`` = 1x1x1 conv with bias."""
    p = prefix + "."
    s = (1, 2, 2)
    c0 = x
    c2 = _cbr3d(_cbr3d(c0, sd, p + "conv1", s, training), sd, p + "conv2", 1, training)
    c4 = _cbr3d(_cbr3d(c2, sd, p + "conv3", s, training), sd, p + "conv4", 1, training)
    y = _cbr3d(_cbr3d(c4, sd, p + "conv5", s, training), sd, p + "conv6", 1, training)

    def up(t, name):
        t = F.conv_transpose3d(t, sd[p + name + ".0.weight"], None, stride=s, padding=1, output_padding=(0, 1, 1))
        return F.relu(_bn(t, sd, p + name + ".1", training))

    y = c4 + up(y, "conv7")
    y = c2 + up(y, "conv9")
    inner = c0 if (p + "inner.weight") not in sd else F.conv3d(c0, sd[p + "inner.weight"], sd[p + "inner.bias"])
    y = inner + up(y, "conv11")
    return F.conv3d(y, sd[p + "prob.weight"], sd[p + "prob.bias"])


# --------------------------------------------------------------------------------------
# a7: depth heads  (module.py:597-619; mvsformer_model.py:110-125)
# --------------------------------------------------------------------------------------
def depth_regression(p: torch.Tensor, depth_values: torch.Tensor) -> torch.Tensor:
    if depth_values.dim() <= 2:
        depth_values = depth_values.reshape(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, dim=1)


def conf_regression(p: torch.Tensor, n: int = 4) -> torch.Tensor:
    """module.py:606-619: sum of ``n`` neighbouring probabilities around the (floored) expected index."""
    D = p.shape[1]
    with torch.no_grad():
        lo = n // 2 if n % 2 == 1 else n // 2 - 1
        padded = F.pad(p.unsqueeze(1), pad=[0, 0, 0, 0, lo, n // 2])
        win = n * F.avg_pool3d(padded, (n, 1, 1), stride=1, padding=0).squeeze(1)
        idx = depth_regression(p, torch.arange(D, dtype=torch.float32).reshape(1, D)).long().clamp(0, D - 1)
        return torch.gather(win, 1, idx.unsqueeze(1)).squeeze(1)


# --------------------------------------------------------------------------------------
# a10: hypothesis schedulers  (module.py:633-653)
# --------------------------------------------------------------------------------------
def init_inverse_range(cur_depth: torch.Tensor, ndepths: int, H: int, W: int) -> torch.Tensor:
    """Uniform in inverse depth between ``cur_depth[:,0]`` (near) and ``cur_depth[:,-1]`` (far); index 0 is FAR."""
    inv_near = 1.0 / cur_depth[:, 0]
    inv_far = 1.0 / cur_depth[:, -1]
    itv = torch.arange(ndepths, dtype=cur_depth.dtype).reshape(1, -1, 1, 1).repeat(1, 1, H, W) / (ndepths - 1)
    inv = inv_far[:, None, None, None] + (inv_near - inv_far)[:, None, None, None] * itv
    return 1.0 / inv


def schedule_inverse_range(depth: torch.Tensor, depth_hypo: torch.Tensor, ndepths: int, split_itv: float,
                           H: int, W: int) -> torch.Tensor:
    """Centre the next stage's inverse-depth samples on the previous depth +- ``split_itv`` previous intervals
    (interval measured between hypotheses 1 and 2), then trilinear x2 upsampling with align_corners=True."""
    last_itv = 1.0 / depth_hypo[:, 2] - 1.0 / depth_hypo[:, 1]
    inv_min = 1.0 / depth + split_itv * last_itv
    inv_max = 1.0 / depth - split_itv * last_itv
    itv = torch.arange(ndepths, dtype=depth.dtype).reshape(1, -1, 1, 1).repeat(1, 1, H // 2, W // 2) / (ndepths - 1)
    inv = inv_max[:, None] + (inv_min - inv_max)[:, None] * itv
    inv = F.interpolate(inv.unsqueeze(1), [ndepths, H, W], mode="trilinear", align_corners=True).squeeze(1)
    return 1.0 / inv


def mixup_head(prob: torch.Tensor, depth_values: torch.Tensor):
    """mvsformer_model.py:126-136 (depth_type 'mixup_ce'): the adjacent hypothesis pair with the largest probability mass;
    confidence = that mass, depth = the pair's hypotheses mixed by the pair probabilities renormalised with +1e-7."""
    left, right = prob[:, :-1], prob[:, 1:]
    conf, idx = torch.max(left + right, dim=1)
    norm = left + right + 1e-7
    mix = depth_values[:, :-1] * (left / norm) + depth_values[:, 1:] * (right / norm)
    return torch.gather(mix, 1, idx.unsqueeze(1)).squeeze(1), conf


# --------------------------------------------------------------------------------------
# StageNet.forward, fusion_type='cnn', all depth_type heads  (mvsformer_model.py:51-160)
# --------------------------------------------------------------------------------------
def stage_forward(features: torch.Tensor, proj: torch.Tensor, depth_values: torch.Tensor, sd: Dict[str, torch.Tensor],
                  *, G: int = 8, ndepth: int, model_th: int = 8, tmp: float = 2.0, training: bool = False,
                  taps: Optional[dict] = None, depth_type: str = "ce") -> Dict[str, torch.Tensor]:
    """``features [B,V,C,H,W]``, ``proj [B,V,2,4,4]``, ``depth_values [B,D,H,W]``.  ``taps`` (optional dict)
    receives the intermediates the golden files pin (per-view in_prod / entropy / vis weight, volume_mean)."""
    ref_feat = features[:, 0].to(torch.float32)
    V = features.shape[1]
    ref_P = compose_projection(proj[:, 0])
    vol_sum, w_sum, sims = 0.0, 0.0, []
    for v in range(1, V):
        src_P = compose_projection(proj[:, v])
        warped, _ = homo_warping_3D_with_mask(features[:, v].to(torch.float32), src_P, ref_P, depth_values)
        ip = group_correlation(ref_feat, warped, G)
        if not training:
            sims.append(group_similarity(ref_feat, warped, G))
        ent = view_entropy(ip.detach())
        w = vis_net(ent, sd, "vis", training)
        if taps is not None:
            taps.setdefault("in_prod", []).append(ip)
            taps.setdefault("entropy", []).append(ent)
            taps.setdefault("vis_weight", []).append(w)
        vol_sum = vol_sum + ip * w.unsqueeze(1)
        w_sum = w_sum + w
    volume_mean = vol_sum / (w_sum.unsqueeze(1) + 1e-6)
    if taps is not None:
        taps["volume_mean"] = volume_mean
    if ndepth <= model_th:
        logits = cost_reg_net_3d(volume_mean, sd, "cost_reg", training)
    else:
        logits = cost_reg_net(volume_mean, sd, "cost_reg", training)
    pre = logits.squeeze(1)
    prob = F.softmax(pre, dim=1)
    if depth_type in ("ce", "was"):                          # mvsformer_model.py:113-125
        if training:
            idx = prob.max(dim=1)[1]
            depth = torch.gather(depth_values, 1, idx.unsqueeze(1)).squeeze(1)
        else:
            depth = depth_regression(F.softmax(pre * tmp, dim=1), depth_values)
        conf = prob.max(dim=1)[0]
    elif depth_type == "mixup_ce":                           # mvsformer_model.py:126-136
        depth, conf = mixup_head(prob, depth_values)
    else:                                                    # mvsformer_model.py:137-146
        depth = depth_regression(prob, depth_values)
        n = {32: 4, 16: 3, 8: 2}.get(ndepth, 4 if ndepth >= 32 else 0)
        conf = conf_regression(prob, n) if n else prob.max(dim=1)[0]
    out = {"depth": depth, "prob_volume": prob, "photometric_confidence": conf.detach(),
           "depth_values": depth_values, "prob_volume_pre": pre}
    if not training:
        tot = torch.stack(sims, dim=1).sum(dim=1)
        if taps is not None:
            taps["similarity_sum"] = tot
        out["sim_depth"] = torch.gather(depth_values, 1, tot.argmax(dim=1, keepdim=True)).squeeze(1)
    return out


# --------------------------------------------------------------------------------------
# a9: the cascade the top models run around StageNet  (mvsformer_model.py:273-306 / 410-449)
# --------------------------------------------------------------------------------------
def cascade_forward(features: Dict[str, torch.Tensor], proj: Dict[str, torch.Tensor], depth_values: torch.Tensor,
                    stage_sds: Sequence[Dict[str, torch.Tensor]], *, ndepths: Sequence[int],
                    depth_interals_ratio: Sequence[float], G: int = 8, model_th: int = 8, tmp=2.0,
                    training: bool = False) -> Dict[str, object]:
    """Inverse-depth cascade: hypotheses -> StageNet -> nearest-upsampled confidences averaged over stages."""
    n = len(ndepths)
    Hf, Wf = features["stage%d" % n].shape[-2:]
    B = depth_values.shape[0]
    conf_acc = torch.zeros(B, Hf, Wf, dtype=torch.float32)
    outputs: Dict[str, object] = {}
    prev = None
    for i in range(n):
        f = features["stage%d" % (i + 1)]
        H, W = f.shape[-2:]
        if i == 0:
            hyp = init_inverse_range(depth_values, ndepths[0], H, W)
        else:
            hyp = schedule_inverse_range(prev["depth"].detach(), prev["depth_values"], ndepths[i],
                                         depth_interals_ratio[i], H, W)
        t = tmp[i] if isinstance(tmp, (list, tuple)) else tmp
        prev = stage_forward(f, proj["stage%d" % (i + 1)], hyp, stage_sds[i], G=G, ndepth=ndepths[i],
                             model_th=model_th, tmp=t, training=training)
        conf = prev["photometric_confidence"]
        if conf.shape[-2:] != (Hf, Wf):
            conf = F.interpolate(conf.unsqueeze(1), [Hf, Wf], mode="nearest").squeeze(1)
            prev["photometric_confidence"] = conf
        conf_acc = conf_acc + conf
        outputs["stage%d" % (i + 1)] = prev
    outputs["refined_depth"] = prev["depth"]
    outputs["photometric_confidence"] = conf_acc / n
    outputs["depth"] = prev["depth"]
    return outputs
