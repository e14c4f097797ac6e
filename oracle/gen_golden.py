#!/usr/bin/env python
"""ORACLE — test infrastructure.  Generates ``tests/golden/*.npz`` by running the REAL reference.

Runs only in the build container, where the read-only reference checkout is
mounted at /root/reference; the GPU box never sees the reference, only the
vectors written here.  Nothing from the reference is copied: its modules are
imported (with stub modules for the absent third-party packages ``timm``,
``torchvision`` and ``omegaconf`` that unrelated files import at module load)
and called on seeded synthetic inputs; inputs and outputs are saved.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Vectors (all float32 unless noted; inputs that are feature maps are stored as
float16-exact values to halve the files):
  state_dict_shapes.json   key -> shape of StageNet.state_dict() for both regularizer kinds
  warp_kat.npz             hand-checkable known-answer case of homo_warping_3D_with_mask
  warp_general.npz         general poses, [B,D] and [B,D,H,W] hypotheses, behind-camera / off-frustum samples
  heads.npz                depth_regression, conf_regression, init/schedule_inverse_range
  costreg.npz              CostRegNet / CostRegNet3D alone, eval BN, odd sizes
  stage_costregnet.npz     StageNet with ndepth=16 (CostRegNet), eval + train, intermediate taps
  stage_costregnet3d.npz   StageNet with ndepth=4 (CostRegNet3D), eval + train, intermediate taps
  cascade_v3.npz, cascade_v5.npz   4-stage inverse-depth cascade, 64x64, tmp=[5,5,5,1]
  train_costregnet.npz, train_costregnet3d.npz   train-mode StageNet (B=2): forward, loss=sum(pre*R), all gradients
  fpn_decoder.npz          FPNDecoder (the step before the path), eval BatchNorm: state_dict, encoder outputs, the 4 feature maps
  fpn_decoder_v2.npz       FPNDecoderV2 (TwinMVSNet's decoder), eval BatchNorm: state_dict (conv weights float16-exact), 7 input maps, 4 outputs
  fpn_encoder.npz          FPNEncoder, eval BatchNorm: state_dict (conv weights float16-exact), image, the 4 encoder outputs
  dinomvsnet_e2e.npz       the REAL DINOMVSNet (configs/config_mvsformer-p.json) in eval mode, images -> depth: 3 views of 128x192
  dinomvsnet_shapes.json   key -> shape of its state_dict() (weights are rebuilt from a seed: oracle/weights.make_model_state_dict)
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m


_stub("timm")
_stub("timm.models")
_stub("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x), trunc_normal_=nn.init.trunc_normal_)
_stub("timm.models.vision_transformer", Block=nn.Module)
_stub("torchvision")
_stub("torchvision.utils")
_stub("omegaconf", OmegaConf=object)

import warnings  # noqa: E402

warnings.filterwarnings("ignore")
import models.mvsformer_model as ref_mm  # noqa: E402
from models.module import (CostRegNet, CostRegNet3D, conf_regression, depth_regression,  # noqa: E402
                           init_inverse_range, schedule_inverse_range)
from models.warping import homo_warping_3D, homo_warping_3D_with_mask  # noqa: E402

from mvsformer_amd import synth  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
ARGS = dict(base_ch=8, fusion_type="cnn", depth_type="ce", model_th=8)


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def f16exact(t):
    return t.to(torch.float16).to(torch.float32)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f kB" % (name, os.path.getsize(path) / 1e3))


# ---------------------------------------------------------------------------------------------
def dump_shapes():
    shapes = {}
    for kind, nd in (("stage_costregnet", 16), ("stage_costregnet3d", 4)):
        net = ref_mm.StageNet(dict(ARGS), nd, 0)
        shapes[kind] = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=0)
    return shapes


def gen_warp_kat():
    K = torch.tensor([[100.0, 0, 2.5, 0], [0, 100.0, 1.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    E_ref = torch.eye(4)
    E_src = torch.eye(4)
    E_src[0, 3] = 1.0
    ref_proj = (K @ E_ref).unsqueeze(0)
    src_proj = (K @ E_src).unsqueeze(0)
    src = torch.arange(2 * 4 * 6, dtype=torch.float32).view(1, 2, 4, 6)
    depth = torch.tensor([[100.0, 200.0, 50.0, -10.0]])
    warped, mask = homo_warping_3D_with_mask(src, src_proj, ref_proj, depth)
    ident, imask = homo_warping_3D_with_mask(src, ref_proj, ref_proj, depth[:, :1])
    save("warp_kat.npz", src=np32(src), src_proj=np32(src_proj), ref_proj=np32(ref_proj), depth=np32(depth),
         warped=np32(warped), mask=mask.numpy(), ident=np32(ident), ident_mask=imask.numpy())


def _general_projs(B, g):
    """World->image 4x4 projections with generic rotations/translations (scene ~ 2-6 units deep)."""
    Ps = []
    for _ in range(2):
        P = torch.zeros(B, 4, 4)
        for b in range(B):
            a = 0.25 * (torch.rand(3, generator=g) - 0.5)
            Rm = synth._rot_xyz(float(a[0]), float(a[1]), float(a[2])).float()
            t = 0.6 * (torch.rand(3, generator=g) - 0.5)
            K = torch.tensor([[22.0, 0, 9.5], [0, 21.0, 5.5], [0, 0, 1]])
            P[b, :3, :3] = K @ Rm
            P[b, :3, 3] = K @ t
            P[b, 3, 3] = 1
        Ps.append(P)
    return Ps


def gen_warp_general():
    g = torch.Generator().manual_seed(11)
    B, C, H, W, D = 2, 16, 12, 20, 6
    src = f16exact(torch.randn(B, C, H, W, generator=g))
    ref_proj, src_proj = _general_projs(B, g)
    depth_bd = torch.tensor([[1.5, 2.5, 4.0, 7.0, -1.0, 0.05]]).repeat(B, 1)
    depth_bd[1] *= 1.3
    depth_map = depth_bd.view(B, D, 1, 1) * (1.0 + 0.2 * torch.rand(B, D, H, W, generator=g))
    w1, m1 = homo_warping_3D_with_mask(src, src_proj, ref_proj, depth_bd)
    w2, m2 = homo_warping_3D_with_mask(src, src_proj, ref_proj, depth_map)
    w3 = homo_warping_3D(src, src_proj, ref_proj, depth_map)
    assert torch.equal(w2, w3)
    save("warp_general.npz", src=np32(src), src_proj=np32(src_proj), ref_proj=np32(ref_proj),
         depth_bd=np32(depth_bd), depth_map=np32(depth_map), warped_bd=np32(w1), mask_bd=m1.numpy(),
         warped_map=np32(w2), mask_map=m2.numpy())


def gen_heads():
    g = torch.Generator().manual_seed(5)
    B, D, H, W = 2, 8, 6, 10
    logits = 3.0 * torch.randn(B, D, H, W, generator=g)
    p = F.softmax(logits, dim=1)
    dv_map = 400.0 + 500.0 * torch.rand(B, D, H, W, generator=g)
    dv_bd = 400.0 + 60.0 * torch.arange(D, dtype=torch.float32).view(1, D).repeat(B, 1)
    out = dict(logits=np32(logits), p=np32(p), dv_map=np32(dv_map), dv_bd=np32(dv_bd),
               reg_map=np32(depth_regression(p, dv_map)), reg_bd=np32(depth_regression(p, dv_bd)))
    for n in (2, 3, 4):
        out["conf_n%d" % n] = np32(conf_regression(p, n=n))
    # schedulers
    cur = synth.depth_range(batch=B)
    cur[1] = cur[1] * 1.1
    init = init_inverse_range(cur, 32, "cpu", torch.float32, 8, 12)
    prev_depth = 500.0 + 300.0 * torch.rand(B, 8, 12, generator=g)
    sched = schedule_inverse_range(prev_depth, init, 16, 2.67, 16, 24)
    out.update(cur_depth=np32(cur), init_inv=np32(init), prev_depth=np32(prev_depth), sched_inv=np32(sched))
    save("heads.npz", **out)


def gen_costreg(shapes):
    g = torch.Generator().manual_seed(21)
    out = {}
    for kind, cls, seed, shp in (("costregnet", CostRegNet, 101, (1, 8, 16, 8, 24)),
                                 ("costregnet3d", CostRegNet3D, 102, (2, 8, 3, 16, 40))):
        sd_stage = make_state_dict(shapes["stage_" + kind], seed)
        sd = {k[len("cost_reg."):]: v for k, v in sd_stage.items() if k.startswith("cost_reg.")}
        net = cls(8, 8)
        net.load_state_dict(sd, strict=True)
        net.eval()
        x = torch.randn(shp, generator=g)
        with torch.no_grad():
            y = net(x)
        out[kind + "_x"] = np32(x)
        out[kind + "_y"] = np32(y)
        out[kind + "_seed"] = np.int64(seed)
    save("costreg.npz", **out)


def _stage_case(kind, ndepth, C, seed, shapes):
    """One StageNet golden: small photo-consistent scene rendered at 1/8 of a 128x192 image."""
    V, Hf, Wf, scale = 4, 128, 192, 8
    scene = synth.make_scene(V, Hf, Wf, seed=seed)
    feats = f16exact(synth.render_features(scene, scale, C, noise=0.05))
    proj = synth.proj_matrices(scene, (scale,))["stage1"]
    H, W = Hf // scale, Wf // scale
    if ndepth > 8:
        hyp = init_inverse_range(synth.depth_range(1), ndepth, "cpu", torch.float32, H, W)
    else:
        # narrow per-pixel hypotheses around the true plane, as a later cascade stage sees them
        g = torch.Generator().manual_seed(seed + 77)
        z = synth.plane_depth(scene, scale)
        centre = (z * (1.0 + 0.01 * torch.randn(z.shape, generator=g))).unsqueeze(0)
        inv = 1.0 / centre.unsqueeze(1) + torch.linspace(-1, 1, ndepth).view(1, -1, 1, 1) * 4e-5
        hyp = 1.0 / inv
    sd = make_state_dict(shapes["stage_" + kind], 1000 + seed)
    net = ref_mm.StageNet(dict(ARGS), ndepth, 1)
    net.load_state_dict(sd, strict=True)
    out = dict(features=feats.numpy().astype(np.float16), proj=np32(proj), depth_values=np32(hyp),
               weight_seed=np.int64(1000 + seed), ndepth=np.int64(ndepth), tmp=np.float32(5.0))

    # taps: re-run the reference's own building blocks the way StageNet.forward chains them
    net.eval()
    with torch.no_grad():
        o = net(feats, proj, hyp, tmp=5.0)
        for k in ("depth", "prob_volume", "photometric_confidence", "prob_volume_pre", "sim_depth"):
            out["eval_" + k] = np32(o[k])
        ref_feat = feats[:, 0]
        ref_pair = proj[:, 0]
        ref_new = ref_pair[:, 0].clone()
        ref_new[:, :3, :4] = torch.matmul(ref_pair[:, 1, :3, :3], ref_pair[:, 0, :3, :4])
        ents, ws, vol, wsum, sims = [], [], 0.0, 0.0, 0.0
        for v in range(1, V):
            sp = proj[:, v]
            src_new = sp[:, 0].clone()
            src_new[:, :3, :4] = torch.matmul(sp[:, 1, :3, :3], sp[:, 0, :3, :4])
            warped, _ = homo_warping_3D_with_mask(feats[:, v], src_new, ref_new, hyp)
            B, Cc, D, Hh, Ww = warped.shape
            wv = warped.view(B, 8, Cc // 8, D, Hh, Ww)
            rv = ref_feat.view(B, 8, Cc // 8, 1, Hh, Ww).repeat(1, 1, 1, D, 1, 1)
            ip = (rv * wv).mean(dim=2)
            sims = sims + (F.normalize(rv, dim=1) * F.normalize(wv, dim=1)).mean(dim=2).sum(dim=1)
            sn = F.softmax(ip.sum(dim=1), dim=1)
            ent = (-sn * torch.log(sn + 1e-7)).sum(dim=1, keepdim=True)
            w = net.vis(ent)
            if v == 1:
                out["tap_in_prod_v1"] = np32(ip)
            ents.append(ent)
            ws.append(w)
            vol = vol + ip * w.unsqueeze(1)
            wsum = wsum + w
        vm = vol / (wsum.unsqueeze(1) + 1e-6)
        chk = net.cost_reg(vm).squeeze(1)
        assert torch.allclose(chk, o["prob_volume_pre"], atol=1e-5), "tap chain diverged from StageNet.forward"
        out["tap_entropy"] = np32(torch.cat(ents, dim=1))
        out["tap_vis_weight"] = np32(torch.cat(ws, dim=1))
        out["tap_volume_mean"] = np32(vm)
        out["tap_similarity_sum"] = np32(sims)
    # training-mode forward (batch-stat BN, argmax depth, no similarity branch); BN buffers restored afterwards
    net.load_state_dict(sd, strict=True)
    net.train()
    with torch.no_grad():
        o = net(feats, proj, hyp, tmp=5.0)
        for k in ("depth", "photometric_confidence", "prob_volume_pre"):
            out["train_" + k] = np32(o[k])
        assert "sim_depth" not in o
    save("stage_%s.npz" % kind, **out)


def gen_other_heads(shapes):
    """heads_other.npz: the real reference StageNet run with depth_type 'mixup_ce' and 'reg' (mvsformer_model.py:126-146) on the
    inputs / weights of the two committed stage cases (eval and train mode; the heads do not depend on the mode, BatchNorm does)."""
    out = {}
    for kind, nd in (("costregnet", 16), ("costregnet3d", 4), ("costregnet", 32), ("costregnet3d", 8)):
        g = np.load(os.path.join(OUT, "stage_%s.npz" % kind))
        feats, proj = torch.from_numpy(g["features"].astype(np.float32)), torch.from_numpy(g["proj"])
        hyp = torch.from_numpy(g["depth_values"])
        if nd != hyp.shape[1]:            # the 32- and 8-plane windows of conf_regression: resample the hypotheses in inverse depth
            inv = F.interpolate((1.0 / hyp).unsqueeze(1), [nd, hyp.shape[2], hyp.shape[3]], mode="trilinear", align_corners=True).squeeze(1)
            hyp = 1.0 / inv
            out["hyp_%s_%d" % (kind, nd)] = np32(hyp)
        sd = make_state_dict(shapes["stage_" + kind], int(g["weight_seed"]))
        for dt in ("mixup_ce", "reg"):
            net = ref_mm.StageNet(dict(ARGS, depth_type=dt), nd, 0)
            net.load_state_dict(sd, strict=True)
            for mode in ("eval", "train"):
                net.load_state_dict(sd, strict=True)
                net.train(mode == "train")
                with torch.no_grad():
                    o = net(feats, proj, hyp, tmp=5.0)
                for k in ("depth", "photometric_confidence"):
                    out["%s_%d_%s_%s_%s" % (kind, nd, dt, mode, k)] = np32(o[k])
    save("heads_other.npz", **out)


def gen_cascade(V, shapes, seed):
    ndepths, ratios, tmps = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0], [5.0, 5.0, 5.0, 1.0]
    Hf = Wf = 64
    feats, proj, dv, scene = synth.make_inputs(V, Hf, Wf, seed=seed)
    feats = {k: f16exact(v) for k, v in feats.items()}
    nets, seeds = [], []
    for i, nd in enumerate(ndepths):
        kind = "stage_costregnet3d" if nd <= 8 else "stage_costregnet"
        s = 2000 + 10 * seed + i
        net = ref_mm.StageNet(dict(ARGS), nd, i)
        net.load_state_dict(make_state_dict(shapes[kind], s), strict=True)
        net.eval()
        nets.append(net)
        seeds.append(s)
    out = dict(depth_range=np32(dv), weight_seeds=np.array(seeds, dtype=np.int64), ndepths=np.array(ndepths),
               ratios=np.array(ratios, dtype=np.float32), tmps=np.array(tmps, dtype=np.float32), scene_seed=np.int64(seed))
    for k, v in feats.items():
        out["features_" + k] = v.numpy().astype(np.float16)
    for k, v in proj.items():
        out["proj_" + k] = np32(v)
    # the cascade loop of DINOMVSNet/TwinMVSNet.forward, driven with the reference's own functions
    B = 1
    prob_maps = torch.zeros(B, Hf, Wf)
    prev = None
    with torch.no_grad():
        for i, nd in enumerate(ndepths):
            f = feats["stage%d" % (i + 1)]
            H, W = f.shape[-2:]
            if i == 0:
                hyp = init_inverse_range(dv, nd, "cpu", torch.float32, H, W)
            else:
                hyp = schedule_inverse_range(prev["depth"].detach(), prev["depth_values"], nd, ratios[i], H, W)
            prev = nets[i].forward(f, proj["stage%d" % (i + 1)], hyp, tmp=tmps)
            conf = prev["photometric_confidence"]
            if conf.shape[1] != Hf or conf.shape[2] != Wf:
                conf = F.interpolate(conf.unsqueeze(1), [Hf, Wf], mode="nearest").squeeze(1)
            prob_maps += conf
            for k in ("depth", "photometric_confidence", "prob_volume_pre", "depth_values", "sim_depth"):
                out["s%d_%s" % (i + 1, k)] = np32(prev[k])
    out["refined_depth"] = np32(prev["depth"])
    out["photometric_confidence"] = np32(prob_maps / len(ndepths))
    out["true_depth"] = np32(synth.plane_depth(scene, 1))
    save("cascade_v%d.npz" % V, **out)


if __name__ == "__main__" and os.environ.get("GEN_EVAL", "1") == "1":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    shapes = dump_shapes()
    gen_warp_kat()
    gen_warp_general()
    gen_heads()
    gen_costreg(shapes)
    _stage_case("costregnet", 16, 16, 3, shapes)
    _stage_case("costregnet3d", 4, 8, 4, shapes)
    gen_cascade(3, shapes, 6)
    gen_cascade(5, shapes, 7)


if __name__ == "__main__" and os.environ.get("GEN_OTHER_HEADS", "1") == "1":
    gen_other_heads(json.load(open(os.path.join(OUT, "state_dict_shapes.json"))))


# ---------------------------------------------------------------------------------------------
# training / autograd goldens: forward in train mode (batch-statistics BN, argmax depth),
# loss = sum(prob_volume_pre * R) with a fixed random R, gradients w.r.t. features and every parameter.
# ---------------------------------------------------------------------------------------------
def gen_train_grads(kind, ndepth, C, seed, shapes, B=2):
    V, Hf, Wf, scale = 3, 128, 192, 8
    feats_l, proj_l, hyp_l = [], [], []
    for bi in range(B):
        scene = synth.make_scene(V, Hf, Wf, seed=seed + 31 * bi)
        feats_l.append(f16exact(synth.render_features(scene, scale, C, noise=0.05)))
        proj_l.append(synth.proj_matrices(scene, (scale,))["stage1"])
        H, W = Hf // scale, Wf // scale
        if ndepth > 8:
            hyp_l.append(init_inverse_range(synth.depth_range(1), ndepth, "cpu", torch.float32, H, W))
        else:
            z = synth.plane_depth(scene, scale)
            inv = 1.0 / z[None, None] + torch.linspace(-1, 1, ndepth).view(1, -1, 1, 1) * 4e-5
            hyp_l.append(1.0 / inv)
    feats = torch.cat(feats_l, 0).requires_grad_(True)
    proj, hyp = torch.cat(proj_l, 0), torch.cat(hyp_l, 0)
    sd = make_state_dict(shapes["stage_" + kind], 3000 + seed)
    net = ref_mm.StageNet(dict(ARGS), ndepth, 0)
    net.load_state_dict(sd, strict=True)
    net.train()
    o = net(feats, proj, hyp, tmp=5.0)
    g = torch.Generator().manual_seed(99 + seed)
    R = torch.randn(o["prob_volume_pre"].shape, generator=g)
    loss = (o["prob_volume_pre"] * R).sum()
    loss.backward()
    out = dict(features=feats.detach().numpy().astype(np.float16), proj=np32(proj), depth_values=np32(hyp), R=np32(R),
               weight_seed=np.int64(3000 + seed), ndepth=np.int64(ndepth), loss=np.float64(loss.item()),
               prob_volume_pre=np32(o["prob_volume_pre"]), depth=np32(o["depth"]), grad_features=np32(feats.grad))
    full = ("vis.", "cost_reg.prob", "cost_reg.conv1.", "cost_reg.conv11.", ".bn.", ".1.weight", ".1.bias")
    for name, p in net.named_parameters():
        gr = p.grad
        out["gsum_" + name] = np.float64(gr.double().sum().item())
        out["gabs_" + name] = np.float64(gr.double().abs().sum().item())
        if any(t in name for t in full) and gr.numel() <= 4096:
            out["grad_" + name] = np32(gr)
    for name, buf in net.named_buffers():
        if "running" in name and ("conv1." in name or "conv7" in name or "vis.0" in name):
            out["buf_" + name] = np32(buf)
    save("train_%s.npz" % kind, **out)


if __name__ == "__main__" and os.environ.get("GEN_TRAIN", "1") == "1":
    shapes = json.load(open(os.path.join(OUT, "state_dict_shapes.json")))
    gen_train_grads("costregnet", 16, 16, 13, shapes)
    gen_train_grads("costregnet3d", 4, 8, 14, shapes)


def gen_fusion():
    """fusion.npz: misc/fusion.py get_reproj / vis_filter / ave_fusion + the point back-projection of test.py:425-434,
    run from the real reference on CPU (its get_pixel_grids hard-codes .cuda(); Tensor.cuda is made a no-op here)."""
    from oracle import ref_fusion
    torch.Tensor.cuda = lambda self, *a, **k: self
    import misc.fusion as rf
    out = {}
    for tag, kw, thr in (("a", dict(n=1, v=4, h=48, w=64, seed=0), (1.0, 0.01, 3)),
                         ("b", dict(n=2, v=3, h=40, w=56, seed=1, noise=0.008), (0.5, 0.02, 2))):
        case = ref_fusion.make_fusion_case(**kw)
        for k, t in case.items():
            out["%s_%s" % (tag, k)] = np32(t)
        reproj, in_range = rf.get_reproj(case["ref_depth"], case["src_depths"], case["ref_cam"], case["src_cams"])
        masks, mask = rf.vis_filter(case["ref_depth"], reproj, in_range, *thr)
        ave = rf.ave_fusion(case["ref_depth"], reproj, masks)
        idx_img = rf.get_pixel_grids(*ave.size()[-2:]).unsqueeze(0)
        points = rf.idx_cam2world(rf.idx_img2cam(idx_img, ave, case["ref_cam"]), case["ref_cam"])[..., :3, 0].permute(0, 3, 1, 2)
        out[tag + "_thresholds"] = np.asarray(thr, np.float32)
        out[tag + "_reproj_xyd"] = np32(reproj)
        out[tag + "_in_range"] = np32(in_range)
        out[tag + "_masks"] = np32(masks)
        out[tag + "_mask"] = mask.numpy()
        out[tag + "_ref_depth_ave"] = np32(ave)
        out[tag + "_points"] = np32(points)
        conf = torch.rand(kw["n"], 3, 1, kw["h"], kw["w"], generator=torch.Generator().manual_seed(5))
        out[tag + "_conf"] = np32(conf)
        out[tag + "_prob_mask"] = rf.prob_filter(conf, [0.3, 0.5, 0.2]).numpy()
        print(tag, "kept", float(mask.float().mean()), "masks", float(masks.mean()))
        # dynamic variant, test.py:494-514 verbatim in effect (reference functions called, driver arithmetic restated here)
        bases = (4, 1300) if tag == "a" else (3, 400)
        v = kw["v"]
        dreproj = rf.get_reproj_dynamic(case["ref_depth"], case["src_depths"], case["ref_cam"], case["src_cams"])
        dmasks, dmask = rf.vis_filter_dynamic(case["ref_depth"], dreproj, dist_base=bases[0], rel_diff_base=bases[1])
        out[tag + "_dyn_bases"] = np.asarray(bases, np.float32)
        out[tag + "_dyn_reproj_xyd"] = np32(dreproj)
        out[tag + "_dyn_masks"] = dmasks.numpy()
        out[tag + "_dyn_vis_mask"] = dmask.numpy()
        rdepth = dreproj[:, :, -1].clone()
        rdepth[~dmask.squeeze(2)] = 0
        sums, vsum = dmasks.sum(dim=1), dmask.sum(dim=1)
        dave = (torch.sum(rdepth, dim=1, keepdim=True) + case["ref_depth"]) / (vsum + 1)
        geo = vsum >= v + 1
        for i in range(2, v + 1):
            geo = torch.logical_or(geo, sums[:, i - 2] >= i)      # [n,1,h,w] | [n,h,w] broadcasts as in the reference
        dpoints = rf.idx_cam2world(rf.idx_img2cam(idx_img, dave, case["ref_cam"]), case["ref_cam"])[..., :3, 0].permute(0, 3, 1, 2)
        out[tag + "_dyn_geo_mask"] = geo.numpy()
        out[tag + "_dyn_ref_depth_ave"] = np32(dave)
        out[tag + "_dyn_points"] = np32(dpoints)
        print(tag, "dynamic kept", float(geo.float().mean()), "level-v views", float(dmask.float().mean()))
    save("fusion.npz", **out)


if __name__ == "__main__" and os.environ.get("GEN_FUSION", "1") == "1":
    gen_fusion()


def gen_io():
    """io.npz: bytes the reference's own writers produce (datasets/data_io.py save_pfm; test.py write_cam) and what its
    readers return (read_pfm, read_camera_parameters, read_pair_file).  test.py cannot be imported (argparse and model
    imports at module level), so the three plain functions are pulled out of its AST and executed as they stand."""
    import ast
    import tempfile
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_data_io", os.path.join(REF, "datasets", "data_io.py"))   # skip datasets/__init__
    ref_io = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_io)
    read_pfm, save_pfm = ref_io.read_pfm, ref_io.save_pfm
    tree = ast.parse(open(os.path.join(REF, "test.py")).read())
    ns = {"np": np}
    wanted = ("write_cam", "read_camera_parameters", "read_pair_file")
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
    exec(compile(mod, "test.py", "exec"), ns)
    rng = np.random.default_rng(0)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        grey = (rng.random((5, 7)) * 900 + 100).astype(np.float32)
        color = rng.random((4, 6, 3)).astype(np.float32)
        for name, img, scale in (("grey", grey, 1), ("color", color, 2.5), ("grey1", grey[:, :, None], 1)):
            p = os.path.join(d, name + ".pfm")
            save_pfm(p, img, scale)
            out["pfm_%s_in" % name] = img
            out["pfm_%s_bytes" % name] = np.frombuffer(open(p, "rb").read(), np.uint8)
            back, sc = read_pfm(p)
            out["pfm_%s_read" % name] = np.ascontiguousarray(back)
            out["pfm_%s_scale" % name] = np.float64(sc)
        # a big-endian file written by hand, for the reader's other branch
        be = os.path.join(d, "be.pfm")
        with open(be, "wb") as f:
            f.write(b"Pf\n7 5\n1.000000\n")
            np.flipud(grey).astype(">f4").tofile(f)
        out["pfm_be_bytes"] = np.frombuffer(open(be, "rb").read(), np.uint8)
        out["pfm_be_read"] = np.ascontiguousarray(read_pfm(be)[0]).astype(np.float32)
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0] = np.eye(4)
        cam[0, :3, :3] = np.linalg.qr(rng.standard_normal((3, 3)))[0]
        cam[0, :3, 3] = rng.standard_normal(3) * 100
        cam[1, :3, :3] = [[2892.33, 0, 823.205], [0, 2883.175, 619.071], [0, 0, 1]]
        cam[1, 3] = [425.0, 2.65, 192, 933.8]
        p = os.path.join(d, "00000000_cam.txt")
        ns["write_cam"](p, cam)
        out["cam_in"] = cam
        out["cam_text"] = np.frombuffer(open(p, "rb").read(), np.uint8)
        K, E = ns["read_camera_parameters"](p)
        out["cam_read_K"], out["cam_read_E"] = K, E
        pair = "4\n0\n3 1 2036.5 2 1243.9 3 703.2\n1\n2 0 2036.5 2 845.0\n2\n0\n3\n1 0 703.2\n"
        p = os.path.join(d, "pair.txt")
        open(p, "w").write(pair)
        out["pair_text"] = np.frombuffer(pair.encode(), np.uint8)
        out["pair_read"] = np.frombuffer(json.dumps(ns["read_pair_file"](p)).encode(), np.uint8)
    save("io.npz", **out)


if __name__ == "__main__" and os.environ.get("GEN_IO", "1") == "1":
    gen_io()


def gen_ce_loss():
    """ce_loss.npz: models/losses.py ce_loss_stage4 (focal=False) values and d loss / d prob_volume_pre from the reference."""
    from models.losses import ce_loss_stage4
    from oracle import ref_losses
    out = {}
    for tag, inverse in (("inv", True), ("fwd", False)):
        inputs, gts, masks = ref_losses.make_loss_case(seed=3 if inverse else 4, inverse_depth=inverse)
        for k in inputs:
            inputs[k]["prob_volume_pre"].requires_grad_(True)
        w = [1.0, 0.5, 2.0, 1.0]
        losses = ce_loss_stage4(inputs, gts, masks, dlossw=w, focal=False, gamma=0.0, inverse_depth=inverse)
        sum(losses.values()).backward()
        out[tag + "_dlossw"] = np.asarray(w, np.float32)
        for k in inputs:
            out["%s_%s_depth_values" % (tag, k)] = np32(inputs[k]["depth_values"])
            out["%s_%s_logits" % (tag, k)] = np32(inputs[k]["prob_volume_pre"])
            out["%s_%s_gt" % (tag, k)] = np32(gts[k])
            out["%s_%s_mask" % (tag, k)] = np32(masks[k])
            out["%s_%s_loss" % (tag, k)] = np.float64(losses[k].item())
            out["%s_%s_grad" % (tag, k)] = np32(inputs[k]["prob_volume_pre"].grad)
            idx, final = ref_losses.gt_bins(inputs[k]["depth_values"], gts[k], masks[k], inverse)
            print(tag, k, "loss %.5f" % losses[k].item(), "valid %.3f" % final.float().mean().item(), "bins", int(idx.min()), int(idx.max()))
    save("ce_loss.npz", **out)


if __name__ == "__main__" and os.environ.get("GEN_LOSS", "1") == "1":
    gen_ce_loss()


def gen_other_losses():
    """other_losses.npz: models/losses.py mixup_ce_loss_stage4 and reg_loss_stage4 (mask_out_range False and True) values and gradients on
    the SAME seeded cases as ce_loss.npz (oracle/ref_losses.make_loss_case, seeds 3 / 4: the inputs are read from that file by the
    tests); the regression head's depth = sum(softmax(logits) * depth_values) and the per-sample depth interval are stored."""
    from models.losses import mixup_ce_loss_stage4, reg_loss_stage4
    from oracle import ref_losses
    out = {}
    w = [1.0, 0.5, 2.0, 1.0]
    for tag, inverse in (("inv", True), ("fwd", False)):
        inputs, gts, masks = ref_losses.make_loss_case(seed=3 if inverse else 4, inverse_depth=inverse)
        for k in inputs:
            inputs[k]["prob_volume_pre"].requires_grad_(True)
        losses = mixup_ce_loss_stage4(inputs, gts, masks, dlossw=w, inverse_depth=inverse)
        sum(losses.values()).backward()
        for k in inputs:
            out["%s_%s_mixup_loss" % (tag, k)] = np.float64(losses[k].item())
            out["%s_%s_mixup_grad" % (tag, k)] = np32(inputs[k]["prob_volume_pre"].grad)
            print(tag, k, "mixup %.5f" % losses[k].item())
        itv = torch.tensor([2.5, 3.5])
        out[tag + "_interval"] = np32(itv)
        for rng in (False, True):
            reg_in = {}
            for k in inputs:
                d = (torch.softmax(inputs[k]["prob_volume_pre"].detach(), 1) * inputs[k]["depth_values"]).sum(1)
                d = d + 2.0 * torch.randn(d.shape, generator=torch.Generator().manual_seed(5))      # both branches of the smooth L1
                reg_in[k] = dict(depth=d.clone().requires_grad_(True), depth_values=inputs[k]["depth_values"])
            losses = reg_loss_stage4(reg_in, gts, masks, w, itv, mask_out_range=rng, inverse_depth=inverse)
            sum(losses.values()).backward()
            for k in inputs:
                out["%s_%s_reg_depth" % (tag, k)] = np32(reg_in[k]["depth"])
                out["%s_%s_reg%d_loss" % (tag, k, int(rng))] = np.float64(losses[k].item())
                out["%s_%s_reg%d_grad" % (tag, k, int(rng))] = np32(reg_in[k]["depth"].grad)
                print(tag, k, "reg range=%d %.5f" % (rng, losses[k].item()))
    out["dlossw"] = np.asarray(w, np.float32)
    save("other_losses.npz", **out)


if __name__ == "__main__" and os.environ.get("GEN_LOSS2", "1") == "1":
    gen_other_losses()


# ---------------------------------------------------------------------------------------------
def gen_fpn_decoder():
    """FPNDecoder (models/module.py:242-270), eval mode, default-initialized weights under a seed + randomized BatchNorm.
    Coarsest level 5x6 -> 10x12, 20x24, 40x48: every level has partial 4x32 tiles; N=1 keeps the file small."""
    from models.module import FPNDecoder
    from oracle import ref_fpn
    torch.manual_seed(7)
    dec = FPNDecoder([8, 16, 32, 64])
    ref_fpn.randomize_bn(dec, 8)
    dec.eval()
    conv01, conv11, conv21, conv31 = ref_fpn.make_case(9, 1, 5, 6)
    with torch.no_grad():
        outs = dec(conv01, conv11, conv21, conv31)
    arrs = {"sd." + k: np32(v) for k, v in dec.state_dict().items() if v.dtype.is_floating_point}
    arrs.update(conv01=np32(conv01), conv11=np32(conv11), conv21=np32(conv21), conv31=np32(conv31))
    arrs.update({"out%d" % i: np32(o) for i, o in enumerate(outs)})
    save("fpn_decoder.npz", **arrs)


if __name__ == "__main__" and os.environ.get("GEN_FPN", "1") == "1":
    gen_fpn_decoder()


def gen_fpn_decoder_v2():
    """FPNDecoderV2 (models/module.py:273-302, TwinMVSNet's decoder), eval mode: seeded default init with the convolution weights rounded
    to float16-exact values (both sides use exactly these) + randomized BatchNorm; encoder maps as gen_fpn_decoder (coarsest 5x6: partial
    tiles at every level), transformer maps vit1..3 at 1/8, 1/4, 1/2 with 64 / 32 / 16 channels."""
    from models.module import FPNDecoderV2
    from oracle import ref_fpn
    torch.manual_seed(17)
    dec = FPNDecoderV2([8, 16, 32, 64])
    ref_fpn.randomize_bn(dec, 18)
    with torch.no_grad():
        for name, p in dec.named_parameters():
            if name.endswith("0.weight"):
                p.copy_(f16exact(p))
    dec.eval()
    conv01, conv11, conv21, conv31 = ref_fpn.make_case(19, 1, 5, 6)
    g = torch.Generator().manual_seed(20)
    vit1, vit2, vit3 = (torch.randn(1, c, 5 * s, 6 * s, generator=g) for c, s in ((64, 1), (32, 2), (16, 4)))
    with torch.no_grad():
        outs = dec(conv01, conv11, conv21, conv31, vit1, vit2, vit3)
    arrs = {}
    for k, v in dec.state_dict().items():
        if v.dtype.is_floating_point:
            arrs["sd." + k] = np32(v).astype(np.float16) if k.endswith("0.weight") else np32(v)
    arrs.update(conv01=np32(conv01), conv11=np32(conv11), conv21=np32(conv21), conv31=np32(conv31), vit1=np32(vit1), vit2=np32(vit2),
                vit3=np32(vit3))
    arrs.update({"out%d" % (i + 1): np32(o) for i, o in enumerate(outs)})
    save("fpn_decoder_v2.npz", **arrs)


if __name__ == "__main__" and os.environ.get("GEN_FPN_V2", "1") == "1":
    gen_fpn_decoder_v2()


def gen_fpn_encoder():
    """FPNEncoder (models/module.py:208-240, norm_type='BN'), eval mode, seeded default init with the convolution weights rounded to
    float16-exact values (halves the file; both sides use exactly these values) + randomized BatchNorm.  40x48 image."""
    from models.module import FPNEncoder
    from oracle import ref_fpn
    torch.manual_seed(11)
    enc = FPNEncoder([8, 16, 32, 64])
    ref_fpn.randomize_bn(enc, 12)
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if name.endswith("conv.weight"):
                p.copy_(f16exact(p))
    enc.eval()
    x = torch.randn(1, 3, 40, 48, generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        outs = enc(x)
    arrs = {}
    for k, v in enc.state_dict().items():
        if v.dtype.is_floating_point:
            arrs["sd." + k] = np32(v).astype(np.float16) if k.endswith("conv.weight") else np32(v)
    arrs["x"] = np32(x)
    arrs.update({"out%d" % i: np32(o) for i, o in enumerate(outs)})
    save("fpn_encoder.npz", **arrs)


if __name__ == "__main__" and os.environ.get("GEN_FPN", "1") == "1":
    gen_fpn_encoder()


def gen_vit():
    """The DINO ViT-small branch of the shipped MVSFormer-P config (configs/config_mvsformer-p.json: vit_small, patch 16, qk_scale
    'default', rescale 0.5, att_fusion, out_ch 64, nhead 6) run with the reference's OWN classes: ``models.vision_transformer`` imports
    nothing but torch and ``utils.trunc_normal_`` (the stubs above cover ``utils``' own imports; ``timm`` is needed by Twins only,
    models/gvt.py:6-7).  One 256x320 image -> bicubic 128x160 -> 8x10 patches.  Weights: oracle/weights.make_vit_state_dict (seeds 21 /
    22), loaded with strict=True; the file holds the image (float16-exact), the resized image, the normalized tokens, the CLS attention
    row of the last block and the decoder output that is added to conv31."""
    import models.vision_transformer as vits
    from models.module import VITDecoderStage4Single
    from oracle.weights import make_vit_state_dict
    vit_args = dict(rescale=0.5, patch_size=16, qk_scale="default", vit_arch="vit_small", vit_ch=384, out_ch=64, att_fusion=True, nhead=6)
    vit = vits.__dict__["vit_small"](patch_size=16, qk_scale="default")
    dec = VITDecoderStage4Single(vit_args)
    shapes = {"vit_small": {k: list(v.shape) for k, v in vit.state_dict().items()},
              "vit_decoder": {k: list(v.shape) for k, v in dec.state_dict().items()}}
    with open(os.path.join(OUT, "vit_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=0)
    vit.load_state_dict(make_vit_state_dict(shapes["vit_small"], 21), strict=True)
    dec.load_state_dict(make_vit_state_dict(shapes["vit_decoder"], 22), strict=True)
    vit.eval(), dec.eval()
    B, H, W = 1, 256, 320
    img = f16exact(torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(23)))
    with torch.no_grad():                                   # mvsformer_model.py:243-262, one view
        vit_h, vit_w = int(H * vit_args["rescale"]), int(W * vit_args["rescale"])
        vit_imgs = F.interpolate(img, (vit_h, vit_w), mode="bicubic", align_corners=False)
        vit_feat, vit_att = vit.forward_with_last_att(vit_imgs)
        feat = vit_feat[:, 1:].reshape(B, vit_h // 16, vit_w // 16, 384).permute(0, 3, 1, 2).contiguous()
        att = vit_att[:, :, 0, 1:].reshape(B, -1, vit_h // 16, vit_w // 16)
        vit_out = dec.forward(feat, att)
    save("vit_small.npz", img=np32(img).astype(np.float16), vit_imgs=np32(vit_imgs), vit_feat=np32(vit_feat),
         att_cls=np32(vit_att[:, :, 0, 1:]), vit_out=np32(vit_out), seeds=np.array([21, 22, 23]))
    print("vit_small params %.1f M, decoder %.1f M" % (sum(p.numel() for p in vit.parameters()) / 1e6, sum(p.numel() for p in dec.parameters()) / 1e6))


if __name__ == "__main__" and os.environ.get("GEN_VIT", "1") == "1":
    gen_vit()


def gen_fpn_train():
    """FPNEncoder + FPNDecoder in TRAINING mode (batch-statistics BatchNorm, models/module.py:208-270 under train(); what TwinMVSNet /
    DINOMVSNet do with their FPN): two 32x40 images through the real modules, loss = sum_i <out_i, R_i> with seeded R, every gradient.
    Weights: oracle/weights.make_state_dict over the modules' own key/shape lists (tests/golden/fpn_shapes.json), seeds 31 / 32."""
    from models.module import FPNDecoder, FPNEncoder
    enc, dec = FPNEncoder([8, 16, 32, 64], norm_type="BN"), FPNDecoder([8, 16, 32, 64])
    shapes = {"encoder": {k: list(v.shape) for k, v in enc.state_dict().items()}, "decoder": {k: list(v.shape) for k, v in dec.state_dict().items()}}
    with open(os.path.join(OUT, "fpn_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=0)
    enc.load_state_dict(make_state_dict(shapes["encoder"], 31), strict=True)
    dec.load_state_dict(make_state_dict(shapes["decoder"], 32), strict=True)
    enc.train(), dec.train()
    g = torch.Generator().manual_seed(33)
    x = f16exact(torch.randn(2, 3, 32, 40, generator=g)).requires_grad_(True)
    feats = enc(x)
    outs = dec(*feats)
    R = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * r).sum() for o, r in zip(outs, R))
    loss.backward()
    arrs = {"x": np32(x).astype(np.float16), "dx": np32(x.grad), "loss": np32(loss)}
    arrs.update({"out%d" % i: np32(o) for i, o in enumerate(outs)})
    arrs.update({"feat%d" % i: np32(o) for i, o in enumerate(feats)})
    for tag, m in (("enc", enc), ("dec", dec)):
        arrs.update({"%s.grad.%s" % (tag, k): np32(p.grad) for k, p in m.named_parameters()})
        arrs.update({"%s.buf.%s" % (tag, k): np32(b) for k, b in m.named_buffers() if b.dtype.is_floating_point})
    arrs["seeds"] = np.array([31, 32, 33])
    save("fpn_train.npz", **arrs)


if __name__ == "__main__" and os.environ.get("GEN_FPN_TRAIN", "1") == "1":
    gen_fpn_train()


def gen_end_to_end():
    """The whole MVSFormer-P model, images -> depth map: the reference's own ``DINOMVSNet`` (models/mvsformer_model.py:163-308) built from the
    shipped ``configs/config_mvsformer-p.json`` arguments, ``eval()``, one reference + two source views of 128 x 192 (the smallest size every
    stage accepts: H/8 x W/8 = 16 x 24 halves three times in CostRegNet, the ViT sees 64 x 96 = 4 x 6 patches), photo-consistent images
    rendered by mvsformer_amd.synth, tmp = [5, 5, 5, 1].  Weights from a seed (oracle/weights.make_model_state_dict, strict load)."""
    from oracle.weights import make_model_state_dict
    args = json.load(open(os.path.join(REF, "configs", "config_mvsformer-p.json")))["arch"]["args"]
    cwd = os.getcwd()
    os.chdir("/tmp")                                        # the constructor probes ./pretrained_weights (absent: it only prints a notice)
    net = ref_mm.DINOMVSNet(args)
    os.chdir(cwd)
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(OUT, "dinomvsnet_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=0)
    seed = 31
    net.load_state_dict(make_model_state_dict(shapes, seed), strict=True)
    net.eval()
    V, H, W = 3, 128, 192
    _, proj, dv, scene = synth.make_inputs(V, H, W, seed=32)
    imgs = f16exact(synth.render_features(scene, 1, 3, noise=0.02))          # [1, V, 3, H, W]: the plane-induced warp of one texture
    tmp = [5.0, 5.0, 5.0, 1.0]
    feats = {}
    real_forward = net.decoder.forward                      # (the model calls decoder.forward(...) directly: a forward hook would not fire)

    def recording(*a):
        o = real_forward(*a)
        feats.setdefault("calls", []).append([t.detach().clone() for t in o])
        return o
    net.decoder.forward = recording
    with torch.no_grad():
        out = net(imgs, proj, dv, tmp=tmp)
    arrs = dict(imgs=np32(imgs).astype(np.float16), depth_range=np32(dv), seed=np.int64(seed), scene_seed=np.int64(32), tmps=np.array(tmp, dtype=np.float32),
                refined_depth=np32(out["refined_depth"]), photometric_confidence=np32(out["photometric_confidence"]),
                features_stage1=np32(torch.stack([c[0] for c in feats["calls"]], dim=1)))      # [1, V, 64, 16, 24]: FPN + ViT, before the path
    for k, v in proj.items():
        arrs["proj_" + k] = np32(v)
    for i in range(4):
        arrs["s%d_depth" % (i + 1)] = np32(out["stage%d" % (i + 1)]["depth"])
    save("dinomvsnet_e2e.npz", **arrs)
    print("DINOMVSNet params %.1f M, depth range %.1f .. %.1f" % (sum(p.numel() for p in net.parameters()) / 1e6, out["refined_depth"].min(), out["refined_depth"].max()))


if __name__ == "__main__" and os.environ.get("GEN_E2E", "1") == "1":
    gen_end_to_end()


GRAD_SAMPLE = 4096


def grad_sample(t, seed):
    """A fixed random subset of a (large) gradient + its L2 norm: what the goldens of the 3.4 M-parameter ViT decoder store per tensor."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= GRAD_SAMPLE:
        return np32(flat), np.arange(flat.numel(), dtype=np.int64), np.float32(flat.double().norm())
    idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(seed))[:GRAD_SAMPLE].sort().values
    return np32(flat[idx]), idx.numpy().astype(np.int64), np.float32(flat.double().norm())


def gen_vit_decoder_train():
    """``VITDecoderStage4Single`` (+ ``AttentionFusionSimple``) of the shipped MVSFormer-P config in TRAINING mode (models/module.py:353-368,
    450-466 under train(); DINOMVSNet trains it, mvsformer_model.py:196-201,225-228): the real module on two 8 x 10 token maps, batch-statistics
    BatchNorm, loss = <out, R> with seeded R, every parameter gradient (the big ones as a fixed 4096-element sample + their L2 norm), the
    gradient of the inputs and the updated running statistics.  Weights: oracle/weights.make_vit_state_dict, seed 41."""
    from models.module import VITDecoderStage4Single
    from oracle.weights import load_vit_shapes, make_vit_state_dict
    vit_args = dict(rescale=0.5, patch_size=16, qk_scale="default", vit_arch="vit_small", vit_ch=384, out_ch=64, att_fusion=True, nhead=6)
    dec = VITDecoderStage4Single(vit_args)
    dec.load_state_dict(make_vit_state_dict(load_vit_shapes("vit_decoder"), 41), strict=True)
    dec.train()
    g = torch.Generator().manual_seed(42)
    feat = f16exact(torch.randn(2, 384, 8, 10, generator=g)).requires_grad_(True)
    att = f16exact(torch.rand(2, 6, 8, 10, generator=g) * 0.05).requires_grad_(True)
    out = dec(feat, att)
    R = torch.randn(out.shape, generator=g)
    loss = (out * R).sum()
    loss.backward()
    arrs = dict(feat=np32(feat).astype(np.float16), att=np32(att).astype(np.float16), out=np32(out), loss=np32(loss), seeds=np.array([41, 42]))
    arrs["dfeat"], arrs["datt"] = np32(feat.grad), np32(att.grad)
    for n, (k, p) in enumerate(dec.named_parameters()):
        v, idx, nrm = grad_sample(p.grad, 100 + n)
        arrs["grad." + k], arrs["idx." + k], arrs["norm." + k] = v, idx, nrm
    arrs.update({"buf." + k: np32(b) for k, b in dec.named_buffers() if b.dtype.is_floating_point})
    save("vit_decoder_train.npz", **arrs)


if __name__ == "__main__" and os.environ.get("GEN_VIT_TRAIN", "1") == "1":
    gen_vit_decoder_train()


def gen_train_step():
    """One TRAINING step of the whole MVSFormer-P model as trainer/mvsformer_trainer.py:104-135 runs it (without AMP: fp32), from the real classes:
    ``DINOMVSNet(config_mvsformer-p)`` in train() ("fix": true - the ViT runs under no_grad), 1 sample x 3 views of 256 x 320,
    ``ce_loss_stage4`` (models/losses.py:304-350) against the scene's true plane depth, ``backward()``.  Stored: the four stage losses, per
    trainable tensor a 512-element gradient sample + index + L2 norm, the stage depths.  Weights: make_model_state_dict(seed 51)."""
    from models.losses import ce_loss_stage4
    from oracle.weights import load_model_shapes, make_model_state_dict
    args = json.load(open(os.path.join(REF, "configs", "config_mvsformer-p.json")))["arch"]["args"]
    cwd = os.getcwd()
    os.chdir("/tmp")
    net = ref_mm.DINOMVSNet(args)
    os.chdir(cwd)
    seed = 51
    net.load_state_dict(make_model_state_dict(load_model_shapes(), seed), strict=True)
    net.train()
    V, H, W = 3, 256, 320
    _, proj, dv, scene = synth.make_inputs(V, H, W, seed=52)
    imgs = f16exact(synth.render_features(scene, 1, 3, noise=0.02))
    tmp = [5.0, 5.0, 5.0, 1.0]
    gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s).to(torch.float32).unsqueeze(0) for i, s in enumerate((8, 4, 2, 1))}
    masks = {k: torch.ones_like(v) for k, v in gts.items()}
    dlossw = [1.0, 1.0, 1.0, 1.0]
    out = net(imgs, proj, dv, tmp=tmp)
    losses = ce_loss_stage4(out, gts, masks, dlossw, focal=False, gamma=0.0, inverse_depth=True)
    sum(losses.values()).backward()
    arrs = dict(imgs=np32(imgs).astype(np.float16), depth_range=np32(dv), seed=np.int64(seed), scene_seed=np.int64(52), tmps=np.array(tmp, dtype=np.float32),
                dlossw=np.array(dlossw, dtype=np.float32), losses=np.array([losses["stage%d" % i].item() for i in range(1, 5)], dtype=np.float64))
    for k, v in proj.items():
        arrs["proj_" + k] = np32(v)
    for i in range(1, 5):                                   # (the ground truth is synth.plane_depth(make_scene(V, H, W, scene_seed), scale): not stored)
        arrs["s%d_depth" % i] = np32(out["stage%d" % i]["depth"])
    n = 0
    global GRAD_SAMPLE
    keep, GRAD_SAMPLE = GRAD_SAMPLE, 512
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        v, idx, nrm = grad_sample(p.grad, 1000 + n)
        arrs["grad." + k], arrs["idx." + k], arrs["norm." + k] = v, idx.astype(np.int32), nrm
        n += 1
    GRAD_SAMPLE = keep
    save("train_step_mvsformer_p.npz", **arrs)
    print("train step: losses", arrs["losses"], "tensors with gradients:", n)


if __name__ == "__main__" and os.environ.get("GEN_TRAIN_STEP", "1") == "1":
    gen_train_step()


def gen_fpn_decoder_v2_train():
    """``FPNDecoderV2`` (TwinMVSNet's decoder, models/module.py:273-302) in TRAINING mode: the real module on two 32 x 40 pyramids, loss =
    sum_i <out_i, R_i>, every gradient (parameters: 4096-element samples + norms; all seven inputs in full), updated running statistics."""
    from models.module import FPNDecoderV2
    dec = FPNDecoderV2([8, 16, 32, 64])
    shapes = {k: list(v.shape) for k, v in dec.state_dict().items()}
    with open(os.path.join(OUT, "fpn_v2_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=0)
    dec.load_state_dict(make_state_dict(shapes, 61), strict=True)
    dec.train()
    g = torch.Generator().manual_seed(62)
    dims = dict(conv01=(8, 32, 40), conv11=(16, 16, 20), conv21=(32, 8, 10), conv31=(64, 4, 5), vit1=(64, 4, 5), vit2=(32, 8, 10), vit3=(16, 16, 20))
    ins = {k: f16exact(torch.randn(2, *d, generator=g)).requires_grad_(True) for k, d in dims.items()}
    outs = dec(*[ins[k] for k in ("conv01", "conv11", "conv21", "conv31", "vit1", "vit2", "vit3")])
    R = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * r).sum() for o, r in zip(outs, R))
    loss.backward()
    arrs = {"loss": np32(loss), "seeds": np.array([61, 62])}
    for k, v in ins.items():
        arrs["in." + k], arrs["din." + k] = np32(v).astype(np.float16), np32(v.grad)
    arrs.update({"out%d" % i: np32(o) for i, o in enumerate(outs)})
    for n, (k, p) in enumerate(dec.named_parameters()):
        v, idx, nrm = grad_sample(p.grad, 200 + n)
        arrs["grad." + k], arrs["idx." + k], arrs["norm." + k] = v, idx.astype(np.int32), nrm
    arrs.update({"buf." + k: np32(b) for k, b in dec.named_buffers() if b.dtype.is_floating_point})
    save("fpn_decoder_v2_train.npz", **arrs)


if __name__ == "__main__" and os.environ.get("GEN_FPN_V2_TRAIN", "1") == "1":
    gen_fpn_decoder_v2_train()


def gen_was_loss():
    """was_loss.npz: models/losses.py wasserstein_loss (ot_iter 10, ot_eps 1, ot_continous False: the trainer's arguments) on the seeded cases of
    ce_loss.npz (oracle/ref_losses.make_loss_case seed 3; the tests read depth_values / logits / gt / mask from that file), prob_volume =
    softmax(logits): the four weighted stage losses and d loss / d prob_volume."""
    from models.losses import wasserstein_loss
    from oracle import ref_losses
    out = {}
    w = [1.0, 0.5, 2.0, 1.0]
    inputs, gts, masks = ref_losses.make_loss_case(seed=3, inverse_depth=True)
    for k in inputs:
        inputs[k]["prob_volume"] = torch.softmax(inputs[k]["prob_volume_pre"], 1).detach().requires_grad_(True)
    losses = wasserstein_loss(inputs, gts, masks, w, ot_iter=10, ot_eps=1, ot_continous=False, inverse=True)
    sum(losses.values()).backward()
    for k in inputs:
        out[k + "_loss"] = np.float64(losses[k].item())
        out[k + "_grad"] = np32(inputs[k]["prob_volume"].grad)
        print(k, "wasserstein %.6f" % losses[k].item())
    out["dlossw"] = np.asarray(w, np.float32)
    save("was_loss.npz", **out)


if __name__ == "__main__" and os.environ.get("GEN_WAS", "1") == "1":
    gen_was_loss()
