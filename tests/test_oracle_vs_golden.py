"""Pins the oracle (oracle/ref_torch.py, oracle/ref_numpy.py) to the golden vectors that
oracle/gen_golden.py produced by running the real reference (/root/reference) in the build container."""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs, rel_err, t
from oracle import ref_numpy, ref_torch
from oracle.weights import load_shapes, make_state_dict

TOL = 2e-5   # oracle and reference run the same ATen CPU kernels; only op grouping differs


def test_warp_kat_literals():
    """SURVEY.md §8(a1) known-answer case, checked against hand-derived rows, the golden file and both oracles."""
    g = load_golden("warp_kat.npz")
    w = g["warped"]
    np.testing.assert_allclose(w[0, 0, 0, 0], [1, 2, 3, 4, 5, 0], atol=1e-4)          # d=100: +1 px
    np.testing.assert_allclose(w[0, 0, 1, 0], [0.5, 1.5, 2.5, 3.5, 4.5, 2.5], atol=1e-4)  # d=200: +0.5 px
    np.testing.assert_allclose(w[0, 0, 2, 0, :3], [2, 3, 4], atol=1e-4)               # d=50: +2 px
    assert np.all(w[0, :, 3] == 0) and g["mask"][0, 3].all()                           # d<0: behind camera
    assert g["mask"][0, 2, 0].tolist() == [False, False, False, True, True, True]
    assert max_abs(g["ident"][0, :, 0], g["src"][0]) < 1e-5
    wt, mt = ref_torch.homo_warping_3D_with_mask(t(g["src"]), t(g["src_proj"]), t(g["ref_proj"]), t(g["depth"]))
    assert max_abs(wt, w) < 1e-6 and np.array_equal(mt.numpy(), g["mask"])
    wn, mn = ref_numpy.plane_sweep_warp(g["src"], g["src_proj"], g["ref_proj"], g["depth"])
    # pixel 3 of the d=50 row lands on x=W-1 to rounding: sampled but value is rounding sensitive
    assert np.abs(wn - w).max() < 1e-4
    assert (mn != g["mask"]).sum() <= 4


@pytest.mark.parametrize("which", ["bd", "map"])
def test_warp_general(which):
    g = load_golden("warp_general.npz")
    depth = g["depth_" + which]
    wt, mt = ref_torch.homo_warping_3D_with_mask(t(g["src"]), t(g["src_proj"]), t(g["ref_proj"]), t(depth))
    assert max_abs(wt, g["warped_" + which]) < TOL
    assert np.array_equal(mt.numpy(), g["mask_" + which])
    wn, mn = ref_numpy.plane_sweep_warp(g["src"], g["src_proj"], g["ref_proj"], depth)
    # numpy inverse/matmul round differently from ATen -> compare away from tap-switch discontinuities
    err = np.abs(wn - g["warped_" + which])
    assert np.quantile(err, 0.999) < 1e-3 and (mn != g["mask_" + which]).mean() < 2e-3
    assert g["mask_" + which].any() and not g["mask_" + which].all()


def test_heads_and_schedulers():
    g = load_golden("heads.npz")
    p = t(g["p"])
    assert max_abs(ref_torch.depth_regression(p, t(g["dv_map"])), g["reg_map"]) < 1e-3
    assert max_abs(ref_torch.depth_regression(p, t(g["dv_bd"])), g["reg_bd"]) < 1e-3
    for n in (2, 3, 4):
        assert max_abs(ref_torch.conf_regression(p, n), g["conf_n%d" % n]) < 1e-6
    init = ref_torch.init_inverse_range(t(g["cur_depth"]), 32, 8, 12)
    assert rel_err(init, g["init_inv"]) < 1e-6
    sched = ref_torch.schedule_inverse_range(t(g["prev_depth"]), t(g["init_inv"]), 16, 2.67, 16, 24)
    assert rel_err(sched, g["sched_inv"]) < 1e-6


@pytest.mark.parametrize("kind", ["costregnet", "costregnet3d"])
def test_regularizers(kind):
    g = load_golden("costreg.npz")
    sd = make_state_dict(load_shapes("stage_" + kind), int(g[kind + "_seed"]))
    fn = ref_torch.cost_reg_net if kind == "costregnet" else ref_torch.cost_reg_net_3d
    with torch.no_grad():
        y = fn(t(g[kind + "_x"]), sd, "cost_reg")
    assert max_abs(y, g[kind + "_y"]) < 1e-4 * max(1.0, float(np.abs(g[kind + "_y"]).max()))


@pytest.mark.parametrize("kind", ["costregnet", "costregnet3d"])
def test_stage(kind):
    g = load_golden("stage_%s.npz" % kind)
    sd = make_state_dict(load_shapes("stage_" + kind), int(g["weight_seed"]))
    nd = int(g["ndepth"])
    taps = {}
    with torch.no_grad():
        o = ref_torch.stage_forward(t(g["features"]), t(g["proj"]), t(g["depth_values"]), sd, ndepth=nd, tmp=5.0,
                                    training=False, taps=taps)
    assert max_abs(taps["in_prod"][0], g["tap_in_prod_v1"]) < TOL
    assert max_abs(torch.cat(taps["entropy"], 1), g["tap_entropy"]) < TOL
    assert max_abs(torch.cat(taps["vis_weight"], 1), g["tap_vis_weight"]) < TOL
    assert max_abs(taps["volume_mean"], g["tap_volume_mean"]) < TOL
    assert max_abs(taps["similarity_sum"], g["tap_similarity_sum"]) < TOL
    assert max_abs(o["prob_volume_pre"], g["eval_prob_volume_pre"]) < 1e-4
    assert rel_err(o["depth"], g["eval_depth"]) < 1e-5
    assert max_abs(o["photometric_confidence"], g["eval_photometric_confidence"]) < 1e-5
    assert (o["sim_depth"].numpy() != g["eval_sim_depth"]).mean() < 0.01
    with torch.no_grad():
        o = ref_torch.stage_forward(t(g["features"]), t(g["proj"]), t(g["depth_values"]), sd, ndepth=nd, tmp=5.0,
                                    training=True)
    assert max_abs(o["prob_volume_pre"], g["train_prob_volume_pre"]) < 1e-4
    assert (o["depth"].numpy() != g["train_depth"]).mean() < 0.01
    assert "sim_depth" not in o


OTHER_HEAD_CASES = [("costregnet", 16), ("costregnet3d", 4), ("costregnet", 32), ("costregnet3d", 8)]


def other_head_inputs(kind, nd):
    """Inputs of one heads_other.npz case: the stage case's features / cameras, hypotheses resampled for the 32- / 8-plane cases."""
    g, go = load_golden("stage_%s.npz" % kind), load_golden("heads_other.npz")
    key = "hyp_%s_%d" % (kind, nd)
    hyp = go[key] if key in go else g["depth_values"]
    return g, go, hyp


@pytest.mark.parametrize("kind,nd", OTHER_HEAD_CASES)
@pytest.mark.parametrize("depth_type", ["mixup_ce", "reg"])
def test_stage_other_heads(kind, nd, depth_type):
    """depth_type 'mixup_ce' / regression heads (mvsformer_model.py:126-146) against the real reference StageNet, eval and train."""
    g, go, hyp = other_head_inputs(kind, nd)
    sd = make_state_dict(load_shapes("stage_" + kind), int(g["weight_seed"]))
    for mode in ("eval", "train"):
        with torch.no_grad():
            o = ref_torch.stage_forward(t(g["features"]), t(g["proj"]), t(hyp), sd, ndepth=nd, tmp=5.0, training=mode == "train",
                                        depth_type=depth_type)
        pre = "%s_%d_%s_%s_" % (kind, nd, depth_type, mode)
        # the pair arg-max (mixup) and floor(E[d]) (conf_regression window) flip on rounding-level ties at isolated pixels
        bad_d = (np.abs(o["depth"].numpy() - go[pre + "depth"]) > 1e-4 * np.abs(go[pre + "depth"])).mean()
        bad_c = (np.abs(o["photometric_confidence"].numpy() - go[pre + "photometric_confidence"]) > 1e-4).mean()
        assert bad_d < 2e-3 and bad_c < 2e-3, (mode, bad_d, bad_c)


@pytest.mark.parametrize("V", [3, 5])
def test_cascade(V):
    g = load_golden("cascade_v%d.npz" % V)
    nds = [int(x) for x in g["ndepths"]]
    sds = [make_state_dict(load_shapes("stage_costregnet3d" if nd <= 8 else "stage_costregnet"), int(s))
           for nd, s in zip(nds, g["weight_seeds"])]
    feats = {"stage%d" % i: t(g["features_stage%d" % i]) for i in range(1, 5)}
    proj = {"stage%d" % i: t(g["proj_stage%d" % i]) for i in range(1, 5)}
    with torch.no_grad():
        out = ref_torch.cascade_forward(feats, proj, t(g["depth_range"]), sds, ndepths=nds,
                                        depth_interals_ratio=[float(x) for x in g["ratios"]],
                                        tmp=[float(x) for x in g["tmps"]])
    for i in range(1, 5):
        assert rel_err(out["stage%d" % i]["depth"], g["s%d_depth" % i]) < 1e-5
        assert rel_err(out["stage%d" % i]["depth_values"], g["s%d_depth_values" % i]) < 1e-5
    assert rel_err(out["refined_depth"], g["refined_depth"]) < 1e-5
    assert max_abs(out["photometric_confidence"], g["photometric_confidence"]) < 1e-5


@pytest.mark.parametrize("kind", ["costregnet", "costregnet3d"])
def test_stage_train_gradients(kind):
    """The oracle in train mode (batch-statistics BN) under torch autograd reproduces the reference's gradients."""
    g = load_golden("train_%s.npz" % kind)
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
          for k, v in make_state_dict(load_shapes("stage_" + kind), int(g["weight_seed"])).items()}
    feats = t(g["features"]).requires_grad_(True)
    o = ref_torch.stage_forward(feats, t(g["proj"]), t(g["depth_values"]), sd, ndepth=int(g["ndepth"]), tmp=5.0, training=True)
    (o["prob_volume_pre"] * t(g["R"])).sum().backward()
    assert max_abs(o["prob_volume_pre"].detach(), g["prob_volume_pre"]) < 1e-4
    scale = float(np.abs(g["grad_features"]).max())
    assert max_abs(feats.grad, g["grad_features"]) < 1e-4 * scale
    for k, v in sd.items():
        if "grad_" + k in g:
            assert max_abs(v.grad, g["grad_" + k]) < 2e-4 * max(1e-6, float(np.abs(g["grad_" + k]).max())), k
        if "buf_" + k in g:
            assert max_abs(v, g["buf_" + k]) < 1e-5, k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fusion_oracle_vs_reference(tag):
    """oracle/ref_fusion.py against misc/fusion.py outputs (tests/golden/fusion.npz)."""
    from oracle import ref_fusion
    g = load_golden("fusion.npz")
    case = {k: t(g["%s_%s" % (tag, k)]) for k in ("ref_depth", "src_depths", "ref_cam", "src_cams")}
    thr = [float(x) for x in g[tag + "_thresholds"]]
    out = ref_fusion.filter_depth_maps(case["ref_depth"], case["src_depths"], case["ref_cam"], case["src_cams"], *thr)
    assert max_abs(out["reproj_xyd"], g[tag + "_reproj_xyd"]) < 1e-3
    assert np.array_equal(out["in_range"].numpy(), g[tag + "_in_range"])
    assert (out["masks"].numpy() != g[tag + "_masks"]).mean() < 1e-4
    assert (out["mask"].numpy() != g[tag + "_mask"]).mean() < 1e-4
    assert rel_err(out["ref_depth_ave"], g[tag + "_ref_depth_ave"]) < 1e-5
    assert rel_err(out["points"], g[tag + "_points"]) < 1e-5
    pm = ref_fusion.prob_filter(t(g[tag + "_conf"]), [0.3, 0.5, 0.2])
    assert np.array_equal(pm.numpy(), g[tag + "_prob_mask"])
    # the synthetic generator itself is deterministic: regenerating gives the stored inputs
    kw = dict(a=dict(n=1, v=4, h=48, w=64, seed=0), b=dict(n=2, v=3, h=40, w=56, seed=1, noise=0.008))[tag]
    again = ref_fusion.make_fusion_case(**kw)
    assert max_abs(again["src_depths"], g[tag + "_src_depths"]) < 1e-2


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fusion_dynamic_oracle_vs_reference(tag):
    """oracle get_reproj_dynamic / vis_filter_dynamic / the test.py:494-514 reduction against the reference's outputs."""
    from oracle import ref_fusion
    g = load_golden("fusion.npz")
    case = {k: t(g["%s_%s" % (tag, k)]) for k in ("ref_depth", "src_depths", "ref_cam", "src_cams")}
    bases = [float(x) for x in g[tag + "_dyn_bases"]]
    out = ref_fusion.dynamic_filter_depth_maps(case["ref_depth"], case["src_depths"], case["ref_cam"], case["src_cams"], *bases)
    want = torch.as_tensor(g[tag + "_dyn_reproj_xyd"])
    assert ((out["reproj_xyd"] - want).abs() <= 1e-3 + 1e-5 * want.abs()).all()
    assert (out["masks"].numpy() != g[tag + "_dyn_masks"]).mean() < 1e-4
    assert (out["vis_mask"].numpy() != g[tag + "_dyn_vis_mask"]).mean() < 1e-4
    # n > 1: the reference broadcasts [n,1,h,w] | [n,h,w] -> [n,n,h,w] with entry [i,j] = sample j's mask
    assert (out["geo_mask"][:, 0].numpy() != g[tag + "_dyn_geo_mask"][0]).mean() < 1e-4
    assert rel_err(out["ref_depth_ave"], g[tag + "_dyn_ref_depth_ave"]) < 1e-5
    assert rel_err(out["points"], g[tag + "_dyn_points"]) < 1e-5


@pytest.mark.parametrize("tag,inverse", [("inv", True), ("fwd", False)])
def test_ce_loss_oracle_vs_reference(tag, inverse):
    """oracle/ref_losses.py against models/losses.py ce_loss_stage4 values and gradients (tests/golden/ce_loss.npz)."""
    from oracle import ref_losses
    g = load_golden("ce_loss.npz")
    w = [float(x) for x in g[tag + "_dlossw"]]
    inputs, gts, masks = {}, {}, {}
    for k in ("stage1", "stage2", "stage3", "stage4"):
        inputs[k] = dict(depth_values=t(g["%s_%s_depth_values" % (tag, k)]), prob_volume_pre=t(g["%s_%s_logits" % (tag, k)]).requires_grad_(True))
        gts[k], masks[k] = t(g["%s_%s_gt" % (tag, k)]), t(g["%s_%s_mask" % (tag, k)])
    losses = ref_losses.ce_loss_stage4(inputs, gts, masks, w, inverse_depth=inverse)
    sum(losses.values()).backward()
    for k in inputs:
        assert abs(losses[k].item() - float(g["%s_%s_loss" % (tag, k)])) < 1e-6
        assert max_abs(inputs[k]["prob_volume_pre"].grad, g["%s_%s_grad" % (tag, k)]) < 1e-8


def _loss_inputs(tag):
    g = load_golden("ce_loss.npz")
    inputs, gts, masks = {}, {}, {}
    for k in ("stage1", "stage2", "stage3", "stage4"):
        inputs[k] = dict(depth_values=t(g["%s_%s_depth_values" % (tag, k)]), prob_volume_pre=t(g["%s_%s_logits" % (tag, k)]))
        gts[k], masks[k] = t(g["%s_%s_gt" % (tag, k)]), t(g["%s_%s_mask" % (tag, k)])
    return inputs, gts, masks


@pytest.mark.parametrize("tag,inverse", [("inv", True), ("fwd", False)])
def test_mixup_ce_and_reg_loss_oracle_vs_reference(tag, inverse):
    """oracle/ref_losses.py against models/losses.py mixup_ce_loss_stage4 / reg_loss_stage4 (tests/golden/other_losses.npz; inputs of ce_loss.npz)."""
    from oracle import ref_losses
    g = load_golden("other_losses.npz")
    w = [float(x) for x in g["dlossw"]]
    inputs, gts, masks = _loss_inputs(tag)
    for k in inputs:
        inputs[k]["prob_volume_pre"].requires_grad_(True)
    losses = ref_losses.mixup_ce_loss_stage4(inputs, gts, masks, w, inverse_depth=inverse)
    sum(losses.values()).backward()
    for k in inputs:
        assert abs(losses[k].item() - float(g["%s_%s_mixup_loss" % (tag, k)])) < 2e-6
        assert max_abs(inputs[k]["prob_volume_pre"].grad, g["%s_%s_mixup_grad" % (tag, k)]) < 1e-8
    itv = t(g[tag + "_interval"])
    for rng in (0, 1):
        reg_in = {k: dict(depth=t(g["%s_%s_reg_depth" % (tag, k)]).requires_grad_(True), depth_values=inputs[k]["depth_values"]) for k in inputs}
        losses = ref_losses.reg_loss_stage4(reg_in, gts, masks, w, itv, mask_out_range=bool(rng), inverse_depth=inverse)
        sum(losses.values()).backward()
        for k in inputs:
            want = float(g["%s_%s_reg%d_loss" % (tag, k, rng)])
            assert abs(losses[k].item() - want) < 2e-6 * max(1.0, want)
            assert max_abs(reg_in[k]["depth"].grad, g["%s_%s_reg%d_grad" % (tag, k, rng)]) < 1e-8


def fpn_golden():
    g = load_golden("fpn_decoder.npz")
    sd = {k[3:]: t(v) for k, v in g.items() if k.startswith("sd.")}
    return g, sd, [t(g[k]) for k in ("conv01", "conv11", "conv21", "conv31")]


def test_fpn_decoder_oracle_vs_reference():
    """oracle/ref_fpn.py against the real FPNDecoder (models/module.py:242-270), eval BatchNorm."""
    from oracle import ref_fpn
    g, sd, feats = fpn_golden()
    outs = ref_fpn.fpn_decoder_forward(sd, *feats)
    assert [tuple(o.shape) for o in outs] == [(1, 64, 5, 6), (1, 32, 10, 12), (1, 16, 20, 24), (1, 8, 40, 48)]
    for i, o in enumerate(outs):
        assert max_abs(o, g["out%d" % i]) < TOL * max(1.0, float(np.abs(g["out%d" % i]).max())), i


def test_fpn_decoder_v2_oracle_vs_reference():
    """oracle/ref_fpn.py against the real FPNDecoderV2 (models/module.py:273-302; tests/golden/fpn_decoder_v2.npz), eval BatchNorm."""
    from oracle import ref_fpn
    g = load_golden("fpn_decoder_v2.npz")
    sd = {k[3:]: t(v.astype(np.float32)) for k, v in g.items() if k.startswith("sd.")}
    outs = ref_fpn.fpn_decoder_v2_forward(sd, *[t(g[k]) for k in ("conv01", "conv11", "conv21", "conv31", "vit1", "vit2", "vit3")])
    assert [tuple(o.shape) for o in outs] == [(1, 64, 5, 6), (1, 32, 10, 12), (1, 16, 20, 24), (1, 8, 40, 48)]
    for i, o in enumerate(outs, start=1):
        assert max_abs(o, g["out%d" % i]) < TOL * max(1.0, float(np.abs(g["out%d" % i]).max())), i


def fpn_encoder_golden():
    g = load_golden("fpn_encoder.npz")
    return g, {k[3:]: t(v) for k, v in g.items() if k.startswith("sd.")}


def test_fpn_encoder_oracle_vs_reference():
    """oracle/ref_fpn.py against the real FPNEncoder (models/module.py:208-240), eval BatchNorm."""
    from oracle import ref_fpn
    g, sd = fpn_encoder_golden()
    outs = ref_fpn.fpn_encoder_forward(sd, t(g["x"]))
    assert [tuple(o.shape) for o in outs] == [(1, 8, 40, 48), (1, 16, 20, 24), (1, 32, 10, 12), (1, 64, 5, 6)]
    for i, o in enumerate(outs):
        assert max_abs(o, g["out%d" % i]) < TOL * max(1.0, float(np.abs(g["out%d" % i]).max())), i


def test_vit_branch_oracle_vs_reference_golden():
    """oracle/ref_vit.py (plain-torch restatement of the DINO ViT-small branch of configs/config_mvsformer-p.json) against the outputs of
    the reference's own ``vits.vit_small`` + ``VITDecoderStage4Single`` (tests/golden/vit_small.npz, oracle/gen_golden.py::gen_vit)."""
    from oracle import ref_vit
    from oracle.weights import load_vit_shapes, make_vit_state_dict
    g = load_golden("vit_small.npz")
    sd_vit = make_vit_state_dict(load_vit_shapes("vit_small"), int(g["seeds"][0]))
    sd_dec = make_vit_state_dict(load_vit_shapes("vit_decoder"), int(g["seeds"][1]))
    assert sum(v.numel() for k, v in sd_vit.items()) > 21_000_000           # the real ViT-small, not a toy
    with torch.no_grad():
        out = ref_vit.vit_branch(sd_vit, sd_dec, torch.from_numpy(g["img"].astype(np.float32)))
    for k, tol in (("vit_imgs", 1e-6), ("vit_feat", 2e-5), ("att_cls", 2e-6), ("vit_out", 2e-5)):
        want = torch.from_numpy(g[k])
        err = (out[k] - want).abs().max().item() / max(1e-6, want.abs().max().item())
        assert err < tol, (k, err)
    # the attention the decoder consumes is a real distribution, not the uniform one of a default-initialised ViT
    att = torch.from_numpy(g["att_cls"])
    assert att.max() > 4.0 / att.shape[-1] and (att.sum(-1) < 1.0).all()


def test_dinomvsnet_oracle_vs_reference():
    """oracle/ref_model.py (FPN encoder -> ViT branch -> FPN decoder -> 4-stage cascade) against the REAL ``DINOMVSNet`` of the shipped
    MVSFormer-P config in eval mode (tests/golden/dinomvsnet_e2e.npz: images -> depth map, 3 views of 128 x 192)."""
    from oracle import ref_model
    from oracle.weights import load_model_shapes, make_model_state_dict
    g = load_golden("dinomvsnet_e2e.npz")
    sd = make_model_state_dict(load_model_shapes(), int(g["seed"]))
    imgs = torch.from_numpy(g["imgs"].astype(np.float32))
    proj = {"stage%d" % i: torch.from_numpy(g["proj_stage%d" % i]) for i in range(1, 5)}
    out = ref_model.dinomvsnet_forward(sd, imgs, proj, torch.from_numpy(g["depth_range"]), tmp=[float(t) for t in g["tmps"]])
    f1 = torch.from_numpy(g["features_stage1"])
    assert (out["features"]["stage1"] - f1).abs().max() < 2e-5 * f1.abs().max()
    for i in range(1, 5):
        want = torch.from_numpy(g["s%d_depth" % i])
        assert ((out["stage%d" % i]["depth"] - want).abs() / want.abs()).max() < 1e-4, i
    want = torch.from_numpy(g["refined_depth"])
    assert ((out["refined_depth"] - want).abs() / want.abs()).max() < 1e-4
    assert (out["photometric_confidence"] - torch.from_numpy(g["photometric_confidence"])).abs().max() < 1e-4


def test_vit_decoder_training_oracle_vs_reference():
    """oracle/ref_vit.vit_decoder(training=True) under torch autograd against the REAL ``VITDecoderStage4Single`` in train() (tests/golden/
    vit_decoder_train.npz): output, loss, the input gradients, every parameter gradient (sampled for the big tensors) and the running statistics."""
    from oracle import ref_vit
    from oracle.weights import load_vit_shapes, make_vit_state_dict
    g = load_golden("vit_decoder_train.npz")
    sd = make_vit_state_dict(load_vit_shapes("vit_decoder"), int(g["seeds"][0]))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    run = {k: (params[k] if k in params else v.clone()) for k, v in sd.items()}
    feat = torch.from_numpy(g["feat"].astype(np.float32)).requires_grad_(True)
    att = torch.from_numpy(g["att"].astype(np.float32)).requires_grad_(True)
    out = ref_vit.vit_decoder(run, feat, att, training=True)
    R = torch.randn(out.shape, generator=_gen_after(int(g["seeds"][1]), [(2, 384, 8, 10), (2, 6, 8, 10)]))
    loss = (out * R).sum()
    loss.backward()
    assert (out - torch.from_numpy(g["out"])).abs().max() < 2e-5 * float(np.abs(g["out"]).max())
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    for name, got in (("dfeat", feat.grad), ("datt", att.grad)):
        want = torch.from_numpy(g[name])
        assert (got - want).abs().max() < 1e-4 * want.abs().max(), name
    for k, p in params.items():
        want, idx = torch.from_numpy(g["grad." + k]), torch.from_numpy(g["idx." + k])
        got = p.grad.reshape(-1)[idx]
        if want.abs().max() < 1e-2:                        # conv bias in front of a batch-statistics BatchNorm: zero up to rounding noise
            assert got.abs().max() < 1e-2, k
            continue
        assert (got - want).abs().max() < 2e-4 * want.abs().max() + 1e-7, k
        assert abs(float(p.grad.double().norm()) - float(g["norm." + k])) < 2e-4 * float(g["norm." + k]) + 1e-7, k
    for k, v in run.items():
        if "running" in k:
            assert (v - torch.from_numpy(g["buf." + k])).abs().max() < 1e-5, k


def _gen_after(seed, shapes_rand):
    """The generator of oracle/gen_golden.py::gen_vit_decoder_train after it drew the inputs (randn for feat, rand for att): R comes next."""
    gen = torch.Generator().manual_seed(seed)
    torch.randn(shapes_rand[0], generator=gen)
    torch.rand(shapes_rand[1], generator=gen)
    return gen


def test_wasserstein_loss_oracle_vs_reference():
    """oracle/ref_losses.sinkhorn_stage (torch autograd through the ten log-domain iterations) against models/losses.py wasserstein_loss
    (tests/golden/was_loss.npz on the inputs of ce_loss.npz): the four weighted stage losses and d loss / d prob_volume."""
    from oracle import ref_losses
    gi, g = load_golden("ce_loss.npz"), load_golden("was_loss.npz")
    w = [float(x) for x in g["dlossw"]]
    inputs, gts, masks = {}, {}, {}
    for k in ("stage1", "stage2", "stage3", "stage4"):
        prob = torch.softmax(torch.from_numpy(gi["inv_%s_logits" % k]), 1).requires_grad_(True)
        inputs[k] = dict(depth_values=torch.from_numpy(gi["inv_%s_depth_values" % k]), prob_volume=prob)
        gts[k], masks[k] = torch.from_numpy(gi["inv_%s_gt" % k]), torch.from_numpy(gi["inv_%s_mask" % k])
    out = ref_losses.wasserstein_loss(inputs, gts, masks, w)
    sum(out.values()).backward()
    for k in inputs:
        assert abs(float(out[k]) - float(g[k + "_loss"])) < 2e-6 * max(1.0, abs(float(g[k + "_loss"]))), k
        wg = torch.from_numpy(g[k + "_grad"])
        assert (inputs[k]["prob_volume"].grad - wg).abs().max() < 2e-5 * wg.abs().max(), k
