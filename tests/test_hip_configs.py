"""GPU parity at the geometries of every BASELINE.json config (through the C ABI, against the CPU oracle):

* config 2 (DTU eval 1536x1152, 5 views): stage 1 at its REAL size straight against the oracle, and the full-size cascade
  through size-independent properties (kernel variants agree, multi-stream == single-stream bit for bit, the
  photo-consistent plane is found by the weight-free similarity arg-max);
* config 3 (training, cascade 32/16/8/8, 5 views): every stage's prob_volume_pre and gradients against torch autograd of
  the oracle's training formulation;
* config 4 (BlendedMVS stress, 7 views) and config 5 (Tanks&Temples, 11 views, 1920x1088): StageNet at the configs' view
  counts and channel widths on CPU-feasible H x W, including the per-view entropy and the aggregated volume.

Tolerances: depth 1e-3 relative per pixel (north star); fp32 intermediates a few 1e-5..1e-4 absolute (summation order);
``sim_depth`` is an arg-max: a pixel may differ from the oracle ONLY where the two candidates' similarity sums are tied to
within fp32 rounding of the sum (checked on every mismatching pixel, not as a mismatch fraction)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import max_abs, rel_err

pytestmark = pytest.mark.gpu
DEPTH_RTOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def assert_sim_depth_only_differs_on_ties(got_sim, hyp, sim_sum, tol_scale=1e-5):
    """``got_sim [B,H,W]`` (a hypothesis value per pixel), ``hyp [B,D,H,W]``, ``sim_sum [B,D,H,W]`` = oracle sum over views of
    the similarity.  Wherever the kernel picked another plane than the oracle, that plane's oracle similarity must equal the
    oracle's maximum to within rounding (the kernel's ``v_rsq_f32`` and summation order move each term by ~1 ulp)."""
    got_sim, hyp, sim_sum = got_sim.cpu().double(), hyp.cpu().double(), sim_sum.cpu().double()
    want_idx = sim_sum.argmax(1, keepdim=True)
    got_idx = (hyp - got_sim.unsqueeze(1)).abs().argmin(1, keepdim=True)
    assert ((torch.gather(hyp, 1, got_idx).squeeze(1) - got_sim).abs() == 0).all(), "sim_depth is not one of the hypotheses"
    gap = (torch.gather(sim_sum, 1, want_idx) - torch.gather(sim_sum, 1, got_idx)).squeeze(1)
    mism = got_idx.squeeze(1) != want_idx.squeeze(1)
    tol = tol_scale * sim_sum.abs().max().item()
    # ... except at the handful of samples whose sampling position is within rounding of the zero-padding border: there a tap of weight
    # ~1e-4 is inside for one side and outside for the other ("never test exact-border pixels for equality"), which moves that
    # plane's similarity by ~1e-4: at most 2e-5 of the pixels, and never by more than 25 tie tolerances
    over = gap[mism] > tol
    assert int(over.sum()) <= max(2, int(2e-5 * mism.numel())) and (gap[mism] <= 25 * tol).all(), \
        "sim_depth picked a plane %.3e below the oracle's maximum (tie tolerance %.1e) on %d px" % (gap[mism].max().item(), tol, int(over.sum()))
    return mism.double().mean().item()


def close_frac(got, want, atol, frac=1e-3, hard=None, what=""):
    """All samples within ``atol`` except a fraction ``frac`` (bilinear samples whose tap sits on the zero-padding border, or
    softmax columns with a near-degenerate maximum, move by more than the typical rounding), none beyond ``hard``."""
    err = (torch.as_tensor(got, dtype=torch.float64).cpu() - torch.as_tensor(want, dtype=torch.float64)).abs()
    bad = (err > atol).double().mean().item()
    assert bad <= frac, "%s: fraction above %g is %g (max %g)" % (what, atol, bad, err.max().item())
    if hard is not None:
        assert err.max().item() <= hard, "%s: max err %g > %g" % (what, err.max().item(), hard)


def run_stage_vs_oracle(dev, C, ndepth, H, W, V, full_hw, seed, B=1, tmp=5.0):
    import mvsformer_amd as m
    from mvsformer_amd import ops, synth
    from oracle import ref_torch
    scale = {64: 8, 32: 4, 16: 2, 8: 1}[C]
    torch.manual_seed(seed)
    net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), ndepth, 0).eval()
    m.randomize_bn_(net, seed + 1)
    # cameras of the FULL image (so per-stage intrinsics, baselines and disparities are the config's), features of a crop
    scene = synth.make_scene(V, full_hw[0], full_hw[1], seed=seed)
    scene.height, scene.width = H * scale, W * scale
    feat = synth.render_features(scene, scale, C, batch=B)
    proj = synth.proj_matrices(scene, (scale,), B)["stage1"]
    if ndepth >= 16:
        hyp = ref_torch.init_inverse_range(synth.depth_range(B), ndepth, H, W)
    else:                                   # a narrow per-pixel band around the true surface, like stages 3-4 sweep
        z = synth.plane_depth(scene, scale)
        hyp = (1.0 / (1.0 / z[None, None] + torch.linspace(1, -1, ndepth).view(1, ndepth, 1, 1) * (0.5e-5 * ndepth))).repeat(B, 1, 1, 1).contiguous()
    taps = {}
    with torch.no_grad():
        want = ref_torch.stage_forward(feat, proj, hyp, net.state_dict(), ndepth=ndepth, tmp=tmp, taps=taps)
    net = net.to(dev)
    with torch.no_grad():
        got = net(feat.to(dev), proj.to(dev), hyp.to(dev), tmp=tmp)
        rt = ops.proj_prepare(proj.to(dev))
        fg, hg = feat.to(dev).contiguous(), hyp.to(dev)
        wg = torch.cat(taps["vis_weight"], 1).to(dev).contiguous()
        fcl = ops.to_channels_last(fg)
        sweeps = {"direct": (ops.cv_entropy(fcl, rt, hg, 8), ops.cv_aggregate(fcl, rt, hg, wg, 8, False)[0]),
                  "direct_fast": (ops.cv_entropy(fcl, rt, hg, 8, exact=False), ops.cv_aggregate(fcl, rt, hg, wg, 8, False, exact=False)[0]),
                  "tiled": (ops.cv_tiled_entropy(fg, rt, hg, 8), ops.cv_tiled_aggregate(fg, rt, hg, wg, 8, False)[0])}
        if ops.cv_store_bytes(fcl, ndepth, 8) > 0:
            ent_s, store = ops.cv_corr(fcl, rt, hg, 8)
            sweeps["stored"] = (ent_s, ops.cv_merge(store, hg, wg, V, C, 8, False)[0])
    # every implementation in its default (reference op order) arithmetic, the direct sweeps also in the shortcut form.  The entropy is a softmax over sim[d] = sum_g in_prod: its rounding
    # error grows with the logits' magnitude (few channels per group -> unaveraged correlations, |sim| up to ~30 at C = 8), so
    # the tolerance is 2e-4 per ~5 units of |sim|; volume_mean is O(1)
    vscale = max(1.0, taps["volume_mean"].abs().max().item())
    sscale = max(1.0, max(ip.sum(1).abs().max().item() for ip in taps["in_prod"]) / 5.0)
    for impl, (ent, vol) in sweeps.items():
        close_frac(ent, torch.cat(taps["entropy"], 1), 2e-4 * sscale, frac=2e-3, hard=2e-3 * sscale, what=impl + " entropy")
        close_frac(vol, taps["volume_mean"], 2e-4 * vscale, frac=1e-3, hard=2e-3 * vscale, what=impl + " volume_mean")
    assert rel_err(got["depth"].cpu(), want["depth"]) < DEPTH_RTOL
    assert max_abs(got["prob_volume_pre"].cpu(), want["prob_volume_pre"]) < 5e-4
    assert max_abs(got["photometric_confidence"].cpu(), want["photometric_confidence"]) < 1e-4
    frac = assert_sim_depth_only_differs_on_ties(got["sim_depth"], hyp, taps["similarity_sum"])
    return frac


def test_config2_stage1_real_size(dev):
    """BASELINE configs[1], stage 1 exactly as benched: C=64, 144x192, D=32, V=5, CostRegNet - direct oracle comparison."""
    run_stage_vs_oracle(dev, 64, 32, 144, 192, 5, (1152, 1536), seed=21)


@pytest.mark.parametrize("C,ndepth,H,W", [(32, 16, 288, 384), (16, 8, 576, 768), (8, 4, 1152, 1536)])
def test_config2_stages_2_to_4_real_size(dev, C, ndepth, H, W):
    """BASELINE configs[1], stages 2-4 at the benched sizes (C=32/16/8, 288x384 ... 1152x1536, D=16/8/4, V=5): every sweep
    implementation, the regularizer (CostRegNet / CostRegNet3D with the fused tail) and the head directly against the CPU oracle
    (~10-20 s of oracle time per stage on the GPU box's host)."""
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    run_stage_vs_oracle(dev, C, ndepth, H, W, 5, (1152, 1536), seed=23 + C)


@pytest.mark.parametrize("C,ndepth,H,W", [(64, 32, 136, 240), (8, 4, 136, 240)])
def test_config5_eleven_views(dev, C, ndepth, H, W):
    """BASELINE configs[4]: 11 views at 1920x1088 - stage 1 at its real size (C=64, 136x240) and the 8-channel finest-stage
    kernels (pipelined (chunk, view) flattening with 10 source views) on a 136x240 crop."""
    run_stage_vs_oracle(dev, C, ndepth, H, W, 11, (1088, 1920), seed=31 + C)


@pytest.mark.parametrize("C,ndepth,H,W", [(16, 8, 96, 128), (32, 16, 48, 64)])
def test_config4_seven_views(dev, C, ndepth, H, W):
    """BASELINE configs[3]: 7 views, 2048x1536 cameras, stage-3 (C=16, D=8) and stage-2 (C=32, D=16) kernels on crops."""
    run_stage_vs_oracle(dev, C, ndepth, H, W, 7, (1536, 2048), seed=41 + C)


# ------------------------------------------------------------------------------------------------ config 3 (training)
def _oracle_stage_train(feat, proj, hyp, sd_tensors, ndepth, R):
    from oracle import ref_torch
    feat = feat.clone().requires_grad_(True)
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd_tensors.items()}
    out = ref_torch.stage_forward(feat, proj, hyp, sd, ndepth=ndepth, tmp=5.0, training=True)
    (out["prob_volume_pre"] * R).sum().backward()
    return out, feat.grad, {k: v.grad for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad}


def test_config3_train_cascade_stages(dev):
    """BASELINE configs[2] geometry (5 views, cascade 32/16/8/8, train mode) on a 256x320 image: every stage of the cascade,
    fed the oracle's own hypotheses, must reproduce the oracle's prob_volume_pre and its gradients w.r.t. features and all
    parameters (stages are coupled only through detached arg-max depths, so per-stage gradients ARE the cascade's)."""
    import mvsformer_amd as m
    from mvsformer_amd import synth
    from oracle import ref_torch
    nds, ratios = [32, 16, 8, 8], [4.0, 2.67, 1.5, 1.0]
    torch.manual_seed(5)
    net = m.CascadeMVS(dict(ndepths=nds, depth_interals_ratio=ratios)).train()
    feats, proj, dv, _ = synth.make_inputs(5, 256, 320, seed=6)
    sds = [{k: v.detach().clone() for k, v in f.state_dict().items()} for f in net.fusions]
    net = net.to(dev)
    prev = None
    for i in range(4):
        f = feats["stage%d" % (i + 1)]
        H, W = f.shape[-2:]
        hyp = ref_torch.init_inverse_range(dv, nds[0], H, W) if i == 0 else \
            ref_torch.schedule_inverse_range(prev["depth"].detach(), prev["depth_values"], nds[i], ratios[i], H, W)
        R = torch.randn(1, nds[i], H, W, generator=torch.Generator().manual_seed(i))
        want, dfeat, dparams = _oracle_stage_train(f, proj["stage%d" % (i + 1)], hyp, sds[i], nds[i], R)
        prev = {"depth": want["depth"].detach(), "depth_values": hyp}
        stage = net.fusions[i]
        stage.zero_grad(set_to_none=True)
        fg = f.to(dev).requires_grad_(True)
        got = stage(fg, proj["stage%d" % (i + 1)].to(dev), hyp.to(dev), tmp=5.0)
        (got["prob_volume_pre"] * R.to(dev)).sum().backward()
        scale = want["prob_volume_pre"].abs().max().item()
        assert max_abs(got["prob_volume_pre"].detach().cpu(), want["prob_volume_pre"].detach()) < 2e-4 * max(1.0, scale), i
        # train-mode depth is an arg-max gather: allow flips only between (near-)tied probabilities
        flips = (got["depth"].cpu() != want["depth"]).double().mean().item()
        assert flips < 0.02, (i, flips)
        # a bilinear tap on the zero-padding border may switch on/off under a 1-ulp coordinate difference and moves ONE scattered
        # gradient element by O(1): robust criterion (as tests/test_hip_training.py::test_aggregate_fn_grads), plus the mean
        gs = dfeat.abs().max().item()
        err = (fg.grad.cpu().double() - dfeat.double()).abs()
        assert (err > 3e-3 * gs).double().mean().item() < 5e-3, (i, err.max().item() / gs)     # 32x40 .. 256x320 maps: many border pixels
        assert err.mean().item() < 5e-4 * gs, (i, err.mean().item() / gs)
        # Parameter gradients in relative L2.  At these small maps ONE ReLU gate whose pre-activation is within an ulp of zero can open
        # on one side and stay shut on the other (measured: 2 of 81,920 voxels of conv9 have |z| < 1e-5 here); that single voxel moves
        # d loss / d beta of its channel by its whole upstream gradient and every gradient upstream of it by ~1 % - a tie, not an error
        # (tools/diag_train.py shows 1e-6 agreement on neighbouring shapes without such a voxel).  The visibility CNN's gradient comes
        # through d loss / d w_v, a difference of nearly equal sums, hence its looser bound.
        for name, p in stage.named_parameters():
            w = dparams[name].double()
            rel = ((p.grad.cpu().double() - w).norm() / (w.norm() + 1e-30)).item()
            assert rel < (5e-2 if name.startswith("vis.") else 3e-2), (i, name, rel)


def test_config3_train_cascade_runs_end_to_end(dev):
    """The same cascade as ONE training step (cascade-scheduled hypotheses, fused CE loss, backward): loss equals the oracle
    cascade's loss to 1e-3 and every parameter receives a finite gradient."""
    import mvsformer_amd as m
    from mvsformer_amd import synth
    from mvsformer_amd.losses import ce_loss_stage4
    from oracle import ref_losses, ref_torch
    nds, ratios = [32, 16, 8, 8], [4.0, 2.67, 1.5, 1.0]
    torch.manual_seed(7)
    net = m.CascadeMVS(dict(ndepths=nds, depth_interals_ratio=ratios)).train()
    feats, proj, dv, scene = synth.make_inputs(5, 128, 192, seed=8)
    gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s)[None] for i, s in enumerate(synth.STAGE_SCALES)}
    masks = {k: torch.ones_like(v) for k, v in gts.items()}
    sds = [{k: v.detach().clone() for k, v in f.state_dict().items()} for f in net.fusions]
    with torch.no_grad():
        ref = ref_torch.cascade_forward(feats, proj, dv, sds, ndepths=nds, depth_interals_ratio=ratios, tmp=[5.0, 5.0, 5.0, 1.0], training=True)
    want = sum(ref_losses.ce_loss_stage(ref["stage%d" % i]["prob_volume_pre"], ref["stage%d" % i]["depth_values"], gts["stage%d" % i],
                                        masks["stage%d" % i]).item() for i in range(1, 5))
    net = net.to(dev)
    out = net({k: v.to(dev).requires_grad_(True) for k, v in feats.items()}, {k: v.to(dev) for k, v in proj.items()}, dv.to(dev),
              tmp=[5.0, 5.0, 5.0, 1.0])
    loss = sum(ce_loss_stage4(out, {k: v.to(dev) for k, v in gts.items()}, {k: v.to(dev) for k, v in masks.items()}, [1, 1, 1, 1]).values())
    loss.backward()
    # later stages see arg-max-scheduled hypotheses: a flipped pixel changes its own column only, so the mean loss moves by O(flips)
    assert abs(loss.item() - want) < 2e-2 * abs(want), (loss.item(), want)
    for n_, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n_


# ------------------------------------------------------------------------------------------------ config 2 at full size
def _set_env(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    return old


def _restore_env(old):
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _reset_caches(net):
    for mod in net.modules():
        if hasattr(mod, "_cache"):
            mod._cache = None
        if hasattr(mod, "_dcache"):
            mod._dcache = {}
        if hasattr(mod, "_vis_cache"):
            mod._vis_cache = None


def test_config2_full_size_cascade_properties(dev):
    """The benched workload itself (1536x1152, 5 views, 32/16/8/4): (a) the fast kernel variants (Winograd convs, Winograd/MFMA
    visibility CNN, fused conv11+prob tail, tiled sweeps) and the plain ones give the same depth far inside the 1e-3 budget;
    (b) three reference views in flight on three HIP streams give bit-identical results to one stream (shared weight caches,
    shared inputs); (c) depths stay inside the swept hypothesis band at every stage and the weight-free similarity arg-max of
    stage 1 lands on the true plane (photo-consistency is found at real size)."""
    import mvsformer_amd as m
    from mvsformer_amd import synth
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, scene = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
    tmp = [5.0, 5.0, 5.0, 1.0]
    a = net(feats, proj, dv, tmp=tmp)
    torch.cuda.synchronize()
    # (c)
    for i in range(1, 5):
        st = a["stage%d" % i]
        lo, hi = st["depth_values"].min(1)[0], st["depth_values"].max(1)[0]
        assert ((st["depth"] >= lo * (1 - 1e-6)) & (st["depth"] <= hi * (1 + 1e-6))).all(), i
        assert torch.isfinite(st["prob_volume_pre"]).all()
    z = synth.plane_depth(scene, 8, device=dev)
    s1 = a["stage1"]
    hyp = s1["depth_values"][0]
    spacing = (hyp[:-1] - hyp[1:]).abs().max(0)[0]
    inner = (slice(8, -8), slice(8, -8))
    ok = ((s1["sim_depth"][0] - z).abs() <= 1.01 * spacing)[inner].double().mean().item()
    assert ok > 0.97, ok
    # (b)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    outs = []
    for i in range(6):
        with torch.cuda.stream(streams[i % 3]):
            outs.append(net(feats, proj, dv, tmp=tmp))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o["refined_depth"], a["refined_depth"])
        assert torch.equal(o["photometric_confidence"], a["photometric_confidence"])
        assert torch.equal(o["stage2"]["sim_depth"], a["stage2"]["sim_depth"])
    # (a)
    old = _set_env({"MVS_CONV_WINO": "0", "MVS_CONV_X3": "0", "MVS_VIS": "valu", "MVS_FUSE_PROB": "0", "MVS_CV_TILED": "0"})
    try:
        _reset_caches(net)
        b = net(feats, proj, dv, tmp=tmp)
        torch.cuda.synchronize()
    finally:
        _restore_env(old)
        _reset_caches(net)
    for i in range(1, 5):
        e = rel_err(b["stage%d" % i]["depth"].cpu(), a["stage%d" % i]["depth"].cpu())
        assert e < 2e-4, (i, e)
    assert max_abs(b["photometric_confidence"].cpu(), a["photometric_confidence"].cpu()) < 1e-3
