"""``StageNet`` — one cascade stage of the plane-sweep path, drop-in for the reference class of the same name
(models/mvsformer_model.py:26-160): same constructor ``StageNet(args, ndepth, stage_idx)``, same
``forward(features, proj_matrices, depth_values, tmp=2.0)``, same output dict keys, same ``state_dict`` keys
(``vis.{0,1,2}.{conv,bn}.*``, ``vis.3.*``, ``cost_reg.*``).

Where the reference runs ~40 ATen launches per source view over materialized [B,C,D,H,W] temporaries, this
forward is 5 + 10 hand-written HIP launches per stage:

    mvs_proj_prepare -> mvs_nchw_to_nhwc (skipped for channel-last inputs) -> sweep A -> mvs_vis_wino_fwd -> sweep B
    -> 9 fused conv/deconv MFMA layers -> (mvs_prob3_fwd) -> mvs_head_fwd

The sweeps are chosen per stage (``_cv_plan``): coarse stages keep the per-view correlation volumes sweep A computes
(``mvs_cv_corr_fwd``), so that sweep B is a pure stream over them (``mvs_cv_merge_fwd``); fine stages recompute the
correlation in sweep B (``mvs_cv_entropy_fwd`` / ``mvs_cv_aggregate_fwd``) because their per-view volumes no longer fit the
Infinity Cache.  ``MVS_CV_TILED=1`` opts into the LDS-tiled sweeps of cost_volume_tiled.hip (they lose on the noisy
hypotheses a random-weight cascade predicts).

``DepthNet`` is the name BASELINE.json uses for the same thing.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from ._lib import MvsHipError
from .module import ConvBnReLU, CostRegNet, CostRegNet3D, _versions, pack_vis_params


VIS_HALO = 3          # receptive-field radius of the visibility CNN (three 3x3 layers): rows a band needs beyond its own


def _store_plan(feat_cl, D, G) -> int:
    """How the stage builds its cost volume: 0 = recompute the correlation in sweep B; 1 = keep the per-view correlation volumes of the whole
    image (stored-correlation sweeps, built for C = 32 | 64); k > 1 = the same in k bands of reference rows, each band's store small enough to
    stay inside the 256 MB Infinity Cache next to the feature maps (MVS_CV_STORE_MAX_MB, default 160; 0 disables the path; MVS_CV_STORE_BANDS caps
    k, default 1 - banding is opt-in until it is measured faster at config-2 stage 2).  Measured at config 2 (profiles/r03_bench_sweeps.txt): stage 1 (127 MB) 0.30 -> 0.20 ms for
    the pair; stage 2 in one piece (254 MB: the round trip goes to HBM) 0.29 -> 0.32 ms."""
    limit = float(os.environ.get("MVS_CV_STORE_MAX_MB", "160")) * 2 ** 20
    max_bands = int(os.environ.get("MVS_CV_STORE_BANDS", "1"))
    nbytes = ops.cv_store_bytes(feat_cl, D, G)
    if nbytes <= 0 or limit <= 0:
        return 0
    if nbytes <= limit:
        return 1
    H = feat_cl.shape[2]
    for k in range(2, max_bands + 1):
        rows = -(-H // k) + 2 * VIS_HALO + 1                  # (+1: a band's first row is rounded down to even)
        if rows < H and nbytes * rows / H <= limit:
            return k
    return 0


class StageNet(nn.Module):
    def __init__(self, args, ndepth, stage_idx):
        super().__init__()
        self.args = args
        self.fusion_type = args.get("fusion_type", "cnn")
        self.ndepth = ndepth
        self.stage_idx = stage_idx
        in_channels = args["base_ch"]
        if self.fusion_type != "cnn":
            raise NotImplementedError(
                "fusion_type=%r: only 'cnn' (every shipped reference config) is built for MI355X" % self.fusion_type)
        model_th = args.get("model_th", 8)
        self.vis = nn.Sequential(ConvBnReLU(1, 16), ConvBnReLU(16, 16), ConvBnReLU(16, 8), nn.Conv2d(8, 1, 1), nn.Sigmoid())
        if ndepth <= model_th:
            self.cost_reg = CostRegNet3D(in_channels, args["base_ch"])
        else:
            self.cost_reg = CostRegNet(in_channels, args["base_ch"])
        self._vis_cache = None
        self._train_pack = None
        self._routed_by_cascade = False

    def _vis_params(self):
        """-> (parameter block, the 3x3 layers' weights prepared for the selected kernel or None for the all-VALU form)."""
        key = _versions(self.vis)
        if self._vis_cache is None or self._vis_cache[0] != key:
            params = pack_vis_params(self.vis)
            # MVS_VIS = x3 (default: split-form bf16 MFMA) | wino (Winograd fp32 MFMA) | valu (the all-VALU reference form)
            mode = os.environ.get("MVS_VIS", "x3" if os.environ.get("MVS_VIS_WINO", "1") != "0" else "valu")
            if mode not in ("x3", "wino", "valu"):
                raise ValueError("MVS_VIS must be x3, wino or valu, not %r" % mode)
            prepared = ops.vis_x3_prepare(params) if mode == "x3" else ops.vis_wino_prepare(params) if mode == "wino" else None
            from .module import _publish_cache
            _publish_cache()                                    # cached tensors are read from any stream afterwards
            self._vis_cache = (key, params, prepared)
        return self._vis_cache[1:]

    @staticmethod
    def _vis_weight(entropy, vis_params, vis_prepared):
        if vis_prepared is None:
            return ops.vis(entropy, vis_params)
        if vis_prepared.dtype == torch.uint8:
            return ops.vis_x3(entropy, vis_params, vis_prepared)
        return ops.vis_wino(entropy, vis_params, vis_prepared)

    def forward(self, features, proj_matrices, depth_values, tmp=2.0):
        """``features [B,V,C,H,W]`` (view 0 = reference), ``proj_matrices [B,V,2,4,4]``, ``depth_values [B,D,H,W]``."""
        depth_type = self.args["depth_type"]
        if features.shape[1] != proj_matrices.shape[1]:
            raise AssertionError("Different number of images and projection matrices")
        G = self.args["base_ch"]
        if self.training:
            return self._forward_train(features, proj_matrices, depth_values, tmp, G)
        feat = features.detach().to(torch.float32)              # layout handled by to_channels_last (zero-copy if NHWC already)
        proj = proj_matrices.detach().to(torch.float32).contiguous()
        hyp = depth_values.detach().to(torch.float32).contiguous()
        if hyp.dim() != 4:
            raise MvsHipError("depth_values must be [B,D,H,W]")

        # step 2 of the reference forward: fused warp + group correlation + visibility-weighted aggregation
        rt = ops.proj_prepare(proj)
        vis_params, vis_prepared = self._vis_params()
        # Sweep plan per stage: MVS_CV_TILED=1 opts into the LDS-tiled sweeps (they lose on the noisy hypotheses a
        # random-weight cascade predicts); coarse stages whose per-view correlation volumes fit the Infinity Cache keep them
        # (mvs_cv_corr_fwd) and merge by streaming (mvs_cv_merge_fwd); everything else recomputes the correlation in sweep B.
        tiled = os.environ.get("MVS_CV_TILED", "0") == "1" and ops.cv_tiled_supported(feat)
        if tiled:                                               # LDS-tiled sweeps straight from the decoder's NCHW maps
            entropy = ops.cv_tiled_entropy(feat, rt, hyp, G)
            weight = self._vis_weight(entropy, vis_params, vis_prepared)
            volume, sim_depth = ops.cv_tiled_aggregate(feat, rt, hyp, weight, G, want_sim_depth=True)
        else:                                                   # direct gather sweeps over channel-last maps (zero-copy if NHWC already)
            feat = ops.to_channels_last(feat)
            bands = _store_plan(feat, hyp.shape[1], G)
            if bands == 1:
                entropy, store = ops.cv_corr(feat, rt, hyp, G)
                weight = self._vis_weight(entropy, vis_params, vis_prepared)
                volume, sim_depth = ops.cv_merge(store, hyp, weight, feat.shape[1], feat.shape[4], G, want_sim_depth=True)
            elif bands > 1:
                # row bands: sweep A' on the band + VIS_HALO rows either side (the rows the visibility CNN's 7 x 7 receptive field reaches),
                # the CNN on the band as an image of its own (exact inside the band: its zero padding only touches the halo rows or the true
                # image border), sweep B' on the band without the halo - the band's store is read back while it is still in the cache
                B_, V_, H_, W_, C_ = feat.shape
                D_ = hyp.shape[1]
                volume = torch.empty(B_, G, D_, H_, W_, device=feat.device, dtype=torch.float32)
                sim_depth = torch.empty(B_, H_, W_, device=feat.device, dtype=torch.float32)
                hb = -(-H_ // bands)
                store = None
                for r0 in range(0, H_, hb):
                    r1 = min(H_, r0 + hb)
                    # an EVEN first row: the MFMA visibility kernels compute output rows in pairs (x3: one tile holds rows 2p and 2p + 1 with the
                    # taps at different K positions; wino: F(2x2,3x3)), so a row's rounding depends on its parity in the image the CNN is given
                    y0, y1 = max(0, (r0 - VIS_HALO) & ~1), min(H_, r1 + VIS_HALO)
                    entropy, store = ops.cv_corr_rows(feat, rt, hyp, G, y0, y1 - y0, store)
                    weight = self._vis_weight(entropy, vis_params, vis_prepared)
                    ops.cv_merge_rows(store, hyp, weight, V_, C_, G, y0, r0 - y0, r1 - r0, volume, sim_depth)
            else:
                entropy = ops.cv_entropy(feat, rt, hyp, G)
                weight = self._vis_weight(entropy, vis_params, vis_prepared)
                volume, sim_depth = ops.cv_aggregate(feat, rt, hyp, weight, G, want_sim_depth=True)

        # step 3: regularization + head
        if type(tmp) == list:
            tmp = tmp[self.stage_idx]
        if isinstance(self.cost_reg, CostRegNet3D):
            logits = self.cost_reg.logits(volume)               # conv11 + 1x1x1 prob fused: the 8-channel volume is never written
        else:
            x = self.cost_reg.features(volume)
            logits = ops.prob3(x, self.cost_reg.prob.weight.detach().to(torch.float32).contiguous())
        if depth_type in ("ce", "was"):
            pre, prob, depth, conf = ops.head(hyp, float(tmp), False, logits=logits)
        elif depth_type == "mixup_ce":                          # mvsformer_model.py:126-136
            pre, prob, _, _ = ops.head(hyp, 1.0, False, logits=logits)
            depth, conf = ops.mixup_head(prob, hyp)
        else:                                                   # regression head, mvsformer_model.py:137-146 (tmp unused)
            pre, prob, depth, conf = ops.head(hyp, 1.0, False, logits=logits)
            n = self._conf_window()
            if n:
                conf = ops.conf_regression(prob, n)
        return {"depth": depth, "prob_volume": prob, "photometric_confidence": conf, "depth_values": depth_values,
                "prob_volume_pre": pre, "sim_depth": sim_depth}


    def _conf_window(self):
        """Window of conf_regression for the regression head (mvsformer_model.py:139-146); 0 = plain max probability."""
        return 4 if self.ndepth >= 32 else 3 if self.ndepth == 16 else 2 if self.ndepth == 8 else 0

    def train_pack(self):
        """This stage's :class:`autograd.StagePack` (bf16 weight layouts of a training step in one launch + weight-gradient routing)."""
        from . import autograd as ag
        if self._train_pack is None or not self._train_pack.valid():
            self._train_pack = ag.StagePack(self)
        return self._train_pack

    def _forward_train(self, features, proj_matrices, depth_values, tmp, G):
        """Training branch (reference mvsformer_model.py:62-125 with ``self.training``): no similarity branch, batch-statistics
        BatchNorm everywhere, depth = hypothesis at the arg-max probability; gradients via :mod:`mvsformer_amd.autograd`."""
        from . import autograd as ag
        from .module import autocast_bf16
        if autocast_bf16() and ag._fused_layers():              # every bf16 weight layout of this stage's step in one launch
            pack = self.train_pack()
            pack.run(route=not self._routed_by_cascade)
            try:
                return self._forward_train_body(features, proj_matrices, depth_values, tmp, G)
            finally:
                if not self._routed_by_cascade:
                    pack.unroute()                              # the routed weights belong to this forward's graph only
        return self._forward_train_body(features, proj_matrices, depth_values, tmp, G)

    def _forward_train_body(self, features, proj_matrices, depth_values, tmp, G):
        from . import autograd as ag
        from .module import autocast_bf16
        proj = proj_matrices.detach().to(torch.float32).contiguous()
        hyp = depth_values.detach().to(torch.float32).contiguous()
        rt = ops.proj_prepare(proj)
        feat_cl = ops.to_channels_last(features.detach().to(torch.float32))
        entropy = ops.cv_entropy(feat_cl, rt, hyp, G, exact=True)           # sim_vol.detach() in the reference
        V = features.shape[1]
        weight = ag.vis_train_views(entropy, self.vis)                     # per-view statistics, one batched pass
        # under autocast the volume leaves the aggregation as bf16 channel-last too (the layout the bf16 regularizer reads) in the same launch
        volume = ag.AggregateFn.apply(features, weight, rt, hyp, G, feat_cl, autocast_bf16() and ag._fused_layers())
        if type(tmp) == list:
            tmp = tmp[self.stage_idx]
        pre = self.cost_reg(volume).squeeze(1)
        depth_type = self.args["depth_type"]
        if depth_type in ("ce", "was"):
            prob, depth, conf = ag.HeadFn.apply(pre, hyp, float(tmp))
        else:
            # The mixup / regression heads are a few elementwise ops on [B,D,H,W]; in training they run as torch ops on the HIP
            # regularizer's logits so that depth and prob_volume are differentiable (mvsformer_model.py:126-146).
            prob = torch.softmax(pre, dim=1)
            if depth_type == "mixup_ce":
                left, right = prob[:, :-1], prob[:, 1:]
                conf, idx = torch.max(left + right, dim=1)
                norm = left + right + 1e-7
                mix = hyp[:, :-1] * (left / norm) + hyp[:, 1:] * (right / norm)
                depth = torch.gather(mix, 1, idx.unsqueeze(1)).squeeze(1)
            else:
                depth = torch.sum(prob * hyp, dim=1)
                n = self._conf_window()
                conf = ops.conf_regression(prob.detach().contiguous(), n) if n else prob.max(1)[0]
        return {"depth": depth, "prob_volume": prob, "photometric_confidence": conf.detach(), "depth_values": depth_values,
                "prob_volume_pre": pre}


DepthNet = StageNet
