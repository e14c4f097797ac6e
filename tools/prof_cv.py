#!/usr/bin/env python
"""Run only the fused cost-volume kernels (and optionally the vis CNN / one conv) at one stage geometry of
config 2 — a short target for rocprofv3 --pmc passes.

    python tools/prof_cv.py --stage 4 --iters 3
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m  # noqa: E402
from mvsformer_amd import ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=4)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--what", default="cv,vis")
args = ap.parse_args()
dev = torch.device("cuda:0")
scale = synth.STAGE_SCALES[args.stage - 1]
C = synth.STAGE_CHANNELS[args.stage - 1]
D = [32, 16, 8, 4][args.stage - 1]
scene = synth.make_scene(5, 1152, 1536, seed=0)
feat = synth.render_features(scene, scale, C, device=dev)
proj = synth.proj_matrices(scene, (scale,), device=dev)["stage1"]
H, W = 1152 // scale, 1536 // scale
z = synth.plane_depth(scene, scale, device=dev)
hyp = (1.0 / (1.0 / z[None, None] + torch.linspace(-1, 1, D, device=dev).view(1, D, 1, 1) * (4e-5 * scale))).contiguous()
net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), D, 0).to(dev).eval()
rt = ops.proj_prepare(proj)
feat = ops.to_channels_last(feat)                       # the sweeps read [B,V,H,W,C]
for _ in range(args.iters):
    if "cv" in args.what:
        ent = ops.cv_entropy(feat, rt, hyp, 8)
    else:
        ent = torch.rand(1, 4, H, W, device=dev)
    if "vis" in args.what:
        vp, vprep = net._vis_params()
        w = net._vis_weight(ent, vp, vprep)
    else:
        w = torch.rand(1, 4, H, W, device=dev)
    if "cv" in args.what:
        vol, sim = ops.cv_aggregate(feat, rt, hyp, w, 8, True)
    if "reg" in args.what:
        x = vol if "cv" in args.what else torch.randn(1, 8, D, H, W, device=dev)
        if hasattr(net.cost_reg, "logits"):
            net.cost_reg.logits(x)                          # CostRegNet3D: the path StageNet takes (conv11 + prob as the fused tail)
        else:
            net.cost_reg.features(x)
torch.cuda.synchronize()
print("done", H, W, C, D)
