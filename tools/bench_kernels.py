#!/usr/bin/env python
"""Time individual hot-path kernels at one config-2 stage geometry (smooth hypotheses): sweeps, vis CNN, head.

    python tools/bench_kernels.py --stage 4
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
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
scale, C, D = synth.STAGE_SCALES[args.stage - 1], synth.STAGE_CHANNELS[args.stage - 1], [32, 16, 8, 4][args.stage - 1]
scene = synth.make_scene(5, 1152, 1536, seed=0)
feat = synth.render_features(scene, scale, C, device=dev)
proj = synth.proj_matrices(scene, (scale,), device=dev)["stage1"]
H, W = 1152 // scale, 1536 // scale
z = synth.plane_depth(scene, scale, device=dev)
hyp = (1.0 / (1.0 / z[None, None] + torch.linspace(-1, 1, D, device=dev).view(1, D, 1, 1) * (4e-5 * scale))).contiguous()
net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), D, 0).to(dev).eval()
rt = ops.proj_prepare(proj)
fcl = ops.to_channels_last(feat)
ent = ops.cv_entropy(fcl, rt, hyp, 8)
w = ops.vis(ent, net._vis_params()[0])
vol, _ = ops.cv_aggregate(fcl, rt, hyp, w, 8, True)


def timeit(name, fn, work=None, unit=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.iters
    extra = "" if work is None else "  %8.1f %s" % (work / ms / (1e6 if unit == "GB/s" else 1e9), unit)
    print("stage%d %-22s %8.4f ms%s" % (args.stage, name, ms, extra))


V = 5
alg = 4.0 * H * W * (V * C + D + 8 * D)
timeit("nchw_to_nhwc", lambda: ops.to_channels_last(feat), 8.0 * feat.numel(), "GB/s")
timeit("cv_entropy", lambda: ops.cv_entropy(fcl, rt, hyp, 8), 4.0 * H * W * (V * C + D), "GB/s")
timeit("vis", lambda: ops.vis(ent, net._vis_params()[0]), 2.0 * 3608 * 4 * H * W, "TFLOP/s")
timeit("vis (default kernel)", lambda: net._vis_weight(ent, *net._vis_params()), 2.0 * 3608 * 4 * H * W, "TFLOP/s")
timeit("cv_aggregate(sim)", lambda: ops.cv_aggregate(fcl, rt, hyp, w, 8, True), alg, "GB/s")
timeit("cv_aggregate(nosim)", lambda: ops.cv_aggregate(fcl, rt, hyp, w, 8, False), alg, "GB/s")
if D <= 8:
    x8 = torch.randn(1, 8, D, H, W, device=dev)
    w1, b1 = net.cost_reg.prob_params()
    timeit("head(fused 1x1x1)", lambda: ops.head(hyp, 5.0, False, x8=x8, w1=w1, b1=b1), 4.0 * H * W * (8 * D + D + 2 * D + 2), "GB/s")
else:
    x8 = torch.randn(1, 8, D, H, W, device=dev)
    pw = net.cost_reg.prob.weight.detach().contiguous()
    timeit("prob3", lambda: ops.prob3(x8, pw), 4.0 * H * W * 9 * D, "GB/s")
    lg = torch.randn(1, D, H, W, device=dev)
    timeit("head", lambda: ops.head(hyp, 5.0, False, logits=lg), 4.0 * H * W * (D + D + D + 2), "GB/s")
