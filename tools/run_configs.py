#!/usr/bin/env python
"""Run the other BASELINE.json configurations through the HIP path (shape coverage, not the judged metric):
config 5 (T&T: 11 views, 1920x1088), config 4 (BlendedMVS hi-res stress: 7 views, 2048x1536), config 1 geometry
(4 views, 640x512, single stage D=48) and the single-stage D=192 stress shapes of SURVEY.md §8(d)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m  # noqa: E402
from mvsformer_amd import synth  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for name, V, H, W in (("config5 T&T 11 views 1920x1088", 11, 1088, 1920), ("config4 BlendedMVS 7 views 2048x1536", 7, 1536, 2048)):
    torch.manual_seed(0)
    net = m.CascadeMVS().to(dev).eval()
    m.randomize_bn_(net, 1)
    feats, proj, dv, _ = synth.make_inputs(V, H, W, seed=0, device=dev)
    ms, out = timed(lambda: net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0]))
    assert torch.isfinite(out["refined_depth"]).all()
    print("%-44s %8.2f ms / depth map  (%.1f depth maps/s)  peak mem %.1f GB" % (name, ms, 1e3 / ms, torch.cuda.max_memory_allocated() / 1e9))
    del net, feats, out
    torch.cuda.empty_cache()

for name, V, C, Hs, Ws, D in (("config1 stage geometry V=4 C=64 64x80 D=48", 4, 64, 64, 80, 48),
                              ("stress V=5 C=64 144x192 D=192", 5, 64, 144, 192, 192), ("stress V=5 C=32 288x384 D=192", 5, 32, 288, 384, 192)):
    scale = 8 if C == 64 else 4
    scene = synth.make_scene(V, Hs * scale, Ws * scale, seed=1)
    feat = synth.render_features(scene, scale, C, device=dev)
    proj = synth.proj_matrices(scene, (scale,), device=dev)["stage1"]
    net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), D, 0).to(dev).eval()
    hyp = m.init_inverse_range(synth.depth_range(1, device=dev), D, dev, torch.float32, Hs, Ws)
    ms, out = timed(lambda: net(feat, proj, hyp, tmp=5.0))
    assert torch.isfinite(out["depth"]).all()
    print("%-44s %8.2f ms / stage" % (name, ms))
