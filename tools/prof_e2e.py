#!/usr/bin/env python
"""images -> depth map through mvsformer_amd.DINOMVSNet at BASELINE configs[1]'s geometry (what bench.py's `end_to_end_mvsformer_p` times), a few
forwards for a kernel trace: `tools/prof_e2e_trace.sh` runs it under `rocprofv3 --kernel-trace --stats` (every launch, the ATen glue included).

    python tools/prof_e2e.py [--views 5] [--height 1152] [--width 1536] [--iters 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import DINOMVSNet, synth  # noqa: E402
from mvsformer_amd.cascade import randomize_bn_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DINOMVSNet(dict(fix=True, depth_type="ce", fusion_type="cnn", inverse_depth=True, base_ch=8, ndepths=[32, 16, 8, 4], feat_chs=[8, 16, 32, 64],
                          depth_interals_ratio=[4.0, 2.67, 1.5, 1.0], multi_scale=False,
                          vit_args=dict(twin=False, rescale=0.5, patch_size=16, qk_scale="default", vit_arch="vit_small", vit_ch=384, out_ch=64,
                                        att_fusion=True, nhead=6))).eval()
    randomize_bn_(net, seed=1)
    net = net.to(dev)
    _, proj, dv, _ = synth.make_inputs(a.views, a.height, a.width, seed=0, device=dev)
    imgs = synth.render_features(synth.make_scene(a.views, a.height, a.width, 0), 1, 3, noise=0.02, device=dev, dtype=torch.float32)
    tmp = [5.0, 5.0, 5.0, 1.0]
    for _ in range(2):
        out = net(imgs, proj, dv, tmp=tmp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        out = net(imgs, proj, dv, tmp=tmp)
    e1.record()
    torch.cuda.synchronize()
    assert torch.isfinite(out["refined_depth"]).all()
    print("images -> depth map: %.3f ms (%d forwards + 2 warm-up)" % (e0.elapsed_time(e1) / a.iters, a.iters))


if __name__ == "__main__":
    main()
