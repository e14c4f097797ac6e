#!/usr/bin/env python
"""Winograd vs direct MFMA convolution on the stride-1 regularizer layers (conv2/conv4/conv6) at config-2 shapes,
plus the max deviation of both from torch's fp64 conv3d on a reduced volume.

    python tools/bench_wino.py [--stages 3,4] [--iters 20]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stages", default="3,4")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
STAGES = {1: (16, 72, 96, 2), 2: (8, 144, 192, 2), 3: (8, 288, 384, 1), 4: (4, 576, 768, 1)}     # dims at conv2's level


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.iters


torch.manual_seed(0)
for c in (16, 32, 64):
    x = torch.randn(1, c, 4, 24, 40, device=dev)
    wt = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
    ref = F.conv3d(x.double(), wt.double(), padding=1)
    a = ops.conv3d(x, ops.conv3d_pack(wt, False), c, c, (1, 1), None, None, None, False)
    b = ops.conv3d_wino(x, ops.conv3d_wino_pack(wt), c, c, None, None, None, False)
    print("C=%d  max|direct-ref| %.2e  max|wino-ref| %.2e  (ref max %.2f)" % (c, (a - ref).abs().max().item(), (b - ref).abs().max().item(),
                                                                                ref.abs().max().item()))
tot = [0.0, 0.0]
for st in [int(s) for s in args.stages.split(",")]:
    D, H, W, sd = STAGES[st]
    for lvl, c in enumerate((16, 32, 64)):
        d, h, w = D // sd ** lvl, H >> lvl, W >> lvl
        x = torch.randn(1, c, d, h, w, device=dev)
        wt = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
        scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        pk, pw = ops.conv3d_pack(wt, False), ops.conv3d_wino_pack(wt)
        t0 = timeit(lambda: ops.conv3d(x, pk, c, c, (1, 1), scale, shift, None, True))
        t1 = timeit(lambda: ops.conv3d_wino(x, pw, c, c, scale, shift, None, True))
        gf = 2.0 * 27 * c * c * d * h * w / 1e9
        tot[0] += t0
        tot[1] += t1
        print("stage%d %2d->%2d %3dx%4dx%4d  direct %.3f ms (%5.1f TF)   wino %.3f ms (%5.1f TF-equivalent)" % (st, c, c, d, h, w, t0, gf / t0,
                                                                                                            t1, gf / t1))
print("TOTAL direct %.3f ms   wino %.3f ms" % tuple(tot))
