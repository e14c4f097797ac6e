#!/usr/bin/env python
"""Per-layer timing of the regularizer convolutions at config-2 shapes (all four stages).

    python tools/bench_conv.py [--stages 3,4] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stages", default="1,2,3,4")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
STAGES = {1: (32, 144, 192, True), 2: (16, 288, 384, True), 3: (8, 576, 768, False), 4: (4, 1152, 1536, False)}


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


tot_ms, tot_gf = 0.0, 0.0
for st in [int(s) for s in args.stages.split(",")]:
    D, H, W, s2 = STAGES[st]
    sd = 2 if s2 else 1
    dims = [(D, H, W)]
    for _ in range(3):
        d, h, w = dims[-1]
        dims.append((d // sd, h // 2, w // 2))
    layers = [("conv1", 8, 16, 0, (sd, 2)), ("conv2", 16, 16, 1, (1, 1)), ("conv3", 16, 32, 1, (sd, 2)), ("conv4", 32, 32, 2, (1, 1)),
              ("conv5", 32, 64, 2, (sd, 2)), ("conv6", 64, 64, 3, (1, 1)), ("conv7", 64, 32, 3, None), ("conv9", 32, 16, 2, None),
              ("conv11", 16, 8, 1, None)]
    for name, cin, cout, lvl, stride in layers:
        d, h, w = dims[lvl]
        x = torch.randn(1, cin, d, h, w, device=dev)
        scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        if stride is not None:
            wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
            pk = ops.conv3d_pack(wt, False)
            fn = lambda: ops.conv3d(x, pk, cin, cout, stride, scale, shift, None, True)
            vox = (d // stride[0]) * (h // stride[1]) * (w // stride[1])
        else:
            wt = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
            pk = ops.conv3d_pack(wt, True, sd)
            fn = lambda: ops.deconv3d(x, pk, cin, cout, sd, scale, shift, None, True)
            vox = d * h * w
        ms = timeit(fn)
        gf = 2.0 * 27 * cin * cout * vox / 1e9
        tot_ms += ms
        tot_gf += gf
        print("stage%d %-6s %2d->%2d in %3dx%4dx%4d %-8s %7.3f ms %6.1f GF %6.1f TF/s" % (st, name, cin, cout, d, h, w,
              "s%s" % (stride,) if stride else "deconv", ms, gf, gf / ms))
print("TOTAL %.3f ms  %.1f GF  %.1f TF/s   (MVS_CONV_TILE=%s)" % (tot_ms, tot_gf, tot_gf / tot_ms, os.environ.get("MVS_CONV_TILE")))
