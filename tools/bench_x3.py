#!/usr/bin/env python
"""The regularizer's stride-1 layers at config-2 shapes: fp32-MFMA direct, Winograd fp32-MFMA and the 3-term bf16 split form.
    python tools/bench_x3.py [--stages 3,4]  -> gpurun_out/bench_x3.txt"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvsformer_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stages", default="1,2,3,4")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="", help="comma list of layer names (conv1 ... conv11, tail); default all")
ap.add_argument("--out", default="bench_x3.txt")
args = ap.parse_args()
dev = torch.device("cuda:0")
STAGES = {1: (32, 144, 192, 2), 2: (16, 288, 384, 2), 3: (8, 576, 768, 1), 4: (4, 1152, 1536, 1)}


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


lines = []
for st in [int(s) for s in args.stages.split(",")]:
    D, H, W, sd = STAGES[st]
    dims = [(D, H, W)]
    for _ in range(3):
        d, h, w = dims[-1]
        dims.append((d // sd, h // 2, w // 2))
    layers = [("conv1", 8, 16, 0, (sd, 2)), ("conv2", 16, 16, 1, (1, 1)), ("conv3", 16, 32, 1, (sd, 2)), ("conv4", 32, 32, 2, (1, 1)),
              ("conv5", 32, 64, 2, (sd, 2)), ("conv6", 64, 64, 3, (1, 1))]
    only = set(args.only.split(",")) if args.only else None
    for name, cin, cout, lvl, stride in layers:
        if only and name not in only:
            continue
        d, h, w = dims[lvl]
        if not ops.conv3d_x3_supported(cin, cout, stride):
            continue
        x = torch.randn(1, cin, d, h, w, device=dev)
        wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        pk, px = ops.conv3d_pack(wt, False), ops.conv3d_x3_pack(wt, stride)
        y0 = ops.conv3d(x, pk, cin, cout, stride, scale, shift, None, True)
        y2 = ops.conv3d_x3(x, px, cin, cout, stride, scale, shift, None, True)
        err = (y2 - y0).abs().max().item() / y0.abs().max().item()
        t0 = timeit(lambda: ops.conv3d(x, pk, cin, cout, stride, scale, shift, None, True))
        t1 = float("nan")
        if stride == (1, 1) and ops.conv3d_wino_supported(cin, cout, d, h, w):
            pw = ops.conv3d_wino_pack(wt)
            t1 = timeit(lambda: ops.conv3d_wino(x, pw, cin, cout, scale, shift, None, True))
        t2 = timeit(lambda: ops.conv3d_x3(x, px, cin, cout, stride, scale, shift, None, True))
        abl = ""
        if os.environ.get("X3_ABLATION"):
            for code, what in ((1, "no loads"), (2, "no split/LDS stores"), (3, "no staging"), (4, "no MFMA"), (7, "empty")):
                os.environ["MVS_X3_ABLATE"] = str(code)
                abl += "  [%s %.4f]" % (what, timeit(lambda: ops.conv3d_x3(x, px, cin, cout, stride, scale, shift, None, True)))
            os.environ.pop("MVS_X3_ABLATE")
        gf = 2.0 * 27 * cin * cout * y0[0, 0].numel() / 1e9
        line = "stage%d %-6s %2d->%2d %3dx%4dx%4d s%s %5.1f GF | direct %.4f ms  wino %.4f ms  x3 %.4f ms (%.1f TFLOP/s direct-form) | x3 vs direct max diff %.1e of scale" % (
            st, name, cin, cout, d, h, w, stride, gf, t0, t1, t2, gf / t2, err) + abl
        print(line, flush=True)
        lines.append(line)
    if sd == 1:
        for name, cin, cout, lvl in (("conv7", 64, 32, 3), ("conv9", 32, 16, 2), ("conv11", 16, 8, 1)):
            if only and name not in only:
                continue
            d, h, w = dims[lvl]
            x = torch.randn(1, cin, d, h, w, device=dev)
            wt = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
            scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
            res = torch.randn(1, cout, d, 2 * h, 2 * w, device=dev)
            pk, px = ops.conv3d_pack(wt, True, 1), ops.deconv3d_x3_pack(wt, 1)
            y0 = ops.deconv3d(x, pk, cin, cout, 1, scale, shift, res, True)
            y2 = ops.deconv3d_x3(x, px, cin, cout, 1, scale, shift, res, True)
            err = (y2 - y0).abs().max().item() / y0.abs().max().item()
            t0 = timeit(lambda: ops.deconv3d(x, pk, cin, cout, 1, scale, shift, res, True))
            t2 = timeit(lambda: ops.deconv3d_x3(x, px, cin, cout, 1, scale, shift, res, True))
            gf = 2.0 * 27 * cin * cout * d * h * w / 1e9
            line = "stage%d %-6s %2d->%2d %3dx%4dx%4d deconv %5.1f GF | direct %.4f ms  x3 %.4f ms (%.1f TFLOP/s direct-form) | max diff %.1e of scale" % (
                st, name, cin, cout, d, h, w, gf, t0, t2, gf / t2, err)
            print(line, flush=True)
            lines.append(line)
        if not only or "tail" in only:                       # conv11 + BatchNorm + ReLU + skip + 1x1x1 prob in one launch: fp32 MFMA vs split form
            d, h, w = dims[1]
            x = torch.randn(1, 16, d, h, w, device=dev)
            wt = torch.randn(16, 8, 3, 3, 3, device=dev) * 0.05
            scale, shift = torch.rand(8, device=dev) + 0.5, torch.randn(8, device=dev)
            res = torch.randn(1, 8, d, 2 * h, 2 * w, device=dev)
            pw, pb = torch.randn(8, device=dev), torch.randn(1, device=dev)
            pk, px = ops.conv3d_pack(wt, True, 1), ops.tail_x3_pack(wt)
            y0 = ops.deconv3d_prob1(x, pk, 16, scale, shift, res, pw, pb, True)
            y2 = ops.tail_x3(x, px, scale, shift, res, pw, pb, True)
            err = (y2 - y0).abs().max().item() / y0.abs().max().item()
            t0 = timeit(lambda: ops.deconv3d_prob1(x, pk, 16, scale, shift, res, pw, pb, True))
            t2 = timeit(lambda: ops.tail_x3(x, px, scale, shift, res, pw, pb, True))
            abl = ""
            if os.environ.get("X3_ABLATION"):
                for code, what in ((1, "no loads"), (4, "no MFMA"), (8, "no epilogue"), (13, "empty")):
                    os.environ["MVS_X3_ABLATE"] = str(code)
                    abl += "  [%s %.4f]" % (what, timeit(lambda: ops.tail_x3(x, px, scale, shift, res, pw, pb, True)))
                os.environ.pop("MVS_X3_ABLATE")
            gf = 2.0 * 27 * 16 * 8 * d * h * w / 1e9
            mb = (16 * d * h * w + 8 * d * 4 * h * w + d * 4 * h * w) * 4 / 1e6
            line = "stage%d tail   16-> 8+prob %3dx%4dx%4d %5.1f GF %6.1f MB | fp32 tail %.4f ms  x3 tail %.4f ms (%.1f TFLOP/s direct-form, %.0f GB/s) | max diff %.1e of scale" % (
                st, d, h, w, gf, mb, t0, t2, gf / t2, mb / t2, err) + abl
            print(line, flush=True)
            lines.append(line)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", args.out), "w").write("\n".join(lines) + "\n")
