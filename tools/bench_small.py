#!/usr/bin/env python
"""CostRegNet's layers at the coarse config-2 stages (stage 1: 32 x 144 x 192, stage 2: 16 x 288 x 384 input volumes): the tiled kernels
(fp32 MFMA / split-form plane sweep) against the small-volume split-form gather kernel (csrc/conv3d_x3_small.hip).
    python tools/bench_small.py  -> gpurun_out/bench_small.txt"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvsformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


_blk = torch.randn(8192, 8192, device=dev)


def timeit(fn, iters=30):
    """GPU duration of one launch (ms): a ~2 ms blocker kernel keeps the GPU busy while the host enqueues `iters` (event, launch, event)
    triples, so every pair brackets the kernel itself and not the host's launch cadence (~10-15 us per call from Python - longer than
    the small layers); the median pair is reported."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _blk @ _blk
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


lines = []
for st, (D, H, W) in ((1, (32, 144, 192)), (2, (16, 288, 384))):
    dims = [(D, H, W)]
    for _ in range(3):
        d, h, w = dims[-1]
        dims.append((d // 2, h // 2, w // 2))
    convs = [("conv1", 8, 16, 0, 2), ("conv2", 16, 16, 1, 1), ("conv3", 16, 32, 1, 2), ("conv4", 32, 32, 2, 1), ("conv5", 32, 64, 2, 2), ("conv6", 64, 64, 3, 1)]
    for name, cin, cout, lvl, s in convs:
        d, h, w = dims[lvl]
        x = torch.randn(1, cin, d, h, w, device=dev)
        wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        pk, ps = ops.conv3d_pack(wt, False), ops.conv3d_small_pack(wt, s, False)
        y0 = ops.conv3d(x, pk, cin, cout, (s, s), scale, shift, None, True)
        y1 = ops.conv3d_small(x, ps, cin, cout, s, False, scale, shift, None, True)
        err = (y1 - y0).abs().max().item() / y0.abs().max().item()
        t0 = timeit(lambda: ops.conv3d(x, pk, cin, cout, (s, s), scale, shift, None, True))
        t1 = timeit(lambda: ops.conv3d_small(x, ps, cin, cout, s, False, scale, shift, None, True))
        tx = float("nan")
        if ops.conv3d_x3_supported(cin, cout, (s, s)):
            px = ops.conv3d_x3_pack(wt, (s, s))
            tx = timeit(lambda: ops.conv3d_x3(x, px, cin, cout, (s, s), scale, shift, None, True))
        line = "stage%d %-6s %2d->%2d s%d out %3dx%3dx%3d (%7d voxels) | fp32 %.4f  x3 sweep %.4f  small %.4f ms | small vs fp32 %.1e" % (
            st, name, cin, cout, s, *y0.shape[2:], y0[0, 0].numel(), t0, tx, t1, err)
        print(line, flush=True)
        lines.append(line)
    for name, cin, cout, lvl in (("conv7", 64, 32, 3), ("conv9", 32, 16, 2), ("conv11", 16, 8, 1)):
        d, h, w = dims[lvl]
        x = torch.randn(1, cin, d, h, w, device=dev)
        wt = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
        scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        res = torch.randn(1, cout, 2 * d, 2 * h, 2 * w, device=dev)
        pk, ps = ops.conv3d_pack(wt, True, 2), ops.conv3d_small_pack(wt, 2, True)
        y0 = ops.deconv3d(x, pk, cin, cout, 2, scale, shift, res, True)
        y1 = ops.conv3d_small(x, ps, cin, cout, 2, True, scale, shift, res, True)
        err = (y1 - y0).abs().max().item() / y0.abs().max().item()
        t0 = timeit(lambda: ops.deconv3d(x, pk, cin, cout, 2, scale, shift, res, True))
        t1 = timeit(lambda: ops.conv3d_small(x, ps, cin, cout, 2, True, scale, shift, res, True))
        line = "stage%d %-6s %2d->%2d deconv out %3dx%3dx%3d (%7d voxels) | fp32 %.4f  small %.4f ms | small vs fp32 %.1e" % (
            st, name, cin, cout, *y0.shape[2:], y0[0, 0].numel(), t0, t1, err)
        print(line, flush=True)
        lines.append(line)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "bench_small.txt"), "w").write("\n".join(lines) + "\n")
