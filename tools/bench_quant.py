#!/usr/bin/env python
"""Wave quantization of the x3 convolutions: time per tile as the grid grows through multiples of the resident-block count."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvsformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for cin, cout, stride, th in ((16, 16, (1, 1), 16), (32, 32, (1, 1), 8), (64, 64, (1, 1), 8)):
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    px = ops.conv3d_x3_pack(wt, stride)
    W = 768 if cin == 16 else 384
    for rows_of_tiles in (8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48):
        H = rows_of_tiles * th
        x = torch.randn(1, cin, 4, H, W, device=dev)
        t = timeit(lambda: ops.conv3d_x3(x, px, cin, cout, stride, None, None, None, True))
        tiles = rows_of_tiles * (W // 16) * max(1, cout // 32)
        print("conv %d->%d D=4 %4dx%4d: %5d blocks  %.4f ms  %.2f us per 100 blocks" % (cin, cout, H, W, tiles, t, t * 1e5 / tiles), flush=True)
