#!/usr/bin/env python
"""Winograd conv kernel time vs number of input-channel chunks (fixed cost vs per-chunk cost)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (d, h, w) in ((4, 576, 768), (4, 288, 384)):
    for cout in (16, 32, 64):
        for cin in (4, 16, 32, 64):
            x = torch.randn(1, cin, d, h, w, device=dev)
            wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
            pw = ops.conv3d_wino_pack(wt)
            t = timeit(lambda: ops.conv3d_wino(x, pw, cin, cout, None, None, None, True))
            mf = 2.0 * 12 * cin * cout * d * h * w / 1e9      # executed MFMA GF (16 xi * 3 kd / 4 outputs = 12 MAC per output/cin/cout)
            print("%dx%dx%d cin %2d cout %2d  %.3f ms  raw MFMA %.1f TF" % (d, h, w, cin, cout, t, mf / t))
