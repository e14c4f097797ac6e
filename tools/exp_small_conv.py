#!/usr/bin/env python
"""Per-shape time of the bf16 training convolution at the coarse levels of config 3 (the K-split launches): graph of 20 calls, per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import ops
dev = torch.device("cuda:0")
shapes = [  # (D, H, W, cin, cout, gather, stride)
    (4, 8, 10, 64, 64, 0, (1, 1)), (2, 16, 20, 64, 64, 0, (1, 1)), (1, 32, 40, 64, 64, 0, (1, 1)), (1, 64, 80, 64, 64, 0, (1, 1)),
    (4, 64, 80, 64, 64, 0, (1, 1)), (2, 64, 80, 64, 64, 0, (1, 1)), (4, 128, 160, 32, 64, 0, (1, 2)), (2, 128, 160, 32, 64, 0, (1, 2)),
    (16, 32, 40, 32, 32, 0, (1, 1)), (8, 64, 80, 32, 32, 0, (1, 1)),
    (8, 16, 20, 32, 32, 0, (1, 1)), (8, 128, 160, 32, 64, 0, (1, 2)), (8, 64, 80, 64, 64, 0, (1, 1)), (4, 8, 10, 64, 32, 1, (2, 2)),
]
for D, H, W, cin, cout, g, st in shapes:
    x = torch.randn(1, D, H, W, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) if g == 0 else torch.randn(cin, cout, 3, 3, 3, device=dev)
    wp = ops.bf16_pack(w, g, cin, cout)
    for _ in range(3):
        y = ops.bf16_conv3d(x, wp, cin, cout, g, st)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(20):
                y = ops.bf16_conv3d(x, wp, cin, cout, g, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.replay(); torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        gr.replay()
    b.record(); torch.cuda.synchronize()
    print("D%d H%d W%d %d->%d g%d s%s: %.1f us per call (out voxels %d)" % (D, H, W, cin, cout, g, st, a.elapsed_time(b) / 100 * 1e3, y.numel() // cout))
