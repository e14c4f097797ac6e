#!/usr/bin/env python
"""The visibility CNN's three kernels (all-VALU, Winograd fp32-MFMA, split-form bf16-MFMA) at the four config-2 stage geometries.
    python tools/bench_vis.py  -> gpurun_out/bench_vis.txt"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvsformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
prm = torch.randn(ops.VIS_PARAM_FLOATS, device=dev) * 0.2
for a, b in ((144, 160), (2480, 2496), (3664, 3672)):
    prm[a:b] = prm[a:b].abs() + 0.5
pw, px = ops.vis_wino_prepare(prm), ops.vis_x3_prepare(prm)


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


lines = []
for stage, (H, W) in enumerate(((144, 192), (288, 384), (576, 768), (1152, 1536)), 1):
    ent = torch.rand(4, H, W, device=dev) * 3.0
    ref = ops.vis(ent, prm)
    tw, tx = timeit(lambda: ops.vis_wino(ent, prm, pw)), timeit(lambda: ops.vis_x3(ent, prm, px))
    tv = timeit(lambda: ops.vis(ent, prm), 5)
    gf = 2.0 * 3608 * ent.numel() * 1e-9
    line = "stage%d 4x%4dx%4d  %5.1f GF | valu %.4f ms  wino %.4f ms  x3 %.4f ms (%.1f TFLOP/s direct-form) | max diff vs valu: wino %.1e  x3 %.1e" % (
        stage, H, W, gf, tv, tw, tx, gf / tx, (ops.vis_wino(ent, prm, pw) - ref).abs().max().item(), (ops.vis_x3(ent, prm, px) - ref).abs().max().item())
    print(line, flush=True)
    lines.append(line)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "bench_vis.txt"), "w").write("\n".join(lines) + "\n")
