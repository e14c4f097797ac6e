#!/usr/bin/env python
"""Time the depth-map consistency filter at test.py's working size: 1152x1536, 10 source views.

    python tools/bench_fusion.py [--views 10] [--cpu]

Prints one JSON line: fused one-pass kernel vs the reference-shaped three-call sequence on the GPU, the algorithmic
HBM bytes (depth maps in, mask/ave/points out) over the kernel time against the 8 TB/s roofline, and (with --cpu) the
torch-CPU oracle timed on a quarter-size sample.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import fusion  # noqa: E402
from oracle import ref_fusion  # noqa: E402  (input generator + cpu_baseline leg only)


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=10)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    case = ref_fusion.make_fusion_case(n=1, v=a.views, h=a.height, w=a.width, seed=0)
    rd, sd, rc, sc = [case[k].to(dev) for k in ("ref_depth", "src_depths", "ref_cam", "src_cams")]
    thr = (1.0, 0.01, 3)
    fused_ms = timed(lambda: fusion.filter_depth_maps(rd, sd, rc, sc, *thr), a.iters)

    def three_calls():
        reproj, in_range = fusion.get_reproj(rd, sd, rc, sc)
        masks, mask = fusion.vis_filter(rd, reproj, in_range, *thr)
        return fusion.ave_fusion(rd, reproj, masks), mask

    three_ms = timed(three_calls, a.iters)
    px = a.height * a.width
    algo = px * (4.0 * (1 + a.views) + 1 + 4 + 12)
    line = dict(metric="consistency_filter_ref_views_per_sec", value=1e3 / fused_ms, unit="reference views/s",
                config=dict(workload="%dx%d, %d source views" % (a.width, a.height, a.views)), fused_ms=fused_ms, three_call_ms=three_ms,
                roofline=dict(bound="hbm", achieved=algo / fused_ms / 1e6, peak=8000.0, unit="GB/s", frac=algo / fused_ms / 1e6 / 8000.0,
                              traffic=None), dtype="f32")
    if a.cpu:
        small = ref_fusion.make_fusion_case(n=1, v=a.views, h=a.height // 2, w=a.width // 2, seed=0)
        torch.set_num_threads(min(64, os.cpu_count() or 1))
        t0 = time.time()
        ref_fusion.filter_depth_maps(small["ref_depth"], small["src_depths"], small["ref_cam"], small["src_cams"], *thr)
        dt = time.time() - t0
        line["cpu_baseline"] = dict(value=1.0 / (dt * 4), unit="reference views/s", cores=torch.get_num_threads(), kind="port",
                                    sample="1 view at %dx%d (quarter of the pixels), time x4" % (a.width // 2, a.height // 2))
    print(json.dumps(line))


if __name__ == "__main__":
    main()
