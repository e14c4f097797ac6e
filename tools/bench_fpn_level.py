#!/usr/bin/env python
"""The FPN decoder's top-down levels alone at BASELINE configs[1]'s geometry (5 views, 1152x1536): median launch time of the split-form
kernels (csrc/fpn_x3.hip) next to the fp32-MFMA kernels of csrc/fpn.hip they replace.  MVS_HIP_LIB selects an experiment build.

    python tools/bench_fpn_level.py [--views 5] [--height 1152] [--width 1536]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import FPNDecoder, ops  # noqa: E402


def timeit(fn, rounds=7, inner=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--levels", default="3")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dec = FPNDecoder([8, 16, 32, 64]).eval().to(dev)
    (w0, s0, h0), levels = dec._prepared()
    for k in [int(c) for c in a.levels.split(",")]:
        ck = {1: 32, 2: 16, 3: 8}[k]
        H, W = a.height >> (3 - k), a.width >> (3 - k)
        prev = torch.randn(a.views, 64, H // 2, W // 2, device=dev)
        lat = torch.randn(a.views, ck, H, W, device=dev)
        w_in, b_in, packed, scale, shift, x3 = levels[k - 1]
        flops = 2.0 * 64 * ck * 10 * a.views * H * W
        med, mn = timeit(lambda: ops.fpn_level(prev, lat, w_in, b_in, packed, scale, shift, want_intra=(k < 3)))
        print("level %d (Ck %2d) fp32-MFMA kernel   %.4f ms (min %.4f)  %6.1f TFLOP/s" % (k, ck, med, mn, flops / med / 1e9))
        if x3 is not None:
            prepared, shift_x, border, prepared_cp = x3
            med, mn = timeit(lambda: ops.fpn_level_x3(prev, lat, prepared, shift_x, border))
            print("level %d (Ck %2d) split form, strips   %.4f ms (min %.4f)  %6.1f TFLOP/s  [%s]" % (k, ck, med, mn, flops / med / 1e9, os.environ.get("MVS_HIP_LIB", "shipped")))
            prev_cl, lat_cl = prev.permute(0, 2, 3, 1).contiguous(), lat.permute(0, 2, 3, 1).contiguous()
            med, mn = timeit(lambda: ops.fpn_level_cp(prev_cl, lat_cl, prepared_cp, shift_x, border))
            print("level %d (Ck %2d) split form, contraction first (channel-last sources)  %.4f ms (min %.4f)  %6.1f TFLOP/s  [%s]" % (k, ck, med, mn, flops / med / 1e9, os.environ.get("MVS_HIP_LIB", "shipped")))
            a, b = ops.fpn_level_x3(prev, lat, prepared, shift_x, border), ops.fpn_level_cp(prev_cl, lat_cl, prepared_cp, shift_x, border)
            print("   the two forms differ by %.2e of scale" % float((a - b).abs().max() / a.abs().max()))


if __name__ == "__main__":
    main()
