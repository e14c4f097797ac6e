#!/usr/bin/env python
"""Times the FPN decoder (csrc/fpn.hip) on BASELINE configs[1]'s geometry (5 views, 1152x1536) with HIP events per launch, next to
the same ops in plain PyTorch (MIOpen convolutions + ATen upsampling) on the same GPU.  Prints one JSON object.

    python tools/bench_fpn.py [--views 5] [--height 1152] [--width 1536] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import FPNDecoder, ops  # noqa: E402


def torch_decoder(sd, conv01, conv11, conv21, conv31):
    """The reference's op sequence (models/module.py:257-270) in plain torch - the comparison leg, not the product."""
    import torch.nn.functional as F

    def out(x, name, pad):
        y = F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=pad)
        y = F.batch_norm(y, sd[name + ".1.running_mean"], sd[name + ".1.running_var"], sd[name + ".1.weight"], sd[name + ".1.bias"], False, 0.0, 1e-5)
        return y * torch.sigmoid(y)

    intra = conv31
    outs = [out(intra, "out0", 0)]
    for k, lat in ((1, conv21), (2, conv11), (3, conv01)):
        intra = F.interpolate(intra, scale_factor=2, mode="bilinear", align_corners=True) + F.conv2d(lat, sd["inner%d.weight" % k], sd["inner%d.bias" % k])
        outs.append(out(intra, "out%d" % k, 1))
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dec = FPNDecoder([8, 16, 32, 64]).eval().to(dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    h, w = a.height // 8, a.width // 8
    feats = [torch.randn(a.views, c, h * 2 ** (3 - i), w * 2 ** (3 - i), generator=g).to(dev) for i, c in enumerate((8, 16, 32, 64))]
    sd = {k: v.detach() for k, v in dec.state_dict().items()}

    def timed(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    ms_hip = timed(lambda: dec(*feats), a.iters)
    with torch.no_grad():
        ms_torch = timed(lambda: torch_decoder(sd, *feats), max(3, a.iters // 4))
    with ops.kernel_timer() as kt:
        for _ in range(a.iters):
            dec(*feats)
    kernels = {}
    for name, s in kt.summary().items():
        wk = kt.work.get(name)
        kernels[name] = {"avg_ms": round(s["avg_ms"], 4), "calls": s["calls"]}
        if wk:
            kernels[name]["tflops"] = round(wk["amount"] / s["calls"] / (s["avg_ms"] * 1e-3) / 1e12, 2)
    flops = sum(kt.work[k]["amount"] / kt.summary()[k]["calls"] for k in kt.work)
    print(json.dumps({"workload": "FPNDecoder eval, %d views %dx%d, fp32" % (a.views, a.height, a.width), "hip_ms": round(ms_hip, 3),
                      "torch_miopen_ms": round(ms_torch, 3), "speedup": round(ms_torch / ms_hip, 2),
                      "algorithmic_tflops": round(flops / (ms_hip * 1e-3) / 1e12, 2), "fp32_mfma_peak_tflops": 157.3, "kernels": kernels}))


if __name__ == "__main__":
    main()
