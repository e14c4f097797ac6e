#!/usr/bin/env python
"""Times the FPN encoder (csrc/conv2d.hip) and decoder (csrc/fpn.hip) on BASELINE configs[1]'s geometry (5 views, 1152x1536) with HIP events per launch, next to
the same ops in plain PyTorch (MIOpen convolutions + ATen upsampling) on the same GPU.  Prints one JSON object.

    python tools/bench_fpn.py [--views 5] [--height 1152] [--width 1536] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import FPNDecoder, FPNEncoder, ops  # noqa: E402


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


def torch_encoder(sd, x):
    """models/module.py:226-240 in plain torch - the comparison leg."""
    import torch.nn.functional as F
    outs = {}
    for name, k, s in FPNEncoder.LAYERS:
        x = F.conv2d(x, sd[name + ".conv.weight"], None, stride=s, padding=k // 2)
        x = F.batch_norm(x, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"], sd[name + ".bn.weight"], sd[name + ".bn.bias"], False, 0.0, 1e-5)
        x = F.leaky_relu(x, 0.1)
        outs[name] = x
    return [outs["conv01"], outs["conv11"], outs["conv21"], outs["conv31"]]


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

    enc = FPNEncoder([8, 16, 32, 64]).eval().to(dev)
    esd = {k: v.detach() for k, v in enc.state_dict().items()}
    img = torch.randn(a.views, 3, a.height, a.width, generator=g).to(dev)
    ms_enc = timed(lambda: enc(img), a.iters)
    with torch.no_grad():
        ms_enc_torch = timed(lambda: torch_encoder(esd, img), max(3, a.iters // 4))
    ms_hip = timed(lambda: dec(*feats), a.iters)
    with torch.no_grad():
        ms_torch = timed(lambda: torch_decoder(sd, *feats), max(3, a.iters // 4))
    with ops.kernel_timer() as kt:
        for _ in range(a.iters):
            enc(img)
            dec(*feats)
    kernels = {}
    for name, s in kt.summary().items():
        wk = kt.work.get(name)
        kernels[name] = {"avg_ms": round(s["avg_ms"], 4), "calls": s["calls"]}
        if wk:
            kernels[name]["tflops"] = round(wk["amount"] / s["calls"] / (s["avg_ms"] * 1e-3) / 1e12, 2)
    summ = kt.summary()
    fl = {True: 0.0, False: 0.0}
    for k in kt.work:
        fl[k.startswith("conv2d")] += kt.work[k]["amount"] / a.iters      # per forward (two layers share some kernel names)
    print(json.dumps({"workload": "FPNEncoder + FPNDecoder eval, %d views %dx%d, fp32" % (a.views, a.height, a.width),
                      "decoder": {"hip_ms": round(ms_hip, 3), "torch_miopen_ms": round(ms_torch, 3), "speedup": round(ms_torch / ms_hip, 2),
                                  "algorithmic_tflops": round(fl[False] / (ms_hip * 1e-3) / 1e12, 2)},
                      "encoder": {"hip_ms": round(ms_enc, 3), "torch_miopen_ms": round(ms_enc_torch, 3), "speedup": round(ms_enc_torch / ms_enc, 2),
                                  "algorithmic_tflops": round(fl[True] / (ms_enc * 1e-3) / 1e12, 2)},
                      "fp32_mfma_peak_tflops": 157.3, "kernels": kernels}))


if __name__ == "__main__":
    main()
