#!/usr/bin/env python
"""LDS-tiled sweeps (cost_volume_tiled.hip) against the direct sweeps (cost_volume.hip) on the GPU: element-wise differences on
ragged and real shapes, LDS fit rate of the tap boxes, and per-stage timings on the hypotheses a real config-2 cascade
produces.  Diagnostic tool (prints a table, writes gpurun_out/check_tiled.json); the pass/fail parity gates live in tests/.

    python tools/check_tiled.py [--no-timing]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def diff_report(name, a, b):
    d = (a.double() - b.double()).abs()
    i = int(d.argmax())
    idx = []
    for s in reversed(a.shape):
        idx.append(i % s)
        i //= s
    return "%s max %.3e mean %.3e at %s (%.6g vs %.6g) finite %s" % (name, d.max().item(), d.mean().item(), tuple(reversed(idx)),
                                                                      a.flatten()[int(d.argmax())].item(), b.flatten()[int(d.argmax())].item(),
                                                                      bool(torch.isfinite(a).all()))


def one_case(dev, C, D, H, W, V, B=1, seed=0, hyp_mode="band", report=None):
    from mvsformer_amd import ops, synth
    scale = {64: 8, 32: 4, 16: 2, 8: 1}[C]
    scene = synth.make_scene(V, H * scale, W * scale, seed=seed)
    feat = synth.render_features(scene, scale, C, batch=B, device=dev).contiguous()
    proj = synth.proj_matrices(scene, (scale,), B, device=dev)["stage1"]
    z = synth.plane_depth(scene, scale, device=dev)
    g = torch.Generator(device="cpu").manual_seed(seed)
    if hyp_mode == "planes":
        hyp = torch.linspace(900.0, 450.0, D, device=dev).view(1, D, 1, 1).expand(B, D, H, W).contiguous()
    elif hyp_mode == "wild":                     # includes points behind the camera / far off the frustum
        hyp = (torch.rand(B, D, H, W, generator=g) * 1500.0 - 200.0).to(dev)
    else:
        jit = 1.0 + 0.01 * torch.randn(B, 1, H, W, generator=g).to(dev)
        hyp = (1.0 / (1.0 / (z[None, None] * jit) + torch.linspace(1, -1, D, device=dev).view(1, D, 1, 1) * (1e-5 * D))).contiguous()
    weight = torch.rand(B, V - 1, H, W, generator=g).to(dev)
    rt = ops.proj_prepare(proj)
    fcl = ops.to_channels_last(feat)
    ent0 = ops.cv_entropy(fcl, rt, hyp, 8)
    vol0, sim0 = ops.cv_aggregate(fcl, rt, hyp, weight, 8, True)
    out = {}
    for exact in (True, False):
        stats = torch.zeros(4, dtype=torch.int32, device=dev)
        ent1 = ops.cv_tiled_entropy(feat, rt, hyp, 8, exact=exact, stats=stats[:2])
        vol1, sim1 = ops.cv_tiled_aggregate(feat, rt, hyp, weight, 8, True, exact=exact, stats=stats[2:])
        vol2, _ = ops.cv_tiled_aggregate(feat, rt, hyp, weight, 8, False, exact=exact)
        torch.cuda.synchronize()
        st = stats.cpu().tolist()
        tag = "exact" if exact else "fast"
        print("  [%s] %s" % (tag, diff_report("entropy", ent1, ent0)))
        print("  [%s] %s" % (tag, diff_report("volume ", vol1, vol0)))
        print("  [%s] sim_depth mismatch %.4f  nosim==sim volume: %s  fit A %d/%d  B %d/%d" % (
            tag, (sim1 != sim0).double().mean().item(), bool(torch.equal(vol1, vol2)), st[0] - st[1], st[0], st[2] - st[3], st[2]))
        out[tag] = {"entropy_max": (ent1 - ent0).abs().max().item(), "volume_max": (vol1 - vol0).abs().max().item(),
                    "sim_mismatch": (sim1 != sim0).double().mean().item(), "rounds_A": st[0], "nofit_A": st[1], "rounds_B": st[2], "nofit_B": st[3]}
    if report is not None:
        report["C%d_D%d_%dx%d_V%d_%s" % (C, D, H, W, V, hyp_mode)] = out


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def timing(dev, report, V=5, H=1152, W=1536):
    import mvsformer_amd as m
    from mvsformer_amd import ops, synth
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, scene = synth.make_inputs(V, H, W, seed=0, device=dev)
    out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    torch.cuda.synchronize()
    rows = []
    for i in range(1, 5):
        f = feats["stage%d" % i].contiguous()
        hyp = out["stage%d" % i]["depth_values"].contiguous()
        zsm = synth.plane_depth(scene, synth.STAGE_SCALES[i - 1], device=dev)
        D = hyp.shape[1]
        lo, hi = hyp.min(1, keepdim=True)[0], hyp.max(1, keepdim=True)[0]
        half = ((1.0 / lo - 1.0 / hi) * 0.5).mean()
        smooth = (1.0 / (1.0 / zsm[None, None] + torch.linspace(-1, 1, D, device=dev).view(1, D, 1, 1) * half)).contiguous()
        rt = ops.proj_prepare(proj["stage%d" % i])
        w = torch.rand(1, V - 1, *f.shape[-2:], device=dev)
        for name, hh in (("cascade", hyp), ("smooth", smooth)):
            fcl = ops.to_channels_last(f)
            stats = torch.zeros(4, dtype=torch.int32, device=dev)
            ops.cv_tiled_entropy(f, rt, hh, 8, stats=stats[:2])
            ops.cv_tiled_aggregate(f, rt, hh, w, 8, True, stats=stats[2:])
            torch.cuda.synchronize()
            st = stats.cpu().tolist()
            r = {"stage": i, "hyp": name,
                 "transpose_ms": timeit(lambda: ops.to_channels_last(f)),
                 "direct_A_ms": timeit(lambda: ops.cv_entropy(fcl, rt, hh, 8)),
                 "direct_B_ms": timeit(lambda: ops.cv_aggregate(fcl, rt, hh, w, 8, True)),
                 "tiled_A_ms": timeit(lambda: ops.cv_tiled_entropy(f, rt, hh, 8)),
                 "tiled_B_ms": timeit(lambda: ops.cv_tiled_aggregate(f, rt, hh, w, 8, True)),
                 "tiled_A_exact_ms": timeit(lambda: ops.cv_tiled_entropy(f, rt, hh, 8, exact=True)),
                 "tiled_B_nosim_ms": timeit(lambda: ops.cv_tiled_aggregate(f, rt, hh, w, 8, False)),
                 "nofit_A": st[1] / max(1, st[0]), "nofit_B": st[3] / max(1, st[2])}
            rows.append(r)
            print("stage%d %-7s transpose %.3f | direct A %.3f B %.3f | tiled A %.3f (exact %.3f) B %.3f (nosim %.3f) | nofit A %.3f B %.3f" % (
                i, name, r["transpose_ms"], r["direct_A_ms"], r["direct_B_ms"], r["tiled_A_ms"], r["tiled_A_exact_ms"], r["tiled_B_ms"],
                r["tiled_B_nosim_ms"], r["nofit_A"], r["nofit_B"]))
    report["timing"] = rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--out", default="gpurun_out/check_tiled.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    report = {}
    cases = [(8, 4, 40, 64, 3, 1, "band"), (8, 2, 5, 64, 2, 1, "band"), (8, 4, 37, 70, 4, 2, "band"), (8, 8, 48, 64, 3, 1, "wild"),
             (16, 8, 32, 48, 3, 1, "band"), (16, 3, 8, 130, 4, 1, "band"), (16, 8, 40, 56, 5, 2, "wild"),
             (32, 16, 24, 32, 3, 1, "band"), (32, 5, 17, 33, 2, 1, "planes"), (32, 16, 36, 48, 5, 1, "wild"),
             (64, 32, 18, 24, 3, 1, "planes"), (64, 6, 9, 70, 3, 2, "band"), (64, 48, 16, 20, 4, 1, "planes"), (64, 32, 16, 24, 5, 1, "wild"),
             (8, 4, 288, 384, 5, 1, "band"), (64, 32, 144, 192, 5, 1, "planes")]
    for C, D, H, W, V, B, mode in cases:
        print("case C=%d D=%d %dx%d V=%d B=%d %s" % (C, D, H, W, V, B, mode))
        try:
            one_case(dev, C, D, H, W, V, B, seed=C + D + W, hyp_mode=mode, report=report)
        except Exception as e:                  # keep going: one launch failure must not hide the other cases
            print("  FAILED: %r" % (e,))
            report["C%d_D%d_%dx%d_V%d_%s" % (C, D, H, W, V, mode)] = {"error": repr(e)}
    if not args.no_timing:
        try:
            timing(dev, report)
        except Exception as e:
            print("timing FAILED: %r" % (e,))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
