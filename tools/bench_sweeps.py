#!/usr/bin/env python
"""Per-stage timings of the cost-volume sweeps at config-2 shapes with the hypotheses the random-weight cascade predicts ("cascade") and a
smooth band around the true surface ("smooth"): recomputing pair (sweep A + sweep B), stored-correlation pair (A' + B'), transposes.

    python tools/bench_sweeps.py [--stages 1 2 3 4] [--exact]     -> gpurun_out/bench_sweeps.txt
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--hyps", nargs="+", default=["cascade", "smooth"])
    args = ap.parse_args()
    import torch
    import mvsformer_amd as m
    from mvsformer_amd import ops, synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, scene = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
    out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    torch.cuda.synchronize()

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.iters

    lines = []
    tot = {}
    for i in args.stages:
        f = feats["stage%d" % i].contiguous()
        B, V, C, H, W = f.shape
        hyp_c = out["stage%d" % i]["depth_values"].contiguous()
        D = hyp_c.shape[1]
        z = synth.plane_depth(scene, synth.STAGE_SCALES[i - 1], device=dev)
        half = ((1.0 / hyp_c.min(1)[0] - 1.0 / hyp_c.max(1)[0]) * 0.5).mean()
        hyp_s = (1.0 / (1.0 / z[None, None] + torch.linspace(-1, 1, D, device=dev).view(1, D, 1, 1) * half)).contiguous()
        rt = ops.proj_prepare(proj["stage%d" % i])
        fcl = ops.to_channels_last(f)
        w = torch.rand(B, V - 1, H, W, device=dev)
        t_tr = timeit(lambda: ops.to_channels_last(f))
        for hname in args.hyps:
            hyp = hyp_c if hname == "cascade" else hyp_s
            for exact in (False, True):
                r = {"A": timeit(lambda: ops.cv_entropy(fcl, rt, hyp, 8, exact=exact)),
                     "B": timeit(lambda: ops.cv_aggregate(fcl, rt, hyp, w, 8, True, exact=exact)),
                     "B(no sim_depth)": timeit(lambda: ops.cv_aggregate(fcl, rt, hyp, w, 8, False, exact=exact))}
                if ops.cv_store_bytes(fcl, D, 8) > 0:
                    ent, store = ops.cv_corr(fcl, rt, hyp, 8, exact=exact)
                    r["A'"] = timeit(lambda: ops.cv_corr(fcl, rt, hyp, 8, exact=exact))
                    r["B'"] = timeit(lambda: ops.cv_merge(store, hyp, w, V, C, 8, True))
                line = "stage%d %-7s %-5s transpose %.3f | " % (i, hname, "exact" if exact else "fast", t_tr) + "  ".join("%s %.3f" % kv for kv in r.items())
                best = min(r["A"] + r["B"], r.get("A'", 9) + r.get("B'", 9))
                r.pop("B(no sim_depth)") if False else None
                line += "  | best pair %.3f ms" % best
                tot[(hname, exact)] = tot.get((hname, exact), 0.0) + best + t_tr
                print(line, flush=True)
                lines.append(line)
    for k, v in tot.items():
        line = "sum over stages %s (%s, %s): sweeps + transposes %.3f ms -> %.4f of the HBM roofline on 1008.6 MB" % (
            args.stages, k[0], "exact" if k[1] else "fast", v, 1008.6e6 / (v * 1e-3) / 8e12)
        print(line)
        lines.append(line)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "bench_sweeps.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
