#!/usr/bin/env python
"""Per-stage and final depth parity of the HIP cascade against the CPU oracle on identical inputs (SURVEY.md §8d
"Reporting": max and p99.9 relative depth error).  Runs the oracle on the host, so keep the size moderate.

    python tools/report_parity.py [--height 256 --width 320 --views 5]

Each stage is compared TWICE: inside the free-running cascade (errors of earlier stages move later hypotheses, arg-max
style choices can flip), and stage by stage with the oracle's own hypotheses fed to the HIP stage (pure kernel error).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m  # noqa: E402
from mvsformer_amd import synth  # noqa: E402
from oracle import ref_torch  # noqa: E402  (the checker)


def stats(got, want):
    rel = ((got - want).abs() / want.abs()).flatten().double()
    return {"max": rel.max().item(), "p99.9": torch.quantile(rel, 0.999).item(), "median": rel.median().item()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--views", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    feats, proj, dv, _ = synth.make_inputs(a.views, a.height, a.width, seed=3)
    tmp = [5.0, 5.0, 5.0, 1.0]
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = ref_torch.cascade_forward(feats, proj, dv, [f.state_dict() for f in net.fusions], ndepths=net.ndepths,
                                        depth_interals_ratio=net.depth_interals_ratio, tmp=tmp)
    net = net.to(dev)
    out = net({k: v.to(dev) for k, v in feats.items()}, {k: v.to(dev) for k, v in proj.items()}, dv.to(dev), tmp=tmp)
    report = {"workload": "%dx%d, %d views, cascade %s, random-init weights, randomized BatchNorm" % (a.width, a.height, a.views, net.ndepths),
              "free_running_cascade": {}, "stage_with_oracle_hypotheses": {}}
    for i in range(4):
        k = "stage%d" % (i + 1)
        report["free_running_cascade"][k] = stats(out[k]["depth"].cpu(), ref[k]["depth"])
        hyp = ref[k]["depth_values"].to(dev)
        with torch.no_grad():
            st = net.fusions[i](feats[k].to(dev), proj[k].to(dev), hyp, tmp=tmp[i])
        report["stage_with_oracle_hypotheses"][k] = stats(st["depth"].cpu(), ref[k]["depth"])
    report["free_running_cascade"]["final"] = stats(out["refined_depth"].cpu(), ref["refined_depth"])
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
