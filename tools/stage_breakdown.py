#!/usr/bin/env python
"""Per-stage, per-kernel time of one config-2 depth map (HIP events around every C-ABI launch, single stream): which launches the
4.7 ms are made of, stage by stage.     python tools/stage_breakdown.py  -> gpurun_out/stage_breakdown.txt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import mvsformer_amd as m  # noqa: E402
from mvsformer_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = m.CascadeMVS().eval()
m.randomize_bn_(net, seed=1)
net = net.to(dev)
feats, proj, dv, scene = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
tmp = [5.0, 5.0, 5.0, 1.0]
out = net(feats, proj, dv, tmp=tmp)
torch.cuda.synchronize()
lines = []
grand = 0.0
REPS = 10
for i in range(1, 5):
    f = feats["stage%d" % i]
    p = proj["stage%d" % i]
    hyp = out["stage%d" % i]["depth_values"].contiguous()
    stage = net.fusions[i - 1]
    for _ in range(2):
        stage(f, p, hyp, tmp=tmp)
    with ops.kernel_timer() as kt:
        for _ in range(REPS):
            stage(f, p, hyp, tmp=tmp)
    summ = kt.summary()
    tot = sum(v["total_ms"] for v in summ.values()) / REPS
    grand += tot
    lines.append("stage %d: %.3f ms in %d launches" % (i, tot, sum(v["calls"] for v in summ.values()) // REPS))
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
        w = kt.work.get(k)
        extra = ""
        if w:
            amt = w["amount"] / v["calls"]
            extra = ("  %7.1f TFLOP/s" % (amt / v["avg_ms"] / 1e9)) if w["kind"] == "flops" else ("  %7.1f GB/s" % (amt / v["avg_ms"] / 1e6))
        lines.append("   %-34s x%d  %8.4f ms each  %8.4f ms%s" % (k, v["calls"] // REPS, v["avg_ms"], v["total_ms"] / REPS, extra))
lines.append("all stages: %.3f ms of kernels per depth map (scheduler / confidence launches between the stages not included)" % grand)
print("\n".join(lines))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "stage_breakdown.txt"), "w").write("\n".join(lines) + "\n")
