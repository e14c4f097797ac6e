#!/usr/bin/env python
"""Per-kernel time of one training step (fwd + bwd, config-3 geometry) from HIP events around every C-ABI launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m
from mvsformer_amd import ops, synth
from mvsformer_amd.losses import ce_loss_stage4
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = m.CascadeMVS(dict(ndepths=[32, 16, 8, 8])).to(dev).train()
feats, proj, dv, scene = synth.make_inputs(5, 512, 640, seed=0, device=dev)
feats = {k: v.requires_grad_(True) for k, v in feats.items()}
gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s, device=dev)[None] for i, s in enumerate(synth.STAGE_SCALES)}
masks = {k: torch.ones_like(v) for k, v in gts.items()}
import contextlib
bf16 = "--bf16" in sys.argv
amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if bf16 else contextlib.nullcontext
def step():
    with amp():
        out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    loss = sum(ce_loss_stage4(out, gts, masks, dlossw=[1, 1, 1, 1]).values())
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
with ops.kernel_timer() as kt:
    step()
tot = 0.0
rows = sorted(kt.summary().items(), key=lambda kv: -kv[1]["total_ms"])
for k, v in rows[:40]:
    print("%-36s calls %3d  total %7.3f ms" % (k, v["calls"], v["total_ms"]))
print("SUM %.3f ms over %d launches" % (sum(v["total_ms"] for _, v in rows), sum(v["calls"] for _, v in rows)))
if "--timeline" in sys.argv:            # every C-ABI launch of the step in order (median of 5 steps), with a running sum
    with ops.kernel_timer() as kt:
        for _ in range(5): step()
    summ = kt.summary()
    n = len(kt.order) // 5
    seen, t = {}, 0.0
    for tag in kt.order[:n]:
        j = seen.get(tag, 0); seen[tag] = j + 1
        per = KT_MED = ops.KernelTimer.per_step_medians(summ[tag]["all_ms"], 5)
        ms = per[j] if per else summ[tag]["all_ms"][j]
        t += ms
        print("%4d %-36s %8.1f us   cum %7.3f ms" % (sum(seen.values()), tag, ms * 1e3, t))
