#!/usr/bin/env python
"""Which torch (ATen) ops still launch kernels of their own in the config-3 training step: one eager step under torch.profiler, ATen ops
that own device time, with the Python line that called them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_train
from torch.profiler import profile, ProfilerActivity
import mvsformer_amd as m
from mvsformer_amd import synth
from mvsformer_amd.losses import ce_loss_stage4
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = m.CascadeMVS(dict(ndepths=[32, 16, 8, 8])).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, fused=True)
feats, proj, dv, scene = synth.make_inputs(5, 512, 640, seed=0, device=dev)
feats = {k: v.requires_grad_(True) for k, v in feats.items()}
gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s, device=dev)[None] for i, s in enumerate(synth.STAGE_SCALES)}
masks = {k: torch.ones_like(v) for k, v in gts.items()}
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    loss = sum(ce_loss_stage4(out, gts, masks, dlossw=[1, 1, 1, 1], inverse_depth=True).values())
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=4)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        st = [x for x in (e.stack or []) if "site-packages" not in x and "dist-packages" not in x]
        rows.append((dt, e.count, e.key, st[0] if st else "(engine / optimizer)"))
for dt, n, name, where in sorted(rows, reverse=True)[:40]:
    print("%-30s x%-3d %7.1f us   %s" % (name, n, dt, where[-120:]))

seen = {}
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add_", "aten::fill_", "aten::sum") and getattr(e, "device_time_total", 0) > 0:
        st = [x for x in (e.stack or []) if "dist-packages" not in x][:2]
        k = (e.name, tuple(st))
        seen[k] = seen.get(k, 0) + 1
for (n, st), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, "x%d" % c, " | ".join(x[-90:] for x in st))
