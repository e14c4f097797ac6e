#!/usr/bin/env python
"""Run-to-run drift of one training step per tensor: which gradients move when only the atomics' order changes."""
import contextlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m
from mvsformer_amd import synth
from mvsformer_amd.losses import ce_loss_stage4
dev = torch.device("cuda:0")
bf16 = "--bf16" in sys.argv
torch.manual_seed(0)
net = m.CascadeMVS(dict(ndepths=[32, 16, 8, 8])).to(dev).train()
feats, proj, dv, scene = synth.make_inputs(3, 256, 320, seed=2, device=dev)
feats = {k: v.requires_grad_(True) for k, v in feats.items()}
gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s, device=dev)[None] for i, s in enumerate(synth.STAGE_SCALES)}
masks = {k: torch.ones_like(v) for k, v in gts.items()}
state = {k: v.clone() for k, v in net.state_dict().items()}
amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if bf16 else contextlib.nullcontext
runs = []
for _ in range(3):
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    for f in feats.values():
        f.grad = None
    with amp():
        out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    loss = sum(ce_loss_stage4(out, gts, masks, dlossw=[1, 1, 1, 1]).values())
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    grads.update({"feat_" + k: f.grad.detach().clone() for k, f in feats.items()})
    runs.append(grads)
rows = []
for n, g in runs[0].items():
    d = max(((r[n] - g).norm() / g.norm().clamp_min(1e-30)).item() for r in runs[1:])
    rows.append((d, n, g.norm().item(), tuple(g.shape)))
rows.sort(reverse=True)
for d, n, nrm, shp in rows[:25]:
    print("%-50s rel drift %.3e  |g| %.3e  %s" % (n, d, nrm, shp))
print("median drift %.3e over %d tensors" % (sorted(r[0] for r in rows)[len(rows) // 2], len(rows)))
