#!/usr/bin/env python
"""Per-parameter gradient errors of one StageNet training step against torch autograd through the CPU oracle (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m  # noqa: E402
from mvsformer_amd import synth  # noqa: E402
from oracle import ref_torch  # noqa: E402

dev = torch.device("cuda:0")
C, nd, H, W, V, full = (int(a) for a in sys.argv[1:7]) if len(sys.argv) > 6 else (64, 32, 32, 40, 5, 256)
scale = {64: 8, 32: 4, 16: 2, 8: 1}[C]
torch.manual_seed(5)
net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), nd, 0).train()
scene = synth.make_scene(V, H * scale, W * scale, seed=6)
feat = synth.render_features(scene, scale, C)
proj = synth.proj_matrices(scene, (scale,))["stage1"]
hyp = ref_torch.init_inverse_range(synth.depth_range(1), nd, H, W)
R = torch.randn(1, nd, H, W, generator=torch.Generator().manual_seed(0))
fr = feat.clone().requires_grad_(True)
sd = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.detach().clone())
      for k, v in net.state_dict().items()}
taps = {}
want = ref_torch.stage_forward(fr, proj, hyp, sd, ndepth=nd, tmp=5.0, training=True, taps=taps)
(want["prob_volume_pre"] * R).sum().backward()
net = net.to(dev)
if os.environ.get("MVS_POISON"):
    # every later allocation is carved out of this freed NaN-filled block: a kernel that reads memory it was supposed to have
    # written (or zeroed) first turns its output into NaN
    poison = torch.full((1 << 29,), float(os.environ["MVS_POISON"]), device=dev)
    del poison
fg = feat.to(dev).requires_grad_(True)
got = net(fg, proj.to(dev), hyp.to(dev), tmp=5.0)
(got["prob_volume_pre"] * R.to(dev)).sum().backward()
print("pre   max err / scale %.3e" % ((got["prob_volume_pre"].detach().cpu() - want["prob_volume_pre"].detach()).abs().max() / want["prob_volume_pre"].abs().max()).item())
e = (fg.grad.cpu() - fr.grad).abs()
print("dfeat max %.3e mean %.3e (rel to max)" % (e.max().item() / fr.grad.abs().max().item(), e.mean().item() / fr.grad.abs().max().item()))
for name, p in net.named_parameters():
    w = sd[name].grad
    print("%-32s max-rel %.3e  l2-rel %.3e  |w|max %.3e" % (name, (p.grad.cpu() - w).abs().max().item() / max(w.abs().max().item(), 1e-12),
                                                           ((p.grad.cpu() - w).norm() / (w.norm() + 1e-30)).item(), w.abs().max().item()))
