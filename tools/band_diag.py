import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mvsformer_amd as m
from mvsformer_amd import ops, synth, stagenet
dev = torch.device("cuda:0")
C, D, H, W, V, B = 32, 16, 40, 56, 5, 1
scale = 4
scene = synth.make_scene(V, H * scale, W * scale, seed=C + H)
feat = synth.render_features(scene, scale, C, batch=B, device=dev).contiguous()
proj = synth.proj_matrices(scene, (scale,), B, device=dev)["stage1"]
z = synth.plane_depth(scene, scale, device=dev)
hyp = (1.0 / (1.0 / z[None, None] + torch.linspace(1, -1, D, device=dev).view(1, D, 1, 1) * (1e-5 * D))).repeat(B, 1, 1, 1).contiguous()
torch.manual_seed(C + H)
net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), D, 0).eval()
m.randomize_bn_(net, 3)
net = net.to(dev)
rt = ops.proj_prepare(proj)
fcl = ops.to_channels_last(feat)
vp, vprep = net._vis_params()
e0, st0 = ops.cv_corr(fcl, rt, hyp, 8)
w0 = net._vis_weight(e0, vp, vprep)
v0, s0 = ops.cv_merge(st0, hyp, w0, V, C, 8, True)
vol = torch.zeros_like(v0); sim = torch.zeros_like(s0)
for r0, r1 in ((0, 20), (20, 40)):
    y0, y1 = max(0, r0 - 3), min(H, r1 + 3)
    e, st = ops.cv_corr_rows(fcl, rt, hyp, 8, y0, y1 - y0)
    print("band", r0, "entropy equal", torch.equal(e, e0[:, :, y0:y1]))
    w = net._vis_weight(e, vp, vprep)
    dw = (w[:, :, r0 - y0:r1 - y0] - w0[:, :, r0:r1]).abs()
    print("  vis weight band rows: max diff", dw.max().item(), "rows with diff", dw.amax((0, 1, 3)).nonzero().flatten().tolist())
    w2 = net._vis_weight(e0[:, :, y0:y1].contiguous(), vp, vprep)
    print("  vis on sliced full entropy equal to vis on band entropy", torch.equal(w, w2))
    for mode in ("valu", "wino"):
        pass
    ops.cv_merge_rows(st, hyp, w0[:, :, y0:y1].contiguous(), V, C, 8, y0, r0 - y0, r1 - r0, vol, sim)
print("volume (full weights) equal", torch.equal(vol, v0), "sim equal", torch.equal(sim, s0))
# the all-VALU vis on the same band
wv0 = ops.vis(e0, vp)
for r0, r1 in ((0, 20), (20, 40)):
    y0, y1 = max(0, r0 - 3), min(H, r1 + 3)
    wv = ops.vis(e0[:, :, y0:y1].contiguous(), vp)
    print("valu vis band", r0, "max diff", (wv[:, :, r0 - y0:r1 - y0] - wv0[:, :, r0:r1]).abs().max().item())
