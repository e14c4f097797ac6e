import os, sys, torch
sys.path.insert(0, "/root/repo")
import mvsformer_amd as m
from mvsformer_amd import synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = m.CascadeMVS().eval(); m.randomize_bn_(net, seed=1); net = net.to(dev)
feats, proj, dv, scene = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
tmp = [5.0, 5.0, 5.0, 1.0]
a = net(feats, proj, dv, tmp=tmp); torch.cuda.synchronize()
a2 = net(feats, proj, dv, tmp=tmp); torch.cuda.synchronize()
print("single-stream repeat equal:", torch.equal(a["refined_depth"], a2["refined_depth"]))
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
outs = []
for i in range(6):
    with torch.cuda.stream(streams[i % 3]):
        outs.append(net(feats, proj, dv, tmp=tmp))
torch.cuda.synchronize()
for j, o in enumerate(outs):
    bad = []
    for k in ("stage1", "stage2", "stage3", "stage4"):
        for key in ("depth", "prob_volume_pre", "sim_depth"):
            if not torch.equal(o[k][key], a[k][key]):
                bad.append((k, key, (o[k][key] - a[k][key]).abs().max().item()))
    print("run", j, "mismatches:", bad[:4])
