import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m
from mvsformer_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
D, H, W = 4, 1152, 1536
net = m.CostRegNet3D(8, 8).eval().to(dev)
vol = torch.randn(1, 8, D, H, W, device=dev)
def run(x):
    outs = {}
    outs["c1"] = net.conv1(x)
    if os.environ.get("RACE_CLONE"):
        outs["c1"] = outs["c1"].clone()
    outs["c2"] = net.conv2(outs["c1"]); outs["c3"] = net.conv3(outs["c2"]); outs["c4"] = net.conv4(outs["c3"])
    outs["c5"] = net.conv5(outs["c4"]); outs["c6"] = net.conv6(outs["c5"])
    return outs
ref = run(vol)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
res = []
for i in range(6):
    with torch.cuda.stream(streams[i % 3]):
        res.append(run(vol))
torch.cuda.synchronize()
for i, r in enumerate(res):
    print("run", i, {k: (0 if torch.equal(v, ref[k]) else ((v - ref[k]).abs().max().item(), int((v != ref[k]).sum()))) for k, v in r.items()})
# where are the differing elements of the first differing layer?
for i, r in enumerate(res):
    for k in ("c1", "c2", "c3", "c4", "c5", "c6"):
        if not torch.equal(r[k], ref[k]):
            idx = (r[k] != ref[k]).nonzero()
            print("run", i, "first bad layer", k, "shape", tuple(r[k].shape), "n", idx.shape[0], "min idx", idx.min(0)[0].tolist(), "max idx", idx.max(0)[0].tolist())
            print("   sample", idx[:5].tolist())
            break
