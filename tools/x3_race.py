import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m
from mvsformer_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [(8, 16, (1, 2), 8, 576, 768), (16, 32, (1, 2), 8, 288, 384), (32, 64, (1, 2), 8, 144, 192), (8, 16, (1, 2), 4, 1152, 1536), (16, 32, (1, 2), 4, 576, 768)]
data = []
for (cin, cout, stride, d, h, w) in cases:
    x = torch.randn(1, cin, d, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    px = ops.conv3d_x3_pack(wt, stride)
    ref = ops.conv3d_x3(x, px, cin, cout, stride)
    data.append((x, px, ref))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
outs = []
for rep in range(4):
    for i, ((cin, cout, stride, d, h, w), (x, px, ref)) in enumerate(zip(cases, data)):
        with torch.cuda.stream(streams[(i + rep) % 3]):
            outs.append((i, ops.conv3d_x3(x, px, cin, cout, stride)))
torch.cuda.synchronize()
for i, o in outs:
    ref = data[i][2]
    if not torch.equal(o, ref):
        print("case", cases[i], "MISMATCH", (o - ref).abs().max().item(), int((o != ref).sum()))
print("mixed-shape concurrency done")
# the regularizer as a whole
for D, H, W in ((8, 576, 768), (4, 1152, 1536)):
    net = m.CostRegNet3D(8, 8).eval().to(dev)
    vol = torch.randn(1, 8, D, H, W, device=dev)
    ref = net.logits(vol)
    torch.cuda.synchronize()
    outs = []
    for i in range(6):
        with torch.cuda.stream(streams[i % 3]):
            outs.append(net.logits(vol))
    torch.cuda.synchronize()
    print("CostRegNet3D", D, H, W, "mismatching runs:", [(i, (o - ref).abs().max().item()) for i, o in enumerate(outs) if not torch.equal(o, ref)])
