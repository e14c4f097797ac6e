"""Like ms_stage_check.py but the stage is run through StageNet.forward (intermediates freed and their memory reused): which stage alone
is not reproducible under 3 streams, and with which kernel switches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m
from mvsformer_amd import ops, synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = m.CascadeMVS().eval(); m.randomize_bn_(net, seed=1); net = net.to(dev)
feats, proj, dv, scene = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
tmp = [5.0, 5.0, 5.0, 1.0]
full = net(feats, proj, dv, tmp=tmp); torch.cuda.synchronize()
which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
def run_stage(i):
    st = net.fusions[i - 1]
    out = st(feats["stage%d" % i], proj["stage%d" % i], full["stage%d" % i]["depth_values"].contiguous(), tmp=tmp)
    return {k: out[k] for k in ("depth", "prob_volume_pre", "sim_depth")}
refs = {i: run_stage(i) for i in which}
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
runs = []
for rep in range(4):
    for k, i in enumerate(which):
        with torch.cuda.stream(streams[(k + rep) % 3]):
            runs.append((i, run_stage(i)))
torch.cuda.synchronize()
for i, t in runs:
    bad = [(k, int((v != refs[i][k]).sum())) for k, v in t.items() if not torch.equal(v, refs[i][k])]
    print("stage", i, "differing:", bad)
if os.environ.get("MS_PATTERN"):
    for i, t in runs:
        d = (t["prob_volume_pre"] != refs[i]["prob_volume_pre"])
        if d.any():
            idx = d.nonzero()
            print("stage", i, "n", idx.shape[0], "of", d.numel(), "depth hist", torch.bincount(idx[:, 1], minlength=d.shape[1]).tolist())
            ys, xs = idx[:, 2], idx[:, 3]
            print("   y range", ys.min().item(), ys.max().item(), "x range", xs.min().item(), xs.max().item())
            # coarse occupancy map on a 64-pixel grid
            occ = torch.zeros(d.shape[2] // 64 + 1, d.shape[3] // 64 + 1, dtype=torch.int32)
            occ.index_put_(((ys // 64).cpu(), (xs // 64).cpu()), torch.ones(idx.shape[0], dtype=torch.int32), accumulate=True)
            nz = occ.nonzero()
            print("   64x64 cells touched:", nz.shape[0], "of", occ.numel(), "first cells", nz[:12].tolist())
            err = (t["prob_volume_pre"] - refs[i]["prob_volume_pre"]).abs()
            print("   max abs diff", err.max().item(), "median of differing", err[d].median().item())
            break
