"""Which kernel's output moves when three streams run the same stage concurrently: every intermediate of StageNet.forward is kept and
compared with a single-stream run (bit for bit).  python tools/ms_stage_check.py [stage]"""
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
    f, p, hyp = feats["stage%d" % i], proj["stage%d" % i], full["stage%d" % i]["depth_values"].contiguous()
    t = {}
    rt = ops.proj_prepare(p.contiguous())
    vp, vprep = st._vis_params()
    fcl = ops.to_channels_last(f)
    t["fcl"] = fcl
    from mvsformer_amd.stagenet import _store_plan
    if _store_plan(fcl, hyp.shape[1], 8):
        t["entropy"], store = ops.cv_corr(fcl, rt, hyp, 8)
        t["weight"] = st._vis_weight(t["entropy"], vp, vprep)
        t["volume"], t["sim"] = ops.cv_merge(store, hyp, t["weight"], fcl.shape[1], fcl.shape[4], 8, True)
    else:
        t["entropy"] = ops.cv_entropy(fcl, rt, hyp, 8)
        t["weight"] = st._vis_weight(t["entropy"], vp, vprep)
        t["volume"], t["sim"] = ops.cv_aggregate(fcl, rt, hyp, t["weight"], 8, True)
    cr = st.cost_reg
    x = t["volume"]
    t["c1"] = cr.conv1(x); t["c2"] = cr.conv2(t["c1"]); t["c3"] = cr.conv3(t["c2"]); t["c4"] = cr.conv4(t["c3"])
    t["c5"] = cr.conv5(t["c4"]); t["c6"] = cr.conv6(t["c5"])
    if isinstance(cr, m.CostRegNet3D):
        t["c7"] = cr._up("conv7", t["c6"], t["c4"]); t["c9"] = cr._up("conv9", t["c7"], t["c2"])
        t["logits"] = cr.logits(x)
    else:
        t["c7"] = cr.conv7(t["c6"], residual=t["c4"]); t["c9"] = cr.conv9(t["c7"], residual=t["c2"]); t["c11"] = cr.conv11(t["c9"], residual=x)
        t["logits"] = ops.prob3(t["c11"], cr.prob.weight.detach().float().contiguous())
    pre, prob, depth, conf = ops.head(hyp, 5.0, False, logits=t["logits"])
    t["depth"] = depth
    return t

if os.environ.get("MS_FREE"):
    _orig = run_stage
    class Taps(dict):
        """keeps a CLONE of every intermediate; the working tensor itself is dropped as soon as the stage no longer needs it"""
    def run_stage(i):                                    # noqa: F811
        st = net.fusions[i - 1]
        f, p, hyp = feats["stage%d" % i], proj["stage%d" % i], full["stage%d" % i]["depth_values"].contiguous()
        keep = {}
        rt = ops.proj_prepare(p.contiguous())
        vp, vprep = st._vis_params()
        fcl = ops.to_channels_last(f)
        ent = ops.cv_entropy(fcl, rt, hyp, 8); keep["entropy"] = ent.clone()
        w = st._vis_weight(ent, vp, vprep); keep["weight"] = w.clone(); del ent
        vol, sim = ops.cv_aggregate(fcl, rt, hyp, w, 8, True); keep["volume"] = vol.clone(); del w, fcl, sim
        cr = st.cost_reg
        c1 = cr.conv1(vol); keep["c1"] = c1.clone()
        c2 = cr.conv2(c1); keep["c2"] = c2.clone(); del c1
        c3 = cr.conv3(c2); keep["c3"] = c3.clone()
        c4 = cr.conv4(c3); keep["c4"] = c4.clone(); del c3
        c5 = cr.conv5(c4); keep["c5"] = c5.clone()
        c6 = cr.conv6(c5); keep["c6"] = c6.clone(); del c5
        c7 = cr._up("conv7", c6, c4); keep["c7"] = c7.clone(); del c6, c4
        c9 = cr._up("conv9", c7, c2); keep["c9"] = c9.clone(); del c7, c2
        lg = cr.logits(vol) if os.environ.get("MS_FREE") == "logits" else ops.deconv3d_prob1(c9, cr._dcache["conv11"][1], 16, cr._dcache["conv11"][2], cr._dcache["conv11"][3], vol, *cr.prob_params(), relu=True)
        keep["logits"] = lg.clone(); del c9, vol
        return keep

refs = {i: run_stage(i) for i in which}
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
runs = []
for rep in range(3):
    for k, i in enumerate(which):
        with torch.cuda.stream(streams[(k + rep) % 3]):
            runs.append((i, run_stage(i)))
torch.cuda.synchronize()
for i, t in runs:
    bad = [(k, int((v != refs[i][k]).sum())) for k, v in t.items() if not torch.equal(v, refs[i][k])]
    print("stage", i, "first differing:", bad[:3])
