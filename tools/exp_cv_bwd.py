#!/usr/bin/env python
"""Backward of the cost-volume aggregation on config-3 stage shapes: direct-atomics kernel vs the LDS-window kernel over a few window
sizes, with the hypotheses the (random-weight) training cascade predicts and with a smooth band around the true plane; prints time and
the fraction of taps that miss the window.   python tools/exp_cv_bwd.py  (GPU box)"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CSRC = os.path.join(REPO, "mvsformer_amd", "csrc")
if "--build" in sys.argv:       # here (no GPU): full libraries with one experiment switch each -> csrc/exp/libmvs_hip_exp<N>.so
    os.makedirs(os.path.join(CSRC, "exp"), exist_ok=True)
    objs = [os.path.join(CSRC, o) for o in os.listdir(CSRC) if o.endswith(".o") and o != "cost_volume_bwd.o"]
    for n in (1, 3, 15, 16, 32):
        obj = os.path.join(CSRC, "exp", "cvb_%d.o" % n)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                               "-DMVS_BWD_EXP=%d" % n, "-c", os.path.join(CSRC, "cost_volume_bwd.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs +
                              ["-o", os.path.join(CSRC, "exp", "libmvs_hip_exp%d.so" % n)])
    sys.exit(0)
import torch
from mvsformer_amd import _lib
if os.environ.get("MVS_EXP_LIB"):
    _lib.LIB_PATH = os.path.join(CSRC, "exp", "libmvs_hip_exp%s.so" % os.environ["MVS_EXP_LIB"])
import mvsformer_amd as m
from mvsformer_amd import ops, synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = m.CascadeMVS(dict(ndepths=[32, 16, 8, 8])).to(dev).train()
feats, proj, dv, scene = synth.make_inputs(5, 512, 640, seed=0, device=dev)
with torch.no_grad():
    out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


WINDOWS = ("6,24",) if os.environ.get("MVS_EXP_LIB") else ("6,24", "6,16", "5,24", "5,16", "7,12", "4,8")
for i in (1, 2, 3, 4):
    f = ops.to_channels_last(feats["stage%d" % i].contiguous())
    B, V, H, W, C = f.shape
    hyp_c = out["stage%d" % i]["depth_values"].contiguous()
    D = hyp_c.shape[1]
    z = synth.plane_depth(scene, synth.STAGE_SCALES[i - 1], device=dev)
    half = ((1.0 / hyp_c.min(1)[0] - 1.0 / hyp_c.max(1)[0]) * 0.5).mean()
    hyp_s = (1.0 / (1.0 / z[None, None] + torch.linspace(-1, 1, D, device=dev).view(1, D, 1, 1) * half)).contiguous()
    rt = ops.proj_prepare(proj["stage%d" % i])
    w = torch.rand(1, V - 1, H, W, device=dev)
    for hname, hyp in (("smooth", hyp_s), ("cascade", hyp_c)):
        vol, _ = ops.cv_aggregate(f, rt, hyp, w, 8, False, exact=True)
        g = torch.randn_like(vol)
        os.environ["MVS_CV_BWD"] = "direct"
        t0 = timeit(lambda: ops.cv_aggregate_bwd(f, rt, hyp, w, vol, g, 8))
        fwd = timeit(lambda: ops.cv_aggregate(f, rt, hyp, w, 8, False, exact=True))
        line = "stage%d C=%d D=%d %dx%d %-7s fwd %.3f direct %.3f |" % (i, C, D, H, W, hname, fwd, t0)
        for mode, win in [("lds", w_) for w_ in WINDOWS] + ([] if os.environ.get("MVS_EXP_LIB") else [("own", w_) for w_ in ("5,12", "5,16", "5,20")]):
            os.environ["MVS_CV_BWD"] = mode
            os.environ["MVS_CV_BWD_WINDOW"] = win
            st = torch.zeros(2, dtype=torch.int32, device=dev)
            ops.cv_aggregate_bwd(f, rt, hyp, w, vol, g, 8, stats=st)
            s = st.cpu().tolist()
            t1 = timeit(lambda: ops.cv_aggregate_bwd(f, rt, hyp, w, vol, g, 8))
            line += " %s %s: %.3f ms miss %.3f |" % (mode, win, t1, s[1] / max(1, s[0]))
        print(line, flush=True)
