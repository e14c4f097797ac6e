#!/usr/bin/env python
"""Tile-shape / LDS-capacity / occupancy scan of the LDS-tiled sweeps: builds cost_volume_tiled.hip several times with different
TileCfg macros into mvsformer_amd/csrc/exp/libtiled_<name>.so (--build, no GPU needed) and times sweep A / sweep B of each
variant on config-2 stage shapes with smooth and cascade-predicted hypotheses (on the GPU box).

    python tools/exp_tiled.py --build            # here
    python tools/exp_tiled.py --stages 3 4       # on the GPU box -> gpurun_out/exp_tiled.json
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CSRC = os.path.join(REPO, "mvsformer_amd", "csrc")
EXP = os.path.join(CSRC, "exp")

# name -> (MVS_T8, MVS_T16, MVS_T32, MVS_T64): TW, TH, S, DCL, CC, CAP, OCC
VARIANTS = {
    "v0_default": ("16,16,1,4,8,1536,3", "16,16,1,4,16,768,3", "16,4,4,1,16,768,3", "16,4,4,1,16,768,3"),
    "v1_small_lds": ("16,16,1,4,8,768,6", "16,16,1,4,16,512,5", "16,4,4,1,16,512,5", "16,4,4,1,16,512,5"),
    "v2_mid_lds": ("16,16,1,4,8,1024,5", "16,16,1,4,16,640,4", "16,4,4,1,16,640,4", "16,4,4,1,16,640,4"),
    "v3_wide_tile": ("32,8,1,4,8,1536,3", "32,8,1,4,16,768,3", "16,8,2,1,16,768,3", "16,8,2,1,16,768,3"),
    "v4_128px_2slots": ("16,8,2,2,8,1024,5", "16,8,2,2,16,640,4", "16,8,2,2,16,768,3", "16,8,2,2,16,768,3"),
    "v5_big_lds": ("16,16,1,4,8,2040,2", "16,16,1,4,16,1020,2", "16,4,4,1,16,768,3", "16,4,4,1,16,768,3"),
}


def build():
    os.makedirs(EXP, exist_ok=True)
    procs = []
    for name, (t8, t16, t32, t64) in VARIANTS.items():
        out = os.path.join(EXP, "libtiled_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
               "-Wno-unused-function", "-shared", "-DMVS_T8=%s" % t8, "-DMVS_T16=%s" % t16, "-DMVS_T32=%s" % t32, "-DMVS_T64=%s" % t64,
               os.path.join(CSRC, "cost_volume_tiled.hip"), os.path.join(CSRC, "common.hip"), "-o", out]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, p in procs:
        o = p.communicate()[0].decode()
        print(name, "rc", p.returncode, o[-400:] if p.returncode else "")


def run(stages, views, H, W):
    import torch
    import mvsformer_amd as m
    from mvsformer_amd import ops, synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, scene = synth.make_inputs(views, H, W, seed=0, device=dev)
    out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    torch.cuda.synchronize()
    P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    libs = {}
    for name in VARIANTS:
        path = os.path.join(EXP, "libtiled_%s.so" % name)
        if not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        lib.mvs_cv_tiled_entropy_fwd.argtypes = [P, P, P, I, I, I, I, I, I, I, P, I, P, P]
        lib.mvs_cv_tiled_aggregate_fwd.argtypes = [P, P, P, P, I, I, I, I, I, I, I, P, P, P, I, P, P]
        lib.mvs_cv_tiled_workspace_bytes.restype = L
        lib.mvs_cv_tiled_workspace_bytes.argtypes = [I] * 6
        libs[name] = lib

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    rows = []
    st = torch.cuda.current_stream().cuda_stream
    for i in stages:
        f = feats["stage%d" % i].contiguous()
        B, V, C, Hs, Ws = f.shape
        hyp_c = out["stage%d" % i]["depth_values"].contiguous()
        D = hyp_c.shape[1]
        z = synth.plane_depth(scene, synth.STAGE_SCALES[i - 1], device=dev)
        half = ((1.0 / hyp_c.min(1)[0] - 1.0 / hyp_c.max(1)[0]) * 0.5).mean()
        hyp_s = (1.0 / (1.0 / z[None, None] + torch.linspace(-1, 1, D, device=dev).view(1, D, 1, 1) * half)).contiguous()
        rt = ops.proj_prepare(proj["stage%d" % i])
        w = torch.rand(1, V - 1, Hs, Ws, device=dev)
        fcl = ops.to_channels_last(f)
        for hname, hyp in (("smooth", hyp_s), ("cascade", hyp_c)):
            ent0 = ops.cv_entropy(fcl, rt, hyp, 8)
            vol0, _ = ops.cv_aggregate(fcl, rt, hyp, w, 8, True)
            base = {"direct_A": timeit(lambda: ops.cv_entropy(fcl, rt, hyp, 8)), "direct_B": timeit(lambda: ops.cv_aggregate(fcl, rt, hyp, w, 8, True))}
            print("stage%d %-7s direct A %.3f B %.3f" % (i, hname, base["direct_A"], base["direct_B"]))
            for name, lib in libs.items():
                ent = torch.empty(B, V - 1, Hs, Ws, device=dev)
                vol = torch.empty(B, 8, D, Hs, Ws, device=dev)
                sim = torch.empty(B, Hs, Ws, device=dev)
                nws = lib.mvs_cv_tiled_workspace_bytes(B, V, C, D, Hs, Ws)
                ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
                stats = torch.zeros(2, dtype=torch.int32, device=dev)

                def A(flags, stp=None):
                    rc = lib.mvs_cv_tiled_entropy_fwd(f.data_ptr(), rt.data_ptr(), hyp.data_ptr(), B, V, C, 8, D, Hs, Ws, ent.data_ptr(), flags, stp, st)
                    assert rc == 0, rc

                def Bf(flags):
                    rc = lib.mvs_cv_tiled_aggregate_fwd(f.data_ptr(), rt.data_ptr(), hyp.data_ptr(), w.data_ptr(), B, V, C, 8, D, Hs, Ws, vol.data_ptr(),
                                                        sim.data_ptr(), ws.data_ptr(), flags, None, st)
                    assert rc == 0, rc
                try:
                    A(0, stats.data_ptr())
                    torch.cuda.synchronize()
                    s = stats.cpu().tolist()
                    e1 = (ent - ent0).abs().max().item()
                    A(4)
                    e2 = (ent - ent0).abs().max().item()
                    Bf(0)
                    e3 = (vol - vol0).abs().max().item()
                    r = {"stage": i, "hyp": hname, "variant": name, "A": timeit(lambda: A(0)), "A_allviews": timeit(lambda: A(4)), "B": timeit(lambda: Bf(0)),
                         "nofit": s[1] / max(1, s[0]), "err": max(e1, e2, e3)}
                except Exception as e:                   # noqa: keep scanning
                    r = {"stage": i, "hyp": hname, "variant": name, "error": repr(e)}
                r.update(base)
                rows.append(r)
                print("   %-16s %s" % (name, " ".join("%s %.3f" % (k, v) for k, v in r.items() if isinstance(v, float))))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(REPO, "gpurun_out", "exp_tiled.json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--stages", type=int, nargs="+", default=[3, 4])
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    a = ap.parse_args()
    if a.build:
        build()
    else:
        run(a.stages, a.views, a.height, a.width)
