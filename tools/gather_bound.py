#!/usr/bin/env python
"""Gather-only bound of the plane sweeps per cascade stage (VERDICT r2 task 1d): tools/probe/gather_probe.hip fetches exactly the
taps a sweep over all source views fetches (config-2 shapes, hypotheses as the random-weight cascade predicts them and a smooth
band around the true surface) with no arithmetic; next to it the real sweeps on the same inputs.

    make -C tools/probe            # here (cross-compiles)
    python tools/gather_bound.py   # on the GPU box -> gpurun_out/gather_bound.{json,txt}
"""
import argparse
import ctypes
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def probe_lib():
    """tools/probe/libgather_probe.so (built by __graft_entry__.build() / make -C tools/probe) or None."""
    path = os.path.join(REPO, "tools", "probe", "libgather_probe.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.probe_taps.argtypes = [P, P, I, I, I, I, I, P, P]
    lib.probe_gather.argtypes = [P, P, I, I, I, I, I, I, P, I, P]
    return lib


def digest():
    """sha256[:16] of the sources that define the measurement (the probe, the shared tap geometry, this file)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("tools/probe/gather_probe.hip", "mvsformer_amd/csrc/geometry.h", "tools/gather_bound.py"):
        h.update(open(os.path.join(REPO, f), "rb").read())
    return h.hexdigest()[:16]


def gather_only_ms(lib, feats, proj, out, stages=(1, 2, 3, 4), iters=20):
    """{stage: ms of ONE gather-only launch} - the taps a sweep over all source views of that stage fetches (the cascade's own
    hypotheses ``out['stageK']['depth_values']``), buffer_load_dwordx4 into registers, no arithmetic.  bench.py calls this after its
    timed region: same box, same run, same inputs."""
    import torch
    from mvsformer_amd import ops
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for i in stages:
        f = feats["stage%d" % i]
        fcl = ops.to_channels_last(f.contiguous() if f.is_contiguous() else f)
        B, V, H, W, C = fcl.shape
        hyp = out["stage%d" % i]["depth_values"].contiguous()
        D = hyp.shape[1]
        rt = ops.proj_prepare(proj["stage%d" % i])
        o00 = torch.empty(B * (V - 1), D, H, W, dtype=torch.int32, device=fcl.device)
        sink = torch.zeros(16, device=fcl.device)
        assert lib.probe_taps(rt.data_ptr(), hyp.data_ptr(), B, V, D, H, W, o00.data_ptr(), st) == 0

        def g():
            assert lib.probe_gather(fcl.data_ptr(), o00.data_ptr(), B, V, C, D, H, W, sink.data_ptr(), 0, st) == 0
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            g()
        b.record()
        torch.cuda.synchronize()
        res[i] = a.elapsed_time(b) / iters
        del o00
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch
    import mvsformer_amd as m
    from mvsformer_amd import ops, synth

    lib = probe_lib()
    assert lib is not None, "make -C tools/probe libgather_probe.so first"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, scene = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
    out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.iters

    rows, lines = [], []
    sink = torch.zeros(16, device=dev)
    for i in args.stages:
        f = feats["stage%d" % i].contiguous()
        B, V, C, H, W = f.shape
        hyp_c = out["stage%d" % i]["depth_values"].contiguous()
        D = hyp_c.shape[1]
        z = synth.plane_depth(scene, synth.STAGE_SCALES[i - 1], device=dev)
        half = ((1.0 / hyp_c.min(1)[0] - 1.0 / hyp_c.max(1)[0]) * 0.5).mean()
        hyp_s = (1.0 / (1.0 / z[None, None] + torch.linspace(-1, 1, D, device=dev).view(1, D, 1, 1) * half)).contiguous()
        rt = ops.proj_prepare(proj["stage%d" % i])
        fcl = ops.to_channels_last(f)
        w = torch.rand(B, V - 1, H, W, device=dev)
        gathered = 4.0 * C * 4 * D * H * W * (V - 1) * B          # bytes through the L1 per sweep
        alg = 4.0 * B * H * W * (V * C + D + 8 * D)
        for hname, hyp in (("cascade", hyp_c), ("smooth", hyp_s)):
            o00 = torch.empty(B * (V - 1), D, H, W, dtype=torch.int32, device=dev)
            rc = lib.probe_taps(rt.data_ptr(), hyp.data_ptr(), B, V, D, H, W, o00.data_ptr(), st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            r = {"stage": i, "hyp": hname, "C": C, "D": D, "H": H, "W": W, "gathered_GB": gathered / 1e9, "algorithmic_MB": alg / 1e6}
            for mode, name in ((0, "gather_regs"), (1, "gather_lds_dma16k"), (2, "gather_lds_dma8k")):
                def g():
                    rc = lib.probe_gather(fcl.data_ptr(), o00.data_ptr(), B, V, C, D, H, W, sink.data_ptr(), mode, st)
                    assert rc == 0, rc
                r[name + "_ms"] = timeit(g)
            if C == 8:
                def g3():
                    rc = lib.probe_gather(fcl.data_ptr(), o00.data_ptr(), B, V, C, D, H, W, sink.data_ptr(), 3, st)
                    assert rc == 0, rc
                r["gather_pair_layout_ms"] = timeit(g3)
            r["sweepA_ms"] = timeit(lambda: ops.cv_entropy(fcl, rt, hyp, 8))
            r["sweepB_ms"] = timeit(lambda: ops.cv_aggregate(fcl, rt, hyp, w, 8, True))
            r["sweepA_exact_ms"] = timeit(lambda: ops.cv_entropy(fcl, rt, hyp, 8, exact=True))
            r["sweepB_exact_ms"] = timeit(lambda: ops.cv_aggregate(fcl, rt, hyp, w, 8, True, exact=True))
            r["gather_regs_TBps"] = gathered / r["gather_regs_ms"] * 1e-9
            r["frac_of_gather_bound_A"] = r["gather_regs_ms"] / r["sweepA_ms"]
            r["frac_of_gather_bound_B"] = r["gather_regs_ms"] / r["sweepB_ms"]
            rows.append(r)
            line = ("stage%d %-7s C=%2d D=%2d  gather-only: regs %.3f ms (%.1f TB/s through L1)  lds-dma %.3f / %.3f ms | sweep A %.3f (exact %.3f)  "
                    "sweep B %.3f (exact %.3f) ms | gather bound / sweep: A %.2f  B %.2f" %
                    (i, hname, C, D, r["gather_regs_ms"], r["gather_regs_TBps"], r["gather_lds_dma16k_ms"], r["gather_lds_dma8k_ms"], r["sweepA_ms"],
                     r["sweepA_exact_ms"], r["sweepB_ms"], r["sweepB_exact_ms"], r["frac_of_gather_bound_A"], r["frac_of_gather_bound_B"]))
            if C == 8:
                line += " | pair layout gather %.3f ms" % r["gather_pair_layout_ms"]
            print(line, flush=True)
            lines.append(line)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    # stamped with the digest of the sources that define the measurement (the probe, the shared tap geometry): bench.py only uses the file
    # while they are unchanged
    json.dump({"digest": digest(), "rows": rows}, open(os.path.join(REPO, "gpurun_out", "gather_bound.json"), "w"), indent=1)
    open(os.path.join(REPO, "gpurun_out", "gather_bound.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
