#!/usr/bin/env python
"""bench.py — depth maps/s of the MI355X plane-sweep cost-volume path (BASELINE.json metric).

One "step" = one full 4-stage hot-path cascade (hypotheses -> fused cost volume -> 3-D U-Net -> softmax
regression, stages 32/16/8/4 hypotheses) for ONE reference view at BASELINE config 2: 1536x1152, 5 views,
192-plane DTU depth range, fp32, given precomputed per-stage feature maps already resident in HBM (feature
extraction is outside the path).  Synthetic photo-consistent DTU-shaped inputs, random-init
weights with randomized BatchNorm statistics.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1 is inference sharding: one process per GPU (spawned here like the reference's train.py:179-191 does when no
launcher set WORLD_SIZE), every rank runs the cascade for its own reference views, no data-path collective (weak
scaling); the only collectives are the timing barrier and the MAX over ranks.

Prints ONE JSON line on rank 0 with
  `roofline`             the dominant kernel (largest time per step), timed live with HIP events on the launch stream;
  `roofline_cost_volume` the north-star figure: SURVEY §8(d) algorithmic bytes of the fused cost-volume build of all four
                         stages / the summed time of every launch that builds it (sweeps, transposes);
  `cpu_baseline`         the torch-CPU oracle on a bounded sample (rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_16x16x4_f32)
BF16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (v_mfma_f32_16x16x32_bf16)
# kernels that do fp32-equivalent work on the bf16 matrix cores in three-term split form execute SIX bf16 MFMAs per fp32-equivalent
# K = 32 step: their roofline is the bf16 peak / 6 in direct-form fp32 FLOPs, not the fp32 matrix peak
SPLIT_FORM_PREFIXES = ("x3_", "vis_x3")
ARITHMETIC = ("fp32 in / fp32 out everywhere; the 3-D regularizer's convolutions and the visibility CNN's 3x3 layers compute every fp32 "
              "product as a 3-term bf16 split (x = h + m + l exactly; six v_mfma_f32_16x16x32_bf16 per K = 32 step, fp32 accumulate, dropped "
              "terms <= 2^-24 of a product; tests/test_hip_x3.py bounds it against fp64 by the fp32-MFMA kernels' own error); cost-volume "
              "sweeps, heads and schedulers are fp32 VALU / fp32 MFMA")
TRAFFIC_FILE = os.path.join(REPO, "profiles", "traffic_by_kernel.json")   # written by tools/pmc_traffic.py from rocprofv3 --pmc passes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph of the cascade per stream instead of launching its ~60 kernels")
    ap.add_argument("--cpu-runs", type=int, default=1, help="timed full-size CPU cascades of the cpu_baseline leg (each 1-2 minutes)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of exactly --steps steps each (barrier + synchronize on both sides of every one); `value` is the "
                         "median region, the others are reported as the spread")
    ap.add_argument("--latency-samples", type=int, default=30,
                    help="single-stream per-sample latency (synchronize, one cascade, synchronize - the reference's own timed region, "
                         "test.py:233-249): number of samples, reported as latency_ms_single_stream (median) beside the throughput")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short config-4 / config-5 runs reported as extra keys")
    ap.add_argument("--batch", type=int, default=1, help="reference views per step (B of the [B,V,C,H,W] inputs)")
    ap.add_argument("--features-layout", choices=["nchw", "nhwc"], default="nchw",
                    help="nchw: what the reference's FPN decoder + torch.stack hand over (default, the BASELINE workload); nhwc: the "
                         "same [B,V,C,H,W] tensors channel-last in memory (decoder run in torch.channels_last), consumed zero-copy")
    ap.add_argument("--input-sets", type=int, default=3,
                    help="distinct synthetic input sets (different scene seeds, ~0.53 GB of feature maps each at config 2) the timed steps rotate "
                         "over, step i runs set i %% input_sets: a rank walks different reference views, so the coarse stages' features "
                         "cannot stay resident in the 256 MB Infinity Cache from one step to the next")
    ap.add_argument("--no-train", action="store_true", help="skip the `train_config3` extra key (BASELINE configs[2]: bf16 training step)")
    ap.add_argument("--train-steps", type=int, default=30)
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams per GPU; step i (one reference view) runs on stream i %% streams, so independent "
                         "reference views overlap (MFMA-bound regularizer of one with the VALU/TA-bound sweeps of another)")
    return ap.parse_args()


def cpu_baseline(net, args):
    """The oracle (torch CPU, all host cores) on the WHOLE config-2 workload: the same inputs the GPU is timed on (seed 0),
    `--cpu-runs` timed cascades (default 1; each is 1-2 minutes on the GPU box's host) after one warm-up cascade on a 512x384 sample
    that only warms the thread pool and the allocator.  Returns the JSON object, the CPU inputs and the oracle's outputs (the
    reference the `parity` object is computed against)."""
    from mvsformer_amd import synth
    from oracle import ref_torch
    cores = min(os.cpu_count() or 1, 64)       # torch's intra-op pool stops scaling (and oversubscribes) beyond this
    torch.set_num_threads(cores)
    sds = [{k: v.detach().cpu() for k, v in f.state_dict().items()} for f in net.fusions]
    kw = dict(ndepths=net.ndepths, depth_interals_ratio=net.depth_interals_ratio, tmp=[5.0, 5.0, 5.0, 1.0])
    with torch.no_grad():
        wf, wp, wd, _ = synth.make_inputs(args.views, 384, 512, seed=0)
        t0 = time.time()
        ref_torch.cascade_forward(wf, wp, wd, sds, **kw)
        warm = time.time() - t0
        del wf, wp, wd
        inputs = synth.make_inputs(args.views, args.height, args.width, seed=0, batch=args.batch)[:3]
        times = []
        for _ in range(max(1, args.cpu_runs)):
            t0 = time.time()
            ref = ref_torch.cascade_forward(*inputs, sds, **kw)
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    cpu = {"value": args.batch / dt, "unit": "depth maps/s", "cores": cores, "kind": "port", "scaled_from_sample": False,
           "seconds_per_cascade": [round(t, 2) for t in times], "warmup_seconds_512x384": round(warm, 2),
           "sample": "oracle/ref_torch.cascade_forward (torch %s CPU, %d threads) on the full workload: %d timed cascade(s) at %dx%d x %d views "
                     "(the inputs the GPU runs, seed 0) after one warm-up cascade on a 512x384 sample; unscaled" % (
                         torch.__version__, cores, len(times), args.width, args.height, args.views)}
    return cpu, inputs, ref


def rel_stats(got, want):
    rel = ((got - want).abs() / want.abs()).flatten().float()
    k = max(1, int(round(rel.numel() * 0.999)))
    return {"max": float(rel.max()), "p99.9": float(rel.kthvalue(k).values), "median": float(rel.median())}


def depth_parity(net, feats, proj, dv, tmp, ref, dev):
    """The second half of the BASELINE metric at the benched size: max per-pixel |depth - ref| / |ref| of the HIP cascade against the
    CPU oracle on identical inputs - free-running (each stage consumes the HIP cascade's own hypotheses) and stage by stage with the
    oracle's hypotheses fed to the HIP stage (pure kernel error), as tools/report_parity.py does at small sizes."""
    with torch.no_grad():
        out = net(feats, proj, dv, tmp=tmp)
        free, fed = {}, {}
        for i in range(4):
            k = "stage%d" % (i + 1)
            want = ref[k]["depth"].to(dev)
            free[k] = rel_stats(out[k]["depth"], want)
            st = net.fusions[i](feats[k], proj[k], ref[k]["depth_values"].to(dev), tmp=tmp)
            fed[k] = rel_stats(st["depth"], want)
        final = rel_stats(out["refined_depth"], ref["refined_depth"].to(dev))
    return {"max_rel_depth_err": final["max"], "p99.9_rel_depth_err": final["p99.9"], "median_rel_depth_err": final["median"], "tolerance": 1e-3,
            "free_running_cascade": free, "stage_with_oracle_hypotheses": fed,
            "reference": "oracle/ref_torch.cascade_forward (the reference's formulation on torch CPU) on the timed inputs, full size"}


def kernel_sources_current(recorded):
    """kernel name -> True while the source files THAT kernel is built from still have the digests the counter passes recorded
    (mvsformer_amd/_sources.py): an edit to an unrelated kernel does not void this kernel's traffic evidence."""
    from mvsformer_amd import _sources
    now = _sources.file_digests()
    return lambda name: _sources.current(name, recorded or {}, now)


def time_steps(run, steps, world, dev):
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    time_steps.own_seconds = dt                          # this rank's own wall time (the returned one is the MAX over ranks)
    return float(tt.item()), out


def make_runner(net, feats, proj, dv, tmp, streams, graphs=False, sets=None):
    """``sets``: list of (feats, proj, dv) the steps of a region rotate over (step i of every ``run(n)`` call takes set i % len(sets));
    ``step()`` alone runs set 0 = (feats, proj, dv)."""
    sets = sets or [(feats, proj, dv)]

    def step(k=0):
        f, p, d = sets[k % len(sets)]
        return net(f, p, d, tmp=tmp)

    if graphs:
        # one captured cascade per stream (mvsformer_amd/graphs.py): a step = one hipGraphLaunch instead of ~60 kernel launches; same
        # kernels, same inputs, bit-identical outputs (tests/test_hip_graph.py::test_captured_eval_cascade_is_bit_equal)
        from mvsformer_amd.graphs import CapturedStep

        sl = streams or [torch.cuda.current_stream()]
        # a captured graph bakes its input pointers in: one graph per (stream, input set) pair that the rotation produces
        import math
        period = len(sl) * len(sets) // math.gcd(len(sl), len(sets))
        caps = []
        for i in range(period):
            with torch.cuda.stream(sl[i % len(sl)]), torch.no_grad():
                caps.append(CapturedStep(lambda k=i: step(k), warmup=1))
        torch.cuda.synchronize()

        def run_g(n):
            out = None
            for i in range(n):
                if streams is None:
                    out = caps[i % period]()
                else:
                    with torch.cuda.stream(streams[i % len(streams)]):
                        out = caps[i % period]()
            return out
        return step, run_g

    def run(n):
        out = None
        if streams is None:
            for i in range(n):
                out = step(i)
        else:
            for i in range(n):
                with torch.cuda.stream(streams[i % len(streams)]):
                    out = step(i)
        return out
    return step, run


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import mvsformer_amd as m
    from mvsformer_amd import ops, synth

    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    cpu, ref = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cpu_inputs, ref = cpu_baseline(net, args)
    net = net.to(dev)

    # each rank owns its reference view(s): different scene seed per rank, inputs resident in HBM before timing
    if ref is not None:                                  # the very tensors the oracle ran on
        feats, proj, dv = ({k: v.to(dev) for k, v in cpu_inputs[0].items()}, {k: v.to(dev) for k, v in cpu_inputs[1].items()}, cpu_inputs[2].to(dev))
        del cpu_inputs
    else:
        feats, proj, dv, _ = synth.make_inputs(args.views, args.height, args.width, seed=rank, batch=args.batch, device=dev)
    if args.features_layout == "nhwc":
        feats = {k: v.reshape(-1, *v.shape[2:]).contiguous(memory_format=torch.channels_last).view(v.shape) for k, v in feats.items()}
    tmp = [5.0, 5.0, 5.0, 1.0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    # the input sets the timed steps rotate over: set 0 = the tensors above (the ones the oracle ran on), the others = other scenes
    sets = [(feats, proj, dv)]
    for j in range(1, max(1, args.input_sets)):
        f_j, p_j, d_j, _ = synth.make_inputs(args.views, args.height, args.width, seed=1000 * j + rank, batch=args.batch, device=dev)
        if args.features_layout == "nhwc":
            f_j = {k: v.reshape(-1, *v.shape[2:]).contiguous(memory_format=torch.channels_last).view(v.shape) for k, v in f_j.items()}
        sets.append((f_j, p_j, d_j))
    sets_desc = {"bytes": sum(v.numel() * v.element_size() for v in feats.values()),
                 "seeds": [0 if ref is not None else rank] + [1000 * j + rank for j in range(1, max(1, args.input_sets))]}
    step, run = make_runner(net, feats, proj, dv, tmp, streams, sets=sets)
    step()                                               # first call builds the weight caches (synchronizes once)
    if args.graph:                                       # measured: 1 stream 188.7 vs 189.1 depth maps/s eager, 3 streams 203 vs 215
        _, run = make_runner(net, feats, proj, dv, tmp, streams, graphs=True, sets=sets)
    run(max(args.warmup, args.streams))
    # `--repeats` timed regions of EXACTLY --steps steps each; the reported one is the median region (a 20-step region is 80 ms: one
    # pre-empted launch would otherwise move the headline), all of them are listed in `ms_per_step_repeats`
    regions = []
    for _ in range(max(1, args.repeats)):
        dt_r, out = time_steps(run, args.steps, world, dev)
        regions.append((dt_r, time_steps.own_seconds))
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt, own_dt = regions[order[(len(order) - 1) // 2]]
    assert torch.isfinite(out["refined_depth"]).all()
    with torch.no_grad():                                # the timed steps computed what a fresh single-stream call computes
        again = step(args.steps - 1)                     # the input set of the region's last step
    assert torch.equal(out["refined_depth"], again["refined_depth"]), "timed step and a fresh call disagree"
    if len(sets) > 1 and (args.steps - 1) % len(sets):   # `out` below (smooth-hypotheses key, nhwc key) is the set-0 result
        with torch.no_grad():
            out = step(0)
    parity = depth_parity(net, feats, proj, dv, tmp, ref, dev) if ref is not None else None
    del ref

    # ---- per-sample latency on ONE stream, the reference's own timed region (test.py:233-249: synchronize, forward, synchronize) ----
    lat = []
    with torch.no_grad():
        for _ in range(max(0, args.latency_samples)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    latency = None
    if lat:
        latency = {"median": round(lat[len(lat) // 2], 3), "min": round(lat[0], 3), "p90": round(lat[min(len(lat) - 1, int(0.9 * len(lat)))], 3),
                   "samples": len(lat), "depth_maps_per_s": round(args.batch * 1e3 / lat[len(lat) // 2], 2),
                   "what": "synchronize, one 4-stage cascade on the default stream, synchronize (host launch time included), as the reference "
                           "times a sample (test.py:233-249)"}

    # ---- per-kernel durations: HIP events around every launch, on the launch stream (single stream, after the timed region) ----
    torch.cuda.synchronize()
    with ops.kernel_timer() as timer:
        for _ in range(args.profile_steps):
            step()
    ksum = timer.summary()
    work = timer.work
    traffic_db, traffic_source = {}, None
    if os.path.exists(TRAFFIC_FILE):
        with open(TRAFFIC_FILE) as f:
            tj = json.load(f)
        is_current = kernel_sources_current(tj.get("source_digests"))
        traffic_db = {k: v for k, v in tj.get("kernels", {}).items() if is_current(k)}
        stale = sorted(k for k in tj.get("kernels", {}) if k not in traffic_db)
        traffic_source = ("profiles/traffic_by_kernel.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/prof_traffic.py) collected at HEAD %s; a kernel's "
                          "traffic is reported while the source files it is built from have the digests recorded then (mvsformer_amd/_sources.py): "
                          "%d kernels current, %d stale%s; not re-measured in this run"
                          % (tj.get("collected_at_head"), len(traffic_db), len(stale), (" (" + ", ".join(stale[:6]) + ")") if stale else ""))
    kernels = []
    for name, s in ksum.items():
        w = work.get(name)
        # The j-th launch of a kernel name within a step has the same shape in every profile step: its duration = the MEDIAN over the
        # steps (an event pair also brackets whatever the host does between recording the start event and enqueueing the kernel, and
        # one pre-empted launch out of five must not move the figure); a step's total for the name = the sum of those medians.
        med = timer.per_step_medians(s["all_ms"], args.profile_steps)
        if med:
            s = dict(s, avg_ms=sum(med) / len(med), total_ms=sum(med) * args.profile_steps)
        e = {"kernel": name, "calls_per_step": s["calls"] // args.profile_steps, "avg_ms": round(s["avg_ms"], 5),
             "ms_per_step": round(s["total_ms"] / args.profile_steps, 4)}
        if w:
            per_launch = w["amount"] / s["calls"]
            if w["kind"] == "bytes":
                ach = per_launch / (s["avg_ms"] * 1e-3) / 1e9
                e.update(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                         algorithmic_per_launch=per_launch)
            else:
                ach = per_launch / (s["avg_ms"] * 1e-3) / 1e12
                split = name.startswith(SPLIT_FORM_PREFIXES)
                peak = round(BF16_MFMA_PEAK_TF / 6.0, 1) if split else FP32_MFMA_PEAK_TF
                e.update(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4), algorithmic_per_launch=per_launch,
                         peak_basis=("bf16/6" if split else "fp32"))
                if split:
                    e["frac_of_fp32_mfma_peak"] = round(ach / FP32_MFMA_PEAK_TF, 4)
            tr = traffic_db.get(name)
            e["traffic"] = tr["hbm_bytes_per_launch"] if tr else None
        kernels.append(e)
    kernels.sort(key=lambda e: -e["ms_per_step"])
    dom = next(e for e in kernels if "bound" in e)
    roofline = {"kernel": dom["kernel"], "bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                "frac": dom["frac"], "traffic": dom.get("traffic"), "avg_launch_ms": dom["avg_ms"],
                "algorithmic_per_launch": dom["algorithmic_per_launch"]}
    if "peak_basis" in dom:
        roofline["peak_basis"] = ("dense bf16 MFMA peak 2500 TFLOP/s / 6 bf16 MFMAs per fp32-equivalent product (three-term split form)"
                                  if dom["peak_basis"] == "bf16/6" else "dense fp32 MFMA peak")

    # ---- the north-star number: the fused cost-volume build of the whole cascade against the HBM roofline ----
    cv = [e for e in kernels if e["kernel"].startswith(("cv_", "nchw_to_nhwc"))]
    cv_ms = sum(e["ms_per_step"] for e in cv)
    G = 8
    cv_bytes = 0.0
    for k, f in feats.items():
        i = int(k.replace("stage", "")) - 1
        B, V, C, H, W = f.shape
        cv_bytes += 4.0 * B * H * W * (V * C + net.ndepths[i] + G * net.ndepths[i])
    cv_traffic = [e.get("traffic") for e in cv]
    roofline_cv = {"bound": "hbm", "algorithmic_bytes_per_depth_map": cv_bytes, "ms_per_depth_map": round(cv_ms, 4),
                   "achieved": round(cv_bytes / (cv_ms * 1e-3) / 1e9, 1) if cv_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(cv_bytes / (cv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if cv_ms > 0 else None,
                   "traffic": (sum(t * e["calls_per_step"] for t, e in zip(cv_traffic, cv)) if cv and all(t is not None for t in cv_traffic) else None),
                   "launches": {e["kernel"]: e["ms_per_step"] for e in cv},
                   "note": "sum over the 4 stages of 4*H*W*(V*C + D + G*D) bytes / summed time of the sweeps (+ feature transposes)"}
    # the measured ceiling of the gathering sweeps, in THIS run on THIS box: tools/gather_bound.py's probe (tools/probe/gather_probe.hip) fetches
    # exactly the taps each gathering sweep fetches - same inputs, the cascade's own hypotheses - with no arithmetic; one launch per sweep
    if rank == 0:
        import importlib.util
        spec = importlib.util.spec_from_file_location("gather_bound", os.path.join(REPO, "tools", "gather_bound.py"))
        gbm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gbm)
        glib = gbm.probe_lib()
        if glib is None:
            roofline_cv["gather_bound_source"] = "none: tools/probe/libgather_probe.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
        else:
            with torch.no_grad():
                ref_out = step(0)
            g_ms = gbm.gather_only_ms(glib, feats, proj, ref_out)
            gathering = [e for e in cv if e["kernel"].startswith(("cv_entropy", "cv_aggregate", "cv_corr"))]
            by_c = {feats["stage%d" % i].shape[2]: g_ms[i] for i in g_ms}
            bound = sum(by_c[4 * int(e["kernel"].split("<")[1].split(",")[0].rstrip(">"))] * e["calls_per_step"] for e in gathering)
            roofline_cv["gather_only_ms_per_depth_map"] = round(bound, 4)
            roofline_cv["gather_only_ms_by_stage"] = {"stage%d" % i: round(v, 4) for i, v in g_ms.items()}
            roofline_cv["frac_of_gather_bound"] = round(bound / sum(e["ms_per_step"] for e in gathering), 4)
            roofline_cv["gather_bound_source"] = ("this run: tools/gather_bound.py gather_only_ms (probe sources %s) after the timed region, on the timed inputs "
                                                  "(input set 0) and the cascade's own hypotheses; one gather-only launch per gathering sweep" % gbm.digest())
            del ref_out

    # ---- extra key: the same cost-volume build on SMOOTH hypotheses (a band around the true surface - what a trained checkpoint predicts,
    #      not the noisy ones of a random-weight cascade): per stage the direct gather pair and the LDS-tiled pair (whose reuse only pays
    #      when neighbouring pixels sample neighbouring texels) are both timed and the faster one is taken.  `value` is NOT affected. ----
    roofline_cv_smooth = None
    if rank == 0 and world == 1 and not args.no_other_configs and (args.height, args.width, args.batch) == (1152, 1536, 1):
        scene = synth.make_scene(args.views, args.height, args.width, seed=0)
        per_stage, total_ms = {}, 0.0

        def t_ms(fn, n=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        for i in range(1, 5):
            k = "stage%d" % i
            f = feats[k].contiguous()
            Bf, Vf, Cf, Hf, Wf = f.shape
            hyp_c = out[k]["depth_values"].contiguous()
            Df = hyp_c.shape[1]
            z = synth.plane_depth(scene, synth.STAGE_SCALES[i - 1], device=dev)
            half = ((1.0 / hyp_c.min(1)[0] - 1.0 / hyp_c.max(1)[0]) * 0.5).mean()
            hyp_s = (1.0 / (1.0 / z[None, None] + torch.linspace(-1, 1, Df, device=dev).view(1, Df, 1, 1) * half)).contiguous()
            rt = ops.proj_prepare(proj[k])
            wgt = torch.rand(Bf, Vf - 1, Hf, Wf, device=dev)
            fcl = ops.to_channels_last(f)
            t_tr = t_ms(lambda: ops.to_channels_last(f))
            direct = t_tr + t_ms(lambda: ops.cv_entropy(fcl, rt, hyp_s, 8)) + t_ms(lambda: ops.cv_aggregate(fcl, rt, hyp_s, wgt, 8, True))
            tiled = None
            if ops.cv_tiled_supported(f):
                tiled = t_ms(lambda: ops.cv_tiled_entropy(f, rt, hyp_s, 8)) + t_ms(lambda: ops.cv_tiled_aggregate(f, rt, hyp_s, wgt, 8, True))
            best = min(direct, tiled) if tiled is not None else direct
            per_stage[k] = {"direct_incl_transpose_ms": round(direct, 4), "tiled_ms": round(tiled, 4) if tiled is not None else None,
                            "chosen": "tiled" if tiled is not None and tiled < direct else "direct"}
            total_ms += best
            del fcl, wgt, hyp_s
        roofline_cv_smooth = {"bound": "hbm", "algorithmic_bytes_per_depth_map": cv_bytes, "ms_per_depth_map": round(total_ms, 4),
                              "achieved": round(cv_bytes / (total_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(cv_bytes / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "stages": per_stage,
                              "note": "hypotheses = a smooth band around the true surface (same band width as the cascade's own); the pair "
                                      "of sweeps is chosen per stage by timing both; standalone launches, not part of `value`"}

    # ---- BASELINE configs[3] / configs[4] shapes through the same cascade (short runs, extra keys; not the judged metric) ----
    other = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        other = {}
        for key, (V2, H2, W2) in (("configs[3] BlendedMVS stress 2048x1536 x7 views", (7, 1536, 2048)),
                                  ("configs[4] Tanks&Temples 1920x1088 x11 views", (11, 1088, 1920))):
            f2, p2, d2, _ = synth.make_inputs(V2, H2, W2, seed=3, device=dev)
            _, run2 = make_runner(net, f2, p2, d2, tmp, streams)
            run2(args.streams + 1)
            n2 = 30
            dt2, o2 = time_steps(run2, n2, 1, dev)
            assert torch.isfinite(o2["refined_depth"]).all()
            other[key] = {"depth_maps_per_s": round(n2 / dt2, 2), "ms_per_depth_map": round(dt2 / n2 * 1e3, 3), "steps": n2}
            del f2, p2, d2, o2
            torch.cuda.empty_cache()

    # ---- the row before the path: FPN decoder emitting channel-last features, timed beside the path (extra key) ----
    before, end_to_end = None, None
    if rank == 0 and world == 1 and not args.no_other_configs:
        from mvsformer_amd import FPNDecoder, FPNEncoder
        torch.manual_seed(0)
        dec = FPNDecoder([8, 16, 32, 64]).eval().to(dev)
        fenc = FPNEncoder([8, 16, 32, 64]).eval().to(dev)
        img = torch.randn(args.views, 3, args.height, args.width, device=dev)
        enc = [torch.randn(args.views, c, args.height >> i, args.width >> i, device=dev) for i, c in enumerate((8, 16, 32, 64))]
        for _ in range(3):
            dec(*enc)
            fenc(img)
        torch.cuda.synchronize(dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(20):
            outs = dec(*enc)
        e1.record()
        for _ in range(20):
            fenc(img)
        e2.record()
        torch.cuda.synchronize(dev)
        flops = sum(2.0 * 64 * c * 10 * args.views * (args.height >> i) * (args.width >> i) for i, c in enumerate((8, 16, 32)))
        ms = e0.elapsed_time(e1) / 20
        eflops = 0.0
        hh, ww, cin = args.height, args.width, 3
        for (_, k, st), cout in zip(FPNEncoder.LAYERS, (8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64)):
            hh, ww = (hh - 1) // st + 1, (ww - 1) // st + 1
            eflops += 2.0 * k * k * cin * cout * args.views * hh * ww
            cin = cout
        ems = e1.elapsed_time(e2) / 20
        before = {"fpn_decoder_ms_per_depth_map": round(ms, 3), "algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 1),
                  "fpn_encoder_ms_per_depth_map": round(ems, 3), "encoder_algorithmic_tflops": round(eflops / (ems * 1e-3) / 1e12, 1),
                  "peak_tflops_fp32_mfma": 157.3, "peak_tflops_split_form": 416.7,
                  "outputs": "channel-last [N,H,W,C]: consumed by the sweeps without nchw_to_nhwc",
                  "note": "FPNEncoder / FPNDecoder.forward (models/module.py:226-270), eval BatchNorm, %d views, random inputs; the full-resolution layers "
                          "(conv00, conv01, the decoder's last level) in three-term bf16 split form (csrc/conv2d_x3.hip, csrc/fpn_x3.hip), the rest on the "
                          "fp32 matrix cores; not in `value`" % args.views}
        del dec, enc, outs, fenc
        # the DINO ViT-small branch of MVSFormer-P (csrc/vit.hip): half-size bicubic resize, 12 blocks, attention-gated decoder, all views batched
        from mvsformer_amd import vit as V
        torch.manual_seed(0)
        vnet = V.vit_small(patch_size=16, qk_scale="default").eval().to(dev)
        vdec = V.VITDecoderStage4Single(dict(out_ch=64, vit_ch=384, att_fusion=True, nhead=6)).eval().to(dev)
        for _ in range(2):
            V.vit_branch(vnet, vdec, img)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(5):
            vo = V.vit_branch(vnet, vdec, img)
        e1.record()
        torch.cuda.synchronize(dev)
        vms = e0.elapsed_time(e1) / 5
        vh, vw = args.height // 2 // 16, args.width // 2 // 16
        ntok = vh * vw + 1
        vflops = args.views * 12 * (2.0 * ntok * 384 * 1152 + 4.0 * 6 * ntok * ntok * 64 + 2.0 * ntok * 384 * 384 + 4.0 * ntok * 384 * 1536) + \
            args.views * 2.0 * (ntok - 1) * (768 * 384 + 9 * 390 * 384 + 9 * 384 * 384 + 384 * 256 + 4 * 4 * 256 * 128 + 16 * 4 * 128 * 64)
        before.update(vit_ms_per_depth_map=round(vms, 3), vit_algorithmic_tflops=round(vflops / (vms * 1e-3) / 1e12, 1),
                      vit_peak_tflops_split_form=round(BF16_MFMA_PEAK_TF / 6.0, 1),
                      vit_note="vits.vit_small(patch 16) + VITDecoderStage4Single on %d views at %dx%d (half size), random weights; fp32-equivalent "
                               "split-form GEMMs (csrc/vit.hip); not in `value`" % (args.views, args.width // 2, args.height // 2))
        del vnet, vdec, vo
        torch.cuda.empty_cache()
        # ---- images -> depth map: the whole MVSFormer-P model composed as models/mvsformer_model.py:205-308 does (FPN encoder -> ViT branch ->
        # FPN decoder -> the judged cascade), single stream = the reference's own timed region (test.py:233-249 includes feature extraction)
        from mvsformer_amd import DINOMVSNet
        from mvsformer_amd.cascade import randomize_bn_
        torch.manual_seed(0)
        e2e_net = DINOMVSNet(dict(fix=True, depth_type="ce", fusion_type="cnn", inverse_depth=True, base_ch=8, ndepths=list(net.ndepths), feat_chs=[8, 16, 32, 64],
                                  depth_interals_ratio=list(net.depth_interals_ratio), multi_scale=False,
                                  vit_args=dict(twin=False, rescale=0.5, patch_size=16, qk_scale="default", vit_arch="vit_small", vit_ch=384, out_ch=64,
                                                att_fusion=True, nhead=6))).eval()
        randomize_bn_(e2e_net, seed=1)
        e2e_net = e2e_net.to(dev)
        imgs = synth.render_features(synth.make_scene(args.views, args.height, args.width, 0), 1, 3, noise=0.02, device=dev, dtype=torch.float32)
        for _ in range(2):
            eo = e2e_net(imgs, proj, dv, tmp=tmp)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(10):
            eo = e2e_net(imgs, proj, dv, tmp=tmp)
        e1.record()
        torch.cuda.synchronize(dev)
        assert torch.isfinite(eo["refined_depth"]).all()
        e2e_ms = e0.elapsed_time(e1) / 10
        end_to_end = {"ms_per_depth_map": round(e2e_ms, 3), "depth_maps_per_s": round(1e3 / e2e_ms, 2), "streams": 1, "steps": 10,
                      "note": "mvsformer_amd.DINOMVSNet.forward(imgs [1,%d,3,%d,%d], proj_matrices, depth_values): FPNEncoder + DINO ViT-small branch + FPNDecoder + "
                              "the 4-stage cascade, eval, random weights, rendered images; parity of this composition against the real reference DINOMVSNet: "
                              "tests/test_hip_vit.py::test_dinomvsnet_images_to_depth_vs_reference_golden; not `value` (the judged metric starts from features)"
                              % (args.views, args.height, args.width)}
        del e2e_net, eo, imgs, img
        torch.cuda.empty_cache()

    # ---- the same workload with channel-last features (what mvsformer_amd.FPNDecoder emits): no nchw_to_nhwc launches (extra key) ----
    nhwc = None
    if rank == 0 and world == 1 and not args.no_other_configs and args.features_layout == "nchw":
        f2 = {k: v.reshape(-1, *v.shape[2:]).contiguous(memory_format=torch.channels_last).view(v.shape) for k, v in feats.items()}
        _, run2 = make_runner(net, f2, proj, dv, tmp, streams)
        run2(args.streams + 1)
        n2 = 60
        dt2, o2 = time_steps(run2, n2, 1, dev)
        assert torch.equal(o2["refined_depth"], out["refined_depth"])      # same numbers: the sweeps read the same NHWC bytes either way
        nhwc = {"depth_maps_per_s": round(n2 * args.batch / dt2, 2), "ms_per_step": round(dt2 / n2 * 1e3, 3), "steps": n2,
                "note": "features handed over channel-last in memory (zero-copy into the sweeps); not `value`: the reference's decoder emits NCHW"}
        del f2, o2

    # ---- BASELINE configs[2]: one bf16-autocast training step (forward + ce loss + backward + AdamW) of the cascade 32/16/8/8 on a 640x512
    #      5-view sample, replayed as one hipGraph; bench_train.py is the stand-alone form (and the N > 1 DDP form) of the same measurement ----
    train3 = None
    if rank == 0 and world == 1 and not args.no_train:
        import bench_train
        del sets, step, run
        torch.cuda.empty_cache()
        t = bench_train.measure(bench_train.parse(["--steps", str(args.train_steps), "--warmup", "3", "--dtype", "bf16", "--graph", "on"]), top=8)
        train3 = {"ms_per_step": t["ms_per_step"], "samples_per_s": t["value"], "steps": t["steps"], "hip_graph": t["hip_graph"],
                  "launches_per_step": t["launches_per_step"], "host_enqueue_ms_per_step": t["host_enqueue_ms_per_step"],
                  "kernel_ms_sum_eager_events": t["kernel_ms_sum"], "final_loss": t["final_loss"], "dtype": "bf16 autocast (fp32 cost volume, "
                  "statistics, head, loss, master weights)", "workload": t["metric"], "top_kernels": t["kernels"],
                  "note": "extra key, not `value`: BASELINE configs[2] geometry on ONE GPU (the 8-GPU form is bench_train.py --gpus 8: DDP + SyncBatchNorm over RCCL)"}

    # ---- who ran: one record per rank (device, its own wall time), so that a scaling run shows N distinct GPUs ----
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": local_rank, "device": props.name, "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
            "ms_per_step": round(own_dt / args.steps * 1e3, 3)}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    if rank == 0:
        total = world * args.steps * args.batch
        # Key order: the bulky per-kernel table FIRST, the judged scalars LAST - the driver's record keeps the tail of this line.
        line = {
            "kernels": kernels, "other_configs": other, "before_the_path": before, "end_to_end_mvsformer_p": end_to_end, "features_layout_nhwc": nhwc, "ranks": ranks, "train_config3": train3,
            "traffic_source": traffic_source, "parity": parity,
            "metric": "depth maps/sec @1536x1152 N=5 D=192", "value": round(total / dt, 3), "unit": "depth maps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: DTU eval %dx%d, %d views, 192-plane range, 4-stage cascade ndepths=32/16/8/4, "
                                   "fp32 in / fp32 out, one reference view per step per GPU, precomputed features resident in HBM"
                                   % (args.width, args.height, args.views),
                       "arithmetic": ARITHMETIC,
                       "parallelism": "inference sharding of reference views, one process per GPU, no collective" if world > 1 else "single GPU",
                       "streams_per_gpu": args.streams, "features_layout": args.features_layout,
                       "reference_views_per_step": args.batch, "input_sets": len(sets_desc["seeds"]), "input_set_bytes": sets_desc["bytes"],
                       "input_set_seeds": sets_desc["seeds"]},
            "repeats": len(regions), "ms_per_step_repeats": [round(r[0] / args.steps * 1e3, 3) for r in regions],
            "value_is": "median of `repeats` timed regions of exactly `steps` steps each, %d streams in flight" % max(1, args.streams),
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "kernel_ms_sum": round(sum(e["ms_per_step"] for e in kernels), 3),
            "latency_ms_single_stream": latency["median"] if latency else None, "latency": latency,
            "roofline_cost_volume_smooth": roofline_cv_smooth,
            "cpu_baseline": cpu, "roofline_cost_volume": roofline_cv, "roofline": roofline,
            "max_rel_depth_err": parity["max_rel_depth_err"] if parity else None,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    _args = parse()
    from mvsformer_amd import sharding
    if not sharding.launch_ranks(main, _args, _args.gpus):
        main(_args)
