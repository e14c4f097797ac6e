#!/usr/bin/env python
"""bench_train.py — training-step throughput of the plane-sweep path at BASELINE config 3 geometry (not the judged
metric; bench.py is).  One step = forward + backward of the 4-stage cascade (ndepths 32/16/8/8 as BASELINE states) on
one 640x512, 5-view sample per GPU with the reference's ``ce_loss_stage4`` (fused HIP kernel) on every stage's ``prob_volume_pre``, plus an AdamW
step.  With N > 1 the cascade is wrapped in DistributedDataParallel: RCCL gradient all-reduce over xGMI,
SyncBatchNorm statistics exchanged by the BatchNorm autograd function.  ``--dtype bf16`` (default) is the config-3 line: forward
under ``torch.autocast(bfloat16)`` as the reference trainer does (trainer/mvsformer_trainer.py:104-106) - bf16 regularizer on the
bf16 matrix cores, fp32 cost volume / statistics / head / loss / master weights; ``--dtype f32`` keeps everything fp32.

    python bench_train.py --steps 10 [--gpus N]          (N > 1 without a launcher: spawns one rank per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench_train.py --gpus 8 --steps 10
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DDP + SyncBatchNorm even with one rank (smoke test)")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16",
                    help="bf16 (BASELINE configs[2]): forward under torch.autocast(bfloat16) - the regularizer runs on bf16 channel-last "
                         "activations and v_mfma_f32_16x16x32_bf16, the cost volume, BatchNorm statistics, head and loss stay fp32, "
                         "master weights fp32 (what the reference's autocast training does); f32: everything fp32")
    ap.add_argument("--optimizer", choices=["hip", "torch"], default="hip", help="AdamW implementation: mvsformer_amd.optim.FusedAdamW or torch.optim.AdamW(fused=True)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the whole step (forward + loss + backward + AdamW) as ONE hipGraph (mvsformer_amd/graphs.py): host cost per "
                         "step 13-22 ms (box dependent) -> 0.4 ms, so the step runs at the GPU's pace whatever the host does; auto = on, with an "
                         "eager fallback if the capture fails.  Under DistributedDataParallel + SyncBatchNorm the RCCL all-reduces are "
                         "captured with the kernels (DDP constructed, warmed up for 12 iterations and captured on one side stream)")
    return ap.parse_args(argv)


def measure(args, top=12):
    """Runs the timed training steps described by ``args`` (the namespace of :func:`parse`; bench.py builds one for its ``train_config3``
    key) and returns the record on rank 0 (None elsewhere)."""
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    import torch.distributed as dist
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ddp = world > 1 or args.force_ddp
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")       # the watchdog's event queries are not capturable
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import mvsformer_amd as m
    from mvsformer_amd import synth
    torch.manual_seed(0)
    net = m.CascadeMVS(dict(ndepths=[32, 16, 8, 8])).to(dev).train()
    model = net
    # the whole step - SyncBatchNorm's and the reducer's RCCL all-reduces included - is captured under DDP too ("auto"; --graph off = eager)
    use_graph = args.graph in ("on", "auto")
    ddp_stream = None
    if ddp:
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
        if use_graph:                                        # whole-step capture under DDP: construct, warm up and capture on ONE side stream
            ddp_stream = torch.cuda.Stream()
            ddp_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ddp_stream):
                model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local], gradient_as_bucket_view=True, broadcast_buffers=False)
        else:
            model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local], gradient_as_bucket_view=True, broadcast_buffers=False)
    # AdamW as the reference trainer builds it (train.py:98), as mvsformer_amd.optim.FusedAdamW: the same update in three launches of 2048-value
    # blocks (--optimizer torch: torch.optim.AdamW(fused=True, capturable=True), five launches of 40 us - one block per tensor chunk; its
    # default capturable form spends 304 elementwise launches and 1.1 ms of the step on bias corrections)
    if args.optimizer == "hip":
        from mvsformer_amd.optim import FusedAdamW
        opt = FusedAdamW(model.parameters(), lr=1e-4)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=use_graph, fused=True)
    feats, proj, dv, scene = synth.make_inputs(args.views, args.height, args.width, seed=rank, device=dev)
    feats = {k: v.requires_grad_(True) for k, v in feats.items()}
    from mvsformer_amd.losses import ce_loss_stage4
    gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s, device=dev)[None] for i, s in enumerate(synth.STAGE_SCALES)}
    masks = {k: torch.ones_like(v) for k, v in gts.items()}

    import contextlib
    amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if args.dtype == "bf16" else contextlib.nullcontext

    def step():
        opt.zero_grad(set_to_none=True)
        for v in feats.values():                             # the inputs' gradients go to the feature networks: consumed, not accumulated
            v.grad = None
        with amp():
            out = model(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
        # the reference's loss for depth_type='ce' (trainer/mvsformer_trainer.py:119-120), fused HIP kernel per stage
        loss = sum(ce_loss_stage4(out, gts, masks, dlossw=[1, 1, 1, 1], inverse_depth=True).values())
        loss.backward()
        opt.step()
        return loss

    eager_step, graph_note = step, None
    if use_graph:
        try:
            from mvsformer_amd.graphs import CapturedStep
            # under DDP: the reducer's logger times its first 10 iterations with events (not capturable) and rebuilds its buckets after the
            # first one - torch's whole-network-capture recipe is >= 11 eager iterations first
            step = CapturedStep(eager_step, warmup=12 if ddp else 3, keep_graph=True, stream=ddp_stream)
        except Exception as e:                               # report, fall back to eager launches
            if args.graph == "on":
                raise
            step, use_graph, graph_note = eager_step, False, "capture failed: %r" % (e,)
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # host cost of ONE step measured against an idle GPU (launch queue empty, so nothing blocks): if it is close to ms_per_step
    # the step is bound by the Python / autograd / launch path, not by the kernels
    t0 = time.perf_counter()
    loss = step()
    t_enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    own_dt = time.perf_counter() - t0
    dt = torch.tensor([own_dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    # ---- who ran: one record per rank (device, its own wall time), so that a scaling run shows N distinct GPUs ----
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": local, "device": props.name, "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
            "ms_per_step": round(own_dt / args.steps * 1e3, 3)}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    # ---- per-kernel durations of ONE eager step (HIP events around every C-ABI launch, on the launch stream) and the roofline of the
    #      dominant kernel that has an algorithmic work figure; rank 0 only, after the timed region ----
    kernels, roofline = [], None
    launches = step.node_counts() if use_graph else None      # nodes of the captured step = launches per step (kernels of torch ops included)
    if rank == 0 and not ddp:
        from mvsformer_amd import ops
        eager_step()
        torch.cuda.synchronize()
        with ops.kernel_timer() as kt:
            eager_step()
        work = kt.work
        for name, st in kt.summary().items():
            e = {"kernel": name, "calls_per_step": st["calls"], "ms_per_step": round(st["total_ms"], 4)}
            w = work.get(name)
            if w:
                per_s = w["amount"] / (st["total_ms"] * 1e-3)
                if w["kind"] == "bytes":
                    e.update(bound="hbm", achieved=round(per_s / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(per_s / 1e9 / 8000.0, 4))
                else:
                    peak = 2500.0 if name.startswith("bf16_") else 157.3         # dense bf16 / fp32 MFMA peaks (MI355X_MICROARCH.md)
                    e.update(bound="mfma", achieved=round(per_s / 1e12, 2), peak=peak, unit="TFLOP/s", frac=round(per_s / 1e12 / peak, 4))
                e["algorithmic_per_step"] = w["amount"]
            kernels.append(e)
        kernels.sort(key=lambda e: -e["ms_per_step"])
        dom = next((e for e in kernels if "bound" in e), None)
        if dom:
            roofline = {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")}
            roofline.update(ms_per_step=dom["ms_per_step"], calls_per_step=dom["calls_per_step"], algorithmic_per_step=dom["algorithmic_per_step"],
                            traffic=None, note="largest kernel of the step with an algorithmic work figure; summed over its launches of one eager step")
    rec = None
    if rank == 0:
        rec = ({"roofline": roofline, "kernel_ms_sum": round(sum(e["ms_per_step"] for e in kernels), 3), "kernels": kernels[:top],
                          "launches_per_step": launches,
                          "ranks_seen": dist.get_world_size() if world > 1 else 1, "ranks": ranks,
                          "metric": "training samples/s (fwd+bwd+AdamW), 640x512, 5 views, cascade 32/16/8/8", "value": round(world * args.steps / dt.item(), 3),
                          "unit": "samples/s", "n_gpus": world, "steps": args.steps, "ms_per_step": round(dt.item() / args.steps * 1e3, 2),
                          "host_enqueue_ms_per_step": round(t_enqueue * 1e3, 2), "hip_graph": use_graph, "graph_note": graph_note, "dtype": args.dtype, "data": "synthetic", "scaling": "weak", "final_loss": round(float(loss.detach()), 4),
                          "parallelism": "DDP + SyncBatchNorm over RCCL" if ddp else "single GPU"})
    if ddp and not getattr(args, "keep_process_group", False):
        dist.destroy_process_group()
    return rec


def main(args):
    rec = measure(args)
    if rec is not None:
        print(json.dumps(rec))


if __name__ == "__main__":
    _args = parse()
    from mvsformer_amd import sharding
    if not sharding.launch_ranks(main, _args, _args.gpus):      # --gpus N without a launcher: spawn N ranks (train.py:179-191)
        main(_args)
