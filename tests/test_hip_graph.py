"""hipGraph capture of a whole training step (mvsformer_amd/graphs.py): a replay must do what the eager step does."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_captured_training_step_matches_eager():
    """From one and the same state (parameters + BatchNorm buffers), one graph replay and one eager step must produce the same loss
    and the same updated state up to the reordering of the backward's float atomics.  (Trajectories over several steps are not
    compared: the training head is an arg-max, so rounding-level parameter differences flip pixels and grow.)"""
    import mvsformer_amd as m
    from mvsformer_amd import synth
    from mvsformer_amd.graphs import CapturedStep
    from mvsformer_amd.losses import ce_loss_stage4
    dev = torch.device("cuda:0")
    feats, proj, dv, scene = synth.make_inputs(3, 128, 192, seed=4, device=dev)
    gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s, device=dev)[None] for i, s in enumerate(synth.STAGE_SCALES)}
    masks = {k: torch.ones_like(v) for k, v in gts.items()}
    torch.manual_seed(0)
    net = m.CascadeMVS(dict(ndepths=[8, 8, 4, 4])).to(dev).train()
    opt = torch.optim.SGD(net.parameters(), lr=1e-2)

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
        loss = sum(ce_loss_stage4(out, gts, masks, dlossw=[1, 1, 1, 1]).values())
        loss.backward()
        opt.step()
        return loss

    graphed = CapturedStep(step, warmup=3)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def restore():
        with torch.no_grad():
            for k, v in net.state_dict().items():
                v.copy_(state[k])

    loss_g = graphed().clone()
    after_g = {k: v.clone() for k, v in net.state_dict().items()}
    loss_g2 = graphed().clone()                              # a second replay really advances the state
    assert not torch.equal(after_g["fusions.0.cost_reg.conv1.conv.weight"], net.state_dict()["fusions.0.cost_reg.conv1.conv.weight"])
    restore()
    loss_e = step().clone()
    assert abs(loss_e.item() - loss_g.item()) <= 1e-6 * abs(loss_e.item()), (loss_e.item(), loss_g.item(), loss_g2.item())
    for k, v in net.state_dict().items():
        if v.dtype.is_floating_point:
            assert (v - after_g[k]).abs().max().item() <= 1e-6 + 1e-4 * v.abs().max().item(), k
        else:
            assert torch.equal(v, after_g[k]), k


def _captured_ddp_worker(rank, port, out):
    """Runs in a process of its own: the capture shares the process with RCCL's watchdog / heartbeat threads, and a process group is
    per process anyway (one rank = one process = one GPU)."""
    import os
    import torch.distributed as dist
    import mvsformer_amd as m
    from mvsformer_amd import synth
    from mvsformer_amd.graphs import CapturedStep
    from mvsformer_amd.losses import ce_loss_stage4
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        feats, proj, dv, scene = synth.make_inputs(3, 128, 192, seed=4, device=dev)
        gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s, device=dev)[None] for i, s in enumerate(synth.STAGE_SCALES)}
        masks = {k: torch.ones_like(v) for k, v in gts.items()}
        torch.manual_seed(0)
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m.CascadeMVS(dict(ndepths=[8, 8, 4, 4]))).to(dev).train()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], gradient_as_bucket_view=True, broadcast_buffers=False)
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o = model(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
            loss = sum(ce_loss_stage4(o, gts, masks, dlossw=[1, 1, 1, 1]).values())
            loss.backward()
            opt.step()
            return loss

        graphed = CapturedStep(step, warmup=12, keep_graph=True, stream=side)
        counts = graphed.node_counts()
        state = {k: v.clone() for k, v in net.state_dict().items()}
        loss_g = graphed().clone()
        after_g = {k: v.clone() for k, v in net.state_dict().items()}
        torch.cuda.synchronize()
        with torch.no_grad():
            for k, v in net.state_dict().items():
                v.copy_(state[k])
        with torch.cuda.stream(side):
            loss_e = step().clone()
        torch.cuda.synchronize()
        worst = []
        for k, v in net.state_dict().items():
            if v.dtype.is_floating_point:
                worst.append(((v - after_g[k]).abs().max().item() / (1e-6 + v.abs().max().item()), k))
            elif not torch.equal(v, after_g[k]):
                worst.append((float("inf"), k))
        out["result"] = (counts, loss_g.item(), loss_e.item(), max(worst))
    finally:
        dist.destroy_process_group()


def test_captured_ddp_syncbn_step_matches_eager():
    """The whole step under DistributedDataParallel + SyncBatchNorm over RCCL (train.py:138-139) captured into ONE hipGraph - the
    SyncBatchNorm all-reduces of forward and backward and the reducer's bucket all-reduce are recorded with the kernels - and replayed:
    same loss and same updated state as the eager DDP step from the same state.  One rank (the box has one GPU; the collectives are real
    RCCL calls on a world of one).  Recipe of torch's whole-network capture: construct DDP, warm up (>= 11 iterations: the reducer's
    logger times its first ten with events and rebuilds its buckets after the first) and capture on ONE side stream."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_captured_ddp_worker, args=(port, out), nprocs=1, join=True)
    counts, loss_g, loss_e, (err, name) = out["result"]
    assert counts is None or counts["kernel"] > 100, counts
    assert abs(loss_e - loss_g) <= 1e-5 * abs(loss_e), (loss_e, loss_g)
    assert err <= 2e-3, (err, name)


def test_captured_eval_cascade_is_bit_equal():
    """The eval cascade (no host synchronization once the weight caches are built) captured into a hipGraph: replays on new inputs
    (copied in place into the captured tensors) reproduce the eager outputs bit for bit."""
    import mvsformer_amd as m
    from mvsformer_amd import synth
    from mvsformer_amd.graphs import CapturedStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, _ = synth.make_inputs(4, 128, 192, seed=1, device=dev)
    feats2, proj2, dv2, _ = synth.make_inputs(4, 128, 192, seed=2, device=dev)

    def step():
        with torch.no_grad():
            out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
        return out["refined_depth"], out["photometric_confidence"], out["stage4"]["sim_depth"]

    eager1 = [t.clone() for t in step()]
    graphed = CapturedStep(step, warmup=2)
    got1 = [t.clone() for t in graphed()]
    for a, b in zip(eager1, got1):
        assert torch.equal(a, b)
    with torch.no_grad():                                   # new scene through the same captured tensors
        for k in feats:
            feats[k].copy_(feats2[k])
            proj[k].copy_(proj2[k])
        dv.copy_(dv2)
    got2 = [t.clone() for t in graphed()]
    eager2 = [t.clone() for t in step()]
    for a, b in zip(eager2, got2):
        assert torch.equal(a, b)
    assert not torch.equal(got1[0], got2[0])
