"""N>1 path on CPU: world_size-2 gloo processes exercise the inference sharding and the timing aggregation that
bench.py uses (no data-path collective exists to test)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvsformer_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    metas = [("scan%d" % (i // 3), i % 3, [1, 2], "scan%d" % (i // 3)) for i in range(11)]
    mine = sharding.shard_samples(metas, rank, world)
    mine_sc = sharding.shard_samples(metas, rank, world, by_scene=True)
    dist.barrier()
    tmax = sharding.timed_region_max(1.0 + rank)              # rank 1 is "slower"
    total, rate = sharding.aggregate_throughput(len(mine), 1.0 + rank)
    out[rank] = (len(mine), sorted(set(m[0] for m in mine_sc)), tmax, total, rate)
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0][0] + out[1][0] == 11 and abs(out[0][0] - out[1][0]) <= 1
    assert set(out[0][1]).isdisjoint(out[1][1]) and len(out[0][1]) + len(out[1][1]) == 4
    for r in range(world):
        assert out[r][2] == 2.0                      # MAX over ranks
        assert out[r][3] == 11 and abs(out[r][4] - 11 / 2.0) < 1e-9


def test_shard_is_a_partition():
    metas = list(range(37))
    for world in (1, 2, 4, 8):
        got = sorted(sum((sharding.shard_samples(metas, r, world) for r in range(world)), []))
        assert got == metas


# ------------------------------------------------------------------------------------------------ SyncBatchNorm exchange
def _syncbn_worker(rank, world, port, out):
    """Ranks hold DIFFERENT element counts (2/4/6-style unequal batches): every rank must issue the same collective and get
    the exact global count - the case that deadlocked a per-rank 'counts are equal' cache."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn as nn
    from mvsformer_amd import autograd as ag
    bn = nn.SyncBatchNorm(3)
    res = []
    for counts in ((4.0, 4.0), (2.0, 6.0), (4.0, 4.0), (float(2 ** 24 + 3), float(2 ** 25 + 5))):
        c = counts[rank]
        sums = torch.arange(6, dtype=torch.float32) + 10.0 * rank
        tot, count_dev = ag._sync_sums(sums, c, bn)
        n = float(count_dev[0].double() * 4096.0 + count_dev[1].double())
        res.append((tot.tolist(), n))
        back, none = ag._sync_sums(torch.ones(6) * (rank + 1), 0.0, bn)      # backward form: sums only
        assert none is None
        res.append((back.tolist(), 0.0))
    out[rank] = res
    # a plain BatchNorm (or no process group) reduces nothing
    same, cd = ag._sync_sums(torch.ones(6), 5.0, nn.BatchNorm3d(3))
    assert cd is None and same.tolist() == [1.0] * 6
    dist.destroy_process_group()


def test_two_rank_syncbn_sums_and_counts():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_syncbn_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == out[1]                                               # every rank ends with the same statistics
    want_sums = [float(2 * i + 10) for i in range(6)]
    totals = [8.0, 8.0, 8.0, float(2 ** 24 + 3 + 2 ** 25 + 5)]
    for k, tot in enumerate(totals):
        sums, n = out[0][2 * k]
        assert sums == want_sums and n == tot, (k, sums, n)
        assert out[0][2 * k + 1][0] == [3.0] * 6


def _launch_probe(args):
    import json
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    with open(os.path.join(args["dir"], "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "world": world, "local": int(os.environ["LOCAL_RANK"]), "sum": t.item()}, f)
    dist.destroy_process_group()


def test_launch_ranks_spawns_one_process_per_gpu(tmp_path, monkeypatch):
    """bench.py --gpus N self-launch (reference train.py:179-191): N ranks with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set;
    under torchrun (WORLD_SIZE present) or with N = 1 it must NOT spawn; more ranks than GPUs is an error."""
    import json
    import pytest
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert sharding.launch_ranks(_launch_probe, {"dir": str(tmp_path)}, 2) is True
    got = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [g["rank"] for g in got] == [0, 1] and all(g["world"] == 2 and g["sum"] == 3.0 for g in got)
    assert sharding.launch_ranks(_launch_probe, {"dir": str(tmp_path)}, 1) is False
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert sharding.launch_ranks(_launch_probe, {"dir": str(tmp_path)}, 2) is False
    monkeypatch.delenv("WORLD_SIZE")
    with pytest.raises(RuntimeError):
        sharding.launch_ranks(_launch_probe, {"dir": str(tmp_path)}, 4)
