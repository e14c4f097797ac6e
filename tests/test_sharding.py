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
