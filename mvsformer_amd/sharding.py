"""Multi-GPU inference sharding for the plane-sweep path: one process per GPU, every rank owns a disjoint slice
of the (scene, reference view) samples, no data-path collective.

The reference evaluates one sample per iteration on one GPU (``test.py:182-249``; samples come from the ``metas``
list of ``datasets/general_eval.py:38-75``) and every sample is independent, so the natural MI355X scale-out is
``metas[rank::world]`` (or whole scenes per rank for Tanks&Temples: 8 intermediate scenes on 8 GPUs).  The only
collectives are bookkeeping: a barrier around timed regions and a MAX-reduce of wall times / a gather of per-rank
sample counts.  Works with any ``torch.distributed`` backend (``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_samples(samples: Sequence, rank: int, world: int, by_scene: bool = False) -> List:
    """Round-robin over samples (default) or over scenes (``sample[0]`` is the scene key, as in the reference's
    ``metas`` tuples ``(scan, ref_view, src_views, scan)``) so that one scene's depth maps stay on one rank for the
    later point-cloud fusion."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    if not by_scene:
        return list(samples[rank::world])
    scenes = []
    for s in samples:
        if s[0] not in scenes:
            scenes.append(s[0])
    mine = set(scenes[rank::world])
    return [s for s in samples if s[0] in mine]


def timed_region_max(seconds: float, device=None) -> float:
    """MAX over ranks of a locally measured wall time (what bench.py reports)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(local_units: int, seconds: float, device=None) -> Tuple[int, float]:
    """(total units over all ranks, units/s using the slowest rank's time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_units, local_units / seconds
    n = torch.tensor([local_units], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    tmax = timed_region_max(seconds, device)
    return int(n.item()), int(n.item()) / tmax


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn_entry(local_rank: int, world: int, port: int, fn, args) -> None:
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(local_rank),
                      LOCAL_RANK=str(local_rank), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    fn(args)


def launch_ranks(fn, args, gpus: int) -> bool:
    """One process per GPU, as the reference's ``train.py:179-191`` does with ``mp.spawn`` + NCCL ``env://``.

    * launched by ``torch.distributed.run`` (``WORLD_SIZE`` in the environment): nothing to do here, returns False and the
      caller runs ``fn(args)`` as the rank it already is;
    * ``gpus == 1``: returns False (single process);
    * ``gpus > 1`` and no launcher: spawns ``gpus`` local ranks (127.0.0.1 rendezvous on a free port), each calling
      ``fn(args)`` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, waits for them, returns True.

    Asking for more ranks than the node has GPUs is an error (never a silent single-rank run)."""
    import os
    if "WORLD_SIZE" in os.environ or gpus <= 1:
        return False
    have = torch.cuda.device_count()
    if have < gpus:
        raise RuntimeError("--gpus %d requested but this node exposes %d GPU(s); one process per GPU is the only mode" % (gpus, have))
    import torch.multiprocessing as mp
    mp.spawn(_spawn_entry, args=(gpus, _free_port(), fn, args), nprocs=gpus, join=True)
    return True
