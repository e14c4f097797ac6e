"""hipGraph capture of a whole training step.

One training step of the cascade is ~570 kernel launches, most of them a few microseconds long; driven from Python (autograd
Functions -> ctypes -> hipLaunchKernel) the HOST needs ~16 ms per step, which is the step time - the GPU idles between launches
(bench_train.py prints ``host_enqueue_ms_per_step``).  The step has no data-dependent control flow and no host synchronization
(SyncBatchNorm counts stay on the device, the optimizer runs ``capturable``), so the whole of forward + loss + backward + optimizer
step is recorded once into a hipGraph through ``torch.cuda.graph`` (stream capture sees the ctypes launches because every C-ABI call
is enqueued on torch's current stream) and replayed with one ``hipGraphLaunch`` per step.

The reference has no counterpart (eager PyTorch, trainer/mvsformer_trainer.py:82-165); the contract is the usual one of captured
graphs: static shapes, inputs updated IN PLACE (``tensor.copy_``) between replays, everything the step allocates lives in the
graph's private pool.
"""
from __future__ import annotations

from typing import Callable

import torch


class CapturedStep:
    """``step_fn()`` (no arguments, returns a tensor or a tuple of tensors) captured into a hipGraph after ``warmup`` eager runs on a
    side stream (lazy initialisation - weight-pack caches, hipFuncSetAttribute, allocator growth - must not happen during capture)."""

    def __init__(self, step_fn: Callable[[], object], warmup: int = 3, keep_graph: bool = False, stream=None):
        """``keep_graph``: keep the captured hipGraph_t beside the executable one, so :meth:`node_counts` can walk it.  ``stream``: the
        side stream to warm up AND capture on (DistributedDataParallel wants the stream it was constructed on: its reducer's
        AccumulateGrad hooks remember it)."""
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedStep needs a GPU")
        side = stream if stream is not None else torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
        self._kept = keep_graph
        # with a caller's stream (DDP): thread-local capture mode - the process group's watchdog thread polls events of earlier
        # collectives while we capture, which the default (global) mode turns into a capture error
        with (torch.cuda.graph(self.graph, stream=stream, capture_error_mode="thread_local") if stream is not None else torch.cuda.graph(self.graph)):
            self.outputs = step_fn()

    def node_counts(self):
        """``{"kernel": n, "memcpy": n, "memset": n, "other": n}`` of the captured graph (hipGraphGetNodes / hipGraphNodeGetType through
        ctypes): the launches one replay stands for.  Needs ``keep_graph=True``; None if the runtime does not hand the graph out."""
        if not self._kept:
            return None
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            g = ctypes.c_void_p(self.graph.raw_cuda_graph())
            n = ctypes.c_size_t(0)
            if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
                return None
            nodes = (ctypes.c_void_p * n.value)()
            if hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) != 0:
                return None
            out = {"kernel": 0, "memcpy": 0, "memset": 0, "other": 0}
            for node in nodes:
                t = ctypes.c_int(-1)
                hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t))
                out[{0: "kernel", 1: "memcpy", 2: "memset"}.get(t.value, "other")] += 1      # hipGraphNodeType enum
            return out
        except Exception:                                    # noqa: BLE001 - diagnostics only
            return None

    def __call__(self):
        """Replay; returns the (static) output tensors of the captured step - clone them if they must outlive the next replay."""
        self.graph.replay()
        from . import ops
        ops.bump_weights_epoch()                             # the replay updates parameters / running statistics without passing through Python
        return self.outputs
