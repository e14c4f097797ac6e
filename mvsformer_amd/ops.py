"""Tensor-level wrappers over the C ABI (include/mvs_hip.h): argument checks, output allocation, stream plumbing.

PyTorch is used here for device memory (caching allocator) and the current HIP stream only; every FLOP of the
path runs in libmvs_hip.so.  All functions require CUDA(ROCm) float32 contiguous tensors and raise otherwise —
there is no fallback.  ``KernelTimer`` (used by bench.py) brackets each launch with HIP events recorded on the
launch stream to get per-kernel durations live.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from collections import defaultdict
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib

VIS_PARAM_FLOATS = 3689


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MvsHipError("%s must be a GPU tensor: the MI355X HIP path is the only implementation (no CPU fallback)" % name)
    if t.dtype != dtype:
        raise _lib.MvsHipError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.MvsHipError("%s must be contiguous" % name)
    if t.device.index != torch.cuda.current_device():
        # kernels are launched on the CURRENT device's current stream; a tensor of another GPU would be dereferenced there
        raise _lib.MvsHipError("%s lives on %s but the current device is cuda:%d - wrap the call in torch.cuda.device(...)"
                               % (name, t.device, torch.cuda.current_device()))
    return t


def _opt(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    """Optional tensor argument: same device / dtype / contiguity rules as the mandatory ones."""
    return None if t is None else _chk(t, name)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------------------------- weights epoch
# Kernels that write parameters or BatchNorm buffers THROUGH RAW POINTERS (mvs_adamw_step, the training-mode BatchNorm kernels' running
# statistics, a replayed hipGraph of a whole step) do not advance torch's per-tensor ``_version``, which the eval-path caches (packed
# weights, folded BatchNorm; module._versions) are keyed on.  Every such write advances this process-wide epoch instead, and the cache keys
# carry it: an eval after a training step re-packs, consecutive evals do not.
_weights_epoch = 0


def bump_weights_epoch() -> None:
    global _weights_epoch
    _weights_epoch += 1


def weights_epoch() -> int:
    return _weights_epoch


# ----------------------------------------------------------------------------------------------- timing hook
class KernelTimer:
    """Collects (start, end) HIP events per C-ABI launch, on the stream the kernels are launched on."""

    def __init__(self):
        self.events: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event]]] = defaultdict(list)
        # algorithmic work per kernel name: {"kind": "bytes"|"flops", "amount": total over the recorded launches}
        self.work: Dict[str, Dict[str, object]] = {}
        self.order: List[str] = []                      # tags in launch order (tools/prof_traffic.py aligns rocprofv3's dispatches with them)

    def add_work(self, tag: str, kind: str, amount: float) -> None:
        w = self.work.setdefault(tag, {"kind": kind, "amount": 0.0})
        w["amount"] += amount

    @staticmethod
    def per_step_medians(all_ms, steps: int):
        """Durations of one kernel name over ``steps`` identical steps (launch order preserved) -> the median over the steps for each
        of the name's launches within a step, or None if the launches do not divide into ``steps`` equal steps.  The j-th launch of a
        name has the same shape in every step, and an event pair also brackets whatever the host does between recording the start
        event and enqueueing the kernel: one pre-empted launch out of five must not move a per-kernel figure."""
        n = len(all_ms)
        if steps < 3 or n == 0 or n % steps:
            return None
        cps = n // steps
        return [sorted(all_ms[j::cps])[steps // 2] for j in range(cps)]

    def summary(self) -> Dict[str, Dict[str, float]]:
        torch.cuda.synchronize()
        out = {}
        for k, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[k] = {"calls": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / max(1, len(ms)), "all_ms": ms}
        return out


_timer: Optional[KernelTimer] = None


@contextlib.contextmanager
def kernel_timer():
    global _timer
    prev, _timer = _timer, KernelTimer()
    try:
        yield _timer
    finally:
        _timer = prev


def _call(name: str, tag, *args):
    """``tag`` is None, a kernel name, or ``(kernel name, 'bytes'|'flops', algorithmic amount of this launch)``."""
    fn = getattr(_lib.load(), name)
    if _timer is not None:
        work = None
        if isinstance(tag, tuple):
            tag, kind, amount = tag
            work = (kind, amount)
        tag = tag or name
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        _timer.events[tag].append((a, b))
        _timer.order.append(tag)
        if work:
            _timer.add_work(tag, *work)
    else:
        rc = fn(*args)
    _lib.check(rc, name)


def _nt(cout: int) -> int:
    nt = (cout + 15) // 16
    return 4 if nt == 3 else nt


# ----------------------------------------------------------------------------------------------- projection
def proj_prepare(proj: torch.Tensor) -> torch.Tensor:
    """``proj [B,V,2,4,4]`` -> ``rt [B,V-1,12]`` (reference mvsformer_model.py:69-72 + warping.py:80-82)."""
    _chk(proj, "proj")
    B, V = proj.shape[0], proj.shape[1]
    if proj.shape[2:] != (2, 4, 4) or V < 2:
        raise _lib.MvsHipError("proj must be [B,V>=2,2,4,4], got %s" % (tuple(proj.shape),))
    rt = torch.empty(B, V - 1, 12, device=proj.device, dtype=torch.float32)
    _call("mvs_proj_prepare", None, _ptr(proj), B, V, _ptr(rt), _stream())
    return rt


def proj_relative(src_proj: torch.Tensor, ref_proj: torch.Tensor) -> torch.Tensor:
    _chk(src_proj, "src_proj"), _chk(ref_proj, "ref_proj")
    B = src_proj.shape[0]
    if src_proj.shape != (B, 4, 4) or ref_proj.shape != (B, 4, 4):
        raise _lib.MvsHipError("src_proj/ref_proj must be [B,4,4]")
    rt = torch.empty(B, 12, device=src_proj.device, dtype=torch.float32)
    _call("mvs_proj_relative", None, _ptr(src_proj), _ptr(ref_proj), B, _ptr(rt), _stream())
    return rt


# ----------------------------------------------------------------------------------------------- warp
def warp(src: torch.Tensor, rt: torch.Tensor, depth: torch.Tensor, with_mask: bool = True):
    _chk(src, "src_fea"), _chk(rt, "rt"), _chk(depth, "depth_values")
    B, C, H, W = src.shape
    D = depth.shape[1]
    per_pixel = depth.dim() == 4
    if per_pixel and depth.shape != (B, D, H, W):
        raise _lib.MvsHipError("depth_values must be [B,D,H,W] or [B,D], got %s" % (tuple(depth.shape),))
    if not per_pixel and depth.shape != (B, D):
        raise _lib.MvsHipError("depth_values must be [B,D,H,W] or [B,D], got %s" % (tuple(depth.shape),))
    warped = torch.empty(B, C, D, H, W, device=src.device, dtype=torch.float32)
    mask = torch.empty(B, D, H, W, device=src.device, dtype=torch.uint8) if with_mask else None
    _call("mvs_warp_fwd", None, _ptr(src), _ptr(rt), _ptr(depth), int(per_pixel), B, C, D, H, W, _ptr(warped), _ptr(mask), _stream())
    return warped, (mask.bool() if with_mask else None)


# ----------------------------------------------------------------------------------------------- cost volume
def _cv_flags(exact: Optional[bool]) -> int:
    """bit 0 of the sweeps' ``flags``: 1 = the reference's op order with IEEE divisions and libm exp/log (the default: measured no
    slower than the shortcut form on MI355X, profiles/r03_bench_sweeps.txt), 0 = one reciprocal + Newton step and hardware
    exp2/log2 (``exact=False``, or MVS_CV_FAST=1 to flip the default)."""
    import os
    if exact is None:
        exact = os.environ.get("MVS_CV_FAST", "0") != "1"
    return 1 if exact else 0


def to_channels_last(feat: torch.Tensor) -> torch.Tensor:
    """``[B,V,C,H,W]`` (FPN decoder layout) -> ``[B,V,H,W,C]`` for the gather sweeps.  A tensor that already IS
    channel-last in memory (an FPN decoder run in ``torch.channels_last`` and viewed as ``[B,V,C,H,W]``) is
    passed through as a view: no kernel, no copy."""
    if feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 5 and not feat.is_contiguous():
        cl = feat.permute(0, 1, 3, 4, 2)
        if cl.is_contiguous():
            return cl
    _chk(feat, "features")
    B, V, C, H, W = feat.shape
    out = torch.empty(B, V, H, W, C, device=feat.device, dtype=torch.float32)
    _call("mvs_nchw_to_nhwc", ("nchw_to_nhwc_kernel<%d>" % C, "bytes", 8.0 * feat.numel()), _ptr(feat), _ptr(out), B * V, C, H * W, _stream())
    return out


def to_channels_last_multi(feats: List[torch.Tensor]) -> List[torch.Tensor]:
    """:func:`to_channels_last` of up to four ``[B,V,C,H,W]`` tensors (the cascade's stages) in ONE launch.  Each result is returned as a
    ``[B,V,C,H,W]`` VIEW of channel-last memory, which every consumer's :func:`to_channels_last` passes through without a copy."""
    import ctypes
    outs, jobs = list(feats), []
    for i, f in enumerate(feats):
        if f.is_cuda and f.dtype == torch.float32 and f.dim() == 5 and not f.is_contiguous() and f.permute(0, 1, 3, 4, 2).is_contiguous():
            continue                                             # channel-last already
        _chk(f, "features")
        jobs.append(i)
    if not jobs:
        return outs
    if len(jobs) > 4:
        raise _lib.MvsHipError("to_channels_last_multi: at most four tensors per launch")
    n = len(jobs)
    bufs = [torch.empty(feats[i].shape[0], feats[i].shape[1], feats[i].shape[3], feats[i].shape[4], feats[i].shape[2],
                        device=feats[i].device, dtype=torch.float32) for i in jobs]
    P = ctypes.c_void_p
    a_in = (P * n)(*[feats[i].data_ptr() for i in jobs])
    a_out = (P * n)(*[b.data_ptr() for b in bufs])
    a_n = (ctypes.c_int * n)(*[feats[i].shape[0] * feats[i].shape[1] for i in jobs])
    a_c = (ctypes.c_int * n)(*[feats[i].shape[2] for i in jobs])
    a_hw = (ctypes.c_int64 * n)(*[feats[i].shape[3] * feats[i].shape[4] for i in jobs])
    nbytes = sum(8.0 * feats[i].numel() for i in jobs)
    _call("mvs_nchw_to_nhwc_multi", ("nchw_to_nhwc_multi", "bytes", nbytes), a_in, a_out, a_n, a_c, a_hw, n, _stream())
    for i, b in zip(jobs, bufs):
        outs[i] = b.permute(0, 1, 4, 2, 3)
    return outs


def cv_entropy(feat: torch.Tensor, rt: torch.Tensor, depth: torch.Tensor, G: int, exact: Optional[bool] = None) -> torch.Tensor:
    """``feat`` is channel-last ``[B,V,H,W,C]`` (see :func:`to_channels_last`)."""
    _chk(feat, "features"), _chk(rt, "rt"), _chk(depth, "depth_values")
    B, V, H, W, C = feat.shape
    D = depth.shape[1]
    if depth.shape != (B, D, H, W):
        raise _lib.MvsHipError("depth_values must be [B,D,H,W]=%s, got %s" % ((B, D, H, W), tuple(depth.shape)))
    ent = torch.empty(B, V - 1, H, W, device=feat.device, dtype=torch.float32)
    # algorithmic bytes of a sweep over ALL source views: features once + hypotheses once; the entropy
    # maps themselves are <1 %
    tag = ("cv_entropy_kernel<%d>" % (C // 4), "bytes", 4.0 * B * H * W * (V * C + D))
    _call("mvs_cv_entropy_fwd", tag, _ptr(feat), _ptr(rt), _ptr(depth), B, V, C, G, D, H, W, _ptr(ent), _cv_flags(exact), _stream())
    return ent


def vis(entropy: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
    _chk(entropy, "entropy"), _chk(params, "vis params")
    if params.numel() != VIS_PARAM_FLOATS:
        raise _lib.MvsHipError("vis params must hold %d floats" % VIS_PARAM_FLOATS)
    H, W = entropy.shape[-2:]
    N = entropy.numel() // (H * W)
    out = torch.empty_like(entropy)
    tag = ("vis_kernel", "flops", 2.0 * 3608 * N * H * W)
    _call("mvs_vis_fwd", tag, _ptr(entropy), _ptr(params), N, H, W, _ptr(out), _stream())
    return out


VIS_WINO_FLOATS = 8192


def vis_wino_prepare(params: torch.Tensor) -> torch.Tensor:
    """Transform-domain weights of the two 3x3 layers, laid out per MFMA lane, for :func:`vis_wino`."""
    _chk(params, "vis params")
    if params.numel() != VIS_PARAM_FLOATS:
        raise _lib.MvsHipError("vis params must hold %d floats" % VIS_PARAM_FLOATS)
    prepared = torch.empty(VIS_WINO_FLOATS, device=params.device, dtype=torch.float32)
    _call("mvs_vis_wino_prepare", None, _ptr(params), _ptr(prepared), _stream())
    return prepared


def vis_wino(entropy: torch.Tensor, params: torch.Tensor, prepared: torch.Tensor) -> torch.Tensor:
    """Same function as :func:`vis`; layers 2-3 as Winograd F(2x2,3x3) GEMMs on the matrix cores."""
    _chk(entropy, "entropy"), _chk(params, "vis params"), _chk(prepared, "prepared vis weights")
    if params.numel() != VIS_PARAM_FLOATS or prepared.numel() != VIS_WINO_FLOATS:
        raise _lib.MvsHipError("vis params / prepared block have the wrong size")
    H, W = entropy.shape[-2:]
    N = entropy.numel() // (H * W)
    out = torch.empty_like(entropy)
    tag = ("vis_wino_kernel", "flops", 2.0 * 3608 * N * H * W)           # credited with the direct form's FLOPs
    _call("mvs_vis_wino_fwd", tag, _ptr(entropy), _ptr(params), _ptr(prepared), N, H, W, _ptr(out), _stream())
    return out


VIS_X3_BYTES = 33792


def vis_x3_prepare(params: torch.Tensor) -> torch.Tensor:
    """The two 3x3 layers' weights as three-term bf16 splits, laid out per MFMA lane, for :func:`vis_x3` (uint8 storage)."""
    _chk(params, "vis params")
    if params.numel() != VIS_PARAM_FLOATS:
        raise _lib.MvsHipError("vis params must hold %d floats" % VIS_PARAM_FLOATS)
    prepared = torch.empty(VIS_X3_BYTES, device=params.device, dtype=torch.uint8)
    _call("mvs_vis_x3_prepare", None, _ptr(params), _ptr(prepared), _stream())
    return prepared


def vis_x3(entropy: torch.Tensor, params: torch.Tensor, prepared: torch.Tensor) -> torch.Tensor:
    """Same function as :func:`vis`; layers 2-3 as direct convolutions on the bf16 matrix cores in three-term split form
    (fp32 in / out, fp32-equivalent; csrc/vis_net_x3.hip)."""
    _chk(entropy, "entropy"), _chk(params, "vis params")
    if params.numel() != VIS_PARAM_FLOATS or prepared.dtype != torch.uint8 or prepared.numel() != VIS_X3_BYTES or not prepared.is_cuda:
        raise _lib.MvsHipError("vis params / prepared block have the wrong size or type")
    H, W = entropy.shape[-2:]
    N = entropy.numel() // (H * W)
    out = torch.empty_like(entropy)
    tag = ("vis_x3_kernel", "flops", 2.0 * 3608 * N * H * W)
    _call("mvs_vis_x3_fwd", tag, _ptr(entropy), _ptr(params), _ptr(prepared), N, H, W, _ptr(out), _stream())
    return out


def cv_aggregate(feat: torch.Tensor, rt: torch.Tensor, depth: torch.Tensor, weight: torch.Tensor, G: int,
                 want_sim_depth: bool, exact: Optional[bool] = None, want_bf16: bool = False):
    """``want_bf16``: also return the volume as bf16 channel-last ``[B,D,H,W,G]`` (written by the same launch) -> ``(vol, sim, vol16)``."""
    _chk(feat, "features"), _chk(rt, "rt"), _chk(depth, "depth_values"), _chk(weight, "vis_weight")
    B, V, H, W, C = feat.shape
    D = depth.shape[1]
    vol = torch.empty(B, G, D, H, W, device=feat.device, dtype=torch.float32)
    sim = torch.empty(B, H, W, device=feat.device, dtype=torch.float32) if want_sim_depth else None
    tag = ("cv_aggregate_kernel<%d,%s>" % (C // 4, "true" if want_sim_depth else "false"), "bytes",
           4.0 * B * H * W * (V * C + D + G * D))          # 4*H*W*(V*C + D + G*D)
    if want_bf16:
        vol16 = torch.empty(B, D, H, W, G, device=feat.device, dtype=torch.bfloat16)
        _call("mvs_cv_aggregate_fwd_bf16", tag, _ptr(feat), _ptr(rt), _ptr(depth), _ptr(weight), B, V, C, G, D, H, W,
              _ptr(vol), _ptr(vol16), _ptr(sim), _cv_flags(exact), _stream())
        return vol, sim, vol16
    _call("mvs_cv_aggregate_fwd", tag, _ptr(feat), _ptr(rt), _ptr(depth), _ptr(weight), B, V, C, G, D, H, W,
          _ptr(vol), _ptr(sim), _cv_flags(exact), _stream())
    return vol, sim


def cv_store_bytes(feat: torch.Tensor, D: int, G: int) -> int:
    """Bytes of the per-view correlation store of the stored-correlation sweeps for channel-last ``feat [B,V,H,W,C]`` (-1: not built
    for this shape - C must be 32 or 64)."""
    B, V, H, W, C = feat.shape
    return int(_lib.load().mvs_cv_corr_store_bytes(B, V, C, G, D, H, W))


def cv_corr(feat: torch.Tensor, rt: torch.Tensor, depth: torch.Tensor, G: int, exact: Optional[bool] = None):
    """Sweep A', coarse stages: entropy ``[B,V-1,H,W]`` as :func:`cv_entropy` plus the per-view correlation store for :func:`cv_merge`."""
    _chk(feat, "features"), _chk(rt, "rt"), _chk(depth, "depth_values")
    B, V, H, W, C = feat.shape
    D = depth.shape[1]
    if depth.shape != (B, D, H, W):
        raise _lib.MvsHipError("depth_values must be [B,D,H,W]=%s, got %s" % ((B, D, H, W), tuple(depth.shape)))
    nbytes = cv_store_bytes(feat, D, G)
    if nbytes <= 0:
        raise _lib.MvsHipError("stored-correlation sweeps are built for C = 32 | 64, G = 8 (got C=%d, G=%d)" % (C, G))
    ent = torch.empty(B, V - 1, H, W, device=feat.device, dtype=torch.float32)
    store = torch.empty(nbytes // 4, device=feat.device, dtype=torch.float32)
    tag = ("cv_corr_kernel<%d>" % (C // 4), "bytes", 4.0 * B * H * W * (V * C + D))
    _call("mvs_cv_corr_fwd", tag, _ptr(feat), _ptr(rt), _ptr(depth), B, V, C, G, D, H, W, _ptr(ent), _ptr(store), _cv_flags(exact), _stream())
    return ent, store


def cv_merge(store: torch.Tensor, depth: torch.Tensor, weight: torch.Tensor, V: int, C: int, G: int, want_sim_depth: bool):
    """Sweep B' over the store of :func:`cv_corr`: ``volume [B,G,D,H,W]`` and the similarity arg-max depth."""
    _chk(store, "correlation store"), _chk(depth, "depth_values"), _chk(weight, "vis_weight")
    B, D, H, W = depth.shape
    if weight.shape != (B, V - 1, H, W) or store.numel() * 4 != int(_lib.load().mvs_cv_corr_store_bytes(B, V, C, G, D, H, W)):
        raise _lib.MvsHipError("cv_merge: weight %s / store size do not match the shape" % (tuple(weight.shape),))
    vol = torch.empty(B, G, D, H, W, device=depth.device, dtype=torch.float32)
    sim = torch.empty(B, H, W, device=depth.device, dtype=torch.float32) if want_sim_depth else None
    tag = ("cv_merge_kernel<%d>" % (256 // C), "bytes", 4.0 * B * H * W * (G * D))     # the rest of SURVEY 8d's bytes is credited to sweep A'
    _call("mvs_cv_merge_fwd", tag, _ptr(store), _ptr(depth), _ptr(weight), B, V, C, G, D, H, W, _ptr(vol), _ptr(sim), _stream())
    return vol, sim


def cv_corr_rows(feat: torch.Tensor, rt: torch.Tensor, depth: torch.Tensor, G: int, y0: int, rows: int, store: Optional[torch.Tensor] = None,
                 exact: Optional[bool] = None):
    """Sweep A' on the band of reference rows ``[y0, y0 + rows)``: band-local entropy ``[B,V-1,rows,W]`` + store (``store``: a buffer to reuse)."""
    _chk(feat, "features"), _chk(rt, "rt"), _chk(depth, "depth_values")
    B, V, H, W, C = feat.shape
    D = depth.shape[1]
    if depth.shape != (B, D, H, W):
        raise _lib.MvsHipError("depth_values must be [B,D,H,W]=%s, got %s" % ((B, D, H, W), tuple(depth.shape)))
    nbytes = int(_lib.load().mvs_cv_corr_store_bytes(B, V, C, G, D, rows, W))
    if nbytes <= 0:
        raise _lib.MvsHipError("stored-correlation sweeps are not built for C=%d, G=%d, D=%d" % (C, G, D))
    if store is None or store.numel() * 4 < nbytes:
        store = torch.empty(nbytes // 4, device=feat.device, dtype=torch.float32)
    ent = torch.empty(B, V - 1, rows, W, device=feat.device, dtype=torch.float32)
    tag = ("cv_corr_kernel<%d>" % (C // 4), "bytes", 4.0 * B * rows * W * (V * C + D))
    _call("mvs_cv_corr_rows_fwd", tag, _ptr(feat), _ptr(rt), _ptr(depth), B, V, C, G, D, H, W, int(y0), int(rows), _ptr(ent), _ptr(store),
          _cv_flags(exact), _stream())
    return ent, store


def cv_merge_rows(store: torch.Tensor, depth: torch.Tensor, weight: torch.Tensor, V: int, C: int, G: int, y0: int, r_lo: int, nrows: int,
                  volume: torch.Tensor, sim_depth: Optional[torch.Tensor]) -> None:
    """Sweep B' over a band's store: band rows ``[r_lo, r_lo + nrows)`` of band-local ``weight [B,V-1,rows,W]`` into the whole-image
    ``volume [B,G,D,H,W]`` / ``sim_depth [B,H,W]`` (rows ``y0 + r_lo ...``)."""
    _chk(store, "correlation store"), _chk(depth, "depth_values"), _chk(weight, "vis_weight"), _chk(volume, "volume"), _opt(sim_depth, "sim_depth")
    B, D, H, W = depth.shape
    rows = weight.shape[2]
    if weight.shape != (B, V - 1, rows, W) or volume.shape != (B, G, D, H, W) or store.numel() * 4 < int(_lib.load().mvs_cv_corr_store_bytes(B, V, C, G, D, rows, W)):
        raise _lib.MvsHipError("cv_merge_rows: weight %s / volume %s / store size do not match the shape" % (tuple(weight.shape), tuple(volume.shape)))
    tag = ("cv_merge_kernel<%d>" % (256 // C), "bytes", 4.0 * B * nrows * W * (G * D))
    _call("mvs_cv_merge_rows_fwd", tag, _ptr(store), _ptr(depth), _ptr(weight), B, V, C, G, D, H, W, int(y0), int(rows), int(r_lo), int(nrows),
          _ptr(volume), _ptr(sim_depth), _stream())


def cv_tiled_supported(feat: torch.Tensor) -> bool:
    """The LDS-tiled sweeps take the FPN decoder's NCHW ``[B,V,C,H,W]`` maps directly (C in 8/16/32/64, contiguous fp32)."""
    return (isinstance(feat, torch.Tensor) and feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 5 and feat.is_contiguous()
            and feat.shape[2] in (8, 16, 32, 64))


def cv_tiled_entropy(feat: torch.Tensor, rt: torch.Tensor, depth: torch.Tensor, G: int, exact: Optional[bool] = None,
                     stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sweep A, LDS-tiled: ``feat`` is NCHW ``[B,V,C,H,W]`` -> entropy ``[B,V-1,H,W]`` (reference mvsformer_model.py:73-79,88-90)."""
    _chk(feat, "features"), _chk(rt, "rt"), _chk(depth, "depth_values")
    B, V, C, H, W = feat.shape
    D = depth.shape[1]
    if depth.shape != (B, D, H, W):
        raise _lib.MvsHipError("depth_values must be [B,D,H,W]=%s, got %s" % ((B, D, H, W), tuple(depth.shape)))
    ent = torch.empty(B, V - 1, H, W, device=feat.device, dtype=torch.float32)
    tag = ("cv_tiled_entropy<%d>" % C, "bytes", 4.0 * B * H * W * (V * C + D))
    _call("mvs_cv_tiled_entropy_fwd", tag, _ptr(feat), _ptr(rt), _ptr(depth), B, V, C, G, D, H, W, _ptr(ent), _cv_flags(exact),
          _ptr(stats), _stream())
    return ent


def cv_tiled_aggregate(feat: torch.Tensor, rt: torch.Tensor, depth: torch.Tensor, weight: torch.Tensor, G: int, want_sim_depth: bool,
                       exact: Optional[bool] = None, stats: Optional[torch.Tensor] = None):
    """Sweep B, LDS-tiled: -> ``volume_mean [B,G,D,H,W]``, ``sim_depth [B,H,W]`` or None (mvsformer_model.py:81-85,101-105,151-158)."""
    _chk(feat, "features"), _chk(rt, "rt"), _chk(depth, "depth_values"), _chk(weight, "vis_weight")
    B, V, C, H, W = feat.shape
    D = depth.shape[1]
    if weight.shape != (B, V - 1, H, W):
        raise _lib.MvsHipError("vis_weight must be %s, got %s" % ((B, V - 1, H, W), tuple(weight.shape)))
    vol = torch.empty(B, G, D, H, W, device=feat.device, dtype=torch.float32)
    sim = torch.empty(B, H, W, device=feat.device, dtype=torch.float32) if want_sim_depth else None
    ws = None
    if want_sim_depth:
        nbytes = _lib.load().mvs_cv_tiled_workspace_bytes(B, V, C, D, H, W)
        if nbytes < 0:
            raise _lib.MvsHipError("mvs_cv_tiled_workspace_bytes: unsupported shape")
        if nbytes > 0:
            ws = torch.empty(nbytes, device=feat.device, dtype=torch.uint8)
    tag = ("cv_tiled_aggregate<%d,%s>" % (C, "sim" if want_sim_depth else "nosim"), "bytes", 4.0 * B * H * W * (V * C + D + G * D))
    _call("mvs_cv_tiled_aggregate_fwd", tag, _ptr(feat), _ptr(rt), _ptr(depth), _ptr(weight), B, V, C, G, D, H, W, _ptr(vol), _ptr(sim),
          _ptr(ws), _cv_flags(exact), _ptr(stats), _stream())
    return vol, sim


# ----------------------------------------------------------------------------------------------- 3-D convs
def conv3d_pack(weight: torch.Tensor, transposed: bool, sd: int = 2) -> torch.Tensor:
    """Re-lay a Conv3d (``[Cout,Cin,3,3,3]``) or ConvTranspose3d (``[Cin,Cout,3,3,3]``, ``sd`` = its depth stride) weight."""
    _chk(weight, "conv weight")
    mode = 0 if not transposed else (2 if sd == 1 else 1)
    if transposed == "dgrad":                      # data gradient of a stride-1 Conv3d: weight stays [Cout,Cin,...] of the layer
        mode = 3
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise _lib.MvsHipError("conv weight must be [*,*,3,3,3], got %s" % (tuple(weight.shape),))
    cin, cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    n = _lib.load().mvs_conv3d_packed_floats(cin, cout, mode)
    if n <= 0:
        raise _lib.MvsHipError("unsupported conv channels Cin=%d Cout=%d" % (cin, cout))
    packed = torch.empty(n, device=weight.device, dtype=torch.float32)
    _call("mvs_conv3d_pack_weights", None, _ptr(weight), cin, cout, mode, _ptr(packed), _stream())
    return packed


def conv3d(x, wpacked, cin, cout, stride, scale=None, shift=None, residual=None, relu=True, tag=None):
    _chk(x, "x"), _chk(wpacked, "packed weights"), _opt(scale, "scale"), _opt(shift, "shift")
    B, C, Di, Hi, Wi = x.shape
    assert C == cin
    sd, shw = stride
    Do, Ho, Wo = (Di - 1) // sd + 1, (Hi - 1) // shw + 1, (Wi - 1) // shw + 1
    y = torch.empty(B, cout, Do, Ho, Wo, device=x.device, dtype=torch.float32)
    if residual is not None:
        _chk(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s" % (tuple(residual.shape), tuple(y.shape)))
    tag = ("conv3d_kernel<%d,%d,%d>" % (_nt(cout), sd, shw), "flops", 2.0 * 27 * cin * cout * B * Do * Ho * Wo)
    _call("mvs_conv3d_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), B, cin, cout,
          Di, Hi, Wi, sd, shw, int(relu), _stream())
    return y


def conv3d_wino_supported(cin: int, cout: int, D: int, H: int, W: int) -> bool:
    return bool(_lib.load().mvs_conv3d_wino_supported(cin, cout, D, H, W))


def conv3d_wino_pack(weight: torch.Tensor) -> torch.Tensor:
    """Conv3d weight ``[Cout,Cin,3,3,3]`` -> Winograd F(2x2,3x3) transform-domain slabs for :func:`conv3d_wino`."""
    _chk(weight, "conv weight")
    cout, cin = weight.shape[0], weight.shape[1]
    n = _lib.load().mvs_conv3d_wino_packed_floats(cin, cout)
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or n <= 0:
        raise _lib.MvsHipError("conv3d_wino_pack: unsupported weight %s" % (tuple(weight.shape),))
    packed = torch.empty(n, device=weight.device, dtype=torch.float32)
    _call("mvs_conv3d_wino_pack_weights", None, _ptr(weight), cin, cout, _ptr(packed), _stream())
    return packed


def conv3d_wino(x, wpacked, cin, cout, scale=None, shift=None, residual=None, relu=True):
    _chk(x, "x"), _chk(wpacked, "packed weights"), _opt(scale, "scale"), _opt(shift, "shift")
    B, C, D, H, W = x.shape
    assert C == cin
    y = torch.empty(B, cout, D, H, W, device=x.device, dtype=torch.float32)
    if residual is not None:
        _chk(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s" % (tuple(residual.shape), tuple(y.shape)))
    # work is credited as the direct convolution's FLOPs (the algorithmic figure), not the 2.25x fewer executed
    tag = ("wino_conv3d_kernel", "flops", 2.0 * 27 * cin * cout * B * D * H * W)
    _call("mvs_conv3d_wino_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), B, cin, cout, D, H, W,
          int(relu), _stream())
    return y


def conv3d_x3_supported(cin: int, cout: int, stride) -> bool:
    """True if the 3-term bf16 split form (csrc/conv3d_x3.hip) is built for this layer shape."""
    return bool(_lib.load().mvs_conv3d_x3_supported(cin, cout, int(stride[0]), int(stride[1])))


def conv3d_x3_pack(weight: torch.Tensor, stride=(1, 1)) -> torch.Tensor:
    """Conv3d weight ``[Cout,Cin,3,3,3]`` -> pre-split (h, m, l) bf16 MFMA fragments for :func:`conv3d_x3` (uint8 storage)."""
    _chk(weight, "conv weight")
    cout, cin = weight.shape[0], weight.shape[1]
    sd, shw = int(stride[0]), int(stride[1])
    n = int(_lib.load().mvs_conv3d_x3_packed_bytes(cin, cout, sd, shw))
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or n <= 0:
        raise _lib.MvsHipError("conv3d_x3_pack: unsupported weight %s / stride %s" % (tuple(weight.shape), (sd, shw)))
    packed = torch.empty(n, device=weight.device, dtype=torch.uint8)
    _call("mvs_conv3d_x3_pack_weights", None, _ptr(weight), cin, cout, sd, shw, _ptr(packed), _stream())
    return packed


def conv3d_x3(x, wpacked, cin, cout, stride=(1, 1), scale=None, shift=None, residual=None, relu=True):
    """``Conv3d`` layer (conv, stride (1,s,s) -> folded BatchNorm -> ReLU [+ residual]) on the bf16 matrix cores in 3-term split form:
    fp32 in, fp32 out, fp32-equivalent (include/mvs_hip.h)."""
    _chk(x, "x"), _chk(wpacked, "packed weights", torch.uint8), _opt(scale, "scale"), _opt(shift, "shift")
    B, C, D, H, W = x.shape
    assert C == cin
    sd, shw = int(stride[0]), int(stride[1])
    Ho, Wo = (H - 1) // shw + 1, (W - 1) // shw + 1
    y = torch.empty(B, cout, D, Ho, Wo, device=x.device, dtype=torch.float32)
    if residual is not None:
        _chk(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s" % (tuple(residual.shape), tuple(y.shape)))
    tag = ("x3_conv_kernel<%d,%d,s%d>" % (cin, cout, shw), "flops", 2.0 * 27 * cin * cout * B * D * Ho * Wo)
    _call("mvs_conv3d_x3_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), B, cin, cout, D, H, W, sd, shw,
          int(relu), _stream())
    return y


def deconv3d_x3_supported(cin: int, cout: int, sd: int) -> bool:
    return bool(_lib.load().mvs_deconv3d_x3_supported(cin, cout, int(sd)))


def deconv3d_x3_pack(weight: torch.Tensor, sd: int = 1) -> torch.Tensor:
    """ConvTranspose3d weight ``[Cin,Cout,3,3,3]`` -> pre-split bf16 MFMA fragments for :func:`deconv3d_x3` (uint8 storage)."""
    _chk(weight, "deconv weight")
    cin, cout = weight.shape[0], weight.shape[1]
    n = int(_lib.load().mvs_deconv3d_x3_packed_bytes(cin, cout, int(sd)))
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or n <= 0:
        raise _lib.MvsHipError("deconv3d_x3_pack: unsupported weight %s / sd %d" % (tuple(weight.shape), sd))
    packed = torch.empty(n, device=weight.device, dtype=torch.uint8)
    _call("mvs_deconv3d_x3_pack_weights", None, _ptr(weight), cin, cout, int(sd), _ptr(packed), _stream())
    return packed


def deconv3d_x3(x, wpacked, cin, cout, sd=1, scale=None, shift=None, residual=None, relu=True):
    """Transposed-conv layer (stride (1,2,2)) -> folded BatchNorm -> ReLU [+ residual] in 3-term bf16 split form (include/mvs_hip.h)."""
    _chk(x, "x"), _chk(wpacked, "packed weights", torch.uint8), _opt(scale, "scale"), _opt(shift, "shift")
    B, C, D, H, W = x.shape
    assert C == cin
    y = torch.empty(B, cout, D, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
    if residual is not None:
        _chk(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s" % (tuple(residual.shape), tuple(y.shape)))
    tag = ("x3_deconv_kernel<%d,%d>" % (cin, cout), "flops", 2.0 * 27 * cin * cout * B * D * H * W)
    _call("mvs_deconv3d_x3_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), B, cin, cout, D, H, W, int(sd),
          int(relu), _stream())
    return y


def deconv3d(x, wpacked, cin, cout, sd, scale=None, shift=None, residual=None, relu=True, tag=None):
    _chk(x, "x"), _chk(wpacked, "packed weights"), _opt(scale, "scale"), _opt(shift, "shift")
    B, C, Di, Hi, Wi = x.shape
    assert C == cin
    y = torch.empty(B, cout, Di * sd, Hi * 2, Wi * 2, device=x.device, dtype=torch.float32)
    if residual is not None:
        _chk(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s (stage H, W must be divisible by 8%s)" % (
                tuple(residual.shape), tuple(y.shape), ", D by 8" if sd == 2 else ""))
    name = "deconv3d_s1_kernel<%d>" % cout if (sd == 1 and cout in (8, 16)) else "deconv3d_kernel<%d,%d>" % (_nt(cout), sd)
    tag = (name, "flops", 2.0 * 27 * cin * cout * B * Di * Hi * Wi)
    _call("mvs_deconv3d_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), B, cin,
          cout, Di, Hi, Wi, sd, int(relu), _stream())
    return y


def deconv3d_prob1(x, wpacked, cin, scale, shift, residual, prob_w, prob_b, relu=True):
    """CostRegNet3D tail: ``prob(residual + relu(bn(conv11(x))))`` -> logits ``[B,D,2H,2W]`` without the 8-channel volume."""
    _chk(x, "x"), _chk(wpacked, "packed weights"), _chk(prob_w, "prob.weight"), _opt(prob_b, "prob.bias")
    _opt(scale, "scale"), _opt(shift, "shift")
    B, C, Di, Hi, Wi = x.shape
    assert C == cin
    if residual is not None:
        _chk(residual, "residual")
        if tuple(residual.shape) != (B, 8, Di, 2 * Hi, 2 * Wi):
            raise _lib.MvsHipError("residual shape %s != %s" % (tuple(residual.shape), (B, 8, Di, 2 * Hi, 2 * Wi)))
    logits = torch.empty(B, Di, 2 * Hi, 2 * Wi, device=x.device, dtype=torch.float32)
    tag = ("deconv3d_s1_kernel<8,prob>", "flops", 2.0 * 27 * cin * 8 * B * Di * Hi * Wi)
    _call("mvs_deconv3d_prob1_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(prob_w), _ptr(prob_b),
          _ptr(logits), B, cin, Di, Hi, Wi, int(relu), _stream())
    return logits


def conv3d_small_supported(cin: int, cout: int, stride: int, transposed: bool) -> bool:
    return bool(_lib.load().mvs_conv3d_small_supported(int(cin), int(cout), int(stride), int(bool(transposed))))


def conv3d_small_pack(weight: torch.Tensor, stride: int, transposed: bool) -> torch.Tensor:
    """Conv3d ``[Cout,Cin,3,3,3]`` / ConvTranspose3d ``[Cin,Cout,3,3,3]`` weight -> pre-split bf16 MFMA fragments for :func:`conv3d_small`."""
    _chk(weight, "conv weight")
    cin, cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    n = int(_lib.load().mvs_conv3d_small_packed_bytes(cin, cout, int(stride), int(bool(transposed))))
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or n <= 0:
        raise _lib.MvsHipError("conv3d_small_pack: unsupported weight %s / stride %d / transposed %s" % (tuple(weight.shape), stride, transposed))
    packed = torch.empty(n, device=weight.device, dtype=torch.uint8)
    _call("mvs_conv3d_small_pack_weights", None, _ptr(weight), cin, cout, int(stride), int(bool(transposed)), _ptr(packed), _stream())
    return packed


def conv3d_small(x, wpacked, cin, cout, stride, transposed, scale=None, shift=None, residual=None, relu=True):
    """Small-volume split-form (transposed) convolution layer -> folded BatchNorm -> ReLU [+ residual] (include/mvs_hip.h)."""
    _chk(x, "x"), _chk(wpacked, "packed weights", torch.uint8), _opt(scale, "scale"), _opt(shift, "shift")
    B, C, D, H, W = x.shape
    assert C == cin
    s = int(stride)
    if transposed:
        oshape = (B, cout, 2 * D, 2 * H, 2 * W)
    else:
        oshape = (B, cout, (D - 1) // s + 1, (H - 1) // s + 1, (W - 1) // s + 1)
    y = torch.empty(oshape, device=x.device, dtype=torch.float32)
    if residual is not None:
        _chk(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s" % (tuple(residual.shape), tuple(y.shape)))
    flops = 2.0 * 27 * cin * cout * B * (D * H * W if transposed else y[0, 0].numel())
    tag = ("x3_small_kernel<%s,%d,%d,s%d>" % ("deconv" if transposed else "conv", cin, cout, s), "flops", flops)
    _call("mvs_conv3d_small_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), B, cin, cout, D, H, W, s,
          int(bool(transposed)), int(relu), _stream())
    return y


def tail_x3_pack(weight: torch.Tensor) -> torch.Tensor:
    """conv11's ConvTranspose3d weight ``[16,8,3,3,3]`` -> pre-split bf16 MFMA fragments for :func:`tail_x3` (uint8 storage)."""
    _chk(weight, "conv11 weight")
    if tuple(weight.shape) != (16, 8, 3, 3, 3):
        raise _lib.MvsHipError("tail_x3_pack: weight %s is not [16,8,3,3,3]" % (tuple(weight.shape),))
    packed = torch.empty(int(_lib.load().mvs_tail_x3_packed_bytes()), device=weight.device, dtype=torch.uint8)
    _call("mvs_tail_x3_pack_weights", None, _ptr(weight), _ptr(packed), _stream())
    return packed


def tail_x3(x, wpacked, scale, shift, residual, prob_w, prob_b, relu=True):
    """CostRegNet3D tail in split form: ``prob(residual + relu(bn(conv11(x))))`` -> logits ``[B,D,2H,2W]`` (include/mvs_hip.h)."""
    _chk(x, "x"), _chk(wpacked, "packed weights", torch.uint8), _chk(prob_w, "prob.weight"), _opt(prob_b, "prob.bias")
    _opt(scale, "scale"), _opt(shift, "shift")
    B, C, D, H, W = x.shape
    if C != 16:
        raise _lib.MvsHipError("tail_x3: %d input channels (built for 16)" % C)
    if residual is not None:
        _chk(residual, "residual")
        if tuple(residual.shape) != (B, 8, D, 2 * H, 2 * W):
            raise _lib.MvsHipError("residual shape %s != %s" % (tuple(residual.shape), (B, 8, D, 2 * H, 2 * W)))
    logits = torch.empty(B, D, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
    tag = ("x3_tail_kernel<16,8,prob>", "flops", 2.0 * 27 * 16 * 8 * B * D * H * W)
    _call("mvs_tail_x3_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(prob_w), _ptr(prob_b),
          _ptr(logits), B, D, H, W, int(relu), _stream())
    return logits


def prob3(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    _chk(x, "x"), _chk(w, "prob weight")
    B, C, D, H, W = x.shape
    out = torch.empty(B, D, H, W, device=x.device, dtype=torch.float32)
    _call("mvs_prob3_fwd", ("prob3_kernel", "flops", 2.0 * 27 * C * B * D * H * W), _ptr(x), _ptr(w), B, C, D, H, W, _ptr(out), _stream())
    return out


# ----------------------------------------------------------------------------------------------- head
def head(depth_values: torch.Tensor, tmp: float, training: bool, logits: Optional[torch.Tensor] = None,
         x8: Optional[torch.Tensor] = None, w1: Optional[torch.Tensor] = None, b1: Optional[torch.Tensor] = None):
    _chk(depth_values, "depth_values")
    B, D, H, W = depth_values.shape
    dev = depth_values.device
    prob = torch.empty(B, D, H, W, device=dev, dtype=torch.float32)
    depth = torch.empty(B, H, W, device=dev, dtype=torch.float32)
    conf = torch.empty(B, H, W, device=dev, dtype=torch.float32)
    if x8 is not None:
        _chk(x8, "x8"), _chk(w1, "prob.weight"), _chk(b1, "prob.bias")
        if x8.shape[0] != B or tuple(x8.shape[2:]) != (D, H, W):
            raise _lib.MvsHipError("x8 %s does not match depth_values %s" % (tuple(x8.shape), tuple(depth_values.shape)))
        pre = torch.empty(B, D, H, W, device=dev, dtype=torch.float32)
        _call("mvs_head_fwd", None, None, _ptr(x8), _ptr(w1), _ptr(b1), x8.shape[1], _ptr(depth_values), float(tmp), int(training),
              B, D, H, W, _ptr(pre), _ptr(prob), _ptr(depth), _ptr(conf), _stream())
    else:
        _chk(logits, "logits")
        if logits.shape != depth_values.shape:
            raise _lib.MvsHipError("logits %s != depth_values %s" % (tuple(logits.shape), tuple(depth_values.shape)))
        pre = logits
        _call("mvs_head_fwd", None, _ptr(logits), None, None, None, 0, _ptr(depth_values), float(tmp), int(training), B, D, H, W,
              None, _ptr(prob), _ptr(depth), _ptr(conf), _stream())
    return pre, prob, depth, conf


def depth_regression(p: torch.Tensor, depth_values: torch.Tensor) -> torch.Tensor:
    _chk(p, "p"), _chk(depth_values, "depth_values")
    B, D, H, W = p.shape
    per_pixel = depth_values.dim() == 4
    if (per_pixel and depth_values.shape != p.shape) or (not per_pixel and depth_values.shape != (B, D)):
        raise _lib.MvsHipError("depth_values %s does not match p %s" % (tuple(depth_values.shape), tuple(p.shape)))
    out = torch.empty(B, H, W, device=p.device, dtype=torch.float32)
    _call("mvs_depth_regression", None, _ptr(p), _ptr(depth_values), int(per_pixel), B, D, H, W, _ptr(out), _stream())
    return out


def conf_regression(p: torch.Tensor, n: int) -> torch.Tensor:
    _chk(p, "p")
    B, D, H, W = p.shape
    out = torch.empty(B, H, W, device=p.device, dtype=torch.float32)
    _call("mvs_conf_regression", None, _ptr(p), int(n), B, D, H, W, _ptr(out), _stream())
    return out


def mixup_head(p: torch.Tensor, depth_values: torch.Tensor):
    """depth_type 'mixup_ce' head (reference mvsformer_model.py:126-136) -> (depth [B,H,W], confidence [B,H,W])."""
    _chk(p, "p"), _chk(depth_values, "depth_values")
    if p.shape != depth_values.shape or p.dim() != 4:
        raise _lib.MvsHipError("mixup_head: p %s / depth_values %s must both be [B,D,H,W]" % (tuple(p.shape), tuple(depth_values.shape)))
    B, D, H, W = p.shape
    depth = torch.empty(B, H, W, device=p.device, dtype=torch.float32)
    conf = torch.empty(B, H, W, device=p.device, dtype=torch.float32)
    _call("mvs_mixup_head", None, _ptr(p), _ptr(depth_values), B, D, H, W, _ptr(depth), _ptr(conf), _stream())
    return depth, conf


def prob1(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    _chk(x, "x"), _chk(w, "prob.weight")
    B, C, D, H, W = x.shape
    out = torch.empty(B, 1, D, H, W, device=x.device, dtype=torch.float32)
    _call("mvs_prob1_fwd", None, _ptr(x), _ptr(w), _ptr(bias), B, C, D * H * W, _ptr(out), _stream())
    return out


def init_inverse_range(depth_range: torch.Tensor, ndepths: int, H: int, W: int) -> torch.Tensor:
    _chk(depth_range, "depth_values")
    B, N = depth_range.shape
    hyp = torch.empty(B, ndepths, H, W, device=depth_range.device, dtype=torch.float32)
    _call("mvs_init_inverse_range", None, _ptr(depth_range), N, B, ndepths, H, W, _ptr(hyp), _stream())
    return hyp


def schedule_inverse_range(prev_depth: torch.Tensor, prev_hyp: torch.Tensor, ndepths: int, split_itv: float, H: int, W: int):
    _chk(prev_depth, "depth"), _chk(prev_hyp, "depth_hypo")
    B, Dp, Hl, Wl = prev_hyp.shape
    if prev_depth.shape != (B, Hl, Wl) or (Hl, Wl) != (H // 2, W // 2):
        raise _lib.MvsHipError("schedule_inverse_range: previous stage %s / %s is not half of (%d,%d)" % (
            tuple(prev_depth.shape), tuple(prev_hyp.shape), H, W))
    hyp = torch.empty(B, ndepths, H, W, device=prev_depth.device, dtype=torch.float32)
    _call("mvs_schedule_inverse_range", None, _ptr(prev_depth), _ptr(prev_hyp), Dp, float(split_itv), B, ndepths, H, W, _ptr(hyp), _stream())
    return hyp


def conf_accumulate(conf: torch.Tensor, acc: torch.Tensor, weight: float = 1.0) -> None:
    _chk(conf, "photometric_confidence"), _chk(acc, "prob_maps")
    B, H, W = conf.shape
    _, Hf, Wf = acc.shape
    _call("mvs_conf_accumulate", None, _ptr(conf), B, H, W, _ptr(acc), Hf, Wf, float(weight), _stream())


# ----------------------------------------------------------------------------------------------- training kernels
def _reduce_ws(fn: str, device, *shape) -> torch.Tensor:
    """Partial-row workspace of the two-launch (atomic-free, bit-reproducible) per-channel reductions."""
    nbytes = getattr(_lib.load(), fn)(*shape)
    if nbytes < 0:
        raise _lib.MvsHipError("%s%r: unsupported shape" % (fn, shape))
    return torch.empty(max(int(nbytes) // 4, 1), device=device, dtype=torch.float32)


def bn_stats(x: torch.Tensor) -> torch.Tensor:
    """``x [B,C,...]`` -> ``sums [2C]`` (sum, sum of squares per channel)."""
    _chk(x, "x")
    B, C = x.shape[0], x.shape[1]
    N = x.numel() // (B * C)
    sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
    ws = _reduce_ws("mvs_bn_reduce_workspace_bytes", x.device, B, C, N)
    _call("mvs_bn_stats", None, _ptr(x), B, C, N, _ptr(sums), _ptr(ws), _stream())
    return sums


def _bn_args(sums, gamma, beta, running_mean, running_var, count_dev):
    _chk(sums, "sums"), _opt(gamma, "bn.weight"), _opt(beta, "bn.bias"), _opt(running_mean, "bn.running_mean")
    _opt(running_var, "bn.running_var"), _opt(count_dev, "count_dev")


def bn_finalize(sums, gamma, beta, running_mean, running_var, momentum, eps, count, count_dev=None):
    """``count``: per-channel element count (host float); ``count_dev`` (SyncBatchNorm): device floats ``[n/4096, n%4096]``
    summed over the ranks with the sums - then ``count`` is ignored and nothing is read back to the host."""
    _bn_args(sums, gamma, beta, running_mean, running_var, count_dev)
    C = sums.numel() // 2
    dev = sums.device
    scale, shift, mean, invstd = (torch.empty(C, device=dev, dtype=torch.float32) for _ in range(4))
    if running_mean is not None:
        bump_weights_epoch()                                 # running statistics are written through raw pointers
    _call("mvs_bn_finalize", None, _ptr(sums), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), float(momentum),
          float(eps), float(count), _ptr(count_dev), C, _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _stream())
    return scale, shift, mean, invstd


def bn_finalize_grouped(sums, gamma, beta, running_mean, running_var, momentum, eps, count, groups: int, count_dev=None):
    """``groups`` BatchNorm calls of one module side by side as channels ``g*C + c``: ``sums [2*groups*C]``, parameters ``[C]``."""
    _bn_args(sums, gamma, beta, running_mean, running_var, count_dev)
    CT = sums.numel() // 2
    C = CT // groups
    dev = sums.device
    scale, shift, mean, invstd = (torch.empty(CT, device=dev, dtype=torch.float32) for _ in range(4))
    if running_mean is not None:
        bump_weights_epoch()                                 # running statistics are written through raw pointers
    _call("mvs_bn_finalize_grouped", None, _ptr(sums), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), float(momentum),
          float(eps), float(count), _ptr(count_dev), C, int(groups), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _stream())
    return scale, shift, mean, invstd


def affine_act(x, scale, shift, residual, relu):
    _chk(x, "x"), _chk(scale, "scale"), _chk(shift, "shift"), _opt(residual, "residual")
    B, C = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    _call("mvs_affine_act", None, _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), int(relu), B, C, x.numel() // (B * C), _ptr(y), _stream())
    return y


def bn_bwd_reduce(dy, x, scale, shift, mean, invstd, relu):
    _chk(dy, "dy"), _chk(x, "x")
    B, C = x.shape[0], x.shape[1]
    N = x.numel() // (B * C)
    sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
    ws = _reduce_ws("mvs_bn_reduce_workspace_bytes", x.device, B, C, N)
    _call("mvs_bn_bwd_reduce", None, _ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), int(relu), B, C, N,
          _ptr(sums), _ptr(ws), _stream())
    return sums


def bn_bwd_apply(dy, x, scale, shift, mean, invstd, gamma, sums, count, relu, count_dev=None):
    _chk(dy, "dy"), _chk(x, "x"), _chk(sums, "sums"), _opt(gamma, "gamma"), _opt(count_dev, "count_dev")
    B, C = x.shape[0], x.shape[1]
    dx = torch.empty_like(x)
    _call("mvs_bn_bwd_apply", None, _ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums),
          float(count), _ptr(count_dev), int(relu), B, C, x.numel() // (B * C), _ptr(dx), _stream())
    return dx


def conv3d_wgrad(A: torch.Tensor, Bt: torch.Tensor, stride) -> torch.Tensor:
    """``dW[a][b][27] = sum A[a,p] * Bt[b, p*s-1+k]``; ``A [N,CA,Dp,Hp,Wp]`` lives on the grid the stride divides."""
    _chk(A, "A"), _chk(Bt, "Bt")
    N, CA, Dp, Hp, Wp = A.shape
    _, CB, Db, Hb, Wb = Bt.shape
    dW = torch.zeros(CA, CB, 3, 3, 3, device=A.device, dtype=torch.float32)
    tag = ("wgrad_kernel<%d>" % ((CA + 15) // 16), "flops", 2.0 * 27 * CA * CB * N * Dp * Hp * Wp)
    _call("mvs_conv3d_wgrad", tag, _ptr(A), _ptr(Bt), _ptr(dW), N, CA, CB, Dp, Hp, Wp, Db, Hb, Wb, stride[0], stride[1], _stream())
    return dW


def cv_aggregate_bwd(feat_cl, rt, depth, weight, volume, gvolume, G: int, stats: Optional[torch.Tensor] = None):
    """``gvolume``: fp32 ``[B,G,D,H,W]``, or bf16 channel-last ``[B,D,H,W,G]`` (the bf16 regularizer's first data gradient: converted here)."""
    for t, n in ((feat_cl, "features"), (rt, "rt"), (depth, "depth_values"), (weight, "vis_weight"), (volume, "volume")):
        _chk(t, n)
    B, V, H, W, C = feat_cl.shape
    D = depth.shape[1]
    dfeat = torch.zeros_like(feat_cl)
    mode = os.environ.get("MVS_CV_BWD", "own")
    if gvolume.dtype == torch.bfloat16:
        # (the default scatter kernel reading this form directly was built and measured: its strided 2-byte loads cost it 0.3 ms per step,
        #  three times what the conversion pass costs - NOTEBOOK.md)
        gvolume = bf16_to_f32(_chk16(gvolume, "grad"))
    _chk(gvolume, "grad")
    if mode not in ("own", "lds", "direct"):
        raise _lib.MvsHipError("MVS_CV_BWD=%r: expected 'own', 'lds' or 'direct'" % mode)
    if mode != "direct":
        # "own" (default): per-wavefront LDS windows with owner election - no atomics in the common case; "lds": block-shared window with
        # LDS float atomics; "direct": global atomics with per-lane run merging (DESIGN.md §4.5 for the measurements).
        # MVS_CV_BWD_WINDOW="log2(WX),WY" overrides the texel window.
        own = mode == "own"
        wxl, wy = (int(v) for v in os.environ.get("MVS_CV_BWD_WINDOW", "5,20" if own else "6,24").split(","))
        part = torch.empty((C // 8,) + tuple(weight.shape), device=weight.device, dtype=torch.float32)
        fn = "mvs_cv_aggregate_bwd_own" if own else "mvs_cv_aggregate_bwd_lds"
        _call(fn, "cv_aggregate_bwd_%s_kernel<%d>" % (mode, C), _ptr(feat_cl), _ptr(rt), _ptr(depth), _ptr(weight),
              _ptr(volume), _ptr(gvolume), B, V, C, G, D, H, W, _ptr(dfeat), _ptr(part), wxl, wy, _ptr(stats), _stream())
        return dfeat, (part[0] if C == 8 else part.sum(0))
    dw = torch.empty_like(weight)
    _call("mvs_cv_aggregate_bwd", "cv_aggregate_bwd_kernel<%d>" % (C // 4), _ptr(feat_cl), _ptr(rt), _ptr(depth), _ptr(weight), _ptr(volume),
          _ptr(gvolume), B, V, C, G, D, H, W, _ptr(dfeat), _ptr(dw), _stream())
    return dfeat, dw


def to_channels_first(feat_cl: torch.Tensor) -> torch.Tensor:
    _chk(feat_cl, "features")
    B, V, H, W, C = feat_cl.shape
    out = torch.empty(B, V, C, H, W, device=feat_cl.device, dtype=torch.float32)
    _call("mvs_nhwc_to_nchw", None, _ptr(feat_cl), _ptr(out), B * V, C, H * W, _stream())
    return out


def softmax_bwd(p, dp):
    _chk(p, "p"), _chk(dp, "dp")
    B, D = p.shape[0], p.shape[1]
    out = torch.empty_like(p)
    _call("mvs_softmax_bwd", None, _ptr(p), _ptr(dp), B, D, p.numel() // (B * D), _ptr(out), _stream())
    return out


def prob1_bwd(x, w, dlogits):
    _chk(x, "x"), _chk(w, "w"), _chk(dlogits, "dlogits")
    B, C = x.shape[0], x.shape[1]
    dx = torch.empty_like(x)
    dwb = torch.zeros(C + 1, device=x.device, dtype=torch.float32)
    _call("mvs_prob1_bwd", None, _ptr(x), _ptr(w), _ptr(dlogits), B, C, x.numel() // (B * C), _ptr(dx), _ptr(dwb), _stream())
    return dx, dwb


def sigmoid(x):
    _chk(x, "x")
    y = torch.empty_like(x)
    _call("mvs_sigmoid_fwd", None, _ptr(x), x.numel(), _ptr(y), _stream())
    return y


def sigmoid_bwd(y, dy):
    _chk(y, "y"), _chk(dy, "dy")
    dx = torch.empty_like(y)
    _call("mvs_sigmoid_bwd", None, _ptr(y), _ptr(dy), y.numel(), _ptr(dx), _stream())
    return dx


# ------------------------------------------------------------------------- depth-map consistency filtering (§8 f2)
def geo_filter(ref_depth, srcs_depth, ref_cam, srcs_cam, img_dist_thresh: float = 1.0, depth_thresh: float = 0.01,
               vthresh: float = 2.0, want=("mask", "ref_depth_ave", "points")) -> Dict[str, torch.Tensor]:
    _chk(ref_depth, "ref_depth"), _chk(srcs_depth, "srcs_depth"), _chk(ref_cam, "ref_cam"), _chk(srcs_cam, "srcs_cam")
    n, v, _, h, w = srcs_depth.shape
    if ref_depth.shape != (n, 1, h, w) or ref_cam.shape != (n, 2, 4, 4) or srcs_cam.shape != (n, v, 2, 4, 4):
        raise _lib.MvsHipError("geo_filter: inconsistent shapes %s %s %s %s" % (tuple(ref_depth.shape), tuple(srcs_depth.shape),
                                                                              tuple(ref_cam.shape), tuple(srcs_cam.shape)))
    dev = ref_depth.device
    shapes = dict(reproj_xyd=(n, v, 3, h, w), in_range=(n, v, 1, h, w), masks=(n, v, 1, h, w), mask=(n, 1, h, w),
                  ref_depth_ave=(n, 1, h, w), points=(n, 3, h, w))
    out = {k: torch.empty(shapes[k], device=dev, dtype=torch.uint8 if k == "mask" else torch.float32) for k in want}
    ws = torch.empty(_lib.load().mvs_geo_filter_workspace_bytes(n, v), device=dev, dtype=torch.uint8)
    algo = 4.0 * n * h * w * (1 + v + sum({"reproj_xyd": 3 * v, "in_range": v, "masks": v, "mask": 0.25, "ref_depth_ave": 1,
                                            "points": 3}[k] for k in want))
    _call("mvs_geo_filter_fwd", ("geo_filter", "bytes", algo), _ptr(ref_depth), _ptr(srcs_depth), _ptr(ref_cam), _ptr(srcs_cam), n, v, h,
          w, float(img_dist_thresh), float(depth_thresh), float(vthresh), _ptr(ws), *[_ptr(out.get(k)) for k in
                                                                                      ("reproj_xyd", "in_range", "masks", "mask",
                                                                                       "ref_depth_ave", "points")], _stream())
    if "mask" in out:
        out["mask"] = out["mask"].view(torch.bool)
    return out


def vis_filter(ref_depth, reproj_xyd, in_range, masks_in, img_dist_thresh, depth_thresh, vthresh, want_masks=True, want_ave=True):
    _chk(ref_depth, "ref_depth"), _chk(reproj_xyd, "reproj_xyd")
    n, v, _, h, w = reproj_xyd.shape
    for t, name in ((in_range, "in_range"), (masks_in, "masks")):
        if t is not None:
            _chk(t, name)
    dev = ref_depth.device
    masks = torch.empty(n, v, 1, h, w, device=dev, dtype=torch.float32) if (want_masks and masks_in is None) else None
    mask = torch.empty(n, 1, h, w, device=dev, dtype=torch.uint8) if masks_in is None else None
    ave = torch.empty(n, 1, h, w, device=dev, dtype=torch.float32) if want_ave else None
    _call("mvs_vis_filter_fwd", "vis_filter", _ptr(ref_depth), _ptr(reproj_xyd), _ptr(in_range), _ptr(masks_in), n, v, h, w,
          float(img_dist_thresh), float(depth_thresh), float(vthresh), _ptr(masks), _ptr(mask), _ptr(ave), _stream())
    return masks, (mask.view(torch.bool) if mask is not None else None), ave


def prob_filter(conf: torch.Tensor, thresh, depth_inplace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``conf [n,C,...]`` (C = len(thresh) <= 4) -> bool ``[n,1,...]`` (the reference's ``ref_prob[:, [i]]`` shape); ``depth_inplace`` (same trailing shape) is zeroed where False."""
    import ctypes
    _chk(conf, "conf")
    n, C = conf.shape[0], len(thresh)
    if conf.shape[1] < C:
        raise _lib.MvsHipError("prob_filter: %d thresholds for %d confidence channels" % (C, conf.shape[1]))
    if conf.shape[1] != C:
        conf = conf[:, :C].contiguous()
    HW = conf.numel() // (n * C)
    mask = torch.empty((n, 1) + tuple(conf.shape[2:]), device=conf.device, dtype=torch.uint8)
    if depth_inplace is not None:
        _chk(depth_inplace, "depth")
        if depth_inplace.numel() != n * HW:
            raise _lib.MvsHipError("prob_filter: depth has %d elements, expected %d" % (depth_inplace.numel(), n * HW))
    th = (ctypes.c_float * C)(*[float(t) for t in thresh])
    _call("mvs_prob_filter", "prob_filter", _ptr(conf), n, C, HW, ctypes.cast(th, ctypes.c_void_p), _ptr(mask), _ptr(depth_inplace),
          _stream())
    return mask.view(torch.bool)


def geo_filter_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam, dist_base: float = 4, rel_diff_base: float = 1300,
                       want=("geo_mask", "ref_depth_ave", "points")) -> Dict[str, torch.Tensor]:
    _chk(ref_depth, "ref_depth"), _chk(srcs_depth, "srcs_depth"), _chk(ref_cam, "ref_cam"), _chk(srcs_cam, "srcs_cam")
    n, v, _, h, w = srcs_depth.shape
    if ref_depth.shape != (n, 1, h, w) or ref_cam.shape != (n, 2, 4, 4) or srcs_cam.shape != (n, v, 2, 4, 4):
        raise _lib.MvsHipError("geo_filter_dynamic: inconsistent shapes %s %s %s %s" % (
            tuple(ref_depth.shape), tuple(srcs_depth.shape), tuple(ref_cam.shape), tuple(srcs_cam.shape)))
    dev = ref_depth.device
    shapes = dict(reproj_xyd=(n, v, 3, h, w), masks=(n, v, v - 1, h, w), vis_mask=(n, v, 1, h, w), geo_mask=(n, 1, h, w),
                  ref_depth_ave=(n, 1, h, w), points=(n, 3, h, w))
    isbool = ("masks", "vis_mask", "geo_mask")
    out = {k: torch.empty(shapes[k], device=dev, dtype=torch.uint8 if k in isbool else torch.float32) for k in want}
    ws = torch.empty(_lib.load().mvs_geo_filter_workspace_bytes(n, v), device=dev, dtype=torch.uint8)
    algo = n * h * w * (4.0 * (1 + v) + sum({"reproj_xyd": 12 * v, "masks": v * (v - 1), "vis_mask": v, "geo_mask": 1,
                                             "ref_depth_ave": 4, "points": 12}[k] for k in want))
    _call("mvs_geo_filter_dynamic_fwd", ("geo_filter_dynamic", "bytes", algo), _ptr(ref_depth), _ptr(srcs_depth), _ptr(ref_cam),
          _ptr(srcs_cam), n, v, h, w, float(dist_base), float(rel_diff_base), _ptr(ws),
          *[_ptr(out.get(k)) for k in ("reproj_xyd", "masks", "vis_mask", "geo_mask", "ref_depth_ave", "points")], _stream())
    for k in isbool:
        if k in out:
            out[k] = out[k].view(torch.bool)
    return out


def vis_filter_dynamic(ref_depth, reproj_xyd, dist_base, rel_diff_base, want=("masks", "vis_mask")) -> Dict[str, torch.Tensor]:
    _chk(ref_depth, "ref_depth"), _chk(reproj_xyd, "reproj_xyd")
    n, v, _, h, w = reproj_xyd.shape
    dev = ref_depth.device
    shapes = dict(masks=(n, v, v - 1, h, w), vis_mask=(n, v, 1, h, w), geo_mask=(n, 1, h, w), ref_depth_ave=(n, 1, h, w))
    out = {k: torch.empty(shapes[k], device=dev, dtype=torch.float32 if k == "ref_depth_ave" else torch.uint8) for k in want}
    _call("mvs_vis_filter_dynamic_fwd", "vis_filter_dynamic", _ptr(ref_depth), _ptr(reproj_xyd), n, v, h, w, float(dist_base),
          float(rel_diff_base), *[_ptr(out.get(k)) for k in ("masks", "vis_mask", "geo_mask", "ref_depth_ave")], _stream())
    return {k: (t if k == "ref_depth_ave" else t.view(torch.bool)) for k, t in out.items()}


# ------------------------------------------------------------------------------------- fused CE loss (§8 f3)
def ce_loss(logits, depth_values, depth_gt, mask, inverse_depth: bool, weight: float = 1.0, want_grad: bool = True,
            want_index: bool = False):
    """-> ``(loss [] , acc [2 + rows] with acc[:2] = (sum, count), grad_unscaled [B,D,H,W] or None)`` (+ ``valid``, ``gt_index`` with
    ``want_index``)."""
    _chk(logits, "prob_volume_pre"), _chk(depth_values, "depth_values"), _chk(depth_gt, "depth_gt"), _chk(mask, "mask")
    B, D, H, W = logits.shape
    if depth_values.shape != logits.shape or depth_gt.shape != (B, H, W) or mask.shape != (B, H, W):
        raise _lib.MvsHipError("ce_loss: shapes %s %s %s %s" % (tuple(logits.shape), tuple(depth_values.shape), tuple(depth_gt.shape),
                                                               tuple(mask.shape)))
    dev = logits.device
    acc = torch.empty(_lib.load().mvs_ce_loss_acc_floats(B, H * W), device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    valid = torch.empty(B, H, W, device=dev, dtype=torch.uint8) if want_index else None
    index = torch.empty(B, H, W, device=dev, dtype=torch.int32) if want_index else None
    algo = 4.0 * B * H * W * (2 * D + 2 + (D if want_grad else 0))
    _call("mvs_ce_loss_fwd", ("ce_loss", "bytes", algo), _ptr(logits), _ptr(depth_values), _ptr(depth_gt), _ptr(mask), B, D, H * W,
          int(inverse_depth), float(weight), _ptr(grad), _ptr(acc), _ptr(loss), _ptr(valid), _ptr(index), _stream())
    if want_index:
        return loss, acc, grad, valid.view(torch.bool), index
    return loss, acc, grad


def mixup_ce_loss(logits, depth_values, depth_gt, mask, inverse_depth: bool, weight: float = 1.0, want_grad: bool = True):
    """models/losses.py:353-408 for one stage -> ``(loss [], acc, grad_unscaled [B,D,H,W] or None)``; ``acc[1]`` = sum(mask) + 1e-6."""
    _chk(logits, "prob_volume_pre"), _chk(depth_values, "depth_values"), _chk(depth_gt, "depth_gt"), _chk(mask, "mask")
    B, D, H, W = logits.shape
    if depth_values.shape != logits.shape or depth_gt.shape != (B, H, W) or mask.shape != (B, H, W):
        raise _lib.MvsHipError("mixup_ce_loss: shapes %s %s %s %s" % (tuple(logits.shape), tuple(depth_values.shape), tuple(depth_gt.shape),
                                                                     tuple(mask.shape)))
    dev = logits.device
    acc = torch.empty(_lib.load().mvs_ce_loss_acc_floats(B, H * W), device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    algo = 4.0 * B * H * W * (2 * D + 2 + (D if want_grad else 0))
    _call("mvs_mixup_ce_loss_fwd", ("mixup_ce_loss", "bytes", algo), _ptr(logits), _ptr(depth_values), _ptr(depth_gt), _ptr(mask), B, D, H * W,
          int(inverse_depth), float(weight), _ptr(grad), _ptr(acc), _ptr(loss), _stream())
    return loss, acc, grad


def reg_loss(depth, depth_gt, mask, interval, depth_values=None, inverse_depth: bool = True, weight: float = 1.0, want_grad: bool = True):
    """models/losses.py:51-85 for one stage: ``depth``, ``depth_gt``, ``mask`` ``[B,H,W]``, ``interval [B]``, ``depth_values [B,D,H,W]`` only
    with ``mask_out_range`` -> ``(loss [], acc, grad_unscaled [B,H,W] or None)``."""
    _chk(depth, "depth"), _chk(depth_gt, "depth_gt"), _chk(mask, "mask"), _chk(interval, "depth_interval"), _opt(depth_values, "depth_values")
    B, H, W = depth.shape
    D = depth_values.shape[1] if depth_values is not None else 0
    if depth_gt.shape != (B, H, W) or mask.shape != (B, H, W) or interval.numel() != B or \
            (depth_values is not None and (depth_values.shape[0] != B or tuple(depth_values.shape[2:]) != (H, W))):
        raise _lib.MvsHipError("reg_loss: shapes %s %s %s %s" % (tuple(depth.shape), tuple(depth_gt.shape), tuple(mask.shape), tuple(interval.shape)))
    dev = depth.device
    acc = torch.empty(_lib.load().mvs_ce_loss_acc_floats(B, H * W), device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    grad = torch.empty_like(depth) if want_grad else None
    _call("mvs_reg_loss_fwd", "reg_loss", _ptr(depth), _ptr(depth_gt), _ptr(mask), _ptr(depth_values), _ptr(interval), B, D, H * W,
          int(inverse_depth), float(weight), _ptr(grad), _ptr(acc), _ptr(loss), _stream())
    return loss, acc, grad


def was_loss(prob, depth_values, depth_gt, mask, ot_iter: int = 10, ot_eps: float = 1.0, weight: float = 1.0, want_grad: bool = True):
    """models/losses.py:88-162 (Sinkhorn, discrete form) for one stage: ``prob``, ``depth_values [B,D,H,W]``, ``depth_gt``, ``mask [B,H,W]`` ->
    ``(loss [], acc, grad_unscaled [B,D,H,W] or None)``."""
    _chk(prob, "prob_volume"), _chk(depth_values, "depth_values"), _chk(depth_gt, "depth_gt"), _chk(mask, "mask")
    B, D, H, W = prob.shape
    if depth_values.shape != prob.shape or depth_gt.shape != (B, H, W) or mask.shape != (B, H, W):
        raise _lib.MvsHipError("was_loss: shapes %s %s %s %s" % (tuple(prob.shape), tuple(depth_values.shape), tuple(depth_gt.shape), tuple(mask.shape)))
    dev = prob.device
    acc = torch.empty(_lib.load().mvs_was_loss_acc_floats(B, H * W), device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    grad = torch.empty_like(prob) if want_grad else None
    _call("mvs_was_loss_fwd", "was_loss", _ptr(prob), _ptr(depth_values), _ptr(depth_gt), _ptr(mask), B, D, H * W, int(ot_iter), float(ot_eps), float(weight),
          _ptr(grad), _ptr(acc), _ptr(loss), _stream())
    return loss, acc, grad


def ce_loss_bwd_scale(grad, acc, gout, weight: float) -> torch.Tensor:
    """-> ``grad * weight * gout / count`` as a NEW tensor (the saved unscaled gradient stays intact for a second backward)."""
    _chk(grad, "grad"), _chk(acc, "acc"), _chk(gout, "grad_out")
    out = torch.empty_like(grad)
    _call("mvs_ce_loss_bwd_scale", "ce_loss_bwd_scale", _ptr(grad), _ptr(out), grad.numel(), _ptr(acc), _ptr(gout), float(weight), _stream())
    return out


def bf16_embed_ch0(x: torch.Tensor) -> torch.Tensor:
    """fp32 ``[...]`` -> bf16 ``[..., 8]`` with the value in channel 0, zeros elsewhere (one launch instead of a fill and a strided copy)."""
    _chk(x, "x")
    out = torch.empty(tuple(x.shape) + (8,), device=x.device, dtype=torch.bfloat16)
    _call("mvs_bf16_embed_ch0", "bf16_embed_ch0", _ptr(x), _ptr(out), x.numel(), _stream())
    return out


# ------------------------------------------------------------------ bf16 channel-last regularizer (training under autocast)
def _chk16(t: torch.Tensor, name: str) -> torch.Tensor:
    return _chk(t, name, dtype=torch.bfloat16)


def bf16_from_f32(x: torch.Tensor) -> torch.Tensor:
    """fp32 ``[B,C,D,H,W]`` -> bf16 channel-last ``[B,D,H,W,C]``."""
    _chk(x, "x")
    B, C = x.shape[0], x.shape[1]
    y = torch.empty((B,) + tuple(x.shape[2:]) + (C,), device=x.device, dtype=torch.bfloat16)
    _call("mvs_bf16_from_f32_ncdhw", "bf16_from_f32", _ptr(x), _ptr(y), B, C, x.numel() // (B * C), _stream())
    return y


def bf16_to_f32(x: torch.Tensor) -> torch.Tensor:
    """bf16 channel-last ``[B,D,H,W,C]`` -> fp32 ``[B,C,D,H,W]``."""
    _chk16(x, "x")
    B, C = x.shape[0], x.shape[-1]
    y = torch.empty((B, C) + tuple(x.shape[1:-1]), device=x.device, dtype=torch.float32)
    _call("mvs_bf16_to_f32_ncdhw", "bf16_to_f32", _ptr(x), _ptr(y), B, C, x.numel() // (B * C), _stream())
    return y


def _taps_of(weight: torch.Tensor) -> int:
    """27 for a ``[*,*,3,3,3]`` parameter, 9 for a 2-D ``[*,*,3,3]`` one (runs as the centre depth tap of a D = 1 volume)."""
    if weight.dim() == 5 and tuple(weight.shape[2:]) == (3, 3, 3):
        return 27
    if weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3):
        return 9
    raise _lib.MvsHipError("conv weight must be [*,*,3,3,3] or [*,*,3,3], got %s" % (tuple(weight.shape),))


def bf16_packed_elems(cin: int, cout: int, taps: int = 27) -> int:
    n = _lib.load().mvs_bf16_packed_elems_taps(cin, cout, taps)
    if n <= 0:
        raise _lib.MvsHipError("bf16 conv: channels must be 8/16/32/64 (Cin=%d Cout=%d, taps %d)" % (cin, cout, taps))
    return int(n)


def bf16_pack(weight: torch.Tensor, src: int, cin: int, cout: int) -> torch.Tensor:
    """fp32 ``[d0,d1,3,3,3]`` parameter -> bf16 MFMA fragments of a ``cin -> cout`` map (``src``: see include/mvs_hip.h)."""
    _chk(weight, "conv weight")
    if _taps_of(weight) != 27:
        raise _lib.MvsHipError("bf16_pack: 3x3x3 weights only (2-D kernels are packed through a PackTable)")
    packed = torch.empty(bf16_packed_elems(cin, cout), device=weight.device, dtype=torch.bfloat16)
    _call("mvs_bf16_pack_weights", None, _ptr(weight), weight.shape[0], weight.shape[1], int(src), cout, cin, _ptr(packed), _stream())
    return packed


def bf16_pack2(weight: torch.Tensor, a, b):
    """Two layouts ``(src, cin, cout)`` of one weight in one launch (forward + data gradient): -> (packed_a, packed_b)."""
    _chk(weight, "conv weight")
    if _taps_of(weight) != 27:
        raise _lib.MvsHipError("bf16_pack2: 3x3x3 weights only (2-D kernels are packed through a PackTable)")
    outs = [torch.empty(bf16_packed_elems(cin, cout), device=weight.device, dtype=torch.bfloat16) for _, cin, cout in (a, b)]
    _call("mvs_bf16_pack_weights2", "mvs_bf16_pack_weights", _ptr(weight), weight.shape[0], weight.shape[1], int(a[0]), a[2], a[1],
          _ptr(outs[0]), int(b[0]), b[2], b[1], _ptr(outs[1]), _stream())
    return outs[0], outs[1]


class PackTable:
    """A table of weight-packing jobs ``(weight, src, cin, cout)`` - fp32 parameter ``[d0,d1,3,3,3]`` or 2-D ``[d0,d1,3,3]`` -> the bf16
    per-lane MFMA fragments of a ``cin -> cout`` map (rows / channels beyond the parameter's own extents pack as zeros) - run by ONE
    launch (``mvs_bf16_pack_table_run``).  Built once (host table -> device, output buffers allocated), run every training step: the
    parameters are updated in place by the optimizer, so their addresses - which the device table holds - do not change;
    :meth:`valid` tells whether they still are the ones the table was built for."""

    def __init__(self, jobs):
        import ctypes
        lib = _lib.load()
        n = len(jobs)
        dev = jobs[0][0].device
        self.weights = [j[0] for j in jobs]
        for w in self.weights:
            _chk(w, "conv weight")
        self.ptrs = tuple(w.data_ptr() for w in self.weights)
        self.outs = []
        nbytes = lib.mvs_bf16_pack_table_bytes(n)
        host = (ctypes.c_uint8 * nbytes)()
        for i, (w, src, cin, cout) in enumerate(jobs):
            taps = _taps_of(w)
            out = torch.empty(bf16_packed_elems(cin, cout, taps), device=dev, dtype=torch.bfloat16)
            self.outs.append(out)
            _lib.check(lib.mvs_bf16_pack_table_fill(host, n, i, _ptr(w), w.shape[0], w.shape[1], int(src), cout, cin, taps, _ptr(out)),
                       "mvs_bf16_pack_table_fill")
        starts = (ctypes.c_int32 * (n + 1)).from_buffer(host, nbytes - 4 * (n + 1))
        self.blocks = int(starts[n])
        self.n = n
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise _lib.MvsHipError("PackTable built inside a hipGraph capture: run the step eagerly once")
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
        torch.cuda.current_stream(dev).synchronize()         # the pageable host buffer is gone after this constructor

    def valid(self) -> bool:
        return all(w.data_ptr() == p for w, p in zip(self.weights, self.ptrs))

    def run(self):
        _call("mvs_bf16_pack_table_run", "mvs_bf16_pack_weights", _ptr(self.table), self.n, self.blocks, _stream())
        for o in self.outs:                                  # written through raw pointers: tell autograd, so that a graph that SAVED an earlier
            torch.autograd.graph.increment_version(o)        # packing (retain_graph / interleaved forwards) raises instead of using new weights
        return self.outs


def bf16_conv3d(x, wpacked, cin, cout, gather: int, stride, scale=None, shift=None, residual=None, relu=False, taps: int = 27):
    """``x [B,D,H,W,cin]`` bf16 -> ``[B,Do,Ho,Wo,cout]`` bf16; ``gather`` 0 = Conv3d, 1 = ConvTranspose3d (k3, p1, op = stride-1).
    ``taps`` = 9: a 2-D kernel (no epilogue arguments then)."""
    _chk16(x, "x"), _chk16(wpacked, "packed weights"), _opt(scale, "scale"), _opt(shift, "shift")
    B, Di, Hi, Wi, C = x.shape
    assert C == cin
    sd, shw = stride
    if gather == 0:
        Do, Ho, Wo = (Di - 1) // sd + 1, (Hi - 1) // shw + 1, (Wi - 1) // shw + 1
    else:
        Do, Ho, Wo = Di * sd, Hi * shw, Wi * shw
    y = torch.empty(B, Do, Ho, Wo, cout, device=x.device, dtype=torch.bfloat16)
    if residual is not None:
        _chk16(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s" % (tuple(residual.shape), tuple(y.shape)))
    # credited with the kernel's useful FLOPs: every output voxel of a Conv3d has 27 taps, of a ConvTranspose3d 27 per INPUT voxel
    flops = 2.0 * 27 * cin * cout * B * (Do * Ho * Wo if gather == 0 else Di * Hi * Wi)
    if taps != 27:
        assert scale is None and shift is None and residual is None and not relu
        tag = ("bf16_conv_kernel<%d,%d,g%d,s%d%d,2d>" % (cin, cout, gather, sd, shw), "flops", flops * taps / 27.0)
        _call("mvs_bf16_conv3d_taps", tag, _ptr(x), _ptr(wpacked), _ptr(y), B, cin, cout, Di, Hi, Wi, int(gather), sd, shw, int(taps), _stream())
        return y
    tag = ("bf16_conv_kernel<%d,%d,g%d,s%d%d>" % (cin, cout, gather, sd, shw), "flops", flops)
    _call("mvs_bf16_conv3d", tag, _ptr(x), _ptr(wpacked), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), B, cin, cout, Di, Hi, Wi,
          int(gather), sd, shw, int(relu), _stream())
    return y


def bf16_conv3d_bnbwd(x, wpacked, cin, cout, gather: int, stride, bn_y, bn4, relu, groups: int = 1, taps: int = 27, addend=None):
    """Raw convolution whose output is the gradient arriving at a BatchNorm(+ReLU) layer (the data gradient of the layer after it) + that
    BatchNorm's backward sums from the convolution's epilogue -> ``(dz, sums [2*groups*cout])``; ``bn_y`` = the BatchNorm's input,
    ``bn4 [4, groups*cout]`` = the forward's (scale, shift, mean, invstd).  ``addend``: the tensor's other gradient (it also fed a skip
    connection), added before the rounding and the sums."""
    _chk16(x, "x"), _chk16(wpacked, "packed weights"), _chk16(bn_y, "bn_y"), _chk(bn4, "bn4")
    B, Di, Hi, Wi, C = x.shape
    assert C == cin
    sd, shw = stride
    if gather == 0:
        Do, Ho, Wo = (Di - 1) // sd + 1, (Hi - 1) // shw + 1, (Wi - 1) // shw + 1
    else:
        Do, Ho, Wo = Di * sd, Hi * shw, Wi * shw
    y = torch.empty(B, Do, Ho, Wo, cout, device=x.device, dtype=torch.bfloat16)
    if tuple(bn_y.shape) != tuple(y.shape) or bn4.dim() != 2 or bn4.shape[0] < 4 or bn4.shape[1] != groups * cout or B % groups:
        raise _lib.MvsHipError("bf16_conv3d_bnbwd: BatchNorm input %s / statistics %s do not match the gradient %s (groups %d)" % (
            tuple(bn_y.shape), tuple(bn4.shape), tuple(y.shape), groups))
    if addend is not None:
        _chk16(addend, "addend")
        if tuple(addend.shape) != tuple(y.shape):
            raise _lib.MvsHipError("bf16_conv3d_bnbwd: addend %s does not match the gradient %s" % (tuple(addend.shape), tuple(y.shape)))
    sums = torch.empty(2 * groups * cout, device=x.device, dtype=torch.float32)
    ws = _reduce_ws("mvs_bf16_conv3d_bn_fwd_workspace_bytes", x.device, B, cout, Do, Ho, Wo)
    flops = 2.0 * taps * cin * cout * B * (Do * Ho * Wo if gather == 0 else Di * Hi * Wi)
    tag = ("bf16_conv_kernel<%d,%d,g%d,s%d%d%s>" % (cin, cout, gather, sd, shw, ",2d" if taps == 9 else ""), "flops", flops)
    _call("mvs_bf16_conv3d_bnbwd", tag, _ptr(x), _ptr(wpacked), _ptr(y), B, cin, cout, Di, Hi, Wi, int(gather), sd, shw, int(taps), _ptr(bn_y),
          _ptr(bn4), int(relu), int(groups), _ptr(addend), _ptr(sums), _ptr(ws), _stream())
    return y, sums


def bf16_conv3d_stats(x, wpacked, cin, cout, gather: int, stride, groups: int = 1):
    """Raw convolution + the batch statistics of its bf16-rounded output in one pass -> ``(y, sums [2*groups*cout])``: the sums are
    what :func:`bf16_bn_stats` ``(y, groups)`` would return, without the extra pass over ``y``."""
    _chk16(x, "x"), _chk16(wpacked, "packed weights")
    B, Di, Hi, Wi, C = x.shape
    assert C == cin
    sd, shw = stride
    if gather == 0:
        Do, Ho, Wo = (Di - 1) // sd + 1, (Hi - 1) // shw + 1, (Wi - 1) // shw + 1
    else:
        Do, Ho, Wo = Di * sd, Hi * shw, Wi * shw
    if B % groups:
        raise _lib.MvsHipError("grouped statistics: batch %d is not a multiple of %d groups" % (B, groups))
    y = torch.empty(B, Do, Ho, Wo, cout, device=x.device, dtype=torch.bfloat16)
    sums = torch.empty(2 * groups * cout, device=x.device, dtype=torch.float32)
    ws = _reduce_ws("mvs_bf16_conv3d_stats_workspace_bytes", x.device, B, cout, Do, Ho, Wo)
    flops = 2.0 * 27 * cin * cout * B * (Do * Ho * Wo if gather == 0 else Di * Hi * Wi)
    tag = ("bf16_conv_kernel<%d,%d,g%d,s%d%d>" % (cin, cout, gather, sd, shw), "flops", flops)
    _call("mvs_bf16_conv3d_stats", tag, _ptr(x), _ptr(wpacked), _ptr(y), B, cin, cout, Di, Hi, Wi, int(gather), sd, shw, int(groups),
          _ptr(sums), _ptr(ws), _stream())
    return y, sums


def bf16_conv3d_wgrad(A: torch.Tensor, Bt: torch.Tensor, stride, taps: int = 27, cb_out: Optional[int] = None) -> torch.Tensor:
    """``dW[a][b][taps] = sum A[p][a] * Bt[p*s-1+k][b]`` (fp32); ``A [N,Dp,Hp,Wp,CA]`` lives on the grid the stride divides.  ``taps`` = 9:
    only the centre depth tap, ``dW [CA,cb_out,3,3]`` (a 2-D kernel's gradient); ``cb_out`` < CB drops ``Bt``'s padding channels."""
    _chk16(A, "A"), _chk16(Bt, "Bt")
    N, Dp, Hp, Wp, CA = A.shape
    _, Db, Hb, Wb, CB = Bt.shape
    cb_out = CB if cb_out is None else int(cb_out)
    dW = torch.empty((CA, cb_out, 3, 3, 3) if taps == 27 else (CA, cb_out, 3, 3), device=A.device, dtype=torch.float32)
    nws = _lib.load().mvs_bf16_conv3d_wgrad_workspace_bytes(N, CA, CB, Dp, Hp, Wp)
    if nws <= 0:
        raise _lib.MvsHipError("bf16 wgrad: channels must be 8/16/32/64 (CA=%d CB=%d)" % (CA, CB))
    ws = torch.empty(nws, device=A.device, dtype=torch.uint8)
    name = "bf16_wgrad_kernel" if os.environ.get("MVS_TAG_SHAPES", "0") != "1" else \
        "bf16_wgrad<%d,%d,s%d%d,%dx%dx%dx%d>" % (CA, CB, stride[0], stride[1], N, Dp, Hp, Wp)
    tag = (name, "flops", 2.0 * taps * CA * CB * N * Dp * Hp * Wp)
    _call("mvs_bf16_conv3d_wgrad_taps", tag, _ptr(A), _ptr(Bt), _ptr(dW), _ptr(ws), N, CA, CB, cb_out, Dp, Hp, Wp, Db, Hb, Wb, stride[0],
          stride[1], int(taps), _stream())
    return dW


def bf16_wgrad_shape(A: torch.Tensor, Bt: torch.Tensor, taps: int = 27, cb_out: Optional[int] = None):
    cb = Bt.shape[-1] if cb_out is None else int(cb_out)
    return (A.shape[-1], cb, 3, 3, 3) if taps == 27 else (A.shape[-1], cb, 3, 3)


def bf16_wgrad_group(jobs) -> None:
    """The weight gradients of several layers in one call (``mvs_bf16_wgrad_group``): one grid per kernel instance with all its jobs side
    by side + one reduce.  ``jobs``: ``(A, Bt, dW, stride, taps, cb_out)`` as :func:`bf16_conv3d_wgrad` takes them, ``dW`` the fp32
    tensor of :func:`bf16_wgrad_shape` to fill.  Bit-identical to the layer-by-layer calls."""
    if not jobs:
        return
    arr = (_lib.WgradJob * len(jobs))()
    flops = 0.0
    for k, (A, Bt, dW, stride, taps, cb_out) in enumerate(jobs):
        _chk16(A, "A"), _chk16(Bt, "Bt"), _chk(dW, "dW")
        N, Dp, Hp, Wp, CA = A.shape
        _, Db, Hb, Wb, CB = Bt.shape
        cb = CB if cb_out is None else int(cb_out)
        if tuple(dW.shape) != bf16_wgrad_shape(A, Bt, taps, cb):
            raise _lib.MvsHipError("bf16_wgrad_group: dW %s for operands %s x %s" % (tuple(dW.shape), tuple(A.shape), tuple(Bt.shape)))
        j = arr[k]
        j.A, j.Bt, j.dW = _ptr(A), _ptr(Bt), _ptr(dW)
        j.nbatch, j.CA, j.CB, j.CBout, j.Dp, j.Hp, j.Wp, j.Db, j.Hb, j.Wb = N, CA, CB, cb, Dp, Hp, Wp, Db, Hb, Wb
        j.sd, j.shw, j.taps, j.reserved = int(stride[0]), int(stride[1]), int(taps), 0
        flops += 2.0 * taps * CA * CB * N * Dp * Hp * Wp
    ptr = ctypes.cast(arr, ctypes.c_void_p)
    nws = _lib.load().mvs_bf16_wgrad_group_workspace_bytes(ptr, len(jobs))
    if nws <= 0:
        raise _lib.MvsHipError("bf16_wgrad_group: a job's channels / stride / taps are not built")
    dev = jobs[0][0].device
    ws = torch.empty(nws, device=dev, dtype=torch.uint8)
    _call("mvs_bf16_wgrad_group", ("bf16_wgrad_kernel", "flops", flops), ptr, len(jobs), _ptr(ws), int(nws), _stream())


def bf16_head_fwd(x: torch.Tensor, w: Optional[torch.Tensor], bias: Optional[torch.Tensor], sigmoid: bool) -> torch.Tensor:
    """``x [..., 8]`` bf16 channel-last -> fp32 ``[...]``: ``act(sum_c w[c]*x[c] + bias)``; ``w`` None selects channel 0."""
    _chk16(x, "x"), _opt(w, "w"), _opt(bias, "bias")
    if x.shape[-1] != 8:
        raise _lib.MvsHipError("bf16 head: 8 input channels, got %d" % x.shape[-1])
    out = torch.empty(x.shape[:-1], device=x.device, dtype=torch.float32)
    _call("mvs_bf16_head_fwd", "bf16_head_fwd", _ptr(x), _ptr(w), _ptr(bias), int(sigmoid), out.numel(), _ptr(out), _stream())
    return out


def bf16_head_bwd(x: torch.Tensor, w: Optional[torch.Tensor], y: Optional[torch.Tensor], dout: torch.Tensor):
    """-> ``(dx bf16 [..., 8], dwb [9] = [dw | dbias] or None when w is None)``; ``y`` = the forward's output if it applied the sigmoid."""
    _chk16(x, "x"), _opt(w, "w"), _opt(y, "y"), _chk(dout, "dout")
    N = dout.numel()
    dx = torch.empty_like(x)
    dwb = ws = None
    if w is not None:
        dwb = torch.empty(9, device=x.device, dtype=torch.float32)
        ws = torch.empty(max(1, _lib.load().mvs_bf16_head_bwd_workspace_bytes(N) // 4), device=x.device, dtype=torch.float32)
    _call("mvs_bf16_head_bwd", "bf16_head_bwd", _ptr(x), _ptr(w), _ptr(y), _ptr(dout), N, _ptr(dx), _ptr(dwb), _ptr(ws), _stream())
    return dx, dwb


def _bf16_bn_shape(x: torch.Tensor, groups: int):
    """(C, R, rows per sample) of a channel-last bf16 batch ``[N,...,C]``; ``groups`` > 1 = grouped BatchNorm over the batch dimension
    (sample n belongs to group n % groups)."""
    C = x.shape[-1]
    R = x.numel() // C
    if groups > 1 and x.shape[0] % groups:
        raise _lib.MvsHipError("grouped BatchNorm: batch %d is not a multiple of %d groups" % (x.shape[0], groups))
    return C, R, R // x.shape[0]


def bf16_bn_stats(x: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """-> ``sums [2*groups*C]``: [sum | sum of squares] per (group, channel)."""
    _chk16(x, "x")
    C, R, rps = _bf16_bn_shape(x, groups)
    sums = torch.empty(2 * groups * C, device=x.device, dtype=torch.float32)
    ws = _reduce_ws("mvs_bf16_bn_reduce_workspace_bytes", x.device, C, R, groups, rps)
    _call("mvs_bf16_bn_stats", "bf16_bn_stats", _ptr(x), C, R, groups, rps, _ptr(sums), _ptr(ws), _stream())
    return sums


def bf16_bn_train_fwd(x, residual, relu, gamma, beta, running_mean, running_var, momentum, eps, groups: int = 1, num_batches_tracked=None):
    """Statistics + finalize (running-stat update) + normalize / ReLU / skip in one call -> ``(y, scale, shift, mean, invstd)``."""
    _chk16(x, "x"), _opt(gamma, "bn.weight"), _opt(beta, "bn.bias"), _opt(running_mean, "bn.running_mean"), _opt(running_var, "bn.running_var")
    if residual is not None:
        _chk16(residual, "residual")
    if num_batches_tracked is not None:
        _chk(num_batches_tracked, "bn.num_batches_tracked", dtype=torch.int64)
    C, R, rps = _bf16_bn_shape(x, groups)
    y = torch.empty_like(x)
    st = torch.empty(5, groups * C, device=x.device, dtype=torch.float32)      # scale | shift | mean | invstd | gamma per (group, channel)
    ws = _reduce_ws("mvs_bf16_bn_reduce_workspace_bytes", x.device, C, R, groups, rps)
    if running_mean is not None:
        bump_weights_epoch()                                 # running statistics are written through raw pointers
    _call("mvs_bf16_bn_train_fwd", "bf16_bn_train_fwd", _ptr(x), _ptr(residual), int(relu), C, R, groups, rps, _ptr(gamma), _ptr(beta),
          _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), _ptr(num_batches_tracked), _ptr(st), _ptr(y), _ptr(ws), _stream())
    return y, st[0], st[1], st[2], st[3]


def bf16_affine_act(x, scale, shift, residual, relu, groups: int = 1):
    _chk16(x, "x"), _chk(scale, "scale"), _chk(shift, "shift")
    if residual is not None:
        _chk16(residual, "residual")
    C, R, rps = _bf16_bn_shape(x, groups)
    y = torch.empty_like(x)
    _call("mvs_bf16_affine_act", "bf16_affine_act", _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), int(relu), C, R, groups, rps, _ptr(y),
          _stream())
    return y


def bf16_bn_bwd_reduce(dy, x, scale, shift, mean, invstd, relu, groups: int = 1):
    _chk16(dy, "dy"), _chk16(x, "x")
    C, R, rps = _bf16_bn_shape(x, groups)
    sums = torch.empty(2 * groups * C, device=x.device, dtype=torch.float32)
    ws = _reduce_ws("mvs_bf16_bn_reduce_workspace_bytes", x.device, C, R, groups, rps)
    _call("mvs_bf16_bn_bwd_reduce", "bf16_bn_bwd_reduce", _ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), int(relu), C,
          R, groups, rps, _ptr(sums), _ptr(ws), _stream())
    return sums


def bf16_conv3d_bn_fwd(x, wpacked, cin, cout, gather: int, stride, residual, relu, gamma, beta, running_mean, running_var, momentum, eps,
                       groups: int = 1, num_batches_tracked=None, taps: int = 27):
    """conv -> batch-statistics BatchNorm -> [ReLU] [+ residual] of one bf16 channel-last training layer in one call (three launches, the
    statistics ride the convolution's epilogue) -> ``(y raw conv output, z, stats [5, groups*cout] = scale | shift | mean | invstd | gamma per (group, channel))``."""
    _chk16(x, "x"), _chk16(wpacked, "packed weights")
    _opt(gamma, "bn.weight"), _opt(beta, "bn.bias"), _opt(running_mean, "bn.running_mean"), _opt(running_var, "bn.running_var")
    B, Di, Hi, Wi, C = x.shape
    assert C == cin
    sd, shw = stride
    if gather == 0:
        Do, Ho, Wo = (Di - 1) // sd + 1, (Hi - 1) // shw + 1, (Wi - 1) // shw + 1
    else:
        Do, Ho, Wo = Di * sd, Hi * shw, Wi * shw
    if B % groups:
        raise _lib.MvsHipError("grouped statistics: batch %d is not a multiple of %d groups" % (B, groups))
    y = torch.empty(B, Do, Ho, Wo, cout, device=x.device, dtype=torch.bfloat16)
    z = torch.empty_like(y)
    if residual is not None:
        _chk16(residual, "residual")
        if residual.shape != y.shape:
            raise _lib.MvsHipError("residual shape %s != output %s" % (tuple(residual.shape), tuple(y.shape)))
    if num_batches_tracked is not None:
        _chk(num_batches_tracked, "bn.num_batches_tracked", dtype=torch.int64)
    st = torch.empty(5, groups * cout, device=x.device, dtype=torch.float32)
    ws = _reduce_ws("mvs_bf16_conv3d_bn_fwd_workspace_bytes", x.device, B, cout, Do, Ho, Wo)
    flops = 2.0 * taps * cin * cout * B * (Do * Ho * Wo if gather == 0 else Di * Hi * Wi)
    tag = ("bf16_conv_bn_fwd<%d,%d,g%d,s%d%d%s>" % (cin, cout, gather, sd, shw, ",2d" if taps == 9 else ""), "flops", flops)
    if running_mean is not None:
        bump_weights_epoch()                                 # running statistics are written through raw pointers
    _call("mvs_bf16_conv3d_bn_fwd", tag, _ptr(x), _ptr(wpacked), _ptr(y), _ptr(z), _ptr(residual), int(relu), B, cin, cout, Di, Hi, Wi,
          int(gather), sd, shw, int(taps), int(groups), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
          _ptr(num_batches_tracked), _ptr(st), _ptr(ws), _stream())
    return y, z, st


def bf16_bn_bwd_apply(dy, x, scale, shift, mean, invstd, gamma, sums, count, relu, count_dev=None, groups: int = 1, want_dgb: bool = False):
    """``want_dgb``: also return ``dgb [2*C] = [dbeta | dgamma]`` of a grouped BatchNorm's shared parameters (the groups' sums added by the
    kernel's first block: no separate reduction launch) -> ``(dx, dgb)``."""
    _chk16(dy, "dy"), _chk16(x, "x"), _chk(sums, "sums"), _opt(gamma, "gamma"), _opt(count_dev, "count_dev")
    C, R, rps = _bf16_bn_shape(x, groups)
    dx = torch.empty_like(x)
    if want_dgb:
        dgb = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        _call("mvs_bf16_bn_bwd_apply_dgb", "bf16_bn_bwd_apply", _ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _ptr(gamma),
              _ptr(sums), float(count), _ptr(count_dev), int(relu), C, R, groups, rps, _ptr(dx), _ptr(dgb), _stream())
        return dx, dgb
    _call("mvs_bf16_bn_bwd_apply", "bf16_bn_bwd_apply", _ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _ptr(gamma),
          _ptr(sums), float(count), _ptr(count_dev), int(relu), C, R, groups, rps, _ptr(dx), _stream())
    return dx


# ----------------------------------------------------------------------------------------------- FPN decoder (row before the path)
FPN_CH = 64


def fpn_pack_weights(w: torch.Tensor) -> torch.Tensor:
    """``out_k.0.weight [Cout,64,3,3]`` (Cout 8 | 16 | 32) -> the MFMA-fragment image ``mvs_fpn_level`` stages through LDS."""
    _chk(w, "fpn 3x3 weight")
    Cout = w.shape[0]
    if w.shape[1:] != (FPN_CH, 3, 3) or Cout not in (8, 16, 32):
        raise _lib.MvsHipError("fpn 3x3 weight must be [8|16|32,64,3,3], got %s" % (tuple(w.shape),))
    packed = torch.empty(int(_lib.load().mvs_fpn_packed_floats(Cout)), device=w.device, dtype=torch.float32)
    _call("mvs_fpn_pack_weights", None, _ptr(w), Cout, _ptr(packed), _stream())
    return packed


def fpn_out0(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """``conv31 [N,64,h,w]`` -> ``Swish(BN(conv1x1)) [N,h,w,64]`` channel-last (reference models/module.py:246,259)."""
    _chk(x, "conv31"), _chk(w, "out0 weight"), _chk(scale, "scale"), _chk(shift, "shift")
    N, C, h, wd = x.shape
    if C != FPN_CH or w.numel() != FPN_CH * FPN_CH or scale.numel() != FPN_CH or shift.numel() != FPN_CH:
        raise _lib.MvsHipError("fpn_out0: expects 64 channels, got x %s w %s" % (tuple(x.shape), tuple(w.shape)))
    out = torch.empty(N, h, wd, FPN_CH, device=x.device, dtype=torch.float32)
    tag = ("fpn_out0_kernel", "flops", 2.0 * FPN_CH * FPN_CH * N * h * wd)
    _call("mvs_fpn_out0", tag, _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), N, h, wd, _ptr(out), _stream())
    return out


def fpn_level(intra_prev: torch.Tensor, lateral: torch.Tensor, w_inner_p: torch.Tensor, b_inner: torch.Tensor, packed: torch.Tensor,
              scale: torch.Tensor, shift: torch.Tensor, want_intra: bool, intra_nhwc: bool = False):
    """One top-down level (models/module.py:262-268): ``(intra_out [N,64,2h,2w] | None, out [N,2h,2w,Ck] channel-last)``.
    ``w_inner_p`` is ``inner_k.weight [64,Ck]`` regrouped by output-channel pair: ``[32,Ck,2]`` (include/mvs_hip.h)."""
    _chk(intra_prev, "intra_prev"), _chk(lateral, "lateral"), _chk(w_inner_p, "inner weight"), _chk(b_inner, "inner bias")
    _chk(packed, "packed weights"), _chk(scale, "scale"), _chk(shift, "shift")
    N, C, h, w = intra_prev.shape
    Ck = lateral.shape[1]
    if C != FPN_CH or lateral.shape != (N, Ck, 2 * h, 2 * w):
        raise _lib.MvsHipError("fpn_level: intra_prev %s needs a lateral [N,Ck,2h,2w], got %s" % (tuple(intra_prev.shape), tuple(lateral.shape)))
    if w_inner_p.shape != (FPN_CH // 2, Ck, 2) or b_inner.numel() != FPN_CH or scale.numel() != Ck or shift.numel() != Ck:
        raise _lib.MvsHipError("fpn_level: parameter sizes do not match Ck=%d" % Ck)
    if Ck not in (8, 16, 32) or packed.numel() != int(_lib.load().mvs_fpn_packed_floats(Ck)):
        raise _lib.MvsHipError("fpn_level: Ck=%d / packed weights of %d floats are not a supported pair" % (Ck, packed.numel()))
    intra = None
    if want_intra:                                           # ``intra_nhwc``: [N,2h,2w,64] memory (what ``fpn_level_cp`` reads), else the reference's [N,64,2h,2w]
        intra = torch.empty((N, 2 * h, 2 * w, FPN_CH) if intra_nhwc else (N, FPN_CH, 2 * h, 2 * w), device=lateral.device, dtype=torch.float32)
    out = torch.empty(N, 2 * h, 2 * w, Ck, device=lateral.device, dtype=torch.float32)
    tag = ("fpn_level_kernel<%d>" % Ck, "flops", 2.0 * FPN_CH * Ck * 10 * N * 4 * h * w)
    _call("mvs_fpn_level_layout", tag, _ptr(intra_prev), _ptr(lateral), _ptr(w_inner_p), _ptr(b_inner), _ptr(packed), _ptr(scale), _ptr(shift),
          N, Ck, h, w, _ptr(intra), 1 if (want_intra and intra_nhwc) else 0, _ptr(out), _stream())
    return intra, out


def fpn_level_x3s_prepare(w3: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """``out_k.0.weight [Ck,64,3,3]`` (Ck 16 | 32) with the folded BatchNorm scale -> the pre-split MFMA fragments of ``fpn_level_x3s``."""
    _chk(w3, "fpn 3x3 weight"), _chk(scale, "scale")
    Ck = w3.shape[0]
    n = int(_lib.load().mvs_fpn_level_x3s_prepared_bytes(Ck))
    if n <= 0 or w3.shape != (Ck, FPN_CH, 3, 3) or scale.numel() != Ck:
        raise _lib.MvsHipError("fpn_level_x3s_prepare: unsupported weight %s" % (tuple(w3.shape),))
    prepared = torch.empty(n, device=w3.device, dtype=torch.uint8)
    _call("mvs_fpn_level_x3s_prepare", None, _ptr(w3.contiguous()), _ptr(scale), Ck, _ptr(prepared), _stream())
    return prepared


def fpn_level_x3s(intra_prev: torch.Tensor, lateral: torch.Tensor, w_inner_p: torch.Tensor, b_inner: torch.Tensor, prepared: torch.Tensor,
                  shift: torch.Tensor, want_intra: bool, intra_nhwc: bool = False):
    """``fpn_level`` for the levels with Ck = 16 | 32, the 3x3 convolution in split form (csrc/fpn_lvl_x3.hip): ``(intra | None, out)``."""
    _chk(intra_prev, "intra_prev"), _chk(lateral, "lateral"), _chk(w_inner_p, "inner weight"), _chk(b_inner, "inner bias")
    _chk(prepared, "prepared", torch.uint8), _chk(shift, "shift")
    N, C, h, w = intra_prev.shape
    Ck = lateral.shape[1]
    if C != FPN_CH or lateral.shape != (N, Ck, 2 * h, 2 * w) or w_inner_p.shape != (FPN_CH // 2, Ck, 2) or b_inner.numel() != FPN_CH or shift.numel() != Ck:
        raise _lib.MvsHipError("fpn_level_x3s: shapes do not match (intra_prev %s, lateral %s)" % (tuple(intra_prev.shape), tuple(lateral.shape)))
    if prepared.numel() != int(_lib.load().mvs_fpn_level_x3s_prepared_bytes(Ck)):
        raise _lib.MvsHipError("fpn_level_x3s: prepared weights do not match Ck=%d" % Ck)
    intra = None
    if want_intra:
        intra = torch.empty((N, 2 * h, 2 * w, FPN_CH) if intra_nhwc else (N, FPN_CH, 2 * h, 2 * w), device=lateral.device, dtype=torch.float32)
    out = torch.empty(N, 2 * h, 2 * w, Ck, device=lateral.device, dtype=torch.float32)
    tag = ("fpn_level_x3s_kernel<%d>" % Ck, "flops", 2.0 * FPN_CH * Ck * 10 * N * 4 * h * w)
    _call("mvs_fpn_level_x3s", tag, _ptr(intra_prev), _ptr(lateral), _ptr(w_inner_p), _ptr(b_inner), _ptr(prepared), _ptr(shift), N, Ck, h, w,
          _ptr(intra), 1 if (want_intra and intra_nhwc) else 0, _ptr(out), _stream())
    return intra, out


def fpn_level_x3_prepare(w3: torch.Tensor, w_inner: torch.Tensor, b_inner: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor):
    """Operands of ``fpn_level_x3`` from ``out_k.0.weight [Ck,64,3,3]``, ``inner_k.weight [64,Ck]`` / ``.bias [64]`` and the folded BatchNorm
    ``(scale, shift)``: the pre-split MFMA fragments, the shift including the inner bias' response, and the per-tap border table.  The
    composition (a [Ck,64,9] x [64,Ck] product, once per weight version) is done in float64."""
    _chk(w3, "fpn 3x3 weight"), _chk(w_inner, "inner weight"), _chk(b_inner, "inner bias"), _chk(scale, "scale"), _chk(shift, "shift")
    Ck = w3.shape[0]
    n = int(_lib.load().mvs_fpn_level_x3_prepared_bytes(Ck))
    if n <= 0 or w3.shape != (Ck, FPN_CH, 3, 3) or w_inner.numel() != FPN_CH * Ck:
        raise _lib.MvsHipError("fpn_level_x3_prepare: unsupported Ck=%d / weight %s" % (Ck, tuple(w3.shape)))
    w3d, wi = w3.double(), w_inner.reshape(FPN_CH, Ck).double()
    wc = torch.einsum("ochw,ci->oihw", w3d, wi).float().contiguous()                       # [Ck,Ck,3,3]
    resp = torch.einsum("ochw,c->ohw", w3d, b_inner.double()).reshape(Ck, 9) * scale.double()[:, None]    # scale * bias response per tap
    shift_x = (shift.double() + resp.sum(1)).float().contiguous()
    border = resp.t().float().contiguous()                                                  # [9][Ck]
    prepared = torch.empty(n, device=w3.device, dtype=torch.uint8)
    _call("mvs_fpn_level_x3_prepare", None, _ptr(w3.contiguous()), _ptr(wc), _ptr(scale), Ck, _ptr(prepared), _stream())
    prepared_cp = torch.empty(int(_lib.load().mvs_fpn_level_cp_prepared_bytes(Ck)), device=w3.device, dtype=torch.uint8)
    _call("mvs_fpn_level_cp_prepare", None, _ptr(w3.contiguous()), _ptr(wc), _ptr(scale), Ck, _ptr(prepared_cp), _stream())
    return prepared, shift_x, border, prepared_cp


def fpn_level_x3(intra_prev: torch.Tensor, lateral: torch.Tensor, prepared: torch.Tensor, shift_x: torch.Tensor, border: torch.Tensor) -> torch.Tensor:
    """The full-resolution top-down level in split form, strip kernel (csrc/fpn_x3.hip): NCHW sources, ``out [N,2h,2w,Ck]`` channel-last; the
    64-channel ``intra`` map is never written."""
    _chk(intra_prev, "intra_prev"), _chk(lateral, "lateral"), _chk(prepared, "prepared", torch.uint8), _chk(shift_x, "shift"), _chk(border, "border")
    N, C, h, w = intra_prev.shape
    Ck = lateral.shape[1]
    if C != FPN_CH or lateral.shape != (N, Ck, 2 * h, 2 * w):
        raise _lib.MvsHipError("fpn_level_x3: intra_prev %s needs a lateral [N,Ck,2h,2w], got %s" % (tuple(intra_prev.shape), tuple(lateral.shape)))
    if prepared.numel() != int(_lib.load().mvs_fpn_level_x3_prepared_bytes(Ck)) or shift_x.numel() != Ck or border.numel() != 9 * Ck:
        raise _lib.MvsHipError("fpn_level_x3: operands do not match Ck=%d" % Ck)
    out = torch.empty(N, 2 * h, 2 * w, Ck, device=lateral.device, dtype=torch.float32)
    tag = ("fpn8_x3_kernel", "flops", 2.0 * FPN_CH * Ck * 10 * N * 4 * h * w)
    _call("mvs_fpn_level_x3", tag, _ptr(intra_prev), _ptr(lateral), _ptr(prepared), _ptr(shift_x), _ptr(border), N, Ck, h, w, _ptr(out), _stream())
    return out


def fpn_level_cp(intra_prev_cl: torch.Tensor, lateral_cl: torch.Tensor, prepared_cp: torch.Tensor, shift_x: torch.Tensor, border: torch.Tensor) -> torch.Tensor:
    """The same level with the channel contraction in front of the upsampling (csrc/fpn_cp.hip).  Both sources CHANNEL-LAST:
    ``intra_prev_cl [N,h,w,64]``, ``lateral_cl [N,2h,2w,Ck]`` -> ``out [N,2h,2w,Ck]``."""
    _chk(intra_prev_cl, "intra_prev"), _chk(lateral_cl, "lateral"), _chk(prepared_cp, "prepared", torch.uint8), _chk(shift_x, "shift"), _chk(border, "border")
    N, h, w, C = intra_prev_cl.shape
    Ck = lateral_cl.shape[3]
    if C != FPN_CH or lateral_cl.shape != (N, 2 * h, 2 * w, Ck):
        raise _lib.MvsHipError("fpn_level_cp: intra_prev [N,h,w,64] %s needs a lateral [N,2h,2w,Ck], got %s" % (tuple(intra_prev_cl.shape), tuple(lateral_cl.shape)))
    if prepared_cp.numel() != int(_lib.load().mvs_fpn_level_cp_prepared_bytes(Ck)) or shift_x.numel() != Ck or border.numel() != 9 * Ck:
        raise _lib.MvsHipError("fpn_level_cp: operands do not match Ck=%d" % Ck)
    out = torch.empty(N, 2 * h, 2 * w, Ck, device=lateral_cl.device, dtype=torch.float32)
    tag = ("fpn8_cp_kernel", "flops", 2.0 * FPN_CH * Ck * 10 * N * 4 * h * w)
    _call("mvs_fpn_level_cp", tag, _ptr(intra_prev_cl), _ptr(lateral_cl), _ptr(prepared_cp), _ptr(shift_x), _ptr(border), N, Ck, h, w, _ptr(out), _stream())
    return out


# ----------------------------------------------------------------------------------------------- FPN encoder layers
def conv2d_x3s_supported(Cin: int, Cout: int, K: int, stride: int) -> bool:
    return bool(_lib.load().mvs_conv2d_x3s_supported(Cin, Cout, K, stride))


def conv2d_x3s_prepare(w: torch.Tensor, scale: torch.Tensor, stride: int) -> torch.Tensor:
    """``conv.weight`` of an encoder layer below full resolution with the folded BatchNorm scale -> the pre-split MFMA fragments of ``conv2d_x3s_bn_lrelu``."""
    _chk(w, "conv2d weight"), _chk(scale, "scale")
    Cout, Cin, K, K2 = w.shape
    n = int(_lib.load().mvs_conv2d_x3s_prepared_bytes(Cin, Cout, K, stride)) if K == K2 else -1
    if n <= 0 or scale.numel() != Cout:
        raise _lib.MvsHipError("conv2d_x3s_prepare: %s stride %d is not a layer of the FPN encoder below full resolution" % (tuple(w.shape), stride))
    prepared = torch.empty(n, device=w.device, dtype=torch.uint8)
    _call("mvs_conv2d_x3s_prepare", None, _ptr(w), _ptr(scale), Cin, Cout, K, stride, _ptr(prepared), _stream())
    return prepared


def conv2d_x3s_bn_lrelu(x: torch.Tensor, prepared: torch.Tensor, shift: torch.Tensor, Cout: int, K: int, stride: int, slope: float, x_nhwc: bool = False) -> torch.Tensor:
    """An encoder layer below full resolution in split form (csrc/conv2d_x3s.hip): fp32 in (``[N,Cin,H,W]``, or ``[N,H,W,8]`` with ``x_nhwc``), NCHW out."""
    _chk(x, "x"), _chk(prepared, "prepared", torch.uint8), _chk(shift, "shift")
    if x_nhwc:
        N, H, W, Cin = x.shape
    else:
        N, Cin, H, W = x.shape
    if prepared.numel() != int(_lib.load().mvs_conv2d_x3s_prepared_bytes(Cin, Cout, K, stride)) or shift.numel() != Cout:
        raise _lib.MvsHipError("conv2d_x3s_bn_lrelu: operands do not match (Cin,Cout,K,stride)=(%d,%d,%d,%d)" % (Cin, Cout, K, stride))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(N, Cout, Ho, Wo, device=x.device, dtype=torch.float32)
    tag = ("conv2d_x3s_kernel<%d,%d,%d,%d>" % (Cin, Cout, K, stride), "flops", 2.0 * K * K * Cin * Cout * N * Ho * Wo)
    _call("mvs_conv2d_x3s_bn_lrelu", tag, _ptr(x), 1 if x_nhwc else 0, _ptr(prepared), _ptr(shift), N, Cin, Cout, K, stride, H, W, float(slope), _ptr(y), _stream())
    return y


def conv2d_x3_supported(Cin: int, Cout: int, K: int, stride: int) -> bool:
    return bool(_lib.load().mvs_conv2d_x3_supported(Cin, Cout, K, stride))


def conv2d_x3_prepare(w: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """``conv.weight [8,Cin,K,K]`` of conv00 / conv01 with the folded BatchNorm scale -> the pre-split MFMA fragments of ``conv2d_x3_bn_lrelu``."""
    _chk(w, "conv2d weight"), _chk(scale, "scale")
    Cout, Cin, K, K2 = w.shape
    n = int(_lib.load().mvs_conv2d_x3_prepared_bytes(Cin, Cout, K)) if K == K2 else -1
    if n <= 0 or scale.numel() != Cout:
        raise _lib.MvsHipError("conv2d_x3_prepare: %s is not conv00 / conv01 of the FPN encoder" % (tuple(w.shape),))
    prepared = torch.empty(n, device=w.device, dtype=torch.uint8)
    _call("mvs_conv2d_x3_prepare", None, _ptr(w), _ptr(scale), Cin, Cout, K, _ptr(prepared), _stream())
    return prepared


def conv2d_x3_bn_lrelu(x: torch.Tensor, prepared: torch.Tensor, shift: torch.Tensor, Cout: int, K: int, slope: float, x_nhwc: bool = False,
                       out: str = "nchw"):
    """conv00 / conv01 in split form (csrc/conv2d_x3.hip): ``leaky_relu(BatchNorm_eval(conv2d(x)))``, fp32, fp32-equivalent.  ``x`` is
    ``[N,Cin,H,W]`` or, with ``x_nhwc`` (Cin = 8), ``[N,H,W,8]``; ``out``: "nchw" -> ``y [N,8,H,W]``, "nhwc" -> ``[N,H,W,8]``, "both" -> the pair."""
    _chk(x, "x"), _chk(prepared, "prepared", torch.uint8), _chk(shift, "shift")
    if x_nhwc:
        N, H, W, Cin = x.shape
    else:
        N, Cin, H, W = x.shape
    if prepared.numel() != int(_lib.load().mvs_conv2d_x3_prepared_bytes(Cin, Cout, K)) or shift.numel() != Cout or out not in ("nchw", "nhwc", "both"):
        raise _lib.MvsHipError("conv2d_x3_bn_lrelu: operands do not match (Cin,Cout,K)=(%d,%d,%d) / out=%r" % (Cin, Cout, K, out))
    y = torch.empty(N, Cout, H, W, device=x.device, dtype=torch.float32) if out != "nhwc" else None
    ycl = torch.empty(N, H, W, Cout, device=x.device, dtype=torch.float32) if out != "nchw" else None
    tag = ("enc_x3_kernel<%d,%d,%d>" % (Cin, Cout, K), "flops", 2.0 * K * K * Cin * Cout * N * H * W)
    _call("mvs_conv2d_x3_bn_lrelu_layout", tag, _ptr(x), 1 if x_nhwc else 0, _ptr(prepared), _ptr(shift), N, Cin, Cout, K, 1, H, W, float(slope),
          _ptr(y), _ptr(ycl), _stream())
    return (y, ycl) if out == "both" else (y if out == "nchw" else ycl)


def conv2d_pack_weights(w: torch.Tensor) -> torch.Tensor:
    """``conv.weight [Cout,Cin,K,K]`` of an FPN encoder layer -> the MFMA-fragment image ``mvs_conv2d_bn_lrelu`` stages through LDS."""
    _chk(w, "conv2d weight")
    Cout, Cin, K, K2 = w.shape
    n = int(_lib.load().mvs_conv2d_packed_floats(Cin, Cout, K)) if K == K2 else -1
    if n <= 0:
        raise _lib.MvsHipError("conv2d weight %s is not an FPN encoder layer shape" % (tuple(w.shape),))
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    _call("mvs_conv2d_pack_weights", None, _ptr(w), Cin, Cout, K, _ptr(packed), _stream())
    return packed


def conv2d_bn_lrelu(x: torch.Tensor, packed: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, cout: int, k: int, stride: int,
                    slope: float = 0.1) -> torch.Tensor:
    """``leaky_relu(BatchNorm2d_eval(conv2d(x, w, stride, padding=k//2)), slope)`` (reference models/module.py:40-73), NCHW."""
    _chk(x, "x"), _chk(packed, "packed weights"), _chk(scale, "scale"), _chk(shift, "shift")
    N, Cin, H, W = x.shape
    if scale.numel() != cout or shift.numel() != cout or packed.numel() != int(_lib.load().mvs_conv2d_packed_floats(Cin, cout, k)):
        raise _lib.MvsHipError("conv2d_bn_lrelu: parameter sizes do not match (Cin,Cout,K)=(%d,%d,%d)" % (Cin, cout, k))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(N, cout, Ho, Wo, device=x.device, dtype=torch.float32)
    tag = ("conv2d_kernel<%d,%d,%d,%d>" % (Cin, cout, k, stride), "flops", 2.0 * k * k * Cin * cout * N * Ho * Wo)
    _call("mvs_conv2d_bn_lrelu", tag, _ptr(x), _ptr(packed), _ptr(scale), _ptr(shift), N, Cin, cout, k, stride, H, W, float(slope), _ptr(y),
          _stream())
    return y


# ------------------------------------------------------------------ DINO ViT branch (csrc/vit.hip)
def gemm_x3(A, B, C, M, N, K, lda, ldb, ldc, nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), b_kn=False, a_mode=0, H=0, W=0, Cp=0, alpha=1.0,
            scale=None, shift=None, act=0, mul=None, res=None, a_off=0, b_off=0, c_off=0):
    """``C[b1][b2] = epi(alpha * A[b1][b2] . B[b1][b2]^T)`` on the bf16 matrix cores in three-term split form (fp32-equivalent); see
    ``mvs_gemm_x3`` in include/mvs_hip.h.  ``?_off``: element offsets into the tensors (a head's slice of a packed qkv row)."""
    for t, n in ((A, "A"), (B, "B"), (C, "C")):
        _chk(t, n)
    _opt(scale, "scale"), _opt(shift, "shift"), _opt(mul, "mul"), _opt(res, "res")
    flops = 2.0 * M * N * K * nb1 * nb2
    _call("mvs_gemm_x3", ("x3_gemm", "flops", flops), A.data_ptr() + 4 * a_off, B.data_ptr() + 4 * b_off, C.data_ptr() + 4 * c_off, M, N, K,
          lda, ldb, ldc, nb1, nb2, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], int(b_kn), int(a_mode), H, W, Cp, float(alpha), _ptr(scale),
          _ptr(shift), int(act), (mul.data_ptr() + 4 * c_off) if mul is not None else None, (res.data_ptr() + 4 * c_off) if res is not None else None,
          _stream())
    return C


def attention_x3(qkv: torch.Tensor, vt: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """``softmax(scale * Q K^T) V`` per (image, head) in flash form: ``qkv [B,N,3C]`` packed, ``vt [B,heads,64,ldv]`` = V transposed with a
    row stride ``ldv >= N`` that is a multiple of 4 (zero / finite padding; :func:`attention_vt` makes it) -> ``[B,N,C]``."""
    _chk(qkv, "qkv"), _chk(vt, "vt")
    B, N, C3 = qkv.shape
    C = C3 // 3
    if vt.dim() != 4 or vt.shape[0] != B or vt.shape[1] != heads or vt.shape[2] * heads != C or not vt.is_contiguous():
        raise _lib.MvsHipError("attention_x3: vt %s for qkv %s, %d heads" % (tuple(vt.shape), tuple(qkv.shape), heads))
    out = torch.empty(B, N, C, device=qkv.device, dtype=torch.float32)
    _call("mvs_attention_x3", ("x3_attention", "flops", 4.0 * B * heads * N * N * (C // heads)), _ptr(qkv), _ptr(vt), _ptr(out), B, N, heads,
          C // heads, int(vt.shape[3]), float(scale), _stream())
    return out


def attention_vt(qkv: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """V of packed ``qkv [B,N,3C]`` transposed to ``[B,heads,hd,ldv]``, ``ldv`` = N rounded up to a multiple of 4 with zero padding (the B
    operand of ``P V`` is then read along the keys, 16 bytes at a time).  ``out``: a buffer of that shape to reuse (its padding is kept)."""
    B, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    ldv = (N + 3) // 4 * 4
    if out is None:
        out = torch.zeros(B, heads, hd, ldv, device=qkv.device, dtype=torch.float32)
    out[..., :N].copy_(qkv[:, :, 2 * C:].reshape(B, N, heads, hd).permute(0, 2, 3, 1))
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    _chk(x, "x"), _chk(gamma, "gamma"), _chk(beta, "beta")
    y = torch.empty_like(x)
    _call("mvs_layernorm", "layernorm", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), x.numel() // x.shape[-1], x.shape[-1], float(eps), _stream())
    return y


def softmax_rows_(x: torch.Tensor, scale: float) -> torch.Tensor:
    """In place: ``softmax(scale * x)`` over the last axis."""
    _chk(x, "x")
    _call("mvs_softmax_rows", "softmax_rows", _ptr(x), _ptr(x), x.numel() // x.shape[-1], x.shape[-1], float(scale), _stream())
    return x


def bicubic_resize(x: torch.Tensor, Ho: int, Wo: int, rscale_h: float, rscale_w: float) -> torch.Tensor:
    """ATen's ``upsample_bicubic2d`` (align_corners=False) of ``[..., H, W]``; ``rscale`` = in / out, or 1 / scale_factor."""
    _chk(x, "x")
    H, W = x.shape[-2:]
    out = torch.empty(tuple(x.shape[:-2]) + (Ho, Wo), device=x.device, dtype=torch.float32)
    _call("mvs_bicubic_resize", "bicubic_resize", _ptr(x), _ptr(out), x.numel() // (H * W), H, W, Ho, Wo, float(rscale_h), float(rscale_w), _stream())
    return out


# ------------------------------------------------------------------ FPN training mode (csrc/vit.hip mvs_conv2d_gemm_x3, csrc/fpn_train.hip)
# ----------------------------------------------------------------------------------------------- pre-split ("packed") operands (csrc/vit_packed.hip)
class Packed:
    """A matrix ``[rows][K]`` stored split into the three bf16 terms and in MFMA fragment order (include/mvs_hip.h, ``mvs_x3p_*``):
    ``buf`` is the raw device buffer, ``rows`` the logical rows, ``rows_alloc`` (a multiple of 16) the rows it holds."""
    __slots__ = ("buf", "rows", "K", "rows_alloc")

    def __init__(self, rows: int, K: int, device, rows_alloc: Optional[int] = None, zero: bool = False):
        if K % 32:
            raise _lib.MvsHipError("Packed: K = %d is not a multiple of 32" % K)
        self.rows, self.K = int(rows), int(K)
        self.rows_alloc = int(rows_alloc) if rows_alloc is not None else (self.rows + 127) // 128 * 128
        n = int(_lib.load().mvs_x3p_bytes(self.rows_alloc, self.K))
        self.buf = (torch.zeros if zero else torch.empty)(n, device=device, dtype=torch.uint8)

    def ptr(self):
        return self.buf.data_ptr()


def x3p_pack(x: torch.Tensor, rows_alloc: Optional[int] = None) -> Packed:
    """fp32 ``[R, K]`` -> :class:`Packed` (rows beyond R zero)."""
    _chk(x, "x")
    R, K = x.shape
    out = Packed(R, K, x.device, rows_alloc)
    _call("mvs_x3p_pack", "x3p_pack", _ptr(x), out.ptr(), R, K, K, out.rows_alloc, _stream())
    return out


def x3p_pack_classes(w: torch.Tensor, rows_alloc: int) -> Packed:
    """``[classes, R, K]`` -> one buffer of ``classes`` packed matrices, ``rows_alloc`` rows each (the transposed convolution's parity classes)."""
    _chk(w, "w")
    ncls, R, K = w.shape
    out = Packed(ncls * rows_alloc, K, w.device, rows_alloc=ncls * rows_alloc)
    per = int(_lib.load().mvs_x3p_bytes(rows_alloc, K))
    for c in range(ncls):
        _call("mvs_x3p_pack", "x3p_pack", w[c].data_ptr(), out.ptr() + c * per, R, K, K, rows_alloc, _stream())
    out.rows, out.rows_alloc = R, rows_alloc                  # per class
    return out


def conv_x3p(X: Packed, W: Packed, mode: int, images: int, H: int, Wd: int, N: int, C: Optional[torch.Tensor] = None, scale=None, shift=None,
             act: int = 0, mul=None, out: Optional[Packed] = None) -> None:
    """Implicit 3x3 (``mode`` 1) / transposed 4x4 stride-2 (``mode`` 2) convolution over the packed channel-last map ``X`` ``[images*H*Wd, Cp]``
    (see ``mvs_conv_x3p``); ``X`` must hold at least one padding row beyond its pixels, all zeros (row ``X.rows``)."""
    M = images * H * Wd
    if X.rows != M or X.rows_alloc <= M:
        raise _lib.MvsHipError("conv_x3p: map of %d rows (%d allocated) for %d pixels + a zero row" % (X.rows, X.rows_alloc, M))
    taps = 9 if mode == 1 else 4
    if W.K != taps * X.K:
        raise _lib.MvsHipError("conv_x3p: weights K = %d for %d taps x %d channels" % (W.K, taps, X.K))
    if C is not None:
        _chk(C, "C")
    _opt(scale, "scale"), _opt(shift, "shift"), _opt(mul, "mul")
    _call("mvs_conv_x3p", ("x3p_gemm", "flops", 2.0 * M * N * W.K * (4 if mode == 2 else 1)), X.ptr(), X.rows_alloc, M, W.ptr(), W.rows_alloc, int(mode), images, H, Wd,
          X.K, N, _ptr(C), C.shape[-1] if C is not None else 0, _ptr(scale), _ptr(shift), int(act), _ptr(mul), out.ptr() if out is not None else None, _stream())


def x3p_unpack(p: Packed) -> torch.Tensor:
    x = torch.empty(p.rows, p.K, device=p.buf.device, dtype=torch.float32)
    _call("mvs_x3p_unpack", "x3p_unpack", p.ptr(), _ptr(x), p.rows, p.K, p.K, p.rows_alloc, _stream())
    return x


def layernorm_x3p(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, Np: int, N: int, out: Optional[Packed] = None) -> Packed:
    """LayerNorm of rows ``[images*Np, C]`` written packed (rows ``t >= N`` of an image: zeros)."""
    _chk(x, "x"), _chk(gamma, "gamma"), _chk(beta, "beta")
    rows, C = x.shape
    if out is None:
        out = Packed(rows, C, x.device)
    _call("mvs_layernorm_x3p", "layernorm_x3p", _ptr(x), _ptr(gamma), _ptr(beta), out.ptr(), rows, C, Np, N, float(eps), _stream())
    return out


def gemm_x3p(A: Packed, B: Packed, N: int, C: Optional[torch.Tensor] = None, scale=None, shift=None, act: int = 0, res=None,
             out: Optional[Packed] = None) -> None:
    """``epi(A . B^T)`` on packed operands -> fp32 ``C [M, N]`` (contiguous rows) and / or packed ``out [M][N]``."""
    if A.K != B.K or B.rows < N:
        raise _lib.MvsHipError("gemm_x3p: A [%d][%d] x B [%d][%d], N = %d" % (A.rows, A.K, B.rows, B.K, N))
    if C is not None:
        _chk(C, "C")
    _opt(scale, "scale"), _opt(shift, "shift"), _opt(res, "res")
    if out is not None and (out.K != N or out.rows_alloc < A.rows):
        raise _lib.MvsHipError("gemm_x3p: packed output [%d][%d] for M = %d, N = %d" % (out.rows_alloc, out.K, A.rows, N))
    _call("mvs_gemm_x3p", ("x3p_gemm", "flops", 2.0 * A.rows * N * A.K), A.ptr(), B.ptr(), A.rows, N, A.K, A.rows_alloc, B.rows_alloc, _ptr(C),
          C.shape[-1] if C is not None else 0, _ptr(scale), _ptr(shift), int(act), _ptr(res), out.ptr() if out is not None else None, _stream())


class QkvPacked:
    """Q (pre-scaled), K packed ``[image][head][Np][64]`` and V^T packed ``[image][head][64][Np]`` (keys permuted per 32-step)."""
    __slots__ = ("q", "k", "vt", "images", "heads", "Np")

    def __init__(self, images: int, heads: int, Np: int, device):
        self.images, self.heads, self.Np = images, heads, Np
        n = images * heads * (Np // 16) * 2 * 3072
        self.q = torch.empty(n, device=device, dtype=torch.uint8)
        self.k = torch.empty(n, device=device, dtype=torch.uint8)
        self.vt = torch.empty(n, device=device, dtype=torch.uint8)


def gemm_x3p_qkv(A: Packed, W: Packed, bias: Optional[torch.Tensor], images: int, Np: int, heads: int, scale: float, out: Optional[QkvPacked] = None) -> QkvPacked:
    """``attn.qkv`` -> packed Q, K, V^T.  ``scale`` = the softmax scale (``head_dim ** -0.5``); Q is stored times ``scale * log2(e)`` because
    :func:`attention_x3p` / :func:`cls_attention_x3p` exponentiate in base 2 (one v_exp_f32 per score, no multiply)."""
    qscale = scale * 1.4426950408889634
    C = heads * 64
    if A.K != C or W.K != C or A.rows != images * Np:
        raise _lib.MvsHipError("gemm_x3p_qkv: A [%d][%d], W [%d][%d] for %d images x %d rows, %d heads" % (A.rows, A.K, W.rows, W.K, images, Np, heads))
    _opt(bias, "bias")
    if out is None:
        out = QkvPacked(images, heads, Np, A.buf.device)
    _call("mvs_gemm_x3p_qkv", ("x3p_gemm", "flops", 2.0 * A.rows * 3 * C * C), A.ptr(), W.ptr(), images, Np, C, heads, A.rows_alloc, W.rows_alloc, _ptr(bias),
          float(qscale), out.q.data_ptr(), out.k.data_ptr(), out.vt.data_ptr(), _stream())
    return out


def attention_x3p(qkv: QkvPacked, N: int, out: Optional[Packed] = None) -> Packed:
    """Flash attention on the packed operands -> packed ``[images*Np][heads*64]``."""
    if out is None:
        out = Packed(qkv.images * qkv.Np, qkv.heads * 64, qkv.q.device)
    _call("mvs_attention_x3p", ("x3p_attention", "flops", 4.0 * qkv.images * qkv.heads * N * N * 64), qkv.q.data_ptr(), qkv.k.data_ptr(), qkv.vt.data_ptr(),
          out.ptr(), qkv.images, N, qkv.Np, qkv.heads, _stream())
    return out


def cls_attention_x3p(qkv: QkvPacked, N: int) -> torch.Tensor:
    """``[images, heads, N]``: the CLS query's attention row per head."""
    att = torch.empty(qkv.images, qkv.heads, N, device=qkv.q.device, dtype=torch.float32)
    _call("mvs_cls_attention_x3p", "x3p_cls_attention", qkv.q.data_ptr(), qkv.k.data_ptr(), _ptr(att), qkv.images, N, qkv.Np, qkv.heads, _stream())
    return att


def _conv2d_out(H, W, KS, S, P):
    return (H + 2 * P - KS) // S + 1, (W + 2 * P - KS) // S + 1


def conv2d_fwd_x3(x: torch.Tensor, w: torch.Tensor, stride: int, pad: int) -> torch.Tensor:
    """Raw ``F.conv2d(x, w, stride=stride, padding=pad)`` (fp32 NCHW, no bias) as a split-form GEMM with an implicit patch matrix."""
    _chk(x, "x"), _chk(w, "weight")
    N, Cin, H, W = x.shape
    Cout, _, KS, _ = w.shape
    Ho, Wo = _conv2d_out(H, W, KS, stride, pad)
    y = torch.empty(N, Cout, Ho, Wo, device=x.device, dtype=torch.float32)
    _call("mvs_conv2d_gemm_x3", ("x3_conv2d_fwd", "flops", 2.0 * N * Cout * Cin * KS * KS * Ho * Wo), 1, _ptr(w), _ptr(x), _ptr(y), N, Cin, Cout, H, W,
          Ho, Wo, KS, stride, pad, 0, _stream())
    return y


def conv2d_dgrad_x3(dy: torch.Tensor, w: torch.Tensor, stride: int, pad: int, H: int, W: int) -> torch.Tensor:
    """Gradient of :func:`conv2d_fwd_x3` with respect to its input ``[N,Cin,H,W]``."""
    _chk(dy, "dy"), _chk(w, "weight")
    N, Cout, Ho, Wo = dy.shape
    Cin, KS = w.shape[1], w.shape[2]
    wt = w.permute(1, 0, 2, 3).contiguous()                 # [Cin][Cout*KS*KS]
    dx = torch.empty(N, Cin, H, W, device=dy.device, dtype=torch.float32)
    _call("mvs_conv2d_gemm_x3", ("x3_conv2d_dgrad", "flops", 2.0 * N * Cout * Cin * KS * KS * Ho * Wo), 2, _ptr(wt), _ptr(dy), _ptr(dx), N, Cin, Cout,
          H, W, Ho, Wo, KS, stride, pad, 0, _stream())
    return dx


def conv2d_wgrad_x3(dy: torch.Tensor, x: torch.Tensor, KS: int, stride: int, pad: int) -> torch.Tensor:
    """Gradient of :func:`conv2d_fwd_x3` with respect to its weight: split-K partial matrices (one per image and pixel range) added in a
    fixed order."""
    _chk(dy, "dy"), _chk(x, "x")
    N, Cout, Ho, Wo = dy.shape
    _, Cin, H, W = x.shape
    K = Ho * Wo
    # enough splits to fill the chip, at most 65535 / N of them, each a multiple of the 32-pixel K step
    want = max(1, min(65535 // N, (2048 + N - 1) // N))
    ksplit = max(32, ((K + want - 1) // want + 31) // 32 * 32)
    nsplit = (K + ksplit - 1) // ksplit
    n = Cout * Cin * KS * KS
    part = torch.empty(N * nsplit, n, device=x.device, dtype=torch.float32)
    _call("mvs_conv2d_gemm_x3", ("x3_conv2d_wgrad", "flops", 2.0 * N * Cout * Cin * KS * KS * Ho * Wo), 3, _ptr(dy), _ptr(x), _ptr(part), N, Cin, Cout,
          H, W, Ho, Wo, KS, stride, pad, ksplit, _stream())
    dw = torch.empty(Cout, Cin, KS, KS, device=x.device, dtype=torch.float32)
    _call("mvs_partials_reduce", "partials_reduce", _ptr(part), N * nsplit, n, _ptr(dw), _stream())
    return dw


def ewise_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``a * b`` elementwise (same shape, fp32)."""
    _chk(a, "a"), _chk(b, "b")
    if a.shape != b.shape:
        raise _lib.MvsHipError("ewise_mul: %s * %s" % (tuple(a.shape), tuple(b.shape)))
    out = torch.empty_like(a)
    _call("mvs_ewise_mul", "ewise_mul", _ptr(a), _ptr(b), a.numel(), _ptr(out), _stream())
    return out


def upsample2x_add(x: torch.Tensor, lateral: Optional[torch.Tensor]) -> torch.Tensor:
    """``F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True) (+ lateral)``, fp32 NCHW."""
    _chk(x, "x"), _opt(lateral, "lateral")
    N, C, h, w = x.shape
    y = torch.empty(N, C, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
    _call("mvs_upsample2x_add", "upsample2x_add", _ptr(x), _ptr(lateral), _ptr(y), N * C, h, w, _stream())
    return y


def upsample2x_bwd(dy: torch.Tensor) -> torch.Tensor:
    _chk(dy, "dy")
    N, C, H, W = dy.shape
    dx = torch.empty(N, C, H // 2, W // 2, device=dy.device, dtype=torch.float32)
    _call("mvs_upsample2x_bwd", "upsample2x_bwd", _ptr(dy), _ptr(dx), N * C, H // 2, W // 2, _stream())
    return dx
