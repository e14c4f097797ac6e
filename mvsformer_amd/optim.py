"""AdamW on the HIP path: ``torch.optim.AdamW``'s update (the reference's optimizer, train.py:98) for all parameter tensors of a group in
a few launches of ``mvs_adamw_step`` (csrc/optim.hip) instead of ATen's multi-tensor kernel.

Same constructor arguments as ``torch.optim.AdamW`` where they apply.  State: ``exp_avg``, ``exp_avg_sq`` per parameter; the step count is
ONE device scalar per group (``param_groups[i]['step']``, fp32), advanced by the kernel, so ``step()`` is capturable in a hipGraph as it is.
``state_dict()`` ALSO carries torch's per-parameter ``'step'`` (a copy of the group's count), so the checkpoint loads into
``torch.optim.AdamW`` and back; ``load_state_dict()`` accepts both layouts (the reference resumes with ``torch.load(map_location='cpu')`` +
``optimizer.load_state_dict``, train.py:110-117): the group's count is moved to the parameters' device, or seeded from the per-parameter
counts of a ``torch.optim.AdamW`` / reference checkpoint (which must agree within a group).
fp32 contiguous GPU parameters only; ``amsgrad`` and sparse gradients are not built; there is no CPU path.

Under hipGraph capture the scalar hyper-parameters (``lr``, betas, ``eps``, ``weight_decay``) travel as kernel arguments and are captured
BY VALUE: a learning-rate schedule needs a re-capture when the rate changes (or the eager ``step()``); the step count, which the bias
corrections depend on, is read from the device and does advance in a replay.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib, ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False):
        if amsgrad:
            raise _lib.MvsHipError("FusedAdamW: amsgrad is not built")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdamW: lr %r, betas %r, eps %r, weight_decay %r" % (lr, betas, eps, weight_decay))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize))

    def state_dict(self):
        sd = super().state_dict()
        packed = {id(p): i for i, p in enumerate(q for g in self.param_groups for q in g["params"])}
        for g_live, g_out in zip(self.param_groups, sd["param_groups"]):
            if "step" not in g_live:
                continue
            count = g_live["step"].detach().to("cpu", torch.float32).clone()
            g_out["step"] = count
            for p in g_live["params"]:
                st = sd["state"].get(packed[id(p)])
                if st is not None:
                    st["step"] = count.clone()               # torch.optim.AdamW's layout: one count per parameter
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            params = [p for p in group["params"]]
            if not params:
                continue
            dev = params[0].device
            counts = {float(self.state[p]["step"]) for p in params if p in self.state and "step" in self.state[p]}
            if "step" in group and group["step"] is not None:
                group["step"] = torch.as_tensor(group["step"], dtype=torch.float32).detach().to(dev).reshape(())
            elif counts:
                if len(counts) != 1:
                    raise _lib.MvsHipError("FusedAdamW.load_state_dict: the parameters of one group carry different step counts %s (one count per group here)"
                                           % sorted(counts))
                group["step"] = torch.tensor(counts.pop(), device=dev, dtype=torch.float32)
            elif any(p in self.state and self.state[p] for p in params):
                raise _lib.MvsHipError("FusedAdamW.load_state_dict: moments without a step count: the bias corrections would restart at 1")
            for p in params:                                 # the per-parameter copies are not state here
                if p in self.state:
                    self.state[p].pop("step", None)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            todo = [p for p in group["params"] if p.grad is not None]
            if not todo:
                continue
            dev = todo[0].device
            if "step" not in group:
                group["step"] = torch.zeros((), device=dev, dtype=torch.float32)
            cnt = group["step"]
            if not (isinstance(cnt, torch.Tensor) and cnt.is_cuda and cnt.device == dev and cnt.dtype == torch.float32 and cnt.numel() == 1):
                raise _lib.MvsHipError("FusedAdamW: param_groups[..]['step'] must be one fp32 scalar on %s (got %r): load checkpoints through load_state_dict" % (dev, cnt))
            arr = (_lib.AdamTensor * len(todo))()
            for k, p in enumerate(todo):
                g = p.grad
                if p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_cuda or p.device != dev or g.is_sparse:
                    raise _lib.MvsHipError("FusedAdamW: fp32 dense GPU parameters of one device (got %s / %s on %s)" % (p.dtype, g.dtype, p.device))
                if not p.is_contiguous():
                    raise _lib.MvsHipError("FusedAdamW: a parameter is not contiguous")
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                t = arr[k]
                t.p, t.g, t.m, t.v, t.n = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            b1, b2 = group["betas"]
            ops._call("mvs_adamw_step", "adamw", ctypes.cast(arr, ctypes.c_void_p), len(todo), float(group["lr"]), float(b1), float(b2),
                      float(group["eps"]), float(group["weight_decay"]), int(bool(group["maximize"])), group["step"].data_ptr(), ops._stream())
            ops.bump_weights_epoch()                         # parameters written through raw pointers: torch's _version does not move
        return loss
