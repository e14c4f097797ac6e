"""Host-side mirror of the reference's ``models/warping.py`` plane-sweep warps (op-level, materializing).

``homo_warping_3D_with_mask`` (reference models/warping.py:69-109), ``homo_warping_3D`` (:155-189) and
``diff_homo_warping_3D_with_mask`` (:112-152) share one HIP kernel (``mvs_warp_fwd``); ``homo_warping`` is the
spelling BASELINE.json uses.  The fused cost-volume build inside :class:`mvsformer_amd.stagenet.StageNet` never
calls these (it does not materialize the warped volume); they exist for callers of the op itself.
"""
from __future__ import annotations

import torch

from . import ops


def _prep(src_fea, src_proj, ref_proj, depth_values):
    src = src_fea.detach().to(torch.float32).contiguous()
    rt = ops.proj_relative(src_proj.detach().to(torch.float32).contiguous(), ref_proj.detach().to(torch.float32).contiguous())
    return src, rt, depth_values.detach().to(torch.float32).contiguous()


def homo_warping_3D_with_mask(src_fea, src_proj, ref_proj, depth_values):
    """``src_fea [B,C,H,W]``, ``src_proj/ref_proj [B,4,4]``, ``depth_values [B,D]`` or ``[B,D,H,W]`` ->
    ``(warped [B,C,D,H,W], proj_mask [B,D,H,W] bool)``; mask True = sample outside the source frustum."""
    src, rt, depth = _prep(src_fea, src_proj, ref_proj, depth_values)
    return ops.warp(src, rt, depth, with_mask=True)


def homo_warping_3D(src_fea, src_proj, ref_proj, depth_values):
    src, rt, depth = _prep(src_fea, src_proj, ref_proj, depth_values)
    return ops.warp(src, rt, depth, with_mask=False)[0]


# the reference's "diff_" variant differs only in autograd scope (grid not under no_grad); forward is identical
diff_homo_warping_3D_with_mask = homo_warping_3D_with_mask
homo_warping = homo_warping_3D
