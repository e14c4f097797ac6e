"""Depth-map geometric consistency filtering on the HIP path — the reference's ``misc/fusion.py`` interface.

Same function names and argument meaning as reference misc/fusion.py:69-114 (``prob_filter``, ``get_reproj``,
``vis_filter``, ``ave_fusion``) plus ``filter_depth_maps``: the whole block of test.py:425-434 (reprojection, masks,
averaged depth, fused world points) in ONE pass over the reference pixels with nothing materialized in between.

    ref_depth [n,1,h,w]   srcs_depth [n,v,1,h,w]   ref_cam [n,2,4,4]   srcs_cam [n,v,2,4,4]   (cam[:,0]=E, cam[:,1,:3,:3]=K)
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from . import ops


def prob_filter(ref_prob: torch.Tensor, prob_thresh: Sequence[float], greater: bool = True) -> torch.Tensor:
    """fusion.py:69-77 — ``ref_prob [n,C,...]`` -> bool ``[n,1,...]``."""
    return ops.prob_filter(ref_prob, prob_thresh)


def get_reproj(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """fusion.py:80-98 -> ``(reproj_xyd [n,v,3,h,w], in_range [n,v,1,h,w])``."""
    out = ops.geo_filter(ref_depth, srcs_depth, ref_cam, srcs_cam, want=("reproj_xyd", "in_range"))
    return out["reproj_xyd"], out["in_range"]


def vis_filter(ref_depth, reproj_xyd, in_range, img_dist_thresh, depth_thresh, vthresh):
    """fusion.py:101-109 -> ``(masks [n,v,1,h,w] float, mask [n,1,h,w] bool)``."""
    masks, mask, _ = ops.vis_filter(ref_depth, reproj_xyd, in_range, None, img_dist_thresh, depth_thresh, vthresh, want_ave=False)
    return masks, mask


def ave_fusion(ref_depth, reproj_xyd, masks):
    """fusion.py:112-114 -> ``[n,1,h,w]``."""
    return ops.vis_filter(ref_depth, reproj_xyd, None, masks, 0.0, 0.0, 0.0, want_masks=False)[2]


def filter_depth_maps(ref_depth, srcs_depth, ref_cam, srcs_cam, thres_disp: float, depth_thresh: float = 0.01,
                      thres_view: float = 2, with_intermediates: bool = False) -> Dict[str, torch.Tensor]:
    """test.py:425-434 fused: ``mask`` (bool), ``ref_depth_ave``, ``points`` [n,3,h,w]; with ``with_intermediates`` also
    ``reproj_xyd``, ``in_range``, ``masks``."""
    want = ("mask", "ref_depth_ave", "points") + (("reproj_xyd", "in_range", "masks") if with_intermediates else ())
    return ops.geo_filter(ref_depth, srcs_depth, ref_cam, srcs_cam, thres_disp, depth_thresh, thres_view, want=want)
