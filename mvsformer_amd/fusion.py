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


def get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """fusion.py:116-150 -> ``reproj_xyd [n,v,3,h,w]``."""
    return ops.geo_filter_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam, want=("reproj_xyd",))["reproj_xyd"]


def vis_filter_dynamic(ref_depth, reproj_xyd, dist_base=4, rel_diff_base=1300):
    """fusion.py:153-165 -> ``(masks [n,v,v-1,h,w] bool, mask [n,v,1,h,w] bool)``."""
    out = ops.vis_filter_dynamic(ref_depth, reproj_xyd, dist_base, rel_diff_base)
    return out["masks"], out["vis_mask"]


def dynamic_filter_depth_maps(ref_depth, srcs_depth, ref_cam, srcs_cam, dist_base=4, rel_diff_base=1300,
                              with_intermediates: bool = False) -> Dict[str, torch.Tensor]:
    """test.py:494-514 fused: ``geo_mask`` (bool ``[n,1,h,w]``: some level k in 2..v is passed by at least k source views),
    ``ref_depth_ave`` (mean of the reference depth and the level-v consistent reprojections), ``points``; with
    ``with_intermediates`` also ``reproj_xyd``, ``masks``, ``vis_mask``.  (For n > 1 the reference's ``geo_mask`` broadcasts
    ``[n,1,h,w] | [n,h,w]`` to ``[n,n,h,w]``; this returns its per-sample diagonal meaning.)"""
    want = ("geo_mask", "ref_depth_ave", "points") + (("reproj_xyd", "masks", "vis_mask") if with_intermediates else ())
    return ops.geo_filter_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam, dist_base, rel_diff_base, want=want)


def filter_scan(pair_folder: str, scan_folder: str, prob_threshold: Sequence[float], method: str = "pcd", thres_disp: float = 1.0,
                thres_view: float = 2, dist_base: float = 4, rel_diff_base: float = 1300, n_src_views: int = 10,
                device: str = "cuda:0"):
    """The per-scan loop of test.py:404-438 (``method='pcd'``, ``filter_depth``) / test.py:475-514 (``'dypcd'``,
    ``dynamic_filter_depth``) over the folder layout ``data_io.save_depth_outputs`` writes: for every reference view of
    ``pair.txt`` load the depth/confidence/camera files, zero the prob-filtered source depths (pcd only, as in the
    reference), run the fused consistency pass, AND with the reference view's own prob mask, and gather the surviving world
    points.  Returns ``{ref_id: (points [M,3] float32 numpy, stats dict)}``; colours and the PLY are the caller's business.
    """
    import torch
    from . import data_io
    if method not in ("pcd", "dypcd"):
        raise ValueError("method must be 'pcd' or 'dypcd'")
    views = {}
    for id_ref, id_srcs in data_io.read_pair_file(pair_folder + "/pair.txt"):
        s = {k: (torch.from_numpy(v).unsqueeze(0).to(device) if hasattr(v, "shape") else v)
             for k, v in data_io.load_filter_sample(scan_folder, id_ref, id_srcs, n_src_views).items()}
        src_depths = s["src_depths"].contiguous()
        if method == "pcd":
            for i in range(src_depths.shape[1]):
                ops.prob_filter(s["src_confs"][:, i].contiguous(), prob_threshold, depth_inplace=src_depths[:, i])
            out = filter_depth_maps(s["ref_depth"], src_depths, s["ref_cam"], s["src_cams"], thres_disp, 0.01, thres_view)
            geo = out["mask"]
        else:
            out = dynamic_filter_depth_maps(s["ref_depth"], src_depths, s["ref_cam"], s["src_cams"], dist_base, rel_diff_base)
            geo = out["geo_mask"]
        prob = prob_filter(s["ref_conf"].contiguous(), prob_threshold)
        keep = prob & geo
        pts = out["points"][0].permute(1, 2, 0)[keep[0, 0]]
        views[id_ref] = (pts.cpu().numpy(), dict(photo=prob.float().mean().item(), geo=geo.float().mean().item(),
                                                 final=keep.float().mean().item()))
    return views
