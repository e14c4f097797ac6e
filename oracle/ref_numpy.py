"""ORACLE — test infrastructure, not product code.

Second, ATen-free restatement of the warp + group correlation in plain numpy
(explicit bilinear taps instead of ``F.grid_sample``), for small cases.  It
pins the exact sampling semantics the HIP kernels implement (reference
models/warping.py:84-106 + ``grid_sample(mode='bilinear', padding_mode='zeros',
align_corners=True)`` as the reference calls it at warping.py:105):

* integer pixel lattice, no half-pixel offset;
* ``u = X0/(X2+1e-6)``; normalize ``u/((W-1)/2)-1``; un-normalize ``((u_n+1)/2)*(W-1)``;
* 4 taps at ``floor`` / ``floor+1``; a tap outside ``[0,W-1]x[0,H-1]`` contributes 0 on its own;
* ``X2 <= 0`` is masked but still sampled.

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import numpy as np


def bilinear_zeros(src: np.ndarray, ix: np.ndarray, iy: np.ndarray) -> np.ndarray:
    """``src [C,H,W]``, un-normalized float32 coordinates ``ix, iy [...]`` -> ``[C,...]``."""
    C, H, W = src.shape
    f32 = np.float32
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1 = x0 + f32(1)
    y1 = y0 + f32(1)
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    out = np.zeros((C,) + ix.shape, dtype=f32)

    def tap(xx, yy, ww):
        with np.errstate(invalid="ignore"):
            ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        xi = np.where(ok, xx, 0).astype(np.int64)
        yi = np.where(ok, yy, 0).astype(np.int64)
        vals = src[:, yi, xi]
        return np.where(ok[None], vals * ww[None].astype(f32), f32(0))

    out += tap(x0, y0, w_nw)
    out += tap(x1, y0, w_ne)
    out += tap(x0, y1, w_sw)
    out += tap(x1, y1, w_se)
    return out


def plane_sweep_warp(src_fea: np.ndarray, src_proj: np.ndarray, ref_proj: np.ndarray, depth: np.ndarray):
    """``src_fea [B,C,H,W]``, ``*_proj [B,4,4]``, ``depth [B,D]`` or ``[B,D,H,W]`` (all float32)
    -> ``warped [B,C,D,H,W]``, ``mask [B,D,H,W]`` (True = outside the source frustum)."""
    f32 = np.float32
    B, C, H, W = src_fea.shape
    D = depth.shape[1]
    warped = np.zeros((B, C, D, H, W), dtype=f32)
    mask = np.zeros((B, D, H, W), dtype=bool)
    ys, xs = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    for b in range(B):
        M = (src_proj[b].astype(f32) @ np.linalg.inv(ref_proj[b].astype(f32)).astype(f32)).astype(f32)
        R, t = M[:3, :3], M[:3, 3]
        ray = [(R[i, 0] * xs + R[i, 1] * ys + R[i, 2]).astype(f32) for i in range(3)]
        for d in range(D):
            dep = depth[b, d] if depth.ndim == 4 else f32(depth[b, d])
            X = [(ray[i] * dep + t[i]).astype(f32) for i in range(3)]
            with np.errstate(divide="ignore", invalid="ignore"):
                u = X[0] / (X[2] + f32(1e-6))
                v = X[1] / (X[2] + f32(1e-6))
                un = (u / f32((W - 1) / 2) - f32(1)).astype(f32)
                vn = (v / f32((H - 1) / 2) - f32(1)).astype(f32)
                mask[b, d] = (un > 1) | (un < -1) | (vn > 1) | (vn < -1) | (X[2] <= 0)
                ix = ((un + f32(1)) / f32(2) * f32(W - 1)).astype(f32)
                iy = ((vn + f32(1)) / f32(2) * f32(H - 1)).astype(f32)
            warped[b, :, d] = bilinear_zeros(src_fea[b].astype(f32), ix, iy)
    return warped, mask


def group_correlation(ref_feat: np.ndarray, warped: np.ndarray, G: int) -> np.ndarray:
    """reference mvsformer_model.py:75-79 -> ``[B,G,D,H,W]``."""
    B, C, D, H, W = warped.shape
    cpg = C // G
    r = ref_feat.reshape(B, G, cpg, 1, H, W).astype(np.float32)
    w = warped.reshape(B, G, cpg, D, H, W)
    return (r * w).mean(axis=2, dtype=np.float32)
