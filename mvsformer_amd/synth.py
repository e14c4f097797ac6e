"""Synthetic, photo-consistent DTU-shaped inputs for the plane-sweep path.

No datasets or checkpoints exist on the build/GPU boxes, so every test and the
bench feed the path with a seeded analytic scene instead of DTU images:

* cameras follow the DTU intrinsics scaled to the requested resolution and the
  per-stage scaling the reference loader applies
  (reference datasets/general_eval.py:90,210-246): ``proj[:, v, 0]`` is the 4x4
  world->camera extrinsic, ``proj[:, v, 1, :3, :3]`` the stage intrinsic;
* the scene is one slanted plane ``n . X = d0`` in the reference camera frame,
  so the reference->source pixel map is an exact homography and per-view
  feature maps can be rendered in closed form: every view samples the same
  band-limited texture (a sum of seeded sinusoids per channel) at the
  reference-image point its pixel sees.  A correct plane sweep therefore gets a
  correlation peak at the true plane depth (pure-noise features would give a
  flat softmax and a weak parity test);
* ``depth_values = 425 + 2.65 * arange(192)`` is the DTU depth range the
  reference loader emits (datasets/general_eval.py:94-104,220).

Everything is torch, device-agnostic and deterministic for a given seed.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch

STAGE_SCALES = (8, 4, 2, 1)          # stage1..4 are 1/8, 1/4, 1/2, 1/1 resolution
STAGE_CHANNELS = (64, 32, 16, 8)     # FPN decoder widths, reference configs feat_chs reversed
DTU_DEPTH_MIN = 425.0
DTU_DEPTH_INTERVAL = 2.5 * 1.06
DTU_NUM_DEPTH = 192


@dataclass
class Scene:
    K: torch.Tensor           # [3,3] full-resolution intrinsics (float64, cpu)
    E: torch.Tensor           # [V,4,4] world->camera extrinsics (float64, cpu)
    plane_n: torch.Tensor     # [3] unit normal in the reference camera frame
    plane_d: float            # n . X = d
    height: int
    width: int
    seed: int


def _rot_xyz(rx: float, ry: float, rz: float) -> torch.Tensor:
    cx, sx, cy, sy, cz, sz = (math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry),
                              math.cos(rz), math.sin(rz))
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
    return Rz @ Ry @ Rx


def make_scene(num_views: int, height: int, width: int, seed: int = 0) -> Scene:
    """DTU-like rig: reference camera at the origin, sources on an arc around the
    scene centre (baselines 60-250 mm, rotations of a few degrees)."""
    g = torch.Generator().manual_seed(1000 + seed)
    K = torch.tensor([[2892.33 * width / 1600.0, 0.0, 823.205 * width / 1600.0],
                      [0.0, 2883.18 * height / 1200.0, 619.071 * height / 1200.0],
                      [0.0, 0.0, 1.0]], dtype=torch.float64)
    E = torch.zeros(num_views, 4, 4, dtype=torch.float64)
    E[0] = torch.eye(4, dtype=torch.float64)
    target = torch.tensor([0.0, 0.0, 680.0], dtype=torch.float64)
    for v in range(1, num_views):
        u = torch.rand(3, generator=g, dtype=torch.float64)
        base = 60.0 + 190.0 * float(u[0])
        phi = 2.0 * math.pi * (float(u[1]) + v / max(1, num_views - 1))
        centre = torch.tensor([base * math.cos(phi), 0.6 * base * math.sin(phi),
                               20.0 * (float(u[2]) - 0.5)], dtype=torch.float64)
        # look at the scene centre, then add a small roll
        z = target - centre
        z = z / z.norm()
        x = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64), z)
        x = x / x.norm()
        y = torch.linalg.cross(z, x)
        R = torch.stack([x, y, z], dim=0)                     # world -> camera
        R = _rot_xyz(0.0, 0.0, 0.03 * (float(u[2]) - 0.5)) @ R
        E[v, :3, :3] = R
        E[v, :3, 3] = -R @ centre
        E[v, 3, 3] = 1.0
    n = torch.tensor([0.15, -0.10, 1.0], dtype=torch.float64)
    n = n / n.norm()
    return Scene(K=K, E=E, plane_n=n, plane_d=650.0 * float(n[2]), height=height, width=width, seed=seed)


def stage_intrinsics(K: torch.Tensor, scale: int) -> torch.Tensor:
    Ks = K.clone()
    Ks[:2, :] = Ks[:2, :] / float(scale)
    return Ks


def proj_matrices(scene: Scene, scales: Sequence[int] = STAGE_SCALES, batch: int = 1,
                  device="cpu", dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """``{'stageN': [B,V,2,4,4]}`` exactly as the reference loader lays them out."""
    out = {}
    V = scene.E.shape[0]
    for i, s in enumerate(scales):
        pm = torch.zeros(V, 2, 4, 4, dtype=torch.float64)
        pm[:, 0] = scene.E
        pm[:, 1, :3, :3] = stage_intrinsics(scene.K, s)
        out["stage%d" % (i + 1)] = pm.unsqueeze(0).repeat(batch, 1, 1, 1, 1).to(device=device, dtype=dtype)
    return out


def depth_range(batch: int = 1, num_depth: int = DTU_NUM_DEPTH, device="cpu", dtype=torch.float32) -> torch.Tensor:
    d = DTU_DEPTH_MIN + DTU_DEPTH_INTERVAL * torch.arange(num_depth, dtype=torch.float64)
    return d.unsqueeze(0).repeat(batch, 1).to(device=device, dtype=dtype)


def plane_depth(scene: Scene, scale: int, device="cpu", dtype=torch.float32) -> torch.Tensor:
    """True depth of the plane at every pixel of the reference view, stage resolution ``[H,W]``."""
    Hs, Ws = scene.height // scale, scene.width // scale
    Kinv = torch.linalg.inv(stage_intrinsics(scene.K, scale))
    ys, xs = torch.meshgrid(torch.arange(Hs, dtype=torch.float64), torch.arange(Ws, dtype=torch.float64), indexing="ij")
    rays = torch.stack([xs, ys, torch.ones_like(xs)], dim=0).reshape(3, -1)
    rays = Kinv @ rays
    z = scene.plane_d / (scene.plane_n @ rays)
    return z.reshape(Hs, Ws).to(device=device, dtype=dtype)


def _texture_params(channels: int, scale: int, seed: int, n_waves: int = 6):
    g = torch.Generator().manual_seed(7919 * seed + 31 * channels + scale)
    # wavelengths between 5 and 24 stage pixels => no aliasing at this stage's sampling
    wl = (5.0 + 19.0 * torch.rand(n_waves, generator=g, dtype=torch.float64)) * scale
    ang = 2.0 * math.pi * torch.rand(n_waves, generator=g, dtype=torch.float64)
    fx = torch.cos(ang) / wl
    fy = torch.sin(ang) / wl
    amp = torch.randn(channels, n_waves, generator=g, dtype=torch.float64) * math.sqrt(2.0 / n_waves)
    phase = 2.0 * math.pi * torch.rand(channels, n_waves, generator=g, dtype=torch.float64)
    return fx, fy, amp, phase


def _texture(xy_full: torch.Tensor, params, dtype) -> torch.Tensor:
    """Evaluate the C-channel texture at full-resolution reference coordinates ``xy_full [2,N]`` -> ``[C,N]``."""
    fx, fy, amp, phase = [p.to(device=xy_full.device, dtype=dtype) for p in params]
    arg = 2.0 * math.pi * (fx[:, None] * xy_full[0][None, :] + fy[:, None] * xy_full[1][None, :])   # [K,N]
    out = torch.zeros(amp.shape[0], xy_full.shape[1], device=xy_full.device, dtype=dtype)
    for k in range(arg.shape[0]):                        # K is tiny; keeps the [C,K,N] temporary away
        out += amp[:, k:k + 1] * torch.sin(arg[k][None, :] + phase[:, k:k + 1])
    return out


def render_features(scene: Scene, scale: int, channels: int, noise: float = 0.05, batch: int = 1,
                    device="cpu", dtype=torch.float32, compute_dtype=None) -> torch.Tensor:
    """Per-view feature maps ``[B,V,C,H/scale,W/scale]`` of the plane scene (NCHW, like the FPN decoder)."""
    cd = compute_dtype or (torch.float64 if str(device) == "cpu" else torch.float32)
    Hs, Ws = scene.height // scale, scene.width // scale
    V = scene.E.shape[0]
    Ks = stage_intrinsics(scene.K, scale)
    Kinv = torch.linalg.inv(Ks)
    params = _texture_params(channels, scale, scene.seed)
    ys, xs = torch.meshgrid(torch.arange(Hs, dtype=cd, device=device), torch.arange(Ws, dtype=cd, device=device), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(Hs * Ws, dtype=cd, device=device)], dim=0)
    feats = []
    Er = scene.E[0]
    for v in range(V):
        Ev = scene.E[v]
        R_rel = Ev[:3, :3] @ Er[:3, :3].T
        t_rel = Ev[:3, 3] - R_rel @ Er[:3, 3]
        Hm = Ks @ (R_rel + torch.outer(t_rel, scene.plane_n) / scene.plane_d) @ Kinv    # ref px -> src px
        Hinv = torch.linalg.inv(Hm).to(device=device, dtype=cd)
        q = Hinv @ pix
        xy = q[:2] / q[2:3]                       # stage-resolution reference coordinates seen by this pixel
        tex = _texture(xy * float(scale), params, cd)
        feats.append(tex.reshape(channels, Hs, Ws))
    f = torch.stack(feats, dim=0)                # [V,C,H,W]
    if noise > 0:
        g = torch.Generator(device="cpu").manual_seed(424243 + scene.seed * 17 + scale)
        nz = torch.randn(f.shape, generator=g, dtype=torch.float32).to(device=device, dtype=cd)
        f = f + noise * nz
    return f.unsqueeze(0).repeat(batch, 1, 1, 1, 1).to(dtype)


def make_inputs(num_views: int, height: int, width: int, seed: int = 0, batch: int = 1, device="cpu",
                dtype=torch.float32, scales: Sequence[int] = STAGE_SCALES,
                channels: Sequence[int] = STAGE_CHANNELS, num_depth: int = DTU_NUM_DEPTH,
                noise: float = 0.05) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], torch.Tensor, Scene]:
    """features{stageN}, proj_matrices{stageN}, depth_values[B,num_depth], scene."""
    scene = make_scene(num_views, height, width, seed)
    feats = {"stage%d" % (i + 1): render_features(scene, s, c, noise=noise, batch=batch, device=device, dtype=dtype)
             for i, (s, c) in enumerate(zip(scales, channels))}
    return feats, proj_matrices(scene, scales, batch, device, dtype), depth_range(batch, num_depth, device, dtype), scene
