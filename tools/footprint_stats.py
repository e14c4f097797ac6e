#!/usr/bin/env python
"""Source-image footprint of reference-pixel tiles in the plane sweep: for every cascade stage and source view, the bounding
box (in source texels) of the bilinear taps of ALL pixels of a TH x TW reference tile over ALL depth hypotheses.  This is
the quantity that decides whether a sweep can stage its source texels through LDS once per tile (design input for
csrc/cost_volume_tiled.hip).

    python tools/footprint_stats.py [--height 1152 --width 1536 --views 5] -> gpurun_out/footprint_stats.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sweep_uv(proj, hyp):
    """proj [V,2,4,4] (one batch entry), hyp [D,H,W] -> u, v [V-1,D,H,W] source pixel coordinates (float64 math is not needed
    for statistics)."""
    V = proj.shape[0]
    D, H, W = hyp.shape
    dev = hyp.device

    def compose(p):
        P = p[0].clone()
        P[:3, :4] = p[1, :3, :3] @ p[0, :3, :4]
        return P
    Pr = compose(proj[0]).double()
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64), torch.arange(W, device=dev, dtype=torch.float64), indexing="ij")
    us, vs = [], []
    for v in range(1, V):
        M = compose(proj[v]).double() @ torch.linalg.inv(Pr)
        R, t = M[:3, :3], M[:3, 3]
        rx = R[0, 0] * xs + R[0, 1] * ys + R[0, 2]
        ry = R[1, 0] * xs + R[1, 1] * ys + R[1, 2]
        rz = R[2, 0] * xs + R[2, 1] * ys + R[2, 2]
        d = hyp.double()
        z = rz[None] * d + t[2]
        us.append(((rx[None] * d + t[0]) / z).float())
        vs.append(((ry[None] * d + t[1]) / z).float())
    return torch.stack(us), torch.stack(vs)


def tile_boxes(u, v, TH, TW, H, W):
    """u, v [Vs,D,H,W] -> box width / height [Vs, H/TH, W/TW] in texels (taps floor(u) .. floor(u)+1, clamped to the image
    with one texel of zero border)."""
    Vs, D = u.shape[:2]
    Hc, Wc = (H // TH) * TH, (W // TW) * TW
    uu = u[:, :, :Hc, :Wc].clamp(-1, W).reshape(Vs, D, Hc // TH, TH, Wc // TW, TW)
    vv = v[:, :, :Hc, :Wc].clamp(-1, H).reshape(Vs, D, Hc // TH, TH, Wc // TW, TW)
    x0 = uu.floor().amin(dim=(1, 3, 5))
    x1 = uu.floor().amax(dim=(1, 3, 5)) + 1
    y0 = vv.floor().amin(dim=(1, 3, 5))
    y1 = vv.floor().amax(dim=(1, 3, 5)) + 1
    return (x1 - x0 + 1), (y1 - y0 + 1)


def summarize(bw, bh, C, TH, TW, D):
    area = (bw * bh).flatten().float()
    q = torch.tensor([0.5, 0.9, 0.99], device=area.device)
    qs = torch.quantile(area[: min(area.numel(), 4_000_000)], q).tolist()
    samples = TH * TW * D
    out = {"tile": [TH, TW], "box_w_mean": bw.float().mean().item(), "box_h_mean": bh.float().mean().item(),
           "box_w_max": bw.max().item(), "box_h_max": bh.max().item(),
           "area_mean": area.mean().item(), "area_p50": qs[0], "area_p90": qs[1], "area_p99": qs[2], "area_max": area.max().item(),
           "taps_per_unique_texel_mean": (4.0 * samples / area).mean().item(),
           "lds_bytes_mean": area.mean().item() * C * 4, "lds_bytes_p99": qs[2] * C * 4}
    for budget in (32, 48, 64, 96):
        out["fits_%dKB" % budget] = (area * C * 4 <= budget * 1024).float().mean().item()
    return out


TILES = {1: [(4, 16), (8, 8), (8, 16), (4, 32)], 2: [(4, 16), (8, 8), (8, 16), (4, 32)], 3: [(8, 16), (8, 32), (4, 32), (16, 16)],
         4: [(8, 16), (8, 32), (4, 64), (16, 16), (8, 64)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/footprint_stats.json")
    args = ap.parse_args()
    import mvsformer_amd as m
    from mvsformer_amd import synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, scene = synth.make_inputs(args.views, args.height, args.width, seed=args.seed, device=dev)
    out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    torch.cuda.synchronize()
    report = {"config": vars(args), "stages": {}}
    for i in range(1, 5):
        hyp = out["stage%d" % i]["depth_values"][0]
        D, H, W = hyp.shape
        C = feats["stage%d" % i].shape[2]
        u, v = sweep_uv(proj["stage%d" % i][0], hyp)
        st = {"C": C, "D": D, "H": H, "W": W, "tiles": []}
        # how far apart are adjacent hypotheses / adjacent pixels in the source image
        du = (u[:, 1:] - u[:, :-1]).abs()
        dvv = (v[:, 1:] - v[:, :-1]).abs()
        st["plane_step_px_mean"] = torch.sqrt(du * du + dvv * dvv).mean().item()
        dx = torch.sqrt((u[..., 1:] - u[..., :-1]) ** 2 + (v[..., 1:] - v[..., :-1]) ** 2)
        st["pixel_step_px_mean"] = dx.mean().item()
        st["pixel_step_px_p99"] = torch.quantile(dx.flatten()[:4_000_000], 0.99).item()
        # smoothness of the hypothesis map itself (stage >= 2: predicted by the random-weight network)
        st["depth_rel_jump_mean"] = ((hyp[:, :, 1:] - hyp[:, :, :-1]).abs() / hyp[:, :, 1:]).mean().item()
        for TH, TW in TILES[i]:
            bw, bh = tile_boxes(u, v, TH, TW, H, W)
            st["tiles"].append(summarize(bw, bh, C, TH, TW, D))
        report["stages"]["stage%d" % i] = st
        print("stage%d C=%d D=%d %dx%d plane step %.2f px, pixel step %.2f px (p99 %.2f), depth jump %.2e" % (
            i, C, D, H, W, st["plane_step_px_mean"], st["pixel_step_px_mean"], st["pixel_step_px_p99"], st["depth_rel_jump_mean"]))
        for t in st["tiles"]:
            print("   tile %2dx%-2d box %.1fx%.1f (max %dx%d) area mean %.0f p99 %.0f  taps/texel %.1f  LDS mean %.1f KB p99 %.1f KB  fits64K %.3f" % (
                t["tile"][0], t["tile"][1], t["box_w_mean"], t["box_h_mean"], t["box_w_max"], t["box_h_max"], t["area_mean"], t["area_p99"],
                t["taps_per_unique_texel_mean"], t["lds_bytes_mean"] / 1024, t["lds_bytes_p99"] / 1024, t["fits_64KB"]))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__" and "--windows" not in sys.argv:
    main()


def window_outliers(u, v, TH, TW, WX, WY, H, W):
    """Fraction of samples whose 2x2 taps do NOT fall inside a fixed WX x WY texel window centred on the projection of the tile's
    centre pixel at its middle plane (the 'software cache' form of the tiled sweep: no bounding-box reduction, per-sample fallback)."""
    Vs, D = u.shape[:2]
    Hc, Wc = (H // TH) * TH, (W // TW) * TW
    uu = u[:, :, :Hc, :Wc].clamp(-1, W).reshape(Vs, D, Hc // TH, TH, Wc // TW, TW)
    vv = v[:, :, :Hc, :Wc].clamp(-1, H).reshape(Vs, D, Hc // TH, TH, Wc // TW, TW)
    cu = uu[:, D // 2, :, TH // 2, :, TW // 2].floor()[:, None, :, None, :, None]     # centre pixel, middle plane
    cv = vv[:, D // 2, :, TH // 2, :, TW // 2].floor()[:, None, :, None, :, None]
    x0, y0 = uu.floor() - (cu - WX // 2), vv.floor() - (cv - WY // 2)
    inside = (x0 >= 0) & (x0 + 1 < WX) & (y0 >= 0) & (y0 + 1 < WY)
    out = (~inside)
    frac = out.float().mean().item()
    # fraction of (wavefront = 64 consecutive pixels of the tile, plane) steps that contain at least one outlier
    per_wave = out.permute(0, 1, 2, 4, 3, 5).reshape(Vs, D, Hc // TH, Wc // TW, (TH * TW) // 64, 64).any(-1).float().mean().item()
    return frac, per_wave


def window_report():
    import mvsformer_amd as m
    from mvsformer_amd import synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = m.CascadeMVS().eval()
    m.randomize_bn_(net, seed=1)
    net = net.to(dev)
    feats, proj, dv, scene = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
    out = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    torch.cuda.synchronize()
    for i in (2, 3, 4):
        hyp = out["stage%d" % i]["depth_values"][0]
        D, H, W = hyp.shape
        u, v = sweep_uv(proj["stage%d" % i][0], hyp)
        for dchunk in ([slice(0, D)] if D <= 4 else [slice(0, 4), slice(0, D)]):
            for (TH, TW) in ((16, 16), (8, 32)):
                for (WX, WY) in ((32, 32), (40, 32), (48, 24), (48, 32), (64, 24), (64, 32), (48, 48)):
                    f, pw = window_outliers(u[:, dchunk], v[:, dchunk], TH, TW, WX, WY, H, W)
                    print("stage%d planes %d tile %2dx%-2d window %2dx%-2d (%4d texels): outlier samples %.4f, (wave,plane) steps with an outlier %.3f" % (
                        i, dchunk.stop - dchunk.start, TH, TW, WX, WY, WX * WY, f, pw))


if __name__ == "__main__" and "--windows" in sys.argv:
    window_report()
