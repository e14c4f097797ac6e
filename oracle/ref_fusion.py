"""TEST INFRASTRUCTURE — CPU oracle for the depth-map consistency filter.

Restates reference misc/fusion.py:8-122 (get_pixel_grids, idx_img2cam, idx_cam2world, idx_world2cam, idx_cam2img,
project_img, prob_filter, get_reproj, vis_filter, ave_fusion) and the driver test.py:404-438 (filter_depth's
per-sample block) for CPU tensors.  Pinned by tests/golden/fusion.npz, generated from the real reference module by
oracle/gen_golden.py (with `Tensor.cuda` made a no-op, the reference hard-codes `.cuda()` in get_pixel_grids).
Never imported by the product path.
"""
import torch
import torch.nn.functional as F


def pixel_centres(h, w):
    """fusion.py:8-13 — [h,w,3,1] homogeneous pixel centres (x+0.5, y+0.5, 1)."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
    return torch.stack([xs, ys, torch.ones_like(xs)], -1).unsqueeze(-1)


def _K(cam):
    return cam[:, 1:2, :3, :3].unsqueeze(1)          # [N,1,1,3,3]


def _E(cam):
    return cam[:, 0:1].unsqueeze(1)                  # [N,1,1,4,4]


def img_to_cam(idx_img, depth, cam):
    """fusion.py:23-28."""
    c = _K(cam).inverse() @ idx_img
    c = c / (c[..., -1:, :] + 1e-9) * depth.permute(0, 2, 3, 1).unsqueeze(4)
    return torch.cat([c, torch.ones_like(c[..., -1:, :])], -2)


def cam_to_world(c, cam):
    """fusion.py:31-34."""
    wld = _E(cam).inverse() @ c
    return wld / (wld[..., -1:, :] + 1e-9)


def world_to_cam(wld, cam):
    """fusion.py:37-40."""
    c = _E(cam) @ wld
    return c / (c[..., -1:, :] + 1e-9)


def cam_to_img(c, cam):
    """fusion.py:43-47."""
    p = _K(cam) @ (c[..., :3, :] / (c[..., 3:4, :] + 1e-9))
    return p / (p[..., -1:, :] + 1e-9)


def project_img(src_img, dst_depth, src_cam, dst_cam):
    """fusion.py:50-66."""
    h, w = src_img.shape[-2:]
    grid = pixel_centres(h, w).unsqueeze(0)
    q = cam_to_img(world_to_cam(cam_to_world(img_to_cam(grid, dst_depth, dst_cam), dst_cam), src_cam), src_cam)
    warp = q[..., :2, 0].clone()
    warp[..., 0] /= w
    warp[..., 1] /= h
    warp = (warp * 2 - 1).clamp(-1.1, 1.1)
    ok = (-1 <= warp[..., 0]) & (warp[..., 0] <= 1) & (-1 <= warp[..., 1]) & (warp[..., 1] <= 1)
    out = F.grid_sample(src_img, warp, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out, ok.to(src_img.dtype).unsqueeze(1)


def prob_filter(ref_prob, prob_thresh):
    """fusion.py:69-77."""
    mask = None
    for i, p in enumerate(prob_thresh):
        m = ref_prob[:, [i]] > p
        mask = m if mask is None else (mask & m)
    return mask


def get_reproj(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """fusion.py:80-98."""
    n, v, _, h, w = srcs_depth.shape
    sd = srcs_depth.reshape(n * v, 1, h, w)
    sc = srcs_cam.reshape(n * v, 2, 4, 4)
    rd = ref_depth.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 1, h, w)
    rc = ref_cam.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 2, 4, 4)
    grid = pixel_centres(h, w).unsqueeze(0)
    s2r_cam = world_to_cam(cam_to_world(img_to_cam(grid, sd, sc), sc), rc)
    s2r_img = cam_to_img(s2r_cam, rc)
    xyd = torch.cat([s2r_img[..., :2, 0], s2r_cam[..., 2:3, 0]], -1).permute(0, 3, 1, 2)
    rep, ok = project_img(xyd, rd, sc, rc)
    return rep.view(n, v, 3, h, w), ok.view(n, v, 1, h, w)


def vis_filter(ref_depth, reproj_xyd, in_range, img_dist_thresh, depth_thresh, vthresh):
    """fusion.py:101-109."""
    n, v, _, h, w = reproj_xyd.shape
    xy = pixel_centres(h, w).permute(3, 2, 0, 1).unsqueeze(1)[:, :, :2]
    dist = (reproj_xyd[:, :, :2] - xy).norm(dim=2, keepdim=True) < img_dist_thresh
    rd = ref_depth.unsqueeze(1)
    dep = (rd - reproj_xyd[:, :, 2:]).abs() < torch.max(rd, reproj_xyd[:, :, 2:]) * depth_thresh
    masks = torch.min(torch.min(in_range, dist.to(ref_depth.dtype)), dep.to(ref_depth.dtype))
    return masks, masks.sum(dim=1) >= (vthresh - 1.1)


def ave_fusion(ref_depth, reproj_xyd, masks):
    """fusion.py:112-114."""
    return ((reproj_xyd[:, :, 2:] * masks).sum(dim=1) + ref_depth) / (masks.sum(dim=1) + 1)


def filter_depth_maps(ref_depth, src_depths, ref_cam, src_cams, thres_disp, depth_thresh, thres_view):
    """test.py:425-434 — the geometric part of filter_depth for one batch."""
    reproj, in_range = get_reproj(ref_depth, src_depths, ref_cam, src_cams)
    masks, mask = vis_filter(ref_depth, reproj, in_range, thres_disp, depth_thresh, thres_view)
    ave = ave_fusion(ref_depth, reproj, masks)
    grid = pixel_centres(*ave.shape[-2:]).unsqueeze(0)
    points = cam_to_world(img_to_cam(grid, ave, ref_cam), ref_cam)[..., :3, 0].permute(0, 3, 1, 2)
    return dict(reproj_xyd=reproj, in_range=in_range, masks=masks, mask=mask, ref_depth_ave=ave, points=points)


def make_fusion_case(n=1, v=4, h=48, w=64, seed=0, noise=0.004, outlier_frac=0.05):
    """Deterministic test scene: a slanted plane seen by 1+v DTU-like cameras, per-view depth maps rendered
    analytically, perturbed by relative noise plus a fraction of gross outliers and zeroed (prob-filtered) pixels."""
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4).repeat(n, 1 + v, 1, 1)
    f = 1.1 * w
    K[..., 0, 0] = f
    K[..., 1, 1] = f
    K[..., 0, 2] = w / 2
    K[..., 1, 2] = h / 2
    E = torch.eye(4).repeat(n, 1 + v, 1, 1)
    for b in range(n):
        for i in range(1, 1 + v):
            ang = (torch.rand(3, generator=g) - 0.5) * 0.12
            cx, sx, cy, sy, cz, sz = ang[0].cos(), ang[0].sin(), ang[1].cos(), ang[1].sin(), ang[2].cos(), ang[2].sin()
            Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
            Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
            E[b, i, :3, :3] = Rz @ Ry @ Rx
            E[b, i, :3, 3] = (torch.rand(3, generator=g) - 0.5) * torch.tensor([120.0, 80.0, 20.0])
    # world plane  nrm . X = d0  (world == reference camera frame)
    nrm = torch.tensor([0.15, -0.1, 1.0])
    nrm = nrm / nrm.norm()
    d0 = 600.0
    grid = pixel_centres(h, w)[..., 0]               # [h,w,3]
    depths = []
    for b in range(n):
        per = []
        for i in range(1 + v):
            R, t = E[b, i, :3, :3], E[b, i, :3, 3]
            rays = grid @ torch.linalg.inv(K[b, i, :3, :3]).T          # cam rays with z=1
            # X_world = R^T (z*ray - t);  nrm.X = d0  ->  z = (d0 + nrm.R^T t) / (nrm.R^T ray)
            nr = R @ nrm
            z = (d0 + (nr * t).sum()) / (rays @ nr)
            per.append(z)
        depths.append(torch.stack(per))
    depth = torch.stack(depths)                      # [n,1+v,h,w]
    depth = depth * (1 + noise * torch.randn(depth.shape, generator=g))
    out = torch.rand(depth.shape, generator=g)
    depth = torch.where(out < outlier_frac, depth * (1 + 0.2 * torch.randn(depth.shape, generator=g)), depth)
    depth = torch.where((out > 1 - outlier_frac) & (torch.arange(1 + v).view(1, -1, 1, 1) > 0), torch.zeros_like(depth), depth)
    cams = torch.stack([E, K], dim=2)                # [n,1+v,2,4,4]
    return dict(ref_depth=depth[:, :1].contiguous(), src_depths=depth[:, 1:].unsqueeze(2).contiguous(),
                ref_cam=cams[:, 0].contiguous(), src_cams=cams[:, 1:].contiguous())


def get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """fusion.py:116-150."""
    n, v, _, h, w = srcs_depth.shape
    sd = srcs_depth.reshape(n * v, 1, h, w)
    sc = srcs_cam.reshape(n * v, 2, 4, 4)
    rc = ref_cam.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 2, 4, 4)
    rd = ref_depth.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 1, h, w)
    grid = pixel_centres(h, w).unsqueeze(0)
    r2s_img = cam_to_img(world_to_cam(cam_to_world(img_to_cam(grid, rd, rc), rc), sc), sc)
    q = r2s_img[..., :2, 0]
    gx = q[..., 0] / ((w - 1) / 2) - 1
    gy = q[..., 1] / ((h - 1) / 2) - 1
    warped = F.grid_sample(sd, torch.stack((gx, gy), -1), mode="bilinear", padding_mode="zeros", align_corners=True)
    qh = torch.cat([q, torch.ones_like(q[..., -1:])], -1).unsqueeze(-1)
    s2r_cam = world_to_cam(cam_to_world(img_to_cam(qh, warped, sc), sc), rc)
    depth = s2r_cam[:, :, :, 2, 0].clone()
    xy = cam_to_img(s2r_cam, rc)
    return torch.cat([xy[..., :2, 0], depth.unsqueeze(-1)], -1).permute(0, 3, 1, 2).reshape(n, v, 3, h, w)


def vis_filter_dynamic(ref_depth, reproj_xyd, dist_base=4, rel_diff_base=1300):
    """fusion.py:153-165 -> (masks [n,v,v-1,h,w] bool, mask [n,v,1,h,w] bool)."""
    n, v, _, h, w = reproj_xyd.shape
    xy = pixel_centres(h, w).permute(3, 2, 0, 1).unsqueeze(1)[:, :, :2]
    cd = (reproj_xyd[:, :, :2] - xy).norm(dim=2, keepdim=True)
    dd = (ref_depth.unsqueeze(1) - reproj_xyd[:, :, 2:]).abs() / ref_depth.unsqueeze(1)
    levels = torch.arange(2, v + 1).reshape(1, 1, -1, 1, 1).repeat(n, v, 1, 1, 1)
    masks = torch.min(cd < levels / dist_base, dd < levels / rel_diff_base)
    return masks, masks[:, :, -1:]


def dynamic_filter_depth_maps(ref_depth, src_depths, ref_cam, src_cams, dist_base=4, rel_diff_base=1300):
    """test.py:494-514 — the geometric part of dynamic_filter_depth for one batch."""
    v = src_depths.shape[1]
    reproj = get_reproj_dynamic(ref_depth, src_depths, ref_cam, src_cams)
    masks, vis_mask = vis_filter_dynamic(ref_depth, reproj, dist_base, rel_diff_base)
    rdepth = reproj[:, :, -1].clone()
    rdepth[~vis_mask.squeeze(2)] = 0
    sums = masks.sum(dim=1)
    vsum = vis_mask.sum(dim=1)
    ave = (rdepth.sum(dim=1, keepdim=True) + ref_depth) / (vsum + 1)
    geo = vsum >= v + 1
    for i in range(2, v + 1):
        geo = torch.logical_or(geo, sums[:, i - 2:i - 1] >= i)
    grid = pixel_centres(*ave.shape[-2:]).unsqueeze(0)
    points = cam_to_world(img_to_cam(grid, ave, ref_cam), ref_cam)[..., :3, 0].permute(0, 3, 1, 2)
    return dict(reproj_xyd=reproj, masks=masks, vis_mask=vis_mask, geo_mask=geo, ref_depth_ave=ave, points=points)
