"""TEST INFRASTRUCTURE — CPU oracle for the training losses: restates reference
models/losses.py:304-350 (ce_loss_stage4, focal=False), :353-408 (mixup_ce_loss_stage4) and :51-85 (reg_loss_stage4) for one stage,
in the order of operations the reference uses.  Pinned by tests/golden/ce_loss.npz and other_losses.npz (values and gradients produced
by the real functions).  Never imported by the product path."""
import torch
import torch.nn.functional as F


def gt_bins(depth_values, depth_gt, mask, inverse_depth=True):
    """losses.py:309-335 -> (gt_index [B,H,W] long, final_mask [B,H,W] bool), index in flipped order when inverse."""
    gt = depth_gt.unsqueeze(1)
    dv = torch.flip(depth_values, dims=[1]) if inverse_depth else depth_values
    half = (dv[:, 1:] - dv[:, :-1]).abs() / 2
    half = torch.cat([half, half[:, -1:]], dim=1)
    lo, hi = dv[:, 0:1] - half[:, 0:1], dv[:, -1:] + half[:, -1:]
    outside = torch.clamp((gt < lo).float() + (gt > hi).float(), 0, 1)
    final = ((1 - outside).squeeze(1) * (mask > 0.5).float()).bool()
    index = ((dv + half) <= gt.expand_as(dv)).float().sum(dim=1, keepdim=True).long()
    return torch.clamp_max(index, dv.shape[1] - 1).squeeze(1), final


def ce_loss_stage(prob_volume_pre, depth_values, depth_gt, mask, inverse_depth=True, weight=1.0):
    index, final = gt_bins(depth_values, depth_gt, mask, inverse_depth)
    logits = torch.flip(prob_volume_pre, dims=[1]) if inverse_depth else prob_volume_pre
    return weight * F.cross_entropy(logits.permute(0, 2, 3, 1)[final, :], index[final], reduction="mean")


def ce_loss_stage4(inputs, depth_gt_ms, mask_ms, dlossw, inverse_depth=True):
    return {k: ce_loss_stage(inputs[k]["prob_volume_pre"].float(), inputs[k]["depth_values"], depth_gt_ms[k], mask_ms[k], inverse_depth,
                             1.0 if dlossw is None else dlossw[int(k[-1]) - 1]) for k in ("stage1", "stage2", "stage3", "stage4")}


def mixup_ce_loss_stage(prob_volume_pre, depth_values, depth_gt, mask, inverse_depth=True, weight=1.0):
    """losses.py:358-399 for one stage."""
    gt = depth_gt.unsqueeze(1)
    maskf = (mask > 0.5).float()
    dv = torch.flip(depth_values, dims=[1]) if inverse_depth else depth_values
    logits = torch.flip(prob_volume_pre, dims=[1]) if inverse_depth else prob_volume_pre
    outside = torch.clamp((gt < dv[:, 0:1]).float() + (gt > dv[:, -1:]).float(), 0, 1)
    final = (1 - outside).squeeze(1) * maskf
    index = (dv[:, 1:] <= gt.expand_as(dv[:, :-1])).float().sum(dim=1, keepdim=True).long()
    index = torch.clamp_max(index, dv.shape[1] - 2).squeeze(1)
    left = torch.gather(dv[:, :-1], 1, index.unsqueeze(1))
    itv = torch.gather((dv[:, 1:] - dv[:, :-1]).abs(), 1, index.unsqueeze(1))
    wl = torch.clamp((gt - left).abs() / itv, 0, 1).squeeze(1)
    wr = 1 - wl
    ll = F.cross_entropy(logits[:, :-1], index, reduction="none")
    lr = F.cross_entropy(logits[:, 1:], index, reduction="none")
    den = final.sum() + 1e-6
    return weight * ((ll * wl * final).sum() / den + (lr * wr * final).sum() / den)


def mixup_ce_loss_stage4(inputs, depth_gt_ms, mask_ms, dlossw, inverse_depth=True):
    return {k: mixup_ce_loss_stage(inputs[k]["prob_volume_pre"].float(), inputs[k]["depth_values"], depth_gt_ms[k], mask_ms[k], inverse_depth,
                                   1.0 if dlossw is None else dlossw[int(k[-1]) - 1]) for k in ("stage1", "stage2", "stage3", "stage4")}


def reg_loss_stage(depth, depth_values, depth_gt, mask, interval, mask_out_range=False, inverse_depth=True, weight=1.0):
    """losses.py:56-84 for one stage; ``interval [B]``."""
    itv = interval.reshape(-1, 1, 1)
    sel = mask > 0.5
    if mask_out_range:
        dv = torch.flip(depth_values, dims=[1]) if inverse_depth else depth_values
        half = (dv[:, 1:] - dv[:, :-1]).abs() / 2
        half = torch.cat([half, half[:, -1:]], dim=1)
        lo, hi = dv[:, 0] - half[:, 0], dv[:, -1] + half[:, -1]
        outside = torch.clamp((depth_gt < lo).float() + (depth_gt > hi).float(), 0, 1)
        sel = sel & (1 - outside).bool()
    return weight * F.smooth_l1_loss((depth / itv)[sel], (depth_gt / itv)[sel], reduction="mean")


def reg_loss_stage4(inputs, depth_gt_ms, mask_ms, dlossw, depth_interval, mask_out_range=False, inverse_depth=True):
    return {k: reg_loss_stage(inputs[k]["depth"], inputs[k]["depth_values"], depth_gt_ms[k], mask_ms[k], depth_interval, mask_out_range,
                              inverse_depth, 1.0 if dlossw is None else dlossw[int(k[-1]) - 1]) for k in ("stage1", "stage2", "stage3", "stage4")}


def make_loss_case(seed=0, B=2, sizes=((32, 8, 12), (16, 16, 24), (8, 24, 32), (4, 40, 56)), inverse_depth=True):
    """Four stages of (D,H,W): inverse-depth hypothesis columns around a smooth surface, logits peaked near the truth plus
    noise, ground truth partly outside the hypothesis range, partly masked out, some landing exactly on a bin edge."""
    g = torch.Generator().manual_seed(seed)
    inputs, gts, masks = {}, {}, {}
    for s, (D, H, W) in enumerate(sizes):
        ys, xs = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        surf = (600 + 120 * xs - 80 * ys + 25 * torch.sin(6 * xs)).expand(B, H, W).clone()
        width = 0.25 / (s + 1)
        inv_c = 1.0 / surf * (1 + 0.02 * torch.randn(B, H, W, generator=g))
        steps = torch.linspace(-1, 1, D).view(1, D, 1, 1)
        inv = inv_c.unsqueeze(1) * (1 + width * steps)                     # ascending inverse depth = descending depth
        dv = 1.0 / inv if inverse_depth else torch.flip(1.0 / inv, dims=[1])
        gt = surf * (1 + 0.5 * width * torch.randn(B, H, W, generator=g))
        gt[:, : H // 8] *= 1.5                                             # far out of range
        sel = torch.rand(B, H, W, generator=g) < 0.02                      # exactly on a right edge: the <= comparison
        dvf = torch.flip(dv, dims=[1]) if inverse_depth else dv
        edge = dvf[:, D // 2] + (dvf[:, D // 2 + 1] - dvf[:, D // 2]).abs() / 2 if D > 2 else dvf[:, 0]
        gt = torch.where(sel, edge, gt)
        mask = (torch.rand(B, H, W, generator=g) > 0.2).float() * torch.rand(B, H, W, generator=g).clamp_min(0.51)
        mask[:, :, : W // 10] = 0.3                                        # below the 0.5 threshold
        logits = -((dv - gt.unsqueeze(1)).abs() / (surf.unsqueeze(1) * width / D * 2)) + 0.7 * torch.randn(B, D, H, W, generator=g)
        key = "stage%d" % (s + 1)
        inputs[key] = dict(depth_values=dv.contiguous(), prob_volume_pre=logits.contiguous())
        gts[key], masks[key] = gt.contiguous(), mask.contiguous()
    return inputs, gts, masks


def sinkhorn_stage(prob, depth_values, depth_gt, mask, iters=10, eps=1.0, weight=1.0):
    """models/losses.py:88-128 with ``continuous=False``: the log-domain Sinkhorn plan between each pixel's probability column and the one-hot of
    the hypothesis nearest to the ground truth, cost |i - j| / eps (the reference's signs: + cost in the exponent), loss = mean over
    ``mask > 0.5`` of sum_ij T_ij |i - j|.  Plain torch (autograd gives the gradient through all iterations)."""
    B, D, H, W = prob.shape
    ar = torch.arange(D, dtype=torch.float32)
    Dm = (ar[:, None] - ar[None, :]).abs()                                   # [D (i: prediction), D (j: ground truth)]
    gi = (depth_values - depth_gt[:, None]).abs().min(1)[1].reshape(B * H * W)
    mu = torch.zeros(B * H * W, D)
    mu[torch.arange(B * H * W), gi] = 1.0
    nu = prob.permute(0, 2, 3, 1).reshape(B * H * W, D)
    log_mu, log_nu = (mu + 1e-12).log(), (nu + 1e-12).log()
    u, v = torch.zeros_like(log_nu), torch.zeros_like(log_mu)
    for _ in range(iters):
        v = log_mu - torch.logsumexp(Dm[None] / eps + u[:, :, None], dim=1)
        u = log_nu - torch.logsumexp(Dm[None] / eps + v[:, None, :], dim=2)
    T = (Dm[None] / eps + u[:, :, None] + v[:, None, :]).exp()
    sel = (mask > 0.5).reshape(-1)
    return weight * (T * Dm[None]).reshape(B * H * W, -1)[sel].sum(-1).mean()


def wasserstein_loss(inputs, depth_gt_ms, mask_ms, dlossw, ot_iter=10, ot_eps=1.0):
    return {k: sinkhorn_stage(inputs[k]["prob_volume"], inputs[k]["depth_values"], depth_gt_ms[k], mask_ms[k], ot_iter, ot_eps, dlossw[i])
            for i, k in enumerate(k for k in inputs if "stage" in k)}
