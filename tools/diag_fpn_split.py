"""Development aid: where does the bf16 split form of the last FPN level differ from the fp32 form?  (GPU only)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mvsformer_amd import FPNDecoder
from oracle import ref_fpn
torch.manual_seed(7)
dec = FPNDecoder([8, 16, 32, 64]); ref_fpn.randomize_bn(dec, 8); dec = dec.eval().cuda()
N, h, w = 2, 144, 192
feats = [f.cuda() for f in ref_fpn.make_case(8, N, h, w)]
os.environ["MVS_FPN_SPLIT"] = "0"; a = dec(*feats)[3].permute(0, 2, 3, 1).contiguous()
os.environ["MVS_FPN_SPLIT"] = "1"; b = dec(*feats)[3].permute(0, 2, 3, 1).contiguous()
b2 = dec(*feats)[3].permute(0, 2, 3, 1).contiguous()
print("split run-to-run identical:", torch.equal(b, b2), " max |b-b2| %.3e" % (b - b2).abs().max().item())
sc = a.abs().max().item()
d = (a - b).abs().amax(3)                                   # [N,H,W]
bad = d > 1e-4 * sc
print("bad pixels", int(bad.sum()), "of", bad.numel())
tiles = bad.view(N, 2 * h * 4 // 4, 4, 2 * w * 4 // 32, 32).any(4).any(2) if False else None
H, W = 8 * h, 8 * w
bt = bad.view(N, H // 4, 4, W // 32, 32).permute(0, 1, 3, 2, 4)      # [N,ty,tx,4,32]
tb = bt.reshape(N, H // 4, W // 32, -1).any(-1)
idx = tb.nonzero()
print("bad tiles", idx.shape[0], "of", tb.numel(), "first", idx[:10].tolist())
print("bad tile ty%2 hist", torch.bincount(idx[:, 1] % 2, minlength=2).tolist(), "tx hist(min,max)", idx[:, 2].min().item(), idx[:, 2].max().item(),
      "ty (min,max)", idx[:, 1].min().item(), idx[:, 1].max().item())
for k in range(min(3, idx.shape[0])):
    n, ty, tx = idx[k].tolist()
    print("tile", n, ty, tx)
    for r in range(4):
        print("   ", "".join("X" if bt[n, ty, tx, r, c] else "." for c in range(32)))
# source coordinate of the first bad tile: is it near a place where the coarse window clamps?
n, ty, tx = idx[0].tolist()
print("tile origin y0,x0 =", ty * 4, tx * 32, " sy*(y0-1) = %.4f  sx*(x0-1) = %.4f" % ((4 * h - 1) / (8 * h - 1) * (ty * 4 - 1), (4 * w - 1) / (8 * w - 1) * (tx * 32 - 1)))
