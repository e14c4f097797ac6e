#!/usr/bin/env python
"""Per-kernel time of the DINO ViT-small branch (csrc/vit.hip) at the bench's size: 5 views, 1536x1152 -> 768x576, from HIP events around every
C-ABI launch + the wall time of the whole branch (torch glue included)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import ops, vit as V
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = V.vit_small(patch_size=16, qk_scale="default").eval().to(dev)
dec = V.VITDecoderStage4Single(dict(out_ch=64, vit_ch=384, att_fusion=True, nhead=6)).eval().to(dev)
img = torch.randn(5, 3, 1152, 1536, device=dev)
for _ in range(2):
    V.vit_branch(net, dec, img)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    V.vit_branch(net, dec, img)
b.record()
torch.cuda.synchronize()
print("whole branch: %.3f ms per 5 views" % (a.elapsed_time(b) / 5))
with ops.kernel_timer() as kt:
    V.vit_branch(net, dec, img)
summ = kt.summary()
order = kt.order
# GEMMs by shape: re-tag through the work table is not available per call; list the launches in order with their times
seen = {}
tot = 0.0
rows = []
for tag in order:
    j = seen.get(tag, 0); seen[tag] = j + 1
    ms = summ[tag]["all_ms"][j]
    rows.append((tag, ms)); tot += ms
agg = {}
for t, ms in rows:
    agg.setdefault(t, [0, 0.0]); agg[t][0] += 1; agg[t][1] += ms
for t, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-20s calls %4d  total %8.3f ms" % (t, n, ms))
print("sum of launches %.3f ms" % tot)
# the launches of block 0 in order (flash form: layernorm, qkv GEMM, attention, proj GEMM, layernorm, fc1, fc2)
print("first launches: " + " | ".join("%s %.3f" % (t, ms) for t, ms in rows[:12]))
