#!/usr/bin/env python
"""Pre-split ViT kernels (csrc/vit_packed.hip) alone at the bench's size (5 views, 768x576 -> 1729 tokens padded to 1760 per image, C = 384, 6 heads):
median launch time over interleaved rounds, TFLOP/s against the split-form peak (2500 / 6), next to the split-on-the-fly kernels of csrc/vit.hip."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N, C, NH = 5, 1729, 384, 6
Np = (N + 31) // 32 * 32
M = B * Np
PEAK = 2500.0 / 6


def timeit(fn, rounds=7, inner=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


x = torch.randn(M, C, device=dev)
g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
rows = []
xp = ops.Packed(M, C, dev)
rows.append(("layernorm_x3p", timeit(lambda: ops.layernorm_x3p(x, g, b, 1e-6, Np, N, out=xp)), 0.0))
rows.append(("layernorm (fp32 out)", timeit(lambda: ops.layernorm(x, g, b, 1e-6)), 0.0))
for name, n, k, act in (("qkv-shaped plain", 3 * C, C, 0), ("proj", C, C, 0), ("fc1+gelu", 4 * C, C, 1), ("fc2", C, 4 * C, 0)):
    A = torch.randn(M, k, device=dev)
    W = torch.randn(n, k, device=dev) * k ** -0.5
    bias = torch.randn(n, device=dev)
    res = torch.randn(M, n, device=dev)
    Ap, Wp = ops.x3p_pack(A), ops.x3p_pack(W)
    Cout = torch.empty(M, n, device=dev)
    outp = ops.Packed(M, n, dev)
    fl = 2.0 * M * n * k
    if name.startswith("fc1"):
        rows.append((name + " x3p -> packed", timeit(lambda: ops.gemm_x3p(Ap, Wp, n, shift=bias, act=act, out=outp)), fl))
    else:
        rows.append((name + " x3p -> fp32+res", timeit(lambda: ops.gemm_x3p(Ap, Wp, n, C=Cout, shift=bias, act=act, res=res)), fl))
    rows.append((name + " split-on-the-fly", timeit(lambda: ops.gemm_x3(A, W, Cout, M, n, k, k, k, n, shift=bias, act=act, res=res)), fl))
W = torch.randn(3 * C, C, device=dev) * C ** -0.5
bias = torch.randn(3 * C, device=dev) * 0.1
Wp = ops.x3p_pack(W)
ops.layernorm_x3p(x, g, b, 1e-6, Np, N, out=xp)
qkv_p = ops.QkvPacked(B, NH, Np, dev)
rows.append(("qkv x3p -> packed Q/K/V^T", timeit(lambda: ops.gemm_x3p_qkv(xp, Wp, bias, B, Np, NH, 0.125, out=qkv_p)), 2.0 * M * 3 * C * C))
ap = ops.Packed(M, C, dev)
afl = 4.0 * B * NH * N * N * 64
rows.append(("attention x3p", timeit(lambda: ops.attention_x3p(qkv_p, N, out=ap)), afl))
qkv = torch.randn(B, N, 3 * C, device=dev)
vt = ops.attention_vt(qkv, NH)
rows.append(("attention split-on-the-fly", timeit(lambda: ops.attention_x3(qkv, vt, NH, 0.125)), afl))
rows.append(("cls attention row", timeit(lambda: ops.cls_attention_x3p(qkv_p, N)), 0.0))
print("%-34s %9s %9s %9s %7s" % ("kernel", "median ms", "min ms", "TFLOP/s", "of peak"))
for name, (med, mn), fl in rows:
    tf = fl / (med * 1e-3) / 1e12 if fl else 0.0
    print("%-34s %9.4f %9.4f %9.1f %7.3f" % (name, med, mn, tf, tf / PEAK))
