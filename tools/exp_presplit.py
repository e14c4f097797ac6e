#!/usr/bin/env python
"""VERDICT r4 item 2 gate: the pre-split activation format (h | m | l bf16 planes channel-last, 6 bytes per value) between conv1 and conv2 of the
stage-4 / stage-3 CostRegNet3D.  Experiment build only:

    make -C mvsformer_amd/csrc exp NAME=presplit EXPSRC=conv3d_x3 EXPFLAGS=-DX3_PRESPLIT
    python tools/exp_presplit.py            # on the GPU box

conv1 (8 -> 16, stride (1,2,2)): as shipped / writing only the pre-split form / writing both; conv2 (16 -> 16, stride 1): as shipped (fp32 NCDHW
in, split in the staging) / staged from the pre-split form by LDS-DMA (no staging registers, no split, no LDS stores).  conv2's two outputs
must be bit-identical (the split is exact)."""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["MVS_HIP_LIB"] = os.path.join(REPO, "mvsformer_amd", "libmvs_hip_presplit.so")
from mvsformer_amd import _lib, ops  # noqa: E402

lib = _lib.load()
P, I = ctypes.c_void_p, ctypes.c_int
lib.mvs_x3_presplit_exp.argtypes = [I, P, P, P, P, P, P, P, I, I, I, I, I, P]
lib.mvs_x3_presplit_exp.restype = I
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for stage, (D, H, W) in (("stage4", (4, 1152, 1536)), ("stage3", (8, 576, 768))):
    torch.manual_seed(0)
    x = torch.randn(1, 8, D, H, W, device=dev)
    w1, w2 = torch.randn(16, 8, 3, 3, 3, device=dev) * 0.1, torch.randn(16, 16, 3, 3, 3, device=dev) * 0.08
    p1, p2 = ops.conv3d_x3_pack(w1, (1, 2)), ops.conv3d_x3_pack(w2, (1, 1))
    sc, sh = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.1
    Ho, Wo = H // 2, W // 2
    y1 = torch.empty(1, 16, D, Ho, Wo, device=dev)
    y1b = torch.empty_like(y1)
    ypre = torch.empty(1, D, Ho, Wo, 3, 16, device=dev, dtype=torch.bfloat16)
    y2a, y2b = torch.empty_like(y1), torch.empty_like(y1)
    st = torch.cuda.current_stream().cuda_stream

    def run(mode, xin, xpre, wp, yout, ypre_out, h, w_):
        rc = lib.mvs_x3_presplit_exp(mode, xin.data_ptr() if xin is not None else None, xpre.data_ptr() if xpre is not None else None, wp.data_ptr(),
                                     sc.data_ptr(), sh.data_ptr(), yout.data_ptr() if yout is not None else None,
                                     ypre_out.data_ptr() if ypre_out is not None else None, 1, D, h, w_, 1, st)
        assert rc == 0, (rc, lib.mvs_last_error())
    run(0, x, None, p1, y1, None, H, W)
    run(2, x, None, p1, y1b, ypre, H, W)
    torch.cuda.synchronize()
    assert torch.equal(y1, y1b), "conv1: fp32 output changed"
    hml = ypre.float().sum(dim=-2).permute(0, 4, 1, 2, 3)          # h + m + l == the fp32 value exactly
    assert torch.equal(hml, y1), "pre-split planes do not add up to the fp32 activations"
    run(3, y1, None, p2, y2a, None, Ho, Wo)
    run(4, None, ypre, p2, y2b, None, Ho, Wo)
    torch.cuda.synchronize()
    same = torch.equal(y2a, y2b)
    t = {
        "conv1 shipped (fp32 out)": timeit(lambda: run(0, x, None, p1, y1, None, H, W)),
        "conv1 pre-split out only": timeit(lambda: run(1, x, None, p1, y1b, ypre, H, W)),
        "conv1 both outputs": timeit(lambda: run(2, x, None, p1, y1b, ypre, H, W)),
        "conv2 shipped (fp32 in)": timeit(lambda: run(3, y1, None, p2, y2a, None, Ho, Wo)),
        "conv2 from pre-split (LDS-DMA staging)": timeit(lambda: run(4, None, ypre, p2, y2b, None, Ho, Wo)),
    }
    print("%s  D=%d %dx%d   conv2 outputs bit-identical: %s" % (stage, D, H, W, same))
    for k, v in t.items():
        print("   %-42s %.4f ms" % (k, v))
    print("   pair: shipped %.4f ms   pre-split %.4f ms" % (t["conv1 shipped (fp32 out)"] + t["conv2 shipped (fp32 in)"],
                                                          t["conv1 pre-split out only"] + t["conv2 from pre-split (LDS-DMA staging)"]))
