#!/usr/bin/env python
"""Where a pass of x3_conv_kernel spends its time (experiment build: make -C mvsformer_amd/csrc exp NAME=tl EXPSRC=conv3d_x3 EXPFLAGS=-DX3_TIMELINE,
run with MVS_HIP_LIB=.../libmvs_hip_tl.so): s_memtime ticks of wavefront 0 per phase, summed over the blocks of one launch."""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvsformer_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
tl = ctypes.CDLL(_lib.LIB_PATH).mvs_x3_timeline
tl.argtypes = [ctypes.c_void_p, ctypes.c_int]
PH = ["issue loads", "barrier (prev reads)", "wait loads", "split + LDS stores", "barrier", "MFMA phase", "store plane"]
STAGES = {3: (8, 576, 768), 4: (4, 1152, 1536)}
for st in (3, 4):
    D, H, W = STAGES[st]
    dims = [(D, H, W), (D, H // 2, W // 2), (D, H // 4, W // 4), (D, H // 8, W // 8)]
    for name, cin, cout, lvl, stride in (("conv1", 8, 16, 0, (1, 2)), ("conv2", 16, 16, 1, (1, 1)), ("conv3", 16, 32, 1, (1, 2)), ("conv4", 32, 32, 2, (1, 1)),
                                         ("conv5", 32, 64, 2, (1, 2)), ("conv6", 64, 64, 3, (1, 1))):
        d, h, w = dims[lvl]
        x = torch.randn(1, cin, d, h, w, device=dev)
        wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        px = ops.conv3d_x3_pack(wt, stride)
        for _ in range(3):
            ops.conv3d_x3(x, px, cin, cout, stride, scale, shift, None, True)
        torch.cuda.synchronize()
        tl(None, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.conv3d_x3(x, px, cin, cout, stride, scale, shift, None, True)
        b.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        tl(buf, 1)
        tot = float(sum(buf[:7])) or 1.0
        print("stage%d %s %d->%d s%s: %.4f ms | " % (st, name, cin, cout, stride, a.elapsed_time(b)) +
              "  ".join("%s %.0f%%" % (PH[i], 100.0 * buf[i] / tot) for i in range(7)) + "  | ticks/launch %.3g" % tot, flush=True)

# the fused tail (conv11 + BatchNorm + ReLU + skip + prob): make ... EXPSRC="conv3d_x3 tail_x3"
ttl = getattr(ctypes.CDLL(_lib.LIB_PATH), "mvs_tail_timeline", None)
if ttl is not None:
    ttl.argtypes = [ctypes.c_void_p, ctypes.c_int]
    TPH = ["issue loads", "MFMA phase", "epilogue + stores", "wait loads", "reduce skip", "split + LDS stores", "barrier"]
    for st in (3, 4):
        D, H, W = STAGES[st]
        h, w = H // 2, W // 2
        x = torch.randn(1, 16, D, h, w, device=dev)
        wt = torch.randn(16, 8, 3, 3, 3, device=dev) * 0.05
        res = torch.randn(1, 8, D, H, W, device=dev)
        scale, shift = torch.rand(8, device=dev) + 0.5, torch.randn(8, device=dev)
        pw, pb = torch.randn(8, device=dev), torch.randn(1, device=dev)
        pk = ops.tail_x3_pack(wt)
        for _ in range(3):
            ops.tail_x3(x, pk, scale, shift, res, pw, pb, True)
        torch.cuda.synchronize()
        ttl(None, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.tail_x3(x, pk, scale, shift, res, pw, pb, True)
        b.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        ttl(buf, 1)
        tot = float(sum(buf[:7])) or 1.0
        print("stage%d tail 16->8+prob: %.4f ms | " % (st, a.elapsed_time(b)) + "  ".join("%s %.0f%%" % (TPH[i], 100.0 * buf[i] / tot) for i in range(7)) +
              "  | ticks/launch %.3g" % tot, flush=True)
