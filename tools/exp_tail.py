#!/usr/bin/env python
"""Where the time of the CostRegNet3D tail (conv11 + BN + ReLU + skip + 1x1x1 prob in one launch) and of conv1 goes at stage-4 shape:
with / without the skip read, against the two-launch form, and a float4 copy of the same bytes for scale.  (GPU box)"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CSRC = os.path.join(REPO, "mvsformer_amd", "csrc")
if "--build" in sys.argv:       # here (no GPU): full libraries with one experiment switch each -> csrc/exp/libmvs_hip_ds1_<N>.so
    os.makedirs(os.path.join(CSRC, "exp"), exist_ok=True)
    srcs = open(os.path.join(CSRC, "Makefile")).read().split("SRCS =")[1].split("\n")[0].split()
    objs = [os.path.join(CSRC, f.replace(".hip", ".o")) for f in srcs if f != "deconv3d_s1.hip"]
    for n in (1, 2, 4, 8, 9):
        obj = os.path.join(CSRC, "exp", "ds1_%d.o" % n)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                               "-DMVS_DS1_EXP=%d" % n, "-c", os.path.join(CSRC, "deconv3d_s1.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs +
                              ["-o", os.path.join(CSRC, "exp", "libmvs_hip_ds1_%d.so" % n)])
    sys.exit(0)
import torch
from mvsformer_amd import _lib
if os.environ.get("MVS_EXP_LIB"):
    _lib.LIB_PATH = os.path.join(CSRC, "exp", "libmvs_hip_ds1_%s.so" % os.environ["MVS_EXP_LIB"])
import mvsformer_amd as m
from mvsformer_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for D, H, W in ((4, 1152, 1536), (8, 576, 768)):
    net = m.CostRegNet3D(8, 8).to(dev).eval()
    m.randomize_bn_(net, seed=1)
    vol = torch.randn(1, 8, D, H, W, device=dev)
    with torch.no_grad():
        y, skip = net._trunk(vol)                       # conv9 output [1,16,D,H/2,W/2], skip = vol
        net.logits(vol)
        key, packed, scale, shift, sd = net._dcache["conv11"]
        w, b = net.prob_params()
        t_full = timeit(lambda: ops.deconv3d_prob1(y, packed, 16, scale, shift, skip, w, b, relu=True))
        t_nores = timeit(lambda: ops.deconv3d_prob1(y, packed, 16, scale, shift, None, w, b, relu=True))
        t_two = timeit(lambda: ops.prob1(ops.deconv3d(y, packed, 16, 8, sd, scale, shift, skip, relu=True), w, b))
        t_conv1 = timeit(lambda: net.conv1(vol))
        t_copy = timeit(lambda: vol.clone())
    gb = vol.numel() * 4 / 1e9
    print("D=%d %dx%d  tail fused %.3f ms (no skip read %.3f, two launches %.3f)  conv1 %.3f ms  clone of the %.0f MB volume %.3f ms" % (
        D, H, W, t_full, t_nores, t_two, t_conv1, gb * 1e3, t_copy))
