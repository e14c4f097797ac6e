#!/usr/bin/env python
"""Target for the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md §HBM): a float4 streaming copy with a known
byte count (the calibration the guide asks for: 1 GiB in, 1 GiB out, larger than the 256 MiB Infinity Cache), then a few
config-2 cascades on ONE stream.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc/fetch -- python $REPO/tools/prof_traffic.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc/write -- python $REPO/tools/prof_traffic.py
    python tools/pmc_traffic.py gpurun_out/pmc/fetch gpurun_out/pmc/write profiles/traffic_by_kernel.json
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvsformer_amd as m  # noqa: E402
from mvsformer_amd import synth  # noqa: E402

CAL_FLOATS = 256 * 1024 * 1024          # 1 GiB of fp32

dev = torch.device("cuda:0")
src = torch.randn(CAL_FLOATS, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)                      # ATen vectorized (16 B/lane) copy kernel: exactly 4*N bytes read, 4*N written
torch.cuda.synchronize()
del src, dst
torch.manual_seed(0)
net = m.CascadeMVS().eval()
m.randomize_bn_(net, seed=1)
net = net.to(dev)
feats, proj, dv, _ = synth.make_inputs(5, 1152, 1536, seed=0, device=dev)
from mvsformer_amd import ops  # noqa: E402
import json  # noqa: E402

tmp = [5.0, 5.0, 5.0, 1.0]
for _ in range(2):
    net(feats, proj, dv, tmp=tmp)
torch.cuda.synchronize()
# the tags bench.py uses, in launch order, for ONE cascade; the LAST cascade of this process (below) is the one tools/pmc_traffic.py aligns with
# them - several launches share a kernel instance (conv4 and conv6, the same layer at two stages), so names alone do not identify a tag
with ops.kernel_timer() as timer:
    net(feats, proj, dv, tmp=tmp)
torch.cuda.synchronize()
out = os.path.join(os.environ.get("MVS_TRAFFIC_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")), "launch_order.json")
json.dump(timer.order, open(out, "w"))
net(feats, proj, dv, tmp=tmp)
torch.cuda.synchronize()
print("done")
