"""Which source files define which kernel: the key of the HBM-traffic evidence (profiles/traffic_by_kernel.json).

The rocprofv3 counter passes record, per source file, the digest the kernels were built from; ``bench.py`` reports a kernel's counter traffic only
while the files THAT kernel is compiled from are unchanged - an edit to a training kernel no longer voids the eval kernels' counters (round 5
keyed everything on one digest over all of csrc/).  Measurement plumbing, not a compute path.
"""
from __future__ import annotations

import hashlib
import os
import re
from typing import Dict, List

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "mvs_hip.h")

# kernel-name / launch-tag prefix -> the .hip file with the __global__ function (+ the headers it includes); first match wins
_RULES = [
    (r"^(cv_|nchw_to_nhwc|mvs_nchw_to_nhwc)", ["cost_volume.hip", "common.h", "geometry.h"]),
    (r"^(vis_x3|mvs_vis)", ["vis_net_x3.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^(x3_tail|tail_x3)", ["tail_x3.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^x3_small", ["conv3d_x3_small.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^(x3_conv|x3_deconv|x3_pack|x3_deconv_pack)", ["conv3d_x3.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^conv3d_kernel", ["conv3d_fwd.hip", "conv_common.h", "common.h"]),
    (r"^(deconv3d_kernel|prob3|prob1)", ["conv3d.hip", "conv_common.h", "common.h"]),
    (r"^pack_deconv_s1", ["deconv3d_s1.hip", "conv_common.h", "common.h"]),
    (r"^(head_|init_inverse|schedule_inverse|conf_accumulate|mvs_head|mvs_init_inverse|mvs_schedule_inverse|mvs_conf)", ["head.hip", "common.h"]),
    (r"^(proj_|mvs_proj)", ["proj.hip", "common.h"]),
    (r"^(fpn_level_x3s|fpn_lvl_x3|mvs_fpn_level_x3s)", ["fpn_lvl_x3.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^(fpn8_cp|mvs_fpn_level_cp)", ["fpn_cp.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^(conv2d_x3s|mvs_conv2d_x3s)", ["conv2d_x3s.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^(enc_x3|mvs_conv2d_x3)", ["conv2d_x3.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^(fpn8_x3|fpn_level_x3|mvs_fpn_level_x3)", ["fpn_x3.hip", "conv_common.h", "common.h", "split3.h"]),
    (r"^(x3p_|gemm_x3p|attention_x3p|layernorm_x3p|cls_attention)", ["vit_packed.hip", "common.h", "split3.h"]),
    (r"^(x3_gemm|x3_attention|gemm_x3|attention_x3|layernorm|softmax_rows|bicubic)", ["vit.hip", "common.h", "geometry.h", "split3.h"]),
]


def files_of(kernel: str) -> List[str]:
    """Source files (names inside csrc/) the kernel or launch tag ``kernel`` is built from; every file of csrc/ when no rule knows it."""
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", kernel)
    for pat, files in _RULES:
        if re.match(pat, name):
            return list(files)
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))


def file_digests() -> Dict[str, str]:
    """sha256[:16] of every kernel source and header (+ the public header, which every object depends on through common.h)."""
    out = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            out[f] = hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:16]
    return out


def current(kernel: str, recorded: Dict[str, str], now: Dict[str, str] = None) -> bool:
    """True while every file ``kernel`` is built from still has the digest ``recorded`` holds."""
    now = now or file_digests()
    return all(recorded.get(f) is not None and recorded.get(f) == now.get(f) for f in files_of(kernel))
