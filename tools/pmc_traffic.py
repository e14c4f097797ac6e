#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into per-kernel HBM bytes
per launch, following MI355X_MICROARCH.md §HBM: the counters are in KiB; on gfx950 FETCH_SIZE under-reports wide
(16 B/lane) coalesced reads by exactly 2x, other access widths must be calibrated on a kernel with a known byte count.
Calibration kernel here: nchw_to_nhwc_kernel<C> (reads 4 B/lane dword streams, writes 16 B/lane; algorithmic bytes =
4*numel each way) from the same run.

    python tools/pmc_traffic.py gpurun_out/final1/fetch gpurun_out/final1/write > profiles/pmc_traffic.json
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(k):
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", k)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", k)
    return m.group(1).replace(" ", "")


def tagname(k):
    """Map a kernel instantiation to the tag bench.py uses."""
    m = re.match(r"conv3d_kernel<(\d+),(\d+),(\d+),(\d+),", k)
    if m:
        np_, nt = int(m.group(2)), {16: 1, 48: 2, 80: 4}[int(m.group(2))]
        return "conv3d_kernel<%d,%s,%s>" % (nt, m.group(3), m.group(4))
    return k


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[tagname(short(row["Kernel_Name"]))].append(float(row["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_units": "bytes per launch (mean over launches); fetch_raw/write_raw = counter*1024",
       "_note": "fetch_corrected doubles FETCH_SIZE for kernels whose dominant reads are 16 B/lane (sweeps, transposes' "
                "writes are not reads); conv kernels stage with dword buffer loads - see calibration entry"}
# config-2 transpose: features [1,5,C,H,W] at each stage = 35.4 MB read + 35.4 MB written
cal = {}
for c in (8, 16, 32, 64):
    k = "nchw_to_nhwc_kernel<%d>" % c
    if k in fetch:
        cal[k] = {"algorithmic_read": 5 * 56623104 / 8 * 4 / 4 * 1.0 if False else 35389440.0, "fetch_raw": fetch[k], "write_raw": write.get(k)}
out["_calibration"] = cal
wide = ("cv_entropy_kernel", "cv_aggregate_kernel")
for k in sorted(set(fetch) | set(write)):
    if not re.search(r"cv_|vis_|conv3d|deconv|head|prob3|nchw|schedule|init_inv", k):
        continue
    fr, wr = fetch.get(k, 0.0), write.get(k, 0.0)
    fc = fr * 2.0 if k.startswith(wide) else fr
    out[k] = {"fetch_raw": fr, "write_raw": wr, "fetch_corrected": fc, "traffic": fc + wr}
print(json.dumps(out, indent=1))
