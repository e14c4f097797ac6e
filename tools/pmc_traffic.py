#!/usr/bin/env python
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) over tools/prof_traffic.py -> HBM bytes
per launch of every kernel, following MI355X_MICROARCH.md §HBM: the counters are in KiB and count fabric requests, not bytes
(FETCH_SIZE reads exactly 1/2 of a coalesced stream on gfx950), so both are CALIBRATED on a kernel of known size in the same run:
the config-2 stage-4 feature transpose (283.1 MB in, 283.1 MB out): bytes = counter * (known bytes / counter of that kernel).
Measured factors: fetch 2.000, write 1.000.

    python tools/pmc_traffic.py <fetch dir> <write dir> <out.json>
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

CAL_BYTES = 256 * 1024 * 1024 * 4.0


def short(k):
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", k)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", k)
    return m.group(1).replace(" ", "") if m else k


def tagname(k):
    """rocprof kernel name -> the tag bench.py uses for the same launch."""
    m = re.match(r"cv_entropy_kernel<(\d+),(true|false)>", k)
    if m:
        return "cv_entropy_kernel<%s>" % m.group(1)
    m = re.match(r"cv_aggregate_kernel<(\d+),(true|false),(true|false)>", k)
    if m:
        return "cv_aggregate_kernel<%s,%s>" % (m.group(1), m.group(2))
    m = re.match(r"cv_corr_kernel<(\d+),(true|false)>", k)
    if m:
        return "cv_corr_kernel<%s>" % m.group(1)
    m = re.match(r"nchw_to_nhwc_kernel<(\d+),", k)
    if m:
        return "nchw_to_nhwc_kernel<%s>" % m.group(1)
    m = re.match(r"conv3d_kernel<(\d+),(\d+),(\d+),(\d+),", k)
    if m:
        nt = {16: 1, 48: 2, 80: 4}.get(int(m.group(2)), 0)
        return "conv3d_kernel<%d,%s,%s>" % (nt, m.group(3), m.group(4))
    return k


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[tagname(short(row["Kernel_Name"]))].append(float(row["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    # calibration on a kernel of THIS library with a known byte count: the stage-4 feature transpose of config 2 reads and writes
    # exactly 4 * (5 views * 8 ch * 1152 * 1536) bytes, streaming, larger than the Infinity Cache
    calk = "nchw_to_nhwc_kernel<8>"
    cal_bytes = 4.0 * 5 * 8 * 1152 * 1536
    ffac = cal_bytes / fetch[calk] if fetch.get(calk) else 2.0
    wfac = cal_bytes / write[calk] if write.get(calk) else 1.0
    import glob as _g
    import hashlib
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(_g.glob(os.path.join(repo, "mvsformer_amd", "csrc", "*.hip")) + _g.glob(os.path.join(repo, "mvsformer_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    out = {"csrc_digest": h.hexdigest()[:16], "_units": "bytes per launch (mean over launches)", "_calibration": {"kernel": calk, "known_bytes_each_way": cal_bytes,
           "fetch_raw": fetch.get(calk), "write_raw": write.get(calk), "fetch_factor": ffac, "write_factor": wfac},
           "_note": "hbm_bytes_per_launch = FETCH_SIZE*1024*fetch_factor + WRITE_SIZE*1024*write_factor; factors from the float4 copy "
                    "of the same run (exact for 16 B/lane streams; 'narrow' kernels load dwords and are only indicative)", "kernels": {}}
    wide = ("cv_entropy_kernel", "cv_aggregate_kernel", "cv_corr_kernel", "cv_merge_kernel", "nchw_to_nhwc", "cv_tiled")
    for k in sorted(set(fetch) | set(write)):
        if not re.search(r"cv_|vis_|conv3d|deconv|head|prob3|nchw|schedule|init_inv|wino|x3_", k):
            continue
        fr, wr = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"fetch_raw": fr, "write_raw": wr, "hbm_bytes_per_launch": fr * ffac + wr * wfac, "narrow": not k.startswith(wide)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["_calibration"]))
    for k, v in out["kernels"].items():
        if k.startswith(("cv_", "nchw")):
            print("%-40s %.1f MB" % (k, v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
