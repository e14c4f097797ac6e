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


def family(tag):
    """tag or kernel name -> the part both spell the same way ('x3_conv', 'cv_entropy', 'vis_x3', 'schedule_inverse', ...)."""
    t = re.sub(r"_kernel.*|<.*", "", tag)
    t = re.sub(r"^mvs_", "", t)
    t = re.sub(r"_fwd$", "", t)
    return {"x3_tail": "tail_x3"}.get(t, t)


def same_family(a, b):
    fa, fb = family(a), family(b)
    return fa.startswith(fb) or fb.startswith(fa)


def collect(d, counter, order=None):
    """Mean counter value per tag.  With `order` (the tags of ONE cascade in launch order, tools/prof_traffic.py) the dispatches of the process's
    LAST cascade are aligned with it from the end, so launches that share a kernel instance (two layers, two stages) land on their own tag;
    a tag with no matching dispatch within the next three is skipped, and kernels are always keyed by (mapped) name as well."""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                rows.append((int(row.get("Dispatch_Id", len(rows))), short(row["Kernel_Name"]), float(row["Counter_Value"]) * 1024.0))
    rows.sort()
    acc = defaultdict(list)
    for _, k, v in rows:
        acc[tagname(k)].append(v)
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    if order:
        lib = [(k, v) for _, k, v in rows if not k.startswith(("at::", "__amd"))]
        j, got, missed = len(lib) - 1, {}, 0
        for tag in reversed(order):
            hit = next((q for q in range(4) if j - q >= 0 and same_family(lib[j - q][0], tag)), None)
            if hit is None:
                missed += 1
                continue
            got.setdefault(tag, []).append(lib[j - hit][1])
            j -= hit + 1
        print("launch-order alignment: %d tags, %d without a dispatch" % (len(order), missed), file=sys.stderr)
        if missed <= len(order) // 10:
            out.update({k: sum(v) / len(v) for k, v in got.items()})
    return out


def main():
    import os
    order_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "launch_order.json")
    order = json.load(open(order_file)) if os.path.exists(order_file) else None
    fetch, write = collect(sys.argv[1], "FETCH_SIZE", order), collect(sys.argv[2], "WRITE_SIZE", order)
    # calibration on a kernel of THIS library with a known byte count: the stage-4 feature transpose of config 2 reads and writes
    # exactly 4 * (5 views * 8 ch * 1152 * 1536) bytes, streaming, larger than the Infinity Cache
    calk = "nchw_to_nhwc_kernel<8>"
    cal_bytes = 4.0 * 5 * 8 * 1152 * 1536
    ffac = cal_bytes / fetch[calk] if fetch.get(calk) else 2.0
    wfac = cal_bytes / write[calk] if write.get(calk) else 1.0
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    from mvsformer_amd import _sources
    try:
        head = subprocess.check_output(["git", "-C", repo, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:                                       # noqa: BLE001 - the GPU box has no .git
        head = os.environ.get("MVS_HEAD", "unknown (no .git on the GPU box)")
    # per-FILE digests: bench.py reports a kernel's traffic while the files that kernel is built from are unchanged (mvsformer_amd/_sources.py)
    out = {"source_digests": _sources.file_digests(), "collected_at_head": head, "_units": "bytes per launch (mean over launches)", "_calibration": {"kernel": calk, "known_bytes_each_way": cal_bytes,
           "fetch_raw": fetch.get(calk), "write_raw": write.get(calk), "fetch_factor": ffac, "write_factor": wfac},
           "_note": "hbm_bytes_per_launch = FETCH_SIZE*1024*fetch_factor + WRITE_SIZE*1024*write_factor; factors from the float4 copy "
                    "of the same run (exact for 16 B/lane streams; 'narrow' kernels load dwords and are only indicative)", "kernels": {}}
    wide = ("cv_entropy_kernel", "cv_aggregate_kernel", "cv_corr_kernel", "cv_merge_kernel", "nchw_to_nhwc", "cv_tiled")
    for k in sorted(set(fetch) | set(write)):
        if not re.search(r"cv_|vis_|conv3d|deconv|head|prob3|nchw|schedule|init_inv|wino|x3_|tail", k):
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
