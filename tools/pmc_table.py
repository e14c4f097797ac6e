#!/usr/bin/env python
"""Pivot rocprofv3 counter_collection CSVs into kernel x counter means.  usage: pmc_table.py dir [dir...]"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    import glob
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::|^void ", "", row["Kernel_Name"])
            k = re.match(r"([\w:]+(?:<[^(]*>)?)", k).group(1)[:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if not re.search(r"cv_|vis_|conv3d|deconv|head|prob3|wino|fpn|x3_|bf16_|gemm", k):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
