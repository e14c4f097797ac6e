#!/usr/bin/env python
"""Summarize a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (CSV on stdout).

    python tools/rocpd_stats.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.csv
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:110]


def main(path, tail=0):
    """``tail`` > 0: only the last ``tail`` kernel dispatches (e.g. the replays of a captured step at the end of a run)."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    if tail > 0:
        rows = rows[-tail:]
    agg = {}
    for n, s, e in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.0f,%d,%d,%.2f' % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
