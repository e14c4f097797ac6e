#!/usr/bin/env python
"""Concurrency of a rocprofv3 kernel trace (rocpd sqlite): per window, sum of kernel durations vs the union of their intervals.

    python tools/rocpd_overlap.py results.db [tail_fraction]
"""
import sqlite3
import sys


def main(path, tail=0.5):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else "0")
    st = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else "0")
    print("columns:", cols)
    rows = sorted(cur.execute("select start, end, %s, %s from kernels" % (q, st)).fetchall())
    rows = rows[int(len(rows) * (1.0 - tail)):]                 # the steady-state tail of the run
    tot = sum(e - s for s, e, _, _ in rows)
    union, cur_s, cur_e = 0, None, None
    for s, e, _, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    span = rows[-1][1] - rows[0][0]
    print("kernels %d  sum of durations %.3f ms  union %.3f ms  span %.3f ms  overlap factor %.3f  queues %s streams %s" % (
        len(rows), tot / 1e6, union / 1e6, span / 1e6, tot / union, sorted({q for _, _, q, _ in rows}), sorted({t for _, _, _, t in rows})))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
