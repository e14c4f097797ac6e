#!/bin/bash
# kernel trace of images -> depth map (tools/prof_e2e.py): two runs of 3 and 13 forwards, their difference / 10 = one forward (set-up kernels cancel),
# every launch incl. the ATen glue -> gpurun_out/${ROUND}_e2e_kernels_per_forward.txt
REPO=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in 1 11; do
  rm -rf $REPO/gpurun_out/prof_e2e_trace_$n
  rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_e2e_trace_$n -o v -- python $REPO/tools/prof_e2e.py --iters $n > $REPO/gpurun_out/prof_e2e_under_rocprof_$n.txt 2>&1
  DB=$(find $REPO/gpurun_out/prof_e2e_trace_$n -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py $DB > $REPO/gpurun_out/e2e_stats_$n.csv 2>> $REPO/gpurun_out/rocpd_stats_e2e.err
  rm -rf $REPO/gpurun_out/prof_e2e_trace_$n
done
cd $REPO
python - <<'PY' > gpurun_out/${ROUND:-r06}_e2e_kernels_per_forward.txt
import csv
def load(n):
    return {r["kernel"]: (int(r["calls"]), float(r["total_ns"])) for r in csv.DictReader(open("gpurun_out/e2e_stats_%d.csv" % n))}
a, b = load(1), load(11)
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0.0))
    if c - c0 > 0:
        rows.append((k, (c - c0) / 10.0, (t - t0) / 10.0 / 1e6))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print("images -> depth map, kernels of ONE forward (difference of a 13- and a 3-forward trace / 10): %.3f ms in %.0f launches" % (tot, sum(r[1] for r in rows)))
for k, c, t in rows:
    print("%-100s x%-6.1f %8.4f ms %5.1f %%" % (k[:100], c, t, 100 * t / tot))
PY
