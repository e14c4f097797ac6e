#!/usr/bin/env python
"""Where a kernel spills: scratch stores/loads per basic block of a -save-temps .s file.  usage: spill_sites.py file.s mangled-substring"""
import re
import sys

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
i = s.index("\n", i)
j = s.index(".Lfunc_end", i)
lab, st, ld, n, ops = "entry", 0, 0, 0, {}
for l in s[i:j].split("\n"):
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        if st or ld:
            print("%-14s insts %4d scratch_store %3d scratch_load %3d  %s" % (lab, n, st, ld, " ".join("%s=%d" % kv for kv in sorted(ops.items()))))
        lab, st, ld, n, ops = m.group(1), 0, 0, 0, {}
        continue
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    n += 1
    op = m.group(1)
    if op.startswith("scratch_store"):
        st += 1
    elif op.startswith("scratch_load"):
        ld += 1
    elif op.startswith(("buffer_load", "ds_read", "ds_write", "global_load", "s_barrier")):
        ops[op] = ops.get(op, 0) + 1
if st or ld:
    print("%-14s insts %4d scratch_store %3d scratch_load %3d  %s" % (lab, n, st, ld, " ".join("%s=%d" % kv for kv in sorted(ops.items()))))
