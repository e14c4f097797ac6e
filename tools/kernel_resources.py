#!/usr/bin/env python
"""Per-kernel VGPR / SGPR / scratch / LDS figures from a hipcc -save-temps .s file (gfx950 metadata block)."""
import re
import subprocess
import sys


def main(path, pattern=""):
    s = open(path).read()
    md = s[s.index("amdhsa.kernels"):]
    for e in md.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", e).group(1)
        get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, e).group(1)
        try:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
        if pattern and not re.search(pattern, dem):
            continue
        print("%-100s vgpr %3s agpr %3s sgpr %3s scratch %4s lds %6s" % (dem[:100], get("vgpr_count"), re.search(r"^:?\s*(\d+)", e).group(1),
                                                                         get("sgpr_count"), get("private_segment_fixed_size"),
                                                                         get("group_segment_fixed_size")))


if __name__ == "__main__":
    main(*sys.argv[1:])
