#!/usr/bin/env python
"""Static cost model of a gfx950 kernel's basic blocks from its assembly, weighted with the issue costs measured by tools/probe/ubench
(profiles/r03_ubench.txt): full-rate fp32 / simple integer ops ~2.3 cycles per wave-instruction per SIMD, "half-rate" ops (DPP, compares,
selects, conversions, min/max, integer multiply, shifts-with-add, any VOP3 with an SGPR source, packed fp32) ~4.5, transcendentals ~8.6.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o k.s file.hip
    python tools/isa_cost.py k.s cv_aggregate_kernelILi2ELb1ELb1 [--min 20]
"""
import re
import sys

FULL = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_ashrrev_i32", "v_lshrrev_b32",
        "v_lshlrev_b32", "v_sub_u32", "v_add_u32", "v_subrev_u32", "v_fmac_f32", "v_mac_f32", "v_not_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def cost(op, operands):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if not base.startswith("v_"):
        return 0.0
    if base in QUARTER:
        return 8.6
    if base.startswith("v_mfma"):
        return 32.0 if "16x16x4" in base or "32x32x2" in base else 16.0
    if op.endswith("_dpp") or "quad_perm" in operands or "row_" in operands:
        return 4.5
    if base.startswith("v_pk_"):
        return 5.0
    if base in FULL:
        # an SGPR / literal source on a 3-operand op measured half rate
        if base == "v_fma_f32" and re.search(r"\bs\d+|\bs\[|0x", operands):
            return 4.5
        return 2.3
    return 4.5


def main():
    path, key = sys.argv[1], sys.argv[2]
    minc = float(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0.0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ": " in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], {"name": "entry", "valu": 0, "cyc": 0.0, "vmem": 0, "lds": 0, "salu": 0, "other": 0, "loop": ""}
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB"):
                blocks.append(cur)
                cur = {"name": t.split(":")[0], "valu": 0, "cyc": 0.0, "vmem": 0, "lds": 0, "salu": 0, "other": 0, "loop": "loop" if "Loop" in t else ""}
            continue
        if t.startswith(".LBB") or re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = {"name": t.split(":")[0], "valu": 0, "cyc": 0.0, "vmem": 0, "lds": 0, "salu": 0, "other": 0, "loop": "loop" if "Loop" in t else ""}
            continue
        parts = t.split(None, 1)
        op, operands = parts[0], (parts[1] if len(parts) > 1 else "")
        operands = operands.split(";")[0]
        if op.startswith("v_"):
            cur["valu"] += 1
            cur["cyc"] += cost(op, operands)
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        else:
            cur["other"] += 1
    blocks.append(cur)
    tot = 0
    for b in blocks:
        tot += b["cyc"]
        if b["cyc"] >= minc:
            print("%-12s %-5s valu %4d  est cycles %7.1f  vmem %3d  lds %3d  salu %3d" % (b["name"], b["loop"], b["valu"], b["cyc"], b["vmem"], b["lds"], b["salu"]))
    print("static total est cycles %.0f" % tot)


if __name__ == "__main__":
    main()
