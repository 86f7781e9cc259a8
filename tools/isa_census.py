#!/usr/bin/env python
"""Static instruction census of a kernel variant between the MKH_MARK comments of ik_kernel.h (compiled with
-DMKH_MARKERS): VALU / SALU / LDS / VMEM / other instruction counts per region, in program order.  Loop bodies
appear once (multiply by the trip count by hand: 18 phase-0 pivots and 13.5 active-set iterations for G1).

    python tools/isa_census.py 62_32_r44
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mink_amd", "csrc"))
import build as hipbuild  # noqa: E402


def classify(op: str) -> str:
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_store")):
        return "SMEM"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait/nop"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    return "other"


def valu_class(op: str, code: str) -> str:
    if op.startswith(("v_fma_f64", "v_fmac_f64")):
        return "fp64 FMA"
    if op.endswith("_f64") or "_f64_" in op:
        return "cmp" if op.startswith("v_cmp") else "other fp64"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "readlane / writelane"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")):
        ops = code.split(",")
        src = ops[1].strip() if len(ops) > 1 else ""
        if "dpp" in op or "row_" in code or "quad_perm" in code:
            return "mov (DPP)"
        if src.startswith("0x") or re.match(r"^-?[0-9.]+$", src):
            return "mov literal / 0"
        return "mov from SGPR" if src.startswith("s") else "mov"
    if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_sub", "v_lshl", "v_lshr", "v_ashr", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i",
                      "v_and", "v_or", "v_xor", "v_add3", "v_lshl_add", "v_add_lshl", "v_bfe", "v_bfi", "v_not", "v_min_", "v_max_",
                      "v_mul_u32", "v_mul_i32", "v_add_nc", "v_mbcnt", "v_alignbit", "v_perm")):
        return "integer / address"
    if op.startswith("v_cvt"):
        return "cvt"
    return "other"


def class_table(asm: str):
    """Static VALU instructions of every function of the code object by class (second table of the census)."""
    func, table, funcs = None, {}, []
    for line in asm.split("\n"):
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not line.startswith(".L"):
            func = m.group(1)
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":") or func is None:
            continue
        op = code.split()[0]
        if not op.startswith("v_"):
            continue
        if func not in table:
            table[func] = {}
            funcs.append(func)
        k = valu_class(op, code)
        table[func][k] = table[func].get(k, 0) + 1
    kinds = ["fp64 FMA", "other fp64", "integer / address", "readlane / writelane", "cndmask", "cmp", "mov literal / 0", "mov from SGPR",
             "mov", "mov (DPP)", "cvt", "other"]
    print()
    print("static VALU instructions by function x class")
    print("| function | total | " + " | ".join(kinds) + " |")
    print("|---|---|" + "---|" * len(kinds))
    tot = {}
    for f in funcs:
        short = re.sub(r"^_ZN3mkh\d+", "", f)
        short = re.split(r"EPK|EPKNS", short)[0]
        print("| %s | %d | " % (short, sum(table[f].values())) + " | ".join(str(table[f].get(k, 0)) for k in kinds) + " |")
        for k, v in table[f].items():
            tot[k] = tot.get(k, 0) + v
    print("| all | %d | " % sum(tot.values()) + " | ".join(str(tot.get(k, 0)) for k in kinds) + " |")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "62_32_r44"
    src = os.path.join(hipbuild.BUILD, f"variant_{name}.hip")
    asm = subprocess.run([hipbuild._hipcc()] + hipbuild.FLAGS + hipbuild.KERNEL_FLAGS +
                         ["-DMKH_MARKERS", "-S", "--cuda-device-only", "-o", "-", src], check=True,
                         capture_output=True, text=True).stdout
    region, counts, order = "prologue", {}, []
    for line in asm.split("\n"):
        m = re.search(r"MKH_MARK (\w+)", line)
        if m:
            region = "after " + m.group(1)
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":") or code.startswith(";"):
            continue
        op = code.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        if region not in counts:
            counts[region] = {}
            order.append(region)
        k = classify(op)
        counts[region][k] = counts[region].get(k, 0) + 1
    kinds = ["VALU", "SALU", "LDS", "VMEM", "SMEM", "wait/nop", "other"]
    print("%-28s" % "region" + "".join("%9s" % k for k in kinds))
    for r in order:
        print("%-28s" % r + "".join("%9d" % counts[r].get(k, 0) for k in kinds))
    class_table(asm)


if __name__ == "__main__":
    main()
