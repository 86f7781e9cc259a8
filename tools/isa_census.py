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


if __name__ == "__main__":
    main()
