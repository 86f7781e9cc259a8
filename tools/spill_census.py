#!/usr/bin/env python
"""Where does hipcc spill?  Scratch stores / loads and SGPR-spill lane writes of a kernel variant per region between the
MKH_MARK comments of ik_kernel.h (compiled with -DMKH_MARKERS), in program order, callees first.  Used to find the
register diet of the 3-waves-per-SIMD variants (a compiler cap of 74 registers): with `--cap N` the variant is compiled
with that cap instead of its own (-DMKH_CAP_PROBE: the code is wrong, the pressure figures are what one wants).

    python tools/spill_census.py 44_32_r44_w3 [--cap 80]
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mink_amd", "csrc"))
import build as hipbuild  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "44_32_r44_w3"
    extra = []
    if "--cap" in sys.argv:
        extra = ["-DMKH_CAP_PROBE=%d" % int(sys.argv[sys.argv.index("--cap") + 1])]
    src = os.path.join(hipbuild.BUILD, f"variant_{name}.hip")
    asm = subprocess.run([hipbuild._hipcc()] + hipbuild.FLAGS + hipbuild.KERNEL_FLAGS + extra +
                         ["-DMKH_MARKERS", "-S", "--cuda-device-only", "-o", "-", src], check=True,
                         capture_output=True, text=True).stdout
    region, counts, order = "prologue", {}, []
    for line in asm.split("\n"):
        m = re.match(r"\s*\.type\s+(\S+),@function", line)
        if m:
            region = "[" + ("kernel" if "ik_solve_kernel" in m.group(1) else re.sub(r"^_ZN3mkh\d+", "", m.group(1))[:12]) + "] prologue"
        m = re.search(r"MKH_MARK (\w+)", line)
        if m:
            region = region.split("]")[0] + "] after " + m.group(1) if "]" in region else "after " + m.group(1)
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        op = code.split()[0]
        if region not in counts:
            counts[region] = {"valu": 0, "scratch_st": 0, "scratch_ld": 0, "sgpr_spill_wr": 0}
            order.append(region)
        c = counts[region]
        c["valu"] += op.startswith("v_")
        c["scratch_st"] += op.startswith("scratch_store")
        c["scratch_ld"] += op.startswith("scratch_load")
        c["sgpr_spill_wr"] += op.startswith("v_writelane")
    print("%-44s %8s %8s %8s %8s" % ("region", "VALU", "scr.st", "scr.ld", "writelane"))
    for r in order:
        c = counts[r]
        print("%-44s %8d %8d %8d %8d" % (r, c["valu"], c["scratch_st"], c["scratch_ld"], c["sgpr_spill_wr"]))
    for k in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        print(k, re.findall(r"\." + k + r":\s+(\d+)", asm))
    print("callee VGPRs:", re.findall(r"\.set \.L\S*?(pre_phases|wood_start)\S*\.num_vgpr, (\d+)", asm))


if __name__ == "__main__":
    main()
