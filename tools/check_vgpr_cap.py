#!/usr/bin/env python
"""Verify on the generated ISA that hipcc stays below the VGPR cap of every kernel variant, i.e. that no
compiler-generated instruction (anything outside the inline-asm blocks) touches the staging / tableau
registers v[256-2·NT-S, 256), S = 32 (16 for NT ≥ 56).  `amdgpu_num_vgpr` is easy to get wrong on gfx90a+ (see ik_kernel.h), and a
violation is silent: the kernel still runs and is only wrong when register pressure happens to be high.

    python tools/check_vgpr_cap.py            # all variants in mink_amd/csrc/_build
"""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "mink_amd", "csrc", "_build")


def max_compiler_vgpr(asm_text: str) -> int:
    """Highest VGPR a compiler-generated instruction of the KERNEL touches.  The 3-waves-per-SIMD variants call real
    functions (ik_kernel.h pre_phases, ...) that run while the tableau is dead and may use the whole register file: their
    bodies (between `.type <name>,@function` and `.Lfunc_end`) are skipped unless the symbol is the kernel's."""
    mx, skip, in_callee = -1, False, False
    for line in asm_text.split("\n"):
        m = re.match(r"\s*\.type\s+(\S+),@function", line)
        if m:
            in_callee = "ik_solve_kernel" not in m.group(1)
        if in_callee:
            continue
        if "#ASMSTART" in line:
            skip = True
        if not skip and not line.lstrip().startswith((";", ".")):
            code = line.split(";")[0]
            for m in re.finditer(r"\bv(\d+)\b", code):
                mx = max(mx, int(m.group(1)))
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
                mx = max(mx, int(m.group(2)))
        if "#ASMEND" in line:
            skip = False
    return mx


def check(src: str):
    sys.path.insert(0, os.path.join(REPO, "mink_amd", "csrc"))
    import build as hipbuild
    import gen_tab_asm as gen
    nt = int(re.search(r"variant_(\d+)_", os.path.basename(src)).group(1))
    # (_w3: the high-occupancy build — TabW3, 168 registers, for NT > 24; TabW4, 128 registers, for NT ≤ 24)
    top = (128 if nt <= 24 else 168) if os.path.basename(src).endswith("_w3.hip") else gen.total_for(nt)
    w3 = os.path.basename(src).endswith("_w3.hip")
    cap = top - 2 * nt - (2 * ((nt + 15) // 16) if w3 else gen.ntmp_for(nt))
    out = subprocess.run([hipbuild._hipcc()] + hipbuild.FLAGS + hipbuild.KERNEL_FLAGS +
                         ["-S", "--cuda-device-only", "-o", "-", src], check=True, capture_output=True, text=True).stdout
    mx = max_compiler_vgpr(out)
    spills = re.findall(r"\.(sgpr|vgpr)_spill_count:\s+(\d+)", out)
    scratch = [int(x) for x in re.findall(r"ScratchSize: (\d+)", out)]     # (one per function: callees first)
    return os.path.basename(src), cap, mx, dict(spills), max(scratch) if scratch else -1


def main():
    srcs = sorted(glob.glob(os.path.join(BUILD, "variant_*.hip")))
    if len(sys.argv) > 1:                       # optional name filters, e.g. 44_0 64_32_r44
        srcs = [s for s in srcs if any(os.path.basename(s) == f"variant_{a}.hip" for a in sys.argv[1:])]
    if not srcs:
        raise SystemExit("run mink_amd/csrc/build.py first")
    bad = 0
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for name, cap, mx, spills, scratch in ex.map(check, srcs):
            ok = mx < cap
            bad += not ok
            print(f"{name:28s} cap v{cap:<4d} highest compiler VGPR v{mx:<4d} {'ok ' if ok else 'VIOLATION'} "
                  f"scratch {scratch:4d} B  spills sgpr {spills.get('sgpr', '?'):>4} vgpr {spills.get('vgpr', '?'):>4}")
    if bad:
        raise SystemExit(f"{bad} variant(s) exceed their VGPR cap")


if __name__ == "__main__":
    main()
