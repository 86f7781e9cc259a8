#!/usr/bin/env python
"""Loops of a kernel variant's ISA whose body waits for memory several times per iteration — `s_waitcnt lgkmcnt(0)` / `vmcnt(0)`
behind single reads: a dependent LDS or L2 round trip each.  This is how round 4 found the 4.5 % of the headline that sat in the
S = I + Jh·Jhᵀ accumulation (eight clamped row pointers: eight reads, each behind its own full wait, per chain bit).

    python tools/isa_wait_scan.py 44_32_r44_w3 [48_72 ...]          (MKH_SCAN_MIN=<n>: report loops with at least n such waits, default 3)
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mink_amd", "csrc"))
import build as hipbuild  # noqa: E402


def scan(asm):
    lines = asm.split("\n")
    func, labels, out = None, {}, []
    for i, l in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w.$]*):", l)
        if m and not l.startswith(".L"):
            func, labels = re.sub(r"^_ZN3mkh\d+", "", m.group(1))[:28], {}
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels:
            body = [x.split(";")[0].strip() for x in lines[labels[m.group(1)]:i + 1]]
            body = [b for b in body if b and not b.endswith(":") and not b.startswith(".")]
            singles, pend_l, pend_v = 0, 0, 0
            for b in body:
                if b.startswith(("ds_read", "ds_bpermute", "ds_swizzle")):
                    pend_l += 1
                elif b.startswith(("global_load", "flat_load")):
                    pend_v += 1
                elif b.startswith("s_waitcnt"):
                    if "lgkmcnt(0)" in b:
                        singles += 1 if 0 < pend_l <= 2 else 0
                        pend_l = 0
                    if "vmcnt(0)" in b:
                        singles += 1 if 0 < pend_v <= 2 else 0
                        pend_v = 0
            valu = sum(b.startswith("v_") for b in body)
            if singles >= int(os.environ.get("MKH_SCAN_MIN", "3")) and len(body) < 500:
                out.append((func, m.group(1), len(body), valu, singles))
    return out


def main():
    for name in sys.argv[1:] or ["44_32_r44_w3"]:
        src = os.path.join(hipbuild.BUILD, f"variant_{name}.hip")
        asm = subprocess.run([hipbuild._hipcc()] + hipbuild.FLAGS + hipbuild.KERNEL_FLAGS + ["-S", "--cuda-device-only", "-o", "-", src],
                             check=True, capture_output=True, text=True).stdout
        print(f"== {name}: loops with >= 3 short full waits per iteration (function, label, instructions, VALU, waits)")
        for r in scan(asm):
            print("   %-28s %-12s %4d %4d %3d" % r)


if __name__ == "__main__":
    main()
