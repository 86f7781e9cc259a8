#!/usr/bin/env python
"""Combines the per-stop PMC summaries of tools/phase_census.sh into the phase x class table (JSON + markdown).

    python tools/phase_census.py <dir with stop<k>.json> <output prefix> [config]
"""
import json
import os
import sys

PHASES = ["FK (load inputs, local transforms, pointer jumping)", "joint axes / dof lanes", "task lanes (pose error, log, jlog)",
          "posture / damping", "box limits", "Jacobian rows", "S = I + Jh Jh^T and w", "eliminations (LDL^T, refinement)",
          "rank-1 updates of the dof block", "active set + write-back"]
CLASSES = [("VALU", "SQ_INSTS_VALU"), ("fp64 FMA", "SQ_INSTS_VALU_FMA_F64"), ("fp64 MUL", "SQ_INSTS_VALU_MUL_F64"),
           ("fp64 ADD", "SQ_INSTS_VALU_ADD_F64"), ("fp64 TRANS", "SQ_INSTS_VALU_TRANS_F64"), ("INT32", "SQ_INSTS_VALU_INT32"),
           ("INT64", "SQ_INSTS_VALU_INT64"), ("CVT", "SQ_INSTS_VALU_CVT"), ("SALU", "SQ_INSTS_SALU"), ("SMEM", "SQ_INSTS_SMEM"),
           ("LDS load", "SQ_INSTS_LDS_LOAD"), ("LDS store", "SQ_INSTS_LDS_STORE"), ("VMEM read", "SQ_INSTS_VMEM_RD"),
           ("branch", "SQ_INSTS_BRANCH"), ("wave cycles", "SQ_WAVE_CYCLES")]


def main():
    d, out = sys.argv[1], sys.argv[2]
    per = {}
    meta = {}
    for stop in list(range(1, 10)) + [0]:
        s = json.load(open(os.path.join(d, "stop%d.json" % stop)))
        ik = s["ik_solve_kernel"]
        meta[stop] = s.get("solve_kernel")
        per[stop] = {c: (sum(ik[c]["per_dispatch"]) / len(ik[c]["per_dispatch"]) if c in ik else None) for _, c in CLASSES}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mink_amd import workloads
    B = workloads.BENCH_CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "g1_c3"]["batch"]
    order = list(range(1, 10)) + [0]
    rows = []
    prev = {c: 0.0 for _, c in CLASSES}
    for k, stop in enumerate(order):
        row = {"phase": PHASES[k], "stop": stop}
        for name, c in CLASSES:
            v = per[stop][c]
            row[name] = None if v is None else (v - prev[c]) / B
            if v is not None:
                prev[c] = v
        fp = sum(row[n] or 0.0 for n in ("fp64 FMA", "fp64 MUL", "fp64 ADD", "fp64 TRANS"))
        integer = sum(row[n] or 0.0 for n in ("INT32", "INT64", "CVT"))
        row["other VALU (moves, selects, compares, lane reads)"] = row["VALU"] - fp - integer
        rows.append(row)
    total = {n: sum(r[n] or 0.0 for r in rows) for n in list(rows[0].keys()) if n not in ("phase", "stop")}
    json.dump({"kernel": meta[0], "batch": B, "per_solve_by_phase": rows, "per_solve_total": total,
               "note": "clock build (-DMKH_CLOCKS): 24 cycle stamps + one stop test per boundary ride along"}, open(out + ".json", "w"), indent=1)
    cols = ["VALU", "fp64 FMA", "fp64 MUL", "fp64 ADD", "fp64 TRANS", "INT32", "INT64", "other VALU (moves, selects, compares, lane reads)",
            "SALU", "SMEM", "LDS load", "LDS store", "VMEM read", "branch"]     # (wave cycles stay in the JSON: not additive across stops)
    with open(out + ".md", "w") as f:
        f.write("# Instructions per solve by phase x class — `%s`, B = %d\n\n" % (meta[0], B))
        f.write("PMC counters of launches of the clock build that abandon every solve after phase boundary k (tools/phase_census.sh); "
                "row k = stop k minus stop k-1, per solve (wave instructions, i.e. one count per 64-lane instruction).\n\n")
        f.write("| phase | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols) + "\n")
        for r in rows + [dict(total, phase="**whole solve**")]:
            f.write("| %s | " % r["phase"] + " | ".join("%.0f" % r[c] if r.get(c) is not None else "–" for c in cols) + " |\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
