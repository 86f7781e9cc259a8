#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small files kept under profiles/.

  python tools/rocprof_summary.py pmc  <out.json> <dir> [<dir> ...]   # --pmc passes (one dir each)
  python tools/rocprof_summary.py stats <out.csv> <dir>               # --kernel-trace --stats pass

The pmc summary lists, per counter, the value of every dispatch of the IK kernel (summed over the
counter's instances/XCDs) and of the calibration copy kernel (tools/pmc_workload.py), and derives
HBM bytes per IK launch from FETCH_SIZE / WRITE_SIZE calibrated on that copy.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short_kernel_name(profiler_name: str) -> str:
    """rocprofv3's demangled kernel name → the key of mink_amd/kernel_resources.json (`ik_solve_kernel_44_32_r44_w3`,
    `ik_quad_kernel<8,0>`, `ik_lane_kernel<6,1>`)."""
    m = re.search(r"(ik_[a-z0-9_]+)(<[^>]*>)?", profiler_name)
    if not m:
        return profiler_name
    name, targs = m.group(1), m.group(2)
    if targs:
        vals = [{"true": "1", "false": "0"}.get(t.strip(), t.strip()) for t in targs[1:-1].split(",")]
        name += "<" + ",".join(vals) + ">"
    return name


def provenance(kernels: list) -> dict:
    """What was profiled: sha256 of the libminkhip.so next to the package (the one the workload loads), the git HEAD it was
    linked at (mink_amd/build_info.json, written by build.py — the GPU box has no .git) and, per kernel of the timed solve,
    the sha256 of the kernel's code object (kernel_resources.json).  bench.py compares these with what IT loaded and ran."""
    out = {"kernels": [short_kernel_name(k) for k in kernels]}
    lib = os.path.join(REPO, "mink_amd", "libminkhip.so")
    try:
        h = hashlib.sha256()
        with open(lib, "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 20), b""):
                h.update(blk)
        out["library_sha256"] = h.hexdigest()
    except OSError:
        out["library_sha256"] = None
    try:
        with open(os.path.join(REPO, "mink_amd", "build_info.json")) as fh:
            info = json.load(fh)
        out["git_head"], out["git_dirty_sources"] = info.get("git_head"), info.get("git_dirty_sources")
        out["library_sha256_at_build"] = info.get("library_sha256")
    except (OSError, ValueError):
        out["git_head"] = None
    try:
        with open(os.path.join(REPO, "mink_amd", "kernel_resources.json")) as fh:
            table = json.load(fh)
        out["kernel_code_sha256"] = {k: (table.get(k) or {}).get("code_sha256") for k in out["kernels"]}
    except (OSError, ValueError):
        out["kernel_code_sha256"] = {}
    return out

COPY_BYTES = 64 * 1024 * 1024 * 8          # tools/pmc_workload.py: 512 MiB read and 512 MiB written


def read_counters(d):
    out = defaultdict(lambda: defaultdict(dict))  # kernel -> counter -> dispatch id -> value
    res = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                c = row["Counter_Name"]
                did = int(row["Dispatch_Id"])
                out[k][c][did] = out[k][c].get(did, 0.0) + float(row["Counter_Value"])
                if k not in res:
                    res[k] = {f: row.get(f) for f in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size",
                                                      "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
    return out, res


def pmc(out_path, dirs):
    """The timed solves are the dispatches of the IK kernel that was launched most often in a pass; the one or two
    launches of a tap variant before them (FK-only target generation, CoM targets: workloads.bench_batch) are listed
    under "setup_dispatches" and do not enter any per-launch figure."""
    ik, setup, copy, resources = defaultdict(list), defaultdict(list), defaultdict(list), {}
    companions = defaultdict(lambda: defaultdict(list))      # other launches of one solve call (the redo launch of tight rows)
    solve_kernel = None
    for d in dirs:
        counters, res = read_counters(d)
        iks = {k: cs for k, cs in counters.items() if "ik_solve_kernel" in k or "ik_lane_kernel" in k or "ik_quad_kernel" in k or "ik_wide_kernel" in k}
        if iks:
            # (a tight-rows solve is two launches — the 48-row build and the full-row redo behind it, equally often: the one
            #  that does the work has the larger counters)
            #  The workgroup-per-problem redo launch behind a wavefront kernel is never the main one: its 256-thread workgroups
            #  win on some counters (SQ_WAVES) and lose on others, and the choice must be the same in every pass.
            count = lambda k: max(len(per) for per in iks[k].values())
            main = max(iks, key=lambda k: (count(k), "ik_wide_kernel" not in k, sum(sum(per.values()) for per in iks[k].values())))
            solve_kernel = main
            for k in iks:
                if k != main and count(k) == count(main):
                    for c, per in iks[k].items():
                        companions[k][c] += [per[i] for i in sorted(per)]
        for k, cs in counters.items():
            if k in iks:
                resources[k] = res[k]
                tgt = ik if k == main else setup
            elif "copy" in k.lower() or "elementwise" in k.lower():
                tgt = copy
            else:
                continue
            for c, per in cs.items():
                vals = [per[i] for i in sorted(per)]
                # the copy kernel list may include small torch copies (H2D staging): keep the largest
                if tgt is copy:
                    tgt[c] = [max(vals + tgt.get(c, []))]
                else:
                    tgt[c] = tgt.get(c, []) + vals
    solve_kernels = [solve_kernel] + sorted(companions) if solve_kernel else []
    summary = {"solve_kernel": solve_kernel,
               "solve_kernels": solve_kernels,            # every launch of ONE solve call (main first)
               "provenance": provenance(solve_kernels),
               "ik_solve_kernel": {c: {"per_dispatch": v} for c, v in sorted(ik.items())},
               "companion_launches": {k: {c: {"per_dispatch": v} for c, v in sorted(cs.items())} for k, cs in companions.items()},
               "setup_dispatches": {c: {"per_dispatch": v} for c, v in sorted(setup.items())},
               "calibration_copy_512MiB": {c: v[0] for c, v in sorted(copy.items())},
               "kernel_resources": resources}
    # HBM bytes per IK launch.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; the calibration
    # factor is (known bytes) / (reported bytes) of the copy kernel in the same pass.
    hbm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c in ik and c in copy and copy[c][0] > 0:
            cal = COPY_BYTES / (copy[c][0] * 1024.0)
            raw = sum(ik[c]) / len(ik[c]) * 1024.0
            hbm[c] = {"raw_bytes_per_launch": raw, "calibration_factor": cal, "bytes_per_launch": raw * cal}
    if len(hbm) == 2:
        main_bytes = hbm["FETCH_SIZE"]["bytes_per_launch"] + hbm["WRITE_SIZE"]["bytes_per_launch"]
        hbm["main_kernel_bytes_per_launch"] = main_bytes
        # the other launches of the same solve call (same calibration): one solve = all of them
        extra = 0.0
        for k, cs in companions.items():
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                if c in cs and cs[c]:
                    extra += sum(cs[c]) / len(cs[c]) * 1024.0 * hbm[c]["calibration_factor"]
        hbm["companion_bytes_per_launch"] = extra
        hbm["traffic_bytes_per_launch"] = main_bytes + extra
    summary["hbm"] = hbm
    # Shares that say what the kernel waits for / runs out of (sums over the solve dispatches; a ratio is only formed from
    # counters of ONE pass, or against SQ_WAVE_CYCLES of the same workload, which every SQ pass of tools/profile.sh carries).
    def tot(c):      # mean per dispatch (a counter collected in two passes lists twice as many dispatches)
        return float(sum(ik[c])) / len(ik[c]) if c in ik and ik[c] else None

    def ratio(a, b):
        x, y = tot(a), tot(b)
        return (x / y) if (x is not None and y) else None

    summary["derived"] = {
        "valu_active_per_wave_cycle": ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
        "any_inst_active_per_wave_cycle": ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
        "wait_any_per_wave_cycle": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
        "wait_inst_any_per_wave_cycle": ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
        "lds_inst_active_per_wave_cycle": ratio("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"),
        "wait_inst_lds_per_wave_cycle": ratio("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"),
        "salu_cycles_per_wave_cycle": ratio("SQ_INST_CYCLES_SALU", "SQ_WAVE_CYCLES"),
        "lds_bank_conflict_per_lds_inst_active": ratio("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"),
        # the CU's one LDS pipe: cycles its index unit is busy (bank-conflict replays included) per busy CU cycle
        "lds_pipe_busy_per_cu_cycle": ratio("SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CU_CYCLES"),
        "lds_bank_conflict_per_cu_cycle": ratio("SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CU_CYCLES"),
        "lds_addr_conflict_per_lds_idx_active": ratio("SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
        "wave_cycles_per_dispatch": tot("SQ_WAVE_CYCLES"), "valu_insts_per_dispatch": tot("SQ_INSTS_VALU"),
        "salu_insts_per_dispatch": tot("SQ_INSTS_SALU"), "lds_insts_per_dispatch": tot("SQ_INSTS_LDS"),
    }
    # VALU instructions by class, per dispatch (the instruction-class counters of gfx950; "other" = moves, selects, lane
    # reads / writes, compares, DPP moves — everything the class counters do not name)
    cls = {k: tot("SQ_INSTS_VALU_" + k) for k in ("FMA_F64", "MUL_F64", "ADD_F64", "TRANS_F64", "INT32", "INT64", "CVT")}
    if tot("SQ_INSTS_VALU") and all(v is not None for v in cls.values()):
        cls["other"] = tot("SQ_INSTS_VALU") - sum(cls.values())
        summary["derived"]["valu_insts_by_class_per_dispatch"] = cls
    for k in ("SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE", "SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_MFMA"):
        if tot(k) is not None:
            summary["derived"][k.lower()[3:] + "_per_dispatch"] = tot(k)
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(hbm, indent=1))


def stats(out_path, d):
    paths = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not paths:
        raise SystemExit(f"no kernel_stats.csv under {d}")
    with open(paths[0]) as fh, open(out_path, "w") as out:
        out.write(fh.read())
    print(open(out_path).read())


if __name__ == "__main__":
    if sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3:])
    else:
        stats(sys.argv[2], sys.argv[3])
