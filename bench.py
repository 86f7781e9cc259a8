#!/usr/bin/env python
"""Headline benchmark: batched differential-IK solves/sec on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config g1_c3|ur5e_c2|g1_full|shadow_c4] [--batch B]

Default workload (BASELINE.json metric, configs[2]): Unitree G1 (nq=44, nv=43), 4 FrameTasks
(feet + palms) + PostureTask + ConfigurationLimit + VelocityLimit box limits, batch
65 536 per GPU, float64.  --config selects the other BASELINE configs (parity-test cases; their
lines carry the same fields).  A "step" is one batched solve_ik over the resident batch
(inputs already in HBM).

N > 1: one rank per GPU.  The driver launches the ranks itself through torch.distributed.run; when
WORLD_SIZE is not set, `python bench.py --gpus N` re-executes itself under torch.distributed.run with N
ranks (and refuses to run if the box has fewer than N GPUs) — a line is never printed with n_gpus != --gpus.
The batch shards by rank (weak scaling, configs[4] = 8 x 65 536) with no data-path collective: the
instances are independent, v stays on the GPU that produced it.  After the compute-only timed region a
second timed region of the same K steps ends every step with the RCCL gather of v to rank 0 that BASELINE
configs[4] mentions; it is reported separately under "gather" (never as `value`).

Prints ONE JSON line (rank 0) with the whole-job solves/sec, the HBM roofline of the kernel (algorithmic
bytes / launch duration from HIP events on the launch stream) and CPU baselines.
"""

import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_SIMDS = 1024                       # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                                   # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6                          # MI355X fp64 vector peak (same guide, chip table)
# algorithmic fp64 flops per G1 config-3 solve, 0.10-0.15 Mflop (FK, Jacobians, log/jlog,
# H accumulation, factorisation, active-set steps of the reference algorithm) — the midpoint
ALGORITHMIC_FLOP_PER_SOLVE = {"g1_c3": 0.125e6}


def issued_flop_per_solve(kernel: str) -> int:
    """Issued fp64 FMA flops of the tableau work per G1 solve (64 lanes x 2 flop per row of a rank-1 update),
    DESIGN.md §3.1.  Low-rank start (kernel name ends in _r44): the 18 residual pivots touch 6 x 34 + 6 x 42 +
    3 x 50 + 3 x 62 = 792 rows (row-sparse updates: dof prefix 16 / 24 / 32 / 44 + 18 residual rows), 13.4
    active-set pivots x 44 rows, and the 19 x 18 chain-sparse dot products of Jh·Jhᵀ (≈14 dofs each); direct
    start: 43 + 13.4 pivots x NT rows + 18 rank-1 updates of the H accumulation."""
    nt = int(kernel.split("_")[3])
    if "_r" in kernel:
        return int((792 + 13.4 * 44) * 64 * 2 + 19 * 18 * 14 * 2)
    return int((43 + 13.4 + 18) * nt * 64 * 2)


def usable_cpus() -> int:
    """Host threads this process may really use: affinity mask ∩ cgroup CPU quota (the GPU box shows 256
    hardware threads but grants a 16-CPU quota; oversubscribing it only gets throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _sha256(path: str):
    import hashlib
    try:
        h = hashlib.sha256()
        with open(path, "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 20), b""):
                h.update(blk)
        return h.hexdigest()
    except OSError:
        return None


def _resource_table():
    try:
        with open(os.path.join(REPO, "mink_amd", "kernel_resources.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def kernel_keys(kernel: str) -> list:
    """Names of every launch behind a `last_kernel()` string, as keyed in mink_amd/kernel_resources.json: the tight-rows call
    `ik_solve_kernel_48_72+redo_64` is two launches (the 48-row build and the full-row redo of the same feature set)."""
    if kernel.startswith("ik_quad_kernel"):
        nt = 32 if "_32" in kernel else (16 if "_16" in kernel else 8)          # (_32: two DPP rows per problem)
        return [f"ik_quad_kernel<{nt},{1 if kernel.endswith('_loop') else 0},{32 if nt == 32 else 16}>"]
    if kernel.startswith("ik_lane_kernel_"):
        return [f"ik_lane_kernel<{kernel.split('_')[3]},{1 if kernel.endswith('_loop') else 0}>"]
    wide = kernel.endswith("+wide")              # (the redo launch of the workgroup-per-problem kernel behind a wavefront kernel)
    if wide:
        kernel = kernel[:-len("+wide")]
    if kernel.startswith("convex_pre+"):         # (general convex pairs evaluated by a kernel in front of the analytic build)
        keys = ["convex_contacts_kernel"] + kernel_keys(kernel[len("convex_pre+"):] + ("+wide" if wide else ""))
        # (a problem with general convex pairs: the redo launch is the workgroup-per-problem kernel's build WITH the convex routine)
        return [("ik_wide_kernel_cvx" if k == "ik_wide_kernel" else k) for k in keys]
    if kernel == "ik_wide_kernel":
        return ["ik_wide_kernel"]
    main, _, redo = kernel.partition("+redo_")
    keys = [main]
    if redo:
        feat = int(main.split("_")[4])
        if "_r" in main[len("ik_solve_kernel_"):]:        # (a tight launch on the low-rank start, `48_40_r48`: its redo is the direct build)
            feat &= ~32
        keys.append(f"ik_solve_kernel_{redo}_{feat}")
    if wide:
        # (FEAT 136 / the all-feature builds 30 / 31 carry the general convex routine, and so does the redo launch behind them)
        feat_main = int(main.split("_")[4]) if main.startswith("ik_solve_kernel_") else 0
        keys.append("ik_wide_kernel_cvx" if (feat_main & 128) else "ik_wide_kernel")
    return keys


_LIB_SHA = {}


def loaded_library_sha256():
    """sha256 of the libminkhip.so THIS process loaded."""
    from mink_amd import _native as nat
    path = getattr(nat.lib(), "_name", None)
    if path not in _LIB_SHA:
        _LIB_SHA[path] = _sha256(path) if path else None
    return _LIB_SHA[path]


def _newest_pmc(config: str, B: int):
    """Newest committed PMC summary of this workload (tools/profile.sh → profiles/<tag>_<config>_b<B>_pmc.json;
    round-1 files of the headline config are named <tag>_g1_b65536_pmc.json)."""
    pats = [f"*_{config}_b{B}_pmc.json"] + ([f"*_g1_b{B}_pmc.json"] if config == "g1_c3" else [])
    paths = sorted(p for pat in pats for p in glob.glob(os.path.join(REPO, "profiles", pat)))
    for path in reversed(paths):
        try:
            with open(path) as fh:
                return json.load(fh), os.path.relpath(path, REPO)
        except (OSError, ValueError):
            continue
    return None, None


def profile_check(js, src, kernel: str) -> dict:
    """Is the committed counter summary a profile of what just ran?  Counters cannot be read inside a timed run, so `traffic`
    and the issue-slot shares come from profiles/ — but only from a summary stamped (tools/rocprof_summary.py "provenance") with
    the code objects of exactly the kernels this run launched; anything else is reported as stale and NOT replayed."""
    keys = kernel_keys(kernel)
    table = _resource_table()
    now = {k: (table.get(k) or {}).get("code_sha256") for k in keys}
    prov = (js or {}).get("provenance") or {}
    then = prov.get("kernel_code_sha256") or {}
    kernels_match = bool(js) and sorted(prov.get("kernels") or []) == sorted(keys)
    code_match = kernels_match and all(now[k] is not None and now[k] == then.get(k) for k in keys)
    lib_now = loaded_library_sha256()
    return {"source": src, "kernels_ran": keys, "kernels_profiled": prov.get("kernels"),
            "kernel_code_sha256": now, "profile_kernel_code_sha256": then or None,
            "library_sha256": lib_now, "profile_library_sha256": prov.get("library_sha256"),
            "library_match": lib_now is not None and lib_now == prov.get("library_sha256"),
            "profile_git_head": prov.get("git_head"),
            "kernel_code_match": code_match, "stale_profile": not code_match}


def measured_traffic(config: str, B: int, kernel: str):
    """HBM bytes per solve call from the PMC passes (FETCH_SIZE + WRITE_SIZE of EVERY launch of the call — a tight-rows solve
    is two —, separate rocprofv3 --pmc runs, calibrated on a known-size copy: tools/profile.sh, tools/rocprof_summary.py), from
    the newest committed summary of the same workload IF it is a profile of the code that just ran (profile_check).
    Returns (bytes or None, check)."""
    js, src = _newest_pmc(config, B)
    chk = profile_check(js, src, kernel)
    hbm = (js or {}).get("hbm", {})
    ok = not chk["stale_profile"] and "traffic_bytes_per_launch" in hbm
    return (hbm["traffic_bytes_per_launch"] if ok else None), chk


def measured_valu_issue(config: str, B: int, waves_per_simd: float, kernel: str):
    """What the kernel runs out of, from the same committed PMC summary as `traffic` (tools/profile.sh →
    tools/rocprof_summary.py "derived"): VALU issue = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a wave's cycles in which
    it issues a VALU instruction) × the waves that share a SIMD; LDS pipe = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES (the CU's one
    LDS index unit, bank-conflict replays included).  None without a summary of the code that ran (profile_check)."""
    js, src = _newest_pmc(config, B)
    if profile_check(js, src, kernel)["stale_profile"]:
        return None
    try:
        d = js.get("derived") or {}
        per_wave = d["valu_active_per_wave_cycle"]
        out = {"valu_active_share_of_wave_cycles": per_wave, "waves_per_simd": waves_per_simd,
               "simd_issue_slots_used": waves_per_simd * per_wave, "source": src}
        for key in ("lds_pipe_busy_per_cu_cycle", "lds_bank_conflict_per_lds_inst_active", "lds_bank_conflict_per_cu_cycle",
                    "wait_any_per_wave_cycle", "any_inst_active_per_wave_cycle", "wait_inst_any_per_wave_cycle",
                    "salu_cycles_per_wave_cycle"):
            if d.get(key) is not None:
                out[key] = d[key]
        if d.get("wave_cycles_per_dispatch"):
            out["per_solve"] = {"wave_quad_cycles": d["wave_cycles_per_dispatch"] / B,
                                "valu_insts": (d.get("valu_insts_per_dispatch") or 0) / B,
                                "salu_insts": (d.get("salu_insts_per_dispatch") or 0) / B,
                                "lds_insts": (d.get("lds_insts_per_dispatch") or 0) / B}
        return out
    except (TypeError, KeyError, ZeroDivisionError):
        return None


def binding_resource(valu):
    """The busier of the two candidates the counters can name: the SIMDs' VALU issue slots and the CU's LDS pipe."""
    if not valu:
        return {"name": None, "frac": None, "source": None}
    cands = {"valu_issue": valu["simd_issue_slots_used"]}
    if valu.get("lds_pipe_busy_per_cu_cycle") is not None:
        cands["lds_pipe"] = valu["lds_pipe_busy_per_cu_cycle"]
    name = max(cands, key=cands.get)
    return {"name": name, "frac": cands[name], "candidates": cands,
            "counters": {"valu_issue": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves per SIMD",
                         "lds_pipe": "SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES"},
            "source": valu["source"]}


def kernel_resources(kernel: str):
    """Registers / spills / scratch of the kernel that does the work as the compiler reported them when the library was built
    (mink_amd/kernel_resources.json, written by mink_amd/csrc/build.py)."""
    e = _resource_table().get(kernel_keys(kernel)[0])
    if e is None:
        return None
    return {"vgprs": e.get("vgprs"), "vgpr_spills": e.get("vgpr_spills_with_callees"), "sgpr_spills": e.get("sgpr_spills_with_callees"),
            "scratch_bytes_per_lane": e.get("scratch_bytes_per_lane"), "occupancy_waves_per_simd": e.get("occupancy_waves_per_simd")}


def measure_side_config(name, dev, steps=20, warmup=3, batch=None, seed=2000):
    """One more workload on the same device, in the same process: W warm-up + K timed launches with inputs resident in
    HBM, HIP events on the launch stream around every launch.  Returns the compact record of `other_configs`."""
    import torch

    from mink_amd import _native as nat
    from mink_amd import workloads

    cfg = workloads.BENCH_CONFIGS[name]
    B = batch or cfg["batch"]
    model = workloads.load_bench_robot(name)
    nm = nat.NativeModel(model, device=dev.index or 0)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    rng = np.random.default_rng(seed)
    q_h, tg_h, pt_h, ct_h = workloads.bench_batch(name, model, nm, prob, rng, B)
    dense_h = workloads.bench_dense(name, model, nm, q_h, rng)
    to = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    q, tg, pt, ct = to(q_h), to(tg_h), to(pt_h if prob.n_posture else None), to(ct_h)
    dense = None if dense_h is None else {k: to(x) for k, x in dense_h.items()}
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
    st = torch.empty((B,), dtype=torch.int32, device=dev)
    ev = []
    for i in range(warmup + steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dense)
        e1.record()
        if i >= warmup:
            ev.append((e0, e1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dense)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in ev]
    kern_ms = sum(ms) / len(ms)
    status = st.cpu().numpy()
    kernel = prob.last_kernel()
    redo = None
    if kernel.endswith("+wide"):
        # how many instances the second launch of the pair re-solved (more contacts in range than tableau rows and one violated, or
        # almost dependent active rows): the same batch on a handle WITHOUT the redo launch (mkh_problem_create_diag, MKH_DIAG_NO_WIDE_REDO)
        with nat.diag_options(nat.DIAG_NO_WIDE_REDO):
            alone, _, _ = workloads.bench_config(name, model, nm, B)
        st1 = torch.empty((B,), dtype=torch.int32, device=dev)
        alone.solve(q, tg, pt, ct, dt, damping, out=torch.empty_like(v), status_out=st1, dense=dense)
        s1 = st1.cpu().numpy()
        redo = {"row_overflow": int(((s1 & 16) != 0).sum()), "almost_dependent_rows": int(((s1 & 32) != 0).sum()),
                "failed_on_the_tableau": int(((s1 & (2 | 8)) != 0).sum()), "of": B}
        alone.close()
    bps = cfg["bytes_per_solve"]
    ach = bps * B / (kern_ms * 1e-3) / 1e9
    traffic, chk = measured_traffic(name, B, kernel)
    res = kernel_resources(kernel) or {}
    out = {"name": name, "batch": B, "kernel": kernel, "value": B * steps / wall, "unit": "solves/s", "steps": steps,
           "kernel_ms": kern_ms, "kernel_ms_median": statistics.median(ms),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": bps * B,
                        "traffic": traffic, "traffic_source": chk["source"], "stale_profile": chk["stale_profile"],
                        "kernel_code_match": chk["kernel_code_match"], "library_match": chk["library_match"]},
           "traffic_over_algorithmic": (traffic / (bps * B)) if traffic else None,
           "vgpr_spills": res.get("vgpr_spills"), "sgpr_spills": res.get("sgpr_spills"),
           "scratch_bytes_per_lane": res.get("scratch_bytes_per_lane"), "waves_per_simd": res.get("occupancy_waves_per_simd"),
           "failed_instances": int(((status & ~1) != 0).sum()),
           "redo_instances": redo, "workload": cfg["workload"]}
    if redo is None:
        out.pop("redo_instances")
    prob.close()
    return out


SIDE_CONFIGS = ("ur5e_c2", "shadow_c4", "g1_full", "g1_plugin", "ur5e_convex", "g1_coll", "h1_c3", "h1_full", "g1_hands", "aloha_coll")


def batch_sweep(prob, q, tg, pt, ct, dt, damping, v, st, batches=(1, 256, 1024, 4096, 16384, 65536), reps=20):
    """The headline problem at smaller batches (RL-sized callers run 1 k - 4 k environments): slices of the resident batch on the
    same handle, `reps` launches each between HIP events; kernel ms = the median launch.  One solve alone is the latency of the
    kernel's dependent instruction stream, a full machine needs 3 072 resident wavefronts."""
    import torch
    out = []
    for b in batches:
        if b > q.shape[0]:
            continue
        qs, ts, vs, ss = q[:b], tg[:b], v[:b], st[:b]
        cs = None if ct is None else (ct[:b] if ct.shape[0] == q.shape[0] else ct)
        for _ in range(3):
            prob.solve(qs, ts, pt, cs, dt, damping, out=vs, status_out=ss)
        ev = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); prob.solve(qs, ts, pt, cs, dt, damping, out=vs, status_out=ss); e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = statistics.median(a.elapsed_time(b_) for a, b_ in ev)
        out.append({"batch": b, "kernel": prob.last_kernel(), "kernel_ms": ms, "solves_per_s": b / (ms * 1e-3)})
    return out


def pcie_inclusive(prob, q_h, tg_h, pt_h, ct_h, dt, damping, reps=5):
    """Host-pointer call (the C ABI stages through its own device buffers: H2D q + targets, D2H v + status; from 32 MB on in
    chunks whose copies overlap the kernels of their neighbours)."""
    for _ in range(2):       # (the first calls also create the handle's copy streams and the runtime's staging for them)
        prob.solve(q_h, tg_h, pt_h, ct_h, dt, damping)
    t0 = time.perf_counter()
    for _ in range(reps):
        prob.solve(q_h, tg_h, pt_h, ct_h, dt, damping)
    return len(q_h) * reps / (time.perf_counter() - t0)


def loop_targets(prob, q, tg, pt, ct, dt, damping, B, max_iters=20, pos_threshold=1e-3, ori_threshold=1e-2, reps=3):
    """Secondary metric: the callers' whole loop — solve, integrate, break when every frame task is
    within (pos_threshold, ori_threshold), at most max_iters times (examples/arm_ur5e_actuators.py:88-97) — as ONE
    launch per batch (mkh_solve_until), device-resident inputs.  Reported twice by main(): `loop20_targets_per_s` with the
    examples' max_iters = 20 (at this workload's velocity limits — π rad/s × 5 ms per step against targets 0.15 rad away — nine
    instances in ten leave on the iteration cap: a 20-step-loop rate), and `converged_targets` with the budget it takes for
    ≥ 90 % of the instances to reach the thresholds."""
    import torch
    res = prob.solve(q, tg, pt, ct, dt, damping, n_steps=max_iters, until=(pos_threshold, ori_threshold))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = prob.solve(q, tg, pt, ct, dt, damping, n_steps=max_iters, until=(pos_threshold, ori_threshold))
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    iters, conv, st = res[3].cpu().numpy(), res[4].cpu().numpy(), res[2].cpu().numpy()
    return {"value": B / el, "unit": "targets/s", "ms_per_batch": 1e3 * el, "max_iters": max_iters,
            "pos_threshold": pos_threshold, "ori_threshold": ori_threshold, "mean_iterations": float(iters.mean()),
            "converged_fraction": float(conv.mean()), "failed_instances": int(((st & ~1) != 0).sum()),
            "kernel": prob.last_kernel(),
            "note": "one mkh_solve_until launch per batch: per-instance (solve, integrate, threshold test) loop on the device"}


def _oracle_specs(config, targets, posture_target, com_target):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_configs as oc

    if config == "g1_full":
        return oc.g1_full(targets, posture_target, com_target)
    return getattr(oc, config)(targets, posture_target)


def cpu_baseline(config, q, targets, posture_target, com_target, budget_s=20.0):
    """Restated reference timed on the host cores over a bounded sample of the same workload: the plain-C
    port of the reference pipeline (oracle/c: dense H, c, G, h + Goldfarb–Idnani, like mink + MuJoCo +
    quadprog; contact rows of plane / sphere / capsule pairs since round 4: the Shadow config) on all host threads.  Boxes
    where it cannot be built use the numpy port on one thread."""
    com0 = None if com_target is None else com_target[0, 0]
    try:
        m, tasks, limits, dt, damping = _oracle_specs(config, targets[0], posture_target[0], com0)
        from oracle import cport
        prob = cport.CProblem(m, tasks, limits)
        threads = usable_cpus()
        ct = lambda n: None if com_target is None else com_target[:n]          # per-instance CoM targets (B, 1, 3)
        n0 = min(len(q), 512 * threads)
        t0 = time.perf_counter()
        prob.solve_batch(q[:n0], targets[:n0], posture_target, dt, damping, com_target=ct(n0), nthreads=threads)
        rate = n0 / (time.perf_counter() - t0)
        total = max(n0, rate * budget_s / threads)        # ≈ budget_s core-seconds of CPU work in all
        n = int(min(len(q), total))
        reps = max(1, int(round(total / n)))
        t0 = time.perf_counter()
        bad = 0
        for _ in range(reps):
            _, st = prob.solve_batch(q[:n], targets[:n], posture_target, dt, damping, com_target=ct(n), nthreads=threads)
            bad += int((st != 0).sum())
        el = time.perf_counter() - t0
        return {"value": n * reps / el, "unit": "solves/s", "cores": threads, "kind": "port",
                "sample": f"{reps} x {n} {config} problems from the benchmark batch ({el * threads:.0f} core-seconds), "
                          f"oracle/c/mink_oracle.c (plain-C port of mink's dense pipeline + Goldfarb-Idnani), "
                          f"{threads} OpenMP threads (host CPU quota); {bad} failed"}
    except (OSError, ImportError, RuntimeError, TypeError, subprocess.CalledProcessError):
        return cpu_baseline_numpy(config, q, targets, posture_target, com_target, budget_s)


def cpu_baseline_numpy(config, q, targets, posture_target, com_target, budget_s=8.0):
    """The numpy restatement (oracle/ik.py): the same per-call structure and dispatch pattern as mink's Python
    (one Configuration update, per-task error/Jacobian, dense build_ik, one QP), one thread."""
    from oracle import ik
    n_done = 0
    t0 = time.perf_counter()
    while n_done < len(q) and time.perf_counter() - t0 < budget_s:
        com = None if com_target is None else com_target[n_done, 0]
        m, tasks, limits, dt, damping = _oracle_specs(config, targets[n_done], posture_target[0], com)
        ik.solve_ik(m, q[n_done], tasks, dt, damping, limits)
        n_done += 1
    el = time.perf_counter() - t0
    return {"value": n_done / el, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"{n_done} {config} problems from the benchmark batch, oracle/ik.py (numpy restatement of "
                      f"mink's per-call Python structure), 1 thread"}


def recorded_reference_baseline(config):
    """Real `mink.solve_ik` (the reference's own Python, /root/reference, over oracle/stubs for the absent mujoco /
    quadprog wheels) timed in the BUILD container by tools/time_reference_mink.py — the reference cannot travel
    to the GPU box, so this is a recorded figure from other host cores, not a live one."""
    path = os.path.join(REPO, "profiles", "r02_cpu_reference_mink.json")
    try:
        with open(path) as fh:
            js = json.load(fh)
        return dict(js[config], source=os.path.relpath(path, REPO))
    except (OSError, ValueError, KeyError):
        return None


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> None:
    """`python bench.py --gpus N` without a launcher: become N ranks under torch.distributed.run."""
    share = os.environ.get("MKH_BENCH_SHARE_GPU") == "1"
    if not share:
        from mink_amd import _native as nat
        have = nat.lib().mkh_device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) visible; refusing to "
                             f"report a {args.gpus}-GPU line from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py")] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="g1_c3", choices=["ur5e_c2", "g1_c3", "g1_full", "shadow_c4", "g1_plugin", "ur5e_convex", "g1_coll", "h1_c3", "h1_full", "g1_hands", "aloha_coll", "ur5e_coll"],
                    help="BASELINE config (default: the headline G1 config 3), or one of the two general routes of the "
                         "boundary: g1_plugin (caller-defined Task + Limit rows, mkh_solve_dense), ur5e_convex (a collision "
                         "pair on the general convex routine)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default g1_c3 line only: skip the `other_configs` records (every other named workload, 20 steps each)")
    ap.add_argument("--batch", type=int, default=None, help="problems per GPU (default: the config's BASELINE batch)")
    ap.add_argument("--gather", action="store_true",
                    help="(kept for compatibility) N>1 always reports the RCCL-gather variant in the 'gather' object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-loop-legs", action="store_true",
                    help="skip the fused-loop figures (loop20_targets_per_s, converged_targets): tools/profile.sh's trace pass, whose "
                         "per-kernel averages should be those of the timed launches")
    ap.add_argument("--no-pcie-leg", action="store_true",
                    help="skip the host-pointer (PCIe-inclusive) leg: it launches the SAME kernel on quarter batches (chunked "
                         "host path), which would pull down that kernel's average in a rocprofv3 --stats run of this command")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                 # does not return

    import torch

    from mink_amd import _native as nat
    from mink_amd import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hook for boxes with fewer GPUs than ranks (the N > 1 control flow can then be exercised on ONE GPU:
    # `MKH_BENCH_SHARE_GPU=1 python bench.py --gpus 2`): every rank uses device 0 and the process group runs on
    # gloo, because RCCL refuses two ranks on one device.
    share_gpu = os.environ.get("MKH_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: a line must report the GPUs it ran on")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # what the process group really is: its own world size and, per rank, the device it computes on (all-gathered), so that a
    # line for N GPUs shows N ranks on N distinct devices
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "name": props.name,
          "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                               getattr(props, "pci_device_id", 0)),
          "uuid": str(getattr(props, "uuid", "")), "host": socket.gethostname()}
    devices = [me]
    pg_world = 1
    if dist is not None:
        pg_world = dist.get_world_size()
        devices = [None] * pg_world
        dist.all_gather_object(devices, me)

    cfg = workloads.BENCH_CONFIGS[args.config]
    B = args.batch or cfg["batch"]
    model = workloads.load_bench_robot(args.config)
    nm = nat.NativeModel(model, device=local_rank)
    prob, dt, damping = workloads.bench_config(args.config, model, nm, B)
    rng = np.random.default_rng(1000 + rank)
    q_h, tg_h, pt_h, ct_h = workloads.bench_batch(args.config, model, nm, prob, rng, B)
    dense_h = workloads.bench_dense(args.config, model, nm, q_h, rng)
    q = torch.from_numpy(q_h).to(dev)
    tg = torch.from_numpy(tg_h).to(dev)
    pt = torch.from_numpy(pt_h).to(dev) if prob.n_posture else None
    ct = None if ct_h is None else torch.from_numpy(ct_h).to(dev)
    dense = None if dense_h is None else {k: torch.from_numpy(np.ascontiguousarray(x)).to(dev) for k, x in dense_h.items()}
    plain = dense is None and args.config not in ("ur5e_convex", "g1_coll", "h1_c3", "h1_full", "g1_hands", "aloha_coll", "ur5e_coll")      # configs the fused loop / host-path / oracle legs cover
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
    st = torch.empty((B,), dtype=torch.int32, device=dev)
    v_all = torch.empty((world * B, model.nv), dtype=torch.float64, device=dev) if (world > 1 and rank == 0) else None
    from mink_amd.distributed import gather_rows

    def timed_region(do_gather: bool):
        """W warm-up steps, then exactly K steps between barrier + synchronize; returns (elapsed s — MAX over
        ranks, per-step kernel ms from HIP events on the launch stream, per-step gather ms)."""
        kern_events, gath_events = [], []
        # (events are created before the timed region: on the short kernels — UR5e at 4 096 instances is 19 µs — creating
        #  two per step inside it costs as much host time as the launch itself)
        pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        # Launches shorter than 100 µs: events around every FOURTH timed step only — recording two events costs the host
        # ≈16 µs per step, more than such a kernel takes, and would put the host, not the device, into `value`.
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dense)
        p0.record()
        prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dense)
        p1.record()
        torch.cuda.synchronize()
        every = 4 if (p0.elapsed_time(p1) < 0.1 and not do_gather) else 1
        n_step = [0]

        def step(timed=False):
            if timed:
                n_step[0] += 1
                timed = (n_step[0] - 1) % every == 0
            if timed:
                e0, e1 = pool[len(kern_events)]
                e0.record()
            prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dense)
            if timed:
                e1.record()                               # HIP events on the launch stream, around the kernel only
                kern_events.append((e0, e1))
            if do_gather:
                gather_rows(v, world * B, dst=0, out=v_all)   # RCCL gather of v to rank 0 (tests/test_distributed_cpu.py)
                if timed:
                    e2 = torch.cuda.Event(enable_timing=True)
                    e2.record()
                    gath_events.append((e1, e2))

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(timed=True)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, [a.elapsed_time(b) for a, b in kern_events], [a.elapsed_time(b) for a, b in gath_events]

    # Python's cyclic GC must not run inside the timed region: with torch loaded a generation-2 collection
    # pauses the host for tens of ms — longer than the whole queue of launches takes to drain — and the GPU
    # idles (measured: always at the 23rd timed step, a 10-40 ms gap that cost 20 % at --steps 30).
    import gc
    gc.collect()
    gc.disable()
    elapsed, kern_list, _ = timed_region(False)
    gather = None
    if world > 1:
        g_el, _, g_list = timed_region(True)
        gather = {"value": world * B * args.steps / g_el, "unit": "solves/s", "ms_per_step": 1e3 * g_el / args.steps,
                  "gather_ms_rank0_median": statistics.median(g_list) if g_list else None,
                  "bytes_per_step_to_rank0": (world - 1) * B * model.nv * 8,
                  "note": "same K steps, each followed by the RCCL gather of v (B x nv f64 per rank) into rank 0"}
    gc.enable()
    kern_ms = sum(kern_list) / len(kern_list)             # average launch duration of the solve kernel (roofline)
    kern_ms_median = statistics.median(kern_list)
    # N > 1: what every rank saw on its own device (all-gathered), so that a scaling curve explains itself — a slow rank, a shared
    # device or a throttled one shows up as ITS kernel time, not as a lower total
    per_rank = [{"rank": rank, "kernel_ms": kern_ms, "kernel_ms_median": kern_ms_median, "value_alone": B / (kern_ms_median * 1e-3)}]
    if dist is not None:
        per_rank = [None] * pg_world
        dist.all_gather_object(per_rank, {"rank": rank, "kernel_ms": kern_ms, "kernel_ms_median": kern_ms_median,
                                          "value_alone": B / (kern_ms_median * 1e-3)})
    if os.environ.get("MKH_BENCH_DEBUG"):
        print("kernel ms per step:", [round(x, 3) for x in kern_list], file=sys.stderr)

    status = st.cpu().numpy()
    n_bad = int(((status & ~1) != 0).sum())
    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        bps = cfg["bytes_per_solve"]
        ach = bps * B / (kern_ms * 1e-3) / 1e9
        info = prob.launch_info(B)
        kernel = prob.last_kernel()
        # resident wavefronts per SIMD: the compiler's occupancy of the kernel that ran (kernel_resources.json), bounded by the
        # wavefronts the launch has (the row kernel at 4 096 instances is ONE wavefront per SIMD)
        occ = (kernel_resources(kernel) or {}).get("occupancy_waves_per_simd") or 1
        waves_per_simd = min(float(occ), max(1.0, info["grid"] / float(N_SIMDS)))
        traffic, chk = measured_traffic(args.config, B, kernel)
        valu = measured_valu_issue(args.config, B, waves_per_simd, kernel)
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": chk["source"], "stale_profile": chk["stale_profile"], "profile_check": chk,
                "kernel": kernel, "kernel_ms": kern_ms, "kernel_ms_median": kern_ms_median,
                "algorithmic_bytes_per_solve": bps, "algorithmic_bytes_per_launch": bps * B,
                # what the kernel really runs out of (PMC: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves
                # per SIMD, from the committed summary of this workload); HBM is the contract's nominal bound
                "binding_resource": binding_resource(valu),
                "note": "not HBM bound by design (one QP per wavefront / lane): fp64 VALU issue + the serial latency of a pivot "
                        "(round 1's LDS-return-bandwidth bound on the rank-1 updates was removed in round 2, "
                        "profiles/r02_ubench_rank1_mix.txt); HBM fraction reported as the contract requires",
                "valu_issue_view": valu}
        if args.config in ALGORITHMIC_FLOP_PER_SOLVE:
            af = ALGORITHMIC_FLOP_PER_SOLVE[args.config]
            flop = issued_flop_per_solve(kernel)
            sec = kern_ms * 1e-3
            roof["algorithmic_flop_view"] = {"flop_per_solve": af, "achieved_tflops": af * B / sec / 1e12,
                                             "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                             "frac": af * B / sec / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                             "source": "SURVEY.md §8(d): 0.10-0.15 Mflop fp64 per solve of the reference "
                                                       "algorithm (midpoint)"}
            # issued fp64 tableau work only (all 64 lanes of every rank-1 row, useful or not)
            roof["fp64_view"] = {"flop_per_solve": flop, "achieved_tflops": flop * B / sec / 1e12,
                                 "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                 "frac": flop * B / sec / 1e12 / FP64_VECTOR_PEAK_TFLOPS}
        metric = "IK solves/sec (whole node), Unitree G1 4 FrameTasks + box limits, batch 65536"
        if args.config != "g1_c3":
            metric = f"IK solves/sec (whole node), {args.config}, batch {B}"
        out = {
            "metric": metric,
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # median of the timed steps (per-step kernel time from HIP events); `value` above is the
            # contract's K steps / wall time between the barriers
            "value_median_of_steps": world * B / (kern_ms_median * 1e-3),
            "config": {"workload": cfg["workload"], "name": args.config,
                       "batch_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"batch-sharded x{world}, no data-path collective",
                       "world_size": pg_world, "backend": (dist.get_backend() if dist is not None else None),
                       "devices": devices,
                       "distinct_devices": len({(d["host"], d["pci_bus_id"], d["uuid"]) for d in devices}),
                       "launch": info, "failed_instances": n_bad},
            "roofline": roof,
        }
        if world > 1:
            ideal = sum(r["value_alone"] for r in per_rank)
            out["per_rank"] = per_rank
            out["scaling_view"] = {"sum_of_rank_rates_alone": ideal, "value_over_that": value / ideal,
                                   "note": "value_alone = batch / this rank's median kernel time: the rate the rank would sustain by itself; "
                                           "the batch shards with no data-path collective, so value / sum is what barriers, launch skew and "
                                           "the slowest rank cost (the driver computes the 1 -> N efficiency from its own N = 1 run)"}
        if gather is not None:
            # the optional gather of v to rank 0 against the links it uses: N - 1 peers, each over its own xGMI link into rank 0
            # (MI355X: 7 links x ~153 GB/s per GPU, point to point — /opt/skills/guides/MI355X_MICROARCH.md)
            gm = gather.get("gather_ms_rank0_median")
            if gm:
                gb = gather["bytes_per_step_to_rank0"] / (gm * 1e-3) / 1e9
                gather["achieved_GBps_into_rank0"] = gb
                gather["xgmi_links_used"] = world - 1
                gather["xgmi_peak_GBps_into_rank0"] = 153.0 * (world - 1)
                gather["frac_of_link_peak"] = gb / (153.0 * (world - 1))
            out["gather"] = gather
        res = kernel_resources(kernel)
        if res is not None:
            out["kernel_resources"] = res
        if world == 1 and plain and not args.no_loop_legs:
            out["loop20_targets_per_s"] = loop_targets(prob, q, tg, pt, ct, dt, damping, B, max_iters=20)
            # ... and a figure that deserves the name: targets that lie inside the joint ranges (the SURVEY §8(d) distribution of
            # the headline batch leaves most G1 targets just outside them, workloads.make_batch), the iteration budget raised
            # until ≥ 90 % of the instances are within the thresholds
            q_r, tg_r, pt_r, _ = workloads.bench_batch(args.config, model, nm, prob, np.random.default_rng(1000 + rank), B, reachable=True)
            q_r, tg_r, pt_r = torch.from_numpy(q_r).to(dev), torch.from_numpy(tg_r).to(dev), torch.from_numpy(pt_r).to(dev)
            conv = None
            for budget in (40, 80, 160, 320):
                conv = loop_targets(prob, q_r, tg_r, pt_r if prob.n_posture else None, ct, dt, damping, B, max_iters=budget, reps=1)
                if conv["converged_fraction"] >= 0.9:
                    break
            conv["note"] = ("consistent, reachable targets: frame targets = FK of a configuration inside the joint ranges, which is also "
                            "the instance's posture target (per-instance posture targets, +nq·8 B per solve); iteration budget "
                            "raised until converged_fraction >= 0.9 (or 320); " + conv["note"])
            out["converged_targets"] = conv
            if not args.no_pcie_leg:
                out["pcie_inclusive_value"] = pcie_inclusive(prob, q_h, tg_h, pt_h, ct_h, dt, damping)
                if args.config == "g1_c3" and args.batch is None:
                    out["batch_sweep"] = batch_sweep(prob, q, tg, pt, ct, dt, damping, v, st)
        if world == 1 and args.config == "g1_c3" and args.batch is None and not args.no_other_configs:
            # every other named workload behind the boundary, measured by the same command (20 launches each, < 0.2 s of
            # GPU time per config): the three other single-GPU BASELINE configs and the two general routes
            out["other_configs"] = [measure_side_config(name, dev) for name in SIDE_CONFIGS]
        if world == 1 and plain and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, q_h, tg_h, pt_h, ct_h)
            # mink's real dispatch pattern (per-call Python over numpy), next to the optimistic C port
            out["cpu_baseline_python"] = (out["cpu_baseline"] if "numpy" in out["cpu_baseline"]["sample"]
                                          else cpu_baseline_numpy(args.config, q_h, tg_h, pt_h, ct_h))
            ref = recorded_reference_baseline(args.config)
            if ref is not None:
                out["cpu_reference_recorded"] = ref
        if world > 1:
            # which fields of an N > 1 line were NOT measured by this run (round-5 review): the roofline's counter traffic and VALU
            # view come from the committed single-GPU profile of the same kernel; the CPU baseline, the closed-loop legs, the
            # PCIe-inclusive rate, the batch sweep and the other workloads are legs of the N = 1 line only
            out["n1_only_fields"] = {"absent_here": ["cpu_baseline", "cpu_baseline_python", "cpu_reference_recorded", "loop20_targets_per_s",
                                                     "converged_targets", "pcie_inclusive_value", "batch_sweep", "other_configs"],
                                     "from_the_single_gpu_profile": ["roofline.traffic", "roofline.valu_issue_view", "roofline.binding_resource"],
                                     "measured_by_this_run": ["value", "ms_per_step", "roofline.kernel_ms (rank 0)", "roofline.achieved", "per_rank",
                                                              "scaling_view", "gather"]}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
