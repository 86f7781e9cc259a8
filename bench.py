#!/usr/bin/env python
"""Headline benchmark: batched differential-IK solves/sec on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric, configs[2]): Unitree G1 (nq=44, nv=43), 4 FrameTasks
(feet + palms) + PostureTask + ConfigurationLimit + VelocityLimit box limits, batch
65 536 per GPU, float64.  A "step" is one batched solve_ik over the resident batch
(inputs already in HBM).  For N > 1 the driver launches one rank per GPU via
torch.distributed.run; the batch shards by rank (weak scaling, configs[4] = 8 x 65 536)
with no data-path collective: the instances are independent, v stays on the GPU that produced it
(--gather adds the RCCL gather of v to rank 0 that BASELINE configs[4] mentions).

Prints ONE JSON line (rank 0) with the whole-job solves/sec, the HBM roofline of the
kernel (algorithmic bytes / launch duration from HIP events) and a CPU baseline.
"""

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BYTES_PER_SOLVE_G1 = 44 * 8 + 4 * 7 * 8 + 43 * 8 + 4   # q + 4 frame targets + v + status = 924 B
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


HBM_PEAK_GBS = 8000.0                                   # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6                          # MI355X fp64 vector peak (same guide, chip table)


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


def measured_traffic():
    """HBM bytes per launch of the IK kernel from the PMC passes (FETCH_SIZE + WRITE_SIZE, separate
    rocprofv3 --pmc runs, calibrated on a known-size copy: tools/profile.sh, tools/rocprof_summary.py).
    Counters cannot be read from inside a timed run, so this comes from the newest committed
    profiles/*_pmc.json of the same workload; None when no calibrated summary exists."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*_g1_b65536_pmc.json"))):
        try:
            with open(path) as fh:
                hbm = json.load(fh).get("hbm", {})
            if "traffic_bytes_per_launch" in hbm:
                best = (hbm["traffic_bytes_per_launch"], os.path.relpath(path, REPO))
        except (OSError, ValueError):
            pass
    return best


def measured_valu_issue():
    """What actually bounds the kernel: VALU issue.  From the same committed PMC summary as `traffic`
    (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the solve dispatches): the share of a wave's cycles in which it issues
    a VALU instruction, times the two waves that share a SIMD for this kernel.  None without a summary."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*_g1_b65536_pmc.json"))):
        try:
            with open(path) as fh:
                k = json.load(fh).get("ik_solve_kernel", {})
            act = k["SQ_ACTIVE_INST_VALU"]["per_dispatch"][1:]
            cyc = k["SQ_WAVE_CYCLES"]["per_dispatch"][1:]
            per_wave = sum(act) / sum(cyc)
            best = {"valu_active_share_of_wave_cycles": per_wave, "waves_per_simd": 2,
                    "simd_issue_slots_used": 2 * per_wave, "source": os.path.relpath(path, REPO)}
        except (OSError, ValueError, KeyError, ZeroDivisionError):
            pass
    return best


def pcie_inclusive(prob, q_h, tg_h, stand, dt, damping, reps=3):
    """Host-pointer call (the C ABI stages through pinned buffers: H2D q + targets, D2H v + status)."""
    prob.solve(q_h, tg_h, stand[None, :], None, dt, damping)
    t0 = time.perf_counter()
    for _ in range(reps):
        prob.solve(q_h, tg_h, stand[None, :], None, dt, damping)
    return len(q_h) * reps / (time.perf_counter() - t0)


def cpu_baseline(model, q, targets, posture_target, budget_s=20.0):
    """Restated reference timed on the host cores over a bounded sample of the same workload: the plain-C
    port of the reference pipeline (oracle/c: dense H, c, G, h + Goldfarb–Idnani, like mink + MuJoCo +
    quadprog) on all host threads; falls back to the numpy port on one thread if the C oracle cannot be built."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_configs as oc

    m, tasks, limits, dt, damping = oc.g1_c3(targets[0], posture_target)
    try:
        from oracle import cport
        prob = cport.CProblem(m, tasks, limits)
        threads = usable_cpus()
        n0 = min(len(q), 512 * threads)
        t0 = time.perf_counter()
        prob.solve_batch(q[:n0], targets[:n0], posture_target[None, :], dt, damping, nthreads=threads)
        rate = n0 / (time.perf_counter() - t0)
        total = max(n0, rate * budget_s / threads)        # ≈ budget_s core-seconds of CPU work in all
        n = int(min(len(q), total))
        reps = max(1, int(round(total / n)))
        t0 = time.perf_counter()
        bad = 0
        for _ in range(reps):
            _, st = prob.solve_batch(q[:n], targets[:n], posture_target[None, :], dt, damping, nthreads=threads)
            bad += int((st != 0).sum())
        el = time.perf_counter() - t0
        return {"value": n * reps / el, "unit": "solves/s", "cores": threads, "kind": "port",
                "sample": f"{reps} x {n} G1 config-3 problems from the benchmark batch ({el * threads:.0f} core-seconds), "
                          f"oracle/c/mink_oracle.c (plain-C port of mink's dense pipeline + Goldfarb-Idnani), "
                          f"{threads} OpenMP threads (host CPU quota); {bad} failed"}
    except (OSError, ImportError, RuntimeError, subprocess.CalledProcessError):
        from oracle import ik
        n_done = 0
        t0 = time.perf_counter()
        while n_done < len(q) and time.perf_counter() - t0 < budget_s:
            m, tasks, limits, dt, damping = oc.g1_c3(targets[n_done], posture_target)
            ik.solve_ik(model, q[n_done], tasks, dt, damping, limits)
            n_done += 1
        el = time.perf_counter() - t0
        return {"value": n_done / el, "unit": "solves/s", "cores": 1, "kind": "port",
                "sample": f"{n_done} G1 config-3 problems from the benchmark batch, oracle/ik.py (numpy), 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="problems per GPU")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: end every step with an RCCL gather of v to rank 0 (BASELINE configs[4] as written); by "
                         "default v stays sharded on the GPU that produced it — the instances are independent, so "
                         "the path has no exchange step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from mink_amd import _native as nat
    from mink_amd import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hook for boxes with fewer GPUs than ranks (the N > 1 control flow can then be exercised on ONE GPU:
    # `MKH_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`): every rank
    # uses device 0 and the process group runs on gloo, because RCCL refuses two ranks on one device.
    share_gpu = os.environ.get("MKH_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    B = args.batch
    model = workloads.load_robot("g1")
    nm = nat.NativeModel(model, device=local_rank)
    prob, dt, damping = workloads.g1_config(model, nm, B)
    rng = np.random.default_rng(1000 + rank)
    stand = model.key_qpos[model.name2id("key", "stand")]
    q_h, tg_h = workloads.make_batch(model, nm, prob, rng, B, base_q=stand)
    q = torch.from_numpy(q_h).to(dev)
    tg = torch.from_numpy(tg_h).to(dev)
    pt = torch.from_numpy(stand[None, :].copy()).to(dev)
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
    st = torch.empty((B,), dtype=torch.int32, device=dev)
    do_gather = world > 1 and args.gather
    v_all = torch.empty((world * B, model.nv), dtype=torch.float64, device=dev) if (do_gather and rank == 0) else None
    from mink_amd.distributed import gather_rows

    kern_events = []

    def step(timed=False):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        prob.solve(q, tg, pt, None, dt, damping, out=v, status_out=st)
        if timed:
            e1.record()                                 # HIP events on the launch stream, around the kernel only
            kern_events.append((e0, e1))
        if do_gather:
            gather_rows(v, world * B, dst=0, out=v_all)   # RCCL gather of v to rank 0 (tests/test_distributed_cpu.py)

    # Python's cyclic GC must not run inside the timed region: with torch loaded a generation-2 collection
    # pauses the host for tens of ms — longer than the whole queue of launches takes to drain — and the GPU
    # idles (measured: always at the 23rd timed step, a 10-40 ms gap that cost 20 % at --steps 30).
    import gc
    gc.collect()
    gc.disable()
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
    gc.enable()
    kern_ms = sum(a.elapsed_time(b) for a, b in kern_events) / args.steps   # average launch duration of ik_solve_kernel
    if os.environ.get("MKH_BENCH_DEBUG"):
        print("kernel ms per step:", [round(a.elapsed_time(b), 3) for a, b in kern_events], file=sys.stderr)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    status = st.cpu().numpy()
    n_bad = int(((status & ~1) != 0).sum())
    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        ach = BYTES_PER_SOLVE_G1 * B / (kern_ms * 1e-3) / 1e9
        info = prob.launch_info(B)
        kernel = prob.last_kernel()
        flop = issued_flop_per_solve(kernel)
        traffic = measured_traffic() if B == 65536 else None
        out = {
            "metric": "IK solves/sec (whole node), Unitree G1 4 FrameTasks + box limits, batch 65536",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Unitree G1 (nq=44,nv=43): 4 FrameTasks(feet+palms)+PostureTask+"
                                   "ConfigurationLimit+VelocityLimit, dt=5e-3, damping=1e-1 (BASELINE configs[2])",
                       "batch_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"batch-sharded x{world}" + (", RCCL gather of v" if do_gather else ""),
                       "launch": info, "failed_instances": n_bad},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": kernel, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_solve": BYTES_PER_SOLVE_G1,
                         "algorithmic_bytes_per_launch": BYTES_PER_SOLVE_G1 * B,
                         "note": "fp64 VALU-issue bound by design (serial pivots of one QP per wavefront); "
                                 "HBM fraction reported as the contract requires, fp64 view alongside",
                         # secondary compute view: issued fp64 tableau work only (issued_flop_per_solve)
                         "fp64_view": {"flop_per_solve": flop,
                                       "achieved_tflops": flop * B / (kern_ms * 1e-3) / 1e12,
                                       "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                       "frac": flop * B / (kern_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS},
                         # ... and every VALU instruction (PMC): the resource the kernel actually runs out of
                         "valu_issue_view": measured_valu_issue()},
        }
        if world == 1:
            out["pcie_inclusive_value"] = pcie_inclusive(prob, q_h, tg_h, stand, dt, damping)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, q_h, tg_h, stand)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
