#!/usr/bin/env python
"""Per-phase shader-clock breakdown of the PRODUCTION kernels (no taps): needs an experiment build of the library compiled
with -DMKH_CLOCKS, in which every variant stamps its phase boundaries into SolveArgs::clk.  GPU only.

    MKH_BUILD_TAG=clk MKH_EXTRA_FLAGS=-DMKH_CLOCKS python -m mink_amd.csrc.build      # (build container)
    MKH_LIB_TAG=clk python tools/phase_clocks.py g1_c3 [shadow_c4 ur5e_c2:4096 ...]   # (GPU box)

The stamps cost a few s_memtime + one store per problem; phases inside real callees (MKH_CALLS builds: FK, axes, task lanes)
are lumped into the first interval."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
path = os.path.join(tempfile.gettempdir(), "mkh_clocks.bin")
os.environ["MKH_DEBUG_CLOCKS"] = path

import torch  # noqa: E402

from mink_amd import _native as nat  # noqa: E402
from mink_amd import workloads  # noqa: E402

# (the workgroup-per-problem redo launch would overwrite the stamps of the wavefront kernel: every handle of this process is
#  created with MKH_DIAG_NO_WIDE_REDO unless MKH_PC_WIDE=1)
if not os.environ.get("MKH_PC_WIDE"):
    nat._diag_default.bits = nat.DIAG_NO_WIDE_REDO

NAMES = ["load+FK", "axes/dof/com", "task lanes", "posture+coll+J cols", "limits", "build T + phase 0", "active set"]


def run(name, B=None):
    cfg = workloads.BENCH_CONFIGS[name]
    B = B or cfg["batch"]
    model = workloads.load_bench_robot(name)
    nm = nat.NativeModel(model)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    rng = np.random.default_rng(0)
    q, tg, pt, ct = workloads.bench_batch(name, model, nm, prob, rng, B)
    dense = workloads.bench_dense(name, model, nm, q, rng)
    dev = torch.device("cuda", 0)
    to = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    args = (to(q), to(tg), to(pt if prob.n_posture else None), to(ct), dt, damping)
    dn = None if dense is None else {k: to(x) for k, x in dense.items()}
    kw = {k: True for k in os.environ.get("MKH_PC_FLAGS", "").split(",") if k}      # e.g. MKH_PC_FLAGS=direct_qp,two_waves
    for _ in range(3):
        prob.solve(*args, dense=dn, **kw)
    torch.cuda.synchronize()
    c = np.fromfile(path, dtype=np.int64).reshape(-1, 24)[:B]
    ok = c[:, 0] != 0
    c = c[ok]
    tot = c[:, 7] - c[:, 0]
    print(f"{name} B={B} kernel {prob.last_kernel()} launch {prob.launch_info(B)}")
    if prob.last_kernel().startswith("ik_quad_kernel"):  # row-per-problem kernel: its own six intervals
        d = np.diff(c[:, :7], axis=1)
        print("  per-row cycles: mean %.0f  p50 %.0f  p99 %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 99)))
        for k, n in enumerate(["load q", "FK (links)", "task lanes", "objective + limits", "QP: all dofs in", "QP: flips"]):
            if k < d.shape[1]:
                print("    %-22s mean %8.0f  (%4.1f%%)" % (n, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))
        prob.close()
        return
    print("  per-problem wave cycles: mean %.0f  p50 %.0f  p99 %.0f  max %d   (%d problems stamped)" % (
        tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max(), len(c)))
    st = c[:, :8].copy()
    for k in range(1, 8):                       # stamps a callee build never wrote: carry the previous one
        z = st[:, k] == 0
        st[z, k] = st[z, k - 1]
    d = np.diff(st, axis=1)
    for k, n in enumerate(NAMES):
        print("    %-22s mean %8.0f  (%4.1f%%)   p50 %7.0f  p90 %7.0f  p99 %7.0f  max %7d" % (
            n, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean(), np.median(d[:, k]), np.percentile(d[:, k], 90),
            np.percentile(d[:, k], 99), d[:, k].max()))
    sub = ["phase-0 publish", "phase-0 rcp+pivot", "GI select", "flip/GI publish", "flip/GI ratio test", "flip/GI pivot"]
    if "_r" in prob.last_kernel():
        sub = ["low-rank: J rows", "low-rank: S, w", "low-rank: elimination", "low-rank: rank-1 dof block (+publish)", "ratio test", "pivot"]
    for k, n in enumerate(sub):
        print("      %-36s mean %8.0f" % (n, c[:, 8 + k].mean()))
    if (c[:, 20] != 0).any():                   # callee builds: pre_phases stamps its own phases
        print("      pre_phases: FK %8.0f | axes / dof / com %8.0f | task lanes %8.0f" % (
            (c[:, 20] - c[:, 0]).mean(), (c[:, 21] - c[:, 20]).mean(), (st[:, 3] - c[:, 21]).mean()))
    if (c[:, 22] != 0).any():
        print("      after the task lanes: posture / damping tasks %8.0f | box limits %8.0f | to the callee's entry %8.0f" % (
            (c[:, 22] - st[:, 3]).mean(), (c[:, 23] - c[:, 22]).mean(), (c[:, 16] - c[:, 23]).mean()))
    if (c[:, 16] != 0).any():
        print("      wood_start: posture / box %8.0f | J rows %8.0f | S, w %8.0f | elimination %8.0f | then rank-1 block … to tick 4: %8.0f" % (
            (c[:, 16] - st[:, 3]).mean(), (c[:, 17] - c[:, 16]).mean(), (c[:, 18] - c[:, 17]).mean(), (c[:, 19] - c[:, 18]).mean(),
            (st[:, 4] - c[:, 19]).mean()))
    prob.close()


if __name__ == "__main__":
    for a in sys.argv[1:] or ["g1_c3"]:
        n, _, b = a.partition(":")
        run(n, int(b) if b else None)
