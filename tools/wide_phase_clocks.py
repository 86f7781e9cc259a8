#!/usr/bin/env python
"""Per-phase cycle breakdown of the workgroup-per-problem kernel (wide_kernel.h MKH_WSTAMP): MKH_DEBUG_CLOCKS makes every launch of
this process synchronous and dumps the shader-clock stamps of every problem.

    MKH_BUILD_TAG=dbg MKH_EXTRA_FLAGS=-DMKH_DEBUG_SWITCHES python -m mink_amd.csrc.build     # (build container; round 6: the product
    MKH_LIB_TAG=dbg python tools/wide_phase_clocks.py [config[:batch]] ...                      #  library reads no environment variable)
(default config: g1_hands)
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
path = os.path.join(tempfile.gettempdir(), "mkh_wide_clk.bin")
os.environ["MKH_DEBUG_CLOCKS"] = path

from mink_amd import _native as nat  # noqa: E402
from mink_amd import workloads  # noqa: E402

NAMES = ["FK", "dof axes / CoM", "task lanes", "posture / LM", "Jacobian rows", "c / box", "contacts", "row selection", "tableau (H, A)",
         "QP phase 0", "QP active set", "write-back"]


def run(name, B=None):
    cfg = workloads.BENCH_CONFIGS[name]
    B = B or cfg["batch"]
    model = workloads.load_bench_robot(name)
    nm = nat.NativeModel(model)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    q, tg, pt, ct = workloads.bench_batch(name, model, nm, prob, np.random.default_rng(2000), B)
    for _ in range(2):
        v, st = prob.solve(q, tg, pt, ct, dt, damping)
    assert prob.last_kernel().endswith("wide_kernel") or prob.last_kernel().endswith("+wide"), prob.last_kernel()
    c = np.fromfile(path, dtype=np.int64).reshape(-1, 24)[:B]
    if prob.last_kernel().endswith("+wide"):           # a redo launch: only the re-solved instances carry stamps
        c = c[c[:, 12] > 0]
        print("redo launch behind %s: %d instances re-solved" % (prob.last_kernel(), len(c)))
        B = len(c)
    d = np.diff(c[:, :13], axis=1).astype(np.float64)
    tot = (c[:, 12] - c[:, 0]).astype(np.float64)
    print("%s, B = %d: per problem %.0f k core-clock cycles; N = %.0f, ratio-test rounds %.1f, pivots after phase 0 %.1f"
          % (name, B, tot.mean() / 1e3, c[:, 16].mean(), c[:, 13].mean(), c[:, 14].mean()))
    print("  per problem: p50 %.0f k, p90 %.0f k, p99 %.0f k, max %.0f k cycles; ratio-test rounds: p50 %.0f, p99 %.0f, max %.0f"
          % (np.percentile(tot, 50) / 1e3, np.percentile(tot, 90) / 1e3, np.percentile(tot, 99) / 1e3, tot.max() / 1e3,
             np.percentile(c[:, 13], 50), np.percentile(c[:, 13], 99), c[:, 13].max()))
    if (c[:, 17:23] > 0).any():
        dn = ["factors + x0", "violated-constraint search", "directions d, z, r", "step lengths + updates", "add (Householder)", "drop (Givens)"]
        ds = c[:, 17:23].astype(np.float64)
        print("  dense Goldfarb–Idnani iteration (%d instances): " % (ds.sum(axis=1) > 0).sum() + "; ".join("%s %.0f k" % (a, b / 1e3) for a, b in zip(dn, ds.mean(axis=0))))
    for k, n in enumerate(NAMES):
        print("  %-16s %9.0f cycles  %5.1f %%" % (n, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))


if __name__ == "__main__":
    for a in (sys.argv[1:] or ["g1_hands"]):
        n, _, b = a.partition(":")
        run(n, int(b) if b else None)
