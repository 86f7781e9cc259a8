#!/usr/bin/env python
"""Round 6: the low-rank start WITH half-space rows against the direct start (MKH_FLAG_DIRECT_QP) on the collision / plugin bench
workloads — same handle, same device-resident batch: kernel names, agreement of v and status, HIP-event time of both.  GPU only.

    python tools/ab_rows_wood.py [g1_coll shadow_c4 g1_plugin ...] [--reps 30]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from mink_amd import _native as nat  # noqa: E402
from mink_amd import workloads  # noqa: E402


def run(name, reps):
    cfg = workloads.BENCH_CONFIGS[name]
    B = cfg["batch"]
    model = workloads.load_bench_robot(name)
    nm = nat.NativeModel(model)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    rng = np.random.default_rng(0)
    q, tg, pt, ct = workloads.bench_batch(name, model, nm, prob, rng, B)
    dense = workloads.bench_dense(name, model, nm, q, rng)
    dev = torch.device("cuda", 0)
    to = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    qd, tgd, ptd, ctd = to(q), to(tg), to(pt), to(ct)
    dd = None if dense is None else {k: to(v) for k, v in dense.items()}
    out = {}
    for label, kw in (("low-rank", {}), ("direct", {"direct_qp": True})):
        v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
        st = torch.empty((B,), dtype=torch.int32, device=dev)
        for _ in range(3):
            prob.solve(qd, tgd, ptd, ctd, dt, damping, out=v, status_out=st, dense=dd, **kw)
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prob.solve(qd, tgd, ptd, ctd, dt, damping, out=v, status_out=st, dense=dd, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        out[label] = (v.cpu().numpy(), st.cpu().numpy(), float(np.median(ms)), prob.last_kernel())
    (v1, s1, t1, k1), (v0, s0, t0, k0) = out["low-rank"], out["direct"]
    ok = (s1 & ~1) == 0
    both = ok & ((s0 & ~1) == 0)
    err = np.abs(v1[both] - v0[both]).max(axis=1) / np.maximum(1.0, np.abs(v0[both]).max(axis=1))
    print(f"{name} B={B}\n  default : {k1:48s} {t1:.4f} ms\n  direct  : {k0:48s} {t0:.4f} ms   x{t0 / t1:.3f}")
    print(f"  status: default {np.unique(s1, return_counts=True)}, direct {np.unique(s0, return_counts=True)}")
    print(f"  max rel |v - v_direct| over {int(both.sum())} instances: {err.max():.2e} (p99 {np.quantile(err, 0.99):.2e}); worst instance {int(np.flatnonzero(both)[err.argmax()])}")
    prob.close()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = 30
    if "--reps" in sys.argv:
        reps = int(sys.argv[sys.argv.index("--reps") + 1]); args = [a for a in args if a != str(reps)]
    for n in (args or ["g1_coll"]):
        run(n, reps)
