#!/usr/bin/env python
"""Stress run of the low-rank QP start on G1 config 3: random batches over step sizes and dampings, the production kernel
(cold-start refinement, 3 waves per SIMD) against the same start without the refinement and against the direct start.
GPU only.      python tools/stress_lowrank.py [batches=8] [B=65536]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import native_configs as nc
    from mink_amd import _native as nat
    from mink_amd import workloads
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    model = workloads.load_robot("g1")
    nm = nat.NativeModel(model)
    prob, dt, _ = nc.build("g1_c3", nm, B)
    with nat.diag_options(nat.DIAG_NO_COLD_REFINE):
        plain, _, _ = nc.build("g1_c3", nm, B)
    stand = model.key_qpos[0]
    worst = 0.0
    for i in range(n_batches):
        rng = np.random.default_rng(1000 + i)
        q, tg = workloads.make_batch(model, nm, prob, rng, B, base_q=stand)
        scale = float(rng.choice([0.25, 1.0, 4.0, 16.0]))
        damping = float(rng.choice([1e-2, 1e-1, 1.0]))
        v, st = prob.solve(q, tg, stand[None, :], None, dt * scale, damping)
        k = prob.last_kernel()
        vp, stp = plain.solve(q, tg, stand[None, :], None, dt * scale, damping)
        vd, std = plain.solve(q, tg, stand[None, :], None, dt * scale, damping, direct_qp=True)
        s = max(1.0, np.abs(vd).max())
        e1, e2 = np.abs(v - vp).max() / s, np.abs(v - vd).max() / s
        worst = max(worst, e1, e2)
        print("batch %d: dt x %-5g damping %-5g %s  failed %d / %d / %d   vs no refinement %.1e   vs direct start %.1e" % (
            i, scale, damping, k, int((st & ~1 != 0).sum()), int((stp & ~1 != 0).sum()), int((std & ~1 != 0).sum()), e1, e2))
    print("worst relative difference %.2e over %d problems" % (worst, n_batches * B))


if __name__ == "__main__":
    main()
