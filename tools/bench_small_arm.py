#!/usr/bin/env python
"""UR5e config 2 (BASELINE.json configs[2]) at a sweep of batch sizes on each of the three kernels a small arm can run
on — wavefront per problem (ik_kernel.h), 16-lane row per problem (quad_kernel.h), lane per problem (lane_kernel.h) —
device-resident inputs, HIP-event timing of back-to-back solves.  Where the dispatch thresholds in minkhip.hip
(`launch`) come from.

    python tools/bench_small_arm.py [reps] [config] [loop]      (loop: the threshold-terminated caller loop, max_iters 20,
                                                                 1 mm / 0.01 rad — M targets/s instead of M solves/s)
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch

    from mink_amd import _native as nat
    from mink_amd import workloads

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    config = sys.argv[2] if len(sys.argv) > 2 else "ur5e_c2"
    loop = len(sys.argv) > 3 and sys.argv[3] == "loop"
    extra = {"n_steps": 20, "until": (1e-3, 1e-2)} if loop else {}
    dev = torch.device("cuda", 0)
    model = workloads.load_bench_robot(config)
    nm = nat.NativeModel(model, device=0)
    print("%8s  %10s %10s %10s   (M solves/s; default dispatch marked *)" % ("B", "wavefront", "row", "lane"))
    for B in (16, 256, 1024, 2048, 4096, 6144, 8192, 16384, 32768, 65536, 131072, 262144, 1048576):
        prob, dt, damping = workloads.bench_config(config, model, nm, B)
        q_h, tg_h, pt_h, ct_h = workloads.bench_batch(config, model, nm, prob, np.random.default_rng(1000), B)
        q, tg, pt = torch.from_numpy(q_h).to(dev), torch.from_numpy(tg_h).to(dev), torch.from_numpy(pt_h).to(dev)
        v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
        st = torch.empty((B,), dtype=torch.int32, device=dev)
        prob.solve(q, tg, pt, None, dt, damping, out=v, status_out=st, **extra)
        default = prob.last_kernel()
        cells = []
        for kw in ({"wave_kernel": True}, {"quad_kernel": True}, {"lane_kernel": True}):
            for _ in range(20):
                prob.solve(q, tg, pt, None, dt, damping, out=v, status_out=st, **kw, **extra)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = max(reps, min(1000, (1 << 20) // B)) // (10 if loop else 1)   # (short kernels: enough launches for the clocks to settle)
            for _ in range(n):
                prob.solve(q, tg, pt, None, dt, damping, out=v, status_out=st, **kw, **extra)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            bad = int(((st.cpu().numpy() & ~1) != 0).sum())
            cells.append("%9.1f%s" % (B / ms / 1e3, "*" if prob.last_kernel() == default else ("!" if bad else " ")))
        print("%8d  %s" % (B, " ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
