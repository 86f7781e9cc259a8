#!/usr/bin/env python
"""Host cost of one device-resident `NativeProblem.solve` call: wall time per call of a loop of asynchronous launches of a
kernel much shorter than the host path (UR5e config 2, B = 64), and the cProfile breakdown of the Python side.

    python tools/bench_call_overhead.py [calls]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from mink_amd import _native as nat
    from mink_amd import workloads

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dev = torch.device("cuda", 0)
    model = workloads.load_bench_robot("ur5e_c2")
    nm = nat.NativeModel(model, device=0)
    for B in (64, 4096):
        prob, dt, damping = workloads.bench_config("ur5e_c2", model, nm, B)
        q_h, tg_h, pt_h, _ = workloads.bench_batch("ur5e_c2", model, nm, prob, np.random.default_rng(0), B)
        q, tg, pt = torch.from_numpy(q_h).to(dev), torch.from_numpy(tg_h).to(dev), torch.from_numpy(pt_h).to(dev)
        v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
        st = torch.empty((B,), dtype=torch.int32, device=dev)
        for _ in range(100):
            prob.solve(q, tg, pt, None, dt, damping, out=v, status_out=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            prob.solve(q, tg, pt, None, dt, damping, out=v, status_out=st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("B=%d: %.2f us per call to issue, %.2f us per call until the device is idle (%s)" %
              (B, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n, prob.last_kernel()))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        prob.solve(q, tg, pt, None, dt, damping, out=v, status_out=st)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)


if __name__ == "__main__":
    main()
