#!/usr/bin/env python
"""Host-pointer (PCIe-inclusive) call rate of a BASELINE config, chunked (default) against single-shot
(MKH_DEBUG_NO_CHUNKS=1 in the environment).  GPU only.     python tools/bench_host_path.py [config] [reps]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from mink_amd import _native as nat
    from mink_amd import workloads
    config = sys.argv[1] if len(sys.argv) > 1 else "g1_c3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = workloads.BENCH_CONFIGS[config]
    B = cfg["batch"] if cfg["batch"] >= 16384 else 65536
    model = workloads.load_robot(cfg["robot"])
    nm = nat.NativeModel(model, device=0)
    prob, dt, damping = workloads.bench_config(config, model, nm, B)
    q, tg, pt, ct = workloads.bench_batch(config, model, nm, prob, np.random.default_rng(1), B)
    ts = []
    for _ in range(reps + 2):
        t0 = time.perf_counter()
        v, st = prob.solve(q, tg, pt, ct, dt, damping)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[2:])
    print("%-10s B=%d %s  %s: median %.3f ms (%.1f M solves/s), min %.3f, max %.3f" % (
        config, B, prob.last_kernel(), "single-shot" if os.environ.get("MKH_DEBUG_NO_CHUNKS") else "chunked    ",
        1e3 * np.median(ts), B / np.median(ts) / 1e6, 1e3 * ts.min(), 1e3 * ts.max()))


if __name__ == "__main__":
    main()
