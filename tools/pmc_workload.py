#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes (see tools/profile.sh).

Dispatch sequence, in order:
  1. a CALIBRATION copy kernel with a known byte count (torch `dst.copy_(src)` of 512 MiB of
     float64: 512 MiB read + 512 MiB written, well past the 256 MiB Infinity Cache), so that
     FETCH_SIZE / WRITE_SIZE can be turned into bytes for THIS access width on gfx950
     (MI355X_MICROARCH.md, HBM section: the counters are uncalibrated for anything but wide reads);
  2. the FK-only launch that manufactures reachable targets (workloads.make_batch);
  3. N timed-style solves of the bench workload (device-resident).

    python tools/pmc_workload.py [n_solves] [batch] [config]      (defaults: 4, the config's batch, g1_c3)
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    config = sys.argv[3] if len(sys.argv) > 3 else "g1_c3"
    import torch

    from mink_amd import _native as nat
    from mink_amd import workloads

    dev = torch.device("cuda", 0)
    src = torch.ones(64 * 1024 * 1024, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)                       # calibration dispatch
    torch.cuda.synchronize()
    del src, dst

    cfg = workloads.BENCH_CONFIGS[config]
    B = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else cfg["batch"]
    model = workloads.load_bench_robot(config)
    nm = nat.NativeModel(model, device=0)
    prob, dt, damping = workloads.bench_config(config, model, nm, B)
    rng = np.random.default_rng(1000)
    q_h, tg_h, pt_h, ct_h = workloads.bench_batch(config, model, nm, prob, rng, B)
    q = torch.from_numpy(q_h).to(dev)
    tg = torch.from_numpy(tg_h).to(dev)
    pt = torch.from_numpy(pt_h).to(dev) if prob.n_posture else None
    ct = None if ct_h is None else torch.from_numpy(ct_h).to(dev)
    dense_h = workloads.bench_dense(config, model, nm, q_h, rng)
    dense = None if dense_h is None else {k: torch.from_numpy(np.ascontiguousarray(x)).to(dev) for k, x in dense_h.items()}
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
    st = torch.empty((B,), dtype=torch.int32, device=dev)
    for _ in range(n):
        prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dense)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
