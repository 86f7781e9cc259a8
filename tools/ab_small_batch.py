#!/usr/bin/env python
"""Round 6: the headline problem (G1 config 3) at small batches — the default dispatch against MKH_FLAG_TWO_WAVES (the 256-register
map, phases inlined: no call prologues in the dependent chain) and MKH_FLAG_DIRECT_QP, launches between HIP events.  GPU only.

    python tools/ab_small_batch.py [--config g1_c3] [--reps 50]"""
import os
import statistics
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from mink_amd import _native as nat  # noqa: E402
from mink_amd import workloads  # noqa: E402


def main():
    name = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "g1_c3"
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
    B = 16384
    model = workloads.load_bench_robot(name)
    nm = nat.NativeModel(model)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    q, tg, pt, ct = workloads.bench_batch(name, model, nm, prob, np.random.default_rng(0), B)
    dev = torch.device("cuda", 0)
    to = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    q, tg, pt, ct = to(q), to(tg), to(pt), to(ct)
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
    st = torch.empty((B,), dtype=torch.int32, device=dev)
    for b in (1, 64, 256, 512, 1024, 2048, 3072, 4096, 8192, 16384):
        row = []
        for label, kw in (("default", {}), ("two_waves", {"two_waves": True}), ("two_waves+direct", {"two_waves": True, "direct_qp": True})):
            cs = None if ct is None else (ct[:b] if ct.shape[0] == B else ct)
            for _ in range(3):
                prob.solve(q[:b], tg[:b], pt, cs, dt, damping, out=v[:b], status_out=st[:b], **kw)
            ev = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); prob.solve(q[:b], tg[:b], pt, cs, dt, damping, out=v[:b], status_out=st[:b], **kw); e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            ms = statistics.median(a.elapsed_time(b_) for a, b_ in ev)
            row.append(f"{label}: {prob.last_kernel()} {ms * 1e3:8.1f} us {b / ms / 1e3:7.2f} M/s")
        print(f"B={b:6d}  " + "  |  ".join(row), flush=True)


if __name__ == "__main__":
    main()
