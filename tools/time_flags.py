#!/usr/bin/env python
"""Kernel ms of a bench config under solve flags (variant pinning): python tools/time_flags.py g1_c3 direct_qp,two_waves"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from mink_amd import _native as nat, workloads  # noqa: E402

name = sys.argv[1]
kw = {k: True for k in (sys.argv[2] if len(sys.argv) > 2 else "").split(",") if k}
cfg = workloads.BENCH_CONFIGS[name]
B = cfg["batch"]
model = workloads.load_bench_robot(name)
nm = nat.NativeModel(model)
prob, dt, damping = workloads.bench_config(name, model, nm, B)
rng = np.random.default_rng(2000)
q, tg, pt, ct = workloads.bench_batch(name, model, nm, prob, rng, B)
dev = torch.device("cuda", 0)
to = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
args = (to(q), to(tg), to(pt if prob.n_posture else None), to(ct), dt, damping)
ev = []
for i in range(25):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); prob.solve(*args, **kw); e1.record()
    if i >= 5:
        ev.append((e0, e1))
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
print(f"{name} {sorted(kw)} {prob.last_kernel()} kernel ms mean {np.mean(ms):.4f} median {np.median(ms):.4f}  ({B / np.mean(ms) / 1e3:.1f} M solves/s)")
