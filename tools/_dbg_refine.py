import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from mink_amd import _native as nat, workloads
for config in ("g1_c3", "g1_full"):
    model = workloads.load_robot(workloads.BENCH_CONFIGS[config]["robot"])
    nm = nat.NativeModel(model)
    B = 16384
    prob, dt, damping = workloads.bench_config(config, model, nm, B)
    q, tg, stand_t, com_t = workloads.bench_batch(config, model, nm, prob, np.random.default_rng(0), B)
    stand = stand_t[0] if stand_t is not None else model.qpos0
    v, st, t = prob.solve(q, tg, stand[None, :], com_t, dt, damping, taps=["qp_iters"], wave_kernel=True)
    pv = t["qp_pivots"]
    print(config, prob.last_kernel(), "pivots mean %.2f" % pv.mean(), "pcts 50/90/99/99.9/max", [int(np.percentile(pv, p)) for p in (50, 90, 99, 99.9, 100)],
          "frac>6 %.4f >9 %.4f >12 %.4f >16 %.4f" % tuple((pv > c).mean() for c in (6, 9, 12, 16)))
