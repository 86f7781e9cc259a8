#!/usr/bin/env python
"""The callers' closed loop — solve_ik, integrate, solve again on the same batch — with single device-resident solves:
ms per step along the loop, cold against MKH_FLAG_WARM_START (the active-set phase starts where the previous solve ended).

    python tools/bench_closed_loop.py [config=g1_c3] [batch] [steps=24]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch

    import native_configs as nc
    from mink_amd import _native as nat
    from mink_amd import workloads

    config = sys.argv[1] if len(sys.argv) > 1 else "g1_c3"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    model = workloads.load_robot(nc.ROBOT_OF[config])
    nm = nat.NativeModel(model)
    dev = torch.device("cuda", 0)
    key = {"g1": "stand", "ur5e": "home"}.get(nc.ROBOT_OF[config], None)
    base = model.key_qpos[model.name2id("key", key)] if key else model.qpos0
    for warm in (False, True):
        prob, dt, damping = nc.build(config, nm, B)
        q, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(5), B, base_q=base)
        qd, tgd = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev)
        pt = torch.from_numpy(base[None, :].copy()).to(dev)
        v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
        st = torch.empty((B,), dtype=torch.int32, device=dev)
        ms = []
        for s in range(steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prob.solve(qd, tgd, pt, None, dt, damping, out=v, status_out=st, warm_start=warm)
            e1.record()
            qd = nm.integrate(qd, v, dt)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        print("%-8s B=%d %-6s %-30s ms per solve at steps 0 / 4 / 10 / %d: %.3f / %.3f / %.3f / %.3f   total %.1f ms  failed %d" %
              (config, B, "warm" if warm else "cold", prob.last_kernel(), steps - 1, ms[0], ms[4], ms[10], ms[-1], sum(ms),
               int(((st.cpu().numpy() & ~1) != 0).sum())), flush=True)


if __name__ == "__main__":
    main()
