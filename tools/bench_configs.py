#!/usr/bin/env python
"""Throughput of every BASELINE config on one GPU, device-resident inputs.  Not the headline
bench (bench.py is): these are the parity-test cases, timed for DESIGN.md's table.

    python tools/bench_configs.py [reps]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch

    import native_configs as nc
    from mink_amd import _native as nat
    from mink_amd import workloads

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    rows = []
    for name, B, key in (("ur5e_c2", 4096, "home"), ("ur5e_c2", 65536, "home"), ("g1_c3", 65536, "stand"),
                         ("g1_full", 65536, "stand"), ("shadow_c4", 16384, "grasp hard"), ("shadow_c4", 65536, "grasp hard")):
        robot = nc.ROBOT_OF[name]
        model = workloads.load_robot(robot)
        nm = nat.NativeModel(model)
        prob, dt, damping = nc.build(name, nm, B)
        base = model.key_qpos[model.name2id("key", key)]
        q, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(5), B, base_q=base)
        if robot == "shadow_left":
            q[::2] = 0.5 * (q[::2] + base)
        com = None
        if prob.n_com:
            _, _, t = prob.solve(q, tg, base[None, :], np.zeros((1, 3)), dt, damping, taps=["subtree_com"], solve_qp=False)
            com = torch.from_numpy(t["subtree_com"][:, None, :] + 0.01).to(dev)
        qd, tgd = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev)
        pt = torch.from_numpy(base[None, :].copy()).to(dev)
        v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
        st = torch.empty((B,), dtype=torch.int32, device=dev)
        for _ in range(2):
            prob.solve(qd, tgd, pt, com, dt, damping, out=v, status_out=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            prob.solve(qd, tgd, pt, com, dt, damping, out=v, status_out=st)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        bad = int(((st.cpu().numpy() & ~1) != 0).sum())
        rows.append((name, B, prob.last_kernel(), el * 1e3, B / el, bad))
        print("%-10s B=%6d  %-28s %8.3f ms  %8.2f M solves/s  failed %d" % rows[-1][:3] + (), end="") if False else None
        print("%-10s B=%6d  %-28s %8.3f ms  %8.2f M solves/s  failed %d" % (name, B, prob.last_kernel(), el * 1e3, B / el / 1e6, bad))


if __name__ == "__main__":
    main()
