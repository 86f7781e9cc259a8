#!/usr/bin/env python
"""Throughput of one of the reference's example robots (tests/golden/models/all/*.json) with the task set of
tests/test_gpu_all_robots.py, device-resident, default kernel choice vs the 2-waves register map.

    python tools/bench_robot.py unitree_h1__scene [batch]
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
    from mink_amd.flatmodel import FlatModel

    name = sys.argv[1] if len(sys.argv) > 1 else "unitree_h1__scene"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    m = FlatModel.load(os.path.join(REPO, "tests", "golden", "models", "all", name + ".json"))
    nm = nat.NativeModel(m)
    sites = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0][-2:]
    frames = [("site", i) for i in sites] if sites else [("body", int(b)) for b in np.argsort(m.body_depth)[-2:]]
    fts = [{"frame_type": ft, "frame_id": fid, "cost": [1.0] * 3 + ([0.5] * 3 if k == 0 else [0.0] * 3), "gain": 1.0,
            "lm_damping": 1.0 if k == 0 else 0.0} for k, (ft, fid) in enumerate(frames)]
    vidx = [int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] in (2, 3)]
    vlim = np.where([m.jnt_type[m.dof_jntid[d]] == 2 for d in vidx], 0.5, np.pi)
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                             velocity_limits=[{"indices": vidx, "limit": vlim}], max_batch=B)
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(1), B, base_q=m.qpos0)
    dev = torch.device("cuda", 0)
    qd, tgd = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev)
    pt = torch.from_numpy(m.qpos0[None, :].copy()).to(dev)
    res = {}
    for two in (True, False):
        v = torch.empty((B, m.nv), dtype=torch.float64, device=dev)
        st = torch.empty((B,), dtype=torch.int32, device=dev)
        for _ in range(3):
            prob.solve(qd, tgd, pt, None, 1e-2, 1e-3, out=v, status_out=st, two_waves=two)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prob.solve(qd, tgd, pt, None, 1e-2, 1e-3, out=v, status_out=st, two_waves=two)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        res[two] = v.cpu().numpy()
        print("%-28s nv %2d  %-12s %-30s %8.3f ms  %8.2f M solves/s  failed %d" %
              (name, m.nv, "plain map" if two else "default", prob.last_kernel(), ms, B / ms / 1e3,
               int(((st.cpu().numpy() & ~1) != 0).sum())), flush=True)
    print("   max |v(default) - v(2 waves)| = %.3e" % np.abs(res[True] - res[False]).max())


if __name__ == "__main__":
    main()
