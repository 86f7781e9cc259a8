#!/usr/bin/env python
"""Robots of the reference's examples between an arm and a humanoid — LEAP hand (16 dofs), mobile Kinova (10), iiwa (7) — at
reinforcement-learning batch sizes: fingertip / end-effector FrameTasks + posture + configuration and velocity limits, default
dispatch against the wavefront kernel (device-resident inputs, HIP events over 100 launches).

    python tools/bench_mid_robots.py
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

    dev = torch.device("cuda", 0)
    for scene in ("leap_hand__scene_right", "stanford_tidybot__scene_mobile_kinova", "kuka_iiwa_14__scene"):
        m = FlatModel.load(os.path.join(REPO, "tests", "golden", "models", "all", scene + ".json"))
        nm = nat.NativeModel(m)
        sites = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0]
        tips = sites[-4:] if len(sites) >= 4 else sites[-1:]
        print(scene, "nv", m.nv, "frames", [m.site_names[i] for i in tips])
        for B in (1024, 4096, 16384, 65536):
            fts = [{"frame_type": "site", "frame_id": i, "cost": [1.0, 1.0, 1.0, 0.0, 0.0, 0.0], "gain": 1.0, "lm_damping": 1.0} for i in tips]
            vidx = [int(m.jnt_dofadr[j]) for j in range(m.njnt)]
            prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                                     velocity_limits=[{"indices": vidx, "limit": np.full(len(vidx), np.pi)}], max_batch=B)
            q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(1), B, base_q=m.qpos0)
            qd, tgd, pt = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (q, tg, m.qpos0[None, :].copy())]
            v = torch.empty((B, m.nv), dtype=torch.float64, device=dev)
            st = torch.empty((B,), dtype=torch.int32, device=dev)
            cells = []
            for kw in ({}, {"wave_kernel": True}):
                for _ in range(10):
                    prob.solve(qd, tgd, pt, None, 5e-3, 1e-3, out=v, status_out=st, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    prob.solve(qd, tgd, pt, None, 5e-3, 1e-3, out=v, status_out=st, **kw)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 100
                bad = int(((st.cpu().numpy() & ~1) != 0).sum())
                cells.append("%-26s %7.1f us %7.1f M/s%s" % (prob.last_kernel(), ms * 1e3, B / ms / 1e3, " FAILED %d" % bad if bad else ""))
            print("   B=%6d  %s | %s" % (B, cells[0], cells[1]))


if __name__ == "__main__":
    main()
