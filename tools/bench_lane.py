#!/usr/bin/env python
"""Throughput of the lane-per-problem kernel on one of the reference's small arms (device-resident inputs).

    python tools/bench_lane.py [scene=kuka_iiwa_14__scene] [batch=262144] [until]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from mink_amd import _native as nat
    from mink_amd import api_specs, workloads
    from mink_amd.flatmodel import FlatModel

    scene = sys.argv[1] if len(sys.argv) > 1 else "kuka_iiwa_14__scene"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    until = len(sys.argv) > 3
    m = FlatModel.load(os.path.join(REPO, "tests", "golden", "models", "all", scene + ".json"))
    nm = nat.NativeModel(m)
    sites = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0]
    ft = {"frame_type": "site", "frame_id": sites[-1], "cost": [1.0] * 6, "gain": 1.0, "lm_damping": 1.0}
    vel = {m.jnt_names[j]: np.pi for j in range(m.njnt)}
    prob = nat.NativeProblem(nm, frame_tasks=[ft], posture_tasks=[{"cost": 1e-2}],
                             configuration_limits=[api_specs.configuration_limit_desc(m)],
                             velocity_limits=[api_specs.velocity_limit_desc(m, vel)], max_batch=B)
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(0), B, base_q=m.qpos0, sigma=0.05 if until else 0.15)
    dev = torch.device("cuda", 0)
    qd, tgd, pt = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev), torch.from_numpy(m.qpos0[None, :].copy()).to(dev)
    kw = dict(n_steps=20, until=(1e-3, 1e-2)) if until else {}
    for wave in (False, True):
        for _ in range(2):
            res = prob.solve(qd, tgd, pt, None, 2e-2, 1e-3, wave_kernel=wave, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            res = prob.solve(qd, tgd, pt, None, 2e-2, 1e-3, wave_kernel=wave, **kw)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        extra = ""
        if until:
            extra = "  mean iterations %.2f, converged %.3f" % (float(res[3].float().mean()), float(res[4].float().mean()))
        print("%-26s B=%8d  %-24s %8.3f ms  %9.2f M %s/s%s" % (scene, B, prob.last_kernel(), el * 1e3, B / el / 1e6,
                                                              "targets" if until else "solves", extra))


if __name__ == "__main__":
    main()
