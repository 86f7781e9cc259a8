#!/usr/bin/env python
"""Time the REAL reference — `mink.solve_ik` from /root/reference, its own Python, one Configuration per call
site — on the BASELINE configs.  Build container only (the reference cannot travel to the GPU box); the third-party
wheels it needs (mujoco, qpsolvers/quadprog) are absent and replaced by oracle/stubs, i.e. MuJoCo's C kernels and
quadprog's Fortran-derived C are replaced by numpy restatements — so this is the reference's Python dispatch cost plus
a SLOWER numeric back end than the real wheels: an upper bound on the per-solve time of the real stack's Python part,
not a measurement of the real stack.  Writes profiles/r02_cpu_reference_mink.json (read by bench.py as
`cpu_reference_recorded`).

    python tools/time_reference_mink.py [seconds per config]
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "oracle", "stubs"), "/root/reference", REPO, os.path.join(REPO, "tests", "golden")]

import mujoco  # noqa: E402  (the stub)
import mink  # noqa: E402  (the real reference)
from make_golden import ROBOTS  # noqa: E402


def configs():
    m = mujoco.MjModel.from_xml_path(ROBOTS["ur5e"])
    ft = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    pt = mink.PostureTask(m, cost=1e-2)
    pt.set_target(m.key_qpos[m.key("home").id])
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: np.pi for n in m.jnt_names})]
    yield "ur5e_c2", m, [ft, pt], lims, [ft], 2e-3, 1e-3, "ik_ur5e_c2.npz"

    m = mujoco.MjModel.from_xml_path(ROBOTS["g1"])
    stand = m.key_qpos[m.key("stand").id]
    feet = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0) for s in ("left_foot", "right_foot")]
    hands = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=0.0, lm_damping=1.0) for s in ("left_palm", "right_palm")]
    pt = mink.PostureTask(m, cost=1.0)
    pt.set_target(stand)
    vel = {m.jnt_names[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] != 0}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)]
    yield "g1_c3", m, feet + hands + [pt], lims, feet + hands, 5e-3, 1e-1, "ik_g1_c3.npz"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    out = {}
    for name, m, tasks, lims, fts, dt, damping, fixture in configs():
        d = np.load(os.path.join(REPO, "tests", "golden", fixture))
        n, t0, k = 0, time.perf_counter(), 0
        cfg = mink.Configuration(m)
        worst = 0.0
        while time.perf_counter() - t0 < budget:
            i = k % len(d["q"])
            for t, tg in zip(fts, d["frame_targets"][i]):
                t.set_target(mink.SE3(wxyz_xyz=tg))
            cfg.update(d["q"][i])
            v = mink.solve_ik(cfg, tasks, dt, "quadprog", damping, limits=lims)
            worst = max(worst, float(np.abs(v - d["v"][i]).max()))
            n += 1; k += 1
        el = time.perf_counter() - t0
        out[name] = {"value": n / el, "unit": "solves/s", "cores": 1, "kind": "reference",
                     "sample": f"{n} calls of the real mink.solve_ik (/root/reference, Python) on the {fixture} inputs in "
                               f"{el:.1f} s, one thread of the BUILD container (not the GPU box); mujoco/quadprog wheels "
                               f"absent: numpy stand-ins (oracle/stubs) do their arithmetic, so the real stack is faster "
                               f"per call than this; max |v - fixture| = {worst:.1e}"}
        print(name, out[name])
    with open(os.path.join(REPO, "profiles", "r02_cpu_reference_mink.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
