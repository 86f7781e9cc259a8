#!/usr/bin/env python
"""The reference's everyday use — ONE configuration in a control loop (examples/humanoid_g1.py, arm_ur5e.py):
`task.set_target(...)`, `vel = solve_ik(configuration, tasks, dt, solver, damping, limits=limits)`,
`configuration.integrate_inplace(vel, dt)` — through mink_amd's public API with numpy in and out.  Wall time per iteration
and the cProfile breakdown of the Python side.

    python tools/bench_control_loop.py [g1|ur5e] [iterations]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import mink_amd as mink
    from mink_amd import workloads

    robot = sys.argv[1] if len(sys.argv) > 1 else "g1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = np.random.default_rng(0)
    if robot == "g1":
        model = workloads.load_robot("g1")
        cfg = mink.Configuration(model)
        cfg.update_from_keyframe("stand")
        frames = [("left_foot", "site"), ("right_foot", "site"), ("left_palm", "site"), ("right_palm", "site")]
        tasks = [mink.FrameTask(f, t, position_cost=200.0 if "foot" in f else 5.0, orientation_cost=10.0 if "foot" in f else 1.0,
                                lm_damping=1.0) for f, t in frames]
        post = mink.PostureTask(model, cost=1e-1)
        post.set_target_from_configuration(cfg)
        limits = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, {j: np.pi for j in model.jnt_names if model.jnt_type[model.name2id("joint", j)] == 3})]
        dt, damping = 5e-3, 1e-1
    else:
        model = workloads.load_robot("ur5e")
        cfg = mink.Configuration(model)
        cfg.update_from_keyframe("home")
        tasks = [mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)]
        post = mink.PostureTask(model, cost=1e-2)
        post.set_target_from_configuration(cfg)
        limits = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, {j: np.pi for j in model.jnt_names})]
        dt, damping = 2e-3, 1e-3
    base = [cfg.get_transform_frame_to_world(t.frame_name, t.frame_type) for t in tasks]

    def iteration(k):
        for t, b in zip(tasks, base):
            w = b.wxyz_xyz.copy()
            w[4:] += 0.02 * np.array([np.sin(0.01 * k), np.cos(0.013 * k), np.sin(0.007 * k)])
            t.set_target(mink.SE3(w))
        vel = mink.solve_ik(cfg, tasks + [post], dt, "mi355x", damping, limits=limits)
        cfg.integrate_inplace(vel, dt)
        return vel

    for k in range(50):
        iteration(k)
    t0 = time.perf_counter()
    for k in range(n):
        v = iteration(k)
    el = time.perf_counter() - t0
    print("%s: %.1f us per control-loop iteration (%d iterations; |v| = %.3f) — %.0f Hz" % (robot, 1e6 * el / n, n, np.abs(v).max(), n / el))
    pr = cProfile.Profile()
    pr.enable()
    for k in range(500):
        iteration(k)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(22)


if __name__ == "__main__":
    main()
