#!/usr/bin/env python
"""The IK set-up of mink's examples/arm_ur5e_actuators.py (:20-97) — end-effector FrameTask, PostureTask,
ConfigurationLimit + VelocityLimit, and the callers' loop "solve, integrate, break when the pose error is within
1e-4 / 1e-4 or after max_iters" — for a batch of UR5e instances, each with its own reachable target, written against the
mink-compatible API of this package (`import mink_amd as mink`).

    python examples/batched_arm_ur5e.py --batch 4096

The first part is the reference's loop with a batch dimension (one `solve_ik` + one `integrate_inplace` per iteration,
numpy in and out); the second runs the whole loop — per-instance break included — as ONE launch (`solve_ik_steps` with
thresholds).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a source checkout
import mink_amd as mink  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--max-iters", type=int, default=20)
    args = ap.parse_args()
    B = args.batch
    rng = np.random.default_rng(0)

    model = mink.load_robot("ur5e")                                  # packaged FlatModel of examples/universal_robots_ur5e/scene.xml
    home = mink.custom_configuration_vector(model, "home")
    q0 = np.tile(home, (B, 1)) + rng.normal(scale=0.05, size=(B, model.nq))
    configuration = mink.Configuration(model, q0)

    tasks = [
        end_effector := mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0),
        posture := mink.PostureTask(model, cost=1e-2),
    ]
    limits = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, {name: np.pi for name in model.jnt_names})]
    posture.set_target(home)
    # reachable targets: the end-effector poses of configurations 0.1 rad (rms) away
    goal = mink.Configuration(model, configuration.integrate(rng.normal(scale=0.1, size=(B, model.nv)), 1.0))
    end_effector.set_target(goal.get_transform_frame_to_world("attachment_site", "site"))

    dt, damping, pos_thr, ori_thr = 2e-2, 1e-3, 1e-4, 1e-4
    iters = np.full(B, args.max_iters)
    live = np.ones(B, bool)
    t0 = time.perf_counter()
    for i in range(args.max_iters):
        vel = mink.solve_ik(configuration, tasks, dt, "quadprog", damping, limits=limits)   # any solver name: one backend
        vel[~live] = 0.0                                             # (an instance that has converged stays where it is)
        configuration.integrate_inplace(vel, dt)
        err = end_effector.compute_error(configuration)
        done = live & (np.linalg.norm(err[:, :3], axis=1) <= pos_thr) & (np.linalg.norm(err[:, 3:], axis=1) <= ori_thr)
        iters[done] = i + 1
        live &= ~done
        if not live.any():
            break
    el = time.perf_counter() - t0
    print(f"{B} targets, the callers' loop from the host: {el * 1e3:.1f} ms ({B / el / 1e6:.2f} M targets/s); "
          f"converged {(~live).sum()} of {B}, {iters[~live].mean():.1f} iterations on average")

    configuration.update(q0)
    t0 = time.perf_counter()
    q_final, vel, n_it, converged = mink.solve_ik_steps(configuration, tasks, dt, args.max_iters, damping=damping, limits=limits,
                                                        pos_threshold=pos_thr, ori_threshold=ori_thr)
    el = time.perf_counter() - t0
    print(f"the same loop as ONE launch: {el * 1e3:.2f} ms ({B / el / 1e6:.2f} M targets/s incl. host copies); "
          f"converged {int(converged.sum())} of {B}, {n_it[converged].mean():.1f} iterations on average; "
          f"iteration counts equal to the host loop's for {int((n_it[~live] == iters[~live]).sum())} of {(~live).sum()} converged instances")


if __name__ == "__main__":
    main()
