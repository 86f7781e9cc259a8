#!/usr/bin/env python
"""The IK set-up of mink's examples/humanoid_g1.py (:13-94) — pelvis orientation, posture, CoM, feet and palm
frame tasks with configuration limits — for a whole batch of G1 instances on one MI355X, written against the
mink-compatible API of this package (`import mink_amd as mink`).

    python examples/batched_humanoid_g1.py --batch 4096 --steps 50

Each instance tracks its own random reachable targets; the loop is the reference's closed loop
(solve_ik → integrate_inplace), and the last lines run the same loop fused on the device (solve_ik_steps).
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
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    B = args.batch
    rng = np.random.default_rng(0)

    model = mink.load_robot("g1")                                   # packaged FlatModel of examples/unitree_g1/scene.xml
    stand = mink.custom_configuration_vector(model, "stand")
    configuration = mink.Configuration(model, np.tile(stand, (B, 1)))

    feet, hands = ["right_foot", "left_foot"], ["right_palm", "left_palm"]
    tasks = [
        pelvis := mink.FrameTask("pelvis", "body", position_cost=0.0, orientation_cost=10.0),
        posture := mink.PostureTask(model, cost=1.0),
        com := mink.ComTask(cost=200.0),
    ]
    feet_tasks = [mink.FrameTask(f, "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0) for f in feet]
    hand_tasks = [mink.FrameTask(h, "site", position_cost=200.0, orientation_cost=0.0, lm_damping=1.0) for h in hands]
    tasks += feet_tasks + hand_tasks
    limits = [mink.ConfigurationLimit(model)]

    # targets: current poses, hands and CoM displaced per instance
    posture.set_target_from_configuration(configuration)
    pelvis.set_target_from_configuration(configuration)
    for t in feet_tasks:
        t.set_target_from_configuration(configuration)
    for t in hand_tasks:
        T = configuration.get_transform_frame_to_world(t.frame_name, t.frame_type).wxyz_xyz.copy()
        T[:, 4:] += rng.normal(scale=0.08, size=(B, 3))
        t.set_target(mink.SE3(T))
    com.set_target(configuration.subtree_com() + rng.normal(scale=0.02, size=(B, 3)))

    dt, damping = 5e-3, 1e-1
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vel = mink.solve_ik(configuration, tasks, dt, "quadprog", damping, limits=limits)   # any solver name: one backend
        configuration.integrate_inplace(vel, dt)
    el = time.perf_counter() - t0
    err = np.linalg.norm(np.stack([t.compute_error(configuration)[:, :3] for t in hand_tasks]), axis=-1)
    print(f"{B} instances x {args.steps} steps in {el * 1e3:.1f} ms ({B * args.steps / el / 1e6:.2f} M solves/s incl. host copies); "
          f"palm position error: mean {err.mean() * 1e3:.2f} mm, max {err.max() * 1e3:.2f} mm")

    # the same closed loop in ONE launch (q stays on the device between the steps)
    q_loop = configuration.q_batch.copy()
    configuration.update(np.tile(stand, (B, 1)))
    t0 = time.perf_counter()
    q_final, vel = mink.solve_ik_steps(configuration, tasks, dt, args.steps, damping=damping, limits=limits)
    el = time.perf_counter() - t0
    dq = np.abs(q_final - q_loop).max(axis=1)
    # (the two loops run different kernel variants, i.e. different rounding; an instance whose active set is about to
    # change amplifies that from step to step, the typical instance does not)
    print(f"fused on the device: {el * 1e3:.1f} ms ({B * args.steps / el / 1e6:.2f} M solves/s); "
          f"|q - q_loop| per instance: median {np.median(dq):.1e}, max {dq.max():.1e}")


if __name__ == "__main__":
    main()
