#!/usr/bin/env python
"""mink's plugin API (tasks/task.py:81-138, limits/limit.py:34-57) on the device path, for a batch of UR5e instances:

  * a Task written from scratch the way a mink user writes one — `compute_error` / `compute_jacobian` with numpy, here "keep the
    tool at a given height" on top of the device's own frame pose and Jacobian;
  * a Task that overrides `compute_qp_objective` — the one method the reference's solve_ik actually calls (solve_ik.py:18-21):
    "stay close to the previous step" as the objective ½·w·‖Δq − Δq_prev‖², i.e. H = w·I, c = −w·Δq_prev, which has no rows of
    its own to speak of (round 5: factored into rows for the device, mink_amd.tasks.objective_to_rows);
  * a Limit written from scratch — a joint-space "keep-out" half-space per instance — next to the built-in limits.

    python examples/batched_plugin_tasks.py --batch 1024
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a source checkout
import mink_amd as mink  # noqa: E402


class ToolHeightTask(mink.Task):
    """e = z(tool) − z*, J = row 2 of the tool's world-aligned translational Jacobian (one row per instance)."""

    def __init__(self, frame_name, frame_type, height, cost, gain=1.0):
        super().__init__(cost=np.array([cost], dtype=np.float64), gain=gain)
        self.frame_name, self.frame_type, self.height = frame_name, frame_type, height

    def compute_error(self, configuration):
        T = configuration.get_transform_frame_to_world(self.frame_name, self.frame_type)   # SE3, batched like the configuration
        return (T.translation()[:, 2] - self.height)[:, None]                    # (B, 1)

    def compute_jacobian(self, configuration):
        R = configuration.get_transform_frame_to_world(self.frame_name, self.frame_type).rotation().as_matrix()   # (B, 3, 3)
        Jb = configuration.get_frame_jacobian(self.frame_name, self.frame_type)  # (B, 6, nv), body frame (configuration.py:112-155)
        return np.einsum("bi,bij->bj", R[:, 2, :], Jb[:, :3, :])[:, None, :]     # world z row of R·J_lin: (B, 1, nv)


class StayCloseTask(mink.Task):
    """½·w·‖Δq − Δq_prev‖² as an objective of its own: H = w·I, c = −w·Δq_prev (per instance)."""

    def __init__(self, nv, weight):
        super().__init__(cost=np.zeros(1))
        self.nv, self.weight, self.dq_prev = nv, weight, None

    def compute_error(self, configuration):            # never called: the reference goes through compute_qp_objective alone
        raise NotImplementedError

    def compute_jacobian(self, configuration):
        raise NotImplementedError

    def compute_qp_objective(self, configuration):
        B = configuration.batch_size
        prev = np.zeros((B, self.nv)) if self.dq_prev is None else self.dq_prev
        return mink.Objective(self.weight * np.eye(self.nv), -self.weight * prev)


class ShoulderKeepOut(mink.Limit):
    """One general half-space per instance: q_shoulder_pan + q_shoulder_lift stays below a ceiling (g·Δq ≤ h)."""

    def __init__(self, model, ceiling, gain=0.9):
        self.nv, self.ceiling, self.gain = model.nv, ceiling, gain

    def compute_qp_inequalities(self, configuration, dt):
        q = configuration.q.reshape(-1, self.nv)
        G = np.zeros((len(q), 1, self.nv)); G[:, 0, 0] = 1.0; G[:, 0, 1] = 1.0
        return mink.Constraint(G=G, h=self.gain * (self.ceiling - (q[:, 0] + q[:, 1]))[:, None])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=60)
    args = ap.parse_args()
    B = args.batch
    rng = np.random.default_rng(0)
    model = mink.load_robot("ur5e")
    home = mink.custom_configuration_vector(model, "home")
    configuration = mink.Configuration(model, np.tile(home, (B, 1)) + rng.normal(scale=0.05, size=(B, model.nq)))

    goal = mink.Configuration(model, configuration.integrate(rng.normal(scale=0.15, size=(B, model.nv)), 1.0))
    target = goal.get_transform_frame_to_world("attachment_site", "site")
    reach = mink.FrameTask("attachment_site", "site", position_cost=[1.0, 1.0, 0.0], orientation_cost=0.5, lm_damping=1.0)
    reach.set_target(target)                                                     # x, y and orientation from the built-in task ...
    height = ToolHeightTask("attachment_site", "site", target.translation()[:, 2], cost=1.0)   # ... z from the user's
    posture = mink.PostureTask(model, cost=1e-2); posture.set_target(home)
    smooth = StayCloseTask(model.nv, weight=0.05)
    tasks = [reach, height, posture, smooth]
    limits = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, {n: np.pi for n in model.jnt_names}),
              ShoulderKeepOut(model, ceiling=float(home[0] + home[1]) + 0.3)]
    dt = 2e-2
    t0 = time.perf_counter()
    for _ in range(args.iters):
        vel = mink.solve_ik(configuration, tasks, dt, "quadprog", 1e-3, limits=limits)
        smooth.dq_prev = vel * dt
        configuration.integrate_inplace(vel, dt)
    el = time.perf_counter() - t0
    err = reach.compute_error(configuration)
    dz = height.compute_error(configuration)[:, 0]
    q = configuration.q.reshape(B, -1)
    exy = np.linalg.norm(err[:, :2], axis=1)
    margin = home[0] + home[1] + 0.3 - q[:, 0] - q[:, 1]
    print(f"{B} instances x {args.iters} iterations with three caller-defined plugins: {el * 1e3:.1f} ms "
          f"({B * args.iters / el / 1e6:.2f} M solves/s incl. the numpy rows); xy error median {np.median(exy):.2e} m, height error "
          f"median {np.median(np.abs(dz)):.2e} m; {int((margin < 1e-6).sum())} instances rest on the keep-out half-space "
          f"(their targets lie behind it), none crosses it (min margin {margin.min():.1e} rad)")


if __name__ == "__main__":
    main()
