"""Fixtures from the REAL reference for the robots of the row kernel's two-row build (17 … 32 dofs, floating bases; build
container only; same method as make_golden_small.py — /root/reference/mink on top of oracle/stubs):

    python tests/golden/make_golden_mid.py

  ik_h1_c.npz    Unitree H1 (examples/unitree_h1/scene.xml, free joint + 19 hinges): the tasks of examples/humanoid_h1.py:22-52
                 without the CoM task — pelvis orientation (body frame), feet (pos 200 / ori 10, lm 1), wrists (pos 200 / ori 0,
                 lm 1), PostureTask(1) — ConfigurationLimit + VelocityLimit(π)
  ik_go1_c.npz   Unitree Go1 (examples/unitree_go1/scene.xml, free joint + 12 hinges) with the tasks of
                 examples/quadruped_go1.py:20-40: trunk pose (body frame), four feet (position), PostureTask(1e-5), ConfigurationLimit
  ik_h1_full.npz examples/humanoid_h1.py:22-52 as written: the H1 tasks above + ComTask(200) with a per-instance CoM target
Every eighth instance has its targets 1e-4 away (the small-angle branch of log / jlog).
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ext as mg  # noqa: E402

mink, mujoco = mg.mink, mg.mujoco
EX = "/root/reference/examples/"


def _record(name, m, fts, post, lims, dt, damping, key, rng, n=32, com=None):
    q0 = np.array(m.key_qpos[m.key(key).id])
    post.set_target(q0)
    qb = mg.sample_q(m, rng, n, base_q=q0)

    def set_targets(i, q):
        ct = mink.Configuration(m, mg.perturbed(m, q, rng, 1e-4 if i % 8 == 7 else 0.15))
        for t in fts:
            t.set_target(ct.get_transform_frame_to_world(t.frame_name, t.frame_type))
        out = {"frame_targets": [t.transform_target_to_world.wxyz_xyz for t in fts]}
        if com is not None:
            com.set_target(ct.data.subtree_com[1].copy())             # per-instance CoM target
            out["com_targets"] = [com.target_com]
        return out

    tasks = fts + [post] + ([com] if com is not None else [])
    mg.record(name, m, tasks, lims, dt, damping, qb, set_targets, store_G=4, extra={"posture_target": q0.copy()})


def h1(rng, full=False):
    m = mujoco.MjModel.from_xml_path(EX + "unitree_h1/scene.xml")
    fts = [mink.FrameTask("pelvis", "body", position_cost=0.0, orientation_cost=10.0)]
    fts += [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0) for s in ("right_foot", "left_foot")]
    fts += [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=0.0, lm_damping=1.0) for s in ("right_wrist", "left_wrist")]
    vel = {m.jnt_names[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] == 3}
    _record("h1_full" if full else "h1_c", m, fts, mink.PostureTask(m, cost=1.0), [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)],
            5e-3, 1e-1, "stand", rng, com=mink.ComTask(cost=200.0) if full else None)


def go1(rng):
    m = mujoco.MjModel.from_xml_path(EX + "unitree_go1/scene.xml")
    fts = [mink.FrameTask("trunk", "body", position_cost=1.0, orientation_cost=1.0)]
    fts += [mink.FrameTask(s, "site", position_cost=1.0, orientation_cost=0.0) for s in ("FL", "FR", "RR", "RL")]
    _record("go1_c", m, fts, mink.PostureTask(m, cost=1e-5), [mink.ConfigurationLimit(m)], 2e-3, 1e-5, "home", rng)


if __name__ == "__main__":
    h1(np.random.default_rng(41))
    go1(np.random.default_rng(42))
    h1(np.random.default_rng(43), full=True)
