"""Fixtures from the REAL reference for the robots of the row-per-problem kernel's sixteen-register build (build container
only; same method as make_golden_ext.py — /root/reference/mink on top of oracle/stubs):

    python tests/golden/make_golden_small.py

  ik_leap_c.npz     LEAP hand (examples/leap_hand/scene_right.xml, 16 hinge dofs, a tree of four fingers): FrameTasks on the
                    four fingertip sites (position 1, orientation 0, lm 1) + PostureTask(1e-2), ConfigurationLimit +
                    VelocityLimit(π)
  ik_kinova_c.npz   Tidybot's mobile Kinova (examples/stanford_tidybot/scene_mobile_kinova.xml, 10 dofs, two slides and a
                    hinge on the base body) with the tasks of examples/mobile_kinova.py:46-63 — end-effector FrameTask,
                    PostureTask with a cost on the base yaw only, DampingTask that holds the base — ConfigurationLimit +
                    VelocityLimit(π / 0.5 m/s on the slides)
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


def leap(rng):
    m = mujoco.MjModel.from_xml_path(EX + "leap_hand/scene_right.xml")
    tips = ["tip_1", "tip_2", "tip_3", "th_tip"]
    fts = [mink.FrameTask(s, "site", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0) for s in tips]
    post = mink.PostureTask(m, cost=1e-2)
    q0 = np.array(m.qpos0)
    post.set_target(q0)
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {m.jnt_names[j]: np.pi for j in range(m.njnt)})]
    qb = mg.sample_q(m, rng, 32, base_q=q0)

    def set_targets(i, q):
        ct = mink.Configuration(m, mg.perturbed(m, q, rng, 1e-4 if i % 8 == 7 else 0.15))
        for t in fts:
            t.set_target(ct.get_transform_frame_to_world(t.frame_name, "site"))
        return {"frame_targets": [t.transform_target_to_world.wxyz_xyz for t in fts]}

    mg.record("leap_c", m, fts + [post], lims, 5e-3, 1e-3, qb, set_targets, store_G=4, extra={"posture_target": q0.copy()})


def kinova(rng):
    m = mujoco.MjModel.from_xml_path(EX + "stanford_tidybot/scene_mobile_kinova.xml")
    ee = mink.FrameTask("pinch_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    pc = np.zeros(m.nv); pc[2] = 1e-3
    post = mink.PostureTask(m, cost=pc)
    dc = np.zeros(m.nv); dc[:2] = 100.0; dc[2] = 1e-3
    damp = mink.DampingTask(m, dc)
    q0 = m.key_qpos[m.key("home").id] if m.nkey else np.array(m.qpos0)
    post.set_target(q0)
    vel = {m.jnt_names[j]: (0.5 if m.jnt_type[j] == 2 else np.pi) for j in range(m.njnt)}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)]
    qb = mg.sample_q(m, rng, 32, base_q=q0)

    def set_targets(i, q):
        ct = mink.Configuration(m, mg.perturbed(m, q, rng, 1e-4 if i % 8 == 7 else 0.15))
        ee.set_target(ct.get_transform_frame_to_world("pinch_site", "site"))
        return {"frame_targets": [ee.transform_target_to_world.wxyz_xyz], "posture_targets": [post.target_q, damp.target_q]}

    mg.record("kinova_c", m, [ee, post, damp], lims, 1e-2, 1e-3, qb, set_targets, store_G=4)


if __name__ == "__main__":
    leap(np.random.default_rng(31))
    kinova(np.random.default_rng(32))
