"""Fixtures from the REAL reference for the robots of the row kernel's two-row build (17 … 32 dofs, floating bases; build
container only; same method as make_golden_small.py — /root/reference/mink on top of oracle/stubs):

    python tests/golden/make_golden_mid.py

  ik_h1_c.npz    Unitree H1 (examples/unitree_h1/scene.xml, free joint + 19 hinges): the tasks of examples/humanoid_h1.py:22-52
                 without the CoM task — pelvis orientation (body frame), feet (pos 200 / ori 10, lm 1), wrists (pos 200 / ori 0,
                 lm 1), PostureTask(1) — ConfigurationLimit + VelocityLimit(π)
  ik_go1_c.npz   Unitree Go1 (examples/unitree_go1/scene.xml, free joint + 12 hinges) with the tasks of
                 examples/quadruped_go1.py:20-40: trunk pose (body frame), four feet (position), PostureTask(1e-5), ConfigurationLimit
  ik_h1_full.npz examples/humanoid_h1.py:22-52 as written: the H1 tasks above + ComTask(200) with a per-instance CoM target
  ik_arm_hand.npz the task set of examples/arm_hand_iiwa_allegro.py:62-94 — FrameTask on the arm's attachment site (pos 1, ori 1, lm 1),
                 PostureTask(5e-2), one RelativeFrameTask per fingertip site measured in the PALM body (pos 1, ori 0, lm 1),
                 ConfigurationLimit — on a 7-dof arm carrying a 16-dof four-finger hand written out below (the example composes
                 its model with dm_control, which is not here; ARM_HAND_XML has the same structure: 23 hinges, fingertip sites,
                 a palm body on the arm's last link); the model is saved as models/arm_hand.json
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


def _arm_hand_xml():
    """7 hinges in series (alternating axes, iiwa-like link lengths), a palm, four fingers of four hinges each."""
    arm = ""
    axes = ["0 0 1", "0 1 0", "0 0 1", "0 -1 0", "0 0 1", "0 1 0", "0 0 1"]
    lens = [0.1575, 0.2025, 0.2045, 0.2155, 0.1845, 0.2155, 0.081]
    close = ""
    for i, (ax, ln) in enumerate(zip(axes, lens)):
        arm += f'<body name="link{i + 1}" pos="0 0 {ln}"><inertial pos="0 0 0.05" mass="{3.0 - 0.3 * i:.2f}" diaginertia="0.01 0.01 0.005"/>' \
               f'<joint name="joint{i + 1}" type="hinge" axis="{ax}" range="-2.9 2.9"/><geom type="capsule" size="0.04 0.05" contype="0" conaffinity="0"/>'
        close += "</body>"
    fingers = ""
    for k, (name, y) in enumerate((("ff", 0.045), ("mf", 0.0), ("rf", -0.045))):
        fingers += f'<body name="{name}_base" pos="0 {y} 0.095" quat="1 0 {0.04 * (k - 1)} 0"><inertial pos="0 0 0.01" mass="0.03" diaginertia="1e-5 1e-5 1e-5"/>' \
                   f'<joint name="{name}j0" type="hinge" axis="0 0 1" range="-0.47 0.47"/>' \
                   f'<body name="{name}_proximal" pos="0 0 0.0164"><inertial pos="0 0 0.027" mass="0.065" diaginertia="1e-5 1e-5 1e-5"/><joint name="{name}j1" type="hinge" axis="0 1 0" range="-0.196 1.61"/>' \
                   f'<body name="{name}_medial" pos="0 0 0.054"><inertial pos="0 0 0.02" mass="0.0355" diaginertia="1e-5 1e-5 1e-5"/><joint name="{name}j2" type="hinge" axis="0 1 0" range="-0.174 1.709"/>' \
                   f'<body name="{name}_distal" pos="0 0 0.0384"><inertial pos="0 0 0.013" mass="0.0096" diaginertia="1e-6 1e-6 1e-6"/><joint name="{name}j3" type="hinge" axis="0 1 0" range="-0.227 1.618"/>' \
                   f'<site name="{name}_tip" pos="0 0 0.0267"/></body></body></body></body>'
    thumb = '<body name="th_base" pos="-0.0182 0.019 0.045" quat="0.477714 -0.521334 -0.521334 -0.477714"><inertial pos="0 0 0.01" mass="0.0176" diaginertia="1e-5 1e-5 1e-5"/>' \
            '<joint name="thj0" type="hinge" axis="-1 0 0" range="0.263 1.396"/>' \
            '<body name="th_proximal" pos="-0.027 0.005 0.0399"><inertial pos="0 0 0.008" mass="0.0119" diaginertia="1e-6 1e-6 1e-6"/><joint name="thj1" type="hinge" axis="0 0 1" range="-0.105 1.163"/>' \
            '<body name="th_medial" pos="0 0 0.0177"><inertial pos="0 0 0.02" mass="0.038" diaginertia="1e-5 1e-5 1e-5"/><joint name="thj2" type="hinge" axis="0 1 0" range="-0.189 1.644"/>' \
            '<body name="th_distal" pos="0 0 0.0514"><inertial pos="0 0 0.02" mass="0.0388" diaginertia="1e-5 1e-5 1e-5"/><joint name="thj3" type="hinge" axis="0 1 0" range="-0.162 1.719"/>' \
            '<site name="th_tip" pos="0 0 0.0423"/></body></body></body></body>'
    hand = f'<site name="attachment_site" pos="0 0 0.045"/><body name="palm" pos="0 0 0.14"><inertial pos="0 0 0.0475" mass="0.4154" diaginertia="1e-4 1e-4 1e-4"/>{fingers}{thumb}</body>'
    key = "0 0.5 0 -1.2 0 0.9 0  0 0.7 0.7 0.5  0 0.7 0.7 0.5  0 0.7 0.7 0.5  0.9 0.4 0.5 0.6"
    return f'<mujoco model="arm_hand"><compiler angle="radian" autolimits="true"/><worldbody>{arm}{hand}{close}</worldbody>' \
           f'<keyframe><key name="home" qpos="{key}"/></keyframe></mujoco>'


def arm_hand(rng):
    xml = _arm_hand_xml()
    m = mujoco.MjModel.from_xml_string(xml)
    m.save(os.path.join(HERE, "models", "arm_hand.json"))
    tips = ["ff_tip", "mf_tip", "rf_tip", "th_tip"]
    ee = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    post = mink.PostureTask(m, cost=5e-2)
    fingers = [mink.RelativeFrameTask(t, "site", "palm", "body", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0) for t in tips]
    q0 = np.array(m.key_qpos[m.key("home").id])
    post.set_target(q0)
    qb = mg.sample_q(m, rng, 32, base_q=q0)

    def set_targets(i, q):
        ct = mink.Configuration(m, mg.perturbed(m, q, rng, 1e-4 if i % 8 == 7 else 0.15))
        ee.set_target(ct.get_transform_frame_to_world("attachment_site", "site"))
        for t, f in zip(tips, fingers):
            f.set_target(ct.get_transform(t, "site", "palm", "body"))
        return {"frame_targets": [ee.transform_target_to_world.wxyz_xyz] + [f.transform_target_to_root.wxyz_xyz for f in fingers]}

    mg.record("arm_hand", m, [ee, post] + fingers, [mink.ConfigurationLimit(m)], 1e-2, 1e-3, qb, set_targets, store_G=4,
              extra={"posture_target": q0.copy()})


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
    arm_hand(np.random.default_rng(44))
