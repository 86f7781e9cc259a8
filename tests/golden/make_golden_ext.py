"""Round-2 extension of make_golden.py: fixtures for the features whose oracle side was only
property-checked in round 1.  Same method: the REAL reference (``/root/reference/mink``) runs on top of
oracle/stubs (the absent ``mujoco`` / ``qpsolvers`` wheels), build container only:

    python tests/golden/make_golden_ext.py

Outputs (np.savez_compressed, next to this file):
  ik_g1_ext.npz      G1: two mink.RelativeFrameTask (moving roots: site↔site and site↔body), a body-frame
                     FrameTask, mink.DampingTask, PostureTask with PER-INSTANCE targets, ComTask with PER-INSTANCE
                     targets, ConfigurationLimit + VelocityLimit                     (tasks/relative_frame_task.py,
                     tasks/damping_task.py; MKH_FLAG_POSTURE_BATCHED / MKH_FLAG_COM_BATCHED on the device)
  ik_ur5e_coll.npz   UR5e, the collision set-up of examples/arm_ur5e.py:20-47: FrameTask on a site + FrameTask on a
                     GEOM frame (wrist_2_link, quat "1 1 0 0" in the MJCF) + ConfigurationLimit +
                     CollisionAvoidanceLimit(wrist_3_link vs floor, wall) + VelocityLimit; capsule–plane, capsule–box
  ik_ballslide.npz   inline MJCF with ball + hinge + slide joints (cf. tests/test_velocity_limit.py:65-89): FrameTask +
                     PostureTask (ball-joint error), VelocityLimit with a 3-vector for the ball joint, ConfigurationLimit
  models/ballslide.json   FlatModel of that MJCF (tests/golden/ballslide.xml is the source text)
  ik_balllimit.npz   the same chain with a LIMITED ball joint (tests/golden/balllimit.xml) under the default
                     ConfigurationLimit: the reference's quaternion-slot arithmetic for limited ball joints
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(REPO, "oracle", "stubs"), "/root/reference", REPO, HERE]

import mujoco  # noqa: E402  (the stub)
import mink  # noqa: E402  (the real reference)

from make_golden import ROBOTS, perturbed, sample_q  # noqa: E402


def record(name, model, tasks, limits, dt, damping, q_batch, set_targets, store_G=4, extra=None):
    """For every q: set_targets(i, q) fixes every task target (returns a dict of per-instance target arrays), then
    the real mink builds and solves; everything the reference exposes is recorded."""
    rec = {k: [] for k in ("q", "v", "H", "c", "h", "G", "task_e", "task_J")}
    tg_rec = {}
    for i, q in enumerate(q_batch):
        for k, val in set_targets(i, q).items():
            tg_rec.setdefault(k, []).append(np.array(val))
        cfg = mink.Configuration(model, q)
        problem = mink.build_ik(cfg, tasks, dt, damping, limits)
        v = mink.solve_ik(cfg, tasks, dt, "quadprog", damping, limits=limits)
        rec["q"].append(q.copy()); rec["v"].append(v); rec["H"].append(problem.P); rec["c"].append(problem.q)
        rec["h"].append(problem.h if problem.h is not None else np.zeros(0))
        if i < store_G and problem.G is not None:
            rec["G"].append(problem.G)
        rec["task_e"].append(np.concatenate([t.compute_error(cfg) for t in tasks]))
        if i < store_G:
            rec["task_J"].append(np.vstack([t.compute_jacobian(cfg) for t in tasks]))
    out = {k: np.array(v) for k, v in rec.items() if len(v)}
    out.update({k: np.array(v) for k, v in tg_rec.items()})
    out["dt"] = np.array(dt); out["damping"] = np.array(damping)
    out.update(extra or {})
    np.savez_compressed(os.path.join(HERE, f"ik_{name}.npz"), **out)
    print(f"ik_{name}.npz", {k: v.shape for k, v in out.items()}, "max|v|", float(np.abs(out["v"]).max()))
    return out


def g1_ext(rng):
    m = mujoco.MjModel.from_xml_path(ROBOTS["g1"])
    stand = m.key_qpos[m.key("stand").id]
    rel_hands = mink.RelativeFrameTask("left_palm", "site", "right_palm", "site", position_cost=100.0,
                                       orientation_cost=5.0, gain=0.8, lm_damping=0.5)
    rel_foot = mink.RelativeFrameTask("left_foot", "site", "pelvis", "body", position_cost=[50.0, 80.0, 120.0],
                                      orientation_cost=0.0, lm_damping=1.0)
    torso = mink.FrameTask("torso_link", "body", position_cost=0.0, orientation_cost=4.0)
    rfoot = mink.FrameTask("right_foot", "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0)
    damp = mink.DampingTask(m, cost=0.3)
    post = mink.PostureTask(m, cost=np.linspace(0.5, 1.5, m.nv))
    com = mink.ComTask(cost=[200.0, 200.0, 50.0], gain=0.9)
    tasks = [rel_hands, damp, rel_foot, post, torso, com, rfoot]
    vel = {m.jnt_names[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] != 0}
    lims = [mink.ConfigurationLimit(m, gain=0.9, min_distance_from_limits=0.01), mink.VelocityLimit(m, vel)]
    qb = sample_q(m, rng, 16, base_q=stand)
    # keep the samples inside the tightened configuration limits (min_distance_from_limits = 0.01)
    for j in range(m.njnt):
        if m.jnt_type[j] != 0 and m.jnt_limited[j]:
            a = m.jnt_qposadr[j]
            qb[:, a] = np.clip(qb[:, a], m.jnt_range[j][0] + 0.011, m.jnt_range[j][1] - 0.011)

    def set_targets(i, q):
        sig = 1e-4 if (i % 8 == 7) else 0.15
        ct = mink.Configuration(m, perturbed(m, q, rng, sig))
        rel_hands.set_target(ct.get_transform("left_palm", "site", "right_palm", "site"))
        rel_foot.set_target(ct.get_transform("left_foot", "site", "pelvis", "body"))
        torso.set_target(ct.get_transform_frame_to_world("torso_link", "body"))
        rfoot.set_target(ct.get_transform_frame_to_world("right_foot", "site"))
        post.set_target(perturbed(m, stand, rng, 0.05))           # per-instance posture target
        com.set_target(ct.data.subtree_com[1].copy())             # per-instance CoM target
        return {"frame_targets": [rel_hands.transform_target_to_root.wxyz_xyz, rel_foot.transform_target_to_root.wxyz_xyz,
                                  torso.transform_target_to_world.wxyz_xyz, rfoot.transform_target_to_world.wxyz_xyz],
                "posture_targets": [damp.target_q, post.target_q], "com_targets": [com.target_com]}

    record("g1_ext", m, tasks, lims, 5e-3, 1e-2, qb, set_targets,
           extra={"posture_cost": post.cost.copy(), "cfg_lower": lims[0].lower, "cfg_upper": lims[0].upper})


def ur5e_coll(rng):
    m = mujoco.MjModel.from_xml_path(ROBOTS["ur5e"])
    home = m.key_qpos[m.key("home").id]
    ee = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    wrist = mink.FrameTask("wrist_2_link", "geom", position_cost=0.5, orientation_cost=[0.1, 0.2, 0.3], gain=0.7)
    col = mink.CollisionAvoidanceLimit(m, [(["wrist_3_link"], ["floor", "wall"])], collision_detection_distance=0.3)
    vel = mink.VelocityLimit(m, {n: np.pi for n in ("shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3")})
    lims = [mink.ConfigurationLimit(m), col, vel]
    dt = 5e-2          # a 20 Hz step: the half-spaces bind (h ∝ 1/dt); the example's 500 Hz step is config 2
    # half of the batch within centimetres of the wall / floor, half anywhere
    pool = sample_q(m, rng, 4096, base_q=home)
    pool[::2] = home + rng.normal(scale=0.4, size=(2048, m.nq))
    hmin = []
    for q in pool:
        _, h = col.compute_qp_inequalities(mink.Configuration(m, q), dt)
        hmin.append(np.where(np.isfinite(h), h, np.inf).min())
    hmin = np.array(hmin)
    near = np.flatnonzero(hmin > 0)
    near = near[np.argsort(hmin[near])][:12]
    qb = np.concatenate([pool[near], pool[rng.choice(np.flatnonzero(hmin > 0), size=12, replace=False)]])
    wall = m.geom("wall").id
    wall_pos = m.body_pos[m.geom_bodyid[wall]] + m.geom_pos[wall]

    def set_targets(i, q):
        ct = mink.Configuration(m, perturbed(m, q, rng, 0.15))
        T = ct.get_transform_frame_to_world("attachment_site", "site").wxyz_xyz.copy()
        if i % 2 == 0:
            T[4:] = wall_pos + rng.normal(scale=0.05, size=3)     # inside / behind the wall
        else:
            T[6] = -0.1                                           # below the floor
        ee.set_target(mink.SE3(wxyz_xyz=T))
        wrist.set_target(ct.get_transform_frame_to_world("wrist_2_link", "geom"))
        return {"frame_targets": [ee.transform_target_to_world.wxyz_xyz, wrist.transform_target_to_world.wxyz_xyz]}

    out = record("ur5e_coll", m, [ee, wrist], lims, dt, 1e-3, qb, set_targets, store_G=24,
                 extra={"geom_id_pairs": np.array(col.geom_id_pairs)})
    n = 6
    hc = out["h"][:, 2 * n:2 * n + 2]
    Gx = np.einsum("bpj,bj->bp", out["G"][:, 2 * n:2 * n + 2], out["v"] * dt)
    print("ur5e_coll: finite rows", np.isfinite(hc).sum(axis=0), "binding", (np.abs(Gx - hc) < 1e-9).sum(axis=0))


BALLSLIDE_XML = open(os.path.join(HERE, "ballslide.xml")).read()


def ballslide(rng):
    m = mujoco.MjModel.from_xml_string(BALLSLIDE_XML)
    m.save(os.path.join(HERE, "models", "ballslide.json"))
    ft = mink.FrameTask("tip", "site", position_cost=2.0, orientation_cost=0.5, lm_damping=0.1)
    body_ft = mink.FrameTask("slider", "body", position_cost=[1.0, 0.0, 0.3], orientation_cost=0.0)
    post = mink.PostureTask(m, cost=[0.3, 0.2, 0.1, 0.05, 0.4, 0.25, 0.15, 0.6])
    post.set_target(m.qpos0)
    vel = mink.VelocityLimit(m, {"ball": (np.pi, np.pi / 2, np.pi / 4), "hinge": (0.5,), "slide": (0.2,),
                                 "ball2": (1.0, 1.0, 1.0)})
    lims = [mink.ConfigurationLimit(m), vel]
    n = 24
    qb = np.tile(m.qpos0, (n, 1))
    for i in range(n):
        qb[i] = perturbed(m, m.qpos0, rng, 0.5)
        for j in range(m.njnt):
            if m.jnt_limited[j] and m.jnt_type[j] in (2, 3):
                a = m.jnt_qposadr[j]
                lo, hi = m.jnt_range[j]
                qb[i, a] = rng.uniform(lo + 0.02 * (hi - lo), hi - 0.02 * (hi - lo))
    # un-normalised ball quaternions must be tolerated exactly like MuJoCo does (mj_kinematics normalises)
    qb[3, 0:4] *= 1.0 + 1e-9
    qb[5, 0:4] *= -1.0                                           # the antipodal quaternion: same rotation

    def set_targets(i, q):
        ct = mink.Configuration(m, perturbed(m, q, rng, 1e-4 if (i % 8 == 7) else 0.2))
        ft.set_target(ct.get_transform_frame_to_world("tip", "site"))
        body_ft.set_target(ct.get_transform_frame_to_world("slider", "body"))
        return {"frame_targets": [ft.transform_target_to_world.wxyz_xyz, body_ft.transform_target_to_world.wxyz_xyz]}

    record("ballslide", m, [ft, post, body_ft], lims, 1e-2, 1e-4, qb, set_targets, store_G=4,
           extra={"posture_target": m.qpos0.copy(), "posture_cost": post.cost.copy(),
                  "vel_indices": np.array(vel.indices), "vel_limit": np.array(vel.limit)})


def balllimit(rng):
    """A LIMITED ball joint under the default ConfigurationLimit.  The reference writes the scalar range ends into the
    joint's four qpos slots and differentiates quaternions (configuration_limit.py:46-52, 94-112), and its check_limits
    compares the quaternion's w with the range (configuration.py:92-99: a warning with safety_break=False) — quirks
    the device path reproduces instead of refusing the model."""
    m = mujoco.MjModel.from_xml_string(open(os.path.join(HERE, "balllimit.xml")).read())
    m.save(os.path.join(HERE, "models", "balllimit.json"))
    ft = mink.FrameTask("tip", "site", position_cost=2.0, orientation_cost=0.5, lm_damping=0.1)
    post = mink.PostureTask(m, cost=0.1)
    post.set_target(m.qpos0)
    lims = [mink.ConfigurationLimit(m, gain=0.9)]
    assert len(lims[0].indices) == 2 + 3            # hinge, slide, and the three dofs of the limited ball joint
    qb = []
    while len(qb) < 16:
        q = perturbed(m, m.qpos0, rng, 0.3)
        for j in range(m.njnt):
            if m.jnt_limited[j] and m.jnt_type[j] in (2, 3):
                lo, hi = m.jnt_range[j]
                q[m.jnt_qposadr[j]] = rng.uniform(lo + 0.02 * (hi - lo), hi - 0.02 * (hi - lo))
        qb.append(q)
    qb = np.array(qb)

    def set_targets(i, q):
        ct = mink.Configuration(m, perturbed(m, q, rng, 0.2))
        ft.set_target(ct.get_transform_frame_to_world("tip", "site"))
        return {"frame_targets": [ft.transform_target_to_world.wxyz_xyz]}

    record("balllimit", m, [ft, post], lims, 1e-2, 1e-4, qb, set_targets, store_G=16,
           extra={"posture_target": m.qpos0.copy(), "cfg_indices": np.array(lims[0].indices)})


def main():
    rng = np.random.default_rng(2)
    g1_ext(rng)
    ur5e_coll(rng)
    ballslide(rng)
    balllimit(np.random.default_rng(5))


if __name__ == "__main__":
    main()
