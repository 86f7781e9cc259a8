"""Fixtures from the REAL reference for the two mesh-dependent collision set-ups (build container only):

    python tests/golden/make_golden_mesh.py

  ik_aloha_coll.npz    examples/arm_aloha.py:76-121 as written: two FrameTasks + PostureTask(1e-4), ConfigurationLimit,
                       VelocityLimit(π on the twelve arm joints), CollisionAvoidanceLimit(wrist subtree × wrist subtree,
                       both arms × metal frame + table; d_min 0.05, d_detect 0.1) — 1 104 geom pairs that mink's constructor
                       filters out of the subtrees' geoms, every arm / frame collision geom a capsule fitted to its mesh.
  ik_shadow_tips.npz   the Shadow hand with its mesh-fitted `*_3` fingertip capsules and the forearm's MESH collision geom:
                       fingertips × fingertips, × middle phalanges, × forearm hull.

mink's own Python runs (pair filters, Contact, compute_qp_inequalities, build_ik, solve_ik); `mujoco` / `qpsolvers` are
oracle/stubs, so mj_geomDistance is the oracle's (mesh geoms through oracle/gjk.py) and the model is the MJCF reader's
(mink_amd/meshes.py for the fitted capsules) — what these fixtures pin is the mink layer on top: which pairs exist, in which
order, with which rows."""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ext as mg  # noqa: E402

mink, mujoco = mg.mink, mg.mujoco
EX = "/root/reference/examples/"


def _pick(model, col, rng, q0, sigma, n_pool, n_keep, dt):
    """Instances inside their joint ranges, the ones with the most detected contacts first (but none that starts closer
    than d_min: those rows are h = relaxation, nothing to pin), plus a few random ones."""
    lo = np.array([model.jnt_range[j][0] for j in range(model.njnt)]); hi = np.array([model.jnt_range[j][1] for j in range(model.njnt)])
    pool = np.clip(q0 + rng.normal(scale=sigma, size=(n_pool, model.nq)), lo + 1e-3, hi - 1e-3)
    cnt, ok = [], []
    for q in pool:
        _, h = col.compute_qp_inequalities(mink.Configuration(model, q), dt)
        fin = np.isfinite(h)
        cnt.append(int(fin.sum())); ok.append(bool((h[fin] > 0).all()))
    cnt, ok = np.array(cnt), np.array(ok)
    order = [i for i in np.argsort(-cnt) if ok[i] and cnt[i] > 0]
    keep = order[: n_keep - 4] + list(rng.choice(np.flatnonzero(ok), size=4, replace=False))
    print("detected contacts of the kept instances:", cnt[keep].tolist())
    return pool[keep]


def aloha_coll(rng):
    m = mujoco.MjModel.from_xml_path(EX + "aloha/scene.xml")
    l_ee = mink.FrameTask("left/gripper", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    r_ee = mink.FrameTask("right/gripper", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    post = mink.PostureTask(m, cost=1e-4)
    q0 = m.key_qpos[m.key("neutral_pose").id]
    post.set_target(q0)
    l_wrist = mink.get_subtree_geom_ids(m, m.body("left/wrist_link").id)
    r_wrist = mink.get_subtree_geom_ids(m, m.body("right/wrist_link").id)
    l_geoms = mink.get_subtree_geom_ids(m, m.body("left/upper_arm_link").id)
    r_geoms = mink.get_subtree_geom_ids(m, m.body("right/upper_arm_link").id)
    frame = mink.get_body_geom_ids(m, m.body("metal_frame").id)
    col = mink.CollisionAvoidanceLimit(m, [(l_wrist, r_wrist), (l_geoms + r_geoms, frame + ["table"])],
                                       minimum_distance_from_collisions=0.05, collision_detection_distance=0.1)
    vel = mink.VelocityLimit(m, {f"{p}/{n}": np.pi for p in ("left", "right")
                                 for n in ("waist", "shoulder", "elbow", "forearm_roll", "wrist_angle", "wrist_rotate")})
    lims = [mink.ConfigurationLimit(m), vel, col]
    dt = 2e-2
    qb = _pick(m, col, rng, q0, 0.5, 400, 12, dt)

    def set_targets(i, q):
        ct = mink.Configuration(m, mg.perturbed(m, q, rng, 0.1))
        l_ee.set_target(ct.get_transform_frame_to_world("left/gripper", "site"))
        r_ee.set_target(ct.get_transform_frame_to_world("right/gripper", "site"))
        return {"frame_targets": [l_ee.transform_target_to_world.wxyz_xyz, r_ee.transform_target_to_world.wxyz_xyz]}

    mg.record("aloha_coll", m, [l_ee, r_ee, post], lims, dt, 1e-4, qb, set_targets, store_G=4,
              extra={"geom_id_pairs": np.array(col.geom_id_pairs), "posture_target": q0.copy()})


def shadow_tips(rng):
    m = mujoco.MjModel.from_xml_path(EX + "shadow_hand/scene_left.xml")
    fingers = ["thumb", "first", "middle", "ring", "little"]
    fts = [mink.FrameTask(f, "site", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0) for f in fingers]
    post = mink.PostureTask(m, cost=1e-2)
    grasp = m.key_qpos[m.key("grasp hard").id]
    post.set_target(grasp)
    fore = [g for g in range(m.ngeom) if m.geom_type[g] == 7 and m.geom_contype[g] != 0]
    assert len(fore) == 1
    tips, mids = [f"{f}_3" for f in fingers], [f"{f}_2" for f in fingers]
    col = mink.CollisionAvoidanceLimit(m, [(tips, tips), (tips, mids), (tips, fore)],
                                       minimum_distance_from_collisions=0.004, collision_detection_distance=0.06)
    lims = [mink.ConfigurationLimit(m), col]
    dt = 2e-2
    qb = _pick(m, col, rng, grasp, 0.15, 300, 12, dt)

    def set_targets(i, q):
        ct = mink.Configuration(m, mg.perturbed(m, q, rng, 0.1))
        out = []
        for t, f in zip(fts, fingers):
            t.set_target(ct.get_transform_frame_to_world(f, "site"))
            out.append(t.transform_target_to_world.wxyz_xyz)
        return {"frame_targets": out}

    mg.record("shadow_tips", m, [post] + fts, lims, dt, 1e-4, qb, set_targets, store_G=4,
              extra={"geom_id_pairs": np.array(col.geom_id_pairs), "posture_target": grasp.copy()})


if __name__ == "__main__":
    aloha_coll(np.random.default_rng(11))
    shadow_tips(np.random.default_rng(12))
