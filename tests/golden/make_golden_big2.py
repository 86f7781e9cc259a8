"""BASELINE-sized slices of configs 2 and 4 from the REAL reference (round 4):

  ik_shadow_c4_big.npz  2 048 Shadow-hand instances (5 fingertip FrameTasks + PostureTask + ConfigurationLimit +
                        CollisionAvoidanceLimit, 40 capsule pairs; half of the instances pulled towards `grasp hard` so that
                        contact rows bind) → v and h of the 40 pairs — for the production tight-rows launch
                        (`ik_solve_kernel_48_72+redo_64`), whose row selection cannot be tapped.
  ik_ur5e_c2_big.npz    4 096 UR5e instances of config 2 (the batch the metric is quoted on) → v — for the row kernel
                        (`ik_quad_kernel`) and, by flag, the lane kernel.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_big2.py

Same set-up, sampling and stubs as make_golden.py / make_golden_big.py (mink's own Python runs; mujoco / qpsolvers are
oracle/stubs).  Only inputs, v (and the contact bounds h) are stored."""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up sys.path: stubs, /root/reference, repo)

mink, mujoco = mg.mink, mg.mujoco


def _run(m, tasks, fts, lims, dt, damping, qb, rng, keep_h=0):
    tg_all, v_all, h_all = [], [], []
    for i, q in enumerate(qb):
        sig = 1e-4 if (i % 8 == 7) else 0.15            # small-angle stress sub-stream, as in make_golden.run_config
        cfg_t = mink.Configuration(m, mg.perturbed(m, q, rng, sig))
        tg = []
        for t in fts:
            T = cfg_t.get_transform_frame_to_world(t.frame_name, t.frame_type)
            t.set_target(T)
            tg.append(T.wxyz_xyz.copy())
        cfg = mink.Configuration(m, q)
        if keep_h:
            h_all.append(mink.build_ik(cfg, tasks, dt, damping, lims).h[-keep_h:].copy())
        v_all.append(mink.solve_ik(cfg, tasks, dt, "quadprog", damping, limits=lims))
        tg_all.append(np.array(tg))
    out = dict(q=qb, frame_targets=np.array(tg_all), v=np.array(v_all), dt=np.array(dt), damping=np.array(damping))
    if keep_h:
        out["coll_h"] = np.array(h_all)
    return out


def shadow(n=2048):
    rng = np.random.default_rng(4044)
    m = mujoco.MjModel.from_xml_path(mg.ROBOTS["shadow_left"])
    grasp = m.key_qpos[m.key("grasp hard").id]
    fingers = ["thumb", "first", "middle", "ring", "little"]
    fts = [mink.FrameTask(f, "site", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0) for f in fingers]
    pt = mink.PostureTask(m, cost=1e-2)
    pt.set_target(grasp)
    groups = [[f"{f}_1", f"{f}_2"] for f in fingers]
    pairs = [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)]
    col = mink.CollisionAvoidanceLimit(m, pairs, collision_detection_distance=0.03)
    assert np.array_equal(np.array(col.geom_id_pairs), np.load(os.path.join(HERE, "shadow_c4_geom_pairs.npy")))
    qb = mg.sample_q(m, rng, n, base_q=grasp)
    qb[::2] = 0.5 * (qb[::2] + grasp)                   # fingers come close: contact rows bind
    t0 = time.time()
    out = _run(m, [pt] + fts, fts, [mink.ConfigurationLimit(m), col], 2e-3, 1e-5, qb, rng, keep_h=len(col.geom_id_pairs))
    out["posture_target"] = grasp.copy()
    np.savez_compressed(os.path.join(HERE, "ik_shadow_c4_big.npz"), **out)
    fin = np.isfinite(out["coll_h"])
    print("ik_shadow_c4_big.npz", {k: v.shape for k, v in out.items()}, f"{time.time() - t0:.1f} s", "max|v|",
          float(np.abs(out["v"]).max()), "contacts in range per instance: mean %.1f max %d" % (fin.sum(1).mean(), fin.sum(1).max()))


def ur5e(n=4096):
    rng = np.random.default_rng(4045)
    m = mujoco.MjModel.from_xml_path(mg.ROBOTS["ur5e"])
    home = m.key_qpos[m.key("home").id]
    ft = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    pt = mink.PostureTask(m, cost=1e-2)
    pt.set_target(home)
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {nm: np.pi for nm in m.jnt_names})]
    qb = mg.sample_q(m, rng, n)
    t0 = time.time()
    out = _run(m, [ft, pt], [ft], lims, 2e-3, 1e-3, qb, rng)
    out["posture_target"] = home.copy()
    np.savez_compressed(os.path.join(HERE, "ik_ur5e_c2_big.npz"), **out)
    print("ik_ur5e_c2_big.npz", {k: v.shape for k, v in out.items()}, f"{time.time() - t0:.1f} s", "max|v|",
          float(np.abs(out["v"]).max()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["shadow", "ur5e"]
    if "shadow" in which:
        shadow()
    if "ur5e" in which:
        ur5e()
