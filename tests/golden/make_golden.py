"""Generate the golden fixtures in this directory by running the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py
    python tests/golden/make_golden.py --real-wheels     # on a machine WITH the mujoco / qpsolvers / quadprog wheels: re-run the
                                                         # committed inputs through the real stack and diff (real_wheels.py);
                                                         # writes nothing into this directory

It imports ``mink`` from /root/reference.  The reference's third-party wheels
``mujoco`` and ``qpsolvers`` are not installed here, so they are replaced by
oracle/stubs (which expose the handful of symbols mink uses, implemented by
oracle/mjmath.py and oracle/qp_gi.py — see oracle/stubs/README.md).  What the
fixtures therefore pin is every line of mink's own Python on the hot path:
mink/lie/*, mink/configuration.py, mink/tasks/*, mink/limits/*, mink/solve_ik.py.

Outputs (np.savez_compressed):
  lie.npz                  SO3/SE3 known-answer vectors from mink.lie
  ik_<config>.npz          per BASELINE config: inputs (q, targets) and the
                           reference's per-task (e, J), (H, c), (h[, G]) and v.
  models/<robot>.json      FlatModel compiled from the reference's example MJCF
                           (the only source of robot models: no network).
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if "--real-wheels" in sys.argv[1:]:
    sys.path.insert(0, HERE)
    import real_wheels
    sys.exit(real_wheels.main([a for a in sys.argv[1:] if a != "--real-wheels"]))
sys.path[:0] = [os.path.join(REPO, "oracle", "stubs"), "/root/reference", REPO]

import mujoco  # noqa: E402  (the stub)
import mink  # noqa: E402  (the real reference)

EX = "/root/reference/examples/"
ROBOTS = {
    "ur5e": EX + "universal_robots_ur5e/scene.xml",
    "g1": EX + "unitree_g1/scene.xml",
    "shadow_left": EX + "shadow_hand/scene_left.xml",
}


# ------------------------------------------------------------------ lie golden
def make_lie(rng):
    n = 64
    out = {}
    np.random.seed(1234)  # mink's sample_uniform uses the global numpy RNG
    T = [mink.SE3.sample_uniform() for _ in range(n)]
    # scale some rotations down to exercise the small-angle branches
    tang = rng.normal(size=(n, 6))
    scales = 10.0 ** rng.uniform(-7, 0.3, size=n)
    tang[:, 3:] *= (scales / np.linalg.norm(tang[:, 3:], axis=1))[:, None]
    tang[0, 3:] = 0.0
    Te = [mink.SE3.exp(t) for t in tang]
    out["se3_params"] = np.array([t.wxyz_xyz for t in T])
    out["tangent"] = tang
    out["se3_exp"] = np.array([t.wxyz_xyz for t in Te])
    out["se3_log"] = np.array([t.log() for t in T])
    out["se3_log_of_exp"] = np.array([t.log() for t in Te])
    out["se3_inverse"] = np.array([t.inverse().wxyz_xyz for t in T])
    out["se3_multiply"] = np.array([(T[i] @ T[(i + 1) % n]).wxyz_xyz for i in range(n)])
    out["se3_adjoint"] = np.array([t.adjoint() for t in T])
    out["se3_jlog"] = np.array([t.jlog() for t in T])
    out["se3_jlog_of_exp"] = np.array([t.jlog() for t in Te])
    out["se3_ljacinv"] = np.array([mink.SE3.ljacinv(t) for t in tang])
    out["se3_rminus"] = np.array([T[i].rminus(T[(i + 1) % n]) for i in range(n)])
    out["se3_as_matrix"] = np.array([t.as_matrix() for t in T])
    out["so3_log"] = np.array([t.rotation().log() for t in T])
    out["so3_exp"] = np.array([mink.SO3.exp(t[3:]).wxyz for t in tang])
    out["so3_as_matrix"] = np.array([t.rotation().as_matrix() for t in T])
    out["so3_from_matrix"] = np.array([mink.SO3.from_matrix(t.rotation().as_matrix()).wxyz for t in T])
    out["so3_ljacinv"] = np.array([mink.SO3.ljacinv(t[3:]) for t in tang])
    pts = rng.normal(size=(n, 3))
    out["points"] = pts
    out["so3_apply"] = np.array([T[i].rotation().apply(pts[i]) for i in range(n)])
    out["se3_apply"] = np.array([T[i].apply(pts[i]) for i in range(n)])
    # quaternions with w<0 and w≈0 (log branch coverage, so3.py:176-191)
    special = np.array([
        [-0.5, 0.5, 0.5, 0.5], [1e-12, 1.0, 0.0, 0.0], [-1e-12, 0.0, 1.0, 0.0],
        [1.0, 1e-7, 0.0, 0.0], [-1.0, 0.0, 1e-6, 0.0], [0.0, 0.0, 0.0, 1.0],
    ])
    out["so3_special"] = special
    out["so3_special_log"] = np.array([mink.SO3(wxyz=s).log() for s in special])
    np.savez_compressed(os.path.join(HERE, "lie.npz"), **out)
    print("lie.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------- sampling
def sample_q(model, rng, n, base_q=None):
    """SURVEY §8(d) input distribution: hinge joints uniform inside the range, 10 %
    of instances with 1–3 joints within 1e-3·range of a bound; free joint near the
    keyframe pose with a random unit quaternion."""
    q = np.tile(np.asarray(model.qpos0 if base_q is None else base_q, dtype=np.float64), (n, 1))
    for j in range(model.njnt):
        a = model.jnt_qposadr[j]
        if model.jnt_type[j] == 0:
            q[:, a:a + 3] = np.array([0, 0, 0.75]) + rng.normal(scale=0.05, size=(n, 3))
            w = rng.normal(scale=0.2, size=(n, 3))
            for i in range(n):
                q[i, a + 3:a + 7] = mink.SO3.exp(w[i]).wxyz
        else:
            lo, hi = model.jnt_range[j]
            if not model.jnt_limited[j]:
                lo, hi = -np.pi, np.pi
            wdt = hi - lo
            q[:, a] = rng.uniform(lo + 0.05 * wdt, hi - 0.05 * wdt, size=n)
    hinge = [j for j in range(model.njnt) if model.jnt_type[j] != 0 and model.jnt_limited[j]]
    for i in range(n):
        if rng.uniform() < 0.10:
            for j in rng.choice(hinge, size=rng.integers(1, 4), replace=False):
                lo, hi = model.jnt_range[j]
                eps = 1e-3 * (hi - lo) * rng.uniform()
                q[i, model.jnt_qposadr[j]] = (lo + eps) if rng.uniform() < 0.5 else (hi - eps)
    return q


def perturbed(model, q, rng, sigma):
    q2 = q.copy()
    v = rng.normal(scale=sigma, size=model.nv)
    mujoco.mj_integratePos(model, q2, v, 1.0)
    return q2


def run_config(name, model, tasks, limits, dt, damping, frame_tasks, q_batch, rng,
               sigma=0.15, posture_task=None, com_task=None, store_G=2):
    """Set targets = FK(q ⊕ δ) per instance, run the real mink, record everything."""
    rec = {k: [] for k in ("q", "frame_targets", "com_target", "v", "H", "c", "h", "G",
                           "task_e", "task_J")}
    for i, q in enumerate(q_batch):
        sig = 1e-4 if (i % 8 == 7) else sigma          # small-angle stress sub-stream
        cfg_t = mink.Configuration(model, perturbed(model, q, rng, sig))
        tg = []
        for t in frame_tasks:
            T = cfg_t.get_transform_frame_to_world(t.frame_name, t.frame_type)
            t.set_target(T)
            tg.append(T.wxyz_xyz.copy())
        if com_task is not None:
            com_task.set_target(cfg_t.data.subtree_com[1].copy())
            rec["com_target"].append(com_task.target_com.copy())
        cfg = mink.Configuration(model, q)
        problem = mink.build_ik(cfg, tasks, dt, damping, limits)
        v = mink.solve_ik(cfg, tasks, dt, "quadprog", damping, limits=limits)
        rec["q"].append(q.copy()); rec["frame_targets"].append(np.array(tg))
        rec["v"].append(v); rec["H"].append(problem.P); rec["c"].append(problem.q)
        rec["h"].append(problem.h if problem.h is not None else np.zeros(0))
        if i < store_G and problem.G is not None:
            rec["G"].append(problem.G)
        rec["task_e"].append(np.concatenate([t.compute_error(cfg) for t in tasks]))
        if i < store_G:
            rec["task_J"].append(np.vstack([t.compute_jacobian(cfg) for t in tasks]))
    out = {k: np.array(v) for k, v in rec.items() if len(v)}
    out["dt"] = np.array(dt); out["damping"] = np.array(damping)
    if posture_task is not None:
        out["posture_target"] = posture_task.target_q.copy()
    np.savez_compressed(os.path.join(HERE, f"ik_{name}.npz"), **out)
    print(f"ik_{name}.npz", {k: v.shape for k, v in out.items()},
          "max|v|", float(np.abs(out["v"]).max()))


def main():
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    models = {}
    for name, path in ROBOTS.items():
        models[name] = mujoco.MjModel.from_xml_path(path)
        models[name].save(os.path.join(HERE, "models", f"{name}.json"))
    make_lie(rng)

    # ---------------- config 1/2: UR5e (examples/arm_ur5e.py:20-47,74)
    m = models["ur5e"]
    ft = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0,
                        lm_damping=1.0)
    pt = mink.PostureTask(m, cost=1e-2)
    home = m.key_qpos[m.key("home").id]
    pt.set_target(home)
    # config 1: keyframe start, +0.1 m z target (tests/test_solve_ik.py:107-109)
    cfg = mink.Configuration(m, home)
    T0 = cfg.get_transform_frame_to_world("attachment_site", "site")
    ft.set_target(T0 @ mink.SE3.from_translation(np.array([0.0, 0.0, 0.1])))
    rec = {"q": [], "v": []}
    for _ in range(20):                       # an IK trajectory (solve + integrate)
        v = mink.solve_ik(cfg, [ft, pt], 2e-3, "quadprog", 1e-3)
        rec["q"].append(cfg.q); rec["v"].append(v)
        cfg.integrate_inplace(v, 2e-3)
    np.savez_compressed(os.path.join(HERE, "ik_ur5e_c1.npz"), q=np.array(rec["q"]),
                        v=np.array(rec["v"]), frame_target=ft.transform_target_to_world.wxyz_xyz,
                        posture_target=home, dt=2e-3, damping=1e-3, q_final=cfg.q)
    print("ik_ur5e_c1.npz final err", np.linalg.norm(ft.compute_error(cfg)))
    vel = {n: np.pi for n in m.jnt_names}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)]
    run_config("ur5e_c2", m, [ft, pt], lims, 2e-3, 1e-3, [ft], sample_q(m, rng, 32), rng,
               posture_task=pt)

    # ---------------- config 3: G1 (examples/humanoid_g1.py:22-52,80,88)
    m = models["g1"]
    stand = m.key_qpos[m.key("stand").id]
    feet = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0)
            for s in ("left_foot", "right_foot")]
    hands = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=0.0, lm_damping=1.0)
             for s in ("left_palm", "right_palm")]
    pt = mink.PostureTask(m, cost=1.0)
    pt.set_target(stand)
    vel = {m.jnt_names[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] != 0}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)]
    qb = sample_q(m, rng, 24, base_q=stand)
    run_config("g1_c3", m, feet + hands + [pt], lims, 5e-3, 1e-1, feet + hands, qb, rng,
               posture_task=pt)
    # full example variant: + pelvis orientation (body frame) + ComTask
    pelvis = mink.FrameTask("pelvis", "body", position_cost=0.0, orientation_cost=10.0)
    com = mink.ComTask(cost=200.0)
    run_config("g1_full", m, [pelvis, pt, com] + feet + hands, lims, 5e-3, 1e-1,
               [pelvis] + feet + hands, qb[:12], rng, posture_task=pt, com_task=com)

    # ---------------- config 4: Shadow hand (examples/hand_shadow.py:18-30,46-63)
    m = models["shadow_left"]
    grasp = m.key_qpos[m.key("grasp hard").id]
    fingers = ["thumb", "first", "middle", "ring", "little"]
    fts = [mink.FrameTask(f, "site", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0)
           for f in fingers]
    pt = mink.PostureTask(m, cost=1e-2)
    pt.set_target(grasp)
    groups = [[f"{f}_1", f"{f}_2"] for f in fingers]
    pairs = [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)]
    col = mink.CollisionAvoidanceLimit(m, pairs, collision_detection_distance=0.03)
    lims = [mink.ConfigurationLimit(m), col]
    qb = sample_q(m, rng, 24, base_q=grasp)
    # pull half of the samples towards the grasp keyframe so fingers come close
    qb[::2] = 0.5 * (qb[::2] + grasp)
    run_config("shadow_c4", m, [pt] + fts, lims, 2e-3, 1e-5, fts, qb, rng, posture_task=pt)
    np.save(os.path.join(HERE, "shadow_c4_geom_pairs.npy"), np.array(col.geom_id_pairs))
    print("shadow pairs", len(col.geom_id_pairs))


if __name__ == "__main__":
    main()
