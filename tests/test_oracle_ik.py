"""oracle/ik.py replayed against fixtures recorded from the real mink
(tests/golden/ik_*.npz).  CPU only."""

import os

import numpy as np
import pytest

import oracle_configs as oc
from oracle import ik, qp_gi


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"ik_{name}.npz"))


def _check(cfgfn, d, extra=None, vtol=1e-9):
    n = len(d["q"])
    for i in range(n):
        args = [d["frame_targets"][i], d["posture_target"]]
        if extra is not None:
            args.append(d[extra][i])
        m, tasks, limits, dt, damping = cfgfn(*args)
        assert dt == float(d["dt"]) and damping == float(d["damping"])
        cfg = ik.Configuration(m, d["q"][i])
        P, c, G, h = ik.build_ik(cfg, tasks, dt, damping, limits)
        scale = max(1.0, np.abs(d["H"][i]).max())
        np.testing.assert_allclose(P, d["H"][i], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(c, d["c"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["c"][i]).max()))
        np.testing.assert_allclose(h, d["h"][i], rtol=0, atol=1e-13)
        if i < len(d["G"]):
            np.testing.assert_allclose(G, d["G"][i], rtol=0, atol=1e-13)
            J = np.vstack([ik.task_error_jacobian(cfg, t)[1] for t in tasks])
            np.testing.assert_allclose(J, d["task_J"][i], rtol=0, atol=1e-11)
        e = np.concatenate([ik.task_error_jacobian(cfg, t)[0] for t in tasks])
        np.testing.assert_allclose(e, d["task_e"][i], rtol=0, atol=1e-13)
        v = ik.solve_ik(m, cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=vtol * max(1.0, np.abs(d["v"][i]).max()))
        assert qp_gi.kkt_residual(P, c, G, h, v * dt) < 1e-7 * max(1.0, np.abs(c).max())


def test_ur5e_c2(golden_dir):
    _check(oc.ur5e_c2, _load(golden_dir, "ur5e_c2"))


def test_g1_c3(golden_dir):
    _check(oc.g1_c3, _load(golden_dir, "g1_c3"))


def test_g1_full(golden_dir):
    _check(oc.g1_full, _load(golden_dir, "g1_full"), extra="com_target")


def test_shadow_c4(golden_dir):
    _check(oc.shadow_c4, _load(golden_dir, "shadow_c4"))


def test_ur5e_c1_trajectory(golden_dir):
    """Config 1: the reference's own convergence behaviour (tests/test_solve_ik.py:95-148)
    replayed: same v at every step of the recorded solve+integrate trajectory."""
    d = _load(golden_dir, "ur5e_c1")
    m, tasks, limits, dt, damping = oc.ur5e_c1([d["frame_target"]], d["posture_target"])
    cfg = ik.Configuration(m, d["q"][0])
    errs = []
    for k in range(len(d["q"])):
        np.testing.assert_allclose(cfg.q, d["q"][k], rtol=0, atol=1e-12)
        v = ik.solve_ik(m, cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(v, d["v"][k], rtol=0, atol=1e-9 * max(1.0, np.abs(v).max()))
        errs.append(np.linalg.norm(ik.task_error_jacobian(cfg, tasks[0])[0]))
        cfg.update(cfg.integrate(v, dt))
    # error contracts until it balances the posture regulariser (≈1.3e-4)
    assert all(b < a for a, b in zip(errs[:4], errs[1:5]))
    assert errs[-1] < 2e-4
    np.testing.assert_allclose(cfg.q, d["q_final"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["g1_ext", "ur5e_coll", "ballslide", "balllimit"])
def test_round2_fixtures(golden_dir, name):
    """RelativeFrameTask / DampingTask / body- and geom-frame tasks / per-instance posture and CoM targets /
    the arm_ur5e.py collision set-up / ball + slide joints, recorded from the real mink by make_golden_ext.py."""
    d = _load(golden_dir, name)
    for i in range(len(d["q"])):
        m, tasks, limits, dt, damping = oc.EXT[name](d, i)
        assert dt == float(d["dt"]) and damping == float(d["damping"])
        cfg = ik.Configuration(m, d["q"][i])
        P, c, G, h = ik.build_ik(cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(P, d["H"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["H"][i]).max()))
        np.testing.assert_allclose(c, d["c"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["c"][i]).max()))
        fin = np.isfinite(d["h"][i])
        assert (np.isfinite(h) == fin).all()
        np.testing.assert_allclose(h[fin], d["h"][i][fin], rtol=0, atol=1e-12 * max(1.0, np.abs(d["h"][i][fin]).max()))
        if i < len(d["G"]):
            np.testing.assert_allclose(G, d["G"][i], rtol=0, atol=1e-13)
            J = np.vstack([ik.task_error_jacobian(cfg, t)[1] for t in tasks])
            np.testing.assert_allclose(J, d["task_J"][i], rtol=0, atol=1e-11)
        e = np.concatenate([ik.task_error_jacobian(cfg, t)[0] for t in tasks])
        np.testing.assert_allclose(e, d["task_e"][i], rtol=0, atol=1e-13)
        v = ik.solve_ik(m, cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][i]).max()))


@pytest.mark.parametrize("name,scene", [("aloha_coll", "aloha__scene"), ("shadow_tips", "shadow_hand__scene_left")])
def test_oracle_vs_real_mink_on_the_mesh_dependent_collision_setups(golden_dir, name, scene):
    """tests/golden/make_golden_mesh.py (the real mink: 1 104 ALOHA pairs of mesh-fitted capsules; Shadow fingertips + forearm
    mesh): the numpy oracle's own pipeline — its limit rows and its Goldfarb–Idnani solve — against mink's Python on the same
    model and the same distance routine."""
    from mink_amd.flatmodel import FlatModel
    d = np.load(os.path.join(golden_dir, f"ik_{name}.npz"))
    m = FlatModel.load(os.path.join(golden_dir, "models", "all", scene + ".json"))
    dt, damping = float(d["dt"]), float(d["damping"])
    if name == "aloha_coll":
        names = ["left/gripper", "right/gripper"]
        mk = lambda i: [ik.FrameTaskSpec(m.name2id("site", s), "site", np.ones(6), d["frame_targets"][i, k], 1.0, 1.0) for k, s in enumerate(names)] + \
            [ik.PostureTaskSpec(np.full(m.nv, 1e-4), d["posture_target"])]
        vidx = [int(m.jnt_dofadr[m.name2id("joint", f"{p}/{n}")]) for p in ("left", "right")
                for n in ("waist", "shoulder", "elbow", "forearm_roll", "wrist_angle", "wrist_rotate")]
        col = ik.CollisionAvoidanceLimitSpec([tuple(p) for p in d["geom_id_pairs"]], minimum_distance_from_collisions=0.05,
                                             collision_detection_distance=0.1)
        limits = [ik.ConfigurationLimitSpec(), ik.VelocityLimitSpec(np.array(vidx), np.full(12, np.pi)), col]
    else:
        names = ["thumb", "first", "middle", "ring", "little"]
        mk = lambda i: [ik.PostureTaskSpec(np.full(m.nv, 1e-2), d["posture_target"])] + \
            [ik.FrameTaskSpec(m.name2id("site", s), "site", np.array([1, 1, 1, 0, 0, 0.0]), d["frame_targets"][i, k], 1.0, 1.0) for k, s in enumerate(names)]
        col = ik.CollisionAvoidanceLimitSpec([tuple(p) for p in d["geom_id_pairs"]], minimum_distance_from_collisions=0.004,
                                             collision_detection_distance=0.06)
        limits = [ik.ConfigurationLimitSpec(), col]
    for i in (0, 5, 11):
        cfg = ik.Configuration(m, d["q"][i])
        P, c, G, h = ik.build_ik(cfg, mk(i), dt, damping, limits)
        np.testing.assert_allclose(P, d["H"][i], rtol=0, atol=1e-12 * np.abs(d["H"][i]).max())
        fin = np.isfinite(d["h"][i])
        assert (np.isfinite(h) == fin).all()
        np.testing.assert_allclose(h[fin], d["h"][i][fin], rtol=0, atol=1e-12 * max(1.0, np.abs(d["h"][i][fin]).max()))
        if i < len(d["G"]):
            np.testing.assert_allclose(G, d["G"][i], rtol=0, atol=1e-12)
        v = ik.solve_ik(m, d["q"][i], mk(i), dt, damping, limits)
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=1e-8 * max(1.0, np.abs(d["v"][i]).max()))


def _small_robot_specs(name, m, d, i):
    """Task / limit specs of the row-kernel fixtures (tests/golden/make_golden_small.py) for instance i."""
    if name == "leap_c":
        sid = [m.name2id("site", s) for s in ("tip_1", "tip_2", "tip_3", "th_tip")]
        tasks = [ik.FrameTaskSpec(s, "site", np.array([1.0, 1, 1, 0, 0, 0]), d["frame_targets"][i, k], 1.0, 1.0) for k, s in enumerate(sid)]
        tasks.append(ik.PostureTaskSpec(np.full(m.nv, 1e-2), d["posture_target"]))
        vlim = np.full(m.nv, np.pi)
    elif name in ("h1_c", "go1_c", "h1_full"):
        # tests/golden/make_golden_mid.py: the tasks of examples/humanoid_h1.py (without the CoM task) / quadruped_go1.py
        c6 = lambda p, o: np.array([p] * 3 + [o] * 3, dtype=np.float64)
        hinge = np.array([int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] == 3])
        if name in ("h1_c", "h1_full"):
            spec = [("pelvis", "body", c6(0, 10), 0.0)] + [(s, "site", c6(200, 10), 1.0) for s in ("right_foot", "left_foot")] + \
                   [(s, "site", c6(200, 0), 1.0) for s in ("right_wrist", "left_wrist")]
            pcost, limits = 1.0, [ik.ConfigurationLimitSpec(), ik.VelocityLimitSpec(hinge, np.full(len(hinge), np.pi))]
        else:
            spec = [("trunk", "body", c6(1, 1), 0.0)] + [(s, "site", c6(1, 0), 0.0) for s in ("FL", "FR", "RR", "RL")]
            pcost, limits = 1e-5, [ik.ConfigurationLimitSpec()]
        tasks = [ik.FrameTaskSpec(m.name2id(ft, n), ft, c, d["frame_targets"][i, k], 1.0, lm) for k, (n, ft, c, lm) in enumerate(spec)]
        tasks.append(ik.PostureTaskSpec(np.full(m.nv, pcost), d["posture_target"]))
        if name == "h1_full":                          # examples/humanoid_h1.py:33: ComTask(cost=200), per-instance target
            tasks.append(ik.ComTaskSpec(np.full(3, 200.0), d["com_targets"][i, 0]))
        return tasks, limits
    else:
        tasks = [ik.FrameTaskSpec(m.name2id("site", "pinch_site"), "site", np.ones(6), d["frame_targets"][i, 0], 1.0, 1.0)]
        pc = np.zeros(m.nv); pc[2] = 1e-3
        dc = np.zeros(m.nv); dc[:2] = 100.0; dc[2] = 1e-3
        tasks.append(ik.PostureTaskSpec(pc, d["posture_targets"][i, 0]))
        tasks.append(ik.PostureTaskSpec(dc, d["posture_targets"][i, 1], gain=0.0))      # DampingTask (damping_task.py:11-20)
        vlim = np.array([0.5 if m.jnt_type[j] == 2 else np.pi for j in range(m.njnt)])
    idx = np.array([int(m.jnt_dofadr[j]) for j in range(m.njnt)])
    return tasks, [ik.ConfigurationLimitSpec(), ik.VelocityLimitSpec(idx, vlim)]


@pytest.mark.parametrize("name,scene", [("leap_c", "leap_hand__scene_right"), ("kinova_c", "stanford_tidybot__scene_mobile_kinova"),
                                        ("h1_c", "unitree_h1__scene"), ("go1_c", "unitree_go1__scene"), ("h1_full", "unitree_h1__scene")])
def test_oracle_vs_real_mink_on_the_hands_and_mobile_arms(golden_dir, name, scene):
    """The real mink on a 16-dof hand (four fingertip tasks) and a 10-dof mobile arm (DampingTask on the base), the robots of
    the row kernel's sixteen-register build: H, c, h, G, e, J and v of the oracle against it."""
    from mink_amd.flatmodel import FlatModel
    d = _load(golden_dir, name)
    m = FlatModel.load(os.path.join(golden_dir, "models", "all", scene + ".json"))
    dt, damping = float(d["dt"]), float(d["damping"])
    for i in range(len(d["q"])):
        tasks, limits = _small_robot_specs(name, m, d, i)
        cfg = ik.Configuration(m, d["q"][i])
        P, c, G, h = ik.build_ik(cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(P, d["H"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["H"][i]).max()))
        np.testing.assert_allclose(c, d["c"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["c"][i]).max()))
        np.testing.assert_allclose(h, d["h"][i], rtol=0, atol=1e-13)
        if i < len(d["G"]):
            np.testing.assert_allclose(G, d["G"][i], rtol=0, atol=1e-13)
        e = np.concatenate([ik.task_error_jacobian(cfg, t)[0] for t in tasks])
        np.testing.assert_allclose(e, d["task_e"][i], rtol=0, atol=1e-13)
        v = ik.solve_ik(m, cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][i]).max()))


def _arm_hand_specs(m, d, i):
    """tests/golden/make_golden_mid.py::arm_hand — the task set of examples/arm_hand_iiwa_allegro.py:62-94."""
    one = np.array([1.0, 1, 1, 0, 0, 0])
    tasks = [ik.FrameTaskSpec(m.name2id("site", "attachment_site"), "site", np.ones(6), d["frame_targets"][i, 0], 1.0, 1.0),
             ik.PostureTaskSpec(np.full(m.nv, 5e-2), d["posture_target"])]
    palm = m.name2id("body", "palm")
    for k, t in enumerate(("ff_tip", "mf_tip", "rf_tip", "th_tip")):
        tasks.append(ik.RelativeFrameTaskSpec(m.name2id("site", t), "site", palm, "body", one, d["frame_targets"][i, 1 + k], 1.0, 1.0))
    return tasks, [ik.ConfigurationLimitSpec()]


def test_oracle_vs_real_mink_on_the_arm_with_a_hand(golden_dir):
    """The real mink on a 7-dof arm carrying a 16-dof hand with the reference's arm + hand task set (one RelativeFrameTask per
    fingertip, measured in the palm): H, c, h, e and v of the oracle against it."""
    from mink_amd.flatmodel import FlatModel
    d = _load(golden_dir, "arm_hand")
    m = FlatModel.load(os.path.join(golden_dir, "models", "arm_hand.json"))
    dt, damping = float(d["dt"]), float(d["damping"])
    for i in range(len(d["q"])):
        tasks, limits = _arm_hand_specs(m, d, i)
        cfg = ik.Configuration(m, d["q"][i])
        P, c, G, h = ik.build_ik(cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(P, d["H"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["H"][i]).max()))
        np.testing.assert_allclose(c, d["c"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["c"][i]).max()))
        np.testing.assert_allclose(h, d["h"][i], rtol=0, atol=1e-13)
        e = np.concatenate([ik.task_error_jacobian(cfg, t)[0] for t in tasks])
        np.testing.assert_allclose(e, d["task_e"][i], rtol=0, atol=1e-13)
        v = ik.solve_ik(m, cfg, tasks, dt, damping, limits)
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][i]).max()))
