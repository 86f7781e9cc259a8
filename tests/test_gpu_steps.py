"""Fused on-device outer loop (mkh_solve_steps) = the reference's solve_ik + integrate_inplace loop."""

import numpy as np
import pytest

import mink_amd as mink
import native_configs as nc
import oracle_configs as oc
from mink_amd import _native as nat
from mink_amd import workloads
from oracle import ik as oik

pytestmark = pytest.mark.gpu


def test_fused_steps_match_stepwise_and_oracle():
    model = workloads.load_robot("g1")
    nm = nat.NativeModel(model)
    B, K = 64, 6
    prob, dt, damping = nc.build("g1_c3", nm, B)
    stand = model.key_qpos[0]
    q0, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(11), B, base_q=stand)
    qK, vK, st = prob.solve(q0, tg, stand[None, :], None, dt, damping, n_steps=K)
    assert (st == 0).all()
    # (a) the same loop driven from the host, one launch per step
    q = q0.copy()
    for _ in range(K):
        v, s1 = prob.solve(q, tg, stand[None, :], None, dt, damping)
        q = nm.integrate(q, v, dt)
    # (two different kernel variants — fused-step and single-step — compiled from the same source: not bitwise)
    np.testing.assert_allclose(qK, q, rtol=0, atol=1e-10)
    np.testing.assert_allclose(vK, v, rtol=0, atol=1e-9)
    # (b) the oracle's loop on a few instances
    m = oc.model("g1")
    for i in (0, 17, 63):
        cfg = oik.Configuration(m, q0[i])
        for _ in range(K):
            mm, tasks, limits, dt_o, damp_o = oc.g1_c3(tg[i], stand)
            v_ref = oik.solve_ik(m, cfg, tasks, dt_o, damp_o, limits)
            cfg.update(cfg.integrate(v_ref, dt_o))
        np.testing.assert_allclose(qK[i], cfg.q, rtol=0, atol=1e-10)
        np.testing.assert_allclose(vK[i], v_ref, rtol=0, atol=1e-7 * max(1.0, np.abs(v_ref).max()))


def test_api_convergence_in_one_launch():
    """reference tests/test_solve_ik.py:95-148 with the loop on the device"""
    m = mink.load_robot("ur5e")
    cfg = mink.Configuration(m)
    cfg.update_from_keyframe("home")
    task = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0)
    target = cfg.get_transform_frame_to_world("attachment_site", "site") @ mink.SE3.from_translation(np.array([0, 0, 0.1]))
    task.set_target(target)
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: np.pi for n in m.jnt_names})]
    q, v = mink.solve_ik_steps(cfg, [task], 5e-3, 20, "mi355x", limits=lims)
    assert np.allclose(v, 0.0, atol=1e-6)
    assert np.linalg.norm(task.compute_error(cfg)) < 1e-6
    np.testing.assert_allclose(cfg.get_transform_frame_to_world("attachment_site", "site").as_matrix(),
                               target.as_matrix(), atol=1e-6)


def test_threshold_terminated_loop_matches_the_callers_loop():
    """mkh_solve_until = the loop of examples/arm_ur5e_actuators.py:88-97 (solve, integrate, compute_error, break on
    pos/ori thresholds, max_iters) per instance: final q, last v, iteration count and converged flag against the
    same loop run on the oracle."""
    model = workloads.load_robot("ur5e")
    om = oc.model("ur5e")
    nm = nat.NativeModel(model)
    B, max_iters, pos_thr, ori_thr = 48, 20, 1e-4, 1e-4
    prob, dt, damping = nc.build("ur5e_c2", nm, B)
    home = model.key_qpos[0]
    rng = np.random.default_rng(21)
    q0 = np.tile(home, (B, 1)) + rng.normal(scale=0.05, size=(B, model.nq))
    # targets at very different distances: some converge in 2-3 iterations, some never within max_iters
    scale = np.repeat([1e-3, 1e-2, 0.05, 0.3], B // 4)[:, None]
    q_t = nm.integrate(q0, rng.normal(size=(B, model.nv)) * scale, 1.0)
    dummy = np.zeros((B, 1, 7)); dummy[:, :, 0] = 1
    _, _, t = prob.solve(q_t, dummy, home[None, :], None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
    tg = t["frame_pose"]
    dt = 2e-2                                                   # (a coarser step than the example's 2 ms: more spread in the counts)
    q, v, st, iters, conv = prob.solve(q0, tg, home[None, :], None, dt, damping, n_steps=max_iters, until=(pos_thr, ori_thr))
    assert prob.last_kernel() == "ik_quad_kernel_loop", prob.last_kernel()   # default for a small arm below 28 672 instances
    assert (st & ~1 == 0).all()
    print("iterations:", np.bincount(iters, minlength=max_iters + 1).tolist(), "converged:", int(conv.sum()), "of", B)
    assert conv.sum() >= B // 4 and (conv == 0).sum() >= 4 and len(set(iters[conv == 1].tolist())) >= 3
    for i in range(B):
        cfg = oik.Configuration(om, q0[i])
        m, tasks, limits, _, damp_o = oc.ur5e_c2(tg[i], home)
        done, n = False, 0
        for n in range(1, max_iters + 1):
            v_ref = oik.solve_ik(om, cfg, tasks, dt, damp_o, limits)
            cfg.update(cfg.integrate(v_ref, dt))
            err = oik.task_error_jacobian(cfg, tasks[0])[0]
            if np.linalg.norm(err[:3]) <= pos_thr and np.linalg.norm(err[3:]) <= ori_thr:
                done = True
                break
        assert (iters[i], bool(conv[i])) == (n, done), (i, iters[i], conv[i], n, done)
        np.testing.assert_allclose(q[i], cfg.q, rtol=0, atol=1e-10)
        np.testing.assert_allclose(v[i], v_ref, rtol=0, atol=1e-7 * max(1.0, np.abs(v_ref).max()))
    # the same loop in the lane-per-problem kernel (small arms, large batches) and in the wavefront kernel: identical
    # iteration counts and flags
    for kw, kernel in (({"lane_kernel": True}, "ik_lane_kernel_6_loop"), ({"wave_kernel": True}, "ik_solve_kernel_8_16")):
        ql, vl, stl, itl, cvl = prob.solve(q0, tg, home[None, :], None, dt, damping, n_steps=max_iters, until=(pos_thr, ori_thr), **kw)
        assert prob.last_kernel() == kernel, prob.last_kernel()
        np.testing.assert_array_equal(itl, iters); np.testing.assert_array_equal(cvl, conv); np.testing.assert_array_equal(stl, st)
        np.testing.assert_allclose(ql, q, rtol=0, atol=1e-10)
        np.testing.assert_allclose(vl, v, rtol=0, atol=1e-7 * max(1.0, np.abs(v).max()))
    # the public API
    cfg = mink.Configuration(model, q0)
    ft = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    ft.set_target(mink.SE3(tg[:, 0]))
    post = mink.PostureTask(model, cost=1e-2); post.set_target(home)
    lims = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, {n: np.pi for n in model.jnt_names})]
    q2, v2, it2, cv2 = mink.solve_ik_steps(cfg, [ft, post], dt, max_iters, damping=damping, limits=lims,
                                           pos_threshold=pos_thr, ori_threshold=ori_thr)
    np.testing.assert_array_equal(it2, iters); np.testing.assert_array_equal(cv2, conv.astype(bool))
    np.testing.assert_allclose(q2, q, rtol=0, atol=1e-12)
    # fixed-count calls are untouched by the feature
    qf, vf, stf = prob.solve(q0, tg, home[None, :], None, dt, damping, n_steps=3)
    assert prob.last_kernel() == "ik_quad_kernel_loop"
    qs = q0.copy()
    for _ in range(3):
        vs, _ = prob.solve(qs, tg, home[None, :], None, dt, damping)
        qs = nm.integrate(qs, vs, dt)
    np.testing.assert_allclose(qf, qs, rtol=0, atol=1e-10)
    np.testing.assert_allclose(vf, vs, rtol=0, atol=1e-8 * max(1.0, np.abs(vs).max()))
    for kw, kernel in (({"lane_kernel": True}, "ik_lane_kernel_6_loop"), ({"wave_kernel": True}, "ik_solve_kernel_8_16")):
        qfl, vfl, stfl = prob.solve(q0, tg, home[None, :], None, dt, damping, n_steps=3, **kw)
        assert prob.last_kernel() == kernel
        np.testing.assert_allclose(qfl, qs, rtol=0, atol=1e-10)
        np.testing.assert_allclose(vfl, vs, rtol=0, atol=1e-8 * max(1.0, np.abs(vs).max()))


def test_warm_start_across_calls_gives_the_cold_answers():
    """MKH_FLAG_WARM_START (solve_ik(..., warm_start=True)): a closed loop of single solves on the same batch — every step's v
    equals the cold solve's (the optimum is unique; the pivot order differs, so to rounding), fewer pivots once the loop runs."""
    import time
    import native_configs as nc
    from mink_amd import _native as nat
    from mink_amd import workloads
    model = workloads.load_robot("g1")
    nm = nat.NativeModel(model)
    B = 2048
    prob, dt, damping = nc.build("g1_c3", nm, B)
    cold, _, _ = nc.build("g1_c3", nm, B)
    stand = model.key_qpos[0]
    q, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(8), B, base_q=stand)
    qw = q.copy()
    worst = 0.0
    for step in range(12):
        vw, stw = prob.solve(qw, tg, stand[None, :], None, dt, damping, warm_start=True)
        vc, stc = cold.solve(qw, tg, stand[None, :], None, dt, damping)
        assert (stw & ~1 == 0).all() and (stc & ~1 == 0).all()
        worst = max(worst, np.abs(vw - vc).max() / max(1.0, np.abs(vc).max()))
        qw = nm.integrate(qw, vw, dt)
    print("closed loop of 12 warm-started solves vs cold solves: max rel |dv| = %.2e" % worst)
    assert worst < 1e-9
    # a different batch size resets the state; a permuted batch still gets the right answers (only slower)
    perm = np.random.default_rng(0).permutation(B)
    vp, stp = prob.solve(qw[perm], tg[perm], stand[None, :], None, dt, damping, warm_start=True)
    vc, _ = cold.solve(qw[perm], tg[perm], stand[None, :], None, dt, damping)
    assert np.abs(vp - vc).max() / max(1.0, np.abs(vc).max()) < 1e-9
    vh, sth = prob.solve(qw[:100], tg[:100], stand[None, :], None, dt, damping, warm_start=True)
    vc, _ = cold.solve(qw[:100], tg[:100], stand[None, :], None, dt, damping)
    assert np.abs(vh - vc).max() / max(1.0, np.abs(vc).max()) < 1e-9


def test_fused_loops_of_the_g1_full_example_on_ten_wavefronts_per_cu():
    """Round 5: the F_COM builds of the low-rank start (ComTask rows, up to 24 task rows) on the one-more-wave register map —
    `44_52_r44_w3` runs the fused loops of the reference's humanoid example.  Same loop on the two-waves build, and as single
    solves + integrate."""
    model = workloads.load_robot("g1")
    nm = nat.NativeModel(model)
    stand = model.key_qpos[0]
    B = 1024
    prob, dt, damping = nc.build("g1_full", nm, B)
    q0, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(9), B, base_q=stand)
    _, _, t = prob.solve(q0, tg, stand[None, :], np.zeros((1, 3)), dt, damping, taps=["subtree_com"], solve_qp=False)
    com = t["subtree_com"][:, None, :] + 0.01
    qf, vf, stf = prob.solve(q0, tg, stand[None, :], com, dt, damping, n_steps=4)
    assert prob.last_kernel() == "ik_solve_kernel_44_52_r44_w3", prob.last_kernel()
    q2, v2, st2 = prob.solve(q0, tg, stand[None, :], com, dt, damping, n_steps=4, two_waves=True)
    assert prob.last_kernel() == "ik_solve_kernel_44_52_r44", prob.last_kernel()
    assert ((stf & ~1) == 0).all() and (st2 == stf).all()
    np.testing.assert_allclose(qf, q2, rtol=0, atol=1e-10)
    np.testing.assert_allclose(vf, v2, rtol=0, atol=1e-8 * max(1.0, np.abs(v2).max()))
    qs = q0.copy()
    for _ in range(4):
        vs, _ = prob.solve(qs, tg, stand[None, :], com, dt, damping)
        qs = nm.integrate(qs, vs, dt)
    np.testing.assert_allclose(qf, qs, rtol=0, atol=1e-10)
    np.testing.assert_allclose(vf, vs, rtol=0, atol=1e-8 * max(1.0, np.abs(vs).max()))
    # the threshold-terminated loop: iteration counts and flags of the two builds agree
    out3 = prob.solve(q0, tg, stand[None, :], com, dt, damping, n_steps=6, until=(5e-2, 2e-1))
    assert prob.last_kernel() == "ik_solve_kernel_44_52_r44_w3", prob.last_kernel()
    out2 = prob.solve(q0, tg, stand[None, :], com, dt, damping, n_steps=6, until=(5e-2, 2e-1), two_waves=True)
    np.testing.assert_array_equal(out3[3], out2[3]); np.testing.assert_array_equal(out3[4], out2[4])
    np.testing.assert_allclose(out3[0], out2[0], rtol=0, atol=1e-10)
