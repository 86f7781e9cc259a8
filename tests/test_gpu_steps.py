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
