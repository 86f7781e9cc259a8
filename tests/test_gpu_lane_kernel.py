"""The lane-per-problem kernel for small arms (mink_amd/csrc/lane_kernel.h): one lane solves one problem start to
finish, 64 problems per wavefront.  Parity against the real-mink fixture, both oracles and the wavefront kernel
(MKH_FLAG_WAVE_KERNEL) on the same inputs, including heavily saturated boxes (many block-pivoting iterations),
7-dof arms, slide joints, status bits and batch-permutation equivariance."""

import os

import numpy as np
import pytest

import native_configs as nc
import oracle_configs as oc
from mink_amd import workloads
from mink_amd.flatmodel import FlatModel
from oracle import cport
from oracle import ik as oik

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from mink_amd import _native
    assert _native.lib().mkh_device_count() >= 1
    return _native


def test_real_mink_fixture(nat):
    d = np.load(os.path.join(oc.GOLDEN, "ik_ur5e_c2.npz"))
    m = oc.model("ur5e")
    nm = nat.NativeModel(m)
    B = len(d["q"])
    prob, dt, damping = nc.build("ur5e_c2", nm, B)
    v, st = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping, lane_kernel=True)
    assert prob.last_kernel() == "ik_lane_kernel_6", prob.last_kernel()
    assert (st & ~1 == 0).all()
    main = np.ones(B, bool); main[7::8] = False
    err = np.abs(v - d["v"]) / np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    print("lane kernel vs real-mink fixture: max rel err main %.1e small-angle %.1e" % (err[main].max(), err[~main].max()))
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5
    vw, stw = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping, wave_kernel=True)
    assert prob.last_kernel() == "ik_solve_kernel_8_0" and (stw == st).all()
    # default dispatch: by batch size (below 73 728 plain solves go to the row kernel, tests/test_gpu_quad_kernel.py)
    prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping)
    assert prob.last_kernel() == "ik_quad_kernel"
    big = nc.build("ur5e_c2", nm, 73728)[0]
    rep = 73728 // B
    vb, _ = big.solve(np.tile(d["q"], (rep, 1)), np.tile(d["frame_targets"], (rep, 1, 1)), d["posture_target"][None, :], None, dt, damping)
    assert big.last_kernel() == "ik_lane_kernel_6" and np.array_equal(vb[:B], v) and np.array_equal(vb[-B:], v)


@pytest.mark.parametrize("vmax,dt", [(np.pi, 2e-3), (0.3, 5e-2)])
def test_ur5e_batch_vs_wave_kernel_and_c_oracle(nat, vmax, dt):
    """B = 4096 (BASELINE config 2); the second parameter set saturates most dofs: the box QP takes several
    block-pivoting rounds, lanes of one wavefront finish at different iterations."""
    m = workloads.load_robot("ur5e")
    om = oc.model("ur5e")
    nm = nat.NativeModel(m)
    B = 4096 + 37                                     # a ragged last wavefront
    idx = [int(m.jnt_dofadr[j]) for j in range(m.njnt)]
    prob = nat.NativeProblem(nm, frame_tasks=[nc._ft(m, "attachment_site", "site", 1.0, 1.0, 1.0)],
                             posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                             velocity_limits=[{"indices": idx, "limit": np.full(6, vmax)}], max_batch=B)
    home = m.key_qpos[0]
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(4), B, base_q=home)
    q[5] = home; q[5, 2] = 3.1415 + 2e-3                # outside the elbow range [-3.1415, 3.1415]: status bit 1, still solved
    v, st = prob.solve(q, tg, home[None, :], None, dt, 1e-3, lane_kernel=True)
    assert prob.last_kernel() == "ik_lane_kernel_6"
    assert st[5] == 1 and (np.delete(st, 5) == 0).all()
    vw, stw = prob.solve(q, tg, home[None, :], None, dt, 1e-3, wave_kernel=True)
    assert (stw == st).all()
    scale = np.maximum(1.0, np.abs(vw).max(axis=1, keepdims=True))
    print("lane vs wavefront kernel: max rel diff %.1e; saturated dofs per instance %.2f" %
          ((np.abs(v - vw) / scale).max(), (np.abs(np.abs(v) - vmax) < 1e-9).sum() / B))
    assert (np.abs(v - vw) / scale).max() < 1e-9
    mm, tasks, _, _, _ = oc.ur5e_c2(tg[0], home)
    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.array(idx), np.full(6, vmax))]
    v_c, st_c = cport.CProblem(om, tasks, limits).solve_batch(q, tg, home[None, :], dt, 1e-3)
    assert (st_c == 0).all()
    err = (np.abs(v - v_c) / np.maximum(1.0, np.abs(v_c).max(axis=1, keepdims=True))).max()
    print("lane kernel vs C oracle on %d instances: max rel err %.1e" % (B, err))
    assert err < 1e-8
    # bitwise batch-permutation equivariance: a problem's answer does not depend on its lane or wavefront
    perm = np.random.default_rng(0).permutation(B)
    vp, _ = prob.solve(q[perm], tg[perm], home[None, :], None, dt, 1e-3, lane_kernel=True)
    assert np.array_equal(vp, v[perm])


@pytest.mark.parametrize("scene,kernel", [("kuka_iiwa_14__scene", "ik_lane_kernel_7"), ("ufactory_xarm7__scene", "ik_lane_kernel_7"),
                                          ("stanford_tidybot__scene_base", "ik_lane_kernel_4"),
                                          ("stanford_tidybot__scene_mobile_kinova", "ik_quad_kernel_16")])
def test_other_small_robots(nat, scene, kernel):
    """7-dof arms (NV = 7), a 3-dof base with slide joints and a body frame (NV = 4); the 10-dof mobile arm does not
    qualify for the lane kernel (nv > 8): MKH_FLAG_LANE_KERNEL leaves it where the dispatch puts it, on the row kernel."""
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", scene + ".json"))
    nm = nat.NativeModel(m)
    B = 512
    sites = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0]
    frame = ("site", sites[-1]) if sites else ("body", int(np.argmax(m.body_depth)))
    ft = {"frame_type": frame[0], "frame_id": frame[1], "cost": [1.0, 1.0, 1.0, 0.3, 0.3, 0.3], "gain": 0.9, "lm_damping": 0.5}
    vidx = [int(m.jnt_dofadr[j]) for j in range(m.njnt)]
    vlim = np.where([m.jnt_type[j] == 2 for j in range(m.njnt)], 0.2, 1.0)
    prob = nat.NativeProblem(nm, frame_tasks=[ft], posture_tasks=[{"cost": 3e-2, "gain": 0.5, "lm_damping": 0.1}],
                             configuration_limits=[nc._cfg_limit(m)], velocity_limits=[{"indices": vidx, "limit": vlim}],
                             max_batch=B)
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(9), B, base_q=m.qpos0)
    ptg = np.tile(m.qpos0, (B, 1, 1)) + np.random.default_rng(1).normal(scale=0.1, size=(B, 1, m.nq))   # per-instance posture target
    dt, damping = 2e-2, 1e-4
    v, st = prob.solve(q, tg, ptg, None, dt, damping, lane_kernel=True)
    assert prob.last_kernel() == kernel, prob.last_kernel()
    assert (st & ~1 == 0).all()
    tasks = [oik.FrameTaskSpec(frame[1], frame[0], np.array(ft["cost"]), tg[0, 0], 0.9, 0.5),
             oik.PostureTaskSpec(np.full(m.nv, 3e-2), ptg[0, 0], 0.5, 0.1)]
    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.array(vidx), vlim)]
    v_c, st_c = cport.CProblem(m, tasks, limits).solve_batch(q, tg, ptg, dt, damping)
    assert (st_c == 0).all()
    err = (np.abs(v - v_c) / np.maximum(1.0, np.abs(v_c).max(axis=1, keepdims=True))).max()
    print("%s (%s): max rel err vs C oracle %.1e" % (scene, kernel, err))
    assert err < 1e-8
    for i in (0, 77, B - 1):
        tasks[0].target = tg[i, 0]; tasks[1].target_q = ptg[i, 0]
        v_ref = oik.solve_ik(m, q[i], tasks, dt, damping, limits)
        assert np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()) < 1e-8


def test_failure_status(nat):
    """Inconsistent box (velocity window that excludes the configuration-limit window) → MKH_ST_INFEASIBLE and NaN,
    like the wavefront kernel (quadprog: "constraints are inconsistent")."""
    m = workloads.load_robot("ur5e")
    nm = nat.NativeModel(m)
    B = 64
    prob, dt, damping = nc.build("ur5e_c2", nm, B)
    home = m.key_qpos[0]
    q = np.tile(home, (B, 1))
    q[3, 2] = 3.1415 + 0.5                             # far outside the range: the box lo ≤ Δq ≤ hi becomes empty with the velocity limit
    tg = np.zeros((B, 1, 7)); tg[:, :, 0] = 1; tg[:, :, 4:] = [0.4, 0.1, 0.4]
    v, st = prob.solve(q, tg, home[None, :], None, dt, damping, lane_kernel=True)
    assert prob.last_kernel() == "ik_lane_kernel_6"
    vw, stw = prob.solve(q, tg, home[None, :], None, dt, damping, wave_kernel=True)
    assert prob.last_kernel() == "ik_solve_kernel_8_0"
    assert (st == stw).all() and st[3] & 2 and np.isnan(v[3]).all() and np.isfinite(np.delete(v, 3, axis=0)).all()


def test_warm_start_across_calls_on_the_lane_kernel(nat):
    """MKH_FLAG_WARM_START on the lane kernel (round 6; it used to start cold): a closed loop of single solves of a UR5e batch whose
    velocity limits bind on most dofs — every step's v equals the cold solve's (the optimum is unique) whatever partition the previous
    call left in the handle; a permuted batch (a wrong prediction for every instance) and a changed batch size (state reset) still
    give the right answers."""
    m = workloads.load_robot("ur5e")
    nm = nat.NativeModel(m)
    B = 1500
    idx = [int(m.jnt_dofadr[j]) for j in range(m.njnt)]
    kw = dict(frame_tasks=[nc._ft(m, "attachment_site", "site", 1.0, 1.0, 1.0)], posture_tasks=[{"cost": 1e-2}],
              configuration_limits=[nc._cfg_limit(m)], velocity_limits=[{"indices": idx, "limit": np.full(6, 0.3)}], max_batch=B)
    prob, cold = nat.NativeProblem(nm, **kw), nat.NativeProblem(nm, **kw)
    home = m.key_qpos[0]
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(8), B, base_q=home)
    pt = home[None, :]
    dt, damping = 5e-2, 1e-3
    rel = lambda a, b: np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))
    qw, worst = q.copy(), 0.0
    for step in range(10):
        vw, stw = prob.solve(qw, tg, pt, None, dt, damping, warm_start=True, lane_kernel=True)
        assert prob.last_kernel() == "ik_lane_kernel_6", prob.last_kernel()
        vc, stc = cold.solve(qw, tg, pt, None, dt, damping, lane_kernel=True)
        assert (stw == stc).all() and (stw & ~1 == 0).all()
        worst = max(worst, rel(vw, vc).max())
        qw = nm.integrate(qw, vw, dt)
    print("UR5e: closed loop of 10 warm-started lane-kernel solves vs cold solves: max rel |dv| = %.2e" % worst)
    assert worst < 1e-9
    perm = np.random.default_rng(0).permutation(B)
    vp, _ = prob.solve(qw[perm], tg[perm], pt, None, dt, damping, warm_start=True, lane_kernel=True)
    vc, _ = cold.solve(qw[perm], tg[perm], pt, None, dt, damping, lane_kernel=True)
    assert rel(vp, vc).max() < 1e-9
    vh, _ = prob.solve(qw[:100], tg[:100], pt, None, dt, damping, warm_start=True, lane_kernel=True)
    assert rel(vh, cold.solve(qw[:100], tg[:100], pt, None, dt, damping, lane_kernel=True)[0]).max() < 1e-9
