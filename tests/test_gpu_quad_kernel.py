"""The row-per-problem kernel for small arms (mink_amd/csrc/quad_kernel.h): one 16-lane DPP row solves one problem,
four problems per wavefront.  Parity against the real-mink fixture, both oracles, the wavefront kernel
(MKH_FLAG_WAVE_KERNEL) and the lane kernel (MKH_FLAG_LANE_KERNEL) on the same inputs: heavily saturated boxes (many
principal pivots in and out of the free set), 7-dof arms, slide joints, branching trees, several frame tasks, status bits,
ragged last wavefront, batch-permutation equivariance, and the default dispatch by batch size."""

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
QUAD = "ik_quad_kernel"


@pytest.fixture(scope="module")
def nat():
    from mink_amd import _native
    assert _native.lib().mkh_device_count() >= 1
    return _native


def _rel(v, ref):
    return np.abs(v - ref) / np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))


def test_real_mink_fixture(nat):
    d = np.load(os.path.join(oc.GOLDEN, "ik_ur5e_c2.npz"))
    m = oc.model("ur5e")
    nm = nat.NativeModel(m)
    B = len(d["q"])
    prob, dt, damping = nc.build("ur5e_c2", nm, B)
    v, st = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping, quad_kernel=True)
    assert prob.last_kernel() == QUAD, prob.last_kernel()
    assert (st & ~1 == 0).all()
    main = np.ones(B, bool); main[7::8] = False
    err = _rel(v, d["v"])
    print("quad kernel vs real-mink fixture: max rel err main %.1e small-angle %.1e" % (err[main].max(), err[~main].max()))
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5
    vw, stw = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping, wave_kernel=True)
    assert prob.last_kernel() == "ik_solve_kernel_8_0" and (stw == st).all()
    assert _rel(v, vw).max() < 1e-9
    # default dispatch of a plain solve: by batch size — row kernel, then lane kernel
    for n, kernel in ((1, QUAD), (255, QUAD), (4096, QUAD), (8192, QUAD), (73727, QUAD), (73728, "ik_lane_kernel_6")):
        p = nc.build("ur5e_c2", nm, n)[0]
        rep = -(-n // B)
        qn, tn = np.tile(d["q"], (rep, 1))[:n], np.tile(d["frame_targets"], (rep, 1, 1))[:n]
        vn, stn = p.solve(qn, tn, d["posture_target"][None, :], None, dt, damping)
        assert p.last_kernel() == kernel, (n, p.last_kernel())
        if kernel == QUAD:                           # bitwise: a problem's answer does not depend on its row / wavefront
            k = min(n, B)
            assert np.array_equal(vn[:k], v[:k])
            if n > B:
                r = n % B or B
                assert np.array_equal(vn[n - r:], v[:r])
    # taps are not this kernel's: the call falls through to the wavefront kernel; fused loops are
    # its own below 8 192 instances (tests/test_gpu_steps.py, test_fused_loop_* below)
    p = nc.build("ur5e_c2", nm, 1024)[0]
    qn, tn = np.tile(d["q"], (32, 1))[:1024], np.tile(d["frame_targets"], (32, 1, 1))[:1024]
    p.solve(qn, tn, d["posture_target"][None, :], None, dt, damping, n_steps=3)
    assert p.last_kernel() == "ik_quad_kernel_loop", p.last_kernel()
    p.solve(qn, tn, d["posture_target"][None, :], None, dt, damping, taps=("H",))
    assert p.last_kernel().removesuffix("+wide").endswith("_31"), p.last_kernel()
    vws, _ = p.solve(qn, tn, d["posture_target"][None, :], None, dt, damping, warm_start=True)
    assert p.last_kernel() == QUAD, p.last_kernel()      # (warm starts are its own: test_warm_start_across_calls below)
    assert np.array_equal(vws[:B], v)                    # (the handle's first warm call starts cold)


@pytest.mark.parametrize("vmax,dt", [(np.pi, 2e-3), (0.3, 5e-2)])
def test_ur5e_batch_vs_other_kernels_and_c_oracle(nat, vmax, dt):
    """B = 4096 (BASELINE config 2) + a ragged last wavefront; the second parameter set saturates most dofs: several
    rounds of principal pivots, rows of one wavefront finish at different iterations."""
    m = workloads.load_robot("ur5e")
    om = oc.model("ur5e")
    nm = nat.NativeModel(m)
    B = 4096 + 3
    idx = [int(m.jnt_dofadr[j]) for j in range(m.njnt)]
    prob = nat.NativeProblem(nm, frame_tasks=[nc._ft(m, "attachment_site", "site", 1.0, 1.0, 1.0)],
                             posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                             velocity_limits=[{"indices": idx, "limit": np.full(6, vmax)}], max_batch=B)
    home = m.key_qpos[0]
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(4), B, base_q=home)
    q[5] = home; q[5, 2] = 3.1415 + 2e-3                # outside the elbow range [-3.1415, 3.1415]: status bit 1, still solved
    v, st = prob.solve(q, tg, home[None, :], None, dt, 1e-3)
    assert prob.last_kernel() == QUAD
    assert st[5] == 1 and (np.delete(st, 5) == 0).all()
    vw, stw = prob.solve(q, tg, home[None, :], None, dt, 1e-3, wave_kernel=True)
    vl, stl = prob.solve(q, tg, home[None, :], None, dt, 1e-3, lane_kernel=True)
    assert (stw == st).all() and (stl == st).all()
    print("quad vs wavefront kernel: max rel diff %.1e, vs lane kernel %.1e; saturated dofs per instance %.2f" %
          (_rel(v, vw).max(), _rel(v, vl).max(), (np.abs(np.abs(v) - vmax) < 1e-9).sum() / B))
    assert _rel(v, vw).max() < 1e-9 and _rel(v, vl).max() < 1e-9
    mm, tasks, _, _, _ = oc.ur5e_c2(tg[0], home)
    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.array(idx), np.full(6, vmax))]
    v_c, st_c = cport.CProblem(om, tasks, limits).solve_batch(q, tg, home[None, :], dt, 1e-3)
    assert (st_c == 0).all()
    err = _rel(v, v_c).max()
    print("quad kernel vs C oracle on %d instances: max rel err %.1e" % (B, err))
    assert err < 1e-8
    perm = np.random.default_rng(0).permutation(B)
    vp, _ = prob.solve(q[perm], tg[perm], home[None, :], None, dt, 1e-3)
    assert np.array_equal(vp, v[perm])


@pytest.mark.parametrize("scene,kernel", [("kuka_iiwa_14__scene", QUAD), ("ufactory_xarm7__scene", QUAD),
                                          ("stanford_tidybot__scene_base", QUAD),
                                          ("stanford_tidybot__scene_mobile_kinova", QUAD + "_16"),
                                          ("leap_hand__scene_right", QUAD + "_16"), ("leap_hand__scene_left", QUAD + "_16"),
                                          ("stanford_tidybot__scene", QUAD + "_32"), ("wonik_allegro__scene_left", QUAD + "_32"),
                                          ("unitree_go1__scene", QUAD + "_32"), ("boston_dynamics_spot__scene", QUAD + "_32"),
                                          ("unitree_h1__scene", QUAD + "_32"), ("shadow_hand__scene_right", QUAD + "_32"),
                                          ("unitree_g1__scene", "ik_solve_kernel")])
def test_other_small_robots(nat, scene, kernel):
    """7-dof arms, a 3-dof base with slide joints and a body frame; 9 … 16 dofs on sixteen column registers: the 10-dof mobile
    arm (three joints on its base body) and the 16-dof LEAP hands (a tree of four fingers); 17 … 32 dofs on TWO DPP rows per
    problem (round 4): the 18-dof Tidybot, the Allegro and Shadow hands, and the floating bases — Go1, Spot, H1 (a free joint =
    three slide links + a quaternion link); the 43-dof G1 stays on the wavefront kernel."""
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", scene + ".json"))
    nm = nat.NativeModel(m)
    B = 514
    sites = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0]
    frame = ("site", sites[-1]) if sites else ("body", int(np.argmax(m.body_depth)))
    ft = {"frame_type": frame[0], "frame_id": frame[1], "cost": [1.0, 1.0, 1.0, 0.3, 0.3, 0.3], "gain": 0.9, "lm_damping": 0.5}
    hs = [j for j in range(m.njnt) if m.jnt_type[j] in (2, 3)]
    vidx = [int(m.jnt_dofadr[j]) for j in hs]
    vlim = np.where([m.jnt_type[j] == 2 for j in hs], 0.2, 1.0)
    prob = nat.NativeProblem(nm, frame_tasks=[ft], posture_tasks=[{"cost": 3e-2, "gain": 0.5, "lm_damping": 0.1}],
                             configuration_limits=[nc._cfg_limit(m)], velocity_limits=[{"indices": vidx, "limit": vlim}],
                             max_batch=B)
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(9), B, base_q=m.qpos0)
    ptg = np.tile(m.qpos0, (B, 1, 1)) + np.random.default_rng(1).normal(scale=0.1, size=(B, 1, m.nq))   # per-instance posture target
    dt, damping = 2e-2, 1e-4
    v, st = prob.solve(q, tg, ptg, None, dt, damping)
    assert prob.last_kernel() == kernel or (kernel == "ik_solve_kernel" and prob.last_kernel().startswith(kernel)), prob.last_kernel()
    assert (st & ~1 == 0).all()
    if kernel.startswith(QUAD):
        vw, stw = prob.solve(q, tg, ptg, None, dt, damping, wave_kernel=True)
        assert prob.last_kernel().startswith("ik_solve_kernel") and (stw == st).all() and _rel(v, vw).max() < 1e-8
    tasks = [oik.FrameTaskSpec(frame[1], frame[0], np.array(ft["cost"]), tg[0, 0], 0.9, 0.5),
             oik.PostureTaskSpec(np.full(m.nv, 3e-2), ptg[0, 0], 0.5, 0.1)]
    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.array(vidx), vlim)]
    v_c, st_c = cport.CProblem(m, tasks, limits).solve_batch(q, tg, ptg, dt, damping)
    assert (st_c == 0).all()
    err = _rel(v, v_c).max()
    print("%s (%s): max rel err vs C oracle %.1e" % (scene, prob.last_kernel(), err))
    assert err < 1e-8
    for i in (0, 77, B - 1):
        tasks[0].target = tg[i, 0]; tasks[1].target_q = ptg[i, 0]
        v_ref = oik.solve_ik(m, q[i], tasks, dt, damping, limits)
        assert np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()) < 1e-8


def test_branching_tree_and_several_frame_tasks(nat):
    """A two-fingered 8-dof tree (slide base, hinge arm, two branches) with a frame task on each branch tip, one on the
    palm and one with zero orientation cost: pointer jumping over a tree whose links have different depths, chains that
    share dofs, rows left out of H by their zero cost."""
    from mink_amd.mjcf import loads_mjcf
    xml = """
    <mujoco><worldbody>
      <body name="base" pos="0 0 0.1"><joint name="s" type="slide" axis="1 0 0" range="-1 1"/><geom size="0.05"/>
        <body name="l1" pos="0 0 0.2"><joint name="h1" type="hinge" axis="0 1 0" range="-2 2"/><geom size="0.04"/>
          <body name="l2" pos="0.3 0 0" quat="0.9239 0 0 0.3827"><joint name="h2" type="hinge" axis="0 0 1" pos="0.01 0.02 0" range="-2 2"/><geom size="0.04"/>
            <body name="palm" pos="0.25 0 0"><joint name="h3" type="hinge" axis="1 0 0" range="-2 2"/><geom size="0.03"/>
              <site name="palm_site" pos="0.02 0 0.01" quat="0.7071 0 0.7071 0"/>
              <body name="fa1" pos="0.05 0.03 0"><joint name="a1" type="hinge" axis="0 0 1" range="-1 1"/><geom size="0.01"/>
                <body name="fa2" pos="0.06 0 0"><joint name="a2" type="hinge" axis="0 0 1" range="-1 1"/><geom size="0.01"/>
                  <site name="tip_a" pos="0.04 0 0"/></body></body>
              <body name="fb1" pos="0.05 -0.03 0"><joint name="b1" type="hinge" axis="0 0 1" range="-1 1"/><geom size="0.01"/>
                <body name="fb2" pos="0.06 0 0"><joint name="b2" type="slide" axis="1 0 0" range="-0.03 0.03"/><geom size="0.01"/>
                  <site name="tip_b" pos="0.04 0 0"/></body></body>
            </body></body></body></body>
    </worldbody></mujoco>"""
    m = loads_mjcf(xml)
    assert m.nv == 8
    nm = nat.NativeModel(m)
    B = 640
    sid = {n: i for i, n in enumerate(m.site_names)}
    fts = [{"frame_type": "site", "frame_id": sid["tip_a"], "cost": [1.0, 1.0, 1.0, 0.0, 0.0, 0.0], "gain": 1.0, "lm_damping": 1.0},
           {"frame_type": "site", "frame_id": sid["tip_b"], "cost": [1.0, 1.0, 1.0, 0.2, 0.2, 0.2], "gain": 0.8, "lm_damping": 0.0},
           {"frame_type": "site", "frame_id": sid["palm_site"], "cost": [0.5, 0.5, 0.5, 1.0, 1.0, 1.0], "gain": 1.0, "lm_damping": 0.3},
           {"frame_type": "body", "frame_id": m.body_names.index("l2"), "cost": [0.1, 0.1, 0.1, 0.1, 0.1, 0.1], "gain": 0.5, "lm_damping": 0.0}]
    vidx = list(range(8))
    vlim = np.array([0.5, 1.0, 1.0, 1.0, 2.0, 2.0, 2.0, 0.1])
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                             velocity_limits=[{"indices": vidx, "limit": vlim}], max_batch=B)
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(2), B, base_q=m.qpos0)
    dt, damping = 3e-2, 1e-6
    v, st = prob.solve(q, tg, m.qpos0[None, :], None, dt, damping)
    assert prob.last_kernel() == QUAD and (st == 0).all()
    vw, stw = prob.solve(q, tg, m.qpos0[None, :], None, dt, damping, wave_kernel=True)
    vl, stl = prob.solve(q, tg, m.qpos0[None, :], None, dt, damping, lane_kernel=True)
    assert prob.last_kernel() == "ik_lane_kernel_8" and (stw == 0).all() and (stl == 0).all()
    print("tree, 4 frame tasks: quad vs wavefront %.1e, vs lane %.1e; saturated dofs per instance %.2f" %
          (_rel(v, vw).max(), _rel(v, vl).max(), (np.abs(np.abs(v) - vlim) < 1e-9).sum() / B))
    assert _rel(v, vw).max() < 1e-8 and _rel(v, vl).max() < 1e-8
    tasks = [oik.FrameTaskSpec(f["frame_id"], f["frame_type"], np.array(f["cost"]), tg[0, k], f["gain"], f["lm_damping"]) for k, f in enumerate(fts)]
    tasks.append(oik.PostureTaskSpec(np.full(m.nv, 1e-2), m.qpos0))
    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.array(vidx), vlim)]
    v_c, st_c = cport.CProblem(m, tasks, limits).solve_batch(q, tg, m.qpos0[None, :], dt, damping)
    assert (st_c == 0).all() and _rel(v, v_c).max() < 1e-8


def test_failure_status(nat):
    """Inconsistent box (velocity window that excludes the configuration-limit window) → MKH_ST_INFEASIBLE and NaN for
    that row only, like the other kernels (quadprog: "constraints are inconsistent")."""
    m = workloads.load_robot("ur5e")
    nm = nat.NativeModel(m)
    B = 66
    prob, dt, damping = nc.build("ur5e_c2", nm, B)
    home = m.key_qpos[0]
    q = np.tile(home, (B, 1))
    q[3, 2] = 3.1415 + 0.5                             # far outside the range: the box lo ≤ Δq ≤ hi becomes empty with the velocity limit
    tg = np.zeros((B, 1, 7)); tg[:, :, 0] = 1; tg[:, :, 4:] = [0.4, 0.1, 0.4]
    v, st = prob.solve(q, tg, home[None, :], None, dt, damping, quad_kernel=True)
    assert prob.last_kernel() == QUAD
    vw, stw = prob.solve(q, tg, home[None, :], None, dt, damping, wave_kernel=True)
    assert (st == stw).all() and st[3] & 2 and np.isnan(v[3]).all() and np.isfinite(np.delete(v, 3, axis=0)).all()
    assert _rel(np.delete(v, 3, axis=0), np.delete(vw, 3, axis=0)).max() < 1e-9


@pytest.mark.parametrize("until", [False, True])
def test_fused_loop_against_the_other_kernels(nat, until):
    """mkh_solve_steps / mkh_solve_until on the row kernel: final q, last v, status, iteration counts and converged flags
    against the lane kernel's and the wavefront kernel's loops on 1 027 instances (a ragged last wavefront, rows of one
    wavefront finishing at different iterations), including an instance whose box becomes inconsistent mid-loop."""
    m = workloads.load_robot("ur5e")
    nm = nat.NativeModel(m)
    B = 1027
    prob, dt, damping = nc.build("ur5e_c2", nm, B)
    home = m.key_qpos[0]
    rng = np.random.default_rng(8)
    q0 = np.tile(home, (B, 1)) + rng.normal(scale=0.05, size=(B, m.nq))
    scale = np.repeat([1e-3, 1e-2, 0.05, 0.3], -(-B // 4))[:B, None]
    q_t = nm.integrate(q0, rng.normal(size=(B, m.nv)) * scale, 1.0)
    dummy = np.zeros((B, 1, 7)); dummy[:, :, 0] = 1
    _, _, t = prob.solve(q_t, dummy, home[None, :], None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
    tg = t["frame_pose"]
    q0[11, 2] = 3.1415 + 0.5                           # outside the range by more than a step: inconsistent box, NaN, status 2
    kw = {"n_steps": 12, "until": (1e-4, 1e-4)} if until else {"n_steps": 5}
    ref = prob.solve(q0, tg, home[None, :], None, 2e-2, damping, wave_kernel=True, **kw)
    assert prob.last_kernel() == "ik_solve_kernel_8_16"
    got = prob.solve(q0, tg, home[None, :], None, 2e-2, damping, **kw)
    assert prob.last_kernel() == "ik_quad_kernel_loop"
    lane = prob.solve(q0, tg, home[None, :], None, 2e-2, damping, lane_kernel=True, **kw)
    for other in (ref, lane):
        np.testing.assert_array_equal(got[2], other[2])                                  # status
        ok = (got[2] & 14) == 0
        assert not ok[11] and ok.sum() == B - 1 and np.isnan(got[1][11]).all()
        np.testing.assert_allclose(got[0][ok], other[0][ok], rtol=0, atol=1e-10)           # q
        np.testing.assert_allclose(got[1][ok], other[1][ok], rtol=0, atol=1e-7 * max(1.0, np.abs(other[1][ok]).max()))
        if until:
            np.testing.assert_array_equal(got[3], other[3]); np.testing.assert_array_equal(got[4], other[4])
    if until:
        print("iterations:", np.bincount(got[3], minlength=13).tolist(), "converged:", int(got[4].sum()), "of", B)
        assert 0 < got[4].sum() < B


@pytest.mark.parametrize("seed", range(int(os.environ.get("MKH_FUZZ_SEEDS", "10"))))
def test_random_small_trees_on_all_three_kernels(nat, seed):
    """Random hinge / slide trees with nv ≤ 16 — branching, fixed bodies, two joints on one body, frames on sites and
    bodies, zero cost rows, per-dof posture costs, random gains / LM damping / limits — through the public API on the default
    dispatch (row kernel), then the same compiled problem on the lane and wavefront kernels and the C oracle."""
    import mink_amd as mink
    from random_models import random_mjcf, rand_q
    rng = np.random.default_rng(7000 + seed)
    for _ in range(50):
        nbody = int(rng.integers(2, 9 if seed % 2 else 15))
        xml, sites = random_mjcf(rng, nbody, free_root=False, no_ball=True)
        m = mink.loads_mjcf(xml)
        if 1 <= m.nv <= 16:
            break
    else:
        pytest.skip("no draw with 1 <= nv <= 16")
    Q = QUAD if m.nv <= 8 else QUAD + "_16"
    B = 67
    q = np.stack([rand_q(m, rng) for _ in range(B)])
    cfg = mink.Configuration(m, q)
    tgt_cfg = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.2, size=(B, m.nv)), 1.0))
    frames = [(s, "site") for s in sites] + [(f"b{i}", "body") for i in range(nbody)]
    picks = [frames[i] for i in rng.choice(len(frames), size=min(int(rng.integers(1, 5 if seed % 3 else 9)), len(frames)), replace=False)]
    tasks, specs = [], []
    for name, typ in picks:
        pc = rng.uniform(0.5, 20.0) * (rng.uniform(size=3) < 0.8)
        oc_ = rng.uniform(0.1, 5.0) * (rng.uniform() < 0.6)
        if not pc.any() and oc_ == 0.0:
            pc = np.ones(3)
        gain, lm = float(rng.uniform(0.3, 1.0)), float(rng.uniform(0.0, 1.0))
        ft = mink.FrameTask(name, typ, position_cost=pc, orientation_cost=oc_, gain=gain, lm_damping=lm)
        ft.set_target(tgt_cfg.get_transform_frame_to_world(name, typ))
        tasks.append(ft)
        specs.append((m.name2id(typ, name), typ, ft.cost.copy(), gain, lm))
    post = mink.PostureTask(m, cost=rng.uniform(0.0, 1.0, size=m.nv) * (rng.uniform(size=m.nv) < 0.8), gain=float(rng.uniform(0.2, 1.0)),
                            lm_damping=float(rng.uniform(0.0, 0.5)))
    post.set_target(rand_q(m, rng))
    vel = {m.jnt_names[j]: float(rng.uniform(0.2, 3.0)) for j in range(m.njnt) if rng.uniform() < 0.7}
    lims = [mink.ConfigurationLimit(m, gain=float(rng.uniform(0.5, 1.0)))] + ([mink.VelocityLimit(m, vel)] if vel else [])
    dt, damping = float(rng.choice([2e-3, 1e-2, 5e-2])), float(rng.choice([1e-6, 1e-3, 1e-1]))
    v = mink.solve_ik(cfg, tasks + [post], dt, "mi355x", damping, limits=lims)
    prob = list(cfg._problems.values())[-1]
    if prob.last_kernel().startswith("ik_solve_kernel"):
        pytest.skip("more than 16 links on the frames' chains: wavefront kernel")
    assert prob.last_kernel() == Q, prob.last_kernel()
    ftg = np.stack([ft.transform_target_to_world.wxyz_xyz for ft in tasks], axis=1)
    ptq = post.target_q[None, :]
    vl, stl = prob.solve(q, ftg, ptq, None, dt, damping, lane_kernel=True)      # (no lane kernel above 8 dofs: the row kernel again)
    assert prob.last_kernel().startswith("ik_lane_kernel") == (m.nv <= 8)
    vw, stw = prob.solve(q, ftg, ptq, None, dt, damping, wave_kernel=True)
    assert prob.last_kernel().startswith("ik_solve_kernel") and (stl & ~1 == 0).all() and (stw & ~1 == 0).all()
    ts = [oik.FrameTaskSpec(fid, typ, cost, ft.transform_target_to_world.wxyz_xyz[0], gain, lm) for (fid, typ, cost, gain, lm), ft in zip(specs, tasks)]
    ts.append(oik.PostureTaskSpec(post.cost, post.target_q, post.gain, post.lm_damping))
    ls = [oik.ConfigurationLimitSpec(lims[0].gain)] + ([oik.VelocityLimitSpec(lims[1].indices, lims[1].limit)] if vel else [])
    v_c, st_c = cport.CProblem(m, ts, ls).solve_batch(q, ftg, ptq, dt, damping)
    assert (st_c == 0).all()
    sat = 0 if not vel else int((np.abs(np.abs(v[:, lims[1].indices]) - lims[1].limit) < 1e-9).sum())
    print("seed %d: nv %d, %d bodies, %d frame tasks, saturated velocity bounds %d: row vs wavefront %.1e, vs lane %.1e, vs C oracle %.1e" % (
        seed, m.nv, nbody, len(tasks), sat, _rel(v, vw).max(), _rel(v, vl).max(), _rel(v, v_c).max()))
    assert _rel(v, vw).max() < 1e-8 and _rel(v, vl).max() < 1e-8 and _rel(v, v_c).max() < 1e-7
    # the fused loops of the three kernels on the same tree (4 steps, and until with loose thresholds)
    for kw in ({"n_steps": 4}, {"n_steps": 6, "until": (2e-2, 5e-2)}):
        r = prob.solve(q, ftg, ptq, None, dt, damping, **kw)
        assert prob.last_kernel() == Q + "_loop", prob.last_kernel()
        for other in (prob.solve(q, ftg, ptq, None, dt, damping, wave_kernel=True, **kw),
                      prob.solve(q, ftg, ptq, None, dt, damping, lane_kernel=True, **kw)):
            np.testing.assert_array_equal(r[2], other[2])
            ok = (r[2] & 14) == 0
            np.testing.assert_allclose(r[0][ok], other[0][ok], rtol=0, atol=1e-9)
            if "until" in kw:
                # (an instance whose error sits within rounding of a threshold may break one iteration apart)
                assert (r[3] != other[3]).sum() <= 1 and (r[4] != other[4]).sum() <= 1


@pytest.mark.parametrize("name,scene", [("leap_c", "leap_hand__scene_right"), ("kinova_c", "stanford_tidybot__scene_mobile_kinova")])
def test_real_mink_fixtures_of_the_sixteen_register_build(nat, name, scene):
    """The REAL mink (tests/golden/make_golden_small.py) on the LEAP hand — four fingertip tasks on a tree of four fingers —
    and on the mobile Kinova with the tasks of examples/mobile_kinova.py (DampingTask holding the base, a posture cost on
    one dof): the public API on its default dispatch (`ik_quad_kernel_16`) against the recorded v; every eighth instance is
    the small-angle stream (tolerance 1e-5)."""
    import mink_amd as mink
    d = np.load(os.path.join(oc.GOLDEN, f"ik_{name}.npz"))
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", scene + ".json"))
    B = len(d["q"])
    cfg = mink.Configuration(m, d["q"])
    if name == "leap_c":
        tasks = [mink.FrameTask(s, "site", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0) for s in ("tip_1", "tip_2", "tip_3", "th_tip")]
        for k, t in enumerate(tasks):
            t.set_target(mink.SE3(d["frame_targets"][:, k]))
        post = mink.PostureTask(m, cost=1e-2); post.set_target(d["posture_target"])
        tasks.append(post)
        vel = {n: np.pi for n in m.jnt_names}
    else:
        ee = mink.FrameTask("pinch_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
        ee.set_target(mink.SE3(d["frame_targets"][:, 0]))
        pc = np.zeros(m.nv); pc[2] = 1e-3
        post = mink.PostureTask(m, cost=pc); post.set_target(d["posture_targets"][0, 0])
        dc = np.zeros(m.nv); dc[:2] = 100.0; dc[2] = 1e-3
        tasks = [ee, post, mink.DampingTask(m, dc)]
        vel = {m.jnt_names[j]: (0.5 if m.jnt_type[j] == 2 else np.pi) for j in range(m.njnt)}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)]
    v = mink.solve_ik(cfg, tasks, float(d["dt"]), "quadprog", float(d["damping"]), limits=lims)
    prob = list(cfg._problems.values())[-1]
    assert prob.last_kernel() == QUAD + "_16", prob.last_kernel()
    main = np.ones(B, bool); main[7::8] = False
    err = _rel(v, d["v"])
    print("%s: row kernel (16 registers) vs real mink: max rel err main %.1e small-angle %.1e; dofs on a velocity bound %d" % (
        name, err[main].max(), err[~main].max(), int((np.abs(np.abs(v) - np.array([vel[n] for n in m.jnt_names])) < 1e-9).sum())))
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5
    # H and c of the same problems (wavefront kernel's taps) against the reference's build_ik
    pr = mink.build_ik(cfg, tasks, float(d["dt"]), float(d["damping"]), lims)
    np.testing.assert_allclose(pr.P[main], d["H"][main], rtol=0, atol=1e-10 * np.abs(d["H"]).max())
    np.testing.assert_allclose(pr.q[main], d["c"][main], rtol=0, atol=1e-10 * max(1.0, np.abs(d["c"]).max()))
    np.testing.assert_allclose(pr.P[~main], d["H"][~main], rtol=0, atol=1e-6 * np.abs(d["H"]).max())      # (small-angle stream)


@pytest.mark.parametrize("name,scene", [("h1_c", "unitree_h1__scene"), ("go1_c", "unitree_go1__scene"), ("h1_full", "unitree_h1__scene")])
def test_real_mink_fixtures_of_the_two_row_build(nat, name, scene):
    """The REAL mink (tests/golden/make_golden_mid.py) on the Unitree H1 — the tasks of examples/humanoid_h1.py without the CoM
    task: a body-frame pelvis task, feet, wrists, posture — and on the Go1 with the tasks of examples/quadruped_go1.py: the public
    API on its default dispatch, which for these floating-base robots is the row kernel on two DPP rows per problem
    (`ik_quad_kernel_32`), against the recorded v; every eighth instance is the small-angle stream (1e-5).  The
    same call pinned to the wavefront kernel gives the same answer."""
    import mink_amd as mink
    d = np.load(os.path.join(oc.GOLDEN, f"ik_{name}.npz"))
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", scene + ".json"))
    B = len(d["q"])
    cfg = mink.Configuration(m, d["q"])
    if name in ("h1_c", "h1_full"):
        tasks = [mink.FrameTask("pelvis", "body", position_cost=0.0, orientation_cost=10.0)]
        tasks += [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0) for s in ("right_foot", "left_foot")]
        tasks += [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=0.0, lm_damping=1.0) for s in ("right_wrist", "left_wrist")]
        post = mink.PostureTask(m, cost=1.0)
        lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {m.jnt_names[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] == 3})]
    else:
        tasks = [mink.FrameTask("trunk", "body", position_cost=1.0, orientation_cost=1.0)]
        tasks += [mink.FrameTask(s, "site", position_cost=1.0, orientation_cost=0.0) for s in ("FL", "FR", "RR", "RL")]
        post = mink.PostureTask(m, cost=1e-5)
        lims = [mink.ConfigurationLimit(m)]
    for k, t in enumerate(tasks):
        t.set_target(mink.SE3(d["frame_targets"][:, k]))
    post.set_target(d["posture_target"])
    tasks.append(post)
    if name == "h1_full":                              # examples/humanoid_h1.py as written: + ComTask(200), per-instance CoM target
        com = mink.ComTask(cost=200.0)
        com.set_target(d["com_targets"][:, 0])
        tasks.append(com)
    v = mink.solve_ik(cfg, tasks, float(d["dt"]), "quadprog", float(d["damping"]), limits=lims)
    prob = list(cfg._problems.values())[-1]
    assert prob.last_kernel() == QUAD + "_32", prob.last_kernel()
    main = np.ones(B, bool); main[7::8] = False
    err = _rel(v, d["v"])
    print("%s: row kernel (two rows per problem) vs real mink: max rel err main %.1e small-angle %.1e" % (name, err[main].max(), err[~main].max()))
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5
    pr = mink.build_ik(cfg, tasks, float(d["dt"]), float(d["damping"]), lims)        # (H, c: the wavefront kernel's taps)
    np.testing.assert_allclose(pr.P[main], d["H"][main], rtol=0, atol=1e-10 * np.abs(d["H"]).max())
    np.testing.assert_allclose(pr.q[main], d["c"][main], rtol=0, atol=1e-10 * max(1.0, np.abs(d["c"]).max()))


@pytest.mark.parametrize("seed", range(int(os.environ.get("MKH_FUZZ_SEEDS", "10"))))
def test_random_trees_on_the_two_row_build(nat, seed):
    """Random trees with 17 … 32 dofs or a FLOATING root (hinge / slide joints elsewhere; branching, fixed bodies, several joints on
    one body, frames on sites and bodies, zero-cost rows, per-dof posture costs, random gains / LM damping / limits) through the
    public API: default dispatch = the row kernel on two DPP rows per problem; the same compiled problem on the wavefront kernel
    and on the C oracle."""
    import mink_amd as mink
    from random_models import random_mjcf, rand_q
    rng = np.random.default_rng(9100 + seed)
    free = seed % 2 == 0
    for _ in range(200):
        nbody = int(rng.integers(4, 20))
        xml, sites = random_mjcf(rng, nbody, free_root=free, no_ball=True)
        m = mink.loads_mjcf(xml)
        if (free and 7 <= m.nv <= 32) or (not free and 17 <= m.nv <= 32):
            break
    else:
        pytest.skip("no draw in range")
    B = 53
    q = np.stack([rand_q(m, rng) for _ in range(B)])
    cfg = mink.Configuration(m, q)
    tgt_cfg = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.2, size=(B, m.nv)), 1.0))
    frames = [(s, "site") for s in sites] + [(f"b{i}", "body") for i in range(nbody)]
    picks = [frames[i] for i in rng.choice(len(frames), size=min(int(rng.integers(1, 7)), len(frames)), replace=False)]
    tasks, specs, n_rel = [], [], 0
    for name, typ in picks:
        pc = rng.uniform(0.5, 20.0) * (rng.uniform(size=3) < 0.8)
        oc_ = rng.uniform(0.1, 5.0) * (rng.uniform() < 0.6)
        if not pc.any() and oc_ == 0.0:
            pc = np.ones(3)
        gain, lm = float(rng.uniform(0.3, 1.0)), float(rng.uniform(0.0, 1.0))
        if seed % 4 in (1, 2) and len(tasks) >= 1 and rng.uniform() < 0.7:
            # RelativeFrameTask: this frame measured in the frame of an earlier pick (chains that share dofs, root-only dofs)
            rname, rtyp = picks[int(rng.integers(0, len(tasks)))]
            if (rname, rtyp) == (name, typ):
                rname, rtyp = "b0", "body"
            ft = mink.RelativeFrameTask(name, typ, rname, rtyp, position_cost=pc, orientation_cost=oc_, gain=gain, lm_damping=lm)
            ft.set_target(tgt_cfg.get_transform(name, typ, rname, rtyp))
            specs.append(("rel", m.name2id(typ, name), typ, m.name2id(rtyp, rname), rtyp, ft.cost.copy(), gain, lm))
            n_rel += 1
        else:
            ft = mink.FrameTask(name, typ, position_cost=pc, orientation_cost=oc_, gain=gain, lm_damping=lm)
            ft.set_target(tgt_cfg.get_transform_frame_to_world(name, typ))
            specs.append((m.name2id(typ, name), typ, ft.cost.copy(), gain, lm))
        tasks.append(ft)
    post = mink.PostureTask(m, cost=rng.uniform(0.0, 1.0, size=m.nv) * (rng.uniform(size=m.nv) < 0.8), gain=float(rng.uniform(0.2, 1.0)),
                            lm_damping=float(rng.uniform(0.0, 0.5)))
    post.set_target(rand_q(m, rng))
    vel = {m.jnt_names[j]: float(rng.uniform(0.2, 3.0)) for j in range(m.njnt) if m.jnt_type[j] in (2, 3) and rng.uniform() < 0.7}
    lims = [mink.ConfigurationLimit(m, gain=float(rng.uniform(0.5, 1.0)))] + ([mink.VelocityLimit(m, vel)] if vel else [])
    dt, damping = float(rng.choice([2e-3, 1e-2, 5e-2])), float(rng.choice([1e-6, 1e-3, 1e-1]))
    com, com_t = None, None
    if seed % 3 == 0:                                   # ... and a ComTask with per-instance targets (zero-cost rows allowed)
        cost = rng.uniform(1.0, 50.0, size=3) * (rng.uniform(size=3) < 0.8)
        com = mink.ComTask(cost=cost if cost.any() else np.ones(3), gain=float(rng.uniform(0.3, 1.0)), lm_damping=float(rng.uniform(0.0, 0.5)))
        com_t = np.asarray(tgt_cfg.subtree_com())
        com.set_target(com_t)
    v = mink.solve_ik(cfg, tasks + [post] + ([com] if com is not None else []), dt, "mi355x", damping, limits=lims)
    prob = list(cfg._problems.values())[-1]
    if prob.last_kernel().startswith("ik_solve_kernel"):
        pytest.skip("more than 32 links on the frames' chains: wavefront kernel")
    assert prob.last_kernel() == QUAD + "_32", prob.last_kernel()
    ftg = np.stack([(ft.transform_target_to_root if isinstance(ft, mink.RelativeFrameTask) else ft.transform_target_to_world).wxyz_xyz
                    for ft in tasks], axis=1)
    ptq = post.target_q[None, :]
    ctg = None if com is None else com_t.reshape(B, 1, 3)
    vw, stw = prob.solve(q, ftg, ptq, ctg, dt, damping, wave_kernel=True)
    assert prob.last_kernel().startswith("ik_solve_kernel") and (stw & ~1 == 0).all()
    ls = [oik.ConfigurationLimitSpec(lims[0].gain)] + ([oik.VelocityLimitSpec(lims[1].indices, lims[1].limit)] if vel else [])

    def oracle_tasks(i):
        ts = []
        for sp, k in zip(specs, range(len(specs))):
            if sp[0] == "rel":
                ts.append(oik.RelativeFrameTaskSpec(sp[1], sp[2], sp[3], sp[4], sp[5], ftg[i, k], sp[6], sp[7]))
            else:
                ts.append(oik.FrameTaskSpec(sp[0], sp[1], sp[2], ftg[i, k], sp[3], sp[4]))
        ts.append(oik.PostureTaskSpec(post.cost, post.target_q, post.gain, post.lm_damping))
        if com is not None:
            ts.append(oik.ComTaskSpec(np.asarray(com.cost, dtype=np.float64), com_t[i], com.gain, com.lm_damping))
        return ts

    if n_rel:                                           # (the C restatement has no RelativeFrameTask: numpy oracle on a sample)
        for i in (0, B // 2, B - 1):
            v_ref = oik.solve_ik(m, q[i], oracle_tasks(i), dt, damping, ls)
            assert np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()) < 1e-8, (i, n_rel)
        v_c = vw
    else:
        v_c, st_c = cport.CProblem(m, oracle_tasks(0), ls).solve_batch(q, ftg, ptq, dt, damping, com_target=ctg)
        assert (st_c == 0).all()
    print("seed %d: nv %d (%s root), %d bodies, %d frame tasks%s: two-row vs wavefront %.1e, vs C oracle %.1e" % (
        seed, m.nv, "free" if free else "fixed", nbody, len(tasks), (" + ComTask" if com is not None else "") + (" (%d relative)" % n_rel if n_rel else ""), _rel(v, vw).max(), _rel(v, v_c).max()))
    assert _rel(v, vw).max() < 1e-8 and _rel(v, v_c).max() < 1e-7


@pytest.mark.parametrize("B", [1, 3, 129])
def test_two_row_build_on_ragged_batches(nat, B):
    """Two problems per wavefront: an odd batch leaves the upper half of the last wavefront idle (it redoes the last problem and
    stores nothing); one instance is half a wavefront.  Go1 (floating base) against the wavefront kernel and the C oracle."""
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", "unitree_go1__scene.json"))
    nm = nat.NativeModel(m)
    fts = [{"frame_type": "site", "frame_id": m.name2id("site", s), "cost": [1.0, 1.0, 1.0, 0.0, 0.0, 0.0], "gain": 1.0, "lm_damping": 0.5}
           for s in ("FL", "FR", "RL", "RR")]
    fts.append({"frame_type": "body", "frame_id": m.name2id("body", "trunk"), "cost": [1.0] * 6, "gain": 1.0, "lm_damping": 0.0})
    hs = [j for j in range(m.njnt) if m.jnt_type[j] == 3]
    vidx = [int(m.jnt_dofadr[j]) for j in hs]
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-3}], configuration_limits=[nc._cfg_limit(m)],
                             velocity_limits=[{"indices": vidx, "limit": np.full(len(vidx), 2.0)}], max_batch=B)
    home = m.key_qpos[m.name2id("key", "home")]
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(B), B, base_q=home)
    v, st = prob.solve(q, tg, home[None, :], None, 1e-2, 1e-4)
    assert prob.last_kernel() == QUAD + "_32" and (st & ~1 == 0).all()
    vw, stw = prob.solve(q, tg, home[None, :], None, 1e-2, 1e-4, wave_kernel=True)
    assert prob.last_kernel().startswith("ik_solve_kernel") and (stw == st).all()
    assert _rel(v, vw).max() < 1e-8
    c6 = lambda f: np.array(f["cost"], dtype=np.float64)
    tasks = [oik.FrameTaskSpec(f["frame_id"], f["frame_type"], c6(f), tg[0, k], 1.0, f["lm_damping"]) for k, f in enumerate(fts)]
    tasks.append(oik.PostureTaskSpec(np.full(m.nv, 1e-3), home))
    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.array(vidx), np.full(len(vidx), 2.0))]
    v_c, st_c = cport.CProblem(m, tasks, limits).solve_batch(q, tg, home[None, :], 1e-2, 1e-4)
    assert (st_c == 0).all() and _rel(v, v_c).max() < 1e-8


def test_real_mink_fixture_of_the_arm_with_a_hand(nat):
    """The task set of examples/arm_hand_iiwa_allegro.py:62-94 — a FrameTask on the arm's attachment site, a posture task, one
    RelativeFrameTask per fingertip measured in the PALM — recorded with the real mink on a 7 + 16-dof arm + hand
    (tests/golden/make_golden_mid.py): the public API on its default dispatch, which since round 4 is the two-row build of the row
    kernel for such a robot (a dof on the chains of both frames drops out of a relative task's Jacobian, one on the root's chain
    only enters with the opposite sign), against the recorded v; the wavefront kernel gives the same."""
    import mink_amd as mink
    d = np.load(os.path.join(oc.GOLDEN, "ik_arm_hand.npz"))
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "arm_hand.json"))
    B = len(d["q"])
    cfg = mink.Configuration(m, d["q"])
    ee = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    ee.set_target(mink.SE3(d["frame_targets"][:, 0]))
    post = mink.PostureTask(m, cost=5e-2); post.set_target(d["posture_target"])
    fingers = []
    for k, t in enumerate(("ff_tip", "mf_tip", "rf_tip", "th_tip")):
        f = mink.RelativeFrameTask(t, "site", "palm", "body", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0)
        f.set_target(mink.SE3(d["frame_targets"][:, 1 + k]))
        fingers.append(f)
    tasks, lims = [ee, post] + fingers, [mink.ConfigurationLimit(m)]
    v = mink.solve_ik(cfg, tasks, float(d["dt"]), "quadprog", float(d["damping"]), limits=lims)
    prob = list(cfg._problems.values())[-1]
    assert prob.last_kernel() == QUAD + "_32", prob.last_kernel()
    main = np.ones(B, bool); main[7::8] = False
    err = _rel(v, d["v"])
    print("arm + hand (4 RelativeFrameTasks): two-row build vs real mink: max rel err main %.1e small-angle %.1e" % (err[main].max(), err[~main].max()))
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5
    pr = mink.build_ik(cfg, tasks, float(d["dt"]), float(d["damping"]), lims)        # (H, c: the wavefront kernel's taps)
    np.testing.assert_allclose(pr.P[main], d["H"][main], rtol=0, atol=1e-10 * np.abs(d["H"]).max())


@pytest.mark.parametrize("scene", ["universal_robots_ur5e__scene", "leap_hand__scene_right", "unitree_h1__scene"])
def test_warm_start_across_calls(nat, scene):
    """MKH_FLAG_WARM_START on the row kernel: a closed loop of single solves on the same batch — every step's v equals the
    cold solve's (the optimum is unique), whatever partition the previous call left in the handle; a permuted batch (a wrong
    prediction for every instance) and a changed batch size (state reset) still give the right answers."""
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", scene + ".json"))
    nm = nat.NativeModel(m)
    B = 1500
    sites = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0]
    tips = sites[-4:] if m.nv > 8 else sites[-1:]
    fts = [{"frame_type": "site", "frame_id": i, "cost": [1.0, 1.0, 1.0, 0.2, 0.2, 0.2], "gain": 1.0, "lm_damping": 1.0} for i in tips]
    vidx = [int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] in (2, 3)]
    kw = dict(frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
              velocity_limits=[{"indices": vidx, "limit": np.full(len(vidx), 1.0)}], max_batch=B)
    prob, cold = nat.NativeProblem(nm, **kw), nat.NativeProblem(nm, **kw)
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(8), B, base_q=m.qpos0)
    pt = m.qpos0[None, :]
    dt, damping = 2e-2, 1e-3
    qw, worst = q.copy(), 0.0
    for step in range(10):
        vw, stw = prob.solve(qw, tg, pt, None, dt, damping, warm_start=True)
        assert prob.last_kernel().startswith(QUAD)
        vc, stc = cold.solve(qw, tg, pt, None, dt, damping)
        assert (stw == stc).all() and (stw & ~1 == 0).all()
        worst = max(worst, _rel(vw, vc).max())
        qw = nm.integrate(qw, vw, dt)
    print("%s: closed loop of 10 warm-started solves vs cold solves: max rel |dv| = %.2e" % (scene, worst))
    assert worst < 1e-9
    perm = np.random.default_rng(0).permutation(B)
    vp, _ = prob.solve(qw[perm], tg[perm], pt, None, dt, damping, warm_start=True)
    vc, _ = cold.solve(qw[perm], tg[perm], pt, None, dt, damping)
    assert _rel(vp, vc).max() < 1e-9
    vh, _ = prob.solve(qw[:100], tg[:100], pt, None, dt, damping, warm_start=True)
    assert _rel(vh, cold.solve(qw[:100], tg[:100], pt, None, dt, damping)[0]).max() < 1e-9
