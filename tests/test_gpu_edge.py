"""Edge cases of the batched path (cf. the reference's tests: empty task lists, default damping 1e-12,
limits=[] vs None, ragged batches) — compared with the oracle where the reference defines a result."""

import numpy as np
import pytest

import mink_amd as mink
import oracle_configs as oc
from mink_amd import _native as nat
from mink_amd import workloads
from oracle import ik as oik

pytestmark = pytest.mark.gpu


def _oracle_v(model_name, q, tasks, limits, dt, damping):
    return oik.solve_ik(oc.model(model_name), q, tasks, dt, damping, limits)


@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 2048, 2049, 5000, 16385])
def test_ragged_batch_sizes(B):
    """grid-stride loop tails: every batch size gives the same per-row answers as B = 1."""
    m = workloads.load_robot("ur5e")
    nm = nat.NativeModel(m)
    home = m.key_qpos[0]
    prob = nat.NativeProblem(nm, frame_tasks=[{"frame_type": "site", "frame_id": 0, "cost": [1.0] * 6, "lm_damping": 1.0}],
                             posture_tasks=[{"cost": 1e-2}], max_batch=B)
    rng = np.random.default_rng(B)
    q, tg = workloads.make_batch(m, nm, prob, rng, B, base_q=home)
    # (the wavefront kernel: the default dispatch of a plain solve for this arm is the row-per-problem kernel, whose own
    # equivariance test is tests/test_gpu_quad_kernel.py)
    v, st = prob.solve(q, tg, home[None, :], None, 2e-3, 1e-3, wave_kernel=True)
    assert (st == 0).all()
    vl, stl = prob.solve(q, tg, home[None, :], None, 2e-3, 1e-3)
    assert prob.last_kernel() == "ik_quad_kernel" and (stl == 0).all()
    np.testing.assert_allclose(vl, v, rtol=0, atol=1e-9 * max(1.0, np.abs(v).max()))
    one = nat.NativeProblem(nm, frame_tasks=[{"frame_type": "site", "frame_id": 0, "cost": [1.0] * 6, "lm_damping": 1.0}],
                            posture_tasks=[{"cost": 1e-2}], max_batch=1)
    for i in {0, B // 2, B - 1}:
        v1, _ = one.solve(q[i:i + 1], tg[i:i + 1], home[None, :], None, 2e-3, 1e-3, wave_kernel=True)
        np.testing.assert_array_equal(v1[0], v[i])
        v1, _ = one.solve(q[i:i + 1], tg[i:i + 1], home[None, :], None, 2e-3, 1e-3)
        np.testing.assert_array_equal(v1[0], vl[i])
    if B > 2:
        # every row: the same batch cut at an odd place (different grid / XCD slices) gives bitwise the same rows
        h = B // 3 + 1
        va, _ = prob.solve(q[:h], tg[:h], home[None, :], None, 2e-3, 1e-3, wave_kernel=True)
        vb, _ = prob.solve(q[h:], tg[h:], home[None, :], None, 2e-3, 1e-3, wave_kernel=True)
        np.testing.assert_array_equal(np.concatenate([va, vb]), v)


def test_default_damping_single_frame_task_vs_oracle():
    """damping=1e-12 and one 6-row task on a 6-dof arm: H = JᵀJ + 1e-12·I is invertible but ill-scaled."""
    m = mink.load_robot("ur5e")
    om = oc.model("ur5e")
    rng = np.random.default_rng(3)
    q = workloads.sample_q(m, rng, 32, base_q=m.key_qpos[0])
    cfg = mink.Configuration(m, q)
    task = mink.FrameTask("attachment_site", "site", 1.0, 1.0)
    tgt = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.05, size=(32, 6)), 1.0))
    task.set_target(tgt.get_transform_frame_to_world("attachment_site", "site"))
    v = mink.solve_ik(cfg, [task], 1e-2, "mi355x", limits=[])            # default damping 1e-12
    sid = om.name2id("site", "attachment_site")
    for i in range(32):
        spec = oik.FrameTaskSpec(sid, "site", np.ones(6), task.transform_target_to_world.wxyz_xyz[i])
        ref = oik.solve_ik(om, q[i], [spec], 1e-2, 1e-12, [])
        # cond(H) can reach ~1e8 near wrist alignment: compare in the task space the QP actually pins
        assert np.abs(v[i] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_only_posture_and_only_limits():
    m = mink.load_robot("g1")
    cfg = mink.Configuration(m)
    cfg.update_from_keyframe("stand")
    post = mink.PostureTask(m, cost=1.0)
    tgt = cfg.q
    tgt[7:] += 0.01
    post.set_target(tgt)
    v = mink.solve_ik(cfg, [post], 1e-2, "mi355x", 1e-3, limits=[])
    # H = (1 + 1e-3) I on hinge dofs, c = −e ⇒ Δq = e / 1.001
    np.testing.assert_allclose(v[6:] * 1e-2, 0.01 / 1.001, rtol=1e-13)
    np.testing.assert_allclose(v[:6], 0.0, atol=0)
    v = mink.solve_ik(cfg, [], 1e-2, "mi355x", 1e-3)                     # only the default ConfigurationLimit
    np.testing.assert_allclose(v, 0.0, atol=0)


def test_status_flags_and_nan_input():
    m = workloads.load_robot("ur5e")
    nm = nat.NativeModel(m)
    home = m.key_qpos[0]
    cl = mink.ConfigurationLimit(m)._native_desc()[1]
    prob = nat.NativeProblem(nm, posture_tasks=[{"cost": 1.0}], configuration_limits=[cl],
                             velocity_limits=[{"indices": [0], "limit": [1e-3]}], max_batch=4)
    q = np.tile(home, (4, 1))
    q[1, 0] = 7.0          # outside the range, and it cannot come back within the velocity limit ⇒ infeasible
    q[2, 1] = 6.5          # outside the range, recoverable ⇒ only the OUTSIDE_LIMITS bit
    v, st = prob.solve(q, None, home[None, :], None, 1e-3, 1e-6)
    assert st[0] == 0 and st[3] == 0
    assert st[1] & nat.ST_INFEASIBLE and st[1] & nat.ST_OUTSIDE_LIMITS and np.isnan(v[1]).all()
    assert st[2] == nat.ST_OUTSIDE_LIMITS and np.isfinite(v[2]).all()
    # zero damping and no task ⇒ H = 0: not positive definite
    p2 = nat.NativeProblem(nm, max_batch=1)
    v, st = p2.solve(home[None, :], None, None, None, 1e-3, 0.0)
    assert st[0] & nat.ST_NOT_PD and np.isnan(v).all()


def test_argument_validation():
    m = workloads.load_robot("ur5e")
    nm = nat.NativeModel(m)
    prob = nat.NativeProblem(nm, frame_tasks=[{"frame_type": "site", "frame_id": 0, "cost": [1.0] * 6}], max_batch=4)
    with pytest.raises(ValueError):
        prob.solve(np.zeros((2, 5)), np.zeros((2, 1, 7)), None, None, 1e-2, 1e-3)
    with pytest.raises(ValueError):
        prob.solve(np.zeros((2, 6)), np.zeros((2, 2, 7)), None, None, 1e-2, 1e-3)
    with pytest.raises(nat.MinkHipError, match="max_batch"):
        prob.solve(np.zeros((8, 6)), np.zeros((8, 1, 7)), None, None, 1e-2, 1e-3)
    # ... on every path: device pointers too (a warm start used to write past the handle's max_batch × nv active sets),
    # checked by the library itself (ctypes call below) and by the binding
    import ctypes as C
    import torch
    qd = torch.zeros((8, 6), dtype=torch.float64, device="cuda"); td = torch.zeros((8, 1, 7), dtype=torch.float64, device="cuda")
    td[:, :, 0] = 1.0
    vd = torch.zeros((8, 6), dtype=torch.float64, device="cuda"); sd = torch.zeros((8,), dtype=torch.int32, device="cuda")
    for flags in (nat.FLAG_DEVICE_PTRS, nat.FLAG_DEVICE_PTRS | nat.FLAG_WARM_START):
        rc = nat.lib().mkh_solve(prob.handle, 8, C.c_void_p(qd.data_ptr()), C.c_void_p(td.data_ptr()), None, None,
                                 C.c_double(1e-2), C.c_double(1e-3), C.c_void_p(vd.data_ptr()), C.c_void_p(sd.data_ptr()), flags, None)
        assert rc == -1 and b"max_batch" in nat.lib().mkh_last_error(), rc       # MKH_E_INVALID
    torch.cuda.synchronize()
    assert float(vd.abs().max()) == 0.0                     # nothing was launched
    with pytest.raises(nat.MinkHipError, match="max_batch"):
        prob.solve(qd, td, None, None, 1e-2, 1e-3, warm_start=True)
    with pytest.raises(nat.MinkHipError, match="dt must be"):
        prob.solve(np.zeros((2, 6)), np.zeros((2, 1, 7)), None, None, 0.0, 1e-3)
    with pytest.raises(nat.MinkHipError, match="site id"):
        nat.NativeProblem(nm, frame_tasks=[{"frame_type": "site", "frame_id": 9, "cost": [1.0] * 6}])
    with pytest.raises(nat.MinkHipError, match="no hull in the model"):     # (a visual mesh geom of the packaged UR5e: no hull was kept for it)
        # a mesh geom has no distance routine (every pair of primitives has one: analytic or the general convex routine)
        nat.NativeProblem(nm, collision_limits=[{"geom_id_pairs": [[1, m.name2id("geom", "wall")]],
                                                 "gain": 0.85, "minimum_distance_from_collisions": 0.005,
                                                 "collision_detection_distance": 0.01, "bound_relaxation": 0.0}])


def test_low_rank_start_conditioning_gate():
    """The low-rank QP start is only used when damping + posture diagonal is not tiny against the task costs
    (DESIGN.md §4.2); below the gate the same handle launches the direct start.  Both agree with the C oracle
    to what the conditioning allows."""
    import native_configs as nc
    import oracle_configs as occ
    from oracle import cport
    m = workloads.load_robot("g1")
    nm = nat.NativeModel(m)
    B = 256
    prob, dt, _ = nc.build("g1_c3", nm, B)
    stand = m.key_qpos[0]
    q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(9), B, base_q=stand)
    # (the floating base carries no posture term: its diagonal is the damping alone; gate at 1e-7·200² = 4e-3)
    for damping, expect_low_rank, tol in ((1e-1, True, 1e-8), (1e-2, True, 1e-8), (1e-3, False, 1e-8)):
        v, st = prob.solve(q, tg, stand[None, :], None, dt, damping)
        assert ("_r44" in prob.last_kernel()) == expect_low_rank, (damping, prob.last_kernel())
        mm, tasks, limits, dt_o, _ = occ.g1_c3(tg[0], stand)
        v_ref, _ = cport.CProblem(mm, tasks, limits).solve_batch(q, tg, stand[None, :], dt, damping)
        err = np.abs(v - v_ref).max(axis=1) / np.maximum(1.0, np.abs(v_ref).max(axis=1))
        assert (st == 0).all() and err.max() < tol, (damping, err.max())
    # posture cost 1e-3 and damping 1e-9: Dg ≈ 1e-6 < 1e-7·200² ⇒ direct start
    weak = nat.NativeProblem(nm, frame_tasks=[{"frame_type": "site", "frame_id": m.name2id("site", "left_palm"),
                                               "cost": [200.0] * 3 + [0.0] * 3, "lm_damping": 0.0}],
                             posture_tasks=[{"cost": 1e-3}], max_batch=B)
    v, st = weak.solve(q, tg[:, 2:3], stand[None, :], None, dt, 1e-9)
    assert "_r" not in weak.last_kernel(), weak.last_kernel()
    assert (st == 0).all() and np.isfinite(v).all()
    v2, st2 = weak.solve(q, tg[:, 2:3], stand[None, :], None, dt, 1e-1)       # same handle, larger damping
    assert "_r44" in weak.last_kernel(), weak.last_kernel()
    assert (st2 == 0).all()


def test_ticket_counter_survives_many_launches_and_odd_batches():
    """Problem distribution: 7/8 of every wave's share is static, the tail is drawn from a ticket counter that the
    launch's last draw zeroes for the next one.  Batches of every shape, launched back to back on one handle,
    must return what a fresh handle returns — bitwise."""
    from mink_amd import _native as nat
    from mink_amd import workloads
    import native_configs as nc
    model = workloads.load_robot("g1")
    nm = nat.NativeModel(model)
    B = 40000
    prob, dt, damping = nc.build("g1_c3", nm, B)
    rng = np.random.default_rng(7)
    stand = model.key_qpos[0]
    q, tg = workloads.make_batch(model, nm, prob, rng, B, base_q=stand)
    ref, st_ref = prob.solve(q, tg, stand[None, :], None, dt, damping)
    assert (st_ref == 0).all()
    for n in (B, 1, 2047, 2048, 2049, 8191, 8192, 16385, 39999, B, 3, B):
        v, st = prob.solve(q[:n], tg[:n], stand[None, :], None, dt, damping)
        np.testing.assert_array_equal(v, ref[:n])
        np.testing.assert_array_equal(st, st_ref[:n])
    fresh, _ = nc.build("g1_c3", nm, B)[0].solve(q, tg, stand[None, :], None, dt, damping)
    np.testing.assert_array_equal(fresh, ref)


def _chain_xml(n_links, seed=0):
    rng = np.random.default_rng(seed)
    xml = ['<mujoco><compiler angle="radian"/><worldbody>']
    for i in range(n_links):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        xml.append(f'<body name="b{i}" pos="{0.03 + 0.02 * rng.uniform():.4f} {0.01 * rng.normal():.4f} {0.01 * rng.normal():.4f}">'
                   f'<joint name="j{i}" type="hinge" axis="{ax[0]:.5f} {ax[1]:.5f} {ax[2]:.5f}" range="-1.5 1.5"/>'
                   f'<geom type="sphere" size="0.01" mass="0.1"/>')
    xml.append('<site name="tip" pos="0.02 0 0"/>')
    xml.append("</body>" * n_links)
    xml.append("</worldbody></mujoco>")
    return "".join(xml)


def test_maximum_sizes_of_one_wavefront():
    """The one-wavefront limits (SURVEY §8a sizes): 64 bodies / 64 dofs.  A 63-link serial chain (64 bodies with the world,
    nv = 63, tree depth 63 ⇒ 6 pointer-jumping rounds, 64-row tableau) solves to oracle accuracy on the wavefront kernel; one
    link more — round 3: refused — runs on the workgroup-per-problem kernel (round 4, tests/test_gpu_wide.py has the 100-dof
    chain): the same answer from either side of the limit."""
    import mink_amd as mink
    from oracle import cport
    from oracle import ik as oik
    m = mink.loads_mjcf(_chain_xml(63))
    assert m.nbody == 64 and m.nv == 63
    rng = np.random.default_rng(1)
    B = 64
    q = rng.uniform(-0.6, 0.6, size=(B, m.nq))
    cfg = mink.Configuration(m, q)
    tgt = mink.Configuration(m, q + rng.normal(scale=0.05, size=q.shape))
    ft = mink.FrameTask("tip", "site", position_cost=1.0, orientation_cost=0.5, lm_damping=1.0)
    ft.set_target(tgt.get_transform_frame_to_world("tip", "site"))
    post = mink.PostureTask(m, cost=1e-1)
    post.set_target(np.zeros(m.nq))
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {f"j{i}": 2.0 for i in range(63)})]
    dt, damping = 1e-2, 1e-3
    v = mink.solve_ik(cfg, [ft, post], dt, "mi355x", damping, limits=lims)
    ts = [oik.FrameTaskSpec(m.name2id("site", "tip"), "site", ft.cost, ft.transform_target_to_world.wxyz_xyz[0], 1.0, 1.0),
          oik.PostureTaskSpec(post.cost, post.target_q, 1.0)]
    ls = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(lims[1].indices, lims[1].limit)]
    v_c, st_c = cport.CProblem(m, ts, ls).solve_batch(q, ft.transform_target_to_world.wxyz_xyz[:, None, :],
                                                        post.target_q[None, :], dt, damping)
    assert (st_c == 0).all()
    err = np.abs(v - v_c).max(axis=1) / np.maximum(1.0, np.abs(v_c).max(axis=1))
    print("63-dof chain vs C oracle: max rel err %.2e" % err.max())
    assert err.max() < 1e-7
    assert list(cfg._problems.values())[-1].last_kernel().startswith("ik_solve_kernel_64")
    big = mink.loads_mjcf(_chain_xml(64))
    assert big.nbody == 65 and big.nv == 64
    qb = rng.uniform(-0.6, 0.6, size=(B, big.nq))
    cfgb = mink.Configuration(big, qb)
    ftb = mink.FrameTask("tip", "site", position_cost=1.0, orientation_cost=0.5, lm_damping=1.0)
    ftb.set_target(mink.Configuration(big, qb + rng.normal(scale=0.05, size=qb.shape)).get_transform_frame_to_world("tip", "site"))
    postb = mink.PostureTask(big, cost=1e-1)
    postb.set_target(np.zeros(big.nq))
    limsb = [mink.ConfigurationLimit(big), mink.VelocityLimit(big, {f"j{i}": 2.0 for i in range(64)})]
    vb = mink.solve_ik(cfgb, [ftb, postb], dt, "mi355x", damping, limits=limsb)
    assert list(cfgb._problems.values())[-1].last_kernel() == "ik_wide_kernel"
    tsb = [oik.FrameTaskSpec(big.name2id("site", "tip"), "site", ftb.cost, ftb.transform_target_to_world.wxyz_xyz[0], 1.0, 1.0),
           oik.PostureTaskSpec(postb.cost, postb.target_q, 1.0)]
    lsb = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(limsb[1].indices, limsb[1].limit)]
    vb_c, stb_c = cport.CProblem(big, tsb, lsb).solve_batch(qb, ftb.transform_target_to_world.wxyz_xyz[:, None, :],
                                                            postb.target_q[None, :], dt, damping)
    errb = np.abs(vb - vb_c).max(axis=1) / np.maximum(1.0, np.abs(vb_c).max(axis=1))
    print("64-dof chain (65 bodies) on the wide kernel vs C oracle: max rel err %.2e" % errb.max())
    assert (stb_c == 0).all() and errb.max() < 1e-8
