"""The reference's own behavioural tests, re-pointed at the device path through the mink-shaped
API (reference tests/test_solve_ik.py:33-148, test_frame_task.py:124-173, test_posture_task.py:87-108,
test_com_task.py:67-92, test_damping_task.py:21-26, test_jacobians.py:41-108,
test_configuration.py:36-118, test_configuration_limit.py:123-156), and checked against the
CPU oracle where the reference compares with MuJoCo."""

import numpy as np
import pytest

import mink_amd as mink
import oracle_configs as oc
from oracle import ik as oik
from oracle import lie as olie

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ur5e():
    return mink.load_robot("ur5e")


@pytest.fixture(scope="module")
def g1():
    return mink.load_robot("g1")


def _ur5e_limits(m):
    return [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: np.pi for n in m.jnt_names})]


# ------------------------------------------------------------------ solve_ik
def test_checks_and_ignores_configuration_limits(ur5e):
    cfg = mink.Configuration(ur5e)
    q = ur5e.key_qpos[0].copy()
    q[0] = 7.0                                  # range is ±6.28319
    cfg.update(q)
    with pytest.raises(mink.NotWithinConfigurationLimits):
        mink.solve_ik(cfg, [], limits=[mink.ConfigurationLimit(ur5e)], dt=1.0, safety_break=True, solver="quadprog")
    v = mink.solve_ik(cfg, [], limits=[mink.ConfigurationLimit(ur5e)], dt=1.0, solver="quadprog", safety_break=False)
    # the limit pushes the joint back inside: Δq ≤ gain·(q_max − q) < 0
    assert v[0] <= 0.95 * (6.28319 - 7.0) + 1e-12


def test_no_limits_and_default_limits(ur5e):
    cfg = mink.Configuration(ur5e)
    p = mink.build_ik(cfg, [], limits=[], dt=1.0)
    assert p.G is None and p.h is None
    p = mink.build_ik(cfg, [], dt=1.0)
    assert p.G.shape == (12, 6) and p.h.shape == (12,)
    np.testing.assert_allclose(p.P, np.eye(6) * 1e-12, atol=0)


def test_trivial_solution(ur5e):
    cfg = mink.Configuration(ur5e)
    v = mink.solve_ik(cfg, [], limits=[], dt=1e-3, solver="quadprog")
    np.testing.assert_allclose(v, np.zeros(6))


def test_single_task_fulfilled(ur5e):
    cfg = mink.Configuration(ur5e)
    task = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0)
    task.set_target(cfg.get_transform_frame_to_world("attachment_site", "site"))
    v = mink.solve_ik(cfg, [task], limits=_ur5e_limits(ur5e), dt=1e-3, solver="quadprog")
    np.testing.assert_allclose(v, np.zeros(6), atol=1e-10)


def test_single_task_convergence(ur5e):
    cfg = mink.Configuration(ur5e)
    cfg.update_from_keyframe("home")
    task = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0)
    T0 = cfg.get_transform_frame_to_world("attachment_site", "site")
    target = T0 @ mink.SE3.from_translation(np.array([0, 0, 0.1]))
    task.set_target(target)
    dt = 5e-3
    lims = _ur5e_limits(ur5e)
    velocity = mink.solve_ik(cfg, [task], limits=lims, dt=dt, solver="quadprog")
    assert not np.allclose(velocity, 0.0)
    assert abs(np.linalg.norm(task.compute_error(cfg)) - 0.1) < 1e-7
    last_error = 1e6
    for nb_steps in range(50):
        error = np.linalg.norm(task.compute_error(cfg))
        if error < 1e-6 and np.allclose(velocity, 0.0):
            break
        assert error < last_error
        last_error = error
        cfg.integrate_inplace(velocity, dt)
        velocity = mink.solve_ik(cfg, [task], limits=lims, dt=dt, solver="quadprog")
    assert np.allclose(velocity, 0.0)
    assert np.linalg.norm(task.compute_error(cfg)) < 1e-7
    np.testing.assert_allclose(cfg.get_transform_frame_to_world("attachment_site", "site").as_matrix(),
                               target.as_matrix(), atol=1e-7)
    assert nb_steps < 20


def test_infeasible_instance_raises(ur5e):
    cfg = mink.Configuration(ur5e)
    q = ur5e.key_qpos[0].copy()
    q[0] = 7.0
    cfg.update(q)
    lims = [mink.ConfigurationLimit(ur5e), mink.VelocityLimit(ur5e, {"shoulder_pan": 1e-3})]
    with pytest.raises(mink.SolverError, match="inconsistent"):
        mink.solve_ik(cfg, [], limits=lims, dt=1e-3)          # must retreat 0.68 rad but may move 1e-6


def test_batched_equals_single(g1):
    rng = np.random.default_rng(0)
    from mink_amd import workloads
    B = 16
    q = workloads.sample_q(g1, rng, B, base_q=g1.key_qpos[0])
    cfgB = mink.Configuration(g1, q)
    tasks = [mink.FrameTask(s, "site", 200.0, 10.0, lm_damping=1.0) for s in ("left_foot", "right_foot")]
    cfgT = mink.Configuration(g1, cfgB.integrate(rng.normal(scale=0.1, size=(B, g1.nv)), 1.0))
    for t in tasks:
        t.set_target(cfgT.get_transform_frame_to_world(t.frame_name, "site"))
    post = mink.PostureTask(g1, 1.0); post.set_target(g1.key_qpos[0])
    lims = [mink.ConfigurationLimit(g1)]
    vB = mink.solve_ik(cfgB, tasks + [post], 5e-3, "mi355x", 1e-1, limits=lims)
    assert vB.shape == (B, g1.nv)
    for i in (0, 5, 15):
        cfg1 = mink.Configuration(g1, q[i])
        t1 = [mink.FrameTask(t.frame_name, "site", 200.0, 10.0, lm_damping=1.0) for t in tasks]
        for a, b in zip(t1, tasks):
            a.set_target(mink.SE3(b.transform_target_to_world.wxyz_xyz[i]))
        v1 = mink.solve_ik(cfg1, t1 + [post], 5e-3, "mi355x", 1e-1, limits=lims)
        assert v1.shape == (g1.nv,)
        np.testing.assert_array_equal(v1, vB[i])          # same kernel, same arithmetic: bitwise


# -------------------------------------------------------------------- tasks
def test_frame_task_objective(g1):
    """unit cost ⇒ H = JᵀJ, c = eᵀJ; zero error at target (reference tests/test_frame_task.py:124-173)."""
    cfg = mink.Configuration(g1)
    cfg.update_from_keyframe("stand")
    task = mink.FrameTask("pelvis", "body", position_cost=1.0, orientation_cost=1.0)
    T = cfg.get_transform_frame_to_world("pelvis", "body")
    task.set_target(T)
    np.testing.assert_allclose(task.compute_error(cfg), np.zeros(6), atol=1e-14)
    task.set_target(T @ mink.SE3.from_translation(np.array([0.0, 0.01, 0.0])))
    J = task.compute_jacobian(cfg)
    e = task.compute_error(cfg)
    H, c = task.compute_qp_objective(cfg)
    np.testing.assert_allclose(H, J.T @ J, atol=1e-13)
    np.testing.assert_allclose(c, e.T @ J, atol=1e-15)
    with pytest.raises(mink.TargetNotSet):
        mink.FrameTask("pelvis", "body", 1.0, 1.0).compute_error(cfg)
    with pytest.raises(mink.InvalidFrame):
        bad = mink.FrameTask("nope", "body", 1.0, 1.0)
        bad.set_target(T)
        bad.compute_error(cfg)
    with pytest.raises(mink.UnsupportedFrame):
        cfg.get_transform_frame_to_world("pelvis", "joint")


def test_posture_com_damping_objectives(g1):
    cfg = mink.Configuration(g1)
    cfg.update_from_keyframe("stand")
    post = mink.PostureTask(g1, cost=1.0)
    post.set_target_from_configuration(cfg)
    np.testing.assert_allclose(post.compute_error(cfg), 0.0, atol=0)
    rng = np.random.default_rng(1)
    tgt = cfg.q
    tgt[7:] += rng.normal(scale=0.1, size=g1.nq - 7)
    post.set_target(tgt)
    J, e = post.compute_jacobian(cfg), post.compute_error(cfg)
    H, c = post.compute_qp_objective(cfg)
    np.testing.assert_allclose(H, J.T @ J, atol=1e-14)
    np.testing.assert_allclose(c, e.T @ J, atol=1e-14)
    assert np.all(e[:6] == 0) and np.all(J[:, :6] == 0)            # floating base untouched
    H0, c0 = mink.PostureTask(g1, cost=0.0).__class__(g1, cost=0.0), None
    dmp = mink.DampingTask(g1, cost=1.0)
    Hd, cd = dmp.compute_qp_objective(cfg)
    ref = np.eye(g1.nv); ref[:6, :6] = 0
    np.testing.assert_allclose(Hd, ref, atol=0)
    np.testing.assert_allclose(cd, 0.0, atol=0)
    com = mink.ComTask(cost=1.0)
    com.set_target_from_configuration(cfg)
    np.testing.assert_allclose(com.compute_error(cfg), 0.0, atol=1e-15)
    ocfg = oik.Configuration(oc.model("g1"), cfg.q)
    np.testing.assert_allclose(cfg.subtree_com(), ocfg.data.subtree_com[1], atol=1e-14)


def _fd_jacobian(cfg, task, h=1e-6):
    J = task.compute_jacobian(cfg)
    e0 = task.compute_error(cfg)
    nv = cfg.nv
    qs = np.stack([cfg.integrate(np.eye(nv)[i] * h, 1.0) for i in range(nv)])
    eh = task.compute_error(mink.Configuration(cfg.model, qs))     # one batched launch
    return J, ((eh - e0) / h).T


def test_task_jacobians_finite_difference(g1):
    """reference tests/test_jacobians.py:41-108 (G1, h=1e-6, ∞-norm 1e-5 / 1e-6)."""
    rng = np.random.default_rng(42)
    from mink_amd import workloads
    q = workloads.sample_q(g1, rng, 1, base_q=g1.key_qpos[0])[0]
    q[3:7] = olie.so3_exp(rng.normal(size=3))
    cfg = mink.Configuration(g1, q)
    for name, ftype in (("left_palm", "site"), ("torso_link", "body"), ("right_foot", "site")):
        task = mink.FrameTask(name, ftype, 1.0, 1.0)
        task.set_target(mink.SE3(olie.se3_exp(rng.normal(size=6))))
        J, Jfd = _fd_jacobian(cfg, task)
        assert np.abs(J - Jfd).max() < 1e-5
    post = mink.PostureTask(g1, 1.0)
    post.set_target(g1.key_qpos[0])
    J, Jfd = _fd_jacobian(cfg, post)
    assert np.abs(J - Jfd)[6:, 6:].max() < 1e-6
    com = mink.ComTask(1.0)
    com.set_target(np.zeros(3))
    J, Jfd = _fd_jacobian(cfg, com)
    assert np.abs(J - Jfd).max() < 1e-6


# ------------------------------------------------------------ configuration
def test_configuration_kinematics_vs_oracle(g1):
    rng = np.random.default_rng(3)
    from mink_amd import workloads
    B = 8
    q = workloads.sample_q(g1, rng, B, base_q=g1.key_qpos[0])
    cfg = mink.Configuration(g1, q)
    m = oc.model("g1")
    for name, ftype in (("head", "site"), ("left_knee_link", "body"), ("right_palm", "site")):
        T = cfg.get_transform_frame_to_world(name, ftype)
        Jb = cfg.get_frame_jacobian(name, ftype)
        fid = m.name2id(ftype, name)
        for i in range(B):
            o = oik.Configuration(m, q[i])
            Tr = o.get_transform_frame_to_world(fid, ftype)
            if np.dot(Tr[:4], T.wxyz_xyz[i, :4]) < 0:
                Tr[:4] = -Tr[:4]                                    # quaternion double cover
            np.testing.assert_allclose(T.wxyz_xyz[i], Tr, atol=1e-14)
            np.testing.assert_allclose(Jb[i], o.get_frame_jacobian(fid, ftype), atol=1e-13)
    v = rng.normal(size=(B, g1.nv))
    qn = cfg.integrate(v, 0.05)
    for i in range(B):
        np.testing.assert_allclose(qn[i], oik.Configuration(m, q[i]).integrate(v[i], 0.05), atol=1e-15)


def test_limit_inequalities_vs_oracle():
    m = mink.load_robot("shadow_left")
    om = oc.model("shadow_left")
    d = np.load(oc.GOLDEN + "/ik_shadow_c4.npz")
    q = d["q"][:6]
    cfg = mink.Configuration(m, q)
    f = list(oc.SHADOW_FINGERS)
    groups = [[f"{x}_1", f"{x}_2"] for x in f]
    col = mink.CollisionAvoidanceLimit(m, [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)],
                                       collision_detection_distance=0.03)
    cl = mink.ConfigurationLimit(m, gain=0.5, min_distance_from_limits=0.01)
    dt = 2e-3
    Gc, hc = col.compute_qp_inequalities(cfg, dt)
    Gl, hl = cl.compute_qp_inequalities(cfg, dt)
    assert Gc.shape == (6, 40, 24) and Gl.shape == (48, 24)
    assert (hc[np.isfinite(hc)] >= 0).all()                         # h ≥ bound_relaxation (reference :47-63)
    for i in range(6):
        o = oik.Configuration(om, q[i])
        G, h = oik.limit_inequalities(o, oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, collision_detection_distance=0.03), dt)
        fin = np.isfinite(h)
        assert (np.isfinite(hc[i]) == fin).all()
        np.testing.assert_allclose(hc[i][fin], h[fin], atol=1e-9)
        np.testing.assert_allclose(Gc[i], G, atol=1e-11)
        G, h = oik.limit_inequalities(o, oik.ConfigurationLimitSpec(0.5, 0.01), dt)
        np.testing.assert_allclose(hl[i], h, atol=1e-15)
        np.testing.assert_array_equal(Gl, G)


def test_relative_frame_task(g1):
    """reference tests/test_relative_frame_task.py:128-154 (root = world ⇒ −FrameTask) + oracle parity + FD."""
    rng = np.random.default_rng(9)
    from mink_amd import workloads
    q = workloads.sample_q(g1, rng, 1, base_q=g1.key_qpos[0])[0]
    cfg = mink.Configuration(g1, q)
    target = mink.SE3(olie.se3_exp(rng.normal(size=6) * 0.3))
    rel = mink.RelativeFrameTask("left_palm", "site", "world", "body", position_cost=1.0, orientation_cost=1.0)
    rel.set_target(target)
    absolute = mink.FrameTask("left_palm", "site", position_cost=1.0, orientation_cost=1.0)
    absolute.set_target(target)
    np.testing.assert_allclose(rel.compute_error(cfg), -absolute.compute_error(cfg), atol=1e-13)
    np.testing.assert_allclose(rel.compute_jacobian(cfg), -absolute.compute_jacobian(cfg), atol=1e-11)
    # moving root: palm relative to torso
    rel = mink.RelativeFrameTask("left_palm", "site", "torso_link", "body", position_cost=[1.0, 2.0, 3.0],
                                 orientation_cost=0.5, gain=0.8, lm_damping=0.3)
    rel.set_target(target)
    m = oc.model("g1")
    spec = oik.RelativeFrameTaskSpec(m.name2id("site", "left_palm"), "site", m.name2id("body", "torso_link"), "body",
                                     rel.cost, target.wxyz_xyz, 0.8, 0.3)
    o = oik.Configuration(m, q)
    e_ref, J_ref = oik.task_error_jacobian(o, spec)
    np.testing.assert_allclose(rel.compute_error(cfg), e_ref, atol=1e-13)
    np.testing.assert_allclose(rel.compute_jacobian(cfg), J_ref, atol=1e-11)
    J, Jfd = _fd_jacobian(cfg, rel)
    assert np.abs(J - Jfd).max() < 1e-5
    H_ref, c_ref = oik.task_qp_objective(o, spec)
    H, c = rel.compute_qp_objective(cfg)
    np.testing.assert_allclose(H, H_ref, atol=1e-11 * np.abs(H_ref).max())
    np.testing.assert_allclose(c, c_ref, atol=1e-12 * max(1.0, np.abs(c_ref).max()))
    post = mink.PostureTask(g1, 1.0); post.set_target(g1.key_qpos[0])
    v = mink.solve_ik(cfg, [rel, post], 1e-2, "mi355x", 1e-3, limits=[mink.ConfigurationLimit(g1)])
    v_ref = oik.solve_ik(m, o, [spec, oik.PostureTaskSpec(np.ones(g1.nv), g1.key_qpos[0])], 1e-2, 1e-3,
                         [oik.ConfigurationLimitSpec()])
    np.testing.assert_allclose(v, v_ref, atol=1e-9 * max(1.0, np.abs(v_ref).max()))
    rel.set_target_from_configuration(cfg)
    np.testing.assert_allclose(rel.compute_error(cfg), 0.0, atol=1e-14)


def test_frame_task_remaining_reference_behaviours(g1):
    """reference tests/test_frame_task.py:81-105,159-173."""
    cfg = mink.Configuration(g1)
    cfg.update_from_keyframe("stand")
    with pytest.raises(mink.TargetNotSet):
        mink.FrameTask("pelvis", "body", 1.0, 1.0).compute_jacobian(cfg)
    task = mink.FrameTask("pelvis", "body", position_cost=1.0, orientation_cost=1.0)
    task.set_target_from_configuration(cfg)
    pose = cfg.get_transform_frame_to_world("pelvis", "body")
    np.testing.assert_array_equal(task.transform_target_to_world.translation(), pose.translation())
    np.testing.assert_array_equal(task.transform_target_to_world.rotation().wxyz, pose.rotation().wxyz)
    # Levenberg–Marquardt damping has no effect when the error is zero
    damped = mink.FrameTask("pelvis", "body", position_cost=1.0, orientation_cost=1.0, lm_damping=1e-3)
    damped.set_target_from_configuration(cfg)
    H0, c0 = task.compute_qp_objective(cfg)
    H1, c1 = damped.compute_qp_objective(cfg)
    np.testing.assert_allclose(H1, H0, atol=1e-13)
    np.testing.assert_allclose(c1, c0, atol=1e-13)


def test_move_mocap_to_frame():
    """reference tests/test_utils.py:61-98: the mocap body lands on the frame's pose (Configuration stands where
    the reference has MjData)."""
    xml = """<mujoco><worldbody>
      <body pos=".1 -.1 0"><joint type="free" name="floating"/><geom type="sphere" size=".1" mass=".1"/>
        <body name="test"><joint type="hinge" name="hinge" range="0 1.57" limited="true"/>
          <geom type="sphere" size=".1" mass=".1"/></body></body>
      <body name="mocap" mocap="true" pos=".5 1 5" quat="1 1 0 0"><geom type="sphere" size=".1" mass=".1"/></body>
    </worldbody></mujoco>"""
    m = mink.loads_mjcf(xml)
    q = m.qpos0.copy()
    q[:3] = [0.3, -0.2, 0.7]
    q[3:7] = np.array([0.8, 0.2, -0.4, 0.1]) / np.linalg.norm([0.8, 0.2, -0.4, 0.1])
    q[7] = 0.5
    cfg = mink.Configuration(m, q)
    body = cfg.get_transform_frame_to_world("test", "body").wxyz_xyz
    mid = int(m.body_mocapid[m.name2id("body", "mocap")])
    assert not np.allclose(m.mocap_pos[mid], body[4:])
    mink.move_mocap_to_frame(m, cfg, "mocap", "test", "body")
    np.testing.assert_allclose(m.mocap_pos[mid], body[4:])
    np.testing.assert_allclose(m.mocap_quat[mid], body[:4])


def test_one_process_drives_several_devices():
    """SURVEY §8(e), second host model: one process, one handle per device, the host batch in contiguous row blocks
    (mink_amd.distributed.ShardedProblem).  A 1-GPU box lists device 0 several times: N handles, N threads, same rows.
    Results must be those of the single-handle call bit for bit (every instance is independent) — plain solve, fused loop,
    per-instance posture targets, taps."""
    import mink_amd as mink
    from mink_amd import _native as nat, workloads
    from mink_amd.distributed import ShardedProblem

    m = workloads.load_robot("g1")
    rng = np.random.default_rng(3)
    stand = m.key_qpos[m.name2id("key", "stand")]
    B = 1003                                               # uneven shards
    q = workloads.sample_q(m, rng, B, base_q=stand)
    one = mink.Configuration(m, q)
    many = mink.Configuration(m, q, device=[0, 0, 0])
    assert many.devices == [0, 0, 0]
    tgt = mink.Configuration(m, one.integrate(rng.normal(scale=0.15, size=(B, m.nv)), 1.0))
    tasks = []
    for s, ori in (("left_foot", 10.0), ("right_foot", 10.0), ("left_palm", 0.0), ("right_palm", 0.0)):
        t = mink.FrameTask(s, "site", 200.0, ori, lm_damping=1.0)
        t.set_target(tgt.get_transform_frame_to_world(s, "site"))
        tasks.append(t)
    post = mink.PostureTask(m, cost=1.0)
    post.set_target(np.tile(stand, (B, 1)) + 0.01 * rng.normal(size=(B, m.nq)))      # per-instance posture target
    tasks.append(post)
    hinge = {m.jnt_names[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] != 0}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, hinge)]
    v1 = mink.solve_ik(one, tasks, 5e-3, "mi355x", 1e-1, limits=lims)
    vN = mink.solve_ik(many, tasks, 5e-3, "mi355x", 1e-1, limits=lims)
    assert any(isinstance(p, ShardedProblem) for p in many._problems.values())
    np.testing.assert_array_equal(vN, v1)
    p1, pN = mink.build_ik(one, tasks, 5e-3, 1e-1, lims), mink.build_ik(many, tasks, 5e-3, 1e-1, lims)
    np.testing.assert_array_equal(pN.P, p1.P); np.testing.assert_array_equal(pN.q, p1.q)
    q1, _ = mink.solve_ik_steps(one, tasks, 5e-3, 5, damping=1e-1, limits=lims, update=False)
    qN, _ = mink.solve_ik_steps(many, tasks, 5e-3, 5, damping=1e-1, limits=lims, update=False)
    np.testing.assert_array_equal(qN, q1)
    assert mink.Configuration(m, q[:4], device="all").devices == list(range(nat.lib().mkh_device_count()))
    with pytest.raises(nat.MinkHipError, match="max_batch"):
        list(many._problems.values())[-1].solve(np.tile(q, (2, 1)), None, None, None, 1e-2, 1e-3)


def test_compile_memo_sees_costs_changed_in_place(g1):
    """solve_ik memoises the compiled handle on the task / limit OBJECTS (control loops call it thousands of times a second
    with the same ones) — by VALUE of everything the device descriptor holds: the reference's setters modify `cost` in place
    (tasks/frame_task.py set_position_cost), and a retuned task must not get the old descriptor."""
    rng = np.random.default_rng(3)
    from mink_amd import workloads
    q = workloads.sample_q(g1, rng, 4, base_q=g1.key_qpos[0])
    cfg = mink.Configuration(g1, q)
    tgt = mink.Configuration(g1, cfg.integrate(rng.normal(scale=0.1, size=(4, g1.nv)), 1.0))

    def make(pos_cost, gain, lm, posture_cost):
        ts = [mink.FrameTask(s, "site", pos_cost, 10.0, gain=gain, lm_damping=lm) for s in ("left_foot", "right_palm")]
        for t in ts:
            t.set_target(tgt.get_transform_frame_to_world(t.frame_name, "site"))
        post = mink.PostureTask(g1, posture_cost); post.set_target(g1.key_qpos[0])
        return ts + [post]

    lims = [mink.ConfigurationLimit(g1)]
    tasks = make(200.0, 1.0, 1.0, 1.0)
    v0 = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=lims)
    assert len(cfg._compile_memo) == 1
    np.testing.assert_array_equal(mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=lims), v0)     # memo hit: same handle
    assert len(cfg._compile_memo) == 1 and len(cfg._problems) == 1
    tasks[0].set_position_cost(20.0)                   # in place
    tasks[1].gain = 0.5
    tasks[1].lm_damping = 0.0
    tasks[2].set_cost(0.1)
    v1 = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=lims)
    fresh = make(200.0, 1.0, 1.0, 0.1)
    fresh[0].set_position_cost(20.0); fresh[1].gain = 0.5; fresh[1].lm_damping = 0.0
    v_ref = mink.solve_ik(mink.Configuration(g1, q), fresh, 5e-3, "mi355x", 1e-1, limits=lims)
    np.testing.assert_array_equal(v1, v_ref)
    assert np.abs(v1 - v0).max() > 1e-3
    # a limit retuned in place, and limits=None (the default ConfigurationLimit is built once per configuration)
    lims[0].gain = 0.5
    v2 = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=lims)
    v2_ref = mink.solve_ik(mink.Configuration(g1, q), fresh, 5e-3, "mi355x", 1e-1, limits=[mink.ConfigurationLimit(g1, gain=0.5)])
    np.testing.assert_array_equal(v2, v2_ref)
    d0 = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1)
    d1 = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1)
    np.testing.assert_array_equal(d0, d1)
    assert cfg._default_limits is not None and len(cfg._default_limits) == 1


def test_small_host_calls_equal_staged_calls(g1):
    """Host-pointer calls below 64 KB go through one pinned, device-visible buffer (the kernel reads and writes it across the
    bus), larger ones through staged copies: the same rows either way, bitwise — single solves, fused steps and the
    threshold-terminated loop."""
    from mink_amd import _native as nat, workloads
    import native_configs as nc
    nm = nat.NativeModel(g1)
    B = 96                                             # 96 x 924 B: staged; halves of 48 and single rows: pinned
    prob, dt, damping = nc.build("g1_c3", nm, B)
    rng = np.random.default_rng(12)
    base = g1.key_qpos[g1.name2id("key", "stand")]
    q, tg = workloads.make_batch(g1, nm, prob, rng, B, base_q=base)
    pt = base[None, :]
    for kw in ({}, {"n_steps": 3}, {"n_steps": 6, "until": (1e-3, 1e-2)}):
        big = prob.solve(q, tg, pt, None, dt, damping, **kw)
        for lo, hi in ((0, 48), (48, 96), (7, 8)):
            small = prob.solve(q[lo:hi], tg[lo:hi], pt, None, dt, damping, **kw)
            for a, b in zip(small, big):
                np.testing.assert_array_equal(a, b[lo:hi])
    # mkh_integrate: small (pinned) against large (staged) calls
    v = rng.normal(size=(B, g1.nv))
    qi = nm.integrate(np.tile(q, (2, 1)), np.tile(v, (2, 1)), 0.1)      # 192 rows: staged
    np.testing.assert_array_equal(nm.integrate(q[:20], v[:20], 0.1), qi[:20])
    np.testing.assert_array_equal(qi[:B], qi[B:])
