"""The Task / Limit PLUGIN API on the device path (north_star: "keeps … its Task/Limit plugin API").

The reference's extension points are Task.compute_error / compute_jacobian (mink/tasks/task.py:81-103) and
Limit.compute_qp_inequalities (mink/limits/limit.py:34-57).  A caller-defined subclass that implements them with
numpy for the whole batch reaches the kernel as dense rows (mkh_solve_dense, include/minkhip.h) and is folded into
the QP exactly like a built-in task — so a user re-implementation of a built-in must give the built-in's answer."""

import numpy as np
import pytest

import mink_amd as mink
import oracle_configs as oc
from mink_amd import workloads
from oracle import ik as oik

pytestmark = pytest.mark.gpu


class UserFrameTask(mink.Task):
    """FrameTask written against the public API only, the way a mink user would (frame_task.py:95-146):
    e = target.minus(frame),  J = −jlog(T_tb)·ᴮJ."""

    def __init__(self, frame_name, frame_type, position_cost, orientation_cost, gain=1.0, lm_damping=0.0):
        super().__init__(cost=np.array([position_cost] * 3 + [orientation_cost] * 3, dtype=np.float64), gain=gain,
                         lm_damping=lm_damping)
        self.frame_name, self.frame_type = frame_name, frame_type
        self.target = None

    def set_target(self, T):
        self.target = T.copy()

    def _poses(self, configuration):
        T = configuration.get_transform_frame_to_world(self.frame_name, self.frame_type).wxyz_xyz
        T = T.reshape(-1, 7)
        tg = np.broadcast_to(self.target.wxyz_xyz.reshape(-1, 7), T.shape)
        return [mink.SE3(t) for t in T], [mink.SE3(t) for t in tg]

    def compute_error(self, configuration):
        Tf, Tt = self._poses(configuration)
        return np.array([t.minus(f) for f, t in zip(Tf, Tt)])

    def compute_jacobian(self, configuration):
        Tf, Tt = self._poses(configuration)
        Jb = configuration.get_frame_jacobian(self.frame_name, self.frame_type).reshape(len(Tf), 6, -1)
        return np.array([-(t.inverse() @ f).jlog() @ J for f, t, J in zip(Tf, Tt, Jb)])


class UserVelocityLimit(mink.Limit):
    """VelocityLimit as generic rows G·Δq ≤ h (velocity_limit.py:71-101) instead of the device's box."""

    def __init__(self, model, vmax):
        self.idx = [int(model.jnt_dofadr[j]) for j in range(model.njnt) if model.jnt_type[j] != 0]
        self.vmax, self.nv = vmax, model.nv

    def compute_qp_inequalities(self, configuration, dt):
        P = np.eye(self.nv)[self.idx]
        return mink.Constraint(G=np.vstack([P, -P]), h=np.full(2 * len(self.idx), dt * self.vmax))


def _ur5e_batch(B, seed=3):
    m = workloads.load_robot("ur5e")
    rng = np.random.default_rng(seed)
    home = m.key_qpos[m.name2id("key", "home")]
    q = workloads.sample_q(m, rng, B, base_q=home)
    cfg = mink.Configuration(m, q)
    tg = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.2, size=(B, m.nv)), 1.0)) \
        .get_transform_frame_to_world("attachment_site", "site")
    return m, cfg, tg, home


def test_user_frame_task_matches_builtin():
    B = 96
    m, cfg, tg, home = _ur5e_batch(B)
    builtin = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=0.7, gain=0.9, lm_damping=1.0)
    user = UserFrameTask("attachment_site", "site", 1.0, 0.7, gain=0.9, lm_damping=1.0)
    builtin.set_target(tg); user.set_target(tg)
    post = mink.PostureTask(m, cost=1e-2); post.set_target(home)
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: np.pi for n in m.jnt_names})]
    assert user._is_dense() and not builtin._is_dense()
    v_ref = mink.solve_ik(cfg, [builtin, post], 2e-3, "mi355x", 1e-3, limits=lims)
    v = mink.solve_ik(cfg, [user, post], 2e-3, "mi355x", 1e-3, limits=lims)
    err = np.abs(v - v_ref).max() / max(1.0, np.abs(v_ref).max())
    print("user FrameTask vs built-in: max rel err %.2e" % err)
    assert err < 1e-10
    # the base-class methods of the plugin task go through the same device route (task.py:105-138)
    H, c = user.compute_qp_objective(cfg)
    Hb, cb = builtin.compute_qp_objective(cfg)
    np.testing.assert_allclose(H, Hb, rtol=0, atol=1e-11 * np.abs(Hb).max())
    np.testing.assert_allclose(c, cb, rtol=0, atol=1e-11 * max(1.0, np.abs(cb).max()))
    # build_ik, unbatched configuration: reference shapes
    one = mink.Configuration(m, cfg.q[0])
    user.set_target(mink.SE3(tg.wxyz_xyz[0])); builtin.set_target(mink.SE3(tg.wxyz_xyz[0]))
    p1, p2 = mink.build_ik(one, [user, post], 2e-3, 1e-3, lims), mink.build_ik(one, [builtin, post], 2e-3, 1e-3, lims)
    assert p1.P.shape == (m.nv, m.nv) and p1.G.shape == p2.G.shape
    np.testing.assert_allclose(p1.P, p2.P, rtol=0, atol=1e-11 * np.abs(p2.P).max())
    v1 = mink.solve_ik(one, [user, post], 2e-3, "mi355x", 1e-3, limits=lims)
    assert v1.shape == (m.nv,)
    np.testing.assert_allclose(v1, v_ref[0], rtol=0, atol=1e-10 * max(1.0, np.abs(v_ref[0]).max()))


def test_user_limit_matches_builtin_box():
    B = 128
    m, cfg, tg, home = _ur5e_batch(B, seed=5)
    ft = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    ft.set_target(tg)
    vmax = 0.6                                         # tight: most instances saturate several dofs
    ref = mink.solve_ik(cfg, [ft], 2e-2, "mi355x", 1e-3,
                        limits=[mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: vmax for n in m.jnt_names})])
    user = UserVelocityLimit(m, vmax)
    assert user._is_dense()
    v = mink.solve_ik(cfg, [ft], 2e-2, "mi355x", 1e-3, limits=[mink.ConfigurationLimit(m), user])
    sat = (np.abs(np.abs(ref) - vmax) < 1e-9).sum()
    print("saturated dofs:", sat, "of", ref.size)
    assert sat > B
    np.testing.assert_allclose(v, ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))


def test_dense_rows_against_oracle_with_collisions_and_inactive_rows():
    """Raw ABI: random dense task rows + dense limit rows (some inactive: h = +inf) next to built-in tasks, box limits
    and collision half-spaces; every instance against the numpy oracle."""
    from mink_amd import _native as nat
    import native_configs as nc
    om = oc.model("ur5e")
    d = np.load(oc.GOLDEN + "/ik_ur5e_coll.npz")
    B, nv = len(d["q"]), om.nv
    rng = np.random.default_rng(8)
    K, M = 9, 5
    e = rng.normal(scale=0.05, size=(B, K)); J = rng.normal(size=(B, K, nv))
    cost = np.array([1.0, 0.5, 0.0, 2.0, 1.0, 1.0, 0.3, 0.3, 0.3])
    G = rng.normal(size=(B, M, nv)); h = rng.uniform(0.0, 0.02, size=(B, M))
    h[rng.uniform(size=(B, M)) < 0.3] = np.inf
    nm = nat.NativeModel(om)
    base, (ft, _, _), _, dt, damping = nc.build_ext("ur5e_coll", nm, d, B)
    g = {"frame_type": "geom", "frame_id": om.name2id("geom", "wrist_2_link"), "cost": [0.5, 0.5, 0.5, 0.1, 0.2, 0.3],
         "gain": 0.7, "lm_damping": 0.0}
    col = {"geom_id_pairs": d["geom_id_pairs"], "gain": 0.85, "minimum_distance_from_collisions": 0.005,
           "collision_detection_distance": 0.3, "bound_relaxation": 0.0}
    prob = nat.NativeProblem(nm, frame_tasks=[nc._ft(om, "attachment_site", "site", 1.0, 1.0, 1.0), g],
                             configuration_limits=[nc._cfg_limit(om)], collision_limits=[col],
                             velocity_limits=[nc._vel_limit(om)], max_batch=B,
                             dense_tasks=[{"cost": cost[:6], "gain": 0.8, "lm_damping": 0.5}, {"cost": cost[6:], "gain": 1.0}],
                             dense_limit_rows=M)
    dense = {"task_e": e, "task_J": J, "limit_G": G, "limit_h": h}
    v, st, taps = prob.solve(d["q"], ft, None, None, dt, damping, taps=["H", "c", "task_e", "task_J"], dense=dense)
    assert prob.last_kernel().removesuffix("+wide").endswith("_31"), prob.last_kernel()
    np.testing.assert_array_equal(taps["task_e"][:, 12:], e)
    np.testing.assert_array_equal(taps["task_J"][:, 12:], J)
    v2, st2 = prob.solve(d["q"], ft, None, None, dt, damping, dense=dense)
    assert prob.last_kernel().removesuffix("+wide").endswith("_30"), prob.last_kernel()
    worst = 0.0
    for i in range(B):
        m, tasks, limits, dt_o, damp_o = oc.ur5e_coll(d, i)
        tasks = tasks + [oik.DenseTaskSpec(e[i, :6], J[i, :6], cost[:6], 0.8, 0.5), oik.DenseTaskSpec(e[i, 6:], J[i, 6:], cost[6:])]
        limits = limits + [oik.DenseLimitSpec(G[i], h[i])]
        cfg = oik.Configuration(m, d["q"][i])
        P, c, Go, ho = oik.build_ik(cfg, tasks, dt_o, damp_o, limits)
        np.testing.assert_allclose(taps["H"][i], P, rtol=0, atol=1e-11 * np.abs(P).max())
        np.testing.assert_allclose(taps["c"][i], c, rtol=0, atol=1e-11 * max(1.0, np.abs(c).max()))
        try:
            v_ref = oik.solve_ik(m, cfg, tasks, dt_o, damp_o, limits)
        except oik.qp_gi.Infeasible:
            assert st[i] & 2, (i, st[i])               # random half-spaces can contradict the box: both must say so
            continue
        assert st[i] & ~1 == 0 and st2[i] == st[i], (i, st[i])
        worst = max(worst, np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()),
                    np.abs(v2[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
    print("dense rows + collisions vs oracle: max rel err %.2e" % worst)
    assert worst < 1e-8


def test_plugin_errors():
    m, cfg, tg, home = _ur5e_batch(4)

    class NoJacobian(mink.Task):
        def compute_error(self, configuration):
            return np.zeros(3)

    with pytest.raises(mink.TaskDefinitionError, match="compute_error and compute_jacobian"):
        mink.solve_ik(cfg, [NoJacobian(cost=np.ones(3))], 1e-2, "mi355x", 1e-3)

    class BadShape(mink.Task):
        def compute_error(self, configuration):
            return np.zeros(2)

        def compute_jacobian(self, configuration):
            return np.zeros((3, configuration.nv))

    with pytest.raises(mink.TaskDefinitionError, match="must return"):
        mink.solve_ik(cfg, [BadShape(cost=np.ones(3))], 1e-2, "mi355x", 1e-3)
    user = UserFrameTask("attachment_site", "site", 1.0, 1.0)
    user.set_target(tg)
    with pytest.raises(mink.TaskDefinitionError, match="solve_ik_steps"):
        mink.solve_ik_steps(cfg, [user], 1e-2, 3, damping=1e-3)


def test_partial_overrides_of_builtin_tasks():
    """The reference calls compute_error / compute_jacobian / compute_qp_objective through the instance
    (/root/reference/mink/solve_ik.py:13-22, tasks/task.py:105-138), so a subclass of a built-in task that overrides
    only ONE of them changes the answer; the half it inherits is the built-in's (round-2 advisor finding: the override
    of compute_jacobian alone used to be ignored, the override of compute_error alone raised)."""
    B = 64
    m, cfg, tg, home = _ur5e_batch(B, seed=11)
    post = mink.PostureTask(m, cost=1e-2); post.set_target(home)
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: np.pi for n in m.jnt_names})]

    class HalfJacobian(mink.FrameTask):                    # J only
        def compute_jacobian(self, configuration):
            return 0.5 * super().compute_jacobian(configuration)

    class PositionError(mink.FrameTask):                   # e only
        def compute_error(self, configuration):
            e = np.array(super().compute_error(configuration))
            e[..., 3:] = 0.0
            return e

    builtin = mink.FrameTask("attachment_site", "site", 1.0, 0.7, gain=0.9, lm_damping=1.0); builtin.set_target(tg)
    e_b, J_b = builtin.compute_error(cfg), builtin.compute_jacobian(cfg)

    class Scratch(mink.Task):                              # the same rows written from scratch: the expected answer
        def __init__(self, e, J, cost, **kw):
            super().__init__(cost=cost, **kw)
            self.e, self.J = e, J

        def compute_error(self, configuration):
            return self.e

        def compute_jacobian(self, configuration):
            return self.J

    for cls, e_x, J_x in ((HalfJacobian, e_b, 0.5 * J_b),
                          (PositionError, np.concatenate([e_b[:, :3], np.zeros((B, 3))], axis=1), J_b)):
        t = cls("attachment_site", "site", 1.0, 0.7, gain=0.9, lm_damping=1.0); t.set_target(tg)
        assert t._is_dense() and t._builtin_class() is mink.FrameTask
        np.testing.assert_allclose(t.compute_error(cfg), e_x, rtol=0, atol=1e-14)
        np.testing.assert_allclose(t.compute_jacobian(cfg), J_x, rtol=0, atol=1e-14)
        v = mink.solve_ik(cfg, [t, post], 2e-3, "mi355x", 1e-3, limits=lims)
        ref = Scratch(e_x, J_x, builtin.cost.copy(), gain=0.9, lm_damping=1.0)
        v_ref = mink.solve_ik(cfg, [ref, post], 2e-3, "mi355x", 1e-3, limits=lims)
        np.testing.assert_allclose(v, v_ref, rtol=0, atol=1e-11 * max(1.0, np.abs(v_ref).max()))
        v_builtin = mink.solve_ik(cfg, [builtin, post], 2e-3, "mi355x", 1e-3, limits=lims)
        assert np.abs(v - v_builtin).max() > 1e-3          # the override is not ignored
        H, c = t.compute_qp_objective(cfg)                 # inherited objective: from the overridden rows
        Hr, cr = ref.compute_qp_objective(cfg)
        np.testing.assert_allclose(H, Hr, rtol=0, atol=1e-12 * np.abs(Hr).max())
        np.testing.assert_allclose(c, cr, rtol=0, atol=1e-12 * max(1.0, np.abs(cr).max()))



def test_compute_qp_objective_overrides():
    """The reference only ever calls Task.compute_qp_objective (/root/reference/mink/solve_ik.py:18-21), so a subclass that
    returns its own (H, c) changes the QP there (round-4 review, missing #5: refused here until round 5).  The device takes
    rows: the override's objective is factored into nv rows with JᵀJ = H, Jᵀe = c (mink_amd.tasks.objective_to_rows)."""
    B = 64
    m, cfg, tg, home = _ur5e_batch(B, seed=17)
    nv = m.nv
    post = mink.PostureTask(m, cost=1e-2); post.set_target(home)
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: np.pi for n in m.jnt_names})]

    class Doubled(mink.FrameTask):                         # an override on top of the inherited objective (super() call)
        def compute_qp_objective(self, configuration):
            H, c = super().compute_qp_objective(configuration)
            return mink.Objective(2.0 * H, 2.0 * c)

    t = Doubled("attachment_site", "site", 1.0, 0.7, gain=0.9, lm_damping=1.0); t.set_target(tg)
    assert t._is_dense() and t._objective_overridden()
    r2 = np.sqrt(2.0)                                      # 2·(JᵀW²J + μI, −JᵀW·We) = the same task with costs·√2
    twin = mink.FrameTask("attachment_site", "site", r2 * 1.0, r2 * 0.7, gain=0.9, lm_damping=1.0); twin.set_target(tg)
    v = mink.solve_ik(cfg, [t, post], 2e-3, "mi355x", 1e-3, limits=lims)
    v_ref = mink.solve_ik(cfg, [twin, post], 2e-3, "mi355x", 1e-3, limits=lims)
    err = np.abs(v - v_ref).max() / max(1.0, np.abs(v_ref).max())
    print("doubled objective vs costs x sqrt 2: max rel err %.2e" % err)
    assert err < 1e-9
    p1, p2 = mink.build_ik(cfg, [t, post], 2e-3, 1e-3, lims), mink.build_ik(cfg, [twin, post], 2e-3, 1e-3, lims)
    np.testing.assert_allclose(p1.P, p2.P, rtol=0, atol=1e-11 * np.abs(p2.P).max())
    np.testing.assert_allclose(p1.q, p2.q, rtol=0, atol=1e-11 * max(1.0, np.abs(p2.q).max()))
    # the rows of such a task are still the built-in's (compute_error / compute_jacobian are inherited)
    np.testing.assert_allclose(t.compute_error(cfg), twin.compute_error(cfg), rtol=0, atol=1e-14)

    class RawObjective(mink.Task):                         # written from scratch: rank-deficient H, c partly outside its range
        def __init__(self, H, c):
            super().__init__(cost=np.zeros(1))
            self.H, self.c = H, c

        def compute_error(self, configuration):
            raise AssertionError("the reference never calls this when compute_qp_objective is overridden")

        def compute_jacobian(self, configuration):
            raise AssertionError("the reference never calls this when compute_qp_objective is overridden")

        def compute_qp_objective(self, configuration):
            return mink.Objective(self.H, self.c)

    rng = np.random.default_rng(5)
    A = rng.normal(size=(B, 2, nv))
    H_raw = 3.0 * np.einsum("bki,bkj->bij", A, A)          # rank 2 of 6, per instance
    c_raw = rng.normal(size=(B, nv))                       # generic: not in the range of H
    builtin = mink.FrameTask("attachment_site", "site", 1.0, 0.7, gain=0.9, lm_damping=1.0); builtin.set_target(tg)
    raw = RawObjective(H_raw, c_raw)
    p0 = mink.build_ik(cfg, [builtin, post], 2e-3, 1e-3, lims)
    p1 = mink.build_ik(cfg, [builtin, raw, post], 2e-3, 1e-3, lims)
    np.testing.assert_allclose(p1.P, p0.P + H_raw, rtol=0, atol=1e-12 * np.abs(p1.P).max())
    np.testing.assert_allclose(p1.q, p0.q + c_raw, rtol=0, atol=1e-12 * max(1.0, np.abs(p1.q).max()))
    v = mink.solve_ik(cfg, [builtin, raw, post], 2e-3, "mi355x", 1e-3, limits=lims)
    from oracle import qp_gi
    worst = 0.0
    for i in range(B):                                     # the reference's QP on the reference's stacked (P, q, G, h)
        dq = qp_gi.solve_qp(p0.P[i] + H_raw[i], p0.q[i] + c_raw[i], p0.G[i] if p0.G.ndim == 3 else p0.G,
                            p0.h[i] if p0.h.ndim == 2 else p0.h)
        worst = max(worst, np.abs(v[i] - dq / 2e-3).max() / max(1.0, np.abs(dq / 2e-3).max()))
    print("raw (H, c) objective vs Goldfarb-Idnani on the stacked QP: max rel err %.2e" % worst)
    assert worst < 1e-8
    # one shared (H, c) for the whole batch, and the unbatched configuration
    raw1 = RawObjective(H_raw[0], c_raw[0])
    v_sh = mink.solve_ik(cfg, [builtin, raw1, post], 2e-3, "mi355x", 1e-3, limits=lims)
    np.testing.assert_allclose(v_sh[0], v[0], rtol=0, atol=1e-10 * max(1.0, np.abs(v[0]).max()))
    # what the reference's solver would refuse is refused with a message, not solved
    with pytest.raises(mink.TaskDefinitionError, match="positive semi-definite"):
        mink.solve_ik(cfg, [builtin, RawObjective(-np.eye(nv), np.zeros(nv)), post], 2e-3, "mi355x", 1e-3, limits=lims)
    with pytest.raises(mink.TaskDefinitionError, match="must return H"):
        mink.solve_ik(cfg, [builtin, RawObjective(np.eye(nv + 1), np.zeros(nv)), post], 2e-3, "mi355x", 1e-3, limits=lims)

    # a from-scratch task that defines ONLY its objective has no error / Jacobian to report (the rows the device folds in are a
    # factorisation of (H, c), not mink's (e, J)): asking for them raises instead of returning those rows (round-5 advisor finding)
    class OnlyObjective(mink.Task):
        def __init__(self):
            super().__init__(cost=np.zeros(1))

        def compute_qp_objective(self, configuration):
            return mink.Objective(np.eye(nv), np.ones(nv))

    only = OnlyObjective()
    for method in (only.compute_error, only.compute_jacobian):
        with pytest.raises(mink.TaskDefinitionError, match="compute_qp_objective only"):
            method(cfg)
    v_only = mink.solve_ik(cfg, [builtin, only, post], 2e-3, "mi355x", 1e-3, limits=lims)
    assert np.isfinite(v_only).all()


def test_user_box_rows_on_g1_do_not_use_tableau_rows():
    """A caller-defined limit [I; −I] on G1 is 74 rows against 64 − 43 = 21 half-space rows per wavefront (round-2 advisor
    finding: it always failed with ROW_OVERFLOW).  Single-entry rows are folded into lo ≤ Δq ≤ hi on the host
    (MkhDenseRows.limit_lo / limit_hi); a few general rows ride along as half-spaces."""
    B = 48
    m = workloads.load_robot("g1")
    rng = np.random.default_rng(4)
    stand = m.key_qpos[m.name2id("key", "stand")]
    q = workloads.sample_q(m, rng, B, base_q=stand)
    cfg = mink.Configuration(m, q)
    tgt = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.15, size=(B, m.nv)), 1.0))
    tasks = []
    for s, ori in (("left_foot", 10.0), ("right_foot", 10.0), ("left_palm", 0.0), ("right_palm", 0.0)):
        t = mink.FrameTask(s, "site", 200.0, ori, lm_damping=1.0)
        t.set_target(tgt.get_transform_frame_to_world(s, "site"))
        tasks.append(t)
    post = mink.PostureTask(m, cost=1.0); post.set_target(stand); tasks.append(post)
    hinge = [m.jnt_names[j] for j in range(m.njnt) if m.jnt_type[j] != 0]
    vmax = 1.5
    ref = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1,
                        limits=[mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: vmax for n in hinge})])
    user = UserVelocityLimit(m, vmax)
    assert len(user.idx) == 37 and 2 * len(user.idx) > 64 - m.nv
    v = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=[mink.ConfigurationLimit(m), user])
    assert (np.abs(np.abs(ref[:, 6:]) - vmax) < 1e-9).sum() > 4 * B      # many dofs sit on the user's bound
    np.testing.assert_allclose(v, ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))

    class Mixed(mink.Limit):                   # box rows + 3 general rows, one of them inactive
        def compute_qp_inequalities(self, configuration, dt):
            P = np.eye(m.nv)[user.idx]
            Gg = np.zeros((3, m.nv)); Gg[0, 6:12] = 1.0; Gg[1, 12:18] = -1.0; Gg[2, 20:24] = 1.0
            return mink.Constraint(G=np.vstack([P, Gg, -P]), h=np.concatenate([np.full(37, dt * vmax), [1e-3, 2e-3, np.inf],
                                                                               np.full(37, dt * vmax)]))

    v_m, st = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=[mink.ConfigurationLimit(m), Mixed()], return_status=True)
    assert (st & ~1 == 0).all()
    dq = v_m * 5e-3
    assert (dq[:, 6:12].sum(axis=1) <= 1e-3 + 1e-12).all() and (-dq[:, 12:18].sum(axis=1) <= 2e-3 + 1e-12).all()
    assert (np.abs(dq[:, 6:]) <= 5e-3 * vmax + 1e-12).all()
    from oracle import ik as oik2
    import oracle_configs as oc2
    # the oracle's dense pipeline on a few instances: every row as a general constraint, like the reference
    om = oc2.model("g1")
    for i in range(0, B, 12):
        fts = np.stack([t.transform_target_to_world.wxyz_xyz[i] for t in tasks[:4]])
        mo, otasks, olims, dt_o, damp_o = oc2.g1_c3(fts, stand)
        c = Mixed().compute_qp_inequalities(cfg, dt_o)
        olims = [olims[0], oik2.DenseLimitSpec(c.G, c.h)]
        v_ref = oik2.solve_ik(mo, q[i], otasks, dt_o, damp_o, olims)
        np.testing.assert_allclose(v_m[i], v_ref, rtol=0, atol=1e-8 * max(1.0, np.abs(v_ref).max()))

    # More general rows than the 64 − nv = 21 a wavefront holds: round 3 refused them (LimitDefinitionError); the reference
    # stacks any number (mink/solve_ik.py:25-40).  Now the wavefront kernel keeps the tightest 21 and the instances in which a
    # dropped row is violated are solved again by the workgroup-per-problem kernel with all 30 — the oracle's answer.
    class Many(mink.Limit):
        def compute_qp_inequalities(self, configuration, dt):
            rng2 = np.random.default_rng(0)
            G30 = np.zeros((30, m.nv)); G30[:, 6:] = rng2.normal(size=(30, m.nv - 6))
            return mink.Constraint(G=G30, h=np.full(30, 2e-3))

    v_w, st_w = mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=[mink.ConfigurationLimit(m), Many()], return_status=True)
    assert (st_w & ~1 == 0).all(), np.unique(st_w)
    assert list(cfg._problems.values())[-1].last_kernel().endswith("+wide")
    cm = Many().compute_qp_inequalities(cfg, 5e-3)
    assert ((cm.G @ (v_w * 5e-3).T).T <= cm.h + 1e-10).all()
    nbind = 0
    for i in range(0, B, 12):
        fts = np.stack([t.transform_target_to_world.wxyz_xyz[i] for t in tasks[:4]])
        mo, otasks, olims, dt_o, damp_o = oc2.g1_c3(fts, stand)
        v_ref = oik2.solve_ik(mo, q[i], otasks, dt_o, damp_o, [olims[0], oik2.DenseLimitSpec(cm.G, cm.h)])
        np.testing.assert_allclose(v_w[i], v_ref, rtol=0, atol=1e-8 * max(1.0, np.abs(v_ref).max()))
        nbind = max(nbind, int((np.abs(cm.G @ (v_ref * dt_o) - cm.h) < 1e-9).sum()))
    print("30 general rows on G1: up to %d binding at the solution" % nbind)

    class TooMany(mink.Limit):
        def compute_qp_inequalities(self, configuration, dt):
            rng2 = np.random.default_rng(0)
            return mink.Constraint(G=rng2.normal(size=(460, m.nv)), h=np.ones(460))

    with pytest.raises(mink.LimitDefinitionError, match="general rows"):
        mink.solve_ik(cfg, tasks, 5e-3, "mi355x", 1e-1, limits=[TooMany()])


def test_nested_evaluations_do_not_evict_the_outer_handle():
    """A caller-defined task may evaluate many built-in ones inside compute_error (each a compiled descriptor in the
    Configuration's LRU cache); the handle of the outer solve must survive that (round-2 advisor finding)."""
    import sys
    sik = sys.modules["mink_amd.solve_ik"]                 # (the package attribute of that name is the function)
    B = 8
    m, cfg, tg, home = _ur5e_batch(B, seed=2)

    class Busy(mink.Task):
        def compute_error(self, configuration):
            acc = np.zeros((configuration.batch_size, 6))
            for i in range(sik.PROBLEM_CACHE_SIZE + 4):            # distinct costs ⇒ distinct descriptors
                t = mink.FrameTask("attachment_site", "site", 1.0 + 0.01 * i, 1.0); t.set_target(tg)
                acc += t.compute_error(configuration)
            return acc / (sik.PROBLEM_CACHE_SIZE + 4)

        def compute_jacobian(self, configuration):
            t = mink.FrameTask("attachment_site", "site", 1.0, 1.0); t.set_target(tg)
            return t.compute_jacobian(configuration)

    builtin = mink.FrameTask("attachment_site", "site", 1.0, 1.0); builtin.set_target(tg)
    v_ref = mink.solve_ik(cfg, [builtin], 2e-3, "mi355x", 1e-3)
    v = mink.solve_ik(cfg, [Busy(cost=np.ones(6))], 2e-3, "mi355x", 1e-3)
    np.testing.assert_allclose(v, v_ref, rtol=0, atol=1e-10 * max(1.0, np.abs(v_ref).max()))
    assert len(cfg._problems) <= sik.PROBLEM_CACHE_SIZE and not cfg._pinned_problems
