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
    assert prob.last_kernel().endswith("_31"), prob.last_kernel()
    np.testing.assert_array_equal(taps["task_e"][:, 12:], e)
    np.testing.assert_array_equal(taps["task_J"][:, 12:], J)
    v2, st2 = prob.solve(d["q"], ft, None, None, dt, damping, dense=dense)
    assert prob.last_kernel().endswith("_30"), prob.last_kernel()
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
