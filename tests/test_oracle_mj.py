"""Property tests that pin the oracle's restated MuJoCo arithmetic the same way the
reference's tests pin MuJoCo itself (the wheel is absent):
finite-difference Jacobians (reference tests/test_jacobians.py:41-108), frame pose
consistency (tests/test_configuration.py:36-53), integrate/differentiate inverses,
collision normal Jacobian vs finite differences of the distance."""

import numpy as np
import pytest

import oracle_configs as oc
from oracle import ik, lie, mjmath


def _rand_q(m, rng):
    q = np.array(m.qpos0)
    for j in range(m.njnt):
        a = m.jnt_qposadr[j]
        if m.jnt_type[j] == 0:
            q[a:a + 3] = rng.normal(size=3) * 0.3
            q[a + 3:a + 7] = lie.so3_exp(rng.normal(size=3))
        else:
            lo, hi = m.jnt_range[j] if m.jnt_limited[j] else (-np.pi, np.pi)
            q[a] = rng.uniform(lo, hi)
    return q


def _fd_jacobian(m, q, task, h=1e-6):
    cfg = ik.Configuration(m, q)
    e0, J = ik.task_error_jacobian(cfg, task)
    Jfd = np.zeros_like(J)
    for i in range(m.nv):
        dv = np.zeros(m.nv); dv[i] = h
        qp = cfg.integrate(dv, 1.0)
        e1, _ = ik.task_error_jacobian(ik.Configuration(m, qp), task)
        Jfd[:, i] = (e1 - e0) / h
    return J, Jfd


@pytest.mark.parametrize("robot", ["g1", "ur5e", "shadow_left"])
def test_frame_task_jacobian_fd(robot):
    m = oc.model(robot)
    rng = np.random.default_rng(42)
    for trial in range(3):
        q = _rand_q(m, rng)
        for ftype, n in (("site", m.nsite), ("body", m.nbody)):
            fid = int(rng.integers(1 if ftype == "body" else 0, n))
            target = lie.se3_exp(rng.normal(size=6))
            task = ik.FrameTaskSpec(fid, ftype, np.ones(6), target)
            J, Jfd = _fd_jacobian(m, q, task)
            assert np.abs(J - Jfd).max() < 1e-5


def test_relative_frame_task_jacobian_fd():
    m = oc.model("g1")
    rng = np.random.default_rng(7)
    q = _rand_q(m, rng)
    task = ik.RelativeFrameTaskSpec(m.name2id("site", "left_palm"), "site",
                                    m.name2id("body", "torso_link"), "body",
                                    np.ones(6), lie.se3_exp(rng.normal(size=6)))
    J, Jfd = _fd_jacobian(m, q, task)
    assert np.abs(J - Jfd).max() < 1e-5


def test_posture_and_com_jacobian_fd():
    m = oc.model("g1")
    rng = np.random.default_rng(5)
    q = _rand_q(m, rng)
    J, Jfd = _fd_jacobian(m, q, ik.PostureTaskSpec(np.ones(m.nv), _rand_q(m, rng)))
    hinge = np.arange(6, m.nv)
    assert np.abs(J - Jfd)[np.ix_(hinge, hinge)].max() < 1e-6
    J, Jfd = _fd_jacobian(m, q, ik.ComTaskSpec(np.ones(3), rng.normal(size=3)))
    assert np.abs(J - Jfd).max() < 1e-6


def test_site_pose_matches_chain_product():
    """Independent FK: compose homogeneous transforms down the chain with scipy-free
    numpy matrices and compare with mj_kinematics' site pose."""
    m = oc.model("ur5e")
    rng = np.random.default_rng(0)
    q = _rand_q(m, rng)
    cfg = ik.Configuration(m, q)
    sid = m.name2id("site", "attachment_site")

    def hom(quat, pos):
        T = np.eye(4); T[:3, :3] = lie.so3_as_matrix(np.asarray(quat)); T[:3, 3] = pos
        return T

    chain = []
    b = int(m.site_bodyid[sid])
    while b:
        chain.append(b); b = int(m.body_parentid[b])
    T = np.eye(4)
    for b in reversed(chain):
        T = T @ hom(m.body_quat[b], m.body_pos[b])
        for j in range(m.body_jntadr[b], m.body_jntadr[b] + m.body_jntnum[b]):
            ang = q[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
            ax = m.jnt_axis[j]
            Rj = hom(lie.so3_exp(ax * ang), np.zeros(3))
            P = np.eye(4); P[:3, 3] = m.jnt_pos[j]
            Pi = np.eye(4); Pi[:3, 3] = -m.jnt_pos[j]
            T = T @ P @ Rj @ Pi
    T = T @ hom(m.site_quat[sid], m.site_pos[sid])
    np.testing.assert_allclose(cfg.data.site_xpos[sid], T[:3, 3], atol=1e-14)
    np.testing.assert_allclose(cfg.data.site_xmat[sid].reshape(3, 3), T[:3, :3], atol=1e-14)
    # UR5e 'home' end-effector position is a well known number for this MJCF
    home = m.key_qpos[m.name2id("key", "home")]
    np.testing.assert_allclose(ik.Configuration(m, home).data.site_xpos[sid],
                               [0.4920, 0.1340, 0.4880], atol=2e-4)


def test_integrate_differentiate_roundtrip():
    m = oc.model("g1")
    rng = np.random.default_rng(1)
    q = _rand_q(m, rng)
    v = rng.normal(size=m.nv) * 0.3
    q2 = q.copy()
    mjmath.mj_integratePos(m, q2, v, 0.7)
    back = np.empty(m.nv)
    mjmath.mj_differentiatePos(m, back, 0.7, q, q2)
    np.testing.assert_allclose(back, v, atol=1e-12)


def test_subtree_com_is_mass_weighted_mean():
    m = oc.model("g1")
    cfg = ik.Configuration(m, _rand_q(m, np.random.default_rng(2)))
    sub = [b for b in range(1, m.nbody) if m.body_rootid[b] == 1]
    com = sum(m.body_mass[b] * cfg.data.xipos[b] for b in sub) / sum(m.body_mass[b] for b in sub)
    np.testing.assert_allclose(cfg.data.subtree_com[1], com, atol=1e-13)


def test_collision_normal_jacobian_fd():
    """G row = −nᵀ(J2−J1) must equal −∂dist/∂q for separated capsules
    (the reference pins it against MuJoCo's efc_J, tests/test_collision_avoidance_limit.py:65-111)."""
    m = oc.model("shadow_left")
    d = np.load(oc.GOLDEN + "/ik_shadow_c4.npz")
    pairs = [tuple(p) for p in np.load(oc.GOLDEN + "/shadow_c4_geom_pairs.npy")]
    spec = ik.CollisionAvoidanceLimitSpec(pairs, collision_detection_distance=0.05)
    q = d["q"][0]
    cfg = ik.Configuration(m, q)
    G, h = ik.limit_inequalities(cfg, spec, 2e-3)
    checked = 0
    for k, (g1, g2) in enumerate(pairs):
        if not np.isfinite(h[k]):
            continue
        d0 = mjmath.mj_geomDistance(m, cfg.data, g1, g2, 0.05, None)
        if d0 <= 1e-4:
            continue
        fd = np.zeros(m.nv)
        for i in range(m.nv):
            dv = np.zeros(m.nv); dv[i] = 1e-7
            c2 = ik.Configuration(m, cfg.integrate(dv, 1.0))
            fd[i] = (mjmath.mj_geomDistance(m, c2.data, g1, g2, 0.06, None) - d0) / 1e-7
        np.testing.assert_allclose(-G[k], fd, atol=2e-5)
        checked += 1
    assert checked >= 3
