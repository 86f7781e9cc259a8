"""Host-side SO3 / SE3 methods that the hot path itself evaluates on the device (jlog, ljacinv, Q) — the classes a
mink caller holds offer them too.  Checked against the oracle restatement (pinned on fixtures recorded from the
real mink.lie) and through the reference's own identities (tests/test_lie_operations.py:22-79)."""

import numpy as np
import pytest

from mink_amd.lie import SE3, SO3, MatrixLieGroup, RollPitchYaw
from oracle import lie as ol


@pytest.mark.parametrize("scale", [1.0, 1e-3, 1e-7, 0.0])
def test_ljacinv_jlog_match_the_oracle(scale):
    rng = np.random.default_rng(3)
    for _ in range(20):
        w = rng.normal(size=3) * scale
        np.testing.assert_allclose(SO3.ljacinv(w), ol.so3_ljacinv(w), atol=1e-14)
        c = np.concatenate([rng.normal(size=3), w])
        np.testing.assert_allclose(SE3.ljacinv(c), ol.se3_ljacinv(c), atol=1e-13)
        T = SE3.exp(c)
        np.testing.assert_allclose(T.jlog(), ol.se3_jlog(T.wxyz_xyz), atol=1e-12)


def test_left_and_right_jacobians_are_inverse_pairs_and_batched():
    rng = np.random.default_rng(4)
    w = rng.normal(size=(7, 3))
    c = rng.normal(size=(7, 6))
    I3, I6 = np.eye(3), np.eye(6)
    np.testing.assert_allclose(SO3.ljac(w) @ SO3.ljacinv(w), np.broadcast_to(I3, (7, 3, 3)), atol=1e-12)
    np.testing.assert_allclose(SO3.rjac(w) @ SO3.rjacinv(w), np.broadcast_to(I3, (7, 3, 3)), atol=1e-12)
    np.testing.assert_allclose(SE3.ljac(c) @ SE3.ljacinv(c), np.broadcast_to(I6, (7, 6, 6)), atol=1e-11)
    np.testing.assert_allclose(SE3.rjac(c) @ SE3.rjacinv(c), np.broadcast_to(I6, (7, 6, 6)), atol=1e-11)
    for i in range(7):                                         # batch = the single-instance results stacked
        np.testing.assert_allclose(SE3.ljacinv(c)[i], SE3.ljacinv(c[i]), atol=0)


@pytest.mark.parametrize("group,dim", [(SO3, 3), (SE3, 6)])
def test_jlog_is_the_derivative_of_log_on_the_right(group, dim):
    """reference tests/test_lie_operations.py:55-79: log(T ⊕ δ) ≈ log(T) + jlog(T)·δ."""
    rng = np.random.default_rng(5)
    T = group.exp(rng.normal(size=dim) * 0.7)
    J = T.jlog()
    eps = 1e-7
    num = np.stack([(T.rplus(eps * e).log() - T.rplus(-eps * e).log()) / (2 * eps) for e in np.eye(dim)], axis=1)
    np.testing.assert_allclose(J, num, atol=1e-7)
    assert isinstance(T, MatrixLieGroup)


def test_roll_pitch_yaw_round_trip():
    """mink/lie/so3.py:116-134 (reference tests/test_lie_operations.py rpy cases)."""
    rng = np.random.default_rng(6)
    for _ in range(20):
        r, p, y = rng.uniform(-np.pi, np.pi), rng.uniform(-1.4, 1.4), rng.uniform(-np.pi, np.pi)
        R = SO3.from_rpy_radians(r, p, y)
        rpy = R.as_rpy_radians()
        assert isinstance(rpy, RollPitchYaw)
        np.testing.assert_allclose([rpy.roll, rpy.pitch, rpy.yaw], [r, p, y], atol=1e-12)
        np.testing.assert_allclose([R.compute_roll_radians(), R.compute_pitch_radians(), R.compute_yaw_radians()],
                                   [r, p, y], atol=1e-12)


def test_se3_from_mocap():
    import mink_amd as mink
    xml = """<mujoco><worldbody>
      <body name="a"><joint type="hinge" axis="0 0 1"/><geom type="sphere" size=".1" mass=".1"/></body>
      <body name="mocap" mocap="true" pos=".5 1 5" quat="1 1 0 0"><geom type="sphere" size=".1" mass=".1"/></body>
    </worldbody></mujoco>"""
    m = mink.loads_mjcf(xml)
    T = SE3.from_mocap_name(m, "mocap")
    np.testing.assert_allclose(T.translation(), [0.5, 1.0, 5.0])
    np.testing.assert_allclose(T.rotation().wxyz, np.array([1.0, 1.0, 0.0, 0.0]) / np.sqrt(2.0), atol=1e-15)
    np.testing.assert_allclose(SE3.from_mocap_id(m, 0).wxyz_xyz, T.wxyz_xyz)
    with pytest.raises(mink.InvalidMocapBody):
        SE3.from_mocap_name(m, "a")
