"""Limit behaviours pinned by the reference's tests (tests/test_configuration_limit.py:48-156,
tests/test_velocity_limit.py:27-63) on the device implementation."""

import numpy as np
import pytest

import mink_amd as mink

pytestmark = pytest.mark.gpu


def test_model_with_no_limit():
    """test_configuration_limit.py:48-55 / test_velocity_limit.py:39-47: nothing to constrain ⇒ (None, None)."""
    g1 = mink.load_robot("g1")
    cfg = mink.Configuration(g1)
    unlimited = mink.loads_mjcf("""
    <mujoco><compiler angle="radian"/><worldbody><body>
      <joint type="hinge" name="free_hinge"/><geom type="sphere" size=".1" mass=".1"/>
    </body></worldbody></mujoco>""")
    lim = mink.ConfigurationLimit(unlimited)
    assert len(lim.indices) == 0 and lim.projection_matrix is None
    G, h = lim.compute_qp_inequalities(mink.Configuration(unlimited), 1e-3)
    assert G is None and h is None
    v = mink.VelocityLimit(g1)
    assert len(v.indices) == 0 and v.projection_matrix is None
    G, h = v.compute_qp_inequalities(cfg, 1e-3)
    assert G is None and h is None


def test_far_from_limit(tol=1e-10):
    """test_configuration_limit.py:123-140: the limit is slack when the configuration is far from it."""
    dt = 1e-3
    m = mink.load_robot("ur5e")
    cfg = mink.Configuration(m)
    G, h = mink.ConfigurationLimit(m).compute_qp_inequalities(cfg, dt)
    names = ["shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3"]   # (vendored examples/ MJCF names)
    vel = mink.VelocityLimit(m, {n: np.pi for n in names})
    assert np.max(+G @ vel.limit * dt - h) < -tol
    assert np.max(-G @ vel.limit * dt - h) < -tol


def test_configuration_limit_repulsion(tol=1e-10):
    """test_configuration_limit.py:142-156: close to a limit the admissible step shrinks with the gain;
    `lower` / `upper` are plain attributes that may be overridden after construction."""
    dt, slack_vel = 1e-3, 5e-4
    g1 = mink.load_robot("g1")
    cfg = mink.Configuration(g1)
    cfg.update_from_keyframe("stand")
    limit = mink.ConfigurationLimit(g1, gain=0.5)
    limit.lower = cfg.integrate(-slack_vel * np.ones(cfg.nv), dt)
    limit.upper = cfg.integrate(+slack_vel * np.ones(cfg.nv), dt)
    _, h = limit.compute_qp_inequalities(cfg, dt)
    assert np.max(h) < slack_vel * dt + tol
    assert np.min(h) > -slack_vel * dt - tol
    np.testing.assert_allclose(h, 0.5 * slack_vel * dt, atol=1e-15)


def test_velocity_limit_rows():
    """test_velocity_limit.py:27-37,49-63: G = [P; −P], h = dt·[v; v] for the listed joints only."""
    m = mink.load_robot("ur5e")
    cfg = mink.Configuration(m)
    vel = mink.VelocityLimit(m, {"elbow": 2.0, "wrist_3": 0.5})
    G, h = vel.compute_qp_inequalities(cfg, 1e-2)
    assert G.shape == (4, m.nv) and h.shape == (4,)
    np.testing.assert_array_equal(vel.indices, [2, 5])
    np.testing.assert_allclose(h, [0.02, 0.005, 0.02, 0.005])
    np.testing.assert_array_equal(G[:2], np.eye(m.nv)[[2, 5]])
    np.testing.assert_array_equal(G[2:], -np.eye(m.nv)[[2, 5]])
