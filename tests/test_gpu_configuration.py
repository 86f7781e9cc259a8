"""mink.Configuration behaviours pinned by the reference's own tests (tests/test_configuration.py:20-120),
replayed on the batched device implementation (single configuration ⇒ reference shapes)."""

import numpy as np
import pytest

import mink_amd as mink
import oracle_configs as oc
from oracle import ik as oik

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ur5e():
    return mink.load_robot("ur5e")


def test_nq_nv_and_initialisation(ur5e):
    q_ref = ur5e.key_qpos[ur5e.name2id("key", "home")]
    cfg = mink.Configuration(ur5e)
    assert (cfg.nq, cfg.nv) == (ur5e.nq, ur5e.nv)                        # :20-23
    np.testing.assert_array_equal(cfg.q, ur5e.qpos0)                      # :29-33 (qpos0 is zero for the UR5e)
    cfg.update_from_keyframe("home")
    np.testing.assert_array_equal(cfg.q, q_ref)
    np.testing.assert_array_equal(mink.Configuration(ur5e, q_ref).q, q_ref)   # :25-27
    q = cfg.q
    q[0] += 1.0                                                           # .q is a copy (configuration.py:243)
    np.testing.assert_array_equal(cfg.q, q_ref)


def test_site_transform_world_frame(ur5e):
    """:36-53 — against the oracle's mj_kinematics instead of MjData."""
    rng = np.random.default_rng(12345)
    lo, hi = ur5e.jnt_range.T
    q = rng.uniform(lo, hi)
    cfg = mink.Configuration(ur5e, q)
    T = cfg.get_transform_frame_to_world("attachment_site", "site")
    o = oik.Configuration(oc.model("ur5e"), q)
    sid = ur5e.name2id("site", "attachment_site")
    np.testing.assert_allclose(T.translation(), o.data.site_xpos[sid], atol=1e-14)
    np.testing.assert_allclose(T.rotation().as_matrix(), o.data.site_xmat[sid].reshape(3, 3), atol=1e-14)


def test_invalid_frames_and_keyframes(ur5e):
    cfg = mink.Configuration(ur5e)
    with pytest.raises(mink.InvalidFrame):                                # :55-59
        cfg.get_transform_frame_to_world("invalid_name", "site")
    with pytest.raises(mink.UnsupportedFrame):                            # :61-65
        cfg.get_transform_frame_to_world("name_does_not_matter", "joint")
    with pytest.raises(mink.InvalidFrame):                                # :67-71
        cfg.get_frame_jacobian("invalid_name", "site")
    with pytest.raises(mink.UnsupportedFrame):                            # :73-77
        cfg.get_frame_jacobian("name_does_not_matter", "joint")
    with pytest.raises(mink.InvalidKeyframe):                             # :79-83
        cfg.update_from_keyframe("invalid_keyframe")


def test_inplace_integration(ur5e):
    """:85-100 (the UR5e only has hinge joints)."""
    q_ref = ur5e.key_qpos[ur5e.name2id("key", "home")]
    cfg = mink.Configuration(ur5e, q_ref)
    dt, qvel = 1e-3, np.ones(ur5e.nv)
    expected = q_ref + dt * qvel
    np.testing.assert_allclose(cfg.integrate(qvel, dt), expected, atol=1e-15)
    np.testing.assert_array_equal(cfg.q, q_ref)                           # integrate() leaves q alone
    cfg.integrate_inplace(qvel, dt)
    np.testing.assert_allclose(cfg.q, expected, atol=1e-15)


def test_check_limits(ur5e):
    """:102-110 and :112-118 (free joints are never checked)."""
    q_ref = ur5e.key_qpos[ur5e.name2id("key", "home")].copy()
    cfg = mink.Configuration(ur5e, q_ref)
    cfg.check_limits()
    q_ref[0] += 1e4
    cfg.update(q_ref)
    with pytest.raises(mink.NotWithinConfigurationLimits):
        cfg.check_limits()
    cfg.check_limits(safety_break=False)                                  # logs a warning, does not raise
    g1 = mink.load_robot("g1")
    c2 = mink.Configuration(g1)
    q = c2.q
    q[0] = 1e4                                                            # x of the free joint
    c2.update(q)
    c2.check_limits(safety_break=True)


def test_integrate_from_two_threads_sharing_one_model(ur5e):
    """Every Configuration of a FlatModel shares ONE native model, ctypes releases the GIL inside the call, and the small
    host-pointer path of mkh_integrate stages through a buffer of that model: two threads with their own Configurations must
    not see each other's q / v (round-3 advisor finding; the path is serialised by a mutex in the library)."""
    import threading
    q_ref = ur5e.key_qpos[ur5e.name2id("key", "home")]
    errs = []

    def work(sign, n=400):
        cfg = mink.Configuration(ur5e, q_ref)
        v = sign * np.arange(1, ur5e.nv + 1, dtype=np.float64)
        for k in range(n):
            dt = 1e-3 * (1 + k % 7)
            out = cfg.integrate(v, dt)
            if not np.allclose(out, q_ref + dt * v, atol=1e-14):
                errs.append((sign, k))
                return

    ts = [threading.Thread(target=work, args=(s,)) for s in (1.0, -1.0, 0.5)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:3]
