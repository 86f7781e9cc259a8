"""The device SO3/SE3 code (mink_amd/csrc/lie_dev.h — what the frame-task lanes run) held DIRECTLY against the
known-answer vectors recorded from the real mink.lie (tests/golden/lie.npz, make_golden.py:49-95), through the C ABI
(mkh_lie_eval), including the special quaternions with w < 0 and w ≈ 0 (so3.py:176-191 branch coverage)."""

import os

import numpy as np
import pytest

import oracle_configs as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lie():
    from mink_amd import _native
    assert _native.lib().mkh_device_count() >= 1
    return _native.lie_eval


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(oc.GOLDEN, "lie.npz"))


def test_se3_log_and_rminus(lie, g):
    np.testing.assert_allclose(lie("se3_log", g["se3_params"]), g["se3_log"], rtol=0, atol=1e-13)
    # exp(ξ) with |ω| from 1e-7 to 2: relative to the reference's own value
    out = lie("se3_log", g["se3_exp"])
    np.testing.assert_allclose(out, g["se3_log_of_exp"], rtol=0, atol=1e-12)
    T = g["se3_params"]
    np.testing.assert_allclose(lie("se3_rminus", T, np.roll(T, -1, axis=0)), g["se3_rminus"], rtol=0, atol=1e-12)


def test_se3_group_operations(lie, g):
    T = g["se3_params"]
    np.testing.assert_allclose(lie("se3_inverse", T), g["se3_inverse"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(lie("se3_multiply", T, np.roll(T, -1, axis=0)), g["se3_multiply"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(lie("se3_apply", T, g["points"]), g["se3_apply"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(lie("so3_matrix", T[:, :4]), g["so3_as_matrix"], rtol=0, atol=1e-15)


def test_jlog_and_ljacinv(lie, g):
    np.testing.assert_allclose(lie("se3_jlog", g["se3_params"]), g["se3_jlog"], rtol=0, atol=1e-12)
    # The reference's closed forms cancel catastrophically for small θ (its own float64 output is off by
    # ≈5e-17·(1+|v|)/θ², 4e-7 at θ = 1.2e-5 — SURVEY §7 hard part 3), so a direct comparison cannot be tighter than
    # that noise.  lie_exact.npz holds the 60-digit values (tests/golden/make_lie_exact.py): the device must be at
    # least as accurate as the reference is (factor 4, floor 2e-12), sample by sample.
    ex = np.load(os.path.join(oc.GOLDEN, "lie_exact.npz"))
    th = np.linalg.norm(g["tangent"][:, 3:], axis=1)
    for op, arg, ref, exact in (("se3_ljacinv", g["tangent"], g["se3_ljacinv"], ex["se3_ljacinv"]),
                                ("se3_jlog", g["se3_exp"], g["se3_jlog_of_exp"], ex["se3_jlog_of_exp"])):
        out = lie(op, arg)
        err_dev = np.abs(out - exact).max(axis=(1, 2))
        err_ref = np.abs(ref - exact).max(axis=(1, 2))
        print(op, "max err vs exact: device %.2e, reference %.2e" % (err_dev.max(), err_ref.max()))
        bad = err_dev > 4.0 * err_ref + 2e-12
        assert not bad.any(), (op, err_dev[bad], err_ref[bad], th[bad])
        # and, for rotations that are not small, plain agreement with the reference
        big = th > 1e-2
        np.testing.assert_allclose(out[big], ref[big], rtol=0, atol=1e-11)
    small = th < 1e-5                                   # θ² < 1e-10: both sides return the identity
    assert small.any()
    np.testing.assert_array_equal(lie("se3_ljacinv", g["tangent"][small]), g["se3_ljacinv"][small])


def test_so3_log_special_quaternions(lie, g):
    np.testing.assert_allclose(lie("so3_log", g["se3_params"][:, :4]), g["so3_log"], rtol=0, atol=1e-14)
    out = lie("so3_log", g["so3_special"])
    print("so3_special:", np.abs(out - g["so3_special_log"]).max(axis=1))
    np.testing.assert_allclose(out, g["so3_special_log"], rtol=0, atol=1e-14)
