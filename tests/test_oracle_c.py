"""oracle/c (plain-C restatement) pinned against (a) the fixtures recorded from the real mink Python
(tests/golden/ik_*.npz) and (b) the numpy restatement oracle/*.py.  CPU only."""

import os

import numpy as np
import pytest

import oracle_configs as oc
from oracle import cport, ik, qp_gi


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"ik_{name}.npz"))


@pytest.mark.parametrize("name,cfgfn,extra", [("ur5e_c2", oc.ur5e_c2, None), ("g1_c3", oc.g1_c3, None),
                                              ("g1_full", oc.g1_full, "com_target")])
def test_c_oracle_vs_real_mink_fixtures(golden_dir, name, cfgfn, extra):
    d = _load(golden_dir, name)
    for i in range(len(d["q"])):
        args = [d["frame_targets"][i], d["posture_target"]]
        if extra is not None:
            args.append(d[extra][i])
        m, tasks, limits, dt, damping = cfgfn(*args)
        prob = cport.CProblem(m, tasks, limits)
        v, (H, c) = prob.solve(d["q"][i], dt, damping, return_problem=True)
        scale = max(1.0, np.abs(d["H"][i]).max())
        np.testing.assert_allclose(H, d["H"][i], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(c, d["c"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["c"][i]).max()))
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][i]).max()))
        # and against the numpy restatement (same operation order): much tighter
        v_np = ik.solve_ik(m, d["q"][i], tasks, dt, damping, limits)
        np.testing.assert_allclose(v, v_np, rtol=0, atol=1e-11 * max(1.0, np.abs(v_np).max()))


def test_c_oracle_ur5e_c1_default_limits(golden_dir):
    """limits=None ⇒ a fresh ConfigurationLimit (solve_ik.py:28-29); every step of the recorded trajectory."""
    d = _load(golden_dir, "ur5e_c1")
    m, tasks, limits, dt, damping = oc.ur5e_c1([d["frame_target"]], d["posture_target"])
    assert limits is None
    prob = cport.CProblem(m, tasks, limits)
    for k in range(len(d["q"])):
        v = prob.solve(d["q"][k], dt, damping)
        np.testing.assert_allclose(v, d["v"][k], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][k]).max()))


def test_c_qp_vs_numpy_gi():
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(2, 20)); m = int(rng.integers(0, 30))
        A = rng.normal(size=(n + 2, n))
        P = A.T @ A + 0.1 * np.eye(n)
        q = rng.normal(size=n)
        G = rng.normal(size=(m, n)) if m else None
        h = (np.abs(rng.normal(size=m)) + 0.05) if m else None     # x = 0 feasible
        x_np = qp_gi.solve_qp(P, q, G, h)
        x_c = cport.solve_qp(P, q, G, h)
        np.testing.assert_allclose(x_c, x_np, rtol=0, atol=1e-10 * max(1.0, np.abs(x_np).max()))
        assert qp_gi.kkt_residual(P, q, G, h, x_c) < 1e-8 * max(1.0, np.abs(q).max())
    with pytest.raises(qp_gi.NotPositiveDefinite):
        cport.solve_qp(np.diag([1.0, -1.0]), np.zeros(2))
    with pytest.raises(qp_gi.Infeasible):
        cport.solve_qp(np.eye(2), np.zeros(2), np.array([[1.0, 0.0], [-1.0, 0.0]]), np.array([-1.0, -1.0]))


def test_c_batch_equals_single_and_threads(golden_dir):
    d = _load(golden_dir, "g1_c3")
    m, tasks, limits, dt, damping = oc.g1_c3(d["frame_targets"][0], d["posture_target"])
    prob = cport.CProblem(m, tasks, limits)
    v1, st1 = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping, nthreads=1)
    v2, st2 = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping, nthreads=2)
    assert (st1 == 0).all() and (st2 == 0).all()
    np.testing.assert_array_equal(v1, v2)
    np.testing.assert_allclose(v1, d["v"], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"]).max()))


def test_c_oracle_vs_2048_real_mink_instances(golden_dir):
    """The BASELINE-sized real-mink slice (tests/golden/make_golden_big.py) pins the checker itself at that size."""
    d = _load(golden_dir, "g1_c3_big")
    m, tasks, limits, dt, damping = oc.g1_c3(d["frame_targets"][0], d["posture_target"])
    prob = cport.CProblem(m, tasks, limits)
    v, st = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping)
    assert (st == 0).all()
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    err = np.abs(v - d["v"]) / vs
    main = np.ones(len(v), bool); main[7::8] = False
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5, (err[main].max(), err[~main].max())
