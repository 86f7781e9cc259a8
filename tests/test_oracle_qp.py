"""oracle/qp_gi.py: KKT certificates + scipy cross-checks (quadprog is absent;
the QP is strictly convex so its optimum is unique — SURVEY.md §7 hard part 3c)."""

import numpy as np
import pytest
from scipy.optimize import lsq_linear

from oracle import qp_gi


def _random_qp(rng, n, box=True, mg=0, damp=1e-3):
    k = int(rng.integers(1, 2 * n))
    A = rng.normal(size=(k, n))
    P = A.T @ A + damp * np.eye(n)
    q = rng.normal(size=n) * 3
    rows, rhs = [], []
    lo = hi = None
    if box:
        lo = -rng.uniform(0, 1, size=n); hi = rng.uniform(0, 1, size=n)
        rows += [np.eye(n), -np.eye(n)]; rhs += [hi, -lo]
    if mg:
        rows.append(rng.normal(size=(mg, n))); rhs.append(rng.uniform(0, 1, size=mg))
    G = np.vstack(rows) if rows else None
    h = np.hstack(rhs) if rhs else None
    return P, q, G, h, lo, hi


def test_unconstrained():
    rng = np.random.default_rng(0)
    P, q, *_ = _random_qp(rng, 12, box=False)
    x = qp_gi.solve_qp(P, q)
    np.testing.assert_allclose(P @ x + q, 0, atol=1e-10)


@pytest.mark.parametrize("seed", range(5))
def test_box_vs_bvls(seed):
    rng = np.random.default_rng(seed)
    for _ in range(20):
        n = int(rng.integers(2, 44))
        P, q, G, h, lo, hi = _random_qp(rng, n)
        x, u, A = qp_gi.solve_qp(P, q, G, h, return_info=True)
        L = np.linalg.cholesky(P)
        ref = lsq_linear(L.T, -np.linalg.solve(L, q), bounds=(lo, hi), method="bvls", tol=1e-14).x
        assert np.abs(ref - x).max() < 1e-6 * max(1, np.abs(x).max())
        lam = np.zeros(len(h)); lam[A] = u[:len(A)]
        assert qp_gi.kkt_residual(P, q, G, h, x, lam) < 1e-8 * max(1, np.abs(q).max())


def test_general_rows_kkt_and_inf_rows():
    rng = np.random.default_rng(11)
    for _ in range(40):
        n = int(rng.integers(3, 30))
        P, q, G, h, *_ = _random_qp(rng, n, mg=int(rng.integers(1, 8)))
        # mink's inactive collision rows: G=0, h=+inf (collision_avoidance_limit.py:192-199)
        G = np.vstack([G, np.zeros((2, n))]); h = np.hstack([h, [np.inf, np.inf]])
        x, u, A = qp_gi.solve_qp(P, q, G, h, return_info=True)
        lam = np.zeros(len(h)); lam[A] = u[:len(A)]
        assert qp_gi.kkt_residual(P, q, G, h, x, lam) < 1e-8 * max(1, np.abs(q).max())


def test_infeasible_and_not_pd():
    P = np.eye(2); q = np.zeros(2)
    G = np.array([[1.0, 0.0], [-1.0, 0.0]]); h = np.array([-1.0, -1.0])  # x<=-1 and x>=1
    with pytest.raises(qp_gi.Infeasible):
        qp_gi.solve_qp(P, q, G, h)
    with pytest.raises(qp_gi.NotPositiveDefinite):
        qp_gi.solve_qp(np.array([[1.0, 2.0], [2.0, 1.0]]), q)
