"""ORACLE (test infrastructure) — Goldfarb–Idnani dual active-set QP, restated.

The reference hands its dense QP to ``qpsolvers.solve_problem(problem,
solver="quadprog")`` (mink/solve_ik.py:65,101).  ``qpsolvers`` (>= 4.3.1) and
``quadprog`` are third-party packages absent from /root/reference and from the
build image, so this file restates the published algorithm quadprog implements —
D. Goldfarb, A. Idnani, "A numerically stable dual method for solving strictly
convex quadratic programs", Math. Programming 27 (1983) — with the J = L^{-T},
R (triangular) factor updates by Givens rotations, as in Turlach's ``qpgen2``
(SURVEY.md Appendix B).  **Parity unpinned against quadprog itself**; because the
QP is strictly convex its optimum is unique, so agreement is certified by KKT
residuals and by scipy cross-checks (tests/test_oracle_qp.py).

min ½ xᵀP x + qᵀx   s.t.  G x ≤ h
"""

from __future__ import annotations

import math
from typing import Optional

import numpy as np


class NotPositiveDefinite(ValueError):
    """quadprog: 'matrix G is not positive definite' (→ qpsolvers ProblemError)."""


class Infeasible(ValueError):
    """quadprog: 'constraints are inconsistent, no solution' (→ x=None, mink
    then fails its ``assert dq is not None``, mink/solve_ik.py:103)."""


def _cholesky_lower(P: np.ndarray) -> np.ndarray:
    n = P.shape[0]
    L = np.zeros_like(P)
    for j in range(n):
        s = P[j, j] - L[j, :j] @ L[j, :j]
        if not s > 0.0:
            raise NotPositiveDefinite("matrix P is not positive definite")
        L[j, j] = math.sqrt(s)
        for i in range(j + 1, n):
            L[i, j] = (P[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
    return L


def solve_qp(P, q, G: Optional[np.ndarray] = None, h: Optional[np.ndarray] = None,
             tol: float = 1e-12, max_iter: Optional[int] = None, return_info: bool = False):
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    n = P.shape[0]
    L = _cholesky_lower(P)
    # J = L^{-T}  (so that Jᵀ P J = I);  x0 = -P^{-1} q
    Linv = np.zeros((n, n))
    for j in range(n):
        e = np.zeros(n); e[j] = 1.0
        for i in range(j, n):
            e[i] = (e[i] - L[i, j:i] @ e[j:i]) / L[i, i]
        Linv[:, j] = e
    J = Linv.T.copy()
    x = -(J @ (J.T @ q))
    if G is None or len(G) == 0:
        return (x, np.zeros(0), []) if return_info else x
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    m = G.shape[0]
    Nn = -G                      # constraint normals n_i (n_iᵀx ≥ b_i)
    b = -h
    nrm = np.sqrt((Nn * Nn).sum(axis=1))
    R = np.zeros((n, n))
    A: list = []                 # active set (constraint indices), ordered
    u = np.zeros(0)              # multipliers of A
    nact = 0
    if max_iter is None:
        max_iter = 50 * (n + m)
    it = 0
    while True:
        # ---- step 1: most violated constraint (normalised slack)
        s = Nn @ x - b           # ≥ 0 when satisfied
        best, p = 0.0, -1
        for i in range(m):
            if i in A or not math.isfinite(b[i]) or nrm[i] == 0.0:
                continue
            v = s[i] / nrm[i]
            if v < -tol * max(1.0, abs(b[i]) / nrm[i]) and v < best:
                best, p = v, i
        if p < 0:
            break
        npv = Nn[p]
        u = np.append(u, 0.0)    # u⁺
        while True:
            it += 1
            if it > max_iter:
                raise Infeasible("iteration limit reached")
            # ---- step 2a: directions
            d = J.T @ npv
            z = J[:, nact:] @ d[nact:]
            r = np.zeros(nact)
            for i in range(nact - 1, -1, -1):      # back-substitution R r = d1
                r[i] = (d[i] - R[i, i + 1:nact] @ r[i + 1:]) / R[i, i]
            # ---- step 2b: step lengths
            t1, l = math.inf, -1
            for k in range(nact):
                if r[k] > 0.0:
                    tk = u[k] / r[k]
                    if tk < t1:
                        t1, l = tk, k
            # zᵀn⁺ = ‖d₂‖²; "|z| = 0" ⇔ n⁺ lies in the span of the active normals
            dd2 = float(d[nact:] @ d[nact:])
            if dd2 > 1e-24 * float(d @ d):
                t2 = -float(npv @ x - b[p]) / dd2
            else:
                t2 = math.inf
            t = min(t1, t2)
            if not math.isfinite(t):
                raise Infeasible("constraints are inconsistent, no solution")
            if not math.isfinite(t2):
                # step in dual space only
                u[:nact] -= t * r
                u[nact] += t
                R, J, A, u, nact = _drop(R, J, A, u, nact, l)
                continue
            x = x + t * z
            u[:nact] -= t * r
            u[nact] += t
            if t2 <= t1:
                # full step: add p to the active set
                R, J = _add(R, J, d, nact)
                A.append(p)
                nact += 1
                break
            # partial step: drop blocking constraint l, retry p
            R, J, A, u, nact = _drop(R, J, A, u, nact, l)
    if return_info:
        return x, u, A
    return x


def _givens(a: float, b: float):
    if b == 0.0:
        return 1.0, 0.0, a
    h = math.hypot(a, b)
    return a / h, b / h, h


def _add(R: np.ndarray, J: np.ndarray, d: np.ndarray, nact: int):
    """Append n⁺: rotate d[nact:] onto its first component, same rotations on J."""
    n = J.shape[0]
    d = d.copy()
    for k in range(n - 1, nact, -1):
        c, s, hh = _givens(d[k - 1], d[k])
        if s == 0.0:
            continue
        d[k - 1], d[k] = hh, 0.0
        a, bcol = J[:, k - 1].copy(), J[:, k].copy()
        J[:, k - 1] = c * a + s * bcol
        J[:, k] = -s * a + c * bcol
    R[: nact + 1, nact] = d[: nact + 1]
    return R, J


def _drop(R: np.ndarray, J: np.ndarray, A: list, u: np.ndarray, nact: int, l: int):
    """Remove active constraint at position l; restore triangular R via Givens."""
    for k in range(l, nact - 1):
        R[:, k] = R[:, k + 1]
        u[k] = u[k + 1]
    u[nact - 1] = u[nact]
    u = u[:nact].copy()          # still carries u⁺ in the last slot
    R[:, nact - 1] = 0.0
    A.pop(l)
    nact -= 1
    for k in range(l, nact):
        c, s, hh = _givens(R[k, k], R[k + 1, k])
        if s == 0.0:
            continue
        rk, rk1 = R[k, k:nact].copy(), R[k + 1, k:nact].copy()
        R[k, k:nact] = c * rk + s * rk1
        R[k + 1, k:nact] = -s * rk + c * rk1
        R[k + 1, k] = 0.0
        a, bcol = J[:, k].copy(), J[:, k + 1].copy()
        J[:, k] = c * a + s * bcol
        J[:, k + 1] = -s * a + c * bcol
    return R, J, A, u, nact


def kkt_residual(P, q, G, h, x, lam=None):
    """max(|stationarity|, primal violation, |complementarity|, dual infeas.)."""
    P = np.asarray(P); q = np.asarray(q)
    g = P @ x + q
    if G is None or len(G) == 0:
        return float(np.abs(g).max())
    G = np.asarray(G); h = np.asarray(h)
    fin = np.isfinite(h)
    slack = np.where(fin, h - G @ x, np.inf)
    if lam is None:
        # least-squares multipliers on the (near-)active rows
        act = fin & (slack < 1e-9 * np.maximum(1.0, np.abs(h)))
        lam = np.zeros(len(h))
        if act.any():
            sol, *_ = np.linalg.lstsq(G[act].T, -g, rcond=None)
            lam[act] = sol
    stat = np.abs(g + G.T @ lam).max()
    prim = max(0.0, float(-(slack[fin]).min())) if fin.any() else 0.0
    dual = max(0.0, float(-lam.min()))
    comp = float(np.abs(lam[fin] * slack[fin]).max()) if fin.any() else 0.0
    return float(max(stat, prim, dual, comp))
