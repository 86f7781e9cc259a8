"""Numpy prototype of the DEVICE QP algorithm (design aid, not product, not oracle).

min ½xᵀHx + cᵀx   s.t.  lo ≤ x ≤ hi (per-dof box),  A x ≤ b (half-spaces)

Dual active set (Goldfarb–Idnani logic) carried on a symmetric *sweep tableau* of
K = [[H, Aᵀ],[A, 0]]: index i<n is dof i, index n+k is half-space k.  A basic index
(in S) has been swept; T = SWP_S(K).  Per index we track (z, w):
  dof i      : z = x_i,   w = (Hx + c + Aᵀλ)_i
  half-space : z = λ_k,   w = A_k x − b_k
basic ⇒ w = 0; nonbasic ⇒ z sits at a bound (x at lo/hi, λ at 0).
One lane of the wavefront owns one index (one tableau column); every step below is
O(1) per lane + one cross-lane reduction, and a pivot is one rank-1 update.
"""

import math

import numpy as np

FREE, AT_LO, AT_HI, ROW_OFF, ROW_ON, ZERO = 0, 1, 2, 3, 4, 5


def sweep(T, k, reverse=False):
    d = T[k, k]
    col = T[:, k].copy()
    T -= np.outer(col, col) / d
    s = -1.0 if reverse else 1.0
    T[:, k] = s * col / d
    T[k, :] = s * col / d
    T[k, k] = -1.0 / d
    return T


def solve(H, c, lo, hi, A=None, b=None, tol=1e-12, max_iter=None, stats=None):
    n = H.shape[0]
    m = 0 if A is None else A.shape[0]
    N = n + m
    T = np.zeros((N, N))
    T[:n, :n] = H
    if m:
        T[n:, :n] = A
        T[:n, n:] = A.T
    z = np.zeros(N)
    w = np.zeros(N)
    w[:n] = c
    if m:
        w[n:] = -b
    state = np.array([ZERO] * n + [ROW_OFF] * m)
    rown = np.ones(N)
    if m:
        rown[n:] = np.sqrt((A * A).sum(1))
        rown[n:][rown[n:] == 0] = 1.0
    npiv = 0

    def step(p, alpha, p_basic):
        nonlocal z, w
        tau = T[:, p].copy()
        basic = np.isin(state, (FREE, ROW_ON))
        z[basic] -= alpha * tau[basic]
        w[~basic] += alpha * tau[~basic]
        if p_basic:
            w[p] += alpha
        else:
            z[p] += alpha

    # ---- phase 0: bring every dof into the basis (x0 = −H⁻¹c), no ratio tests
    for k in range(n):
        d = T[k, k]
        if not d > 0:
            return None, "not_pd"
        step(k, -w[k] / d, False)
        w[k] = 0.0
        sweep(T, k); state[k] = FREE; npiv += 1
    ref = np.abs(np.diag(T)).copy()      # (H⁻¹)_ii and A_k H⁻¹ A_kᵀ: pivot scale references
    ref[ref == 0] = 1.0
    if max_iter is None:
        max_iter = 20 * (N + 4)
    it = 0
    while True:
        # ---- most violated primal condition among basic dofs / inactive rows
        viol = np.zeros(N)
        for i in range(N):
            if state[i] == FREE:
                viol[i] = max(z[i] - hi[i], lo[i] - z[i])
            elif state[i] == ROW_OFF and math.isfinite(b[i - n]):
                viol[i] = w[i] / rown[i]
        p = int(np.argmax(viol))
        scale = max(1.0, abs(hi[p]) if p < n and math.isfinite(hi[p]) else 1.0)
        if viol[p] <= tol:
            break
        p_basic = state[p] == FREE
        if p_basic:
            upper = z[p] - hi[p] > lo[p] - z[p]
            beta = hi[p] if upper else lo[p]
        while True:
            it += 1
            if it > max_iter:
                return None, "max_iter"
            tau = T[:, p]
            tpp = tau[p]
            # direction sign: (A) basic dof → α so that z_p moves to β: z_p −= α τ_pp
            #                 (B) row     → α>0 raises λ_p; w_p += α τ_pp (τ_pp<0)
            if p_basic:
                sgn = -1.0 if upper else 1.0     # w_p must go ≤0 at upper, ≥0 at lower
                full = (z[p] - beta) / tpp if abs(tpp) > 1e-12 * ref[p] else math.inf
            else:
                sgn = 1.0
                full = -w[p] / tpp if -tpp > 1e-12 * ref[p] else math.inf
            # ratio test on dual feasibility, α = sgn·t, t ≥ 0
            t1, l = math.inf, -1
            for i in range(N):
                if i == p:
                    continue
                r = sgn * tau[i]
                if state[i] == ROW_ON:          # λ_i − α τ_i ≥ 0
                    if r > 0:
                        t = z[i] / r
                        if t < t1: t1, l = t, i
                elif state[i] == AT_HI:         # w_i + α τ_i ≤ 0
                    if r > 0:
                        t = -w[i] / r
                        if t < t1: t1, l = t, i
                elif state[i] == AT_LO:         # w_i + α τ_i ≥ 0
                    if r < 0:
                        t = w[i] / -r
                        if t < t1: t1, l = t, i
            t2 = abs(full)
            if not math.isfinite(min(t1, t2)):
                return None, "infeasible"
            if t2 <= t1:
                step(p, sgn * t2, p_basic)
                if p_basic:
                    z[p] = beta
                    sweep(T, p, reverse=True); state[p] = AT_HI if upper else AT_LO
                else:
                    w[p] = 0.0
                    sweep(T, p); state[p] = ROW_ON
                npiv += 1
                break
            step(p, sgn * t1, p_basic)
            if state[l] == ROW_ON:
                z[l] = 0.0
                sweep(T, l, reverse=True); state[l] = ROW_OFF
            else:
                w[l] = 0.0
                sweep(T, l); state[l] = FREE
            npiv += 1
    if stats is not None:
        stats["pivots"] = npiv
        stats["iters"] = it
    return z[:n].copy(), "ok"
