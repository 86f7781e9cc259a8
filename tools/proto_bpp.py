"""Numpy prototype of the block-principal-pivoting active-set phase (design aid, not product).

Box-constrained strictly convex QP  min ½xᵀHx + cᵀx, lo ≤ x ≤ hi  as a bound-constrained LCP with a P-matrix
(H ≻ 0).  Júdice & Pires, "A block principal pivoting algorithm for large-scale strictly monotone linear
complementarity problems" (Computers & OR 21, 1994): keep a partition F (free) / L (at lower) / U (at upper);
the infeasible indices are the free ones outside their bounds and the bound ones whose multiplier has the
wrong sign; flip ALL of them while that keeps reducing the number of infeasibilities (with a budget of p̄
non-improving block steps), otherwise flip only the infeasible index with the largest number (Murty's
single-pivot rule, finite for P-matrices).  On the sweep tableau every flip is one rank-1 pivot WITHOUT
selection reduction or ratio test: clamp (basic → bound β): α = (x−β)/d, un-sweep;  release (bound → basic):
α = −w/d, sweep.

Statistics that shaped the kernel (G1 benchmark-like problems, 150 samples): 3 block steps on average (≤ 6),
as many pivots as Goldfarb–Idnani (12.9 vs 12.0).  On the ill-conditioned, heavily saturated fixture samples
(cond(H) ≈ 2e5, 25 of 37 bounds active) the pure block method flip-flops for ~250 pivots and loses accuracy, so
the KERNEL differs from this prototype in its fallback: block steps continue only while each halves the number
of infeasibilities (≤ 3), then wrong-signed multipliers are released one by one and Goldfarb–Idnani finishes
(ik_kernel.h "phase 1a").
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))
import proto_tableau_qp as pt  # noqa: E402


def solve_bpp(H, c, lo, hi, tol=1e-12, pbar=3, stats=None):
    n = len(c)
    T = H.copy()
    x = c.copy()                  # free value: w when nonbasic, z when basic
    basic = np.zeros(n, bool)
    at_hi = np.zeros(n, bool)
    bound = np.zeros(n)           # z of a nonbasic index
    npiv = 0

    def pivot(k, beta=None, up=False):
        nonlocal T, x, npiv
        d = T[k, k]
        tau = T[:, k].copy()
        if basic[k]:
            alpha = (x[k] - beta) / d
        else:
            alpha = -x[k] / d
        upd = np.where(basic, -alpha, alpha) * tau
        x = x + upd
        if basic[k]:
            x[k] = alpha; bound[k] = beta; at_hi[k] = up
            pt.sweep(T, k, reverse=True); basic[k] = False
        else:
            x[k] = bound[k] + alpha
            pt.sweep(T, k); basic[k] = True
        npiv += 1

    for k in range(n):            # phase 0
        pivot(k)
    hmax = np.abs(np.diag(H)).max()
    tolw = 1e-16 * hmax
    best, p, outer = n + 1, pbar, 0
    while True:
        outer += 1
        over = basic & (x - hi > tol)
        under = basic & (lo - x > tol)
        wrong = (~basic) & np.where(at_hi, x > tolw, x < -tolw)
        inf = over | under | wrong
        cnt = int(inf.sum())
        if cnt == 0:
            break
        if outer > 50 * n:
            return None, "cycle"
        if cnt < best:
            best, p = cnt, pbar
            flips = np.nonzero(inf)[0]
        elif p > 0:
            p -= 1
            flips = np.nonzero(inf)[0]
        else:
            flips = np.nonzero(inf)[0][-1:]
        tgt_hi = over.copy()
        tgt_beta = np.where(over, hi, lo)
        for k in flips:
            if basic[k]:
                pivot(k, tgt_beta[k], tgt_hi[k])
            else:
                pivot(k)
    z = np.where(basic, x, bound)
    if stats is not None:
        stats["pivots"] = npiv - n
        stats["outer"] = outer
    return z, "ok"


def main():
    import oracle_configs as oc
    d = np.load(os.path.join(REPO, "tests", "golden", "ik_g1_c3.npz"))
    m = oc.model("g1")
    idx = np.array([int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] != 0 and m.jnt_limited[j]])
    n = len(idx)
    tot = np.zeros(4)
    for b in range(len(d["q"])):
        H, c, h = d["H"][b], d["c"][b], d["h"][b]
        hi = np.full(43, np.inf); lo = np.full(43, -np.inf)
        hi[idx] = np.minimum(h[:n], h[2 * n:3 * n]); lo[idx] = np.maximum(-h[n:2 * n], -h[3 * n:4 * n])
        s1, s2 = {}, {}
        x1, _ = pt.solve(H, c, lo, hi, stats=s1)
        x2, st = solve_bpp(H, c, lo, hi, stats=s2)
        assert st == "ok", st
        err = np.abs(x1 - x2).max() / max(1e-30, np.abs(x1).max())
        assert err < 1e-9, err
        tot += [s1["iters"], s1["pivots"] - 43, s2["pivots"], s2["outer"]]
        print(b, "GI iters %d pivots %d | BPP pivots %d outer %d | rel err %.1e" % (s1["iters"], s1["pivots"] - 43, s2["pivots"], s2["outer"], err))
    print("totals: GI iters %d pivots %d | BPP pivots %d outer %d" % tuple(tot))
    # random box QPs incl. tight / degenerate boxes
    rng = np.random.default_rng(3)
    worst = 0
    for trial in range(300):
        nn = int(rng.integers(2, 30))
        A = rng.normal(size=(nn + 2, nn))
        H = A.T @ A + 10 ** rng.uniform(-4, 0) * np.eye(nn)
        c = rng.normal(size=nn) * 10 ** rng.uniform(-1, 2)
        lo = -np.abs(rng.normal(size=nn)) * 10 ** rng.uniform(-3, 0); hi = np.abs(rng.normal(size=nn)) * 10 ** rng.uniform(-3, 0)
        if trial % 5 == 0:
            k = rng.integers(0, nn); lo[k] = hi[k] = 0.1      # fixed variable
        if trial % 7 == 0:
            lo[:] = -np.inf                                      # one-sided
        x1, s = pt.solve(H, c, lo, hi)
        st2 = {}
        x2, st = solve_bpp(H, c, lo, hi, stats=st2)
        assert st == "ok", (trial, st)
        worst = max(worst, np.abs(x1 - x2).max() / max(1.0, np.abs(x1).max()))
    print("random box QPs: worst abs/rel difference GI vs BPP %.2e" % worst)


if __name__ == "__main__":
    main()
