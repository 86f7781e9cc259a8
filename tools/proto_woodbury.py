"""Numpy prototype of the low-rank ("Woodbury") start of the device QP (design aid, not product).

H = Dg + JwᵀJw with Dg diagonal (damping + Σ LM terms + posture task) and Jw the n_μ weighted rows of
the frame / CoM tasks.  Instead of sweeping the nv dof indices of K = [[H, ·],[·, ·]] one by one
(nv rank-1 pivots), start from the augmented quasi-definite matrix

      K' = [[Dg, Jwᵀ],[Jw, −I]]            (μ = Jw x − r are the task residuals)

whose dof block is diagonal, so sweeping ALL dofs is closed form; only the n_μ residual indices need
rank-1 pivots (18 instead of 43 for the G1 benchmark).  After that the dof × dof block of the tableau
is −H⁻¹, z = x0 = −H⁻¹c, and the residual indices are dropped — the Goldfarb–Idnani phase runs on
exactly the tableau the direct start produces.

Symmetric scaling used on the device: rows of Jw are stored as Jh[r][k] = Jw[r][k]/√Dg_k, the lazy
scale of dof k is σ_k = 1/√Dg_k, so S = I + Jh·Jhᵀ needs one array only.
"""

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))

import proto_tableau_qp as direct  # noqa: E402


def woodbury_start(Dg, c_d, Jw, r):
    """Returns (T_dd, z, w) of the dof block after all dof + residual sweeps: T_dd = −H⁻¹, z = x0."""
    nv, nm = len(Dg), Jw.shape[0]
    N = nv + nm
    sg = np.ones(N)
    sg[:nv] = 1.0 / np.sqrt(Dg)
    Jh = Jw * sg[:nv][None, :]
    R = np.zeros((N, N))                   # raw entries; true T[i][j] = σ_i σ_j R[i][j], diagonal in D
    R[nv:, :nv] = Jh
    R[:nv, nv:] = Jh.T
    S = np.eye(nm) + Jh @ Jh.T
    R[nv:, nv:] = -S
    D = np.concatenate([-1.0 / Dg, -np.diag(S)])
    z = np.concatenate([-c_d / Dg, np.zeros(nm)])
    w = np.concatenate([np.zeros(nv), Jw @ z[:nv] - r])
    basic = np.concatenate([np.ones(nv, bool), np.zeros(nm, bool)])
    for k in range(nv, N):
        d = D[k]
        assert d < 0
        own = R[:, k].copy()
        own[k] = 0.0
        tau = sg * sg[k] * own
        tau[k] = d
        alpha = -w[k] / d
        z[basic] -= alpha * tau[basic]
        w[~basic] += alpha * tau[~basic]
        z[k] += alpha
        w[k] = 0.0
        basic[k] = True
        inv = 1.0 / d
        ck = sg * sg[k] * own
        g = sg[k] * sg[k] * own * inv
        R -= np.outer(own, g)
        Dn = D - ck * inv * ck
        Dn[k] = -inv
        D = Dn
        sg[k] = sg[k] * inv
    T = (sg[:nv, None] * sg[None, :nv]) * R[:nv, :nv]
    T[np.arange(nv), np.arange(nv)] = D[:nv]
    return T, z[:nv], w[:nv]


def main():
    import oracle_configs as oc
    from oracle import ik

    d = np.load(os.path.join(REPO, "tests", "golden", "ik_g1_c3.npz"))
    worst_T = worst_x = 0.0
    for b in range(d["q"].shape[0]):
        m, tasks, limits, dt, damping = oc.g1_c3(d["frame_targets"][b], d["posture_target"])
        cfg = ik.Configuration(m, d["q"][b])
        nv = m.nv
        Dg = np.full(nv, damping)
        c_d = np.zeros(nv)
        rows, rhs = [], []
        for t in tasks:
            e, J = ik.task_error_jacobian(cfg, t)
            W = np.asarray(t.cost, float)
            we = W * (-t.gain * e)
            Dg += t.lm_damping * (we @ we)
            Jw = W[:, None] * J
            if isinstance(t, ik.PostureTaskSpec):
                Dg += np.diag(Jw.T @ Jw)
                c_d += -we @ Jw
            else:
                keep = W != 0
                rows.append(Jw[keep])
                rhs.append(we[keep])
        Jw = np.vstack(rows)
        r = np.hstack(rhs)
        H = np.diag(Dg) + Jw.T @ Jw
        c = c_d - Jw.T @ r
        assert np.abs(H - d["H"][b]).max() < 1e-9 * np.abs(H).max()
        T, x0, w0 = woodbury_start(Dg, c_d, Jw, r)
        Hinv = np.linalg.inv(H)
        x_ref = -Hinv @ c
        worst_T = max(worst_T, np.abs(T + Hinv).max() / np.abs(Hinv).max())
        worst_x = max(worst_x, np.abs(x0 - x_ref).max() / max(1.0, np.abs(x_ref).max()))
    print(f"n_mu={Jw.shape[0]}  max rel |T + H^-1| = {worst_T:.2e}   max rel |x0 - ref| = {worst_x:.2e}"
          f"   min Dg/max Jw^2 = {Dg.min() / (Jw * Jw).max():.1e}")


if __name__ == "__main__":
    main()
