"""Exact (60-digit, mpmath) values of SE3.ljacinv on the tangents of lie.npz, rounded to float64.

The reference's closed forms (mink/lie/se3.py:210-249, so3.py:214-226) cancel catastrophically for small θ: its own
float64 output is off by ≈5e-17·(1+|v|)/θ² (4e-7 at θ = 1.2e-5).  A direct device-vs-reference comparison therefore
cannot be tighter than that noise; with the exact values the device is held to "at least as accurate as the reference".

    python tests/golden/make_lie_exact.py        # writes lie_exact.npz (needs mpmath; no reference import)
"""
import os

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 60


def skew(v):
    return mp.matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def ljacinv_exact(x):
    v = [mp.mpf(float(a)) for a in x[:3]]
    w = [mp.mpf(float(a)) for a in x[3:]]
    thsq = sum(a * a for a in w)
    out = np.eye(6)
    if thsq < mp.mpf("1e-10"):
        return out                                   # the reference's identity branch (se3.py:213-214)
    th = mp.sqrt(thsq)
    W, V = skew(w), skew(v)
    Jinv = mp.eye(3) - W / 2 + (1 / thsq - (1 + mp.cos(th)) / (2 * th * mp.sin(th))) * (W * W)
    A = mp.mpf(1) / 2
    B = (th - mp.sin(th)) / (thsq * th)
    C = (1 - thsq / 2 - mp.cos(th)) / (thsq * thsq)
    D = (2 * th - 3 * mp.sin(th) + th * mp.cos(th)) / (2 * thsq * thsq * th)
    VW = V * W
    WV = VW.T
    WVW = WV * W
    VWW = VW * W
    Q = A * V + B * (WV + VW + WVW) - C * (VWW - VWW.T - 3 * WVW) + D * (WVW * W + W * WVW)
    Bk = -(Jinv * Q * Jinv)
    out = np.zeros((6, 6))
    for i in range(3):
        for j in range(3):
            out[i, j] = out[i + 3, j + 3] = float(Jinv[i, j])
            out[i, j + 3] = float(Bk[i, j])
    return out


def main():
    g = np.load(os.path.join(HERE, "lie.npz"))
    t = g["tangent"]
    np.savez_compressed(os.path.join(HERE, "lie_exact.npz"),
                        se3_ljacinv=np.array([ljacinv_exact(x) for x in t]),
                        se3_jlog_of_exp=np.array([ljacinv_exact(-x) for x in t]))


if __name__ == "__main__":
    main()
