"""ORACLE (test infrastructure) — mink.lie restated as plain numpy functions.

Follows /root/reference/mink/lie/{so3,se3,base,utils}.py function by function
(citations per function).  Pinned against the reference itself: the committed
fixtures tests/golden/lie_*.npz were produced by importing the real
``mink.lie`` classes (tests/golden/make_golden.py) and tests/test_oracle_lie.py
checks every function here against them.

Quaternions are (w, x, y, z); SE3 parameters are (wxyz, xyz); se(3) tangents are
(v, ω) (mink/lie/se3.py:20-21).
"""

from __future__ import annotations

import math

import numpy as np

from . import mjmath

EPS64 = 1e-10  # mink/lie/utils.py:4-8 get_epsilon(float64)


def skew(x):
    """mink/lie/utils.py:11-21."""
    wx, wy, wz = x
    return np.array([[0.0, -wz, wy], [wz, 0.0, -wx], [-wy, wx, 0.0]])


# ----------------------------------------------------------------------- SO3
def so3_from_matrix(R):
    """mink/lie/so3.py:80-84."""
    q = np.zeros(4)
    mjmath.mju_mat2Quat(q, np.asarray(R, dtype=np.float64).ravel())
    return q


def so3_as_matrix(wxyz):
    """mink/lie/so3.py:111-114."""
    mat = np.zeros(9)
    mjmath.mju_quat2Mat(mat, wxyz)
    return mat.reshape(3, 3)


def so3_inverse(wxyz):
    """mink/lie/so3.py:136-137."""
    return wxyz * np.array([1.0, -1.0, -1.0, -1.0])


def so3_multiply(a, b):
    """mink/lie/so3.py:148-151."""
    res = np.empty(4)
    mjmath.mju_mulQuat(res, a, b)
    return res


def so3_apply(wxyz, target):
    """mink/lie/so3.py:143-146 (un-normalised sandwich product)."""
    padded = np.concatenate([np.zeros(1), target])
    return so3_multiply(so3_multiply(wxyz, padded), so3_inverse(wxyz))[1:]


def so3_exp(tangent):
    """mink/lie/so3.py:158-173."""
    theta_squared = float(tangent @ tangent)
    theta_pow_4 = theta_squared * theta_squared
    use_taylor = theta_squared < EPS64
    safe_theta = 1.0 if use_taylor else math.sqrt(theta_squared)
    safe_half_theta = 0.5 * safe_theta
    if use_taylor:
        real = 1.0 - theta_squared / 8.0 + theta_pow_4 / 384.0
        imaginary = 0.5 - theta_squared / 48.0 + theta_pow_4 / 3840.0
    else:
        real = math.cos(safe_half_theta)
        imaginary = math.sin(safe_half_theta) / safe_theta
    return np.concatenate([np.array([real]), imaginary * tangent])


def so3_log(wxyz):
    """mink/lie/so3.py:176-191."""
    w = float(wxyz[0])
    norm_sq = float(wxyz[1:] @ wxyz[1:])
    use_taylor = norm_sq < EPS64
    norm_safe = 1.0 if use_taylor else math.sqrt(norm_sq)
    w_safe = w if use_taylor else 1.0
    atan_n_over_w = math.atan2(-norm_safe if w < 0 else norm_safe, abs(w))
    if use_taylor:
        atan_factor = 2.0 / w_safe - 2.0 / 3.0 * norm_sq / w_safe**3
    else:
        if abs(w) < EPS64:
            scl = 1.0 if w > 0.0 else -1.0
            atan_factor = scl * math.pi / norm_safe
        else:
            atan_factor = 2.0 * atan_n_over_w / norm_safe
    return atan_factor * wxyz[1:]


def so3_ljacinv(other):
    """mink/lie/so3.py:214-226 (note: Taylor switch is on θ, not θ²)."""
    theta = math.sqrt(float(other @ other))
    if theta < EPS64:
        t2 = theta**2
        A = (1.0 / 12.0) * (1.0 + t2 / 60.0 * (1.0 + t2 / 42.0 * (1.0 + t2 / 40.0)))
    else:
        A = (1.0 / theta**2) * (
            1.0 - (theta * math.sin(theta) / (2.0 * (1.0 - math.cos(theta))))
        )
    S = skew(other)
    return np.eye(3) - 0.5 * S + A * (S @ S)


# ----------------------------------------------------------------------- SE3
def se3_inverse(T):
    """mink/lie/se3.py:136-141."""
    R_inv = so3_inverse(T[:4])
    return np.concatenate([R_inv, -so3_apply(R_inv, T[4:])])


def se3_multiply(a, b):
    """mink/lie/se3.py:153-157."""
    return np.concatenate([so3_multiply(a[:4], b[:4]), so3_apply(a[:4], b[4:]) + a[4:]])


def se3_exp(tangent):
    """mink/lie/se3.py:109-134."""
    rotation = so3_exp(tangent[3:])
    theta_squared = float(tangent[3:] @ tangent[3:])
    use_taylor = theta_squared < EPS64
    theta_squared_safe = 1.0 if use_taylor else theta_squared
    theta_safe = math.sqrt(theta_squared_safe)
    S = skew(tangent[3:])
    if use_taylor:
        V = so3_as_matrix(rotation)
    else:
        V = (
            np.eye(3)
            + (1.0 - math.cos(theta_safe)) / theta_squared_safe * S
            + (theta_safe - math.sin(theta_safe)) / (theta_squared_safe * theta_safe) * (S @ S)
        )
    return np.concatenate([rotation, V @ tangent[:3]])


def se3_log(T):
    """mink/lie/se3.py:159-185."""
    omega = so3_log(T[:4])
    theta_squared = float(omega @ omega)
    use_taylor = theta_squared < EPS64
    S = skew(omega)
    theta_squared_safe = 1.0 if use_taylor else theta_squared
    theta_safe = math.sqrt(theta_squared_safe)
    half_theta_safe = 0.5 * theta_safe
    SS = S @ S
    if use_taylor:
        V_inv = np.eye(3) - 0.5 * S + SS / 12.0
    else:
        V_inv = (
            np.eye(3)
            - 0.5 * S
            + (1.0 - theta_safe * math.cos(half_theta_safe) / (2.0 * math.sin(half_theta_safe)))
            / theta_squared_safe
            * SS
        )
    return np.concatenate([V_inv @ T[4:], omega])


def se3_adjoint(T):
    """mink/lie/se3.py:187-194."""
    R = so3_as_matrix(T[:4])
    return np.block([[R, skew(T[4:]) @ R], [np.zeros((3, 3)), R]])


def _getQ(c):
    """mink/lie/se3.py:222-249."""
    theta_sq = float(c[3:] @ c[3:])
    A = 0.5
    if theta_sq < EPS64:
        B = (1.0 / 6.0) + (1.0 / 120.0) * theta_sq
        C = -(1.0 / 24.0) + (1.0 / 720.0) * theta_sq
        D = -(1.0 / 60.0)
    else:
        theta = math.sqrt(theta_sq)
        sin_theta = math.sin(theta)
        cos_theta = math.cos(theta)
        B = (theta - sin_theta) / (theta_sq * theta)
        C = (1.0 - theta_sq / 2.0 - cos_theta) / (theta_sq * theta_sq)
        D = (2 * theta - 3 * sin_theta + theta * cos_theta) / (2 * theta_sq * theta_sq * theta)
    V = skew(c[:3])
    W = skew(c[3:])
    VW = V @ W
    WV = VW.T
    WVW = WV @ W
    VWW = VW @ W
    return A * V + B * (WV + VW + WVW) - C * (VWW - VWW.T - 3 * WVW) + D * (WVW @ W + W @ WVW)


def se3_ljacinv(other):
    """mink/lie/se3.py:210-218."""
    theta = other[3:]
    if float(theta @ theta) < EPS64:
        return np.eye(6)
    Q = _getQ(other)
    J_inv = so3_ljacinv(theta)
    return np.block([[J_inv, -J_inv @ Q @ J_inv], [np.zeros((3, 3)), J_inv]])


def se3_jlog(T):
    """mink/lie/base.py:150-156: jlog(T) = rjacinv(log T) = ljacinv(−log T)."""
    return se3_ljacinv(-se3_log(T))


def se3_rminus(a, b):
    """mink/lie/base.py:113-114: a ⊖ b = log(b⁻¹ a)."""
    return se3_log(se3_multiply(se3_inverse(b), a))


def se3_from_rotation_matrix_and_translation(R, t):
    """mink/lie/se3.py:46-52 with SO3.from_matrix."""
    return np.concatenate([so3_from_matrix(R), np.asarray(t, dtype=np.float64)])
