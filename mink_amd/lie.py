"""SO3 / SE3 value classes of the mink API (mink/lie/so3.py, se3.py, base.py), batched.

Host-side utilities for *constructing targets* (`FrameTask.set_target(SE3)`); the per-solve Lie
algebra of the hot path (log, jlog, adjoint action on Jacobians) runs on the device
(csrc/lie_dev.h).  Parameters may carry any leading batch shape: SO3.wxyz (..., 4),
SE3.wxyz_xyz (..., 7).  Tangents are (v, ω) for SE3 as in the reference (se3.py:20-21).
"""

from __future__ import annotations

import abc
from dataclasses import dataclass
from typing import NamedTuple

import numpy as np

_EPS = 1e-10  # get_epsilon(float64), mink/lie/utils.py:4-8


def _qmul(a, b):
    aw, ax, ay, az = np.moveaxis(a, -1, 0)
    bw, bx, by, bz = np.moveaxis(b, -1, 0)
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def get_epsilon(dtype) -> float:
    """mink/lie/utils.py:4-8: the small-angle threshold of the log / Jacobian branches (the device code uses the
    float64 value)."""
    return {np.dtype("float32"): 1e-5, np.dtype("float64"): 1e-10}[np.dtype(dtype)]


def skew(x):
    x = np.asarray(x, dtype=np.float64)
    z = np.zeros_like(x[..., 0])
    return np.stack([np.stack([z, -x[..., 2], x[..., 1]], -1), np.stack([x[..., 2], z, -x[..., 0]], -1),
                     np.stack([-x[..., 1], x[..., 0], z], -1)], -2)


class RollPitchYaw(NamedTuple):
    """Roll, pitch and yaw Euler angles (mink/lie/so3.py:16-21)."""
    roll: float
    pitch: float
    yaw: float


@dataclass(frozen=True)
class SO3:
    wxyz: np.ndarray

    def __post_init__(self):
        w = np.asarray(self.wxyz, dtype=np.float64)
        if w.shape[-1] != 4:
            raise ValueError(f"Expeced wxyz to be a length 4 vector but got {w.shape[-1]}.")
        object.__setattr__(self, "wxyz", w)

    def __repr__(self):
        return f"SO3(wxyz={np.round(self.wxyz, 5)})"

    def parameters(self):
        return self.wxyz

    def copy(self):
        return SO3(self.wxyz.copy())

    @classmethod
    def identity(cls):
        return SO3(np.array([1.0, 0.0, 0.0, 0.0]))

    @classmethod
    def from_matrix(cls, R):
        R = np.asarray(R, dtype=np.float64)
        m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
        q = np.empty(R.shape[:-2] + (4,))
        tr = m00 + m11 + m22
        c0 = tr > 0
        c1 = ~c0 & (m00 > m11) & (m00 > m22)
        c2 = ~c0 & ~c1 & (m11 > m22)
        c3 = ~c0 & ~c1 & ~c2
        for cond, vals in (
            (c0, (1 + tr, R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1])),
            (c1, (R[..., 2, 1] - R[..., 1, 2], 1 + m00 - m11 - m22, R[..., 0, 1] + R[..., 1, 0], R[..., 0, 2] + R[..., 2, 0])),
            (c2, (R[..., 0, 2] - R[..., 2, 0], R[..., 0, 1] + R[..., 1, 0], 1 - m00 + m11 - m22, R[..., 1, 2] + R[..., 2, 1])),
            (c3, (R[..., 1, 0] - R[..., 0, 1], R[..., 0, 2] + R[..., 2, 0], R[..., 1, 2] + R[..., 2, 1], 1 - m00 - m11 + m22)),
        ):
            v = np.stack(np.broadcast_arrays(*vals), -1)
            q = np.where(cond[..., None], v, q)
        return SO3(q / np.linalg.norm(q, axis=-1, keepdims=True))

    @classmethod
    def from_rpy_radians(cls, roll, pitch, yaw):
        x = cls.exp(np.array([roll, 0.0, 0.0])); y = cls.exp(np.array([0.0, pitch, 0.0]))
        z = cls.exp(np.array([0.0, 0.0, yaw]))
        return z @ y @ x

    @classmethod
    def from_x_radians(cls, t): return cls.exp(np.array([t, 0.0, 0.0]))
    @classmethod
    def from_y_radians(cls, t): return cls.exp(np.array([0.0, t, 0.0]))
    @classmethod
    def from_z_radians(cls, t): return cls.exp(np.array([0.0, 0.0, t]))

    @classmethod
    def sample_uniform(cls, rng=None):
        u1, u2, u3 = (np.random if rng is None else rng).uniform(0, [1.0, 2 * np.pi, 2 * np.pi])
        a, b = np.sqrt(1 - u1), np.sqrt(u1)
        return SO3(np.array([a * np.sin(u2), a * np.cos(u2), b * np.sin(u3), b * np.cos(u3)]))

    def as_matrix(self):
        w, x, y, z = np.moveaxis(self.wxyz, -1, 0)
        return np.stack([
            np.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
            np.stack([2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)], -1),
            np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z], -1)], -2)

    def inverse(self):
        return SO3(self.wxyz * np.array([1.0, -1.0, -1.0, -1.0]))

    def normalize(self):
        return SO3(self.wxyz / np.linalg.norm(self.wxyz, axis=-1, keepdims=True))

    def apply(self, target):
        target = np.asarray(target, dtype=np.float64)
        p = np.concatenate([np.zeros(target.shape[:-1] + (1,)), target], -1)
        return _qmul(_qmul(self.wxyz, p), self.inverse().wxyz)[..., 1:]

    def multiply(self, other):
        return SO3(_qmul(self.wxyz, other.wxyz))

    def __matmul__(self, other):
        if isinstance(other, np.ndarray):
            return self.apply(other)
        return self.multiply(other)

    @classmethod
    def exp(cls, tangent):
        t = np.asarray(tangent, dtype=np.float64)
        th2 = (t * t).sum(-1, keepdims=True)
        small = th2 < _EPS
        th = np.sqrt(np.where(small, 1.0, th2))
        real = np.where(small, 1 - th2 / 8 + th2 * th2 / 384, np.cos(0.5 * th))
        imag = np.where(small, 0.5 - th2 / 48 + th2 * th2 / 3840, np.sin(0.5 * th) / th)
        return SO3(np.concatenate([real, imag * t], -1))

    def log(self):
        w = self.wxyz[..., :1]
        v = self.wxyz[..., 1:]
        n2 = (v * v).sum(-1, keepdims=True)
        small = n2 < _EPS
        n = np.sqrt(np.where(small, 1.0, n2))
        ws = np.where(small, w, 1.0)
        at = np.arctan2(np.where(w < 0, -n, n), np.abs(w))
        f_t = 2.0 / ws - 2.0 / 3.0 * n2 / ws ** 3
        f_pi = np.where(w > 0, 1.0, -1.0) * np.pi / n
        f = np.where(small, f_t, np.where(np.abs(w) < _EPS, f_pi, 2.0 * at / n))
        return f * v

    def adjoint(self):
        return self.as_matrix()

    # ---- Euler angles (mink/lie/so3.py:116-134)
    def compute_roll_radians(self):
        q0, q1, q2, q3 = np.moveaxis(self.wxyz, -1, 0)
        return np.arctan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 ** 2 + q2 ** 2))

    def compute_pitch_radians(self):
        q0, q1, q2, q3 = np.moveaxis(self.wxyz, -1, 0)
        return np.arcsin(2 * (q0 * q2 - q3 * q1))

    def compute_yaw_radians(self):
        q0, q1, q2, q3 = np.moveaxis(self.wxyz, -1, 0)
        return np.arctan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 ** 2 + q3 ** 2))

    def as_rpy_radians(self):
        return RollPitchYaw(roll=self.compute_roll_radians(), pitch=self.compute_pitch_radians(),
                            yaw=self.compute_yaw_radians())

    # ---- Jacobians of exp (mink/lie/so3.py:199-226, base.py:146-156); tangents may carry batch dimensions.  The
    # Taylor branch is taken where θ < 1e-10, like the reference (θ itself, not θ²).
    @classmethod
    def ljac(cls, other):
        t = np.asarray(other, dtype=np.float64)
        th = np.sqrt((t * t).sum(-1))[..., None, None]
        small = th < _EPS
        ths = np.where(small, 1.0, th)
        t2 = th * th
        A = np.where(small, 0.5 * (1.0 - t2 / 12.0 * (1.0 - t2 / 30.0 * (1.0 - t2 / 56.0))), (1 - np.cos(ths)) / ths ** 2)
        B = np.where(small, (1.0 / 6.0) * (1.0 - t2 / 20.0 * (1.0 - t2 / 42.0 * (1.0 - t2 / 72.0))),
                     (ths - np.sin(ths)) / ths ** 3)
        S = skew(t)
        return np.eye(3) + A * S + B * (S @ S)

    @classmethod
    def ljacinv(cls, other):
        t = np.asarray(other, dtype=np.float64)
        th = np.sqrt((t * t).sum(-1))[..., None, None]
        small = th < _EPS
        ths = np.where(small, 1.0, th)
        t2 = th * th
        A = np.where(small, (1.0 / 12.0) * (1.0 + t2 / 60.0 * (1.0 + t2 / 42.0 * (1.0 + t2 / 40.0))),
                     (1.0 / ths ** 2) * (1.0 - ths * np.sin(ths) / (2.0 * (1.0 - np.cos(ths)))))
        S = skew(t)
        return np.eye(3) - 0.5 * S + A * (S @ S)

    @classmethod
    def rjac(cls, other): return cls.ljac(-np.asarray(other, dtype=np.float64))
    @classmethod
    def rjacinv(cls, other): return cls.ljacinv(-np.asarray(other, dtype=np.float64))
    def jlog(self): return SO3.rjacinv(self.log())

    def rminus(self, other): return (other.inverse() @ self).log()
    def minus(self, other): return self.rminus(other)
    def rplus(self, t): return self @ SO3.exp(t)
    def plus(self, t): return self.rplus(t)
    def lplus(self, t): return SO3.exp(t) @ self
    def lminus(self, other): return (self @ other.inverse()).log()


@dataclass(frozen=True)
class SE3:
    wxyz_xyz: np.ndarray

    def __post_init__(self):
        w = np.asarray(self.wxyz_xyz, dtype=np.float64)
        if w.shape[-1] != 7:
            raise ValueError(f"Expected wxyz_xyz to be a length 7 vector but got {w.shape[-1]}.")
        object.__setattr__(self, "wxyz_xyz", w)

    def __repr__(self):
        return f"SE3(wxyz={np.round(self.wxyz_xyz[..., :4], 5)}, xyz={np.round(self.wxyz_xyz[..., 4:], 5)})"

    def copy(self): return SE3(np.array(self.wxyz_xyz))
    def parameters(self): return self.wxyz_xyz

    @classmethod
    def identity(cls): return SE3(np.array([1.0, 0, 0, 0, 0, 0, 0]))

    @classmethod
    def from_rotation_and_translation(cls, rotation: SO3, translation):
        t = np.asarray(translation, dtype=np.float64)
        q, t = np.broadcast_arrays(rotation.wxyz, t[..., :1] * 0 + 1)[0], t
        shape = np.broadcast_shapes(rotation.wxyz.shape[:-1], t.shape[:-1])
        return SE3(np.concatenate([np.broadcast_to(rotation.wxyz, shape + (4,)),
                                   np.broadcast_to(t, shape + (3,))], -1))

    @classmethod
    def from_rotation(cls, rotation: SO3):
        return cls.from_rotation_and_translation(rotation, np.zeros(3))

    @classmethod
    def from_translation(cls, translation):
        return cls.from_rotation_and_translation(SO3.identity(), translation)

    @classmethod
    def from_matrix(cls, M):
        M = np.asarray(M, dtype=np.float64)
        return cls.from_rotation_and_translation(SO3.from_matrix(M[..., :3, :3]), M[..., :3, 3])

    @classmethod
    def sample_uniform(cls, rng=None):
        g = np.random if rng is None else rng
        return cls.from_rotation_and_translation(SO3.sample_uniform(rng), g.uniform(-1.0, 1.0, size=3))

    def rotation(self): return SO3(self.wxyz_xyz[..., :4])
    def translation(self): return self.wxyz_xyz[..., 4:]

    def as_matrix(self):
        R = self.rotation().as_matrix()
        M = np.zeros(R.shape[:-2] + (4, 4))
        M[..., :3, :3] = R
        M[..., :3, 3] = self.translation()
        M[..., 3, 3] = 1.0
        return M

    def inverse(self):
        Ri = self.rotation().inverse()
        return SE3.from_rotation_and_translation(Ri, -Ri.apply(self.translation()))

    def normalize(self):
        return SE3.from_rotation_and_translation(self.rotation().normalize(), self.translation())

    def apply(self, target):
        return self.rotation().apply(target) + self.translation()

    def multiply(self, other):
        return SE3.from_rotation_and_translation(self.rotation() @ other.rotation(),
                                                 self.rotation().apply(other.translation()) + self.translation())

    def __matmul__(self, other):
        if isinstance(other, np.ndarray):
            return self.apply(other)
        return self.multiply(other)

    @classmethod
    def exp(cls, tangent):
        t = np.asarray(tangent, dtype=np.float64)
        w = t[..., 3:]
        rot = SO3.exp(w)
        th2 = (w * w).sum(-1)[..., None, None]
        small = th2 < _EPS
        th2s = np.where(small, 1.0, th2)
        th = np.sqrt(th2s)
        S = skew(w)
        V = np.where(small, rot.as_matrix(),
                     np.eye(3) + (1 - np.cos(th)) / th2s * S + (th - np.sin(th)) / (th2s * th) * (S @ S))
        return cls.from_rotation_and_translation(rot, (V @ t[..., :3, None])[..., 0])

    def log(self):
        w = self.rotation().log()
        th2 = (w * w).sum(-1)[..., None, None]
        small = th2 < _EPS
        th2s = np.where(small, 1.0, th2)
        th = np.sqrt(th2s)
        S = skew(w)
        k = np.where(small, 1.0 / 12.0, (1.0 - th * np.cos(0.5 * th) / (2.0 * np.sin(0.5 * th))) / th2s)
        Vinv = np.eye(3) - 0.5 * S + k * (S @ S)
        return np.concatenate([(Vinv @ self.translation()[..., None])[..., 0], w], -1)

    def adjoint(self):
        R = self.rotation().as_matrix()
        A = np.zeros(R.shape[:-2] + (6, 6))
        A[..., :3, :3] = R
        A[..., :3, 3:] = skew(self.translation()) @ R
        A[..., 3:, 3:] = R
        return A

    # ---- mocap bodies (mink/lie/se3.py:77-91): the pose of a mocap body of the model (the reference reads MjData)
    @classmethod
    def from_mocap_id(cls, model, mocap_id: int):
        return cls(np.concatenate([np.asarray(model.mocap_quat[mocap_id], dtype=np.float64),
                                   np.asarray(model.mocap_pos[mocap_id], dtype=np.float64)]))

    @classmethod
    def from_mocap_name(cls, model, mocap_name: str):
        from .exceptions import InvalidMocapBody
        body = model.name2id("body", mocap_name)
        if body < 0 or int(model.body_mocapid[body]) == -1:
            raise InvalidMocapBody(mocap_name, model)
        return cls.from_mocap_id(model, int(model.body_mocapid[body]))

    # ---- Jacobians of exp (mink/lie/se3.py:196-249): blocks [[J, Q],[0, J]] with Barfoot's Q; identity where θ² < 1e-10
    @staticmethod
    def _Q(c):
        v, w = c[..., :3], c[..., 3:]
        th2 = (w * w).sum(-1)[..., None, None]
        small = th2 < _EPS
        th2s = np.where(small, 1.0, th2)
        th = np.sqrt(th2s)
        sn, cs = np.sin(th), np.cos(th)
        B = np.where(small, 1.0 / 6.0 + th2 / 120.0, (th - sn) / (th2s * th))
        C = np.where(small, -1.0 / 24.0 + th2 / 720.0, (1.0 - th2s / 2.0 - cs) / (th2s * th2s))
        D = np.where(small, -1.0 / 60.0, (2 * th - 3 * sn + th * cs) / (2 * th2s * th2s * th))
        V, W = skew(v), skew(w)
        VW = V @ W
        WV = np.swapaxes(VW, -1, -2)
        WVW = WV @ W
        VWW = VW @ W
        return 0.5 * V + B * (WV + VW + WVW) - C * (VWW - np.swapaxes(VWW, -1, -2) - 3 * WVW) + D * (WVW @ W + W @ WVW)

    @classmethod
    def _blocks(cls, other, inverse: bool):
        c = np.asarray(other, dtype=np.float64)
        w = c[..., 3:]
        small = ((w * w).sum(-1) < _EPS)[..., None, None]
        Q = cls._Q(c)
        J = SO3.ljacinv(w) if inverse else SO3.ljac(w)
        out = np.zeros(c.shape[:-1] + (6, 6))
        out[..., :3, :3] = J
        out[..., 3:, 3:] = J
        out[..., :3, 3:] = -(J @ Q @ J) if inverse else Q
        return np.where(small, np.eye(6), out)

    @classmethod
    def ljac(cls, other): return cls._blocks(other, False)
    @classmethod
    def ljacinv(cls, other): return cls._blocks(other, True)
    @classmethod
    def rjac(cls, other): return cls.ljac(-np.asarray(other, dtype=np.float64))
    @classmethod
    def rjacinv(cls, other): return cls.ljacinv(-np.asarray(other, dtype=np.float64))
    def jlog(self): return SE3.rjacinv(self.log())

    def rminus(self, other): return (other.inverse() @ self).log()
    def minus(self, other): return self.rminus(other)
    def rplus(self, t): return self @ SE3.exp(t)
    def plus(self, t): return self.rplus(t)
    def lplus(self, t): return SE3.exp(t) @ self
    def lminus(self, other): return (self @ other.inverse()).log()


__all__ = ("SE3", "SO3", "MatrixLieGroup", "RollPitchYaw", "get_epsilon", "skew")


class MatrixLieGroup(abc.ABC):
    """Interface shared by SO3 and SE3 (mink/lie/base.py:8-156).  The two groups are frozen dataclasses with
    batched parameters; they are registered as virtual subclasses, so `isinstance(x, MatrixLieGroup)` and
    annotations written against the reference keep working."""

    matrix_dim: int
    parameters_dim: int
    tangent_dim: int
    space_dim: int

    @classmethod
    @abc.abstractmethod
    def identity(cls): ...

    @classmethod
    @abc.abstractmethod
    def from_matrix(cls, matrix): ...

    @classmethod
    @abc.abstractmethod
    def exp(cls, tangent): ...

    @abc.abstractmethod
    def as_matrix(self): ...

    @abc.abstractmethod
    def parameters(self): ...

    @abc.abstractmethod
    def log(self): ...

    @abc.abstractmethod
    def inverse(self): ...

    @abc.abstractmethod
    def adjoint(self): ...


MatrixLieGroup.register(SO3)
MatrixLieGroup.register(SE3)
