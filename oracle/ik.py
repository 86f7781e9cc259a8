"""ORACLE (test infrastructure) — mink's ``solve_ik`` restated on CPU.

Restates, on top of :mod:`oracle.mjmath` (MuJoCo arithmetic) and
:mod:`oracle.qp_gi` (quadprog's Goldfarb–Idnani), the reference hot path:

* mink/configuration.py:53-64,77-110,112-185   (FK update, limit check, frame pose, body Jacobian)
* mink/tasks/task.py:105-138                    (H, c assembly)
* mink/tasks/frame_task.py:95-146, posture_task.py:87-142, com_task.py:71-97
* mink/tasks/relative_frame_task.py:106-142, damping_task.py:11-20
* mink/limits/configuration_limit.py:69-124, velocity_limit.py:71-101,
  collision_avoidance_limit.py:187-229
* mink/solve_ik.py:13-105

The mink layer of this restatement is pinned against the real mink Python code:
tests/golden/make_golden.py runs /root/reference's ``mink.solve_ik`` (with the
absent third-party ``mujoco``/``qpsolvers`` wheels replaced by oracle/stubs, i.e.
by this package's own restatement of their arithmetic) and commits (q, targets) →
(H, c, G, h, v) fixtures that tests/test_oracle_ik.py replays.  The MuJoCo /
quadprog layers remain unpinned against the real wheels (see their headers).

Problems are described by plain spec objects so the oracle shares no code with
the product package.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import lie, mjmath, qp_gi

JNT_FREE = 0
_QW = {0: 7, 1: 4, 2: 1, 3: 1}
_DW = {0: 6, 1: 3, 2: 1, 3: 1}


# ------------------------------------------------------------------- specs
@dataclass
class FrameTaskSpec:
    frame_id: int
    frame_type: str                 # "body" | "geom" | "site"
    cost: np.ndarray                # (6,) [position ×3, orientation ×3]
    target: np.ndarray              # (7,) wxyz_xyz, transform_target_to_world
    gain: float = 1.0
    lm_damping: float = 0.0


@dataclass
class RelativeFrameTaskSpec:
    frame_id: int
    frame_type: str
    root_id: int
    root_type: str
    cost: np.ndarray
    target: np.ndarray              # transform_target_to_root
    gain: float = 1.0
    lm_damping: float = 0.0


@dataclass
class PostureTaskSpec:
    cost: np.ndarray                # (nv,)
    target_q: np.ndarray            # (nq,)
    gain: float = 1.0
    lm_damping: float = 0.0


@dataclass
class ComTaskSpec:
    cost: np.ndarray                # (3,)
    target: np.ndarray              # (3,)
    gain: float = 1.0
    lm_damping: float = 0.0


@dataclass
class ConfigurationLimitSpec:
    gain: float = 0.95
    min_distance_from_limits: float = 0.0


@dataclass
class VelocityLimitSpec:
    indices: np.ndarray             # dof ids
    limit: np.ndarray               # max |velocity| per entry


@dataclass
class CollisionAvoidanceLimitSpec:
    geom_id_pairs: Sequence[Tuple[int, int]]
    gain: float = 0.85
    minimum_distance_from_collisions: float = 0.005
    collision_detection_distance: float = 0.01
    bound_relaxation: float = 0.0


@dataclass
class DenseTaskSpec:
    """A caller-defined mink.Task subclass reduced to what the reference's base class consumes
    (mink/tasks/task.py:81-138): the values of compute_error / compute_jacobian at this configuration."""
    e: np.ndarray                   # (k,)
    J: np.ndarray                   # (k, nv)
    cost: np.ndarray                # (k,)
    gain: float = 1.0
    lm_damping: float = 0.0


@dataclass
class DenseLimitSpec:
    """A caller-defined mink.Limit subclass: the (G, h) its compute_qp_inequalities returns
    (mink/limits/limit.py:34-57); rows with h = +inf are inactive."""
    G: np.ndarray                   # (m, nv)
    h: np.ndarray                   # (m,)


# ----------------------------------------------------------- configuration
class Configuration:
    """mink/configuration.py:21-64."""

    def __init__(self, model, q=None):
        self.model = model
        self.data = mjmath.Data(model)
        self.update(q)

    def update(self, q=None):
        if q is not None:
            self.data.qpos = np.array(q, dtype=np.float64)
        mjmath.mj_kinematics(self.model, self.data)
        mjmath.mj_comPos(self.model, self.data)

    @property
    def q(self):
        return self.data.qpos.copy()

    def limit_violations(self, tol: float = 1e-6) -> List[int]:
        """Joint ids that mink/configuration.py:86-110 would raise/warn about."""
        m = self.model
        out = []
        for jnt in range(m.njnt):
            if m.jnt_type[jnt] == JNT_FREE or not m.jnt_limited[jnt]:
                continue
            qval = self.data.qpos[m.jnt_qposadr[jnt]]
            if qval < m.jnt_range[jnt, 0] - tol or qval > m.jnt_range[jnt, 1] + tol:
                out.append(jnt)
        return out

    def _frame(self, frame_id: int, frame_type: str):
        d = self.data
        if frame_type == "body":
            return d.xpos[frame_id], d.xmat[frame_id]
        if frame_type == "geom":
            return d.geom_xpos[frame_id], d.geom_xmat[frame_id]
        if frame_type == "site":
            return d.site_xpos[frame_id], d.site_xmat[frame_id]
        raise ValueError(f"{frame_type} is not supported.")

    def get_transform_frame_to_world(self, frame_id: int, frame_type: str) -> np.ndarray:
        """mink/configuration.py:157-185 → wxyz_xyz."""
        xpos, xmat = self._frame(frame_id, frame_type)
        return lie.se3_from_rotation_matrix_and_translation(xmat.reshape(3, 3), xpos)

    def get_frame_jacobian(self, frame_id: int, frame_type: str) -> np.ndarray:
        """mink/configuration.py:112-155: body-frame Jacobian (6, nv)."""
        m, d = self.model, self.data
        jac = np.empty((6, m.nv))
        if frame_type == "body":
            mjmath.mj_jacBody(m, d, jac[:3], jac[3:], frame_id)
        elif frame_type == "geom":
            mjmath.mj_jacGeom(m, d, jac[:3], jac[3:], frame_id)
        elif frame_type == "site":
            mjmath.mj_jacSite(m, d, jac[:3], jac[3:], frame_id)
        else:
            raise ValueError(f"{frame_type} is not supported.")
        _, xmat = self._frame(frame_id, frame_type)
        R_wf = lie.so3_from_matrix(xmat.reshape(3, 3))
        T = np.concatenate([lie.so3_inverse(R_wf), np.zeros(3)])
        return lie.se3_adjoint(T) @ jac

    def get_transform(self, src_id, src_type, dst_id, dst_type) -> np.ndarray:
        """mink/configuration.py:187-212: pose of source in dest."""
        a = self.get_transform_frame_to_world(src_id, src_type)
        b = self.get_transform_frame_to_world(dst_id, dst_type)
        return lie.se3_multiply(lie.se3_inverse(b), a)

    def integrate(self, velocity, dt) -> np.ndarray:
        """mink/configuration.py:214-226."""
        q = self.data.qpos.copy()
        mjmath.mj_integratePos(self.model, q, np.asarray(velocity, dtype=np.float64), dt)
        return q


# -------------------------------------------------------------------- tasks
def task_error_jacobian(cfg: Configuration, task) -> Tuple[np.ndarray, np.ndarray]:
    m = cfg.model
    if isinstance(task, FrameTaskSpec):
        # mink/tasks/frame_task.py:95-146
        T_frame = cfg.get_transform_frame_to_world(task.frame_id, task.frame_type)
        e = lie.se3_rminus(task.target, T_frame)          # target.minus(frame)
        jac = cfg.get_frame_jacobian(task.frame_id, task.frame_type)
        T_tb = lie.se3_multiply(lie.se3_inverse(task.target), T_frame)
        return e, -lie.se3_jlog(T_tb) @ jac
    if isinstance(task, RelativeFrameTaskSpec):
        # mink/tasks/relative_frame_task.py:106-142
        T_fr = cfg.get_transform(task.frame_id, task.frame_type, task.root_id, task.root_type)
        e = lie.se3_rminus(T_fr, task.target)             # frame_to_root.rminus(target)
        jac_frame = cfg.get_frame_jacobian(task.frame_id, task.frame_type)
        jac_root = cfg.get_frame_jacobian(task.root_id, task.root_type)
        T_tf = lie.se3_multiply(lie.se3_inverse(task.target), T_fr)
        T_fr_inv = lie.se3_inverse(T_fr)
        return e, lie.se3_jlog(T_tf) @ (jac_frame - lie.se3_adjoint(T_fr_inv) @ jac_root)
    if isinstance(task, PostureTaskSpec):
        # mink/tasks/posture_task.py:87-142
        qvel = np.empty(m.nv)
        mjmath.mj_differentiatePos(m, qvel, 1.0, cfg.q, task.target_q)
        jac = -np.eye(m.nv)
        for j in range(m.njnt):
            if m.jnt_type[j] == JNT_FREE:
                va = int(m.jnt_dofadr[j])
                qvel[va:va + 6] = 0.0
                jac[:, va:va + 6] = 0.0
        return qvel, jac
    if isinstance(task, DenseTaskSpec):
        return np.asarray(task.e, dtype=np.float64), np.asarray(task.J, dtype=np.float64)
    if isinstance(task, ComTaskSpec):
        # mink/tasks/com_task.py:71-97 (subtree of body 1)
        e = cfg.data.subtree_com[1] - task.target
        jac = np.empty((3, m.nv))
        mjmath.mj_jacSubtreeCom(m, cfg.data, jac, 1)
        return e, jac
    raise TypeError(type(task))


def task_qp_objective(cfg: Configuration, task) -> Tuple[np.ndarray, np.ndarray]:
    """mink/tasks/task.py:105-138."""
    e, jacobian = task_error_jacobian(cfg, task)
    minus_gain_error = -task.gain * e
    weight = np.diag(np.asarray(task.cost, dtype=np.float64))
    weighted_jacobian = weight @ jacobian
    weighted_error = weight @ minus_gain_error
    mu = task.lm_damping * weighted_error @ weighted_error
    H = weighted_jacobian.T @ weighted_jacobian + mu * np.eye(cfg.model.nv)
    c = -weighted_error.T @ weighted_jacobian
    return H, c


# ------------------------------------------------------------------- limits
def configuration_limit_arrays(m, spec: ConfigurationLimitSpec):
    """mink/limits/configuration_limit.py:41-67 (constructor)."""
    index_list: List[int] = []
    lower = np.full(m.nq, -mjmath.mjMAXVAL)
    upper = np.full(m.nq, mjmath.mjMAXVAL)
    for jnt in range(m.njnt):
        jt = int(m.jnt_type[jnt])
        if jt == JNT_FREE or not m.jnt_limited[jnt]:
            continue
        padr = int(m.jnt_qposadr[jnt])
        lower[padr:padr + _QW[jt]] = m.jnt_range[jnt, 0] + spec.min_distance_from_limits
        upper[padr:padr + _QW[jt]] = m.jnt_range[jnt, 1] - spec.min_distance_from_limits
        va = int(m.jnt_dofadr[jnt])
        index_list.extend(range(va, va + _DW[jt]))
    return np.array(index_list, dtype=np.int64), lower, upper


def limit_inequalities(cfg: Configuration, spec, dt: float):
    """→ (G, h) or (None, None) when the limit is inactive."""
    m = cfg.model
    if isinstance(spec, ConfigurationLimitSpec):
        # mink/limits/configuration_limit.py:69-124
        idx, lower, upper = configuration_limit_arrays(m, spec)
        if len(idx) == 0:
            return None, None
        delta_q_max = np.zeros(m.nv)
        mjmath.mj_differentiatePos(m, delta_q_max, 1.0, cfg.q, upper)
        delta_q_min = np.zeros(m.nv)
        mjmath.mj_differentiatePos(m, delta_q_min, 1.0, lower, cfg.q)
        Pm = np.eye(m.nv)[idx]
        G = np.vstack([Pm, -Pm])
        h = np.hstack([spec.gain * delta_q_max[idx], spec.gain * delta_q_min[idx]])
        return G, h
    if isinstance(spec, DenseLimitSpec):
        return np.asarray(spec.G, dtype=np.float64), np.asarray(spec.h, dtype=np.float64)
    if isinstance(spec, VelocityLimitSpec):
        # mink/limits/velocity_limit.py:71-101
        if len(spec.indices) == 0:
            return None, None
        Pm = np.eye(m.nv)[np.asarray(spec.indices, dtype=np.int64)]
        lim = np.asarray(spec.limit, dtype=np.float64)
        return np.vstack([Pm, -Pm]), np.hstack([dt * lim, dt * lim])
    if isinstance(spec, CollisionAvoidanceLimitSpec):
        # mink/limits/collision_avoidance_limit.py:187-210
        npair = len(spec.geom_id_pairs)
        upper_bound = np.full(npair, np.inf)
        coeff = np.zeros((npair, m.nv))
        for k, (g1, g2) in enumerate(spec.geom_id_pairs):
            fromto = np.empty(6)
            dist = mjmath.mj_geomDistance(m, cfg.data, g1, g2, spec.collision_detection_distance, fromto)
            if dist == spec.collision_detection_distance:     # Contact.inactive
                continue
            if dist > spec.minimum_distance_from_collisions:
                d = dist - spec.minimum_distance_from_collisions
                upper_bound[k] = (spec.gain * d / dt) + spec.bound_relaxation
            else:
                upper_bound[k] = spec.bound_relaxation
            # compute_contact_normal_jacobian (:59-72)
            normal = fromto[3:] - fromto[:3]
            mjmath.mju_normalize3(normal)
            jac2 = np.empty((3, m.nv)); jac1 = np.empty((3, m.nv))
            mjmath.mj_jac(m, cfg.data, jac2, None, fromto[3:], int(m.geom_bodyid[g2]))
            mjmath.mj_jac(m, cfg.data, jac1, None, fromto[:3], int(m.geom_bodyid[g1]))
            coeff[k] = -(normal @ (jac2 - jac1))
        return coeff, upper_bound
    raise TypeError(type(spec))


# ----------------------------------------------------------------- solve_ik
def build_ik(cfg: Configuration, tasks, dt: float, damping: float = 1e-12, limits=None):
    """mink/solve_ik.py:13-65 → (P, q, G, h); G,h None when unconstrained."""
    nv = cfg.model.nv
    H = np.eye(nv) * damping
    c = np.zeros(nv)
    for task in tasks:
        H_task, c_task = task_qp_objective(cfg, task)
        H += H_task
        c += c_task
    if limits is None:
        limits = [ConfigurationLimitSpec()]
    G_list, h_list = [], []
    for lim in limits:
        G, h = limit_inequalities(cfg, lim, dt)
        if G is not None:
            G_list.append(G)
            h_list.append(h)
    if not G_list:
        return H, c, None, None
    return H, c, np.vstack(G_list), np.hstack(h_list)


def solve_ik(model, q, tasks, dt: float, damping: float = 1e-12, limits=None,
             return_problem: bool = False):
    """mink/solve_ik.py:68-105 for one problem instance → v (nv,)."""
    cfg = q if isinstance(q, Configuration) else Configuration(model, q)
    P, c, G, h = build_ik(cfg, tasks, dt, damping, limits)
    dq = qp_gi.solve_qp(P, c, G, h)
    v = dq / dt
    if return_problem:
        return v, (P, c, G, h)
    return v
