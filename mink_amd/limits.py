"""Limit plugin classes of the mink API (mink/limits/*.py): same constructors and validation.
Constructors run on the host once; the per-solve arithmetic is on the device."""

from __future__ import annotations

import abc
import itertools
from typing import List, Mapping, NamedTuple, Optional, Sequence, Union

import numpy as np

from . import api_specs
from .configuration import Configuration, as_flat_model
from .exceptions import LimitDefinitionError
from .flatmodel import JNT_FREE, dof_width


class Constraint(NamedTuple):
    """mink/limits/limit.py:11-23."""
    G: Optional[np.ndarray] = None
    h: Optional[np.ndarray] = None

    @property
    def inactive(self) -> bool:
        return self.G is None and self.h is None


class Limit(abc.ABC):
    """mink/limits/limit.py:26-57.  A subclass that only implements the reference's extension point
    `compute_qp_inequalities(configuration, dt)` → Constraint(G, h) with G (m, nv) or (B, m, nv) and h (m,) or (B, m)
    (h = +inf: row inactive) reaches the device as dense half-space rows (mkh_solve_dense); the built-in limits
    override `_native_desc` and are evaluated on the device."""

    def _native_desc(self):
        """(kind, descriptor dict) for mkh_problem_create."""
        return "dense_limit", None

    def _fingerprint(self):
        """Cheap hashable stand-in for the device descriptor (see Task._fingerprint); None: no shortcut."""
        return None

    def _is_dense(self) -> bool:
        return type(self)._native_desc is Limit._native_desc or \
            type(self).compute_qp_inequalities not in _BUILTIN_INEQUALITIES

    def _eval(self, configuration: Configuration, dt: float, taps):
        from .solve_ik import _compile
        prob, _ = _compile(configuration, [], limits=[self], batch=configuration.batch_size)
        _, _, out = prob.solve(configuration.q_batch, None, None, None, dt, 1.0, taps=taps, solve_qp=False)
        return out

    @abc.abstractmethod
    def compute_qp_inequalities(self, configuration: Configuration, dt: float) -> Constraint:
        raise NotImplementedError


class ConfigurationLimit(Limit):
    """mink/limits/configuration_limit.py:18-124."""

    def __init__(self, model, gain: float = 0.95, min_distance_from_limits: float = 0.0):
        if not 0.0 < gain <= 1.0:
            raise LimitDefinitionError(f"{self.__class__.__name__} gain must be in the range (0, 1]")
        m = as_flat_model(model)
        d = api_specs.configuration_limit_desc(m, gain, min_distance_from_limits)
        self.indices = d["indices"].astype(np.int64)
        self.indices.setflags(write=False)
        dim = len(self.indices)
        self.projection_matrix = np.eye(m.nv)[self.indices] if dim > 0 else None
        self.lower = d["lower"]
        self.upper = d["upper"]
        self.model = m
        self.gain = gain

    def _native_desc(self):
        return "cfg", {"gain": self.gain, "lower": self.lower, "upper": self.upper, "indices": self.indices}

    def _fingerprint(self):
        if self._is_dense():
            return None
        return (id(self), float(self.gain), self.lower.tobytes(), self.upper.tobytes(), np.asarray(self.indices).tobytes())

    def compute_qp_inequalities(self, configuration: Configuration, dt: float) -> Constraint:
        if self.projection_matrix is None:
            return Constraint()
        out = self._eval(configuration, dt, ["box_lo", "box_hi"])
        G = np.vstack([self.projection_matrix, -self.projection_matrix])
        h = np.concatenate([out["box_hi"][:, self.indices], -out["box_lo"][:, self.indices]], axis=1)
        return Constraint(G=G, h=configuration._unbatch(h))


class VelocityLimit(Limit):
    """mink/limits/velocity_limit.py:18-101."""

    def __init__(self, model, velocities: Mapping[str, object] = {}):
        m = as_flat_model(model)
        limit_list: List[float] = []
        index_list: List[int] = []
        for joint_name, max_vel in velocities.items():
            jid = m.joint(joint_name).id
            jnt_type = int(m.jnt_type[jid])
            if jnt_type == JNT_FREE:
                raise LimitDefinitionError(f"Free joint {joint_name} is not supported")
            vadr = int(m.jnt_dofadr[jid])
            vdim = dof_width(jnt_type)
            max_vel = np.atleast_1d(max_vel)
            if max_vel.shape != (vdim,):
                raise LimitDefinitionError(
                    f"Joint {joint_name} must have a limit of shape ({vdim},). Got: {max_vel.shape}")
            index_list.extend(range(vadr, vadr + vdim))
            limit_list.extend(max_vel.tolist())
        self.indices = np.array(index_list, dtype=np.int64)
        self.indices.setflags(write=False)
        self.limit = np.array(limit_list, dtype=np.float64)
        self.limit.setflags(write=False)
        nb = len(self.indices)
        self.projection_matrix = np.eye(m.nv)[self.indices] if nb > 0 else None
        self.model = m

    def _native_desc(self):
        return "vel", {"indices": self.indices, "limit": self.limit}

    def _fingerprint(self):
        if self._is_dense():
            return None
        return (id(self), np.asarray(self.indices).tobytes(), np.asarray(self.limit, dtype=np.float64).tobytes())

    def compute_qp_inequalities(self, configuration: Configuration, dt: float) -> Constraint:
        if self.projection_matrix is None:
            return Constraint()
        G = np.vstack([self.projection_matrix, -self.projection_matrix])
        h = np.hstack([dt * self.limit, dt * self.limit])
        return Constraint(G=G, h=h)


Geom = Union[int, str]
GeomSequence = Sequence[Geom]
CollisionPair = tuple
CollisionPairs = Sequence[CollisionPair]


class Contact(NamedTuple):
    """mink/limits/collision_avoidance_limit.py:20-56: what mj_geomDistance reports for one geom pair.  On the device
    a pair lane holds exactly these fields for its pair (collide_dev.h, ik_kernel.h "collision half-space rows");
    the class exists for callers that build or inspect contacts on the host."""
    dist: float
    fromto: np.ndarray
    geom1: int
    geom2: int
    distmax: float

    @property
    def normal(self) -> np.ndarray:
        """Unit vector from the closest point on geom1 to the one on geom2; (1, 0, 0) when they coincide
        (mju_normalize3)."""
        n = np.asarray(self.fromto[3:], dtype=np.float64) - np.asarray(self.fromto[:3], dtype=np.float64)
        norm = float(np.linalg.norm(n))
        return np.array([1.0, 0.0, 0.0]) if norm < 1e-15 else n / norm

    @property
    def inactive(self) -> bool:
        """No distance smaller than distmax was found."""
        return self.dist == self.distmax


def _is_welded_together(m, g1: int, g2: int) -> bool:
    return m.body_weldid[m.geom_bodyid[g1]] == m.body_weldid[m.geom_bodyid[g2]]


def _are_geom_bodies_parent_child(m, g1: int, g2: int) -> bool:
    w1 = m.body_weldid[m.geom_bodyid[g1]]
    w2 = m.body_weldid[m.geom_bodyid[g2]]
    wp1 = m.body_weldid[m.body_parentid[w1]]
    wp2 = m.body_weldid[m.body_parentid[w2]]
    return w1 == wp2 or w2 == wp1


def _is_pass_contype_conaffinity_check(m, g1: int, g2: int) -> bool:
    return bool(m.geom_contype[g1] & m.geom_conaffinity[g2]) or bool(m.geom_contype[g2] & m.geom_conaffinity[g1])


class CollisionAvoidanceLimit(Limit):
    """mink/limits/collision_avoidance_limit.py:118-278."""

    def __init__(self, model, geom_pairs: CollisionPairs, gain: float = 0.85,
                 minimum_distance_from_collisions: float = 0.005,
                 collision_detection_distance: float = 0.01, bound_relaxation: float = 0.0):
        self.model = as_flat_model(model)
        self.gain = gain
        self.minimum_distance_from_collisions = minimum_distance_from_collisions
        self.collision_detection_distance = collision_detection_distance
        self.bound_relaxation = bound_relaxation
        self.geom_id_pairs = self._construct_geom_id_pairs(geom_pairs)
        self.max_num_contacts = len(self.geom_id_pairs)

    def _homogenize_geom_id_list(self, geom_list: GeomSequence) -> List[int]:
        out: List[int] = []
        for g in geom_list:
            out.append(g if isinstance(g, (int, np.integer)) else self.model.geom(g).id)
        return [int(g) for g in out]

    def _construct_geom_id_pairs(self, geom_pairs):
        m = self.model
        pairs = []
        for pair in geom_pairs:
            a = list(set(self._homogenize_geom_id_list(pair[0])))
            b = list(set(self._homogenize_geom_id_list(pair[1])))
            for ga, gb in itertools.product(a, b):
                if (not _is_welded_together(m, ga, gb) and not _are_geom_bodies_parent_child(m, ga, gb)
                        and _is_pass_contype_conaffinity_check(m, ga, gb)):
                    # (checked on the pairs that survive the filters only: the reference's examples hand over whole
                    #  subtrees, visual mesh geoms included — examples/arm_aloha.py:95-104 — which contype / conaffinity
                    #  = 0 then drops)
                    for g in (ga, gb):
                        if not m.geom_valid[g]:
                            raise LimitDefinitionError(
                                f"geom {g} ('{m.geom_names[g]}') takes its size and frame from a mesh asset that was "
                                "not available when the model was read (MJCF subset reader): it cannot be a collision "
                                "candidate")
                    pairs.append((min(ga, gb), max(ga, gb)))
        return pairs

    def _native_desc(self):
        return "col", {"geom_id_pairs": np.array(self.geom_id_pairs, dtype=np.int32).reshape(-1, 2),
                       "gain": self.gain,
                       "minimum_distance_from_collisions": self.minimum_distance_from_collisions,
                       "collision_detection_distance": self.collision_detection_distance,
                       "bound_relaxation": self.bound_relaxation}

    def _fingerprint(self):
        if self._is_dense():
            return None
        return (id(self), float(self.gain), float(self.minimum_distance_from_collisions),
                float(self.collision_detection_distance), float(self.bound_relaxation),
                np.asarray(self.geom_id_pairs, dtype=np.int32).tobytes())

    def compute_qp_inequalities(self, configuration: Configuration, dt: float) -> Constraint:
        out = self._eval(configuration, dt, ["coll_G", "coll_h"])
        return Constraint(G=configuration._unbatch(out["coll_G"]), h=configuration._unbatch(out["coll_h"]))


_BUILTIN_INEQUALITIES = (ConfigurationLimit.compute_qp_inequalities, VelocityLimit.compute_qp_inequalities,
                         CollisionAvoidanceLimit.compute_qp_inequalities)
