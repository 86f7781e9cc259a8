"""Task plugin classes of the mink API: same constructors, validation and error strings as
mink/tasks/{task,frame_task,posture_task,com_task,damping_task}.py; targets may be batched.
`compute_error` / `compute_jacobian` / `compute_qp_objective` evaluate on the device."""

from __future__ import annotations

import abc
import threading
from typing import NamedTuple, Optional

import numpy as np

from . import _native as nat
from .configuration import Configuration, as_flat_model
from .exceptions import (InvalidDamping, InvalidGain, InvalidTarget, TargetNotSet, TaskDefinitionError)
from .lie import SE3


class Objective(NamedTuple):
    """mink/tasks/task.py:12-22."""
    H: np.ndarray
    c: np.ndarray

    def value(self, x: np.ndarray) -> float:
        return x.T @ self.H @ x + self.c @ x


_DENSE_BY_CLASS: dict = {}


def objective_to_rows(H: np.ndarray, c: np.ndarray, name: str = "task"):
    """Rows (e, J) — nv of them, cost 1, gain 1, no Levenberg–Marquardt term — whose contribution to the QP under mink's own
    formula (tasks/task.py:125-138: H += JᵀJ, c += Jᵀe for that cost and gain) IS the given objective: JᵀJ = H and Jᵀe = c.
    This is how a `compute_qp_objective` override (the method mink/solve_ik.py:18-21 actually calls) reaches the device, whose
    plugin route takes rows.  H: (nv, nv) or (B, nv, nv) symmetric positive semi-definite; c: (nv,) or (B, nv).

    H = V·Λ·Vᵀ gives J = Λ^½·Vᵀ (rows of vanishing eigenvalues are zero) and e_i = v_iᵀc / √λ_i.  A linear term with a
    component c_n outside the range of H — no least-squares objective has one, but the reference's QP accepts it — takes one
    of the zero rows: J_s = s·n̂ᵀ, e_s = ‖c_n‖ / s with s² = 1e-20·λ_max, i.e. the term enters c exactly and perturbs H four
    orders of magnitude below its own rounding."""
    H = np.asarray(H, dtype=np.float64); c = np.asarray(c, dtype=np.float64)
    nv = H.shape[-1]
    if H.shape[-2:] != (nv, nv) or H.ndim not in (2, 3) or c.shape[-1] != nv or c.ndim not in (1, 2):
        raise TaskDefinitionError(f"{name}.compute_qp_objective must return H of shape (nv, nv) or (B, nv, nv) and c of shape "
                                  f"(nv,) or (B, nv); got {H.shape}, {c.shape}")
    batched = H.ndim == 3 or c.ndim == 2
    Hb = H if H.ndim == 3 else H[None]
    cb = c if c.ndim == 2 else c[None]
    if not (np.isfinite(Hb).all() and np.isfinite(cb).all()):
        raise TaskDefinitionError(f"{name}.compute_qp_objective returned non-finite entries")
    scale = np.abs(Hb).max(axis=(1, 2))                                    # (b,)
    if (np.abs(Hb - np.swapaxes(Hb, 1, 2)).max(axis=(1, 2)) > 1e-9 * scale + 1e-300).any():
        raise TaskDefinitionError(f"{name}.compute_qp_objective: H must be symmetric")
    w, V = np.linalg.eigh(0.5 * (Hb + np.swapaxes(Hb, 1, 2)))             # ascending
    w, V = w[:, ::-1], V[:, :, ::-1]                                       # largest first: the nonzero rows lead
    lam = np.maximum(w[:, 0], 0.0)
    if (w[:, -1] < -1e-9 * np.maximum(lam, scale)).any():
        raise TaskDefinitionError(f"{name}.compute_qp_objective: H must be positive semi-definite (quadprog, the reference's "
                                  f"solver, needs a convex objective)")
    tol = (8.0 * nv * np.finfo(np.float64).eps) * lam
    keep = w > tol[:, None]                                                # (b, nv)
    sq = np.sqrt(np.where(keep, w, 0.0))
    J = sq[:, :, None] * np.swapaxes(V, 1, 2)                              # row i = √λ_i·v_iᵀ
    nb = max(Hb.shape[0], cb.shape[0])
    if Hb.shape[0] != nb:
        J, V, keep, sq, lam = (np.broadcast_to(a, (nb,) + a.shape[1:]) for a in (J, V, keep, sq, lam))
    cc = np.broadcast_to(cb, (nb, nv))
    vc = np.einsum("bki,bk->bi", V, cc)                                    # v_iᵀc
    e = np.where(keep, vc / np.where(keep, sq, 1.0), 0.0)
    c_n = cc - np.einsum("bik,bi->bk", J, e)                               # what the range of H cannot carry
    n_n = np.linalg.norm(c_n, axis=1)
    need = n_n > 1e-13 * np.maximum(np.linalg.norm(cc, axis=1), 1e-300)
    if need.any():
        J = np.array(J); e = np.array(e)
        rank = keep.sum(axis=1)
        idx = np.nonzero(need & (rank < nv))[0]                            # (full rank: c_n is rounding)
        s = np.where(lam[idx] > 0.0, 1e-10 * np.sqrt(lam[idx]), 1e-12)
        J[idx, rank[idx]] = s[:, None] * (c_n[idx] / n_n[idx, None])
        e[idx, rank[idx]] = n_n[idx] / s
    if not batched:
        return np.ascontiguousarray(e[0]), np.ascontiguousarray(J[0])
    return np.ascontiguousarray(e), np.ascontiguousarray(J)


# Re-entrancy marks of Task._eval / compute_qp_objective ("evaluate THIS instance through its rows" / "through the built-in
# descriptor"), per THREAD: as instance attributes two threads sharing a task object (distributed workers) could pair an nv-row
# descriptor with k-row data (round-5 advisor finding).
_marks = threading.local()


def _mark(kind: str, obj, delta: int = 0) -> int:
    d = _marks.__dict__.setdefault(kind, {})
    n = d.get(id(obj), 0) + delta
    if n:
        d[id(obj)] = n
    else:
        d.pop(id(obj), None)
    return n


class Task(abc.ABC):
    """mink/tasks/task.py:25-138."""

    def __init__(self, cost: np.ndarray, gain: float = 1.0, lm_damping: float = 0.0):
        if not 0.0 <= gain <= 1.0:
            raise InvalidGain("`gain` must be in the range [0, 1]")
        if lm_damping < 0.0:
            raise InvalidDamping("`lm_damping` must be >= 0")
        self.cost = cost
        self.gain = gain
        self.lm_damping = lm_damping

    # -- device plumbing -------------------------------------------------
    # The extension point is the reference's (mink/tasks/task.py:81-103): a subclass that implements
    # `compute_error(configuration)` → (k,) or (B, k) and `compute_jacobian(configuration)` → (k, nv) or (B, k, nv)
    # — with numpy, for the whole batch of the Configuration — reaches the device as DENSE ROWS (mkh_solve_dense,
    # include/minkhip.h) and is folded into the QP by the kernel like any built-in task.  The built-in classes below
    # override `_native_desc` / `_native_target` instead: their error and Jacobian are computed on the device.
    def _native_desc(self, configuration: Configuration):
        """(kind, descriptor dict) for mkh_problem_create."""
        if self._objective_overridden():
            # nv rows that carry the override's (H, c) exactly (objective_to_rows): unit cost and gain, no LM term
            return "dense", {"cost": np.ones(configuration.nv), "gain": 1.0, "lm_damping": 0.0}
        return "dense", {"cost": np.array(self.cost, dtype=np.float64).reshape(-1), "gain": float(self.gain),
                         "lm_damping": float(self.lm_damping)}

    def _objective_overridden(self) -> bool:
        """Does the (H, c) of this task come from a `compute_qp_objective` override?  Not while the base-class method itself is
        being evaluated for this instance (an override that calls `super().compute_qp_objective(configuration)`)."""
        if _mark("rows_only", self):
            return False
        return type(self).compute_qp_objective is not self._builtin_class().compute_qp_objective

    def _native_target(self, configuration: Configuration) -> np.ndarray:
        """Target rows for mkh_solve (built-in tasks only)."""
        raise NotImplementedError

    def _fingerprint(self):
        """A cheap, hashable stand-in for this task's device descriptor (solve_ik._compile's memo: a control loop calls
        solve_ik with the same task objects thousands of times a second, and building + hashing the descriptors cost more
        than the launch).  Everything `_native_desc` reads must be in it BY VALUE — costs are arrays the setters modify in
        place.  None: no shortcut for this task (caller-defined tasks: their rows are evaluated per call anyway)."""
        return None

    def _fp_common(self):
        return (id(self), np.asarray(self.cost, dtype=np.float64).tobytes(), float(self.gain), float(self.lm_damping),
                _mark("force_builtin", self))

    _PLUGIN_METHODS = ("compute_error", "compute_jacobian", "compute_qp_objective")

    def _builtin_class(self) -> type:
        """The class whose device descriptor this task uses: the nearest one in the MRO that defines `_native_desc`
        (FrameTask for a caller's FrameTask subclass; Task itself for a task written from scratch)."""
        for c in type(self).__mro__:
            if "_native_desc" in c.__dict__:
                return c
        return Task

    def _is_dense(self) -> bool:
        """Does this task reach the device as dense rows?  Yes for a task written from scratch, and for a subclass of a
        built-in task that overrides ANY of compute_error / compute_jacobian / compute_qp_objective: the reference calls
        those through the instance (mink/solve_ik.py:13-22 → tasks/task.py:105-138), so an override must change the
        answer here too — the built-in device descriptor would silently ignore it."""
        if _mark("force_builtin", self):
            return False
        cls = type(self)
        hit = _DENSE_BY_CLASS.get(cls)
        if hit is None or hit[0] != tuple(getattr(cls, n) for n in Task._PLUGIN_METHODS):     # (methods patched onto a class: re-derive)
            base = self._builtin_class()
            methods = tuple(getattr(cls, n) for n in Task._PLUGIN_METHODS)
            hit = (methods, base is Task or any(m is not getattr(base, n) for m, n in zip(methods, Task._PLUGIN_METHODS)))
            _DENSE_BY_CLASS[cls] = hit
        return hit[1]

    def _dense_rows(self, configuration: Configuration):
        """(e, J) of this caller-defined task for every instance: (B, k), (B, k, nv).  A half the subclass inherits from a
        built-in task (say, compute_jacobian of FrameTask under an overridden compute_error) is evaluated on the device
        through the built-in descriptor (Task.compute_error / compute_jacobian below)."""
        base, cls = self._builtin_class(), type(self)
        B, nv = configuration.batch_size, configuration.nv
        if self._objective_overridden():
            # The reference only ever calls compute_qp_objective (mink/solve_ik.py:18-21), so a subclass that returns its own
            # (H, c) changes the QP there.  The device folds tasks in from rows: the override's objective is factored into nv
            # rows with JᵀJ = H, Jᵀe = c (objective_to_rows), evaluated here with numpy like any caller-defined row.
            obj = self.compute_qp_objective(configuration)
            try:
                H, c = obj
            except (TypeError, ValueError):
                raise TaskDefinitionError(f"{cls.__name__}.compute_qp_objective must return an Objective (H, c)") from None
            H, c = np.asarray(H, dtype=np.float64), np.asarray(c, dtype=np.float64)
            if H.shape not in ((nv, nv), (B, nv, nv)) or c.shape not in ((nv,), (B, nv)):
                raise TaskDefinitionError(f"{cls.__name__}.compute_qp_objective must return H ({nv}, {nv}) or ({B}, {nv}, {nv}) "
                                          f"and c ({nv},) or ({B}, {nv}); got {H.shape}, {c.shape}")
            e, J = objective_to_rows(H, c, cls.__name__)
            return np.broadcast_to(e, (B, nv)), np.broadcast_to(J, (B, nv, nv))
        if base is Task and (cls.compute_error is Task.compute_error or cls.compute_jacobian is Task.compute_jacobian):
            raise TaskDefinitionError(f"{cls.__name__} must implement compute_error and compute_jacobian "
                                      "(mink's Task plugin interface)")
        k = len(np.atleast_1d(self.cost))
        e = np.asarray(self.compute_error(configuration), dtype=np.float64)
        J = np.asarray(self.compute_jacobian(configuration), dtype=np.float64)
        if e.shape not in ((k,), (B, k)) or J.shape not in ((k, nv), (B, k, nv)):
            raise TaskDefinitionError(f"{cls.__name__}: compute_error must return ({k},) or ({B}, {k}) and "
                                      f"compute_jacobian ({k}, {nv}) or ({B}, {k}, {nv}); got {e.shape}, {J.shape}")
        return np.broadcast_to(e, (B, k)), np.broadcast_to(J, (B, k, nv))

    def _eval(self, configuration: Configuration, taps, builtin: bool = False):
        """Evaluate this task alone on the device.  `builtin`: through the built-in descriptor even when the instance
        overrides a plugin method (the inherited half of a partially overridden built-in task)."""
        from .solve_ik import _compile, _dense_inputs, _gather_targets, _pin
        force = builtin and self._builtin_class() is not Task
        if force:
            _mark("force_builtin", self, +1)
        try:
            prob, layout = _compile(configuration, [self], limits=[], batch=configuration.batch_size)
            ft, pt, ct = _gather_targets(configuration, layout)
            with _pin(configuration, layout):
                _, _, out = prob.solve(configuration.q_batch, ft, pt, ct, 1.0, 0.0, taps=taps, solve_qp=False,
                                       dense=_dense_inputs(configuration, layout, 1.0))
        finally:
            if force:
                _mark("force_builtin", self, -1)
        return out

    def _no_rows_of_an_objective(self, what: str) -> None:
        # a task written from scratch that only overrides compute_qp_objective has no (e, J) of its own: the rows the device
        # sees are a factorisation of its (H, c) (objective_to_rows, possibly with a synthetic row of norm 1e10·‖c_n‖), not an
        # error or a Jacobian in mink's sense (round-5 advisor finding: they used to be returned)
        if self._builtin_class() is Task and self._objective_overridden():
            raise TaskDefinitionError(f"{type(self).__name__} defines its objective through compute_qp_objective only: it has no "
                                      f"{what} (implement compute_error and compute_jacobian to get one)")

    def compute_error(self, configuration: Configuration) -> np.ndarray:
        self._no_rows_of_an_objective("compute_error")
        return configuration._unbatch(self._eval(configuration, ["task_e"], builtin=True)["task_e"])

    def compute_jacobian(self, configuration: Configuration) -> np.ndarray:
        self._no_rows_of_an_objective("compute_jacobian")
        return configuration._unbatch(self._eval(configuration, ["task_J"], builtin=True)["task_J"])

    def compute_qp_objective(self, configuration: Configuration) -> Objective:
        # mink's formula from this task's rows (tasks/task.py:105-138) — also when a subclass's override calls it through
        # super(): the rows are then the instance's compute_error / compute_jacobian, not the override's own (H, c)
        base, cls = self._builtin_class(), type(self)
        own_rows = base is Task or cls.compute_error is not base.compute_error or cls.compute_jacobian is not base.compute_jacobian
        _mark("rows_only", self, +1)
        try:
            out = self._eval(configuration, ["H", "c"], builtin=not own_rows)   # damping = 0 ⇒ exactly this task's (H, c)
        finally:
            _mark("rows_only", self, -1)
        return Objective(configuration._unbatch(out["H"]), configuration._unbatch(out["c"]))


class FrameTask(Task):
    """mink/tasks/frame_task.py:16-146."""

    k: int = 6

    def __init__(self, frame_name: str, frame_type: str, position_cost, orientation_cost, gain: float = 1.0,
                 lm_damping: float = 0.0):
        super().__init__(cost=np.zeros((self.k,)), gain=gain, lm_damping=lm_damping)
        self.frame_name = frame_name
        self.frame_type = frame_type
        self.position_cost = position_cost
        self.orientation_cost = orientation_cost
        self.transform_target_to_world: Optional[SE3] = None
        self.set_position_cost(position_cost)
        self.set_orientation_cost(orientation_cost)

    def set_position_cost(self, position_cost) -> None:
        position_cost = np.atleast_1d(position_cost)
        if position_cost.ndim != 1 or position_cost.shape[0] not in (1, 3):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} position cost should be a vector of shape "
                "1 (aka identical cost for all coordinates) or (3,) but got "
                f"{position_cost.shape}")
        if not np.all(position_cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} position cost should be >= 0")
        self.cost[:3] = position_cost

    def set_orientation_cost(self, orientation_cost) -> None:
        orientation_cost = np.atleast_1d(orientation_cost)
        if orientation_cost.ndim != 1 or orientation_cost.shape[0] not in (1, 3):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} orientation cost should be a vector of "
                "shape 1 (aka identical cost for all coordinates) or (3,) but got "
                f"{orientation_cost.shape}")
        if not np.all(orientation_cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} position cost should be >= 0")
        self.cost[3:] = orientation_cost

    def set_target(self, transform_target_to_world: SE3) -> None:
        """Single pose or a batch (B, 7); copied (mink/tasks/frame_task.py:77-83)."""
        self.transform_target_to_world = transform_target_to_world.copy()

    def set_target_from_configuration(self, configuration: Configuration) -> None:
        self.set_target(configuration.get_transform_frame_to_world(self.frame_name, self.frame_type))

    def _native_desc(self, configuration):
        fid = configuration._frame_id(self.frame_name, self.frame_type)
        return "frame", {"frame_type": self.frame_type, "frame_id": fid, "cost": self.cost.tolist(),
                         "gain": self.gain, "lm_damping": self.lm_damping}

    def _fingerprint(self):
        return None if self._is_dense() else self._fp_common() + (self.frame_name, self.frame_type)

    def _native_target(self, configuration):
        if self.transform_target_to_world is None:
            raise TargetNotSet(self.__class__.__name__)
        t = self.transform_target_to_world.wxyz_xyz
        B = configuration.batch_size
        if t.ndim == 1:
            return np.broadcast_to(t, (B, 7))
        if t.shape != (B, 7):
            raise InvalidTarget(f"Expected target pose batch to have shape ({B}, 7) but got {t.shape}")
        return t


class PostureTask(Task):
    """mink/tasks/posture_task.py:17-142."""

    def __init__(self, model, cost, gain: float = 1.0, lm_damping: float = 0.0):
        m = as_flat_model(model)
        super().__init__(cost=np.zeros((m.nv,)), gain=gain, lm_damping=lm_damping)
        self.target_q: Optional[np.ndarray] = None
        _, v_ids = m.freejoint_dims()
        self._v_ids = np.asarray(v_ids) if v_ids else None
        self.k = m.nv
        self.nq = m.nq
        self.set_cost(cost)

    def set_cost(self, cost) -> None:
        cost = np.atleast_1d(cost)
        if cost.ndim != 1 or cost.shape[0] not in (1, self.k):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} cost must be a vector of shape (1,) "
                f"(aka identical cost for all dofs) or ({self.k},). Got {cost.shape}")
        if not np.all(cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} cost should be >= 0")
        self.cost[: self.k] = cost

    def set_target(self, target_q) -> None:
        target_q = np.atleast_1d(target_q)
        if target_q.ndim == 2 and target_q.shape[1] == self.nq:
            self.target_q = np.array(target_q, dtype=np.float64)      # batched posture target
            return
        if target_q.ndim != 1 or target_q.shape[0] != (self.nq):
            raise InvalidTarget(f"Expected target posture to have shape ({self.nq},) but got {target_q.shape}")
        self.target_q = np.array(target_q, dtype=np.float64)

    def set_target_from_configuration(self, configuration: Configuration) -> None:
        self.set_target(configuration.q)

    def _native_desc(self, configuration):
        return "posture", {"cost": self.cost.copy(), "gain": self.gain, "lm_damping": self.lm_damping}

    def _fingerprint(self):
        return None if self._is_dense() else self._fp_common()

    def _native_target(self, configuration):
        if self.target_q is None:
            raise TargetNotSet(self.__class__.__name__)
        return self.target_q


class DampingTask(PostureTask):
    """mink/tasks/damping_task.py:11-20."""

    def __init__(self, model, cost):
        m = as_flat_model(model)
        super().__init__(model=m, cost=cost, gain=0.0, lm_damping=0.0)
        self.target_q = np.array(m.qpos0)


class ComTask(Task):
    """mink/tasks/com_task.py:16-97."""

    k: int = 3

    def __init__(self, cost, gain: float = 1.0, lm_damping: float = 0.0):
        super().__init__(cost=np.zeros((self.k,)), gain=gain, lm_damping=lm_damping)
        self.target_com: Optional[np.ndarray] = None
        self.set_cost(cost)

    def set_cost(self, cost) -> None:
        cost = np.atleast_1d(cost)
        if cost.ndim != 1 or cost.shape[0] not in (1, self.k):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} cost must be a vector of shape (1,) "
                f"(aka identical cost for all coordinates) or ({self.k},). "
                f"Got {cost.shape}")
        if not np.all(cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} cost must be >= 0")
        self.cost[:] = cost

    def set_target(self, target_com) -> None:
        target_com = np.atleast_1d(target_com)
        if target_com.ndim == 2 and target_com.shape[1] == self.k:
            self.target_com = np.array(target_com, dtype=np.float64)
            return
        if target_com.ndim != 1 or target_com.shape[0] != (self.k):
            raise InvalidTarget(f"Expected target CoM to have shape ({self.k},) but got {target_com.shape}")
        self.target_com = np.array(target_com, dtype=np.float64)

    def set_target_from_configuration(self, configuration: Configuration) -> None:
        self.set_target(configuration.subtree_com())

    def _native_desc(self, configuration):
        configuration.model.require_valid_masses("ComTask")
        return "com", {"cost": self.cost.tolist(), "gain": self.gain, "lm_damping": self.lm_damping}

    def _fingerprint(self):
        return None if self._is_dense() else self._fp_common()

    def _native_target(self, configuration):
        if self.target_com is None:
            raise TargetNotSet(self.__class__.__name__)
        return self.target_com


class RelativeFrameTask(FrameTask):
    """mink/tasks/relative_frame_task.py:16-142: pose of a frame relative to another moving frame."""

    def __init__(self, frame_name: str, frame_type: str, root_name: str, root_type: str, position_cost,
                 orientation_cost, gain: float = 1.0, lm_damping: float = 0.0):
        super().__init__(frame_name, frame_type, position_cost, orientation_cost, gain=gain, lm_damping=lm_damping)
        self.root_name = root_name
        self.root_type = root_type

    @property
    def transform_target_to_root(self) -> Optional[SE3]:
        return self.transform_target_to_world

    def set_target_from_configuration(self, configuration: Configuration) -> None:
        self.set_target(configuration.get_transform(self.frame_name, self.frame_type, self.root_name, self.root_type))

    def _fingerprint(self):
        fp = super()._fingerprint()
        return None if fp is None else fp + (self.root_name, self.root_type)

    def _native_desc(self, configuration):
        kind, d = super()._native_desc(configuration)
        d["root_type"] = self.root_type
        d["root_id"] = configuration._frame_id(self.root_name, self.root_type)
        return kind, d
