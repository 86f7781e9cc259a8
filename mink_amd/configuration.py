"""Configuration: a robot model plus a *batch* of configurations on one GPU.

Mirrors mink.Configuration (mink/configuration.py:21-253).  `q` may be (nq,) — then every
method behaves, shape-wise, like the reference — or (B, nq) for a batch; results gain the
same leading dimension.  All kinematics run on the device through libminkhip.so.
"""

from __future__ import annotations

import logging
import weakref
from typing import Optional

import numpy as np

from . import _native as nat
from . import exceptions
from .constants import SUPPORTED_FRAMES
from .flatmodel import JNT_FREE, FlatModel
from .lie import SE3

_native_models = weakref.WeakKeyDictionary()


def as_flat_model(model) -> FlatModel:
    if isinstance(model, FlatModel):
        return model
    if type(model).__module__.split(".")[0] == "mujoco":
        cached = getattr(model, "_mink_amd_flat", None)
        return cached if cached is not None else FlatModel.from_mjmodel(model)
    raise TypeError(f"unsupported model type {type(model)!r}: pass a mink_amd FlatModel or a mujoco.MjModel")


def native_model(model: FlatModel, device: int = 0) -> "nat.NativeModel":
    per = _native_models.setdefault(model, {})
    if device not in per:
        per[device] = nat.NativeModel(model, device)
    return per[device]


class Configuration:
    def __init__(self, model, q: Optional[np.ndarray] = None, device=0):
        """`device`: a device index, or — one process driving several GPUs — a sequence of indices or "all"
        (every visible device): solve_ik / solve_ik_steps / build_ik then split the batch into contiguous row blocks, one
        per listed device (mink_amd.distributed.ShardedProblem; a device may be listed more than once)."""
        self.model = as_flat_model(model)
        if isinstance(device, str):
            if device != "all":
                raise ValueError("device must be an index, a sequence of indices or 'all'")
            device = list(range(max(1, nat.lib().mkh_device_count())))
        self.devices = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
        self.device = self.devices[0]             # kinematics queries (frame poses, Jacobians, CoM) run on the first one
        # compiled device descriptors of this configuration's call sites (insertion-ordered: LRU in solve_ik._compile).
        # A descriptor owns one ticket counter and one set of staging buffers, so calls on ONE Configuration must not
        # run concurrently from several threads/streams (include/minkhip.h "One in-flight call per MkhProblem").
        self._problems = {}
        self._compile_memo = {}          # (task / limit fingerprints, batch) → (cache key, layout): solve_ik._compile's shortcut
        self._default_limits = None      # the [ConfigurationLimit(model)] of solve_ik(limits=None), built once
        self._pinned_problems = {}       # cache key → nesting count of callers about to solve on that handle (never evicted)
        self._q = None
        self.update(q if q is not None else self.model.qpos0)

    # ------------------------------------------------------------------ state
    @property
    def native(self) -> "nat.NativeModel":
        return native_model(self.model, self.device)

    @property
    def batched(self) -> bool:
        return self._batched

    @property
    def batch_size(self) -> int:
        return self._q.shape[0]

    def update(self, q: Optional[np.ndarray] = None) -> None:
        """mink/configuration.py:53-64 (FK itself is recomputed on the device in every call)."""
        if q is None:
            return
        q = np.array(q, dtype=np.float64)
        if q.ndim == 1:
            self._batched = False
            q = q[None, :]
        elif q.ndim == 2:
            self._batched = True
        else:
            raise ValueError(f"q must have shape ({self.model.nq},) or (B, {self.model.nq})")
        if q.shape[1] != self.model.nq:
            raise ValueError(f"q must have {self.model.nq} columns, got {q.shape[1]}")
        self._q = np.ascontiguousarray(q)

    def update_from_keyframe(self, key_name: str) -> None:
        key_id = self.model.name2id("key", key_name)
        if key_id == -1:
            raise exceptions.InvalidKeyframe(key_name, self.model)
        self.update(q=self.model.key_qpos[key_id])

    @property
    def q(self) -> np.ndarray:
        return self._q.copy() if self._batched else self._q[0].copy()

    @property
    def q_batch(self) -> np.ndarray:
        return self._q

    @property
    def nv(self) -> int:
        return self.model.nv

    @property
    def nq(self) -> int:
        return self.model.nq

    def _unbatch(self, x):
        return x if self._batched else x[0]

    # ------------------------------------------------------------ limit check
    def check_limits(self, tol: float = 1e-6, safety_break: bool = True) -> None:
        """mink/configuration.py:77-110 for every instance of the batch."""
        m = self.model
        for jnt in range(m.njnt):
            if m.jnt_type[jnt] == JNT_FREE or not m.jnt_limited[jnt]:
                continue
            qv = self._q[:, m.jnt_qposadr[jnt]]
            qmin, qmax = m.jnt_range[jnt]
            bad = np.nonzero((qv < qmin - tol) | (qv > qmax + tol))[0]
            if len(bad):
                val = float(qv[bad[0]])
                if safety_break:
                    raise exceptions.NotWithinConfigurationLimits(joint_id=jnt, value=val, lower=qmin,
                                                                  upper=qmax, model=m)
                logging.warning(f"Value {val:.2f} at index {jnt} is outside of its limits: "
                                f"[{qmin:.2f}, {qmax:.2f}]" + (f" ({len(bad)} instances)" if self._batched else ""))

    # -------------------------------------------------------------- frames
    def _frame_id(self, frame_name: str, frame_type: str) -> int:
        if frame_type not in SUPPORTED_FRAMES:
            raise exceptions.UnsupportedFrame(frame_type, SUPPORTED_FRAMES)
        fid = self.model.name2id(frame_type, frame_name)
        if fid == -1:
            raise exceptions.InvalidFrame(frame_name=frame_name, frame_type=frame_type, model=self.model)
        if frame_type == "geom" and self.model.geom_valid[fid] != 1:     # (2: a mesh-derived frame of the MJCF reader, see mjcf.py)
            # primitive fitted to a mesh (type="capsule" mesh=...): its frame comes from the mesh asset, which the
            # MJCF subset reader does not have — refuse rather than report a pose computed from placeholder values
            raise exceptions.InvalidFrame(frame_name=frame_name, frame_type=frame_type, model=self.model,
                                          reason="its local frame is derived from a mesh asset that was not available "
                                                 "when the model was read")
        return fid

    def _frame_problem(self, fid: int, frame_type: str) -> "nat.NativeProblem":
        key = ("frame", fid, frame_type, self._q.shape[0])
        return self._cached_problem(key, lambda: nat.NativeProblem(
            self.native, frame_tasks=[{"frame_type": frame_type, "frame_id": fid, "cost": [1.0] * 6}],
            max_batch=self._q.shape[0]))

    def _cached_problem(self, key, make) -> "nat.NativeProblem":
        """Helper descriptors share the LRU cache of solve_ik._compile: every use re-inserts the entry as most recently
        used, so that a control loop which retunes task costs every step does not evict and rebuild them each time."""
        prob = self._problems.pop(key, None)
        if prob is None:
            prob = make()
        self._problems[key] = prob
        return prob

    def _frame_pose_batch(self, frame_name: str, frame_type: str) -> np.ndarray:
        fid = self._frame_id(frame_name, frame_type)
        prob = self._frame_problem(fid, frame_type)
        B = self._q.shape[0]
        dummy = np.zeros((B, 1, 7)); dummy[:, :, 0] = 1.0
        _, _, taps = prob.solve(self._q, dummy, None, None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
        return taps["frame_pose"][:, 0]

    def get_transform_frame_to_world(self, frame_name: str, frame_type: str) -> SE3:
        """mink/configuration.py:157-185."""
        return SE3(self._unbatch(self._frame_pose_batch(frame_name, frame_type)))

    def get_frame_jacobian(self, frame_name: str, frame_type: str) -> np.ndarray:
        """mink/configuration.py:112-155: body Jacobian ᴮJ (6, nv).  Evaluated as the FrameTask
        Jacobian with the target at the frame itself (jlog = I there), J_task = −ᴮJ."""
        fid = self._frame_id(frame_name, frame_type)
        pose = self._frame_pose_batch(frame_name, frame_type)
        prob = self._frame_problem(fid, frame_type)
        _, _, taps = prob.solve(self._q, pose[:, None, :], None, None, 1.0, 1.0, taps=["task_J"], solve_qp=False)
        return self._unbatch(-taps["task_J"])

    def get_transform(self, source_name: str, source_type: str, dest_name: str, dest_type: str) -> SE3:
        """mink/configuration.py:187-212."""
        a = self.get_transform_frame_to_world(source_name, source_type)
        b = self.get_transform_frame_to_world(dest_name, dest_type)
        return b.inverse() @ a

    def subtree_com(self) -> np.ndarray:
        """data.subtree_com[1] (used by ComTask.set_target_from_configuration)."""
        self.model.require_valid_masses("subtree_com")
        key = ("com", self._q.shape[0])
        prob = self._cached_problem(key, lambda: nat.NativeProblem(self.native, com_tasks=[{"cost": 1.0}],
                                                                   max_batch=self._q.shape[0]))
        _, _, taps = prob.solve(self._q, None, None, np.zeros((1, 3)), 1.0, 1.0,
                                               taps=["subtree_com"], solve_qp=False)
        return self._unbatch(taps["subtree_com"])

    # ------------------------------------------------------------- integrate
    def integrate(self, velocity: np.ndarray, dt: float) -> np.ndarray:
        """mink/configuration.py:214-226."""
        v = np.asarray(velocity, dtype=np.float64)
        v = np.broadcast_to(v if v.ndim == 2 else v[None, :], (self._q.shape[0], self.model.nv))
        return self._unbatch(self.native.integrate(self._q, np.ascontiguousarray(v), dt))

    def integrate_inplace(self, velocity: np.ndarray, dt: float) -> None:
        """mink/configuration.py:228-236."""
        q = self.integrate(velocity, dt)
        self._q = np.ascontiguousarray(q if self._batched else q[None, :])
