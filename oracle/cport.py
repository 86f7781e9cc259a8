"""ORACLE (test infrastructure) — ctypes binding of oracle/c/libmink_oracle.so, the plain-C restatement of the
reference's solve_ik pipeline (see oracle/c/mink_oracle.h for the reference citations).

Takes the same spec objects as oracle/ik.py (FrameTaskSpec, PostureTaskSpec, ComTaskSpec,
ConfigurationLimitSpec, VelocityLimitSpec).  Used by tests/ (full-batch parity of the GPU path at BASELINE
sizes) and by bench.py's cpu_baseline leg — never by the product (mink_amd/).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

from . import ik

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "c", "libmink_oracle.so")
_lib = None

PI32 = C.POINTER(C.c_int32)
PF64 = C.POINTER(C.c_double)
_INT_FIELDS = ("body_parentid", "body_rootid", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum", "body_mocapid")
_MODEL_FIELDS = (
    [(n, C.c_int32) for n in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite")]
    + [(n, PI32) for n in _INT_FIELDS]
    + [(n, PF64) for n in ("body_pos", "body_quat", "body_ipos", "body_mass", "body_subtreemass")]
    + [(n, PI32) for n in ("jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited")]
    + [(n, PF64) for n in ("jnt_pos", "jnt_axis", "jnt_range")]
    + [("dof_parentid", PI32), ("qpos0", PF64), ("site_bodyid", PI32), ("site_pos", PF64), ("site_quat", PF64),
       ("geom_bodyid", PI32), ("geom_pos", PF64), ("geom_quat", PF64), ("mocap_pos", PF64), ("mocap_quat", PF64),
       ("geom_type", PI32), ("geom_size", PF64)]
)


class MkoModel(C.Structure):
    _fields_ = _MODEL_FIELDS


class MkoFrameTask(C.Structure):
    _fields_ = [("frame_type", C.c_int32), ("frame_id", C.c_int32), ("cost", C.c_double * 6), ("gain", C.c_double),
                ("lm_damping", C.c_double), ("root_type", C.c_int32), ("root_id", C.c_int32)]


class MkoPostureTask(C.Structure):
    _fields_ = [("cost", PF64), ("gain", C.c_double), ("lm_damping", C.c_double)]


class MkoComTask(C.Structure):
    _fields_ = [("cost", C.c_double * 3), ("gain", C.c_double), ("lm_damping", C.c_double)]


class MkoCollisionLimit(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("pairs", PI32), ("gain", C.c_double), ("minimum_distance", C.c_double),
                ("detection_distance", C.c_double), ("bound_relaxation", C.c_double)]


class MkoDenseTask(C.Structure):
    _fields_ = [("k", C.c_int32), ("cost", PF64), ("gain", C.c_double), ("lm_damping", C.c_double)]


class MkoDenseRows(C.Structure):
    _fields_ = [("task_e", PF64), ("task_J", PF64), ("limit_G", PF64), ("limit_h", PF64)]


class MkoProblem(C.Structure):
    _fields_ = [("n_frame", C.c_int32), ("frame", C.POINTER(MkoFrameTask)), ("n_posture", C.c_int32),
                ("posture", C.POINTER(MkoPostureTask)), ("n_com", C.c_int32), ("com", C.POINTER(MkoComTask)),
                ("has_cfg_limit", C.c_int32), ("cfg_gain", C.c_double), ("cfg_min_distance", C.c_double),
                ("n_vel", C.c_int32), ("vel_idx", PI32), ("vel_limit", PF64),
                ("n_coll", C.c_int32), ("coll", C.POINTER(MkoCollisionLimit)),
                ("n_dense", C.c_int32), ("dense", C.POINTER(MkoDenseTask)), ("n_dense_limit_rows", C.c_int32)]


def build() -> str:
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "c")], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "c", "mink_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            build()
        L = C.CDLL(LIB)
        L.mko_solve_ik_batch.restype = C.c_int32
        L.mko_solve_ik_batch_dense.restype = C.c_int32
        L.mko_solve_ik.restype = C.c_int32
        L.mko_solve_qp.restype = C.c_int32
        L.mko_collision_rows.restype = C.c_int32
        _lib = L
    return _lib


_FRAME_TYPES = {"body": 0, "geom": 1, "site": 2}
# (mjtGeom pairs, smaller type first, with a routine in oracle/c/mink_oracle.c::geom_distance)
_C_PAIR_TYPES = {(3, 3), (2, 2), (2, 3), (0, 2), (0, 3), (0, 6), (0, 5), (2, 6), (2, 5), (3, 6), (3, 5)}


class CProblem:
    """One solve_ik call site (model + tasks + limits) compiled into the C structs."""

    def __init__(self, model, tasks: Sequence, limits: Optional[Sequence], dense_tasks: Sequence = (),
                 dense_limit_rows: int = 0):
        """dense_tasks: caller-defined tasks as dicts {cost (k,), gain, lm_damping} whose e / J arrive per instance in
        solve_batch(dense=...); dense_limit_rows: number of rows of the caller-defined limits (their G, h likewise)."""
        self.model = model
        self._keep = []          # numpy buffers the C structs point into
        mm = MkoModel()
        for n in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite"):
            setattr(mm, n, int(getattr(model, n)))
        for name, typ in _MODEL_FIELDS[6:]:
            dt = np.int32 if typ is PI32 else np.float64
            a = np.ascontiguousarray(np.asarray(getattr(model, name), dtype=dt).ravel())
            if a.size == 0:
                a = np.zeros(1, dtype=dt)
            self._keep.append(a)
            setattr(mm, name, a.ctypes.data_as(typ))
        self.cmodel = mm
        # (FrameTasks and RelativeFrameTasks share the frame-target slots, in the caller's order)
        frames = [t for t in tasks if isinstance(t, (ik.FrameTaskSpec, ik.RelativeFrameTaskSpec))]
        postures = [t for t in tasks if isinstance(t, ik.PostureTaskSpec)]
        coms = [t for t in tasks if isinstance(t, ik.ComTaskSpec)]
        if len(frames) + len(postures) + len(coms) != len(tasks):
            raise TypeError("the C restatement covers FrameTask, RelativeFrameTask, PostureTask and ComTask")
        # the C side adds objectives grouped by kind; float addition is not associative, so keep the
        # caller's order within each kind (H differs from the numpy oracle only at the 1e-16 level)
        self.frames, self.postures, self.coms = frames, postures, coms
        fa = (MkoFrameTask * max(1, len(frames)))()
        for i, t in enumerate(frames):
            fa[i].frame_type = _FRAME_TYPES[t.frame_type]
            fa[i].frame_id = int(t.frame_id)
            fa[i].cost[:] = [float(x) for x in t.cost]
            fa[i].gain = float(t.gain)
            fa[i].lm_damping = float(t.lm_damping)
            rel = isinstance(t, ik.RelativeFrameTaskSpec)
            fa[i].root_type = _FRAME_TYPES[t.root_type] if rel else -1
            fa[i].root_id = int(t.root_id) if rel else 0
        pa = (MkoPostureTask * max(1, len(postures)))()
        for i, t in enumerate(postures):
            c = np.ascontiguousarray(np.asarray(t.cost, dtype=np.float64))
            self._keep.append(c)
            pa[i].cost = c.ctypes.data_as(PF64)
            pa[i].gain = float(t.gain)
            pa[i].lm_damping = float(t.lm_damping)
        ca = (MkoComTask * max(1, len(coms)))()
        for i, t in enumerate(coms):
            ca[i].cost[:] = [float(x) for x in t.cost]
            ca[i].gain = float(t.gain)
            ca[i].lm_damping = float(t.lm_damping)
        self._keep += [fa, pa, ca]
        pr = MkoProblem()
        pr.n_frame, pr.frame = len(frames), fa
        pr.n_posture, pr.posture = len(postures), pa
        pr.n_com, pr.com = len(coms), ca
        if limits is None:
            limits = [ik.ConfigurationLimitSpec()]                 # solve_ik.py:28-29
        cfg = [l for l in limits if isinstance(l, ik.ConfigurationLimitSpec)]
        vel = [l for l in limits if isinstance(l, ik.VelocityLimitSpec)]
        col = [l for l in limits if isinstance(l, ik.CollisionAvoidanceLimitSpec)]
        if len(cfg) > 1 or len(vel) > 1 or len(cfg) + len(vel) + len(col) != len(limits):
            raise TypeError("the C restatement covers one ConfigurationLimit, one VelocityLimit and CollisionAvoidanceLimits")
        # (collision rows are stacked after the box rows whatever their place in `limits`: the optimum of the strictly
        # convex QP does not depend on the row order)
        gt = np.asarray(model.geom_type)
        for l in col:
            for g1, g2 in l.geom_id_pairs:
                a, b = sorted((int(gt[g1]), int(gt[g2])))
                if (a, b) not in _C_PAIR_TYPES:
                    raise TypeError("the C restatement covers plane / sphere / capsule pairs and box / cylinder against "
                                    f"plane / sphere / capsule, not geom types ({a}, {b})")
        if cfg and vel and limits.index(cfg[0]) > limits.index(vel[0]):
            raise TypeError("row order: ConfigurationLimit first")
        pr.has_cfg_limit = 1 if cfg else 0
        if cfg:
            pr.cfg_gain, pr.cfg_min_distance = float(cfg[0].gain), float(cfg[0].min_distance_from_limits)
        if vel:
            vi = np.ascontiguousarray(np.asarray(vel[0].indices, dtype=np.int32))
            vl = np.ascontiguousarray(np.asarray(vel[0].limit, dtype=np.float64))
            self._keep += [vi, vl]
            pr.n_vel, pr.vel_idx, pr.vel_limit = len(vi), vi.ctypes.data_as(PI32), vl.ctypes.data_as(PF64)
        cl = (MkoCollisionLimit * max(1, len(col)))()
        for i, l in enumerate(col):
            pa_ = np.ascontiguousarray(np.asarray(l.geom_id_pairs, dtype=np.int32).reshape(-1, 2))
            self._keep.append(pa_)
            cl[i].n_pairs, cl[i].pairs = len(pa_), pa_.ctypes.data_as(PI32)
            cl[i].gain, cl[i].minimum_distance = float(l.gain), float(l.minimum_distance_from_collisions)
            cl[i].detection_distance, cl[i].bound_relaxation = float(l.collision_detection_distance), float(l.bound_relaxation)
        self._keep.append(cl)
        pr.n_coll, pr.coll = len(col), cl
        self.collisions = col
        da = (MkoDenseTask * max(1, len(dense_tasks)))()
        for i, t in enumerate(dense_tasks):
            c = np.ascontiguousarray(np.asarray(t["cost"], dtype=np.float64).ravel())
            self._keep.append(c)
            da[i].k, da[i].cost = len(c), c.ctypes.data_as(PF64)
            da[i].gain, da[i].lm_damping = float(t.get("gain", 1.0)), float(t.get("lm_damping", 0.0))
        self._keep.append(da)
        pr.n_dense, pr.dense, pr.n_dense_limit_rows = len(dense_tasks), da, int(dense_limit_rows)
        self.dense_K = sum(int(da[i].k) for i in range(len(dense_tasks)))
        self.cproblem = pr

    def collision_rows(self, q, dt: float, which: int = 0):
        """(G, h) of the `which`-th CollisionAvoidanceLimit at q — every pair a row, h = +inf for an inactive pair."""
        m = self.model
        q = np.ascontiguousarray(q, dtype=np.float64)
        n = self.cproblem.coll[which].n_pairs
        G = np.empty((n, m.nv))
        h = np.empty(n)
        rc = lib().mko_collision_rows(C.byref(self.cmodel), C.byref(self.cproblem.coll[which]), q.ctypes.data_as(PF64),
                                      C.c_double(dt), G.ctypes.data_as(PF64), h.ctypes.data_as(PF64))
        if rc:
            raise RuntimeError(f"mko_collision_rows: {rc}")
        return G, h

    def _targets(self):
        ft = np.array([t.target for t in self.frames], dtype=np.float64).reshape(len(self.frames), 7)
        pt = np.array([t.target_q for t in self.postures], dtype=np.float64).reshape(len(self.postures), self.model.nq)
        ct = np.array([t.target for t in self.coms], dtype=np.float64).reshape(len(self.coms), 3)
        return ft, pt, ct

    def solve(self, q, dt: float, damping: float, return_problem: bool = False):
        """One instance, targets taken from the specs (mirror of oracle.ik.solve_ik)."""
        m = self.model
        ft, pt, ct = self._targets()
        q = np.ascontiguousarray(q, dtype=np.float64)
        v = np.empty(m.nv)
        H = np.empty((m.nv, m.nv))
        c = np.empty(m.nv)
        rc = lib().mko_solve_ik(C.byref(self.cmodel), C.byref(self.cproblem), q.ctypes.data_as(PF64),
                                ft.ctypes.data_as(PF64), pt.ctypes.data_as(PF64), ct.ctypes.data_as(PF64),
                                C.c_double(dt), C.c_double(damping), v.ctypes.data_as(PF64), H.ctypes.data_as(PF64),
                                c.ctypes.data_as(PF64))
        if rc == 4:
            raise ik.qp_gi.NotPositiveDefinite("matrix P is not positive definite")
        if rc:
            raise ik.qp_gi.Infeasible(f"C oracle status {rc}")
        return (v, (H, c)) if return_problem else v

    def solve_batch(self, q, frame_targets, posture_target, dt: float, damping: float, com_target=None,
                    nthreads: int = 1, dense: Optional[dict] = None):
        """q (B, nq), frame_targets (B, n_frame, 7), posture_target (n_posture, nq) or (B, n_posture, nq), com_target
        (n_com, 3) or (B, n_com, 3); dense: {task_e (B, K), task_J (B, K, nv), limit_G (B, M, nv), limit_h (B, M)}."""
        m = self.model
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        ft = np.ascontiguousarray(frame_targets, dtype=np.float64).reshape(B, max(len(self.frames), 0) * 7) \
            if len(self.frames) else np.zeros((B, 1))
        pt = np.ascontiguousarray(posture_target, dtype=np.float64) if len(self.postures) else np.zeros(1)
        batched = 1 if (len(self.postures) and pt.ndim == 3) else 0
        ct = np.ascontiguousarray(com_target, dtype=np.float64) if len(self.coms) else np.zeros(3)
        com_batched = 1 if (len(self.coms) and ct.ndim == 3) else 0
        if com_batched and ct.shape[0] != B:
            raise ValueError("com_target (B, n_com, 3) does not match the batch")
        v = np.empty((B, m.nv))
        st = np.empty(B, dtype=np.int32)
        rows, keep = None, []
        K, M = self.dense_K, int(self.cproblem.n_dense_limit_rows)
        if K or M:
            rows = MkoDenseRows()
            for name, shape in (("task_e", (B, K)), ("task_J", (B, K, m.nv)), ("limit_G", (B, M, m.nv)), ("limit_h", (B, M))):
                a = np.ascontiguousarray(dense[name], dtype=np.float64) if shape[1] else np.zeros(1)
                if shape[1] and a.shape != shape:
                    raise ValueError(f"dense[{name!r}]: expected {shape}, got {a.shape}")
                keep.append(a)
                setattr(rows, name, a.ctypes.data_as(PF64))
        rc = lib().mko_solve_ik_batch_dense(C.byref(self.cmodel), C.byref(self.cproblem), C.c_int32(B), q.ctypes.data_as(PF64),
                                            ft.ctypes.data_as(PF64), pt.ctypes.data_as(PF64), C.c_int32(batched),
                                            ct.ctypes.data_as(PF64), C.c_int32(com_batched),
                                            C.byref(rows) if rows is not None else None, C.c_double(dt), C.c_double(damping),
                                            C.c_int32(nthreads), v.ctypes.data_as(PF64), st.ctypes.data_as(PI32))
        if rc:
            raise RuntimeError(f"mko_solve_ik_batch: {rc}")
        return v, st


def solve_qp(P, q, G=None, h=None):
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n = P.shape[0]
    m = 0 if G is None else len(G)
    Gc = np.ascontiguousarray(G, dtype=np.float64) if m else np.zeros((1, n))
    hc = np.ascontiguousarray(h, dtype=np.float64) if m else np.zeros(1)
    x = np.empty(n)
    rc = lib().mko_solve_qp(C.c_int32(n), C.c_int32(m), P.ctypes.data_as(PF64), q.ctypes.data_as(PF64),
                            Gc.ctypes.data_as(PF64), hc.ctypes.data_as(PF64), x.ctypes.data_as(PF64))
    if rc == 4:
        raise ik.qp_gi.NotPositiveDefinite("matrix P is not positive definite")
    if rc:
        raise ik.qp_gi.Infeasible(f"status {rc}")
    return x
