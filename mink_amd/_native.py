"""ctypes binding of libminkhip.so (include/minkhip.h).

This is the whole Python↔device boundary: plain pointers and sizes, no torch types.
numpy arrays are passed as host pointers (the library stages them); torch CUDA
tensors are passed as device pointers with MKH_FLAG_DEVICE_PTRS (asynchronous on the
current torch stream).  There is no CPU fallback: if the library or a GPU is missing
every call raises.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, Optional, Sequence

import numpy as np

from .flatmodel import FlatModel

# (MKH_LIB_TAG: an experiment build made with MKH_BUILD_TAG, mink_amd/csrc/build.py — same-box A/B runs only)
_LIB_TAG = os.environ.get("MKH_LIB_TAG", "")
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                         f"libminkhip_{_LIB_TAG}.so" if _LIB_TAG else "libminkhip.so")

MKH_OK = 0
FLAG_DEVICE_PTRS, FLAG_POSTURE_BATCHED, FLAG_COM_BATCHED, FLAG_DIRECT_QP, FLAG_WAVE_KERNEL, FLAG_LANE_KERNEL = 1, 2, 4, 8, 16, 32
FLAG_TWO_WAVES = 64
FLAG_WARM_START = 128
FLAG_QUAD_KERNEL = 256
FLAG_FULL_ROWS = 512
# per-handle diagnostic switches of mkh_problem_create_diag (include/minkhip.h MKH_DIAG_*): parity tests and measurements only
DIAG_NO_WIDE_REDO, DIAG_NO_TIGHT_REDO, DIAG_NO_COLD_REFINE, DIAG_NO_PAIR_CULL = 1, 2, 4, 8
_diag_default = threading.local()


class diag_options:
    """`with diag_options(DIAG_NO_WIDE_REDO): prob = NativeProblem(...)` — every handle created by THIS thread inside the block
    gets these MKH_DIAG_* bits (tests and bench.py build their problems through helper functions; the switch travels in the
    call, not in the process environment).  A `diag=` argument of NativeProblem wins."""

    def __init__(self, bits: int):
        self.bits = int(bits)

    def __enter__(self):
        self.prev = getattr(_diag_default, "bits", 0)
        _diag_default.bits = self.bits
        return self

    def __exit__(self, *exc):
        _diag_default.bits = self.prev
        return False
ST_OUTSIDE_LIMITS, ST_INFEASIBLE, ST_NOT_PD, ST_ITER_LIMIT, ST_ROW_OVERFLOW = 1, 2, 4, 8, 16
FRAME_TYPE_ID = {"body": 0, "geom": 1, "site": 2}

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


class MkhFlatModel(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite")]
        + [(n, _pi) for n in ("body_parentid", "body_rootid", "body_jntnum", "body_jntadr",
                              "body_dofnum", "body_dofadr")]
        + [(n, _pd) for n in ("body_pos", "body_quat", "body_ipos", "body_mass", "body_subtreemass")]
        + [(n, _pi) for n in ("jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited")]
        + [(n, _pd) for n in ("jnt_pos", "jnt_axis", "jnt_range", "qpos0")]
        + [(n, _pi) for n in ("dof_bodyid", "dof_jntid", "dof_parentid", "site_bodyid")]
        + [(n, _pd) for n in ("site_pos", "site_quat")]
        + [(n, _pi) for n in ("geom_bodyid", "geom_type")]
        + [(n, _pd) for n in ("geom_size", "geom_pos", "geom_quat")]
        + [("nmesh", C.c_int32), ("nmeshvert", C.c_int32)]
        + [(n, _pi) for n in ("geom_dataid", "mesh_vertadr", "mesh_vertnum")]
        + [("mesh_vert", _pd)]
    )


class MkhFrameTaskDesc(C.Structure):
    _fields_ = [("frame_type", C.c_int32), ("frame_id", C.c_int32), ("cost", C.c_double * 6),
                ("gain", C.c_double), ("lm_damping", C.c_double), ("root_type", C.c_int32), ("root_id", C.c_int32)]


class MkhPostureTaskDesc(C.Structure):
    _fields_ = [("cost", _pd), ("gain", C.c_double), ("lm_damping", C.c_double)]


class MkhComTaskDesc(C.Structure):
    _fields_ = [("cost", C.c_double * 3), ("gain", C.c_double), ("lm_damping", C.c_double)]


class MkhConfigurationLimitDesc(C.Structure):
    _fields_ = [("gain", C.c_double), ("lower", _pd), ("upper", _pd), ("n_indices", C.c_int32),
                ("indices", _pi)]


class MkhVelocityLimitDesc(C.Structure):
    _fields_ = [("n_indices", C.c_int32), ("indices", _pi), ("limit", _pd)]


class MkhCollisionLimitDesc(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("geom_id_pairs", _pi), ("gain", C.c_double),
                ("minimum_distance_from_collisions", C.c_double),
                ("collision_detection_distance", C.c_double), ("bound_relaxation", C.c_double)]


class MkhDenseTaskDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("cost", _pd), ("gain", C.c_double), ("lm_damping", C.c_double)]


class MkhDenseRows(C.Structure):
    _fields_ = [("task_e", C.c_void_p), ("task_J", C.c_void_p), ("limit_G", C.c_void_p), ("limit_h", C.c_void_p),
                ("limit_lo", C.c_void_p), ("limit_hi", C.c_void_p)]


class MkhProblemDesc(C.Structure):
    _fields_ = [
        ("n_frame_tasks", C.c_int32), ("frame_tasks", C.POINTER(MkhFrameTaskDesc)),
        ("n_posture_tasks", C.c_int32), ("posture_tasks", C.POINTER(MkhPostureTaskDesc)),
        ("n_com_tasks", C.c_int32), ("com_tasks", C.POINTER(MkhComTaskDesc)),
        ("n_configuration_limits", C.c_int32), ("configuration_limits", C.POINTER(MkhConfigurationLimitDesc)),
        ("n_velocity_limits", C.c_int32), ("velocity_limits", C.POINTER(MkhVelocityLimitDesc)),
        ("n_collision_limits", C.c_int32), ("collision_limits", C.POINTER(MkhCollisionLimitDesc)),
        ("n_dense_tasks", C.c_int32), ("dense_tasks", C.POINTER(MkhDenseTaskDesc)),
        ("n_dense_limit_rows", C.c_int32), ("dense_limit_box", C.c_int32),
    ]


TAP_NAMES = ("xpos", "xquat", "frame_pose", "subtree_com", "task_e", "task_J", "H", "c", "box_lo",
             "box_hi", "coll_G", "coll_h", "qp_iters", "cycles")


class MkhTaps(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TAP_NAMES]


class MinkHipError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load libminkhip.so (built by mink_amd/csrc/build.py).  Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MinkHipError(
            f"{_LIB_PATH} not found: build it with `python -m mink_amd.csrc.build` "
            "(there is no CPU fallback for the solve path)")
    try:
        # PyTorch-ROCm ships its own libamdhip64; load it first so that the process has ONE HIP
        # runtime (loading /opt/rocm's copy first makes a later torch.cuda init fail with
        # "No HIP GPUs are available").  torch is only plumbing here: memory, streams, RCCL.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB_PATH)
    L.mkh_version.restype = C.c_int32
    L.mkh_last_error.restype = C.c_char_p
    L.mkh_device_count.restype = C.c_int32
    L.mkh_model_create.argtypes = [C.POINTER(MkhFlatModel), C.c_int32, C.POINTER(C.c_void_p)]
    L.mkh_model_destroy.argtypes = [C.c_void_p]
    L.mkh_model_destroy.restype = None
    L.mkh_problem_create.argtypes = [C.c_void_p, C.POINTER(MkhProblemDesc), C.c_int32, C.POINTER(C.c_void_p)]
    L.mkh_problem_create_diag.argtypes = [C.c_void_p, C.POINTER(MkhProblemDesc), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.mkh_problem_create_diag.restype = C.c_int32
    L.mkh_problem_destroy.argtypes = [C.c_void_p]
    L.mkh_problem_destroy.restype = None
    L.mkh_problem_num_task_rows.argtypes = [C.c_void_p]
    L.mkh_problem_num_collision_pairs.argtypes = [C.c_void_p]
    L.mkh_problem_last_kernel.argtypes = [C.c_void_p]
    L.mkh_problem_last_kernel.restype = C.c_char_p
    common = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
              C.c_void_p, C.c_void_p]
    L.mkh_solve.argtypes = common + [C.c_int32, C.c_void_p]
    L.mkh_eval.argtypes = common + [C.POINTER(MkhTaps), C.c_int32, C.c_void_p]
    L.mkh_solve_until.argtypes = common[:8] + [C.c_int32, C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int32, C.c_void_p]
    L.mkh_solve_until.restype = C.c_int32
    L.mkh_solve_dense.argtypes = common[:6] + [C.POINTER(MkhDenseRows), C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                               C.POINTER(MkhTaps), C.c_int32, C.c_void_p]
    L.mkh_solve_dense.restype = C.c_int32
    L.mkh_solve_steps.argtypes = common[:8] + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.mkh_integrate.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                C.c_int32, C.c_void_p]
    L.mkh_lie_eval.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.mkh_lie_eval.restype = C.c_int32
    L.mkh_geom_distance_eval.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mkh_geom_distance_eval.restype = C.c_int32
    L.mkh_problem_launch_info.argtypes = [C.c_void_p, C.c_int32] + [C.POINTER(C.c_int32)] * 4
    for f in ("mkh_model_create", "mkh_problem_create", "mkh_problem_num_task_rows",
              "mkh_problem_num_collision_pairs", "mkh_solve", "mkh_eval", "mkh_integrate",
              "mkh_problem_launch_info", "mkh_solve_steps"):
        getattr(L, f).restype = C.c_int32
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "mkh_version", "mkh_last_error", "mkh_device_count", "mkh_model_create", "mkh_model_destroy",
    "mkh_problem_create", "mkh_problem_destroy", "mkh_problem_num_task_rows",
    "mkh_problem_num_collision_pairs", "mkh_solve", "mkh_eval", "mkh_integrate", "mkh_problem_launch_info",
    "mkh_solve_steps", "mkh_problem_last_kernel", "mkh_lie_eval", "mkh_solve_dense", "mkh_solve_until",
    "mkh_geom_distance_eval", "mkh_problem_create_diag",
)

LIE_OPS = {"se3_log": (0, 7, 0, (6,)), "se3_jlog": (1, 7, 0, (6, 6)), "se3_ljacinv": (2, 6, 0, (6, 6)),
           "se3_multiply": (3, 7, 7, (7,)), "se3_inverse": (4, 7, 0, (7,)), "se3_rminus": (5, 7, 7, (6,)),
           "so3_log": (6, 4, 0, (3,)), "so3_matrix": (7, 4, 0, (3, 3)), "se3_apply": (8, 7, 3, (3,))}


def lie_eval(op: str, a, b=None, device: int = 0) -> np.ndarray:
    """The device SO3/SE3 functions of the hot path over a batch of inputs (mkh_lie_eval; host arrays)."""
    code, na, nb, oshape = LIE_OPS[op]
    a = _f64(a).reshape(-1, na)
    n = len(a)
    if nb:
        b = _f64(b).reshape(-1, nb)
        if len(b) != n:
            raise ValueError("a and b must have the same leading dimension")
    out = np.empty((n,) + oshape)
    _check(lib().mkh_lie_eval(int(device), code, n, a.ctypes.data, b.ctypes.data if nb else None, out.ctypes.data, 0, None))
    return out


def geom_distance_eval(type1, size1, pos1, quat1, type2, size2, pos2, quat2, distmax: float, device: int = 0):
    """mj_geomDistance of n primitive geom pairs on the device routines of the collision phase (mkh_geom_distance_eval):
    types (n,), sizes (n, 3), world positions (n, 3), world quaternions wxyz (n, 4) → dist (n,), fromto (n, 6)."""
    t1 = _f64(type1).reshape(-1, 1)
    n = len(t1)
    rec = np.concatenate([t1, _f64(size1).reshape(n, 3), _f64(pos1).reshape(n, 3), _f64(quat1).reshape(n, 4),
                          _f64(type2).reshape(n, 1), _f64(size2).reshape(n, 3), _f64(pos2).reshape(n, 3),
                          _f64(quat2).reshape(n, 4)], axis=1)
    rec = np.ascontiguousarray(rec)
    dist, fromto = np.empty(n), np.empty((n, 6))
    _check(lib().mkh_geom_distance_eval(int(device), n, rec.ctypes.data, float(distmax), dist.ctypes.data, fromto.ctypes.data, None))
    return dist, fromto


def _check(rc: int) -> None:
    if rc != MKH_OK:
        raise MinkHipError(f"libminkhip error {rc}: {lib().mkh_last_error().decode()}")


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _raw_stream(torch, dev) -> int:
    """hipStream_t of torch's current stream on `dev` (the private accessor costs a tenth of building a Stream object)."""
    try:
        return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream(dev).cuda_stream


class NativeModel:
    """Device copy of a FlatModel (mkh_model_create)."""

    def __init__(self, model: FlatModel, device: int = 0):
        self.model = model
        self.device = int(device)
        m = model
        self._keep = {}
        fm = MkhFlatModel()
        for n in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite"):
            setattr(fm, n, int(getattr(m, n)))
        fm.nmesh = int(len(m.mesh_vertnum))
        fm.nmeshvert = int(len(m.mesh_vert))
        for n, ctype in MkhFlatModel._fields_[6:]:
            if n in ("nmesh", "nmeshvert"):
                continue
            arr = getattr(m, n)
            arr = _i32(arr) if ctype is _pi else _f64(arr)
            if arr.size == 0:
                arr = np.zeros(1, dtype=arr.dtype)
            self._keep[n] = arr
            setattr(fm, n, arr.ctypes.data_as(ctype))
        h = C.c_void_p()
        _check(lib().mkh_model_create(C.byref(fm), self.device, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            lib().mkh_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def integrate(self, q, v, dt: float, out=None):
        """Configuration.integrate for a batch (mkh_integrate)."""
        if _is_torch(q):
            import torch
            q = q.contiguous(); v = v.contiguous()
            out = torch.empty_like(q) if out is None else out
            stream = torch.cuda.current_stream(q.device).cuda_stream
            _check(lib().mkh_integrate(self.handle, q.shape[0], q.data_ptr(), v.data_ptr(), float(dt),
                                       out.data_ptr(), FLAG_DEVICE_PTRS, stream))
            return out
        q = _f64(q); v = _f64(v)
        out = np.empty_like(q)
        _check(lib().mkh_integrate(self.handle, q.shape[0], q.ctypes.data, v.ctypes.data, float(dt),
                                   out.ctypes.data, 0, None))
        return out


class NativeProblem:
    """Device descriptor of one solve_ik call site (mkh_problem_create)."""

    def __init__(self, nmodel: NativeModel, frame_tasks: Sequence[dict] = (), posture_tasks: Sequence[dict] = (),
                 com_tasks: Sequence[dict] = (), configuration_limits: Sequence[dict] = (),
                 velocity_limits: Sequence[dict] = (), collision_limits: Sequence[dict] = (),
                 max_batch: int = 1, dense_tasks: Sequence[dict] = (), dense_limit_rows: int = 0,
                 dense_limit_box: bool = False, diag: Optional[int] = None):
        self.nmodel = nmodel
        m = nmodel.model
        keep = []
        d = MkhProblemDesc()

        def arr(ctype_struct, items):
            a = (ctype_struct * max(1, len(items)))()
            keep.append(a)
            return a

        ft = arr(MkhFrameTaskDesc, frame_tasks)
        for i, t in enumerate(frame_tasks):
            ft[i].frame_type = FRAME_TYPE_ID[t["frame_type"]]
            ft[i].frame_id = int(t["frame_id"])
            ft[i].cost = (C.c_double * 6)(*[float(x) for x in t["cost"]])
            ft[i].gain = float(t.get("gain", 1.0)); ft[i].lm_damping = float(t.get("lm_damping", 0.0))
            ft[i].root_type = FRAME_TYPE_ID[t["root_type"]] if t.get("root_type") is not None else -1
            ft[i].root_id = int(t.get("root_id", 0))
        pt = arr(MkhPostureTaskDesc, posture_tasks)
        for i, t in enumerate(posture_tasks):
            c = _f64(np.broadcast_to(t["cost"], (m.nv,))); keep.append(c)
            pt[i].cost = c.ctypes.data_as(_pd)
            pt[i].gain = float(t.get("gain", 1.0)); pt[i].lm_damping = float(t.get("lm_damping", 0.0))
        ct = arr(MkhComTaskDesc, com_tasks)
        for i, t in enumerate(com_tasks):
            ct[i].cost = (C.c_double * 3)(*[float(x) for x in np.broadcast_to(t["cost"], (3,))])
            ct[i].gain = float(t.get("gain", 1.0)); ct[i].lm_damping = float(t.get("lm_damping", 0.0))
        cl = arr(MkhConfigurationLimitDesc, configuration_limits)
        for i, t in enumerate(configuration_limits):
            lo, up, idx = _f64(t["lower"]), _f64(t["upper"]), _i32(t["indices"])
            keep += [lo, up, idx]
            cl[i].gain = float(t["gain"]); cl[i].lower = lo.ctypes.data_as(_pd); cl[i].upper = up.ctypes.data_as(_pd)
            cl[i].n_indices = len(idx); cl[i].indices = idx.ctypes.data_as(_pi)
        vl = arr(MkhVelocityLimitDesc, velocity_limits)
        for i, t in enumerate(velocity_limits):
            idx, lim = _i32(t["indices"]), _f64(t["limit"])
            keep += [idx, lim]
            vl[i].n_indices = len(idx); vl[i].indices = idx.ctypes.data_as(_pi); vl[i].limit = lim.ctypes.data_as(_pd)
        co = arr(MkhCollisionLimitDesc, collision_limits)
        for i, t in enumerate(collision_limits):
            pairs = _i32(np.asarray(t["geom_id_pairs"]).reshape(-1, 2)); keep.append(pairs)
            co[i].n_pairs = len(pairs); co[i].geom_id_pairs = pairs.ctypes.data_as(_pi)
            co[i].gain = float(t["gain"])
            co[i].minimum_distance_from_collisions = float(t["minimum_distance_from_collisions"])
            co[i].collision_detection_distance = float(t["collision_detection_distance"])
            co[i].bound_relaxation = float(t["bound_relaxation"])
        d.n_frame_tasks, d.frame_tasks = len(frame_tasks), ft
        d.n_posture_tasks, d.posture_tasks = len(posture_tasks), pt
        d.n_com_tasks, d.com_tasks = len(com_tasks), ct
        d.n_configuration_limits, d.configuration_limits = len(configuration_limits), cl
        d.n_velocity_limits, d.velocity_limits = len(velocity_limits), vl
        d.n_collision_limits, d.collision_limits = len(collision_limits), co
        dn = arr(MkhDenseTaskDesc, dense_tasks)
        for i, t in enumerate(dense_tasks):
            c = _f64(np.atleast_1d(t["cost"])); keep.append(c)
            dn[i].k = len(c); dn[i].cost = c.ctypes.data_as(_pd)
            dn[i].gain = float(t.get("gain", 1.0)); dn[i].lm_damping = float(t.get("lm_damping", 0.0))
        d.n_dense_tasks, d.dense_tasks = len(dense_tasks), dn
        d.n_dense_limit_rows = int(dense_limit_rows)
        d.dense_limit_box = 1 if dense_limit_box else 0
        self.dense_limit_box = bool(dense_limit_box)
        self.n_dense_rows = int(sum(len(np.atleast_1d(t["cost"])) for t in dense_tasks))
        self.n_dense_limit_rows = int(dense_limit_rows)
        h = C.c_void_p()
        # (diag: DIAG_* bits — which launches stand behind this handle; 0 = the product's own choice, mkh_problem_create)
        if diag is None:
            diag = getattr(_diag_default, "bits", 0)
        self.diag = int(diag)
        if diag:
            _check(lib().mkh_problem_create_diag(nmodel.handle, C.byref(d), int(max_batch), int(diag), C.byref(h)))
        else:
            _check(lib().mkh_problem_create(nmodel.handle, C.byref(d), int(max_batch), C.byref(h)))
        self.handle = h
        # one in-flight call per handle (it owns the staging buffers and the ticket counter): host-pointer calls are
        # synchronous, so a lock makes them safe from several threads; device-pointer calls are asynchronous and must
        # be ordered by the caller (same stream or events) — include/minkhip.h
        self._lock = threading.Lock()
        self.max_batch = int(max_batch)
        self.n_frame, self.n_posture, self.n_com = len(frame_tasks), len(posture_tasks), len(com_tasks)
        self.n_rows = lib().mkh_problem_num_task_rows(h)
        self.n_pairs = lib().mkh_problem_num_collision_pairs(h)

    def close(self):
        if getattr(self, "handle", None):
            lib().mkh_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_kernel(self) -> str:
        """Kernel variant launched by the last solve on this handle (diagnostic)."""
        return lib().mkh_problem_last_kernel(self.handle).decode()

    def launch_info(self, B: int) -> Dict[str, int]:
        g, b, l, t = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().mkh_problem_launch_info(self.handle, int(B), C.byref(g), C.byref(b), C.byref(l), C.byref(t)))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value, "tableau_rows": t.value}

    # ------------------------------------------------------------------ solve
    def _tap_shapes(self, B: int) -> Dict[str, tuple]:
        m = self.nmodel.model
        return {
            "xpos": (B, m.nbody, 3), "xquat": (B, m.nbody, 4), "frame_pose": (B, self.n_frame, 7),
            "subtree_com": (B, 3), "task_e": (B, self.n_rows), "task_J": (B, self.n_rows, m.nv),
            "H": (B, m.nv, m.nv), "c": (B, m.nv), "box_lo": (B, m.nv), "box_hi": (B, m.nv),
            "coll_G": (B, self.n_pairs, m.nv), "coll_h": (B, self.n_pairs), "qp_iters": (B,),
            "cycles": (B, 16),
        }

    def solve(self, q, frame_targets=None, posture_target=None, com_target=None, dt: float = 1e-2,
              damping: float = 1e-12, taps: Sequence[str] = (), solve_qp: bool = True,
              out=None, status_out=None, n_steps: Optional[int] = None, q_out=None, direct_qp: bool = False,
              dense: Optional[dict] = None, until: Optional[tuple] = None, wave_kernel: bool = False, lane_kernel: bool = False,
              two_waves: bool = False, warm_start: bool = False, quad_kernel: bool = False, full_rows: bool = False):
        """Returns (v, status[, taps dict]).  numpy in → numpy out (synchronous);
        torch CUDA tensors in → torch tensors out (asynchronous on the current stream).
        `until` = (pos_threshold, ori_threshold) with n_steps = max_iters: the threshold-terminated loop
        (mkh_solve_until), returns (q_final, v_last, status, iters, converged).
        `dense`: the plugin rows of a problem created with dense_tasks / dense_limit_rows —
        {"task_e": (B, K), "task_J": (B, K, nv), "limit_G": (B, M, nv), "limit_h": (B, M)} (mkh_solve_dense)."""
        with self._lock:
            return self._solve(q, frame_targets, posture_target, com_target, dt, damping, taps, solve_qp, out,
                               status_out, n_steps, q_out, direct_qp, dense, until, wave_kernel, lane_kernel, two_waves, warm_start, quad_kernel, full_rows)

    def _solve(self, q, frame_targets, posture_target, com_target, dt, damping, taps, solve_qp, out, status_out,
               n_steps, q_out, direct_qp, dense=None, until=None, wave_kernel=False, lane_kernel=False, two_waves=False,
               warm_start=False, quad_kernel=False, full_rows=False):
        m = self.nmodel.model
        use_torch = _is_torch(q)
        B = int(q.shape[0])
        if B > self.max_batch:      # the handle's per-instance state (staging, warm-start sets) is sized by max_batch
            raise MinkHipError(f"B={B} exceeds max_batch={self.max_batch} of this problem")
        flags = (FLAG_DIRECT_QP if direct_qp else 0) | (FLAG_WAVE_KERNEL if wave_kernel else 0) | \
            (FLAG_LANE_KERNEL if lane_kernel else 0) | (FLAG_TWO_WAVES if two_waves else 0) | \
            (FLAG_WARM_START if warm_start else 0) | (FLAG_QUAD_KERNEL if quad_kernel else 0) | \
            (FLAG_FULL_ROWS if full_rows else 0)

        def tgt(x, per, name):
            nonlocal flags
            if x is None:
                return None
            return x

        if use_torch:
            import torch
            dev = q.device

            f64 = torch.float64

            def prep(x):     # (the common case — already float64, on the device, contiguous — must cost nothing: a UR5e
                #              solve of 4 096 instances is a 19 µs kernel, the host side of this call has to stay below it)
                if x is None or (x.dtype is f64 and x.device == dev and x.is_contiguous()):
                    return x
                return x.to(device=dev, dtype=f64).contiguous()

            q = prep(q); frame_targets = prep(frame_targets); posture_target = prep(posture_target)
            com_target = prep(com_target)
            ptr = lambda x: 0 if x is None else x.data_ptr()
            flags |= FLAG_DEVICE_PTRS
            v = (torch.empty((B, m.nv), dtype=torch.float64, device=dev) if out is None else out) if solve_qp else None
            st = (torch.empty((B,), dtype=torch.int32, device=dev) if status_out is None else status_out) if solve_qp else None
            stream = _raw_stream(torch, dev)
            tapbufs = {}
            shapes = self._tap_shapes(B) if taps else None
            for n in taps:
                dt_ = torch.int32 if n == "qp_iters" else (torch.int64 if n == "cycles" else torch.float64)
                tapbufs[n] = torch.empty(shapes[n], dtype=dt_, device=dev)
                if n in ("task_e", "task_J", "subtree_com"):
                    tapbufs[n].zero_()
        else:
            q = _f64(q)
            frame_targets = None if frame_targets is None else _f64(frame_targets)
            posture_target = None if posture_target is None else _f64(posture_target)
            com_target = None if com_target is None else _f64(com_target)
            ptr = lambda x: None if x is None else x.ctypes.data
            v = (np.empty((B, m.nv)) if out is None else out) if solve_qp else None
            st = (np.zeros((B,), dtype=np.int32) if status_out is None else status_out) if solve_qp else None
            stream = None
            shapes = self._tap_shapes(B)
            tapbufs = {n: np.zeros(shapes[n], dtype=np.int32 if n == "qp_iters" else (np.int64 if n == "cycles" else np.float64)) for n in taps}
        if q.shape != (B, m.nq):
            raise ValueError(f"q must have shape (B, {m.nq}), got {tuple(q.shape)}")
        if self.n_frame and (frame_targets is None or tuple(frame_targets.shape) != (B, self.n_frame, 7)):
            raise ValueError(f"frame_targets must have shape ({B}, {self.n_frame}, 7)")
        if self.n_posture:
            if posture_target is None:
                raise ValueError("posture_target is required")
            if tuple(posture_target.shape) == (B, self.n_posture, m.nq):
                flags |= FLAG_POSTURE_BATCHED
            elif tuple(posture_target.shape) != (self.n_posture, m.nq):
                raise ValueError(f"posture_target must have shape ({self.n_posture}, {m.nq}) or (B, ...)")
        if self.n_com:
            if com_target is None:
                raise ValueError("com_target is required")
            if tuple(com_target.shape) == (B, self.n_com, 3):
                flags |= FLAG_COM_BATCHED
            elif tuple(com_target.shape) != (self.n_com, 3):
                raise ValueError(f"com_target must have shape ({self.n_com}, 3) or (B, ...)")
        args = [self.handle, B, ptr(q), ptr(frame_targets), ptr(posture_target), ptr(com_target), float(dt),
                float(damping), ptr(v), ptr(st)]
        if self.n_dense_rows or self.n_dense_limit_rows or self.dense_limit_box:
            if n_steps is not None:
                raise ValueError("dense (plugin) rows are evaluated by the caller at q: no fused steps")
            dense = dense or {}
            shapes_d = {"task_e": (B, self.n_dense_rows), "task_J": (B, self.n_dense_rows, m.nv),
                        "limit_G": (B, self.n_dense_limit_rows, m.nv), "limit_h": (B, self.n_dense_limit_rows)}
            if self.dense_limit_box:
                shapes_d.update({"limit_lo": (B, m.nv), "limit_hi": (B, m.nv)})
            dr, keep_d = MkhDenseRows(), []
            for name, shp in shapes_d.items():
                if 0 in shp:
                    continue
                x = dense.get(name)
                if x is None and name in ("limit_lo", "limit_hi"):
                    continue                                   # one-sided box rows
                if x is None or tuple(x.shape) != shp:
                    raise ValueError(f"dense['{name}'] must have shape {shp}")
                x = prep(x) if use_torch else _f64(x)
                keep_d.append(x)
                setattr(dr, name, x.data_ptr() if use_torch else x.ctypes.data)
            tp = None
            if taps or not solve_qp:
                tp = MkhTaps()
                for n in taps:
                    setattr(tp, n, tapbufs[n].data_ptr() if use_torch else tapbufs[n].ctypes.data)
            _check(lib().mkh_solve_dense(*args[:6], C.byref(dr), float(dt), float(damping), ptr(v), ptr(st),
                                         C.byref(tp) if tp is not None else None, flags, stream))
            return (v, st, tapbufs) if tp is not None else (v, st)
        if n_steps is not None:
            # fused (solve, integrate) x n_steps on the device: returns (q_final, v_last, status)
            if use_torch:
                import torch
                qo = torch.empty_like(q) if q_out is None else q_out
            else:
                qo = np.empty_like(q) if q_out is None else q_out
            if until is not None:
                if use_torch:
                    it = torch.empty((B,), dtype=torch.int32, device=dev); cv = torch.empty_like(it)
                else:
                    it = np.zeros((B,), dtype=np.int32); cv = np.zeros((B,), dtype=np.int32)
                _check(lib().mkh_solve_until(*args[:8], int(n_steps), float(until[0]), float(until[1]), ptr(qo), ptr(v),
                                             ptr(st), ptr(it), ptr(cv), flags, stream))
                return qo, v, st, it, cv
            _check(lib().mkh_solve_steps(*args[:8], int(n_steps), ptr(qo), ptr(v), ptr(st), flags, stream))
            return qo, v, st
        if taps or not solve_qp:
            tp = MkhTaps()
            for n in taps:
                setattr(tp, n, tapbufs[n].data_ptr() if use_torch else tapbufs[n].ctypes.data)
            _check(lib().mkh_eval(*args, C.byref(tp), flags, stream))
            if "qp_iters" in tapbufs:
                # packed by the kernel: active-set selections | loop iterations << 10 | rank-1 pivots << 20
                raw = tapbufs["qp_iters"]
                tapbufs["qp_loops"] = (raw >> 10) & 1023
                tapbufs["qp_pivots"] = (raw >> 20) & 1023
                tapbufs["qp_iters"] = raw & 1023
            return v, st, tapbufs
        _check(lib().mkh_solve(*args, flags, stream))
        return v, st
