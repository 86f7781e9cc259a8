"""Stand-in for the ``mujoco`` wheel: just the symbols /root/reference/mink uses,
backed by oracle/mjmath.py.  See oracle/stubs/README.md."""

from __future__ import annotations

import enum

import numpy as np

from mink_amd.flatmodel import FlatModel
from mink_amd.mjcf import load_mjcf, loads_mjcf
from oracle import mjmath as _mj

mjMAXVAL = _mj.mjMAXVAL
mjMINVAL = _mj.mjMINVAL


class mjtObj(enum.IntEnum):
    mjOBJ_BODY = 1
    mjOBJ_JOINT = 3
    mjOBJ_GEOM = 5
    mjOBJ_SITE = 6
    mjOBJ_KEY = 23


class mjtJoint(enum.IntEnum):
    mjJNT_FREE = 0
    mjJNT_BALL = 1
    mjJNT_SLIDE = 2
    mjJNT_HINGE = 3


class MjModel(FlatModel):
    @staticmethod
    def from_xml_path(path):
        m = load_mjcf(str(path))
        m.__class__ = MjModel
        return m

    @staticmethod
    def from_xml_string(xml):
        m = loads_mjcf(xml)
        m.__class__ = MjModel
        return m


class MjData(_mj.Data):
    def __init__(self, model):
        object.__setattr__(self, "_qpos", np.array(model.qpos0, dtype=np.float64))
        super().__init__(model)

    # real MjData copies into its buffer on assignment (no aliasing of the rhs)
    @property
    def qpos(self):
        return self._qpos

    @qpos.setter
    def qpos(self, value):
        self._qpos[:] = np.asarray(value, dtype=np.float64)


_KIND = {mjtObj.mjOBJ_BODY: "body", mjtObj.mjOBJ_JOINT: "joint", mjtObj.mjOBJ_GEOM: "geom",
         mjtObj.mjOBJ_SITE: "site", mjtObj.mjOBJ_KEY: "key"}


def mj_name2id(m, kind, name):
    return m.name2id(_KIND[mjtObj(kind)], name)


def mj_kinematics(m, d):
    _mj.mj_kinematics(m, d)


def mj_comPos(m, d):
    _mj.mj_comPos(m, d)


def mj_jacBody(m, d, jacp, jacr, body):
    _mj.mj_jacBody(m, d, jacp, jacr, int(body))


def mj_jacGeom(m, d, jacp, jacr, geom):
    _mj.mj_jacGeom(m, d, jacp, jacr, int(geom))


def mj_jacSite(m, d, jacp, jacr, site):
    _mj.mj_jacSite(m, d, jacp, jacr, int(site))


def mj_jac(m, d, jacp, jacr, point, body):
    _mj.mj_jac(m, d, jacp, jacr, point, int(body))


def mj_jacSubtreeCom(m, d, jacp, body):
    _mj.mj_jacSubtreeCom(m, d, jacp, int(body))


def mj_differentiatePos(m, qvel, dt, qpos1, qpos2):
    _mj.mj_differentiatePos(m, qvel, dt, qpos1, qpos2)


def mj_integratePos(m, qpos, qvel, dt):
    _mj.mj_integratePos(m, qpos, qvel, dt)


def mj_geomDistance(m, d, geom1, geom2, distmax, fromto):
    return _mj.mj_geomDistance(m, d, geom1, geom2, distmax, fromto)


def mj_resetData(m, d):
    d.qpos = m.qpos0


def mj_resetDataKeyframe(m, d, key):
    d.qpos = m.key_qpos[key]


def mju_mat2Quat(quat, mat):
    _mj.mju_mat2Quat(quat, np.asarray(mat, dtype=np.float64).ravel())


def mju_quat2Mat(mat, quat):
    _mj.mju_quat2Mat(mat, quat)


def mju_mulQuat(res, a, b):
    _mj.mju_mulQuat(res, a, b)


def mju_normalize3(v):
    return _mj.mju_normalize3(v)


_NAMES = {mjtObj.mjOBJ_BODY: "body_names", mjtObj.mjOBJ_JOINT: "jnt_names", mjtObj.mjOBJ_GEOM: "geom_names",
          mjtObj.mjOBJ_SITE: "site_names", mjtObj.mjOBJ_KEY: "key_names"}


def mj_id2name(m, kind, i):
    """None for unnamed objects, like the real binding."""
    names = getattr(m, "_names")[_NAMES[mjtObj(kind)]] if hasattr(m, "_names") else getattr(m, _NAMES[mjtObj(kind)])
    return names[i] or None


class RawMjModel:
    """A plain attribute bag with the REAL ``mujoco.MjModel`` field names, shapes and dtypes (int32 ids, uint8
    ``jnt_limited``, float64 reals, names only through ``mj_id2name``) — and nothing of FlatModel's own interface —
    so that the production ingest ``FlatModel.from_mjmodel`` can be executed without the wheel
    (tests/test_api_cpu.py::test_from_mjmodel_ingest)."""

    _INT = ("body_parentid", "body_rootid", "body_weldid", "body_mocapid", "body_jntnum", "body_jntadr", "body_dofnum",
            "body_dofadr", "body_geomnum", "body_geomadr", "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid",
            "dof_bodyid", "dof_jntid", "dof_parentid", "site_bodyid", "geom_bodyid", "geom_type", "geom_contype",
            "geom_conaffinity")
    _F64 = ("body_pos", "body_quat", "body_ipos", "body_mass", "body_subtreemass", "jnt_pos", "jnt_axis", "jnt_range",
            "qpos0", "site_pos", "site_quat", "geom_size", "geom_pos", "geom_quat", "key_qpos")

    def __init__(self, flat: FlatModel):
        for n in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite", "nmocap", "nkey"):
            setattr(self, n, int(getattr(flat, n)))
        for n in self._INT:
            setattr(self, n, np.array(getattr(flat, n), dtype=np.int32))
        for n in self._F64:
            setattr(self, n, np.array(getattr(flat, n), dtype=np.float64))
        self.jnt_limited = np.array(flat.jnt_limited, dtype=np.uint8)       # mjtByte in the real struct
        # meshes in the real layout: float32 vertices, per-mesh address / count, the convex-hull graph (here: a graph
        # whose vertex list is the identity — the FlatModel's mesh_vert ARE the hull's vertices — preceded, for the
        # first mesh, by a dummy interior vertex that only the graph can tell apart)
        self.nmesh = int(len(flat.mesh_vertnum))
        self.geom_dataid = np.array(flat.geom_dataid, dtype=np.int32)
        verts, adr, num, graph, gadr = [], [], [], [], []
        for k in range(self.nmesh):
            hull = flat.mesh_vert[int(flat.mesh_vertadr[k]): int(flat.mesh_vertadr[k]) + int(flat.mesh_vertnum[k])]
            extra = 1 if k == 0 else 0
            adr.append(sum(len(v) for v in verts)); num.append(len(hull) + extra)
            if extra:
                verts.append(hull.mean(axis=0, keepdims=True))
            verts.append(hull)
            gadr.append(len(graph))
            nvh = len(hull)
            graph += [nvh, 0] + [0] * nvh + list(range(extra, extra + nvh))
        self.mesh_vertadr = np.array(adr, dtype=np.int32); self.mesh_vertnum = np.array(num, dtype=np.int32)
        self.mesh_vert = (np.concatenate(verts, axis=0) if verts else np.zeros((0, 3))).astype(np.float32)
        self.mesh_graph = np.array(graph, dtype=np.int32); self.mesh_graphadr = np.array(gadr, dtype=np.int32)
        self._names = {k: list(getattr(flat, k)) for k in _NAMES.values()}
