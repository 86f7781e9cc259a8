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
