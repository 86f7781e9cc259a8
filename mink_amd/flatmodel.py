"""FlatModel: the one-time flattened copy of the mjModel kinematic tree.

This is the host-side snapshot of every ``mujoco.MjModel`` field the reference's
hot path reads (call sites: mink/configuration.py:53-155,
mink/limits/configuration_limit.py:41-67, mink/limits/velocity_limit.py:45-69,
mink/tasks/posture_task.py:44, mink/limits/collision_avoidance_limit.py:75-115).
Field names and semantics follow MuJoCo's ``mjModel`` so that
``FlatModel.from_mjmodel(m)`` is a field-for-field copy when ``mujoco`` is
importable, and :mod:`mink_amd.mjcf` builds the same arrays from MJCF when it is
not.  Nothing here runs per solve: the device library receives these arrays once
at ``mkh_model_create``.
"""

from __future__ import annotations

import json
from dataclasses import dataclass, field, fields
from types import SimpleNamespace
from typing import Dict, List

import numpy as np

# mjtJoint (mink/constants.py:27-34 uses the integer values).
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
# mjtGeom
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE = 0, 1, 2, 3
GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 4, 5, 6, 7
GEOM_TYPE_NAMES = {
    "plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE,
    "capsule": GEOM_CAPSULE, "ellipsoid": GEOM_ELLIPSOID,
    "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH,
}
mjMINVAL = 1e-15
mjMAXVAL = 1e10

_QPOS_WIDTH = {JNT_FREE: 7, JNT_BALL: 4, JNT_SLIDE: 1, JNT_HINGE: 1}
_DOF_WIDTH = {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}


def qpos_width(jnt_type: int) -> int:
    return _QPOS_WIDTH[int(jnt_type)]


def dof_width(jnt_type: int) -> int:
    return _DOF_WIDTH[int(jnt_type)]


_INT_FIELDS = (
    "body_parentid", "body_rootid", "body_weldid", "body_mocapid", "body_jntnum",
    "body_jntadr", "body_dofnum", "body_dofadr", "body_geomnum", "body_geomadr",
    "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited",
    "dof_bodyid", "dof_jntid", "dof_parentid", "site_bodyid", "geom_bodyid",
    "geom_type", "geom_contype", "geom_conaffinity", "geom_valid", "body_mass_valid",
    "geom_dataid", "mesh_vertadr", "mesh_vertnum",
)
_F64_FIELDS = (
    "body_pos", "body_quat", "body_ipos", "body_mass", "body_subtreemass",
    "jnt_pos", "jnt_axis", "jnt_range", "qpos0", "site_pos", "site_quat",
    "geom_size", "geom_pos", "geom_quat", "key_qpos", "mocap_pos", "mocap_quat", "mesh_vert",
)
_NAME_FIELDS = ("body_names", "jnt_names", "site_names", "geom_names", "key_names")


class _Named(SimpleNamespace):
    """Tiny stand-in for mujoco's named accessors (``model.joint(name).id``)."""


@dataclass(eq=False)   # identity hash: models are cache keys (device handles, compiled problems)
class FlatModel:
    nq: int = 0
    nv: int = 0
    nbody: int = 0
    njnt: int = 0
    ngeom: int = 0
    nsite: int = 0
    nmocap: int = 0
    nkey: int = 0
    # bodies
    body_parentid: np.ndarray = None
    body_rootid: np.ndarray = None
    body_weldid: np.ndarray = None
    body_mocapid: np.ndarray = None
    body_jntnum: np.ndarray = None
    body_jntadr: np.ndarray = None
    body_dofnum: np.ndarray = None
    body_dofadr: np.ndarray = None
    body_geomnum: np.ndarray = None
    body_geomadr: np.ndarray = None
    body_pos: np.ndarray = None
    body_quat: np.ndarray = None
    body_ipos: np.ndarray = None
    body_mass: np.ndarray = None
    body_subtreemass: np.ndarray = None
    body_mass_valid: np.ndarray = None  # 0 for bodies whose mass / inertial frame needs mesh data (no <inertial>)
    # joints / dofs
    jnt_type: np.ndarray = None
    jnt_qposadr: np.ndarray = None
    jnt_dofadr: np.ndarray = None
    jnt_bodyid: np.ndarray = None
    jnt_limited: np.ndarray = None
    jnt_pos: np.ndarray = None
    jnt_axis: np.ndarray = None
    jnt_range: np.ndarray = None
    dof_bodyid: np.ndarray = None
    dof_jntid: np.ndarray = None
    dof_parentid: np.ndarray = None
    qpos0: np.ndarray = None
    # sites / geoms
    site_bodyid: np.ndarray = None
    site_pos: np.ndarray = None
    site_quat: np.ndarray = None
    geom_bodyid: np.ndarray = None
    geom_type: np.ndarray = None
    geom_contype: np.ndarray = None
    geom_conaffinity: np.ndarray = None
    geom_valid: np.ndarray = None  # 0 for geoms whose local frame/size need mesh data
    geom_size: np.ndarray = None
    geom_pos: np.ndarray = None
    geom_quat: np.ndarray = None
    # mesh geoms (MuJoCo's names; what the collision path reads of a mesh is its CONVEX HULL — mj_geomDistance collides
    # the hull): geom_dataid = mesh of a type-mesh geom (−1 else); hull vertices of mesh k in the geom frame (MuJoCo
    # re-expresses a mesh in its inertial frame and folds that frame into geom_pos / geom_quat) are
    # mesh_vert[mesh_vertadr[k] : + mesh_vertnum[k]].  Here mesh_vert holds the hull's vertices only, as float64.
    geom_dataid: np.ndarray = None
    mesh_vertadr: np.ndarray = None
    mesh_vertnum: np.ndarray = None
    mesh_vert: np.ndarray = None
    # keyframes / mocap
    key_qpos: np.ndarray = None
    mocap_pos: np.ndarray = None
    mocap_quat: np.ndarray = None
    # names
    body_names: List[str] = field(default_factory=list)
    jnt_names: List[str] = field(default_factory=list)
    site_names: List[str] = field(default_factory=list)
    geom_names: List[str] = field(default_factory=list)
    key_names: List[str] = field(default_factory=list)

    # ------------------------------------------------------------------ setup
    def finalize(self) -> "FlatModel":
        if self.body_mass_valid is None:        # (models serialised before the field existed; real MjModels)
            self.body_mass_valid = np.ones(self.nbody, dtype=np.int32)
        if self.geom_dataid is None:            # (models serialised before the mesh fields existed)
            self.geom_dataid = -np.ones(self.ngeom, dtype=np.int32)
        if self.mesh_vertadr is None:
            self.mesh_vertadr, self.mesh_vertnum, self.mesh_vert = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 3))
        for f in _INT_FIELDS:
            setattr(self, f, np.ascontiguousarray(getattr(self, f), dtype=np.int32))
        for f in _F64_FIELDS:
            setattr(self, f, np.ascontiguousarray(getattr(self, f), dtype=np.float64))
        shapes = {
            "body_pos": (self.nbody, 3), "body_quat": (self.nbody, 4),
            "body_ipos": (self.nbody, 3), "jnt_pos": (self.njnt, 3),
            "jnt_axis": (self.njnt, 3), "jnt_range": (self.njnt, 2),
            "site_pos": (self.nsite, 3), "site_quat": (self.nsite, 4),
            "geom_size": (self.ngeom, 3), "geom_pos": (self.ngeom, 3),
            "geom_quat": (self.ngeom, 4), "key_qpos": (self.nkey, self.nq),
            "mocap_pos": (self.nmocap, 3), "mocap_quat": (self.nmocap, 4), "mesh_vert": (-1, 3),
        }
        for k, shp in shapes.items():
            setattr(self, k, getattr(self, k).reshape(shp))
        # hull vertices carry the precision of the compiled model (mjModel.mesh_vert is float32): a value that went
        # through the short decimals of to_json comes back as exactly the same float32
        self.mesh_vert = np.ascontiguousarray(self.mesh_vert.astype(np.float32).astype(np.float64))
        self._name_maps: Dict[str, Dict[str, int]] = {
            "body": {n: i for i, n in enumerate(self.body_names) if n},
            "joint": {n: i for i, n in enumerate(self.jnt_names) if n},
            "site": {n: i for i, n in enumerate(self.site_names) if n},
            "geom": {n: i for i, n in enumerate(self.geom_names) if n},
            "key": {n: i for i, n in enumerate(self.key_names) if n},
        }
        # derived: depth of each body and ancestor-dof bit masks (used by the device
        # descriptor and by the oracle's Jacobian walk).
        depth = np.zeros(self.nbody, dtype=np.int32)
        for b in range(1, self.nbody):
            depth[b] = depth[self.body_parentid[b]] + 1
        self.body_depth = depth
        return self

    # ---------------------------------------------------------- named lookups
    def name2id(self, kind: str, name: str) -> int:
        """``mj_name2id`` semantics: -1 when absent (mink/configuration.py:72,133)."""
        return self._name_maps[kind].get(name, -1)

    def _named(self, kind: str, key, names, extra=None) -> _Named:
        if isinstance(key, str):
            idx = self.name2id(kind, key)
            if idx < 0:
                raise KeyError(
                    f"Invalid name '{key}'. Valid names: {[n for n in names if n]}"
                )
        else:
            idx = int(key)
            if not 0 <= idx < len(names):
                raise IndexError(f"Invalid index {idx}")
        ns = _Named(id=idx, name=names[idx])
        if extra:
            for k, v in extra(idx).items():
                setattr(ns, k, v)
        return ns

    def joint(self, key) -> _Named:
        return self._named("joint", key, self.jnt_names)

    def body(self, key) -> _Named:
        return self._named(
            "body", key, self.body_names,
            lambda i: {"mocapid": np.array([self.body_mocapid[i]], dtype=np.int32)},
        )

    def site(self, key) -> _Named:
        return self._named("site", key, self.site_names)

    def geom(self, key) -> _Named:
        return self._named("geom", key, self.geom_names)

    def key(self, key) -> _Named:
        return self._named("key", key, self.key_names)

    # ---------------------------------------------------------- derived masks
    def dof_ancestor_mask(self, body_id: int) -> int:
        """Bit i set iff dof i lies on the chain world→``body_id`` (mj_jac's walk)."""
        mask = 0
        b = int(body_id)
        while b > 0 and self.body_dofnum[b] == 0:
            b = int(self.body_parentid[b])
        if b == 0:
            return 0
        i = int(self.body_dofadr[b] + self.body_dofnum[b] - 1)
        while i >= 0:
            mask |= 1 << i
            i = int(self.dof_parentid[i])
        return mask

    def require_valid_masses(self, what: str) -> None:
        """Mass-dependent quantities (subtree CoM, ComTask) of a model whose MJCF leaves a body's inertia to its
        mesh geoms cannot be computed without the mesh assets: fail instead of using mass 0."""
        bad = [self.body_names[b] or f"#{b}" for b in range(1, self.nbody)
               if not self.body_mass_valid[b] and self.body_rootid[b] == 1]
        if bad:
            raise ValueError(f"{what} needs body masses, but {bad[:4]}{' ...' if len(bad) > 4 else ''} have no "
                             "<inertial> and mesh geoms (mesh-derived inertia is not available to the MJCF subset "
                             "reader): give those bodies an <inertial> element or ingest a compiled mujoco.MjModel")

    def freejoint_dims(self):
        """mink/utils.py:38-56."""
        q_ids, v_ids = [], []
        for j in range(self.njnt):
            if self.jnt_type[j] == JNT_FREE:
                qa, va = int(self.jnt_qposadr[j]), int(self.jnt_dofadr[j])
                q_ids.extend(range(qa, qa + 7))
                v_ids.extend(range(va, va + 6))
        return q_ids, v_ids

    # ------------------------------------------------------------ (de)serialise
    def to_json(self) -> str:
        out = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if f.name == "mesh_vert":      # float32 values: their shortest decimals (a third of the file size of float64 reprs)
                v = [[float(str(x)) for x in row] for row in np.asarray(v, dtype=np.float32)]
            out[f.name] = v.tolist() if isinstance(v, np.ndarray) else v
        return json.dumps(out)

    @classmethod
    def from_json(cls, text: str) -> "FlatModel":
        d = json.loads(text)
        m = cls(**d)
        return m.finalize()

    def save(self, path: str) -> None:
        with open(path, "w") as fh:
            fh.write(self.to_json())

    @classmethod
    def load(cls, path: str) -> "FlatModel":
        with open(path) as fh:
            return cls.from_json(fh.read())

    # ------------------------------------------------------ real mujoco ingest
    @staticmethod
    def _hulls_of_mjmodel(m) -> dict:
        """Convex-hull vertices of the meshes that type-mesh geoms refer to, from the compiled model: `mesh_graph` lists
        the hull's vertices (layout at mesh_graphadr[k]: numvert, numface, vert_edgeadr[numvert], vert_globalid[numvert],
        …); a mesh without a graph is collided with all of its vertices."""
        ngeom = int(m.ngeom)
        dataid = -np.ones(ngeom, dtype=np.int32)
        if not hasattr(m, "mesh_vert") or int(getattr(m, "nmesh", 0)) == 0:
            return dict(geom_dataid=dataid, mesh_vertadr=np.zeros(0, np.int32), mesh_vertnum=np.zeros(0, np.int32),
                        mesh_vert=np.zeros((0, 3)))
        verts_all = np.asarray(m.mesh_vert, dtype=np.float64).reshape(-1, 3)
        graph = np.asarray(getattr(m, "mesh_graph", np.zeros(0, np.int32)), dtype=np.int64).reshape(-1)
        remap, hulls = {}, []
        for g in range(ngeom):
            k = int(m.geom_dataid[g])
            if int(m.geom_type[g]) != GEOM_MESH or k < 0:
                continue
            if k not in remap:
                v = verts_all[int(m.mesh_vertadr[k]): int(m.mesh_vertadr[k]) + int(m.mesh_vertnum[k])]
                ga = int(m.mesh_graphadr[k]) if hasattr(m, "mesh_graphadr") else -1
                if ga >= 0:
                    nvh = int(graph[ga])
                    v = v[graph[ga + 2 + nvh: ga + 2 + 2 * nvh]]
                remap[k] = len(hulls)
                hulls.append(np.ascontiguousarray(v))
            dataid[g] = remap[k]
        num = np.array([len(h) for h in hulls], dtype=np.int32)
        return dict(geom_dataid=dataid, mesh_vertadr=(np.cumsum(num) - num).astype(np.int32), mesh_vertnum=num,
                    mesh_vert=np.concatenate(hulls, axis=0) if hulls else np.zeros((0, 3)))

    def mesh_hull(self, geom_id: int) -> np.ndarray:
        """Hull vertices (n, 3) of a type-mesh geom in the geom's frame."""
        k = int(self.geom_dataid[geom_id])
        if k < 0:
            raise ValueError(f"geom {geom_id} is not a mesh geom (or its mesh asset was not available)")
        a = int(self.mesh_vertadr[k])
        return self.mesh_vert[a: a + int(self.mesh_vertnum[k])]

    @classmethod
    def from_mjmodel(cls, m) -> "FlatModel":
        """Field-for-field copy of a real ``mujoco.MjModel`` (production ingest)."""
        import mujoco  # noqa: F401  (only reachable when mujoco is installed)

        def names(kind, n):
            return [mujoco.mj_id2name(m, kind, i) or "" for i in range(n)]

        fm = cls(
            nq=m.nq, nv=m.nv, nbody=m.nbody, njnt=m.njnt, ngeom=m.ngeom,
            nsite=m.nsite, nmocap=m.nmocap, nkey=m.nkey,
            body_parentid=m.body_parentid, body_rootid=m.body_rootid,
            body_weldid=m.body_weldid, body_mocapid=m.body_mocapid,
            body_jntnum=m.body_jntnum, body_jntadr=m.body_jntadr,
            body_dofnum=m.body_dofnum, body_dofadr=m.body_dofadr,
            body_geomnum=m.body_geomnum, body_geomadr=m.body_geomadr,
            body_pos=m.body_pos, body_quat=m.body_quat, body_ipos=m.body_ipos,
            body_mass=m.body_mass, body_subtreemass=m.body_subtreemass,
            jnt_type=m.jnt_type, jnt_qposadr=m.jnt_qposadr, jnt_dofadr=m.jnt_dofadr,
            jnt_bodyid=m.jnt_bodyid, jnt_limited=m.jnt_limited, jnt_pos=m.jnt_pos,
            jnt_axis=m.jnt_axis, jnt_range=m.jnt_range, dof_bodyid=m.dof_bodyid,
            dof_jntid=m.dof_jntid, dof_parentid=m.dof_parentid, qpos0=m.qpos0,
            site_bodyid=m.site_bodyid, site_pos=m.site_pos, site_quat=m.site_quat,
            geom_bodyid=m.geom_bodyid, geom_type=m.geom_type,
            geom_contype=m.geom_contype, geom_conaffinity=m.geom_conaffinity,
            geom_valid=np.ones(m.ngeom, dtype=np.int32),
            geom_size=m.geom_size, geom_pos=m.geom_pos, geom_quat=m.geom_quat,
            **cls._hulls_of_mjmodel(m),
            key_qpos=m.key_qpos,
            mocap_pos=np.array([m.body_pos[b] for b in range(m.nbody) if m.body_mocapid[b] >= 0]).reshape(-1, 3),
            mocap_quat=np.array([m.body_quat[b] for b in range(m.nbody) if m.body_mocapid[b] >= 0]).reshape(-1, 4),
            body_names=names(mujoco.mjtObj.mjOBJ_BODY, m.nbody),
            jnt_names=names(mujoco.mjtObj.mjOBJ_JOINT, m.njnt),
            site_names=names(mujoco.mjtObj.mjOBJ_SITE, m.nsite),
            geom_names=names(mujoco.mjtObj.mjOBJ_GEOM, m.ngeom),
            key_names=names(mujoco.mjtObj.mjOBJ_KEY, m.nkey),
        )
        return fm.finalize()
