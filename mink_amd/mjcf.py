"""Minimal MJCF subset reader → :class:`FlatModel` (host-side, setup time only).

Used when the ``mujoco`` package is not importable (it is not in the build image):
``FlatModel.from_mjmodel`` is the production ingest, this reader covers the MJCF
subset the benchmark robots use following
MuJoCo's documented compile semantics: ``<include>`` splicing, nested
``<default class>`` + ``childclass``, depth-first body numbering, normalised
quaternions/axes, ``autolimits``, ``<freejoint>``, keyframes, mocap bodies.

Not supported (raises or marks ``geom_valid=0``): mesh-derived geom frames/sizes,
``<frame>``/``<replicate>``/``<attach>``, ``<composite>``, tendons, equality.
"""

from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

from .flatmodel import (
    GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_ELLIPSOID, GEOM_MESH, GEOM_PLANE,
    GEOM_SPHERE, GEOM_TYPE_NAMES, JNT_BALL, JNT_FREE, JNT_HINGE, JNT_SLIDE,
    FlatModel, mjMINVAL,
)

_JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}


class MjcfError(ValueError):
    pass


# ------------------------------------------------------------------ small math
def _vec(text: Optional[str], n: Optional[int] = None, default=None) -> np.ndarray:
    if text is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(t) for t in text.split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise MjcfError(f"expected {n} numbers, got '{text}'")
    return v


def _normalize(v: np.ndarray) -> np.ndarray:
    n = np.linalg.norm(v)
    if n < mjMINVAL:
        raise MjcfError("zero-length vector cannot be normalised")
    return v / n


def _quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def _axisangle_quat(axis, angle):
    axis = _normalize(np.asarray(axis, dtype=np.float64))
    s = math.sin(0.5 * angle)
    return np.array([math.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def _mat_quat(R: np.ndarray) -> np.ndarray:
    """Rotation matrix → unit quaternion (Shepperd)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        q = np.array([1 + t, R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    elif R[0, 0] >= R[1, 1] and R[0, 0] >= R[2, 2]:
        q = np.array([R[2, 1] - R[1, 2], 1 + R[0, 0] - R[1, 1] - R[2, 2],
                      R[0, 1] + R[1, 0], R[0, 2] + R[2, 0]])
    elif R[1, 1] >= R[2, 2]:
        q = np.array([R[0, 2] - R[2, 0], R[0, 1] + R[1, 0],
                      1 - R[0, 0] + R[1, 1] - R[2, 2], R[1, 2] + R[2, 1]])
    else:
        q = np.array([R[1, 0] - R[0, 1], R[0, 2] + R[2, 0], R[1, 2] + R[2, 1],
                      1 - R[0, 0] - R[1, 1] + R[2, 2]])
    return _normalize(q)


def _quat_rotate(q, v) -> np.ndarray:
    w, u = q[0], np.asarray(q[1:], dtype=np.float64)
    t = 2.0 * np.cross(u, v)
    return np.asarray(v, dtype=np.float64) + w * t + np.cross(u, t)


def _z2quat(vec) -> np.ndarray:
    """Quaternion rotating +z onto ``vec`` (MuJoCo ``zaxis`` / ``fromto`` rule)."""
    vec = _normalize(np.asarray(vec, dtype=np.float64))
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if vec[2] > 0 else np.array([0.0, 1.0, 0, 0])
    ang = math.atan2(s, float(vec[2]))
    return _axisangle_quat(axis / s, ang)


class _Compiler:
    def __init__(self):
        self.angle_scale = math.pi / 180.0  # MJCF default: degrees
        self.autolimits = True  # MuJoCo >= 3.0 default
        self.eulerseq = "xyz"
        self.meshdir = ""
        self.exactmeshinertia = False  # the 3.1.x default: "legacy" pyramid sums (mink_amd/meshes.py mass_properties)

    def update(self, el: ET.Element):
        em = el.get("exactmeshinertia")
        if em is not None:
            self.exactmeshinertia = em == "true"
        md = el.get("meshdir")
        if md is not None:
            self.meshdir = md
        a = el.get("angle")
        if a is not None:
            self.angle_scale = 1.0 if a == "radian" else math.pi / 180.0
        al = el.get("autolimits")
        if al is not None:
            self.autolimits = al == "true"
        es = el.get("eulerseq")
        if es is not None:
            self.eulerseq = es


def _orientation(attrs: Dict[str, str], comp: _Compiler) -> np.ndarray:
    if "quat" in attrs:
        return _normalize(_vec(attrs["quat"], 4))
    if "axisangle" in attrs:
        v = _vec(attrs["axisangle"], 4)
        return _axisangle_quat(v[:3], v[3] * comp.angle_scale)
    if "euler" in attrs:
        e = _vec(attrs["euler"], 3) * comp.angle_scale
        q = np.array([1.0, 0, 0, 0])
        for ax_char, ang in zip(comp.eulerseq, e):
            ax = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}[ax_char.lower()]
            r = _axisangle_quat(ax, ang)
            # lower-case = rotating (intrinsic) axes: post-multiply; upper = fixed.
            q = _quat_mul(q, r) if ax_char.islower() else _quat_mul(r, q)
        return _normalize(q)
    if "xyaxes" in attrs:
        v = _vec(attrs["xyaxes"], 6)
        x = _normalize(v[:3])
        y = v[3:] - x * np.dot(x, v[3:])
        y = _normalize(y)
        z = np.cross(x, y)
        return _mat_quat(np.stack([x, y, z], axis=1))
    if "zaxis" in attrs:
        return _z2quat(_vec(attrs["zaxis"], 3))
    return np.array([1.0, 0.0, 0.0, 0.0])


# ------------------------------------------------------------------ xml loading
def _load_with_includes(path: str) -> ET.Element:
    root = ET.parse(path).getroot()
    base = os.path.dirname(os.path.abspath(path))

    def splice(parent: ET.Element):
        i = 0
        while i < len(parent):
            child = parent[i]
            if child.tag == "include":
                inc = _load_with_includes(os.path.join(base, child.get("file")))
                parent.remove(child)
                for k, sub in enumerate(list(inc)):
                    parent.insert(i + k, sub)
                i += len(inc)
            else:
                splice(child)
                i += 1

    splice(root)
    return root


class _Defaults:
    """Nested ``<default class>`` tree: class → {tag → attrs} with inheritance."""

    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}

    def add(self, el: ET.Element, parent_cls: Optional[str]):
        if parent_cls is None:
            # top-level <default> blocks (one per included file) accumulate in "main"
            name, base = "main", self.classes["main"]
        else:
            name, base = el.get("class"), self.classes[parent_cls]
            if not name:
                raise MjcfError("nested <default> requires a class attribute")
        merged = {t: dict(a) for t, a in base.items()}
        for child in el:
            if child.tag == "default":
                continue
            merged.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = merged
        for child in el:
            if child.tag == "default":
                self.add(child, name)

    def resolve(self, tag: str, el: ET.Element, childclass: Optional[str]) -> Dict[str, str]:
        cls = el.get("class") or childclass or "main"
        if cls not in self.classes:
            raise MjcfError(f"unknown default class '{cls}'")
        attrs = dict(self.classes[cls].get(tag, {}))
        attrs.update(el.attrib)
        return attrs


# ------------------------------------------------------------------ geom helpers
def _geom_volume(gtype: int, size: np.ndarray) -> float:
    if gtype == GEOM_SPHERE:
        return 4.0 / 3.0 * math.pi * size[0] ** 3
    if gtype == GEOM_CAPSULE:
        return math.pi * size[0] ** 2 * (2 * size[1]) + 4.0 / 3.0 * math.pi * size[0] ** 3
    if gtype == GEOM_CYLINDER:
        return math.pi * size[0] ** 2 * (2 * size[1])
    if gtype == GEOM_BOX:
        return 8.0 * size[0] * size[1] * size[2]
    if gtype == GEOM_ELLIPSOID:
        return 4.0 / 3.0 * math.pi * size[0] * size[1] * size[2]
    return 0.0


# ------------------------------------------------------------------ the reader
def load_mjcf(path: str) -> FlatModel:
    """Compile the MJCF file at ``path`` into a :class:`FlatModel`."""
    root = _load_with_includes(path)
    if root.tag != "mujoco":
        raise MjcfError("root element must be <mujoco>")
    comp = _Compiler()
    for el in root.findall("compiler"):
        comp.update(el)
    defaults = _Defaults()
    for el in root.findall("default"):
        defaults.add(el, None)

    B: Dict[str, list] = {k: [] for k in (
        "parent", "pos", "quat", "ipos", "mass", "mocap", "name", "jntnum", "jntadr",
        "geomnum", "geomadr")}
    J: Dict[str, list] = {k: [] for k in (
        "type", "body", "pos", "axis", "range", "limited", "name", "ref")}
    S: Dict[str, list] = {k: [] for k in ("body", "pos", "quat", "name")}
    G: Dict[str, list] = {k: [] for k in (
        "body", "type", "size", "pos", "quat", "contype", "conaffinity", "name", "valid", "dataid")}
    # mesh assets (mink_amd/meshes.py): read lazily — only when a collision candidate refers to one (a visual geom,
    # contype = conaffinity = 0, can never be in a pair: mink/limits/collision_avoidance_limit.py:109-115)
    from . import meshes as _meshes
    mesh_elems = [m for asset in root.findall("asset") for m in asset.findall("mesh")]
    assets = _meshes.load_assets(mesh_elems, os.path.join(os.path.dirname(os.path.abspath(path)), comp.meshdir),
                                 lambda el: defaults.resolve("mesh", el, None), exact=comp.exactmeshinertia)
    mesh_ids: Dict[str, int] = {}          # asset name → index in the FlatModel's mesh arrays (meshes of type="mesh" geoms)
    mesh_hulls: List[np.ndarray] = []

    mass_valid: Dict[int, int] = {}
    # world body
    for k, v in (("parent", 0), ("pos", np.zeros(3)), ("quat", np.array([1.0, 0, 0, 0])),
                 ("ipos", np.zeros(3)), ("mass", 0.0), ("mocap", False), ("name", "world"),
                 ("jntnum", 0), ("jntadr", -1), ("geomnum", 0), ("geomadr", -1)):
        B[k].append(v)

    def add_geom(el, body_id, childclass, gmass_acc):
        a = defaults.resolve("geom", el, childclass)
        has_mesh = "mesh" in a
        tname = a.get("type", "sphere")
        if tname not in GEOM_TYPE_NAMES:
            raise MjcfError(f"unsupported geom type '{tname}'")
        gtype = GEOM_TYPE_NAMES[tname]
        size = np.zeros(3)
        sv = _vec(a.get("size"))
        if sv is not None:
            size[: sv.size] = sv
        pos = _vec(a.get("pos"), 3, [0, 0, 0])
        quat = _orientation(a, comp)
        valid = 1
        if "fromto" in a:
            ft = _vec(a["fromto"], 6)
            d = ft[3:] - ft[:3]
            pos = 0.5 * (ft[:3] + ft[3:])
            quat = _z2quat(d)
            half = 0.5 * np.linalg.norm(d)
            if gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
                size[1] = half
            else:
                size[2] = half
        contype, conaff = int(a.get("contype", 1)), int(a.get("conaffinity", 1))
        dataid = -1
        pos_attr, quat_attr = pos.copy(), quat.copy()          # the geom's own frame, before a mesh's inertial frame is composed in
        if gtype == GEOM_MESH or has_mesh:
            valid = 0  # local frame / fitted size need the mesh asset
            lazy = assets.get(a.get("mesh", ""))
            if lazy is not None and (contype or conaff) and os.path.exists(lazy.path):
                # The compiler re-expresses a mesh in its inertial frame (centre of mass, principal axes) and composes
                # that frame into the geom's; a primitive with a mesh attribute is FITTED to the mesh's inertia box
                # (mink_amd/meshes.py).  valid = 2: usable as a collision geom — the axes of the inertial frame are
                # only defined up to half turns (eigenvectors), which no primitive and no hull distance can see, but a
                # FrameTask on this geom could: frames on it stay refused (Configuration._frame_id).
                # An asset this reader cannot compile (a format it does not read, an open surface, a flat hull) leaves the
                # geom at valid = 0 — refused where it is USED, with CollisionAvoidanceLimit's own message — instead of
                # failing the whole model load (round-3 advisor finding).
                try:
                    asset = lazy.get()
                    hull = None
                    if gtype == GEOM_MESH:
                        hull = asset.hull_vert
                    else:
                        fitted = _meshes.fit_primitive(gtype, asset.boxsz) * float(a.get("fitscale", 1.0))
                except (_meshes.MeshError, ValueError, RuntimeError, OSError):        # (scipy's QhullError is a RuntimeError)
                    asset = None
                if asset is not None:
                    pos = pos + _quat_rotate(quat, asset.pos)
                    quat = _normalize(_quat_mul(quat, asset.quat))
                    if gtype == GEOM_MESH:
                        name_m = a["mesh"]
                        if name_m not in mesh_ids:
                            mesh_ids[name_m] = len(mesh_hulls)
                            mesh_hulls.append(hull)
                        dataid = mesh_ids[name_m]
                    else:
                        size = fitted
                    valid = 2
        if gtype not in (GEOM_MESH, GEOM_PLANE) and valid:
            need = {GEOM_SPHERE: 1, GEOM_CAPSULE: 2, GEOM_CYLINDER: 2,
                    GEOM_BOX: 3, GEOM_ELLIPSOID: 3}[gtype]
            if np.any(size[:need] <= 0):
                raise MjcfError(f"geom size missing for type '{tname}'")
        G["body"].append(body_id); G["type"].append(gtype); G["size"].append(size)
        G["pos"].append(pos); G["quat"].append(quat)
        G["contype"].append(contype)
        G["conaffinity"].append(conaff)
        G["name"].append(a.get("name", "")); G["valid"].append(valid); G["dataid"].append(dataid)
        unknown = False
        if "mass" in a:
            m = float(a["mass"])
        elif valid == 1:
            m = float(a.get("density", 1000.0)) * _geom_volume(gtype, size)
        elif valid == 2 and gtype != GEOM_MESH:
            m = float(a.get("density", 1000.0)) * _geom_volume(gtype, size)      # a primitive fitted to a mesh weighs what the primitive weighs
        else:
            m = 0.0
            density = float(a.get("density", 1000.0))
            lazy = assets.get(a.get("mesh", ""))
            if density != 0.0:
                # mass = density x volume of the mesh (or of the primitive fitted to it), at the mesh's centre of mass: needs the
                # asset — compiled only if the body turns out to have no <inertial> (visit_body), the one case in which it is used
                def unknown(lazy=lazy, gtype=gtype, p0=pos_attr, q0=quat_attr, density=density, fs=float(a.get("fitscale", 1.0))):
                    if lazy is None or not os.path.exists(lazy.path):
                        raise _meshes.MeshError("mesh asset not available")
                    asset = lazy.get()
                    vol = asset.volume if gtype == GEOM_MESH else _geom_volume(gtype, _meshes.fit_primitive(gtype, asset.boxsz) * fs)
                    return density * vol, p0 + _quat_rotate(q0, asset.pos)
        gmass_acc.append((m, pos, unknown))

    def add_site(el, body_id, childclass):
        a = defaults.resolve("site", el, childclass)
        pos = _vec(a.get("pos"), 3, [0, 0, 0])
        quat = _orientation(a, comp)
        if "fromto" in a:
            ft = _vec(a["fromto"], 6)
            pos = 0.5 * (ft[:3] + ft[3:]); quat = _z2quat(ft[3:] - ft[:3])
        S["body"].append(body_id); S["pos"].append(pos); S["quat"].append(quat)
        S["name"].append(a.get("name", ""))

    def add_joint(el, body_id, childclass, free=False):
        if free:
            a = dict(el.attrib); a["type"] = "free"
        else:
            a = defaults.resolve("joint", el, childclass)
        jt = _JNT_TYPES[a.get("type", "hinge")]
        rng = _vec(a.get("range"), 2, [0, 0])
        limited_attr = a.get("limited", "auto")
        if limited_attr == "auto":
            limited = comp.autolimits and ("range" in a)
        else:
            limited = limited_attr == "true"
        if jt in (JNT_HINGE, JNT_BALL):
            rng = rng * comp.angle_scale
        ref = float(a.get("ref", 0.0)) * (comp.angle_scale if jt == JNT_HINGE else 1.0)
        axis = _vec(a.get("axis"), 3, [0, 0, 1])
        if jt in (JNT_HINGE, JNT_SLIDE):
            axis = _normalize(axis)
        if jt == JNT_FREE:
            limited = False
            axis = np.array([0.0, 0.0, 1.0])
        J["type"].append(jt); J["body"].append(body_id)
        J["pos"].append(np.zeros(3) if jt == JNT_FREE else _vec(a.get("pos"), 3, [0, 0, 0]))
        J["axis"].append(axis); J["range"].append(rng); J["limited"].append(int(limited))
        J["name"].append(a.get("name", "")); J["ref"].append(ref)

    def visit_body_children(el, body_id, childclass):
        """Handle non-body children of a body element (order: document)."""
        gmass: list = []
        jstart = len(J["type"])
        gstart = len(G["type"])
        inertial = None
        for ch in el:
            if ch.tag == "inertial":
                inertial = ch
            elif ch.tag == "joint":
                add_joint(ch, body_id, childclass)
            elif ch.tag == "freejoint":
                add_joint(ch, body_id, childclass, free=True)
            elif ch.tag == "geom":
                add_geom(ch, body_id, childclass, gmass)
            elif ch.tag == "site":
                add_site(ch, body_id, childclass)
            elif ch.tag in ("frame", "replicate", "attach", "composite", "flexcomp"):
                raise MjcfError(f"<{ch.tag}> is not supported by the MJCF subset reader")
        B["jntnum"][body_id] = len(J["type"]) - jstart
        B["jntadr"][body_id] = jstart if len(J["type"]) > jstart else -1
        B["geomnum"][body_id] = len(G["type"]) - gstart
        B["geomadr"][body_id] = gstart if len(G["type"]) > gstart else -1
        if inertial is not None:
            B["mass"][body_id] = float(inertial.get("mass"))
            B["ipos"][body_id] = _vec(inertial.get("pos"), 3, [0, 0, 0])
        elif body_id != 0:
            # no <inertial>: mass and centre of mass from the geoms (inertiafromgeom's default); a mesh geom contributes density x
            # the mesh's volume at the mesh's centre of mass (meshes.py: the compiler's legacy volume).  An asset that cannot be
            # compiled leaves the body flagged (FlatModel.require_valid_masses refuses it for ComTask) instead of weighing 0.
            parts = []
            for m, p, u in gmass:
                if callable(u):
                    try:
                        m, p = u()
                    except (_meshes.MeshError, ValueError, RuntimeError, OSError):
                        mass_valid[body_id] = 0
                        continue
                parts.append((m, p))
            mt = sum(m for m, _ in parts)
            B["mass"][body_id] = mt
            if mt > mjMINVAL:
                B["ipos"][body_id] = sum(m * p for m, p in parts) / mt
        for ch in el:
            if ch.tag == "body":
                visit_body(ch, body_id, childclass)

    def visit_body(el, parent_id, childclass):
        childclass = el.get("childclass", childclass)
        body_id = len(B["parent"])
        B["parent"].append(parent_id)
        B["pos"].append(_vec(el.get("pos"), 3, [0, 0, 0]))
        B["quat"].append(_orientation(el.attrib, comp))
        B["ipos"].append(np.zeros(3)); B["mass"].append(0.0)
        B["mocap"].append(el.get("mocap", "false") == "true")
        B["name"].append(el.get("name", ""))
        B["jntnum"].append(0); B["jntadr"].append(-1)
        B["geomnum"].append(0); B["geomadr"].append(-1)
        visit_body_children(el, body_id, childclass)

    # multiple <worldbody> sections concatenate in document order.  World-level
    # geoms/sites belong to body 0; MuJoCo numbers a body's geoms contiguously, so
    # collect world geoms first, then descend.
    worldbodies = root.findall("worldbody")
    gm: list = []
    gstart = 0
    for wb in worldbodies:
        for ch in wb:
            if ch.tag == "geom":
                add_geom(ch, 0, wb.get("childclass"), gm)
            elif ch.tag == "site":
                add_site(ch, 0, wb.get("childclass"))
            elif ch.tag in ("joint", "freejoint", "inertial"):
                raise MjcfError("world body cannot have joints or inertial")
    B["geomnum"][0] = len(G["type"]) - gstart
    B["geomadr"][0] = 0 if B["geomnum"][0] else -1
    for wb in worldbodies:
        for ch in wb:
            if ch.tag == "body":
                visit_body(ch, 0, wb.get("childclass"))

    nbody, njnt = len(B["parent"]), len(J["type"])
    from .flatmodel import dof_width, qpos_width

    jnt_qposadr, jnt_dofadr = [], []
    nq = nv = 0
    for j in range(njnt):
        jnt_qposadr.append(nq); jnt_dofadr.append(nv)
        nq += qpos_width(J["type"][j]); nv += dof_width(J["type"][j])

    body_dofnum = [0] * nbody
    body_dofadr = [-1] * nbody
    dof_bodyid, dof_jntid, dof_parentid = [], [], []
    last_dof_of_body = [-1] * nbody  # last dof on chain up to and incl. this body
    for b in range(nbody):
        p = B["parent"][b]
        last = last_dof_of_body[p] if b > 0 else -1
        if B["jntnum"][b] > 0:
            body_dofadr[b] = jnt_dofadr[B["jntadr"][b]]
        for j in range(B["jntadr"][b], B["jntadr"][b] + B["jntnum"][b]) if B["jntnum"][b] else ():
            for _ in range(dof_width(J["type"][j])):
                dof_bodyid.append(b); dof_jntid.append(j); dof_parentid.append(last)
                last = len(dof_bodyid) - 1
                body_dofnum[b] += 1
        last_dof_of_body[b] = last

    qpos0 = np.zeros(nq)
    for j in range(njnt):
        a, t, b = jnt_qposadr[j], J["type"][j], J["body"][j]
        if t == JNT_FREE:
            if B["jntnum"][b] != 1 or B["parent"][b] != 0:
                raise MjcfError("free joint must be the only joint of a top-level body")
            qpos0[a:a + 3] = B["pos"][b]; qpos0[a + 3:a + 7] = B["quat"][b]
        elif t == JNT_BALL:
            qpos0[a:a + 4] = [1, 0, 0, 0]
        else:
            qpos0[a] = J["ref"][j]

    body_rootid = [0] * nbody
    body_weldid = [0] * nbody
    for b in range(1, nbody):
        p = B["parent"][b]
        body_rootid[b] = b if p == 0 else body_rootid[p]
        body_weldid[b] = b if B["jntnum"][b] > 0 else body_weldid[p]
    subtreemass = np.array(B["mass"], dtype=np.float64)
    for b in range(nbody - 1, 0, -1):
        subtreemass[B["parent"][b]] += subtreemass[b]

    mocapid, mocap_pos, mocap_quat = [], [], []
    for b in range(nbody):
        if B["mocap"][b]:
            if B["parent"][b] != 0 or B["jntnum"][b] != 0:
                raise MjcfError("mocap body must be a jointless child of the world")
            mocapid.append(len(mocap_pos))
            mocap_pos.append(B["pos"][b]); mocap_quat.append(B["quat"][b])
        else:
            mocapid.append(-1)

    key_names, key_qpos = [], []
    for kf in root.findall("keyframe"):
        for key in kf.findall("key"):
            key_names.append(key.get("name", ""))
            qp = _vec(key.get("qpos"))
            if qp is None:
                qp = qpos0.copy()
            if qp.size != nq:
                raise MjcfError(f"keyframe '{key.get('name')}' has {qp.size} qpos, model nq={nq}")
            key_qpos.append(qp)

    def arr(lst, width):
        return np.array(lst, dtype=np.float64).reshape(-1, width) if lst else np.zeros((0, width))

    fm = FlatModel(
        nq=nq, nv=nv, nbody=nbody, njnt=njnt, ngeom=len(G["type"]), nsite=len(S["body"]),
        nmocap=len(mocap_pos), nkey=len(key_qpos),
        body_parentid=B["parent"], body_rootid=body_rootid, body_weldid=body_weldid,
        body_mocapid=mocapid, body_jntnum=B["jntnum"], body_jntadr=B["jntadr"],
        body_dofnum=body_dofnum, body_dofadr=body_dofadr, body_geomnum=B["geomnum"],
        body_geomadr=B["geomadr"], body_pos=arr(B["pos"], 3), body_quat=arr(B["quat"], 4),
        body_ipos=arr(B["ipos"], 3), body_mass=np.array(B["mass"]), body_subtreemass=subtreemass,
        body_mass_valid=np.array([mass_valid.get(b, 1) for b in range(len(B["parent"]))], dtype=np.int32),
        jnt_type=J["type"], jnt_qposadr=jnt_qposadr, jnt_dofadr=jnt_dofadr,
        jnt_bodyid=J["body"], jnt_limited=J["limited"], jnt_pos=arr(J["pos"], 3),
        jnt_axis=arr(J["axis"], 3), jnt_range=arr(J["range"], 2), dof_bodyid=dof_bodyid,
        dof_jntid=dof_jntid, dof_parentid=dof_parentid, qpos0=qpos0,
        site_bodyid=S["body"], site_pos=arr(S["pos"], 3), site_quat=arr(S["quat"], 4),
        geom_bodyid=G["body"], geom_type=G["type"], geom_contype=G["contype"],
        geom_conaffinity=G["conaffinity"], geom_valid=G["valid"], geom_size=arr(G["size"], 3),
        geom_pos=arr(G["pos"], 3), geom_quat=arr(G["quat"], 4),
        geom_dataid=np.array(G["dataid"], dtype=np.int32),
        mesh_vertadr=np.cumsum([0] + [len(h) for h in mesh_hulls[:-1]]).astype(np.int32) if mesh_hulls else np.zeros(0, np.int32),
        mesh_vertnum=np.array([len(h) for h in mesh_hulls], dtype=np.int32),
        mesh_vert=np.concatenate(mesh_hulls, axis=0) if mesh_hulls else np.zeros((0, 3)),
        key_qpos=arr(key_qpos, nq) if key_qpos else np.zeros((0, nq)),
        mocap_pos=arr(mocap_pos, 3), mocap_quat=arr(mocap_quat, 4),
        body_names=B["name"], jnt_names=J["name"], site_names=S["name"],
        geom_names=G["name"], key_names=key_names,
    )
    return fm.finalize()


def loads_mjcf(xml_text: str, tmpdir: Optional[str] = None) -> FlatModel:
    """Compile an MJCF string (no relative includes)."""
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".xml", dir=tmpdir, delete=False) as fh:
        fh.write(xml_text)
        name = fh.name
    try:
        return load_mjcf(name)
    finally:
        os.unlink(name)
