"""Model composition: attach one FlatModel to a site (or body) of another.

The reference's arm + hand examples build their models this way before mink ever sees them —
``attach_site.attach(hand_mjcf)`` (dm_control) in ``examples/arm_hand_iiwa_allegro.py:32-42``,
``examples/arm_hand_xarm_leap.py`` and ``examples/mobile_kinova_leap.py`` — and hand the compiled
``mujoco.MjModel`` to ``mink.Configuration``.  Neither dm_control nor mujoco is importable here, so the
same operation is offered on the flattened model: every body below the child's world body is re-parented
to the body that carries the attachment site (its pose composed with the site's), the tree is put back
into MuJoCo's depth-first order (a subtree is a contiguous id range — the device kernels rely on it), and
every address array (joints, dofs, qpos, geoms, sites, meshes) is rebuilt.  Host-side, one-time; nothing
here runs per solve.
"""

from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .flatmodel import JNT_BALL, JNT_FREE, FlatModel, dof_width, qpos_width


def _qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _qrot(q, v):
    w, x, y, z = q
    t = 2.0 * np.cross([x, y, z], v)
    return v + w * t + np.cross([x, y, z], t)


class _Body:
    __slots__ = ("name", "pos", "quat", "ipos", "mass", "mass_valid", "mocap", "joints", "geoms", "sites", "children")

    def __init__(self):
        self.joints, self.geoms, self.sites, self.children = [], [], [], []


def _tree(m: FlatModel, prefix: str = "", tag: str = "parent") -> _Body:
    """The model as a tree of bodies carrying their own joints / geoms / sites (qpos0 slices stay with their joint).
    `tag`: which argument of attach() the model is — joints and meshes are identified by (tag, index), so attach(m, m, ...)
    keeps the two copies apart (round-5 advisor finding: keys on id(model) collided there)."""
    nodes = []
    name = lambda n: (prefix + n) if (prefix and n) else n
    for b in range(m.nbody):
        k = _Body()
        k.name = name(m.body_names[b]) if b else "world"
        k.pos, k.quat, k.ipos = m.body_pos[b].copy(), m.body_quat[b].copy(), m.body_ipos[b].copy()
        k.mass, k.mass_valid = float(m.body_mass[b]), int(m.body_mass_valid[b])
        k.mocap = None if m.body_mocapid[b] < 0 else (m.mocap_pos[m.body_mocapid[b]].copy(), m.mocap_quat[m.body_mocapid[b]].copy())
        nodes.append(k)
        if b:
            nodes[int(m.body_parentid[b])].children.append(k)
    for j in range(m.njnt):
        a, t = int(m.jnt_qposadr[j]), int(m.jnt_type[j])
        nodes[int(m.jnt_bodyid[j])].joints.append(dict(
            type=t, pos=m.jnt_pos[j].copy(), axis=m.jnt_axis[j].copy(), range=m.jnt_range[j].copy(),
            limited=int(m.jnt_limited[j]), name=name(m.jnt_names[j]), qpos0=m.qpos0[a:a + qpos_width(t)].copy(), src=(tag, j)))
    for g in range(m.ngeom):
        nodes[int(m.geom_bodyid[g])].geoms.append(dict(
            type=int(m.geom_type[g]), size=m.geom_size[g].copy(), pos=m.geom_pos[g].copy(), quat=m.geom_quat[g].copy(),
            contype=int(m.geom_contype[g]), conaffinity=int(m.geom_conaffinity[g]), valid=int(m.geom_valid[g]),
            name=name(m.geom_names[g]), hull=(m.mesh_hull(g) if int(m.geom_dataid[g]) >= 0 else None), mesh=(tag, int(m.geom_dataid[g]))))
    for s in range(m.nsite):
        nodes[int(m.site_bodyid[s])].sites.append(dict(pos=m.site_pos[s].copy(), quat=m.site_quat[s].copy(), name=name(m.site_names[s])))
    return nodes[0]


def _flatten(world: _Body, keys: Sequence = ()) -> FlatModel:
    """Depth-first numbering of the tree and every address array of a FlatModel (the last third of mjcf.load_mjcf, on a tree).
    `keys`: (name, {joint source → qpos slice}) keyframes to carry over; a joint without an entry keeps its qpos0."""
    order, parent = [], []

    def visit(k, p):
        i = len(order)
        order.append(k); parent.append(p)
        for c in k.children:
            visit(c, i)

    visit(world, 0)
    nbody = len(order)
    J, G, S = [], [], []
    jntnum, jntadr, geomnum, geomadr = [0] * nbody, [-1] * nbody, [0] * nbody, [-1] * nbody
    # (MuJoCo numbers a body's geoms contiguously and the world's first; joints and sites follow the body order)
    for b, k in enumerate(order):
        if k.joints:
            jntadr[b], jntnum[b] = len(J), len(k.joints)
        for j in k.joints:
            J.append(dict(j, body=b))
        if k.geoms:
            geomadr[b], geomnum[b] = len(G), len(k.geoms)
        for g in k.geoms:
            G.append(dict(g, body=b))
        for s in k.sites:
            S.append(dict(s, body=b))
    # MuJoCo's compiler (and dm_control's attach) refuse a model in which two elements of one kind share a name ("repeated
    # name"); FlatModel.finalize() would silently keep the last one, and a FrameTask("ff_tip", "site") or a collision geom list
    # would bind to the wrong hand (round-5 advisor finding: the same hand attached to both palms without prefixes)
    for kind, names in (("body", [k.name for k in order]), ("joint", [j["name"] for j in J]), ("geom", [g["name"] for g in G]),
                        ("site", [s["name"] for s in S]), ("key", [k for k, _ in keys])):
        seen = set()
        for n in names:
            if n and n in seen:
                raise ValueError(f"attach: repeated {kind} name '{n}' in the composed model — give the child a distinct `prefix`")
            seen.add(n)
    njnt = len(J)
    qadr, dadr, nq, nv = [], [], 0, 0
    for j in J:
        qadr.append(nq); dadr.append(nv)
        nq += qpos_width(j["type"]); nv += dof_width(j["type"])
    dofnum, dofadr = [0] * nbody, [-1] * nbody
    dof_body, dof_jnt, dof_parent, last_of = [], [], [], [-1] * nbody
    for b in range(nbody):
        last = last_of[parent[b]] if b else -1
        if jntnum[b]:
            dofadr[b] = dadr[jntadr[b]]
        for j in range(jntadr[b], jntadr[b] + jntnum[b]) if jntnum[b] else ():
            for _ in range(dof_width(J[j]["type"])):
                dof_body.append(b); dof_jnt.append(j); dof_parent.append(last)
                last = len(dof_body) - 1
                dofnum[b] += 1
        last_of[b] = last
    qpos0 = np.zeros(nq)
    for j, a in zip(J, qadr):
        if j["type"] == JNT_FREE and (jntnum[j["body"]] != 1 or parent[j["body"]] != 0):
            raise ValueError("attach: a free joint must stay the only joint of a top-level body (attach the floating model as the parent)")
        qpos0[a:a + qpos_width(j["type"])] = j["qpos0"]
    rootid, weldid = [0] * nbody, [0] * nbody
    for b in range(1, nbody):
        rootid[b] = b if parent[b] == 0 else rootid[parent[b]]
        weldid[b] = b if jntnum[b] > 0 else weldid[parent[b]]
    mass = np.array([k.mass for k in order], dtype=np.float64)
    stm = mass.copy()
    for b in range(nbody - 1, 0, -1):
        stm[parent[b]] += stm[b]
    mocapid, mpos, mquat = [], [], []
    for k in order:
        if k.mocap is None:
            mocapid.append(-1)
        else:
            mocapid.append(len(mpos)); mpos.append(k.mocap[0]); mquat.append(k.mocap[1])
    # meshes: one entry per distinct (source model, mesh id), in geom order
    mesh_ids, hulls, dataid = {}, [], []
    for g in G:
        if g["hull"] is None:
            dataid.append(-1)
            continue
        if g["mesh"] not in mesh_ids:
            mesh_ids[g["mesh"]] = len(hulls); hulls.append(np.asarray(g["hull"], dtype=np.float64).reshape(-1, 3))
        dataid.append(mesh_ids[g["mesh"]])
    key_names, key_qpos = [], []
    for kname, values in keys:
        qk = qpos0.copy()
        for j, a in zip(J, qadr):
            if j["src"] in values:
                qk[a:a + qpos_width(j["type"])] = values[j["src"]]
        key_names.append(kname); key_qpos.append(qk)
    arr = lambda rows, w: np.array(rows, dtype=np.float64).reshape(-1, w) if len(rows) else np.zeros((0, w))
    fm = FlatModel(
        nq=nq, nv=nv, nbody=nbody, njnt=njnt, ngeom=len(G), nsite=len(S), nmocap=len(mpos), nkey=len(key_qpos),
        body_parentid=parent, body_rootid=rootid, body_weldid=weldid, body_mocapid=mocapid, body_jntnum=jntnum, body_jntadr=jntadr,
        body_dofnum=dofnum, body_dofadr=dofadr, body_geomnum=geomnum, body_geomadr=geomadr,
        body_pos=arr([k.pos for k in order], 3), body_quat=arr([k.quat for k in order], 4), body_ipos=arr([k.ipos for k in order], 3),
        body_mass=mass, body_subtreemass=stm, body_mass_valid=np.array([k.mass_valid for k in order], dtype=np.int32),
        jnt_type=[j["type"] for j in J], jnt_qposadr=qadr, jnt_dofadr=dadr, jnt_bodyid=[j["body"] for j in J],
        jnt_limited=[j["limited"] for j in J], jnt_pos=arr([j["pos"] for j in J], 3), jnt_axis=arr([j["axis"] for j in J], 3),
        jnt_range=arr([j["range"] for j in J], 2), dof_bodyid=dof_body, dof_jntid=dof_jnt, dof_parentid=dof_parent, qpos0=qpos0,
        site_bodyid=[s["body"] for s in S], site_pos=arr([s["pos"] for s in S], 3), site_quat=arr([s["quat"] for s in S], 4),
        geom_bodyid=[g["body"] for g in G], geom_type=[g["type"] for g in G], geom_contype=[g["contype"] for g in G],
        geom_conaffinity=[g["conaffinity"] for g in G], geom_valid=[g["valid"] for g in G], geom_size=arr([g["size"] for g in G], 3),
        geom_pos=arr([g["pos"] for g in G], 3), geom_quat=arr([g["quat"] for g in G], 4), geom_dataid=np.array(dataid, dtype=np.int32),
        mesh_vertadr=np.cumsum([0] + [len(h) for h in hulls[:-1]]).astype(np.int32) if hulls else np.zeros(0, np.int32),
        mesh_vertnum=np.array([len(h) for h in hulls], dtype=np.int32),
        mesh_vert=np.concatenate(hulls, axis=0) if hulls else np.zeros((0, 3)),
        key_qpos=arr(key_qpos, nq) if key_qpos else np.zeros((0, nq)), mocap_pos=arr(mpos, 3), mocap_quat=arr(mquat, 4),
        body_names=[k.name for k in order], jnt_names=[j["name"] for j in J], site_names=[s["name"] for s in S],
        geom_names=[g["name"] for g in G], key_names=key_names,
    )
    return fm.finalize()


def _find(k: _Body, pred):
    if pred(k):
        return k
    for c in k.children:
        r = _find(c, pred)
        if r is not None:
            return r
    return None


def attach(parent: FlatModel, child: FlatModel, site: Optional[str] = None, body: Optional[str] = None, prefix: str = "",
           pos: Sequence[float] = (0.0, 0.0, 0.0), quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0),
           child_key: Optional[str] = None) -> FlatModel:
    """``site.attach(child)`` of the reference's arm + hand examples (examples/arm_hand_iiwa_allegro.py:32-42) on FlatModels.

    The bodies below ``child``'s world body become children of the body that carries ``site`` (or of ``body``), their poses
    composed with the site's pose and with (``pos``, ``quat``) — the example moves the palm before attaching; geoms and sites
    of the child's world body move to that body too.  Names of the child get ``prefix`` (dm_control writes ``<model>/<name>``).
    Keyframes: the parent's are kept, the child's joints take ``child_key`` of the child when given, else their qpos0; the
    child's own keyframes are dropped.  Two elements of one kind with the same name raise ValueError, as MuJoCo's compiler and
    dm_control do ("repeated name"): attach the same child twice with two prefixes.
    Difference from dm_control: ``site.attach`` wraps the child in an extra jointless frame body named after the child model;
    here the child's top-level bodies hang directly under the body that carries the site, so ``nbody`` is one less per
    attachment and body ids after the attachment point are shifted by one against the reference's compiled model — kinematics,
    dofs, qpos layout and every named lookup (bodies, joints, sites, geoms) are the same.
    Returns a new FlatModel in MuJoCo's depth-first body order; neither argument is modified."""
    if (site is None) == (body is None):
        raise ValueError("attach: give exactly one of `site` and `body`")
    root = _tree(parent)
    sub = _tree(child, prefix, "child")
    if site is not None:
        host = _find(root, lambda k: any(s["name"] == site for s in k.sites))
        if host is None:
            raise KeyError(f"attach: no site named '{site}' in the parent model")
        s = next(s for s in host.sites if s["name"] == site)
        fpos, fquat = s["pos"], s["quat"]
    else:
        host = _find(root, lambda k: k.name == body)
        if host is None:
            raise KeyError(f"attach: no body named '{body}' in the parent model")
        fpos, fquat = np.zeros(3), np.array([1.0, 0.0, 0.0, 0.0])
    fpos = fpos + _qrot(fquat, np.asarray(pos, dtype=np.float64))
    fquat = _qmul(fquat, np.asarray(quat, dtype=np.float64))
    fquat = fquat / np.linalg.norm(fquat)
    for c in sub.children:
        if c.mocap is not None:
            raise ValueError("attach: the child model carries a mocap body (mocap bodies must stay children of the world)")
        c.pos, c.quat = fpos + _qrot(fquat, c.pos), _qmul(fquat, c.quat)
        host.children.append(c)
    for g in sub.geoms:
        host.geoms.append(dict(g, pos=fpos + _qrot(fquat, g["pos"]), quat=_qmul(fquat, g["quat"])))
    for s in sub.sites:
        host.sites.append(dict(s, pos=fpos + _qrot(fquat, s["pos"]), quat=_qmul(fquat, s["quat"])))
    # keyframes of the parent, completed with the child's joints
    ck = {}
    if child_key is not None:
        kq = child.key_qpos[child.name2id("key", child_key)]
        for j in range(child.njnt):
            a = int(child.jnt_qposadr[j])
            ck[("child", j)] = kq[a:a + qpos_width(int(child.jnt_type[j]))].copy()
    keys = []
    for i, kname in enumerate(parent.key_names):
        vals = dict(ck)
        for j in range(parent.njnt):
            a = int(parent.jnt_qposadr[j])
            vals[("parent", j)] = parent.key_qpos[i][a:a + qpos_width(int(parent.jnt_type[j]))].copy()
        keys.append((kname, vals))
    return _flatten(root, keys)
