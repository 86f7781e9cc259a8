"""ORACLE (test infrastructure, not product code) — MuJoCo arithmetic restated in numpy.

Parity status: **unpinned against MuJoCo itself**.  The `mujoco` wheel (reference
pin: mujoco >= 3.1.6, /root/reference/pyproject.toml:27-28) is a third-party
dependency that is not vendored under /root/reference and is not installed in the
build image, so these functions restate MuJoCo's published algorithms
(engine_core_smooth.c mj_kinematics/mj_comPos, engine_support.c mj_jac*/
mj_differentiatePos/mj_integratePos, engine_util_spatial.c quaternion helpers,
engine_collision_primitive.c capsule/sphere routines) from SURVEY.md Appendix A.
They are pinned indirectly by the reference's own property tests re-run on top of
them (finite-difference Jacobians tests/test_jacobians.py:41-68, site pose
tests/test_configuration.py:36-53) — see tests/test_oracle_*.py.

Each function cites the reference call site it serves.  Single-problem, loop-based,
float64; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.
"""

from __future__ import annotations

import math

import numpy as np

mjMINVAL = 1e-15
mjMAXVAL = 1e10
mjPI = math.pi

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX = 0, 2, 3, 4, 5, 6


# ------------------------------------------------------------- mju_* helpers
def mju_normalize3(v: np.ndarray) -> float:
    """In place; returns the norm (mink/limits/collision_avoidance_limit.py:49)."""
    n = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    if n < mjMINVAL:
        v[0], v[1], v[2] = 1.0, 0.0, 0.0
    else:
        inv = 1.0 / n
        v[0] *= inv; v[1] *= inv; v[2] *= inv
    return n


def mju_normalize4(q: np.ndarray) -> float:
    n = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    if n < mjMINVAL:
        q[0], q[1], q[2], q[3] = 1.0, 0.0, 0.0, 0.0
    elif abs(n - 1.0) > mjMINVAL:
        inv = 1.0 / n
        q *= inv
    return n


def mju_mulQuat(res: np.ndarray, a: np.ndarray, b: np.ndarray) -> None:
    """Hamilton product (mink/lie/so3.py:150)."""
    r0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3]
    r1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2]
    r2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1]
    r3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]
    res[0], res[1], res[2], res[3] = r0, r1, r2, r3


def mju_negQuat(res: np.ndarray, q: np.ndarray) -> None:
    res[0], res[1], res[2], res[3] = q[0], -q[1], -q[2], -q[3]


def mju_quat2Mat(res: np.ndarray, q: np.ndarray) -> None:
    """Row-major 3x3 (mink/lie/so3.py:113)."""
    q00, q01, q02, q03 = q[0] * q[0], q[0] * q[1], q[0] * q[2], q[0] * q[3]
    q11, q12, q13 = q[1] * q[1], q[1] * q[2], q[1] * q[3]
    q22, q23, q33 = q[2] * q[2], q[2] * q[3], q[3] * q[3]
    res[0] = q00 + q11 - q22 - q33
    res[4] = q00 - q11 + q22 - q33
    res[8] = q00 - q11 - q22 + q33
    res[1] = 2 * (q12 - q03)
    res[2] = 2 * (q13 + q02)
    res[3] = 2 * (q12 + q03)
    res[5] = 2 * (q23 - q01)
    res[6] = 2 * (q13 - q02)
    res[7] = 2 * (q23 + q01)


def mju_mat2Quat(quat: np.ndarray, mat: np.ndarray) -> None:
    """Row-major 3x3 → unit quaternion (mink/lie/so3.py:83, mink/utils.py:35)."""
    m = mat
    if m[0] + m[4] + m[8] > 0:
        quat[0] = 0.5 * math.sqrt(1 + m[0] + m[4] + m[8])
        quat[1] = 0.25 * (m[7] - m[5]) / quat[0]
        quat[2] = 0.25 * (m[2] - m[6]) / quat[0]
        quat[3] = 0.25 * (m[3] - m[1]) / quat[0]
    elif m[0] > m[4] and m[0] > m[8]:
        quat[1] = 0.5 * math.sqrt(1 + m[0] - m[4] - m[8])
        quat[0] = 0.25 * (m[7] - m[5]) / quat[1]
        quat[2] = 0.25 * (m[1] + m[3]) / quat[1]
        quat[3] = 0.25 * (m[2] + m[6]) / quat[1]
    elif m[4] > m[8]:
        quat[2] = 0.5 * math.sqrt(1 - m[0] + m[4] - m[8])
        quat[0] = 0.25 * (m[2] - m[6]) / quat[2]
        quat[1] = 0.25 * (m[1] + m[3]) / quat[2]
        quat[3] = 0.25 * (m[5] + m[7]) / quat[2]
    else:
        quat[3] = 0.5 * math.sqrt(1 - m[0] - m[4] + m[8])
        quat[0] = 0.25 * (m[3] - m[1]) / quat[3]
        quat[1] = 0.25 * (m[2] + m[6]) / quat[3]
        quat[2] = 0.25 * (m[5] + m[7]) / quat[3]
    mju_normalize4(quat)


def mju_axisAngle2Quat(res: np.ndarray, axis: np.ndarray, angle: float) -> None:
    if angle == 0:
        res[0], res[1], res[2], res[3] = 1.0, 0.0, 0.0, 0.0
    else:
        s = math.sin(angle * 0.5)
        res[0] = math.cos(angle * 0.5)
        res[1], res[2], res[3] = axis[0] * s, axis[1] * s, axis[2] * s


def _mulMatVec3(mat9: np.ndarray, v: np.ndarray) -> np.ndarray:
    return np.array([
        mat9[0] * v[0] + mat9[1] * v[1] + mat9[2] * v[2],
        mat9[3] * v[0] + mat9[4] * v[1] + mat9[5] * v[2],
        mat9[6] * v[0] + mat9[7] * v[1] + mat9[8] * v[2],
    ])


def mju_rotVecQuat(vec: np.ndarray, quat: np.ndarray) -> np.ndarray:
    if vec[0] == 0 and vec[1] == 0 and vec[2] == 0:
        return np.zeros(3)
    if quat[0] == 1 and quat[1] == 0 and quat[2] == 0 and quat[3] == 0:
        return np.array(vec, dtype=np.float64)
    mat = np.empty(9)
    mju_quat2Mat(mat, quat)
    return _mulMatVec3(mat, vec)


def mju_quat2Vel(quat: np.ndarray, dt: float) -> np.ndarray:
    axis = np.array([quat[1], quat[2], quat[3]])
    sin_a_2 = mju_normalize3(axis)
    speed = 2 * math.atan2(sin_a_2, quat[0])
    if speed > mjPI:
        speed -= 2 * mjPI
    speed /= dt
    return axis * speed


def mju_quatIntegrate(quat: np.ndarray, vel: np.ndarray, scale: float) -> None:
    tmp = np.array(vel, dtype=np.float64)
    angle = scale * mju_normalize3(tmp)
    qrot = np.empty(4)
    mju_axisAngle2Quat(qrot, tmp, angle)
    mju_normalize4(quat)
    mju_mulQuat(quat, quat.copy(), qrot)


# ------------------------------------------------------------------- MjData
class Data:
    """The subset of ``mjData`` mink reads (mink/configuration.py:50-64)."""

    def __init__(self, m):
        self.qpos = np.array(m.qpos0, dtype=np.float64)
        self.mocap_pos = np.array(m.mocap_pos, dtype=np.float64).reshape(-1, 3)
        self.mocap_quat = np.array(m.mocap_quat, dtype=np.float64).reshape(-1, 4)
        self.xpos = np.zeros((m.nbody, 3))
        self.xquat = np.zeros((m.nbody, 4)); self.xquat[:, 0] = 1
        self.xmat = np.zeros((m.nbody, 9))
        self.xipos = np.zeros((m.nbody, 3))
        self.xanchor = np.zeros((m.njnt, 3))
        self.xaxis = np.zeros((m.njnt, 3))
        self.geom_xpos = np.zeros((m.ngeom, 3))
        self.geom_xmat = np.zeros((m.ngeom, 9))
        self.site_xpos = np.zeros((m.nsite, 3))
        self.site_xmat = np.zeros((m.nsite, 9))
        self.subtree_com = np.zeros((m.nbody, 3))
        self.cdof = np.zeros((m.nv, 6))


def _local2global(d: Data, pos, quat, body):
    xpos = _mulMatVec3(d.xmat[body], pos) + d.xpos[body]
    tmp = np.empty(4)
    mju_mulQuat(tmp, d.xquat[body], quat)
    xmat = np.empty(9)
    mju_quat2Mat(xmat, tmp)
    return xpos, xmat


def mj_kinematics(m, d: Data) -> None:
    """Tree forward kinematics (mink/configuration.py:63); SURVEY Appendix A.1."""
    d.xpos[0] = 0
    d.xquat[0] = (1, 0, 0, 0)
    d.xmat[0] = (1, 0, 0, 0, 1, 0, 0, 0, 1)
    qpos = d.qpos
    for i in range(1, m.nbody):
        jntadr, jntnum = int(m.body_jntadr[i]), int(m.body_jntnum[i])
        if jntnum == 1 and m.jnt_type[jntadr] == JNT_FREE:
            qadr = int(m.jnt_qposadr[jntadr])
            xpos = np.array(qpos[qadr:qadr + 3])
            xquat = np.array(qpos[qadr + 3:qadr + 7])
            mju_normalize4(xquat)
            d.xanchor[jntadr] = xpos
            d.xaxis[jntadr] = m.jnt_axis[jntadr]
        else:
            pid = int(m.body_parentid[i])
            mid = int(m.body_mocapid[i])
            if mid >= 0:
                bodypos = d.mocap_pos[mid]
                bodyquat = np.array(d.mocap_quat[mid])
                mju_normalize4(bodyquat)
            else:
                bodypos, bodyquat = m.body_pos[i], m.body_quat[i]
            if pid:
                xpos = _mulMatVec3(d.xmat[pid], bodypos) + d.xpos[pid]
                xquat = np.empty(4)
                mju_mulQuat(xquat, d.xquat[pid], bodyquat)
            else:
                xpos = np.array(bodypos, dtype=np.float64)
                xquat = np.array(bodyquat, dtype=np.float64)
            for jid in range(jntadr, jntadr + jntnum):
                qadr, jtype = int(m.jnt_qposadr[jid]), int(m.jnt_type[jid])
                xaxis = mju_rotVecQuat(m.jnt_axis[jid], xquat)
                xanchor = mju_rotVecQuat(m.jnt_pos[jid], xquat) + xpos
                d.xaxis[jid] = xaxis
                d.xanchor[jid] = xanchor
                if jtype == JNT_SLIDE:
                    xpos = xpos + xaxis * (qpos[qadr] - m.qpos0[qadr])
                elif jtype in (JNT_BALL, JNT_HINGE):
                    qloc = np.empty(4)
                    if jtype == JNT_BALL:
                        qloc[:] = qpos[qadr:qadr + 4]
                        mju_normalize4(qloc)
                    else:
                        mju_axisAngle2Quat(qloc, m.jnt_axis[jid], qpos[qadr] - m.qpos0[qadr])
                    mju_mulQuat(xquat, xquat.copy(), qloc)
                    vec = mju_rotVecQuat(m.jnt_pos[jid], xquat)
                    xpos = xanchor - vec
                else:
                    raise ValueError("free joint must be the only joint of its body")
        mju_normalize4(xquat)
        d.xquat[i] = xquat
        d.xpos[i] = xpos
        mju_quat2Mat(d.xmat[i], xquat)
    for i in range(m.nbody):
        d.xipos[i] = _mulMatVec3(d.xmat[i], m.body_ipos[i]) + d.xpos[i]
    for g in range(m.ngeom):
        d.geom_xpos[g], d.geom_xmat[g] = _local2global(d, m.geom_pos[g], m.geom_quat[g], int(m.geom_bodyid[g]))
    for s in range(m.nsite):
        d.site_xpos[s], d.site_xmat[s] = _local2global(d, m.site_pos[s], m.site_quat[s], int(m.site_bodyid[s]))


def _dofCom(axis, offset):
    res = np.zeros(6)
    if offset is not None:
        res[:3] = axis
        res[3:] = np.cross(axis, offset)
    else:
        res[3:] = axis
    return res


def mj_comPos(m, d: Data) -> None:
    """subtree_com and cdof (mink/configuration.py:64); SURVEY Appendix A.2."""
    d.subtree_com[:] = 0
    for i in range(m.nbody - 1, -1, -1):
        d.subtree_com[i] += d.xipos[i] * m.body_mass[i]
        if i:
            d.subtree_com[int(m.body_parentid[i])] += d.subtree_com[i]
        if m.body_subtreemass[i] < mjMINVAL:
            d.subtree_com[i] = d.xipos[i]
        else:
            d.subtree_com[i] = d.subtree_com[i] * (1.0 / max(mjMINVAL, m.body_subtreemass[i]))
    for j in range(m.njnt):
        da, bi = int(m.jnt_dofadr[j]), int(m.jnt_bodyid[j])
        offset = d.subtree_com[int(m.body_rootid[bi])] - d.xanchor[j]
        jt = int(m.jnt_type[j])
        skip = 0
        if jt == JNT_FREE:
            d.cdof[da:da + 3] = 0
            for i in range(3):
                d.cdof[da + i, 3 + i] = 1
            skip = 3
        if jt in (JNT_FREE, JNT_BALL):
            for i in range(3):
                axis = np.array([d.xmat[bi, i], d.xmat[bi, i + 3], d.xmat[bi, i + 6]])
                d.cdof[da + skip + i] = _dofCom(axis, offset)
        elif jt == JNT_SLIDE:
            d.cdof[da] = _dofCom(d.xaxis[j], None)
        elif jt == JNT_HINGE:
            d.cdof[da] = _dofCom(d.xaxis[j], offset)


def mj_jac(m, d: Data, jacp, jacr, point, body: int) -> None:
    """World-aligned point Jacobian (collision_avoidance_limit.py:69,71); A.3."""
    if jacp is not None:
        jacp[:] = 0
    if jacr is not None:
        jacr[:] = 0
    offset = np.asarray(point) - d.subtree_com[int(m.body_rootid[body])]
    while body and not m.body_dofnum[body]:
        body = int(m.body_parentid[body])
    if not body:
        return
    i = int(m.body_dofadr[body] + m.body_dofnum[body] - 1)
    while i >= 0:
        cdof = d.cdof[i]
        if jacr is not None:
            jacr[:, i] = cdof[:3]
        if jacp is not None:
            jacp[:, i] = cdof[3:] + np.cross(cdof[:3], offset)
        i = int(m.dof_parentid[i])


def mj_jacBody(m, d, jacp, jacr, body):
    mj_jac(m, d, jacp, jacr, d.xpos[body], body)


def mj_jacBodyCom(m, d, jacp, jacr, body):
    mj_jac(m, d, jacp, jacr, d.xipos[body], body)


def mj_jacGeom(m, d, jacp, jacr, geom):
    mj_jac(m, d, jacp, jacr, d.geom_xpos[geom], int(m.geom_bodyid[geom]))


def mj_jacSite(m, d, jacp, jacr, site):
    """mink/constants.py:11-13 → mink/configuration.py:144-145."""
    mj_jac(m, d, jacp, jacr, d.site_xpos[site], int(m.site_bodyid[site]))


def mj_jacSubtreeCom(m, d: Data, jacp, body: int) -> None:
    """mink/tasks/com_task.py:96; SURVEY Appendix A.4."""
    jacp[:] = 0
    jacp_b = np.zeros((3, m.nv))
    for b in range(body, m.nbody):
        if b > body and m.body_parentid[b] < body:
            break
        mj_jacBodyCom(m, d, jacp_b, None, b)
        jacp += jacp_b * m.body_mass[b]
    jacp *= 1.0 / m.body_subtreemass[body]


def mj_differentiatePos(m, qvel, dt, qpos1, qpos2) -> None:
    """qvel = (qpos2 ⊖ qpos1)/dt (posture_task.py:107, configuration_limit.py:100,110)."""
    for j in range(m.njnt):
        padr, vadr, jt = int(m.jnt_qposadr[j]), int(m.jnt_dofadr[j]), int(m.jnt_type[j])
        if jt == JNT_FREE:
            for i in range(3):
                qvel[vadr + i] = (qpos2[padr + i] - qpos1[padr + i]) / dt
            vadr += 3; padr += 3
        if jt in (JNT_FREE, JNT_BALL):
            neg = np.empty(4); dif = np.empty(4)
            mju_negQuat(neg, qpos1[padr:padr + 4])
            mju_mulQuat(dif, neg, qpos2[padr:padr + 4])
            qvel[vadr:vadr + 3] = mju_quat2Vel(dif, dt)
        else:
            qvel[vadr] = (qpos2[padr] - qpos1[padr]) / dt


def mj_integratePos(m, qpos, qvel, dt) -> None:
    """In place q ← q ⊕ v·dt (mink/configuration.py:225,235); A.6."""
    for j in range(m.njnt):
        padr, vadr, jt = int(m.jnt_qposadr[j]), int(m.jnt_dofadr[j]), int(m.jnt_type[j])
        if jt == JNT_FREE:
            for i in range(3):
                qpos[padr + i] += dt * qvel[vadr + i]
            padr += 3; vadr += 3
        if jt in (JNT_FREE, JNT_BALL):
            quat = np.array(qpos[padr:padr + 4])
            mju_quatIntegrate(quat, qvel[vadr:vadr + 3], dt)
            qpos[padr:padr + 4] = quat
        else:
            qpos[padr] += dt * qvel[vadr]


# --------------------------------------------------------------- collisions
def _sphere_sphere(p1, r1, p2, r2, margin):
    """mjraw_SphereSphere: returns list of (dist, pos, normal)."""
    dif = p2 - p1
    cdist = math.sqrt(float(dif @ dif))
    dist = cdist - r1 - r2
    if dist > margin:
        return []
    n = dif.copy()
    if cdist < mjMINVAL:
        n = np.array([1.0, 0.0, 0.0])
    else:
        n = n / cdist
    pos = p1 + n * (r1 + 0.5 * dist)
    return [(dist, pos, n)]


def _capsule_capsule(pos1, mat1, size1, pos2, mat2, size2, margin):
    """mjc_CapsuleCapsule restated: segment–segment closest
    points then sphere–sphere; the parallel case yields up to two contacts."""
    axis1 = np.array([mat1[2], mat1[5], mat1[8]])
    axis2 = np.array([mat2[2], mat2[5], mat2[8]])
    dif = pos1 - pos2
    ma = float(axis1 @ axis1); mb = -float(axis1 @ axis2); mc = float(axis2 @ axis2)
    u = -float(axis1 @ dif); v = float(axis2 @ dif)
    det = ma * mc - mb * mb
    out = []
    if abs(det) >= mjMINVAL:
        x1 = (mc * u - mb * v) / det
        x2 = (ma * v - mb * u) / det
        if x1 > size1[1]:
            x1 = size1[1]; x2 = (v - mb * size1[1]) / mc
        elif x1 < -size1[1]:
            x1 = -size1[1]; x2 = (v + mb * size1[1]) / mc
        if x2 > size2[1]:
            x2 = size2[1]; x1 = (u - mb * size2[1]) / ma
        elif x2 < -size2[1]:
            x2 = -size2[1]; x1 = (u + mb * size2[1]) / ma
        x1 = min(max(x1, -size1[1]), size1[1])
        x2 = min(max(x2, -size2[1]), size2[1])
        out += _sphere_sphere(pos1 + axis1 * x1, size1[0], pos2 + axis2 * x2, size2[0], margin)
    else:
        # parallel axes: test both ends of each capsule against the other segment
        cands = []
        for s in (+1.0, -1.0):
            x1 = s * size1[1]
            x2 = (v - mb * x1) / mc
            if -size2[1] <= x2 <= size2[1]:
                cands.append((x1, x2))
        for s in (+1.0, -1.0):
            x2 = s * size2[1]
            x1 = (u - mb * x2) / ma
            if -size1[1] <= x1 <= size1[1]:
                cands.append((x1, x2))
        # (no end projects inside the other segment ⇒ no contact is generated)
        for x1, x2 in cands[:2]:
            out += _sphere_sphere(pos1 + axis1 * x1, size1[0], pos2 + axis2 * x2, size2[0], margin)
    return out


def _sphere_capsule(pos1, size1, pos2, mat2, size2, margin):
    axis = np.array([mat2[2], mat2[5], mat2[8]])
    x = float(axis @ (pos1 - pos2))
    x = min(max(x, -size2[1]), size2[1])
    return _sphere_sphere(pos1, size1[0], pos2 + axis * x, size2[0], margin)


def _plane_sphere(pos1, mat1, pos2, size2, margin):
    n = np.array([mat1[2], mat1[5], mat1[8]])
    cdist = float(n @ (pos2 - pos1))
    dist = cdist - size2[0]
    if dist > margin:
        return []
    pos = pos2 - n * (size2[0] + 0.5 * dist)
    return [(dist, pos, n)]


def _plane_capsule(pos1, mat1, pos2, mat2, size2, margin):
    axis = np.array([mat2[2], mat2[5], mat2[8]])
    out = []
    for s in (+1.0, -1.0):
        out += _plane_sphere(pos1, mat1, pos2 + axis * s * size2[1], size2, margin)
    return out


# ---- box / cylinder against plane, sphere and capsule.  MuJoCo's native routines (engine_collision_primitive.c
# mjc_PlaneBox / mjc_PlaneCylinder / mjc_SphereCylinder, engine_collision_box.c mjc_SphereBox / mjc_CapsuleBox)
# return the Euclidean distance between the two convex shapes for separated geoms; these restate that distance
# exactly (and are checked against a brute-force minimisation in tests/test_oracle_collision_shapes.py).  Where the
# closest pair is not unique (an edge parallel to a face) MuJoCo reports several equally distant contacts and
# mj_geomDistance keeps the first: the tie rule here is our own and says so.
def _mat3(mat9):
    return np.asarray(mat9, dtype=np.float64).reshape(3, 3)


def _plane_box(pos1, mat1, pos2, mat2, size2, margin):
    """Lowest corner of the box (mjc_PlaneBox keeps the corners below the centre; the lowest has the smallest
    distance).  A face or edge parallel to the plane ties towards the −size corner, MuJoCo's enumeration order."""
    n = np.array([mat1[2], mat1[5], mat1[8]])
    R = _mat3(mat2)
    nb = R.T @ n
    vec = np.where(nb < 0.0, size2[:3], -np.asarray(size2[:3]))
    dist = float(n @ (pos2 - pos1)) + float(nb @ vec)
    if dist > margin:
        return []
    return [(dist, pos2 + R @ vec - n * (0.5 * dist), n)]


def _plane_cylinder(pos1, mat1, pos2, mat2, size2, margin):
    """Lowest rim point of the cylinder; a cap parallel to the plane ties to the centre of that cap."""
    n = np.array([mat1[2], mat1[5], mat1[8]])
    axis = np.array([mat2[2], mat2[5], mat2[8]])
    c = float(n @ axis)
    radial = n - c * axis
    rl = math.sqrt(float(radial @ radial))
    pt = pos2 - axis * (-size2[1] if c < 0.0 else size2[1])
    if rl > mjMINVAL:
        pt = pt - radial * (size2[0] / rl)
    dist = float(n @ (pt - pos1))
    if dist > margin:
        return []
    return [(dist, pt - n * (0.5 * dist), n)]


GEOM_MESH = 7
_CONVEX = (GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH)


def _plane_mesh(pos1, mat1, pos2, mat2, hull, margin):
    """Lowest vertex of the hull against the plane normal (mjc_PlaneConvex keeps the deepest point)."""
    n = np.array([mat1[2], mat1[5], mat1[8]])
    R = _mat3(mat2)
    hull = np.asarray(hull, dtype=np.float64).reshape(-1, 3)
    pt = pos2 + R @ hull[int(np.argmax(hull @ (R.T @ -n)))]
    dist = float(n @ (pt - pos1))
    if dist > margin:
        return []
    return [(dist, pt - n * (0.5 * dist), n)]


def _plane_ellipsoid(pos1, mat1, pos2, mat2, size2, margin):
    """Lowest point of the ellipsoid: its support point against the plane normal (mjc_PlaneConvex with the ellipsoid's
    support mapping)."""
    n = np.array([mat1[2], mat1[5], mat1[8]])
    R = _mat3(mat2)
    e = np.asarray(size2[:3]) * (R.T @ n)
    pt = pos2 - R @ (np.asarray(size2[:3]) * e) / math.sqrt(float(e @ e))
    dist = float(n @ (pt - pos1))
    if dist > margin:
        return []
    return [(dist, pt - n * (0.5 * dist), n)]


def _ball_box_local(p, r, s, margin):
    """Ball of radius r centred at p (box frame) against the box ±s (mjc_SphereBox): clamp the centre onto the
    box; a centre inside the box leaves through the nearest face (order −x,+x,−y,+y,−z,+z, strict <)."""
    s = np.asarray(s[:3], dtype=np.float64)
    cl = np.minimum(np.maximum(p, -s), s)
    d = cl - p
    dl = math.sqrt(float(d @ d))
    if dl - r > margin:
        return []
    if dl > mjMINVAL:
        n = d / dl
        dist = dl - r
        return [(dist, p + n * (r + 0.5 * dist), n)]
    face = s - np.abs(p)
    k = 0
    for i in (1, 2):
        if face[i] < face[k]:
            k = i
    n = np.zeros(3)
    n[k] = 1.0 if p[k] <= 0.0 else -1.0
    closest = float(face[k])
    return [(-closest - r, p + n * (0.5 * (r - closest)), n)]


def _seg_box_param(c, a, l, s):
    """argmin over |t| ≤ l of dist(c + t·a, box ±s): the derivative g(t) = Σ aᵢ·excessᵢ(t) of ½·dist² is piecewise
    linear and non-decreasing; scan the sorted breakpoints for the interval holding its root.  A flat stretch
    (axis parallel to a face / through the box) takes its midpoint; g > 0 at −l takes −l, g < 0 at +l takes +l."""
    s = np.asarray(s[:3], dtype=np.float64)

    def g(t, on=-1):
        # `on`: the axis whose slab plane t lies on — its excess is 0 by construction (inside the box g must be
        # exactly 0 at the entry and exit points, or rounding noise would pick the point of the chord)
        p = c + t * a
        e = p - np.minimum(np.maximum(p, -s), s)
        if on >= 0:
            e[on] = 0.0
        v = float(a @ e)
        return 0.0 if abs(v) <= tol else v     # parallel up to rounding is parallel: the flat-stretch rule, not the noise's pick

    tol = 1e-13 * (l + float(np.max(np.abs(c))) + float(np.max(s)))

    cand = [(-l, g(-l)), (l, g(l))]
    for i in range(3):
        if abs(a[i]) >= mjMINVAL:
            for e in (-s[i], s[i]):
                tb = (e - c[i]) / a[i]
                if -l < tb < l:
                    cand.append((tb, g(tb, i)))
    cand.sort(key=lambda x: x[0])
    ts = [x[0] for x in cand]
    gs = [x[1] for x in cand]
    if gs[0] > 0.0:
        return -l
    if gs[-1] < 0.0:
        return l
    lo = max(i for i in range(len(ts)) if gs[i] <= 0.0)       # last point with g ≤ 0
    hi = min(i for i in range(len(ts)) if gs[i] >= 0.0)       # first point with g ≥ 0
    tL, gL, tR, gR = ts[lo], gs[lo], ts[hi], gs[hi]
    if gR - gL > 0.0:
        return tL + (tR - tL) * (-gL / (gR - gL))
    return 0.5 * (tL + tR)


def _ball_cylinder_local(p, r, rad, half, margin):
    """Ball against a cylinder (its frame; mjc_SphereCylinder): side / cap / rim; a centre inside leaves through
    the nearer of the side wall and the caps."""
    rho = math.hypot(p[0], p[1])
    sc = rad / rho if rho > rad else 1.0
    cl = np.array([p[0] * sc, p[1] * sc, min(max(p[2], -half), half)])
    d = cl - p
    dl = math.sqrt(float(d @ d))
    if dl - r > margin:
        return []
    if dl > mjMINVAL:
        n = d / dl
        dist = dl - r
        return [(dist, p + n * (r + 0.5 * dist), n)]
    fr, fz = rad - rho, half - abs(p[2])
    if fz < fr:
        closest, n = fz, np.array([0.0, 0.0, 1.0 if p[2] <= 0.0 else -1.0])
    else:
        closest = fr
        n = np.array([-p[0] / rho, -p[1] / rho, 0.0]) if rho > mjMINVAL else np.array([-1.0, 0.0, 0.0])
    return [(-closest - r, p + n * (0.5 * (r - closest)), n)]


def _to_world(cons, R, o):
    return [(dist, o + R @ pos, R @ n) for dist, pos, n in cons]


def _sphere_box(pos1, size1, pos2, mat2, size2, margin):
    R = _mat3(mat2)
    return _to_world(_ball_box_local(R.T @ (pos1 - pos2), size1[0], size2, margin), R, pos2)


def _sphere_cylinder(pos1, size1, pos2, mat2, size2, margin):
    R = _mat3(mat2)
    return _to_world(_ball_cylinder_local(R.T @ (pos1 - pos2), size1[0], size2[0], size2[1], margin), R, pos2)


def _capsule_box(pos1, mat1, size1, pos2, mat2, size2, margin):
    R = _mat3(mat2)
    c = R.T @ (pos1 - pos2)
    a = R.T @ np.array([mat1[2], mat1[5], mat1[8]])
    t = _seg_box_param(c, a, size1[1], size2)
    return _to_world(_ball_box_local(c + t * a, size1[0], size2, margin), R, pos2)


def _seg_seg(p1, q1, p2, q2):
    """Closest points of two segments of non-zero length (clamped line–line solution; Ericson, Real-Time Collision
    Detection §5.1.9).  Returns (squared distance, point on 1, point on 2)."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = float(d1 @ d1), float(d2 @ d2), float(d2 @ r)
    c, b = float(d1 @ r), float(d1 @ d2)
    den = a * e - b * b
    sp = min(max((b * f - c * e) / den, 0.0), 1.0) if den > mjMINVAL * a * e else 0.0
    t = (b * sp + f) / e
    if t < 0.0:
        t, sp = 0.0, min(max(-c / a, 0.0), 1.0)
    elif t > 1.0:
        t, sp = 1.0, min(max((b - c) / a, 0.0), 1.0)
    x1, x2 = p1 + sp * d1, p2 + t * d2
    return float((x1 - x2) @ (x1 - x2)), x1, x2


def _box_edges(s):
    """The 12 edges of the box ±s as (start, end) pairs: axis e, the four sign choices of the other two axes."""
    out = []
    for e in range(3):
        j, k = (e + 1) % 3, (e + 2) % 3
        for sj in (-1.0, 1.0):
            for sk in (-1.0, 1.0):
                a = np.zeros(3); b = np.zeros(3)
                a[e], b[e] = -s[e], s[e]
                a[j] = b[j] = sj * s[j]
                a[k] = b[k] = sk * s[k]
                out.append((a, b))
    return out


def _box_box(pos1, mat1, size1, pos2, mat2, size2, margin):
    """Two boxes (MuJoCo: mjc_BoxBox, engine_collision_box.c).  Separated boxes: the exact Euclidean distance — the
    minimum over vertex–box (both ways) and edge–edge pairs, which covers every closest-feature combination of two
    convex polytopes.  Overlapping boxes (no separating axis among the 15 of the SAT): the smallest overlap and its
    axis, with the contact placed midway between the two centres' projections — MuJoCo clips faces there and reports
    several contacts; ours is one contact with the SAT depth (documented approximation: a collision-AVOIDANCE limit
    only needs a sane normal once d ≤ d_min)."""
    R1, R2 = _mat3(mat1), _mat3(mat2)
    sa, sb = np.asarray(size1[:3], dtype=np.float64), np.asarray(size2[:3], dtype=np.float64)
    R = R2.T @ R1                      # A's axes in B's frame
    c = R2.T @ (pos1 - pos2)           # A's centre in B's frame
    # ---- separating-axis test: separation along unit axis L = |c·L| − (rA + rB)
    best_sep, best_axis = -np.inf, None
    axes = [np.eye(3)[i] for i in range(3)] + [R[:, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            ax = np.cross(R[:, i], np.eye(3)[j])
            n = math.sqrt(float(ax @ ax))
            if n > 1e-9:
                axes.append(ax / n)
    for L in axes:
        ra = float(np.abs(R.T @ L) @ sa)
        rb = float(np.abs(L) @ sb)
        sep = abs(float(c @ L)) - (ra + rb)
        if sep > best_sep:
            best_sep, best_axis = sep, (L if float(c @ L) <= 0.0 else -L)      # from A towards B
    if best_sep <= 0.0:
        dist = best_sep
        if dist > margin:
            return []
        n = best_axis
        ca = float(c @ n) + float(np.abs(R.T @ n) @ sa)       # A's extent towards B along n
        cb = -float(np.abs(n) @ sb)                            # B's extent towards A along n
        mid = 0.5 * c + n * (0.5 * (ca + cb) - 0.5 * float(c @ n))
        return _to_world([(dist, mid, n)], R2, pos2)
    # ---- separated: closest features
    best = (np.inf, None, None)
    for i in range(8):
        sg = np.array([1.0 if i & 1 else -1.0, 1.0 if i & 2 else -1.0, 1.0 if i & 4 else -1.0])
        va = c + R @ (sg * sa)                                  # vertex of A against box B
        cl = np.minimum(np.maximum(va, -sb), sb)
        d2 = float((va - cl) @ (va - cl))
        if d2 < best[0]:
            best = (d2, va, cl)
        ub = sg * sb                                            # vertex of B against box A (A's frame)
        ul = R.T @ (ub - c)
        cl = np.minimum(np.maximum(ul, -sa), sa)
        d2 = float((ul - cl) @ (ul - cl))
        if d2 < best[0]:
            best = (d2, c + R @ cl, ub)
    for ea0, ea1 in _box_edges(sa):
        pa, qa = c + R @ ea0, c + R @ ea1
        for eb0, eb1 in _box_edges(sb):
            d2, x1, x2 = _seg_seg(pa, qa, eb0, eb1)
            if d2 < best[0]:
                best = (d2, x1, x2)
    dist = math.sqrt(best[0])
    if dist > margin:
        return []
    n = (best[2] - best[1]) / dist if dist > mjMINVAL else (best_axis)
    return _to_world([(dist, 0.5 * (best[1] + best[2]), n)], R2, pos2)


def _cylinder_closest(p, rad, half):
    """Closest point of the solid cylinder (radius rad, half height half, axis z) to p."""
    rho = math.hypot(p[0], p[1])
    sc = rad / rho if rho > rad else 1.0
    return np.array([p[0] * sc, p[1] * sc, min(max(p[2], -half), half)])


def _capsule_cylinder(pos1, mat1, size1, pos2, mat2, size2, margin):
    """Capsule against cylinder (MuJoCo: libccd).  The squared distance from the capsule axis c + t·a to the solid
    cylinder is convex in t, so its derivative g(t) = a·(p(t) − closest(p(t))) is non-decreasing: 64 bisection steps
    on g locate the closest axis point to the last bit, then ball-against-cylinder.  Axis through the cylinder
    (g = 0 on an interval): the bisection lands on the interval's lower end."""
    R = _mat3(mat2)
    c = R.T @ (pos1 - pos2)
    a = R.T @ np.array([mat1[2], mat1[5], mat1[8]])
    l, rad, half = size1[1], size2[0], size2[1]

    def g(t):
        p = c + t * a
        return float(a @ (p - _cylinder_closest(p, rad, half)))

    lo, hi = -l, l
    if g(lo) >= 0.0:
        t = lo
    elif g(hi) <= 0.0:
        t = hi
    else:
        for _ in range(64):
            mid = 0.5 * (lo + hi)
            if g(mid) < 0.0:
                lo = mid
            else:
                hi = mid
        t = hi
    return _to_world(_ball_cylinder_local(c + t * a, size1[0], rad, half, margin), R, pos2)


def mj_geomDistance(m, d: Data, geom1: int, geom2: int, distmax: float, fromto) -> float:
    """Smallest signed distance between two geoms and the connecting segment
    (mink/limits/collision_avoidance_limit.py:219); SURVEY Appendix A.8.
    Restated: plane/sphere/capsule pairs, box against plane/sphere/capsule/box, cylinder against plane/sphere/capsule,
    plane–ellipsoid; every other pair of convex primitives (cylinder–box, cylinder–cylinder, ellipsoid–*) through the
    general convex routine of oracle/gjk.py."""
    g1, g2 = int(geom1), int(geom2)
    t1, t2 = int(m.geom_type[g1]), int(m.geom_type[g2])
    flip = t1 > t2
    if flip:
        g1, g2, t1, t2 = g2, g1, t2, t1
    p1, p2 = d.geom_xpos[g1], d.geom_xpos[g2]
    R1, R2 = d.geom_xmat[g1], d.geom_xmat[g2]
    s1, s2 = m.geom_size[g1], m.geom_size[g2]
    if not (m.geom_valid[g1] and m.geom_valid[g2]):
        raise NotImplementedError("geom needs mesh data")
    # a mesh geom is its convex hull: the hull's vertices (geom frame) stand in for the size
    if t1 == GEOM_MESH:
        s1 = m.mesh_hull(g1)
    if t2 == GEOM_MESH:
        s2 = m.mesh_hull(g2)
    if (t1, t2) == (GEOM_CAPSULE, GEOM_CAPSULE):
        cons = _capsule_capsule(p1, R1, s1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_SPHERE, GEOM_SPHERE):
        cons = _sphere_sphere(p1, s1[0], p2, s2[0], distmax)
    elif (t1, t2) == (GEOM_SPHERE, GEOM_CAPSULE):
        cons = _sphere_capsule(p1, s1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_PLANE, GEOM_SPHERE):
        cons = _plane_sphere(p1, R1, p2, s2, distmax)
    elif (t1, t2) == (GEOM_PLANE, GEOM_CAPSULE):
        cons = _plane_capsule(p1, R1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_PLANE, GEOM_BOX):
        cons = _plane_box(p1, R1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_PLANE, GEOM_CYLINDER):
        cons = _plane_cylinder(p1, R1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_SPHERE, GEOM_BOX):
        cons = _sphere_box(p1, s1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_SPHERE, GEOM_CYLINDER):
        cons = _sphere_cylinder(p1, s1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_CAPSULE, GEOM_BOX):
        cons = _capsule_box(p1, R1, s1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_CAPSULE, GEOM_CYLINDER):
        cons = _capsule_cylinder(p1, R1, s1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_BOX, GEOM_BOX):
        cons = _box_box(p1, R1, s1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_PLANE, GEOM_ELLIPSOID):
        cons = _plane_ellipsoid(p1, R1, p2, R2, s2, distmax)
    elif (t1, t2) == (GEOM_PLANE, GEOM_MESH):
        cons = _plane_mesh(p1, R1, p2, R2, s2, distmax)
    elif t1 in _CONVEX and t2 in _CONVEX:
        # no native pair routine in MuJoCo either: the general convex collider (oracle/gjk.py states what it approximates)
        from . import gjk
        c = gjk.convex_distance(t1, s1, p1, _mat3(R1), t2, s2, p2, _mat3(R2), distmax)
        cons = [c] if c is not None else []
    else:
        raise NotImplementedError(f"geom pair types ({t1},{t2}) not restated")
    if fromto is not None:
        fromto[:] = 0
    if not cons:
        return distmax
    dist, pos, n = min(cons, key=lambda c: c[0])
    if fromto is not None:
        s = -1.0 if flip else 1.0
        fromto[0:3] = pos - n * (0.5 * s * dist)
        fromto[3:6] = pos + n * (0.5 * s * dist)
    return dist
