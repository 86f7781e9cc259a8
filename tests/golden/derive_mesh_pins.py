#!/usr/bin/env python
"""Second, independent derivation of what MuJoCo's model compiler makes of a `<geom type="capsule" mesh=...>` — for the
constants typed into tests/test_mjcf_pinned.py.  Deliberately shares NOTHING with mink_amd/mjcf.py / mink_amd/meshes.py:
its own OBJ / STL readers, plain Python loops over the triangles (no vectorised formulas), a Jacobi eigen-solver written out
here, quaternions composed by hand.  Needs /root/reference (the mesh files); its printed output is what was typed into the
test, so the committed model fixtures (tests/golden/models/**, written by the reader) are checked against numbers that did
not come from the reader.

The RULES are the same upstream-recalled ones (user_mesh.cc, mujoco 3.1.x — unpinned against the wheel, see meshes.py):
legacy mesh inertia (|pyramid volumes|, apex = area-weighted surface centroid for volume / centre of mass, apex = centre of
mass for the second moments), principal axes ordered by decreasing inertia, the box of the same inertia, capsule radius =
mean of the two short half-sizes, half-length = long half-size − radius / 2; geom frame = geom's own pos / quat ∘ mesh frame.

    python tests/golden/derive_mesh_pins.py
"""
import math
import re
import struct

EX = "/root/reference/examples"


def read_obj(path):
    verts, faces = [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            verts.append((float(t[1]), float(t[2]), float(t[3])))
        elif t[0] == "f":
            idx = [int(re.split("/", x)[0]) for x in t[1:]]
            idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
            for k in range(1, len(idx) - 1):
                faces.append((idx[0], idx[k], idx[k + 1]))
    return verts, faces


def read_stl(path):
    raw = open(path, "rb").read()
    n = struct.unpack_from("<I", raw, 80)[0]
    assert len(raw) == 84 + 50 * n, "binary STL expected"
    verts, faces = [], []
    for k in range(n):
        f = struct.unpack_from("<12f", raw, 84 + 50 * k)
        base = len(verts)
        verts += [f[3:6], f[6:9], f[9:12]]
        faces.append((base, base + 1, base + 2))
    return verts, faces


def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def legacy_mass_properties(verts, faces):
    # pass 0: area-weighted centroid of the surface
    A, cen = 0.0, [0.0, 0.0, 0.0]
    for i, j, k in faces:
        a, b, c = verts[i], verts[j], verts[k]
        n = cross(sub(b, a), sub(c, a))
        ar = 0.5 * math.sqrt(dot(n, n))
        A += ar
        for d in range(3):
            cen[d] += ar * (a[d] + b[d] + c[d]) / 3.0
    apex = tuple(x / A for x in cen)
    # pass 1: |volume| and centroid of every pyramid (apex, a, b, c)
    V, com = 0.0, [0.0, 0.0, 0.0]
    for i, j, k in faces:
        a, b, c = sub(verts[i], apex), sub(verts[j], apex), sub(verts[k], apex)
        vol = abs(dot(a, cross(b, c))) / 6.0
        V += vol
        for d in range(3):
            com[d] += vol * (a[d] + b[d] + c[d]) / 4.0
    com = tuple(apex[d] + com[d] / V for d in range(3))
    # pass 2: second moments about the centre of mass; tetrahedron with one vertex at the origin:
    #   ∫ x xᵀ dV = V/20 · (Σ p pᵀ + (Σ p)(Σ p)ᵀ) over its other three vertices p
    C = [[0.0] * 3 for _ in range(3)]
    for i, j, k in faces:
        p = [sub(verts[i], com), sub(verts[j], com), sub(verts[k], com)]
        vol = abs(dot(p[0], cross(p[1], p[2]))) / 6.0
        s = tuple(p[0][d] + p[1][d] + p[2][d] for d in range(3))
        for r in range(3):
            for c_ in range(3):
                C[r][c_] += vol / 20.0 * (p[0][r] * p[0][c_] + p[1][r] * p[1][c_] + p[2][r] * p[2][c_] + s[r] * s[c_])
    return V, com, C


def jacobi_eig(M):
    """Eigenvalues / eigenvectors (columns) of a symmetric 3x3 by cyclic Jacobi rotations."""
    a = [row[:] for row in M]
    v = [[1.0 if i == j else 0.0 for j in range(3)] for i in range(3)]
    for _ in range(60):
        off = abs(a[0][1]) + abs(a[0][2]) + abs(a[1][2])
        if off < 1e-30:
            break
        for p, q in ((0, 1), (0, 2), (1, 2)):
            if abs(a[p][q]) < 1e-300:
                continue
            th = 0.5 * math.atan2(2.0 * a[p][q], a[q][q] - a[p][p])
            c, s = math.cos(th), math.sin(th)
            for k in range(3):
                akp, akq = a[k][p], a[k][q]
                a[k][p], a[k][q] = c * akp - s * akq, s * akp + c * akq
            for k in range(3):
                apk, aqk = a[p][k], a[q][k]
                a[p][k], a[q][k] = c * apk - s * aqk, s * apk + c * aqk
            for k in range(3):
                vkp, vkq = v[k][p], v[k][q]
                v[k][p], v[k][q] = c * vkp - s * vkq, s * vkp + c * vkq
    return [a[0][0], a[1][1], a[2][2]], v


def quat_mul(a, b):
    return (a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0])


def quat_rot(q, v):
    w, x, y, z = q
    R = ((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)),
         (2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)),
         (2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)))
    return tuple(dot(R[r], v) for r in range(3))


def fitted_capsule(path, scale, geom_pos=(0, 0, 0), geom_quat=(1, 0, 0, 0)):
    verts, faces = (read_obj if path.endswith(".obj") else read_stl)(path)
    verts = [(v[0] * scale[0], v[1] * scale[1], v[2] * scale[2]) for v in verts]
    V, com, C = legacy_mass_properties(verts, faces)
    tr = C[0][0] + C[1][1] + C[2][2]
    inertia = [[(tr if r == c else 0.0) - C[r][c] for c in range(3)] for r in range(3)]
    w, vec = jacobi_eig(inertia)
    order = sorted(range(3), key=lambda i: -w[i])                 # decreasing principal inertia: the long axis comes last (z)
    w = [w[i] for i in order]
    half = [math.sqrt(max(0.0, 1.5 * (w[1] + w[2] - w[0]) / V)), math.sqrt(max(0.0, 1.5 * (w[0] + w[2] - w[1]) / V)),
            math.sqrt(max(0.0, 1.5 * (w[0] + w[1] - w[2]) / V))]
    radius = 0.5 * (half[0] + half[1])
    halflen = half[2] - 0.5 * radius
    zaxis_mesh = tuple(vec[r][order[2]] for r in range(3))        # long axis of the solid in the mesh file's frame
    n = math.sqrt(sum(x * x for x in geom_quat))
    gq = tuple(x / n for x in geom_quat)
    pos = tuple(geom_pos[d] + quat_rot(gq, com)[d] for d in range(3))
    zaxis = quat_rot(gq, zaxis_mesh)
    return dict(volume=V, radius=radius, halflen=halflen, pos=pos, zaxis=zaxis, nvert=len(verts))


def extremes(path, scale):
    verts, _ = (read_obj if path.endswith(".obj") else read_stl)(path)
    verts = sorted(set((v[0] * scale[0], v[1] * scale[1], v[2] * scale[2]) for v in verts))
    out = {}
    for d, name in enumerate("xyz"):
        out["min_" + name] = min(verts, key=lambda v: (v[d], v))
        out["max_" + name] = max(verts, key=lambda v: (v[d], v))
    return len(verts), out


if __name__ == "__main__":
    mm = (0.001, 0.001, 0.001)
    cases = [
        # left_hand.xml:8 (mesh scale 0.001), :149 `<geom name="first_3" class="plastic_collision" type="capsule" mesh="f_distal_pst"/>`
        ("shadow first_3", EX + "/shadow_hand/assets/f_distal_pst.obj", mm, (0, 0, 0), (1, 0, 0, 0)),
        # :263 `<geom name="thumb_3" ... type="capsule" mesh="th_distal_pst"/>`
        ("shadow thumb_3", EX + "/shadow_hand/assets/th_distal_pst.obj", mm, (0, 0, 0), (1, 0, 0, 0)),
        # aloha.xml:12 (scale 0.001), :86-87 class collision = capsule, :123 `<geom class="collision" mesh="vx300s_4_upper_forearm"/>`
        ("aloha left/upper_forearm_link", EX + "/aloha/assets/vx300s_4_upper_forearm.stl", mm, (0, 0, 0), (1, 0, 0, 0)),
        # :117 `<geom quat="1 0 0 1" class="collision" mesh="vx300s_3_upper_arm"/>`
        ("aloha left/upper_arm_link", EX + "/aloha/assets/vx300s_3_upper_arm.stl", mm, (0, 0, 0), (1, 0, 0, 1)),
        # :111 `<geom pos="0 0 -0.003" quat="1 0 0 1" mesh="vx300s_2_shoulder" class="collision"/>`
        ("aloha left/shoulder_link", EX + "/aloha/assets/vx300s_2_shoulder.stl", mm, (0, 0, -0.003), (1, 0, 0, 1)),
    ]
    for name, path, scale, gp, gq in cases:
        r = fitted_capsule(path, scale, gp, gq)
        print("%s:\n  radius %.12g  half-length %.12g\n  pos (%.12g, %.12g, %.12g)\n  axis (%.9f, %.9f, %.9f)  [volume %.6e, %d file vertices]"
              % ((name, r["radius"], r["halflen"]) + r["pos"] + r["zaxis"] + (r["volume"], r["nvert"])))
    # left_hand.xml:101 `<geom class="plastic_collision" type="mesh" mesh="forearm_collision"/>`
    n, ex = extremes(EX + "/shadow_hand/assets/forearm_collision.obj", mm)
    print("shadow forearm_collision.obj: %d distinct vertices" % n)
    for k, v in ex.items():
        print("  %s (%.9g, %.9g, %.9g)" % ((k,) + v))
    try:
        from scipy.spatial import ConvexHull
        import numpy as np
        verts, _ = read_obj(EX + "/shadow_hand/assets/forearm_collision.obj")
        print("  hull of the file's vertices (qhull): %d vertices" % len(ConvexHull(np.array(sorted(set(verts)))).vertices))
    except ImportError:
        pass
    # Body mass and centre of mass from mesh geoms (round 5): wonik_allegro/left_hand.xml — class "allegro_left" sets
    # `<geom density="800"/>` (:10), the visual classes are `type="mesh"` (:13), no body has an <inertial>, every collision geom has
    # `mass="0"` (:38); so a finger link weighs 800 x the legacy volume of its visual mesh (no mesh scale in this file) at the
    # mesh's centre of mass.  :136-138 rf_proximal = link_1.0, :148-149 rf_tip = link_3.0_tip at pos "0 0 0.0267" (:30).
    for name, file, off in (("allegro rf_proximal (link_1.0)", "link_1.0.stl", (0, 0, 0)),
                            ("allegro rf_tip (link_3.0_tip)", "link_3.0_tip.stl", (0, 0, 0.0267)),
                            ("allegro rf_medial (link_2.0)", "link_2.0.stl", (0, 0, 0))):
        verts, faces = read_stl(EX + "/wonik_allegro/assets/" + file)
        V, com, _ = legacy_mass_properties(verts, faces)
        print("%s:\n  mass %.12g  (volume %.12g)\n  ipos (%.12g, %.12g, %.12g)" % (name, 800.0 * V, V, com[0] + off[0], com[1] + off[1], com[2] + off[2]))
