"""mink_amd/meshes.py — the restated slice of MuJoCo's mesh compiler (inertial frame, inertia box, primitive fit, hull) —
pinned by properties that do not need the wheel: exact mass properties of solids tessellated on the spot, invariance under
rigid motions, file readers on files written here; FlatModel.from_mjmodel's hull extraction on an object with the real
MjModel mesh field layout.  CPU only."""

import os
import struct

import numpy as np
import pytest

from mink_amd import meshes
from mink_amd.flatmodel import GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_SPHERE, FlatModel


def _box(hx, hy, hz):
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=float)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = []
    for a, b, c, d in quads:
        f += [(a, b, c), (a, c, d)]
    return v, np.array(f)


def _cylinder(r, h, n=256):
    ang = 2 * np.pi * np.arange(n) / n
    ring = np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)
    v = np.concatenate([np.c_[ring, np.full(n, -h)], np.c_[ring, np.full(n, h)], [[0, 0, -h], [0, 0, h]]])
    f = []
    for i in range(n):
        j = (i + 1) % n
        f += [(i, j, n + j), (i, n + j, n + i), (2 * n, j, i), (2 * n + 1, n + i, n + j)]
    return v, np.array(f)


def _random_rotation(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_box_mass_properties_and_fit_are_exact():
    rng = np.random.default_rng(0)
    hx, hy, hz = 0.03, 0.05, 0.11
    v, f = _box(hx, hy, hz)
    R, t = _random_rotation(rng), rng.normal(size=3)
    pos, A, box, vol = meshes.inertial_frame(v @ R.T + t, f)
    np.testing.assert_allclose(vol, 8 * hx * hy * hz, rtol=1e-12)
    np.testing.assert_allclose(pos, t, atol=1e-13)
    # principal inertias decreasing ⇒ box half-sizes increasing: (x, y, z) = (smallest … largest extent)
    np.testing.assert_allclose(box, [hx, hy, hz], rtol=1e-10)
    # axes: the box's own, up to sign
    np.testing.assert_allclose(np.abs(A.T @ R), np.eye(3), atol=1e-9)
    assert np.linalg.det(A) > 0
    np.testing.assert_allclose(meshes.fit_primitive(GEOM_BOX, box), box)
    np.testing.assert_allclose(meshes.fit_primitive(GEOM_SPHERE, box)[0], (hx + hy + hz) / 3)
    cap = meshes.fit_primitive(GEOM_CAPSULE, box)
    np.testing.assert_allclose(cap[:2], [(hx + hy) / 2, hz - (hx + hy) / 4])
    np.testing.assert_allclose(meshes.fit_primitive(GEOM_CYLINDER, box)[:2], [(hx + hy) / 2, hz])
    # inward-wound triangles describe the same solid
    pos2, _, box2, vol2 = meshes.inertial_frame(v, f[:, ::-1])
    np.testing.assert_allclose([vol2, *box2], [vol, *box], rtol=1e-12)


def test_cylinder_inertia_box():
    r, h = 0.02, 0.09
    v, f = _cylinder(r, h, 720)
    pos, A, box, vol = meshes.inertial_frame(v, f)
    np.testing.assert_allclose(vol, np.pi * r * r * 2 * h, rtol=1e-4)
    # solid cylinder: I_z = m r²/2, I_x = m (3 r² + (2h)²)/12 ⇒ box with the same inertia: sx = sy = r·√3/2, sz = h
    np.testing.assert_allclose(box, [r * np.sqrt(3) / 2, r * np.sqrt(3) / 2, h], rtol=2e-4)
    assert abs(abs(A[2, 2]) - 1.0) < 1e-9                       # long axis → z


def test_mesh_files_and_asset(tmp_path):
    v, f = _box(0.01, 0.02, 0.04)
    tri = v[f]
    stl = tmp_path / "b.stl"
    with open(stl, "wb") as fh:
        fh.write(b"\0" * 80 + struct.pack("<I", len(tri)))
        for t in tri:
            fh.write(struct.pack("<12fH", 0, 0, 0, *t.reshape(-1), 0))
    obj = tmp_path / "b.obj"
    with open(obj, "w") as fh:
        for p in v * 1000.0:
            fh.write("v %.9g %.9g %.9g\n" % tuple(p))
        for a, b, c in f:
            fh.write(f"f {a + 1}//1 {b + 1}//1 {c + 1}//1\n")
    a1 = meshes.MeshAsset(str(stl))
    a2 = meshes.MeshAsset(str(obj), scale=(0.001, 0.001, 0.001))
    for a in (a1, a2):
        assert len(a.vert) == 8 and len(a.hull_vert) == 8        # an STL's repeated corners are merged
        np.testing.assert_allclose(a.boxsz, [0.01, 0.02, 0.04], rtol=1e-6)
        np.testing.assert_allclose(np.sort(np.abs(a.vert), axis=0)[-1], [0.01, 0.02, 0.04], rtol=1e-6)
    with pytest.raises(meshes.MeshError):
        meshes.load_mesh_file(str(tmp_path / "b.msh"))


def test_from_mjmodel_takes_the_hull_from_mesh_graph():
    """mjModel layout: mesh_vert float32 (nmeshvert, 3); mesh_graph at mesh_graphadr[k] = [numvert, numface,
    vert_edgeadr[numvert], vert_globalid[numvert], edge_localid[...], face_globalid[3·numface]]."""
    class Raw:
        pass
    m = Raw()
    m.ngeom, m.nmesh = 3, 2
    m.geom_type = np.array([7, 3, 7], dtype=np.int32)
    m.geom_dataid = np.array([1, -1, 0], dtype=np.int32)
    m.mesh_vertadr = np.array([0, 5], dtype=np.int32); m.mesh_vertnum = np.array([5, 4], dtype=np.int32)
    m.mesh_vert = np.arange(27, dtype=np.float32).reshape(9, 3)
    # mesh 0: hull = vertices 0, 2, 3, 4 (vertex 1 interior); mesh 1: no graph ⇒ all four vertices
    g0 = [4, 4] + [0, 0, 0, 0] + [0, 2, 3, 4] + [0] * (4 + 12) + [0] * 12
    m.mesh_graph = np.array(g0, dtype=np.int32)
    m.mesh_graphadr = np.array([0, -1], dtype=np.int32)
    out = FlatModel._hulls_of_mjmodel(m)
    assert out["geom_dataid"].tolist() == [0, -1, 1]            # renumbered in geom order
    assert out["mesh_vertnum"].tolist() == [4, 4] and out["mesh_vertadr"].tolist() == [0, 4]
    np.testing.assert_array_equal(out["mesh_vert"][:4], m.mesh_vert[5:9])
    np.testing.assert_array_equal(out["mesh_vert"][4:], m.mesh_vert[[0, 2, 3, 4]])


def _l_prism(a=0.3, b=0.1, h=0.05):
    """A non-convex solid: the L-shaped polygon (0,0) (a,0) (a,b) (b,b) (b,a) (0,a) extruded over z ∈ [−h, h]."""
    poly = np.array([[0, 0], [a, 0], [a, b], [b, b], [b, a], [0, a]], dtype=float)
    n = len(poly)
    v = np.concatenate([np.c_[poly, np.full(n, -h)], np.c_[poly, np.full(n, h)]])
    f = []
    for i in range(n):                                   # side walls (outward for a counter-clockwise polygon)
        j = (i + 1) % n
        f += [(i, j, n + j), (i, n + j, n + i)]
    tris = [(0, 1, 2), (0, 2, 3), (0, 3, 5), (3, 4, 5)]  # the polygon, triangulated (all inside the L)
    for t in tris:
        f.append((t[0], t[2], t[1]))                     # bottom: normal −z
        f.append((n + t[0], n + t[1], n + t[2]))         # top: normal +z
    return v, np.array(f), a, b, h


def test_legacy_and_exact_mesh_inertia_rules():
    """`<compiler exactmeshinertia>`: the default ("legacy") sums every face pyramid with the ABSOLUTE value of its volume —
    exact for a convex mesh only —, "true" sums signed volumes — exact for any closed surface (mink_amd/meshes.py
    mass_properties; round-3 advisor finding: the fitted capsules of non-convex collision meshes depend on it)."""
    rng = np.random.default_rng(3)
    # convex: both rules give the exact values
    v, f = _box(0.03, 0.05, 0.11)
    for exact in (False, True):
        vol, com, C = meshes.mass_properties(v + 0.2, f, exact)
        np.testing.assert_allclose(vol, 8 * 0.03 * 0.05 * 0.11, rtol=1e-12)
        np.testing.assert_allclose(com, 0.2, atol=1e-13)
        np.testing.assert_allclose(np.diag(C), vol * np.array([0.03, 0.05, 0.11]) ** 2 / 3, rtol=1e-10)
    # non-convex: the signed rule is exact ...
    v, f, a, b, h = _l_prism()
    vol, com, C = meshes.mass_properties(v, f, exact=True)
    area = a * b + b * (a - b)                                           # two rectangles: [0,a]×[0,b] and [0,b]×[b,a]
    np.testing.assert_allclose(vol, area * 2 * h, rtol=1e-12)
    cx = (a * b * a / 2 + b * (a - b) * b / 2) / area
    np.testing.assert_allclose(com, [cx, cx, 0.0], atol=1e-13)
    # ... the legacy rule is not: pyramids that face away from the surface centroid count positive too — a larger volume
    vol_l, com_l, C_l = meshes.mass_properties(v, f, exact=False)
    assert vol_l > vol * 1.01
    # recomputed from its definition: Σ |pyramid| from the area-weighted surface centroid
    A, B, Cc = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    ar = 0.5 * np.linalg.norm(np.cross(B - A, Cc - A), axis=1)
    apex = ((A + B + Cc) / 3 * ar[:, None]).sum(0) / ar.sum()
    pyr = np.abs(np.einsum("ij,ij->i", A - apex, np.cross(B - apex, Cc - apex))) / 6
    np.testing.assert_allclose(vol_l, pyr.sum(), rtol=1e-12)
    np.testing.assert_allclose(com_l, (pyr[:, None] * (0.75 * (A + B + Cc) / 3 + 0.25 * apex)).sum(0) / pyr.sum(), atol=1e-13)
    # both rules are invariant under rigid motions of the input
    R, t = _random_rotation(rng), rng.normal(size=3)
    for exact in (False, True):
        v0, c0, C0 = meshes.mass_properties(v, f, exact)
        v1, c1, C1 = meshes.mass_properties(v @ R.T + t, f, exact)
        np.testing.assert_allclose(v1, v0, rtol=1e-11)
        np.testing.assert_allclose(c1, R @ c0 + t, atol=1e-12)
        np.testing.assert_allclose(C1, R @ C0 @ R.T, atol=1e-14)
    # the MJCF reader honours the compiler flag
    from mink_amd import mjcf
    assert mjcf._Compiler().exactmeshinertia is False


def test_an_asset_the_reader_cannot_compile_does_not_fail_the_model_load(tmp_path):
    """A collision geom on a mesh that cannot be compiled (here: a flat sheet — no volume) is marked geom_valid = 0 and refused
    where it is used; the model itself loads (round-3 advisor finding)."""
    from mink_amd import mjcf
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[1, 0, 0], [1, 1, 0], [0, 1, 0]]], dtype="<f4")
    with open(tmp_path / "sheet.stl", "wb") as fh:
        fh.write(b"\0" * 80 + struct.pack("<I", len(tri)))
        for t in tri:
            fh.write(struct.pack("<3f", 0, 0, 1) + t.tobytes() + b"\0\0")
    xml = f"""<mujoco><compiler meshdir="{tmp_path}"/><asset><mesh name="sheet" file="sheet.stl"/></asset>
    <worldbody><body name="b"><joint type="hinge" axis="0 0 1"/><geom name="g" type="mesh" mesh="sheet"/>
    <geom name="c" type="capsule" mesh="sheet"/><geom name="s" type="sphere" size="0.1"/></body></worldbody></mujoco>"""
    p = tmp_path / "m.xml"
    p.write_text(xml)
    m = mjcf.load_mjcf(str(p))
    assert m.nv == 1 and list(np.asarray(m.geom_valid)) == [0, 0, 1]


def test_body_mass_from_mesh_geoms_when_there_is_no_inertial(tmp_path):
    """A body without <inertial> gets mass and centre of mass from its geoms (`inertiafromgeom` default); a mesh geom weighs
    density x the mesh's volume at the mesh's centre of mass, a primitive fitted to a mesh what the fitted primitive weighs — what a
    ComTask on the Allegro hands (21 such bodies, `<geom density="800"/>` on visual meshes) needs from the MJCF reader (round-3
    review, missing item 5; a compiled MjModel carries these numbers).  A mesh file that is not there leaves the body flagged."""
    from mink_amd import mjcf
    v, f = _box(0.02, 0.03, 0.05)
    v = v + np.array([0.1, -0.2, 0.3])                               # centre of mass away from the file's origin
    with open(tmp_path / "brick.obj", "w") as fh:
        for p in v:
            fh.write("v %.9f %.9f %.9f\n" % tuple(p))
        for t in f:
            fh.write("f %d %d %d\n" % tuple(t + 1))
    xml = f"""<mujoco><compiler meshdir="{tmp_path}"/>
    <asset><mesh name="brick" file="brick.obj"/><mesh name="gone" file="gone.obj"/></asset>
    <worldbody>
      <body name="a" pos="1 0 0"><joint type="hinge" axis="0 0 1"/>
        <geom type="mesh" mesh="brick" density="800" pos="0 0 0.5" contype="0" conaffinity="0"/>
        <geom type="sphere" size="0.1" mass="0.25" pos="0 1 0"/>
        <geom type="capsule" mesh="brick" pos="0.2 0 0"/>
        <geom type="box" size="0.1 0.1 0.1" mass="0" pos="5 5 5"/>
      </body>
      <body name="b"><joint type="hinge" axis="0 0 1"/><geom type="mesh" mesh="gone" contype="0" conaffinity="0"/></body>
      <body name="c"><inertial pos="0 0 0.1" mass="2"/><joint type="hinge" axis="0 0 1"/><geom type="mesh" mesh="gone"/></body>
    </worldbody></mujoco>"""
    p = tmp_path / "m.xml"
    p.write_text(xml)
    m = mjcf.load_mjcf(str(p))
    a, b, c = (m.name2id("body", n) for n in "abc")
    vol = 8 * 0.02 * 0.03 * 0.05
    # the fitted capsule: inertia box of the brick = the brick; radius = mean of the two short half-sizes, half-length = long − r/2
    r, h = 0.025, 0.05 - 0.0125
    m_caps = 1000.0 * (np.pi * r * r * 2 * h + 4.0 / 3.0 * np.pi * r ** 3)
    parts = [(800.0 * vol, np.array([0.1, -0.2, 0.3 + 0.5])), (0.25, np.array([0.0, 1.0, 0.0])),
             (m_caps, np.array([0.2 + 0.1, -0.2, 0.3]))]
    mt = sum(w for w, _ in parts)
    assert list(np.asarray(m.body_mass_valid)[[a, b, c]]) == [1, 0, 1]
    np.testing.assert_allclose(m.body_mass[a], mt, rtol=1e-12)
    np.testing.assert_allclose(m.body_ipos[a], sum(w * x for w, x in parts) / mt, rtol=0, atol=1e-12)
    assert m.body_mass[c] == 2.0 and m.body_mass[b] == 0.0
