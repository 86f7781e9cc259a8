"""mink_amd/mjcf.py pinned against constants READ OFF THE MJCF BY HAND.

The golden fixtures, both oracles and the product all receive their robots from the same MJCF reader, so a wrong
default-class / childclass / quaternion-normalisation / freejoint rule in it would be invisible to every parity
test.  Here one non-trivial chain of each benchmark robot is written down from the XML text
(/root/reference/examples/{universal_robots_ur5e/ur5e.xml, unitree_g1/g1.xml, shadow_hand/left_hand.xml} and their
scene files) — every number below was typed from those files, nothing is computed by the reader — and compared with
the committed FlatModel fixtures (tests/golden/models/*.json = what make_golden.py got from the reader; the packaged
mink_amd/robots/*.json must be identical).  The UR5e `home` end-effector pose is then derived in the test from those
typed constants alone (plain 4x4 products) and compared with the oracle's FK on the reader's model.
"""

import numpy as np
import pytest

import oracle_configs as oc
from oracle import ik

S = np.sqrt(0.5)


def _unit(*q):
    q = np.array(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def _rotmat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _zaxis(q):
    return _rotmat(q)[:, 2]


def _check_chain(m, bodies, joints, atol=1e-15):
    for name, parent, pos, quat, mass, ipos in bodies:
        b = m.name2id("body", name)
        assert b >= 0, name
        assert m.body_names[m.body_parentid[b]] == parent or (parent == "world" and m.body_parentid[b] == 0), name
        np.testing.assert_allclose(m.body_pos[b], pos, rtol=0, atol=atol, err_msg=name)
        np.testing.assert_allclose(m.body_quat[b], quat, rtol=0, atol=2e-16, err_msg=name)
        if mass is not None:
            assert m.body_mass[b] == mass, name
            np.testing.assert_allclose(m.body_ipos[b], ipos, rtol=0, atol=atol, err_msg=name)
    for name, body, jtype, axis, rng, limited in joints:
        j = m.name2id("joint", name)
        assert j >= 0, name
        assert m.body_names[m.jnt_bodyid[j]] == body and m.jnt_type[j] == jtype, name
        np.testing.assert_allclose(m.jnt_axis[j], axis, rtol=0, atol=1e-15, err_msg=name)
        assert bool(m.jnt_limited[j]) == limited, name
        if limited:
            np.testing.assert_array_equal(m.jnt_range[j], rng, err_msg=name)


# ------------------------------------------------------------------------------------------------ UR5e
# ur5e.xml: <default class="ur5e"> sets joint axis "0 1 0" range ±6.28319; class joint_limited (nested) only
# overrides the range; <body name="base" quat="1 0 0 1" childclass="ur5e">; compiler autolimits="true".
UR5E_BODIES = [  # name, parent, pos, quat (normalised by hand), mass, inertial pos
    ("base", "world", (0, 0, 0), (S, 0, 0, S), 4.0, (0, 0, 0)),
    ("shoulder_link", "base", (0, 0, 0.163), (1, 0, 0, 0), 3.7, (0, 0, 0)),
    ("upper_arm_link", "shoulder_link", (0, 0.138, 0), (S, 0, S, 0), 8.393, (0, 0, 0.2125)),
    ("forearm_link", "upper_arm_link", (0, -0.131, 0.425), (1, 0, 0, 0), 2.275, (0, 0, 0.196)),
    ("wrist_1_link", "forearm_link", (0, 0, 0.392), (S, 0, S, 0), 1.219, (0, 0.127, 0)),
    ("wrist_2_link", "wrist_1_link", (0, 0.127, 0), (1, 0, 0, 0), 1.219, (0, 0, 0.1)),
    ("wrist_3_link", "wrist_2_link", (0, 0, 0.1), (1, 0, 0, 0), 0.1879, (0, 0.0771683, 0)),
    ("target", "world", (0.5, 0, 0.5), (0, 1, 0, 0), None, None),           # scene.xml mocap body
    ("wall", "world", (0.5, 0, 0.1), (1, 0, 0, 0), None, None),
]
UR5E_JOINTS = [  # name, body, type (3 = hinge), axis, range, limited
    ("shoulder_pan", "shoulder_link", 3, (0, 0, 1), (-6.28319, 6.28319), True),
    ("shoulder_lift", "upper_arm_link", 3, (0, 1, 0), (-6.28319, 6.28319), True),
    ("elbow", "forearm_link", 3, (0, 1, 0), (-3.1415, 3.1415), True),          # class joint_limited
    ("wrist_1", "wrist_1_link", 3, (0, 1, 0), (-6.28319, 6.28319), True),
    ("wrist_2", "wrist_2_link", 3, (0, 0, 1), (-6.28319, 6.28319), True),
    ("wrist_3", "wrist_3_link", 3, (0, 1, 0), (-6.28319, 6.28319), True),
]
UR5E_HOME = (-1.5708, -1.5708, 1.5708, -1.5708, -1.5708, 0.0)


@pytest.mark.parametrize("source", ["golden", "packaged"])
def test_ur5e(source):
    from mink_amd import workloads
    m = oc.model("ur5e") if source == "golden" else workloads.load_robot("ur5e")
    assert (m.nq, m.nv, m.njnt, m.nbody, m.nmocap) == (6, 6, 6, 10, 1)
    _check_chain(m, UR5E_BODIES, UR5E_JOINTS)
    np.testing.assert_array_equal(m.qpos0, np.zeros(6))
    np.testing.assert_array_equal(m.key_qpos[m.name2id("key", "home")], UR5E_HOME)
    s = m.name2id("site", "attachment_site")
    assert m.body_names[m.site_bodyid[s]] == "wrist_3_link"
    np.testing.assert_array_equal(m.site_pos[s], (0, 0.1, 0))
    np.testing.assert_allclose(m.site_quat[s], (-0.5, 0.5, 0.5, 0.5), rtol=0, atol=2e-16)     # quat="-1 1 1 1"
    # <geom name="wrist_3_link" class="collision" pos="0 0.025 0" quat="1 1 0 0" size="0.04 0.05"/>: class
    # "collision" makes it a capsule (type 3) and leaves contype/conaffinity at 1
    g = m.name2id("geom", "wrist_3_link")
    assert (m.geom_type[g], m.geom_contype[g], m.geom_conaffinity[g], m.geom_valid[g]) == (3, 1, 1, 1)
    np.testing.assert_array_equal(m.geom_size[g], (0.04, 0.05, 0.0))
    np.testing.assert_array_equal(m.geom_pos[g], (0, 0.025, 0))
    np.testing.assert_allclose(m.geom_quat[g], (S, S, 0, 0), rtol=0, atol=2e-16)
    g = m.name2id("geom", "wall")
    assert m.geom_type[g] == 6 and m.body_names[m.geom_bodyid[g]] == "wall"
    np.testing.assert_array_equal(m.geom_size[g], (0.1, 0.1, 0.1))
    g = m.name2id("geom", "floor")
    assert m.geom_type[g] == 0 and m.geom_bodyid[g] == 0
    np.testing.assert_array_equal(m.geom_size[g], (1, 1, 0.01))
    # visual geoms: class "visual" → contype = conaffinity = 0 (never collision candidates)
    vis = [i for i in range(m.ngeom) if m.geom_type[i] == 7]
    assert len(vis) == 20 and all(m.geom_contype[i] == 0 and m.geom_conaffinity[i] == 0 for i in vis)
    # subtree masses: plain sums of the <inertial mass=...> values typed above
    masses = [4.0, 3.7, 8.393, 2.275, 1.219, 1.219, 0.1879]
    for k, (name, *_rest) in enumerate(UR5E_BODIES[:7]):
        np.testing.assert_allclose(m.body_subtreemass[m.name2id("body", name)], sum(masses[k:]), rtol=1e-15)


def _hom(quat, pos):
    w, x, y, z = quat
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = pos
    return T


def _rot(axis, ang):
    ax = np.array(axis, dtype=np.float64)
    return _hom((np.cos(ang / 2), *(np.sin(ang / 2) * ax)), (0, 0, 0))


def test_ur5e_home_pose_from_typed_constants():
    """FK of `home` from the typed constants only (joint anchors are the body origins: no <joint pos>), against the
    oracle's mj_kinematics on the reader's model and against the well-known UR5e home pose of this MJCF."""
    T = np.eye(4)
    for (name, _p, pos, quat, _m, _i), (_j, _b, _t, axis, _r, _l), q in zip(UR5E_BODIES[1:7], UR5E_JOINTS, UR5E_HOME):
        T = T @ _hom(quat, pos) @ _rot(axis, q)
    T = _hom(UR5E_BODIES[0][3], UR5E_BODIES[0][2]) @ T @ _hom((-0.5, 0.5, 0.5, 0.5), (0, 0.1, 0))
    m = oc.model("ur5e")
    d = ik.Configuration(m, np.array(UR5E_HOME)).data
    s = m.name2id("site", "attachment_site")
    np.testing.assert_allclose(d.site_xpos[s], T[:3, 3], rtol=0, atol=1e-15)
    np.testing.assert_allclose(d.site_xmat[s].reshape(3, 3), T[:3, :3], rtol=0, atol=1e-14)
    np.testing.assert_allclose(T[:3, 3], [0.4920, 0.1340, 0.4880], atol=2e-4)


# ------------------------------------------------------------------------------------------------ G1
# g1.xml: <body name="pelvis" pos="0 0 0.755" childclass="g1"> with <freejoint>; explicit <inertial> everywhere;
# joint axes / ranges are attributes of each <joint>; quaternions such as "0.984807 0 -0.17365 0" are NOT unit in
# the text (norm 1.0000001...) and must be normalised like MuJoCo does.
G1_BODIES = [
    ("pelvis", "world", (0, 0, 0.755), (1, 0, 0, 0), 2.86, (0, 0, -0.07605)),
    ("left_hip_pitch_link", "pelvis", (0, 0.06445, -0.1027), _unit(0.984807, 0, -0.17365, 0), 1.299,
     (0.001962, 0.049392, -0.000941)),
    ("left_hip_roll_link", "left_hip_pitch_link", (0, 0.0523, 0), (1, 0, 0, 0), 1.446, (0.024757, -0.001036, -0.086323)),
    ("left_hip_yaw_link", "left_hip_roll_link", (0.01966, -0.0012139, -0.1241), (1, 0, 0, 0), 2.052,
     (-0.053554, -0.011477, -0.14067)),
    ("left_knee_link", "left_hip_yaw_link", (-0.078292, -0.0017335, -0.177225), _unit(0.967714, 0, 0.252052, 0), 2.252,
     (0.005505, 0.006534, -0.116629)),
    ("left_ankle_pitch_link", "left_knee_link", (0, 0.0040687, -0.30007), _unit(0.99678, 0, -0.0801788, 0), 0.074,
     (-0.007269, 0, 0.011137)),
    ("left_ankle_roll_link", "left_ankle_pitch_link", (0, 0, -0.017558), (1, 0, 0, 0), 0.391, (0.024762, 2e-05, -0.012526)),
]
G1_JOINTS = [
    ("floating_base_joint", "pelvis", 0, (0, 0, 1), None, False),
    ("left_hip_pitch_joint", "left_hip_pitch_link", 3, (0, 1, 0), (-2.35, 3.05), True),
    ("left_hip_roll_joint", "left_hip_roll_link", 3, (1, 0, 0), (-0.26, 2.53), True),
    ("left_hip_yaw_joint", "left_hip_yaw_link", 3, (0, 0, 1), (-2.75, 2.75), True),
    ("left_knee_joint", "left_knee_link", 3, (0, 1, 0), (-0.33489, 2.5449), True),
    ("left_ankle_pitch_joint", "left_ankle_pitch_link", 3, (0, 1, 0), (-0.68, 0.73), True),
    ("left_ankle_roll_joint", "left_ankle_roll_link", 3, (1, 0, 0), (-0.2618, 0.2618), True),
]


@pytest.mark.parametrize("source", ["golden", "packaged"])
def test_g1(source):
    from mink_amd import workloads
    m = oc.model("g1") if source == "golden" else workloads.load_robot("g1")
    assert (m.nq, m.nv, m.njnt) == (44, 43, 38)
    _check_chain(m, G1_BODIES, G1_JOINTS)
    # <freejoint>: qpos0 = the body's pose in its parent, quaternion first of the last four (MuJoCo)
    np.testing.assert_array_equal(m.qpos0[:7], (0, 0, 0.755, 1, 0, 0, 0))
    np.testing.assert_array_equal(m.qpos0[7:], np.zeros(37))
    j = m.name2id("joint", "floating_base_joint")
    assert (m.jnt_qposadr[j], m.jnt_dofadr[j]) == (0, 0)
    # depth-first order: the left leg's joints take qpos 7..12 / dofs 6..11
    for k, (name, *_r) in enumerate(G1_JOINTS[1:]):
        j = m.name2id("joint", name)
        assert (m.jnt_qposadr[j], m.jnt_dofadr[j]) == (7 + k, 6 + k), name
        assert m.dof_parentid[6 + k] == 5 + k                                # chain back into the free joint's last dof
    s = m.name2id("site", "left_foot")
    assert m.body_names[m.site_bodyid[s]] == "left_ankle_roll_link"
    np.testing.assert_array_equal(m.site_pos[s], (0, 0, 0))
    np.testing.assert_array_equal(m.site_quat[s], (1, 0, 0, 0))
    # subtree masses of the leg: sums of the typed <inertial mass> values, leaf upwards
    leg = [1.299, 1.446, 2.052, 2.252, 0.074, 0.391]
    for k, (name, *_rest) in enumerate(G1_BODIES[1:]):
        np.testing.assert_allclose(m.body_subtreemass[m.name2id("body", name)], sum(leg[k:]), rtol=1e-15)
    # stand keyframe (g1.xml:422-431): base at z = 0.75 with the identity quaternion
    np.testing.assert_array_equal(m.key_qpos[m.name2id("key", "stand")][:7], (0, 0, 0.75, 1, 0, 0, 0))
    # foot contact spheres: class "foot" nested in "collision" → sphere of radius 0.001 (g1.xml:21-23,124-127)
    foot = [i for i in range(m.ngeom) if m.body_names[m.geom_bodyid[i]] == "left_ankle_roll_link" and m.geom_type[i] == 2]
    assert len(foot) == 4 and all(m.geom_size[i][0] == 0.001 for i in foot)
    np.testing.assert_array_equal(m.geom_pos[foot[0]], (-0.06, 0.02, -0.03))


# ------------------------------------------------------------------------------------------------ Shadow hand
# left_hand.xml: <body name="lh_forearm" childclass="left_hand" quat="0 1 0 1">; class left_hand sets joint axis
# "1 0 0"; nested classes wrist/wrist_y (axis 0 -1 0 + range), wrist_x (range only: axis inherited from left_hand),
# knuckle (axis 0 -1 0) overridden in place by axis="0 1 0" on lh_FFJ4, proximal / middle_distal (range only),
# thumb classes; metacarpal axis "0.573576 0 0.819152" overridden by axis="0.573576 0 -0.819152" on lh_LFJ5.
SH_BODIES = [
    ("lh_forearm", "world", (0, 0, 0), (0, S, 0, S), 3.0, (0, 0, 0.09)),
    ("lh_wrist", "lh_forearm", (0.01, 0, 0.21301), (S, 0, 0, S), 0.1, (0, 0, 0.029)),
    ("lh_palm", "lh_wrist", (0, 0, 0.034), (1, 0, 0, 0), 0.3, (0, 0, 0.035)),
    ("lh_ffknuckle", "lh_palm", (-0.033, 0, 0.095), (1, 0, 0, 0), 0.008, (0, 0, 0)),
    ("lh_ffproximal", "lh_ffknuckle", (0, 0, 0), (1, 0, 0, 0), 0.03, (0, 0, 0.0225)),
    ("lh_ffmiddle", "lh_ffproximal", (0, 0, 0.045), (1, 0, 0, 0), 0.017, (0, 0, 0.0125)),
    ("lh_ffdistal", "lh_ffmiddle", (0, 0, 0.025), (1, 0, 0, 0), 0.013, (0, 0, 0.0130769)),
    ("lh_thbase", "lh_palm", (-0.034, -0.00858, 0.029), _unit(0, -0.382683, 0, 0.92388), 0.01, (0, 0, 0)),
    ("lh_thproximal", "lh_thbase", (0, 0, 0), (1, 0, 0, 0), 0.04, (0, 0, 0.019)),
    ("lh_thhub", "lh_thproximal", (0, 0, 0.038), (1, 0, 0, 0), 0.005, (0, 0, 0)),
    ("lh_thmiddle", "lh_thhub", (0, 0, 0), (1, 0, 0, 0), 0.02, (0, 0, 0.016)),
    ("lh_thdistal", "lh_thmiddle", (0, 0, 0.032), (S, 0, 0, -S), 0.017, (0, 0, 0.0145588)),
    ("thumb_target", "world", (0.5, 0, 0.5), (0, 1, 0, 0), None, None),
]
SH_JOINTS = [
    ("lh_WRJ2", "lh_wrist", 3, (0, -1, 0), (-0.523599, 0.174533), True),
    ("lh_WRJ1", "lh_palm", 3, (1, 0, 0), (-0.698132, 0.488692), True),
    ("lh_FFJ4", "lh_ffknuckle", 3, (0, 1, 0), (-0.349066, 0.349066), True),
    ("lh_FFJ3", "lh_ffproximal", 3, (1, 0, 0), (-0.261799, 1.5708), True),
    ("lh_FFJ2", "lh_ffmiddle", 3, (1, 0, 0), (0, 1.5708), True),
    ("lh_FFJ1", "lh_ffdistal", 3, (1, 0, 0), (0, 1.5708), True),
    ("lh_THJ5", "lh_thbase", 3, (0, 0, 1), (-1.0472, 1.0472), True),
    ("lh_THJ4", "lh_thproximal", 3, (-1, 0, 0), (0, 1.22173), True),
    ("lh_THJ3", "lh_thhub", 3, (-1, 0, 0), (-0.20944, 0.20944), True),
    ("lh_THJ2", "lh_thmiddle", 3, (0, -1, 0), (-0.698132, 0.698132), True),
    ("lh_THJ1", "lh_thdistal", 3, (1, 0, 0), (-0.261799, 1.5708), True),
    ("lh_LFJ5", "lh_lfmetacarpal", 3, _unit(0.573576, 0, -0.819152), (0, 0.785398), True),
]


@pytest.mark.parametrize("source", ["golden", "packaged"])
def test_shadow_left(source):
    from mink_amd import workloads
    m = oc.model("shadow_left") if source == "golden" else workloads.load_robot("shadow_left")
    assert (m.nq, m.nv, m.njnt, m.nmocap) == (24, 24, 24, 5)
    _check_chain(m, SH_BODIES, SH_JOINTS)
    for site, body, pos in (("first", "lh_ffdistal", (0, 0, 0.025)), ("thumb", "lh_thdistal", (0, 0, 0.032))):
        s = m.name2id("site", site)
        assert m.body_names[m.site_bodyid[s]] == body
        np.testing.assert_array_equal(m.site_pos[s], pos)
    for geom, body, size, pos in (("first_1", "lh_ffproximal", (0.009, 0.02, 0), (0, 0, 0.025)),
                                  ("first_2", "lh_ffmiddle", (0.009, 0.0125, 0), (0, 0, 0.0125)),
                                  ("thumb_1", "lh_thproximal", (0.0105, 0.009, 0), (0, 0, 0.02)),
                                  ("thumb_2", "lh_thmiddle", (0.009, 0.009, 0), (0, 0, 0.012))):
        g = m.name2id("geom", geom)
        assert m.geom_type[g] == 3 and m.body_names[m.geom_bodyid[g]] == body and m.geom_valid[g] == 1
        np.testing.assert_array_equal(m.geom_size[g], size)
        np.testing.assert_array_equal(m.geom_pos[g], pos)
    # mesh-fitted capsules (left_hand.xml:149 / :263 `type="capsule" mesh="f_distal_pst"` / `"th_distal_pst"`, mesh scale 0.001 :8):
    # radius, half-length, centre and axis TYPED from the output of tests/golden/derive_mesh_pins.py — a second derivation that
    # shares no code with the reader (own OBJ parser, per-triangle loops, Jacobi eigen-solver); flag 2 = fitted, frame axes only
    # up to half turns.  (This pin is what found, in round 4, that this fixture still carried the exact-inertia fit after the
    # reader had moved to the compiler's legacy default.)
    for geom, body, radius, halflen, pos, axis in (
            ("first_3", "lh_ffdistal", 0.00668597552241, 0.00779683667472,
             (-2.39844179808e-05, 0.000420732553707, 0.0176464755128), (-0.000011283, 0.002690022, 0.999996382)),
            ("thumb_3", "lh_thdistal", 0.00817286083005, 0.00785656372807,
             (-8.25444041129e-07, 0.000802518760052, 0.0195533730824), (0.000106926, -0.003173490, -0.999994959))):
        g = m.name2id("geom", geom)
        assert m.geom_valid[g] == 2 and m.geom_type[g] == 3 and m.body_names[m.geom_bodyid[g]] == body
        np.testing.assert_allclose(m.geom_size[g][:2], (radius, halflen), rtol=1e-9)
        np.testing.assert_allclose(m.geom_pos[g], pos, rtol=0, atol=1e-12)
        zaxis = _zaxis(m.geom_quat[g])
        assert abs(abs(zaxis @ np.array(axis)) - 1.0) < 1e-9, (geom, zaxis)   # capsule axis, up to its direction
    # the one true mesh collision geom of the hand (left_hand.xml:101 `type="mesh" mesh="forearm_collision"`): its hull.
    # forearm_collision.obj has 228 distinct vertices, 98 of them on the hull (qhull on the file's own coordinates, same script);
    # the six extreme vertices below are READ OFF THE FILE (× 0.001) — the model stores the hull in the mesh's inertial frame, so
    # geom_pos + R(geom_quat)·hull must reproduce them in the body frame (float32 storage: 1e-8)
    fm = [k for k in range(m.ngeom) if m.geom_type[k] == 7 and m.geom_dataid[k] >= 0]
    assert len(fm) == 1 and m.body_names[m.geom_bodyid[fm[0]]] == "lh_forearm"
    hull = m.mesh_hull(fm[0])
    assert len(hull) == 98
    in_body = m.geom_pos[fm[0]] + hull @ _rotmat(m.geom_quat[fm[0]]).T
    for d, lo, hi in ((0, (-0.074762627, -2.3e-08, 0.156506897), (0.074762581, -2.3e-08, 0.156506897)),
                      (1, (-0.006898247, -0.07444371, 0.156506897), (0.006898201, 0.074443657, 0.156506897)),
                      (2, None, (0.067499977, -2.3e-08, 0.182800003))):
        if lo is not None:
            assert np.abs(in_body - np.array(lo)).sum(axis=1).min() < 3e-8, (d, lo)
            assert abs(in_body[:, d].min() - lo[d]) < 1e-8
        assert np.abs(in_body - np.array(hi)).sum(axis=1).min() < 3e-8, (d, hi)
        assert abs(in_body[:, d].max() - hi[d]) < 1e-8
    assert abs(in_body[:, 2].min() - (-4e-09)) < 1e-8                       # the base of the forearm: the plane z = 0 of the file
    g = m.name2id("geom", "floor")                          # scene_left.xml: pos="0 0 -0.1" size="0 0 0.05"
    np.testing.assert_array_equal(m.geom_pos[g], (0, 0, -0.1))
    np.testing.assert_array_equal(m.geom_size[g], (0, 0, 0.05))
    assert m.body_mocapid[m.name2id("body", "thumb_target")] == 0
    np.testing.assert_array_equal(m.mocap_pos[0], (0.5, 0, 0.5))


# ------------------------------------------------------------------------------------------------ ALOHA
# aloha.xml: meshes scaled 0.001 (:9-14); class "collision" = `<geom group="3" type="capsule" .../>` (:86-87), so every
# `<geom class="collision" mesh=...>` is a capsule FITTED to its mesh; the geom's own pos / quat (typed from the XML, quat
# normalised by hand) compose with the mesh's inertial frame.  Radius / half-length / centre / axis: typed from
# tests/golden/derive_mesh_pins.py (see the Shadow test).  Body placement typed from aloha.xml:100-125.
ALOHA_CAPSULES = [  # body, parent, body pos, radius, half-length, centre in the body frame, axis in the body frame
    ("left/shoulder_link", "left/base_link", (0, 0, 0.079), 0.0346084511082, 0.0294520521564,
     (0.000632571956375, 7.75043501511e-11, 0.020391457864), (0, 1, 0)),                      # :111 pos="0 0 -0.003" quat="1 0 0 1"
    ("left/upper_arm_link", "left/shoulder_link", (0, 0, 0.04805), 0.0340304611975, 0.152895847956,
     (0.018794847185, 1.06888765445e-09, 0.203813886889), (0.137037749, -0.000000001, 0.990565826)),   # :117 quat="1 0 0 1"
    ("left/upper_forearm_link", "left/upper_arm_link", (0.05955, 0, 0.3), 0.0255845380061, 0.0864867043566,
     (0.0915029827223, 4.669695453e-08, 5.33518918792e-08), (1, 0, 0)),                       # :123 no pos / quat
]


def test_aloha_fitted_capsules():
    import os
    m = oc.FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", "aloha__scene.json"))
    for body, parent, bpos, radius, halflen, pos, axis in ALOHA_CAPSULES:
        b = m.name2id("body", body)
        assert m.body_names[m.body_parentid[b]] == parent
        np.testing.assert_allclose(m.body_pos[b], bpos, rtol=0, atol=1e-15)
        gs = [k for k in range(m.ngeom) if m.geom_bodyid[k] == b and m.geom_type[k] == 3]
        assert len(gs) == 1 and m.geom_valid[gs[0]] == 2, body            # one collision capsule per link, fitted
        g = gs[0]
        np.testing.assert_allclose(m.geom_size[g][:2], (radius, halflen), rtol=1e-9)
        np.testing.assert_allclose(m.geom_pos[g], pos, rtol=0, atol=1e-12)
        assert abs(abs(_zaxis(m.geom_quat[g]) @ np.array(axis)) - 1.0) < 1e-8, body


def test_allegro_body_masses_from_mesh_geoms_second_derivation():
    """Bodies WITHOUT <inertial> whose mass comes from mesh geoms (round-4 review: pinned only by the reader's own unit test).
    wonik_allegro/left_hand.xml: class "allegro_left" sets `<geom density="800"/>` (:10), the visual classes are `type="mesh"`
    (:13) with no mesh scale, every collision geom carries `mass="0"` (:38), no body has an <inertial> — so a finger link weighs
    800 x the legacy volume of its visual mesh, at that mesh's centre of mass (+ the geom's own pos: the fingertip's
    `pos="0 0 0.0267"`, :30).  The numbers below are TYPED from the output of tests/golden/derive_mesh_pins.py — its own STL
    reader, plain loops over the triangles, nothing shared with mink_amd/mjcf.py / meshes.py — and compared with the committed
    model fixture (what the reader compiled from the same files)."""
    from mink_amd.flatmodel import FlatModel
    import os
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", "wonik_allegro__scene_left.json"))
    pins = {  # body: (mass, ipos)
        "rf_proximal": (0.0272760563087, (2.75233141553e-08, -9.05955355725e-05, 0.0269997695608)),   # link_1.0.stl
        "rf_medial": (0.018458520143, (-0.000192881578363, -7.79041643551e-05, 0.0225860114871)),    # link_2.0.stl
        "rf_tip": (0.00684452138085, (7.27630463767e-12, -1.15602789957e-07, 0.0254027647694)),       # link_3.0_tip.stl at z + 0.0267
    }
    for name, (mass, ipos) in pins.items():
        b = m.name2id("body", name)
        assert m.body_mass_valid[b] == 1
        np.testing.assert_allclose(m.body_mass[b], mass, rtol=1e-11)
        np.testing.assert_allclose(m.body_ipos[b], ipos, rtol=0, atol=1e-13)
    # the three fingers of the same build share the meshes: same masses (left_hand.xml:131-198)
    for other in ("mf", "ff"):
        for link in ("proximal", "medial", "tip"):
            np.testing.assert_allclose(m.body_mass[m.name2id("body", f"{other}_{link}")], pins[f"rf_{link}"][0], rtol=1e-11)
