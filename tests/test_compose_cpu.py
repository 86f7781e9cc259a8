"""mink_amd.compose.attach — the model composition of the reference's arm + hand examples
(examples/arm_hand_iiwa_allegro.py:32-42: `attach_site.attach(hand_mjcf)`) on FlatModels — against the same robot written
out as ONE nested MJCF, and against the numpy oracle's kinematics."""

import json

import numpy as np
import pytest

from mink_amd import compose, mjcf
from oracle import ik as oik

ARM = """<mujoco><compiler angle="radian" autolimits="true"/><worldbody>
<geom name="floor" type="plane" size="0 0 0.01"/>
<body name="l1" pos="0 0 0.1"><inertial pos="0 0 0.05" mass="2" diaginertia="1 1 1"/><joint name="j1" axis="0 0 1" range="-2 2"/>
 <geom name="g1" type="capsule" size="0.03 0.1"/>
 <body name="l2" pos="0 0.1 0.2" quat="0.9238795 0.3826834 0 0"><inertial pos="0 0 0.1" mass="1" diaginertia="1 1 1"/><joint name="j2" axis="0 1 0" range="-1.5 1.5"/>
  <site name="tool" pos="0.01 0.02 0.25" quat="0.7071068 0 0 0.7071068"/>
  <body name="l3" pos="0.1 0 0"><inertial pos="0 0 0" mass="0.5" diaginertia="1 1 1"/><joint name="j3" type="slide" axis="1 0 0" range="-0.1 0.1"/></body>
 </body>
</body>
<body name="other" pos="1 0 0"><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/><joint name="jo" axis="1 0 0"/></body>
</worldbody><keyframe><key name="home" qpos="0.1 0.2 0.03 0.4"/></keyframe></mujoco>"""

HAND = """<mujoco><compiler angle="radian" autolimits="true"/><worldbody>
<site name="base_mark" pos="0 0 0.01"/>
<body name="palm" pos="0 0 0.05" quat="0.7071068 0.7071068 0 0"><inertial pos="0 0 0.02" mass="0.4" diaginertia="1 1 1"/>
 <geom name="palm_g" type="box" size="0.04 0.01 0.05"/>
 <body name="f1" pos="0.02 0 0.06"><inertial pos="0 0 0.01" mass="0.05" diaginertia="1 1 1"/><joint name="f1j" axis="0 1 0" range="0 1.5" ref="0.1"/>
  <site name="f1_tip" pos="0 0 0.04"/><body name="f1b" pos="0 0 0.03"><inertial pos="0 0 0.01" mass="0.02" diaginertia="1 1 1"/><joint name="f1k" type="ball"/></body></body>
 <body name="f2" pos="-0.02 0 0.06"><inertial pos="0 0 0.01" mass="0.05" diaginertia="1 1 1"/><joint name="f2j" axis="0 1 0" range="0 1.5"/><site name="f2_tip" pos="0 0 0.04"/></body>
</body></worldbody><keyframe><key name="open" qpos="0.3 1 0 0 0 0.2"/></keyframe></mujoco>"""

# the same robot as one file: the hand's palm below l2, its pose = the site's pose ∘ the palm's own; quat(z90°) ⊗ quat(x90°) =
# (0.5, 0.5, 0.5, 0.5); pos = site pos + Rz90·(0, 0, 0.05); the child's world-level site rides on l2 at the composed pose too
WHOLE = """<mujoco><compiler angle="radian" autolimits="true"/><worldbody>
<geom name="floor" type="plane" size="0 0 0.01"/>
<body name="l1" pos="0 0 0.1"><inertial pos="0 0 0.05" mass="2" diaginertia="1 1 1"/><joint name="j1" axis="0 0 1" range="-2 2"/>
 <geom name="g1" type="capsule" size="0.03 0.1"/>
 <body name="l2" pos="0 0.1 0.2" quat="0.9238795 0.3826834 0 0"><inertial pos="0 0 0.1" mass="1" diaginertia="1 1 1"/><joint name="j2" axis="0 1 0" range="-1.5 1.5"/>
  <site name="tool" pos="0.01 0.02 0.25" quat="0.7071068 0 0 0.7071068"/>
  <site name="hand/base_mark" pos="0.01 0.02 0.26" quat="0.7071068 0 0 0.7071068"/>
  <body name="l3" pos="0.1 0 0"><inertial pos="0 0 0" mass="0.5" diaginertia="1 1 1"/><joint name="j3" type="slide" axis="1 0 0" range="-0.1 0.1"/></body>
  <body name="hand/palm" pos="0.01 0.02 0.30" quat="0.5 0.5 0.5 0.5"><inertial pos="0 0 0.02" mass="0.4" diaginertia="1 1 1"/>
   <geom name="hand/palm_g" type="box" size="0.04 0.01 0.05"/>
   <body name="hand/f1" pos="0.02 0 0.06"><inertial pos="0 0 0.01" mass="0.05" diaginertia="1 1 1"/><joint name="hand/f1j" axis="0 1 0" range="0 1.5" ref="0.1"/>
    <site name="hand/f1_tip" pos="0 0 0.04"/><body name="hand/f1b" pos="0 0 0.03"><inertial pos="0 0 0.01" mass="0.02" diaginertia="1 1 1"/><joint name="hand/f1k" type="ball"/></body></body>
   <body name="hand/f2" pos="-0.02 0 0.06"><inertial pos="0 0 0.01" mass="0.05" diaginertia="1 1 1"/><joint name="hand/f2j" axis="0 1 0" range="0 1.5"/><site name="hand/f2_tip" pos="0 0 0.04"/></body>
  </body>
 </body>
</body>
<body name="other" pos="1 0 0"><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/><joint name="jo" axis="1 0 0"/></body>
</worldbody><keyframe><key name="home" qpos="0.1 0.2 0.03 0.3 1 0 0 0 0.2 0.4"/></keyframe></mujoco>"""


def test_attach_equals_the_same_robot_written_as_one_file():
    arm, hand, whole = mjcf.loads_mjcf(ARM), mjcf.loads_mjcf(HAND), mjcf.loads_mjcf(WHOLE)
    m = compose.attach(arm, hand, site="tool", prefix="hand/", child_key="open")
    a, b = json.loads(m.to_json()), json.loads(whole.to_json())
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], list) and a[k] and isinstance(a[k][0], (int, float, list)):
            np.testing.assert_allclose(np.array(a[k], dtype=float), np.array(b[k], dtype=float), rtol=0, atol=2e-7, err_msg=k)   # (7-digit quaternions in the files)
        else:
            assert a[k] == b[k], k
    # depth-first order: every subtree is a contiguous range, and the hand sits INSIDE l2's range, in front of `other`
    assert m.body_names == ["world", "l1", "l2", "l3", "hand/palm", "hand/f1", "hand/f1b", "hand/f2", "other"]
    assert m.jnt_names == ["j1", "j2", "j3", "hand/f1j", "hand/f1k", "hand/f2j", "jo"]
    assert m.nq == 10 and m.nv == 9 and m.dof_parentid.tolist() == [-1, 0, 1, 1, 3, 4, 5, 6, 1, -1][:0] + [-1, 0, 1, 1, 3, 4, 5, 1, -1]
    # neither argument was modified
    assert arm.nbody == 5 and hand.nbody == 5 and hand.body_names[1] == "palm"


def test_kinematics_of_an_attached_model_are_the_parts_kinematics():
    """FK of the composed model = FK of the arm, then the hand's own FK carried by the attachment frame."""
    arm, hand = mjcf.loads_mjcf(ARM), mjcf.loads_mjcf(HAND)
    m = compose.attach(arm, hand, site="tool", prefix="h/")
    rng = np.random.default_rng(0)
    from oracle import lie
    for _ in range(5):
        qa = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.1, 0.1), rng.uniform(-1, 1)])
        w = rng.normal(size=4)
        qh = np.concatenate([[rng.uniform(0, 1.5)], w / np.linalg.norm(w), [rng.uniform(0, 1.5)]])
        q = np.concatenate([qa[:3], qh, qa[3:]])
        T_site = oik.Configuration(arm, qa).get_transform_frame_to_world(arm.name2id("site", "tool"), "site")
        T_tip_h = oik.Configuration(hand, qh).get_transform_frame_to_world(hand.name2id("site", "f1_tip"), "site")
        T_tip = oik.Configuration(m, q).get_transform_frame_to_world(m.name2id("site", "h/f1_tip"), "site")
        ref = lie.se3_multiply(T_site, T_tip_h)
        if ref[:4] @ T_tip[:4] < 0:
            ref[:4] = -ref[:4]
        np.testing.assert_allclose(T_tip, ref, rtol=0, atol=1e-12)


def test_the_g1_with_two_allegro_hands():
    """The `g1_hands` bench model (mink_amd/workloads.py): 43 + 2 x 16 dofs, 44 + 2 x 21 bodies, both past one wavefront."""
    from mink_amd import workloads
    m = workloads.load_bench_robot("g1_hands")
    assert (m.nv, m.nq, m.nbody) == (75, 76, 86)
    assert m.name2id("site", "lh/ff_tip") >= 0 and m.name2id("site", "rh/th_tip") >= 0
    for b in range(1, m.nbody):                                  # parents first, subtrees contiguous
        assert m.body_parentid[b] < b
    last = np.arange(m.nbody)
    for b in range(m.nbody - 1, 0, -1):
        last[m.body_parentid[b]] = max(last[m.body_parentid[b]], last[b])
    for b in range(1, m.nbody):
        assert all(_is_below(m, c, b) for c in range(b, last[b] + 1))
    g1 = workloads.load_robot("g1")
    np.testing.assert_allclose(m.body_subtreemass[1], g1.body_subtreemass[1] + 2 * workloads.load_robot("allegro_left").body_subtreemass[1], rtol=1e-12)
    stand = m.key_qpos[m.name2id("key", "stand")]
    assert stand.shape == (76,) and np.isfinite(stand).all()


def _is_below(m, c, b):
    while c > 0 and c != b:
        c = int(m.body_parentid[c])
    return c == b


def test_attach_refuses_repeated_names():
    """MuJoCo's compiler and dm_control raise "repeated name" when two bodies / joints / sites / geoms of a composed model share a
    name; FlatModel.finalize() alone would keep the last entry and a FrameTask or a collision geom list would silently bind to the
    wrong copy (round-5 advisor finding: the same hand on both palms of the G1 without prefixes)."""
    from mink_amd import workloads
    from mink_amd.compose import attach
    hand = workloads.load_robot("allegro_left")
    m = attach(workloads.load_robot("g1"), hand, site="left_palm", pos=(0.0, 0.0, 0.02))
    with pytest.raises(ValueError, match="repeated .* name .*prefix"):
        attach(m, hand, site="right_palm", pos=(0.0, 0.0, 0.02))
    ok = attach(m, hand, site="right_palm", prefix="rh/", pos=(0.0, 0.0, 0.02))
    assert len(set(n for n in ok.body_names if n)) == len([n for n in ok.body_names if n])


def test_attach_a_model_to_itself_keeps_the_two_copies_apart():
    """attach(m, m, ...): joints and meshes are keyed by which ARGUMENT they come from, not by id(model) — the parent's keyframe
    values stay on the parent's joints, the child's joints take `child_key` (or their qpos0)."""
    arm = mjcf.loads_mjcf(ARM)
    home = arm.key_qpos[arm.name2id("key", "home")]
    assert np.abs(home - arm.qpos0).min() > 0.01
    two = compose.attach(arm, arm, site="tool", prefix="b/")
    assert (two.nv, two.nq, two.nbody) == (2 * arm.nv, 2 * arm.nq, 2 * arm.nbody - 1)
    k = two.key_qpos[two.name2id("key", "home")]
    first = [int(two.jnt_qposadr[two.name2id("joint", n)]) for n in arm.jnt_names]
    second = [int(two.jnt_qposadr[two.name2id("joint", "b/" + n)]) for n in arm.jnt_names]
    np.testing.assert_array_equal(k[first], home)
    np.testing.assert_array_equal(k[second], arm.qpos0)                # (no child_key: the child's joints keep their qpos0)
    two_k = compose.attach(arm, arm, site="tool", prefix="b/", child_key="home")
    k2 = two_k.key_qpos[two_k.name2id("key", "home")]
    np.testing.assert_array_equal(k2[first], home)
    np.testing.assert_array_equal(k2[second], home)
