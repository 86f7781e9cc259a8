"""mink.utils helpers on a FlatModel — the reference's tests/test_utils.py re-run (same cases, same expectations)."""

import numpy as np
import pytest

import mink_amd as mink
from mink_amd import utils, workloads
from mink_amd.exceptions import InvalidKeyframe, InvalidMocapBody


@pytest.fixture(scope="module")
def g1():
    return workloads.load_robot("g1")


def test_custom_configuration_vector_throws_error_if_keyframe_invalid(g1):       # test_utils.py:23-25
    with pytest.raises(InvalidKeyframe):
        utils.custom_configuration_vector(g1, "stand123")


def test_custom_configuration_vector_from_keyframe(g1):                          # :27-29
    q = utils.custom_configuration_vector(g1, "stand")
    np.testing.assert_allclose(q, g1.key_qpos[g1.name2id("key", "stand")])


def test_custom_configuration_vector_raises_error_if_jnt_shape_invalid(g1):      # :31-37
    with pytest.raises(ValueError):
        utils.custom_configuration_vector(g1, "stand", left_ankle_pitch_joint=(0.1, 0.1))


def test_custom_configuration_vector(g1):                                        # :39-49
    custom = dict(left_ankle_pitch_joint=0.2, right_ankle_roll_joint=0.1)
    q = utils.custom_configuration_vector(g1, **custom)
    q_expected = g1.qpos0.copy()
    for name, value in custom.items():
        q_expected[g1.jnt_qposadr[g1.name2id("joint", name)]] = value
    np.testing.assert_array_almost_equal(q, q_expected)


def test_move_mocap_to_frame_throws_error_if_body_not_mocap(g1):                 # :51-59
    with pytest.raises(InvalidMocapBody):
        utils.move_mocap_to_frame(g1, None, "left_ankle_roll_link", "unused_frame_name", "unused_frame_type")


def test_get_freejoint_dims(g1):                                                 # :100-109
    q_ids, v_ids = utils.get_freejoint_dims(g1)
    assert q_ids == list(range(0, 7)) and v_ids == list(range(0, 6))


SUBTREE_GEOMS = """<mujoco><worldbody>
  <body name="b1" pos=".1 -.1 0"><joint type="free"/>
    <geom name="b1/g1" type="sphere" size=".1" mass=".1"/>
    <geom name="b1/g2" type="sphere" size=".1" mass=".1" pos="0 0 .5"/>
    <body name="b2"><joint type="hinge" range="0 1.57" limited="true"/>
      <geom name="b2/g1" type="sphere" size=".1" mass=".1"/></body></body>
  <body name="b3" pos="1 1 1"><joint type="free"/>
    <geom name="b3/g1" type="sphere" size=".1" mass=".1"/>
    <body name="b4"><joint type="hinge" range="0 1.57" limited="true"/>
      <geom name="b4/g1" type="sphere" size=".1" mass=".1"/></body></body>
  <body name="geomless"><inertial pos="0 0 0" mass=".1" diaginertia="1 1 1"/></body>
</worldbody></mujoco>"""


def test_get_subtree_geom_ids():                                                 # :111-155
    model = mink.loads_mjcf(SUBTREE_GEOMS)
    gid = lambda n: model.name2id("geom", n)
    bid = lambda n: model.name2id("body", n)
    assert set(utils.get_subtree_geom_ids(model, bid("b1"))) == {gid("b1/g1"), gid("b1/g2"), gid("b2/g1")}
    assert set(utils.get_subtree_geom_ids(model, bid("b3"))) == {gid("b3/g1"), gid("b4/g1")}
    assert utils.get_subtree_geom_ids(model, bid("geomless")) == []
    assert set(utils.get_subtree_geom_ids(model, 0)) == set(range(model.ngeom))
    assert utils.get_body_geom_ids(model, bid("b1")) == [gid("b1/g1"), gid("b1/g2")]


SUBTREE_BODIES = """<mujoco><worldbody>
  <body name="b1" pos=".1 -.1 0"><joint type="free"/>
    <geom name="b1/g1" type="sphere" size=".1" mass=".1"/>
    <body name="b3"><joint type="hinge" range="0 1.57" limited="true"/>
      <geom name="b3/g1" type="sphere" size=".1" mass=".1"/>
      <body name="b4" pos="1 1 1"><geom name="b4/g1" type="sphere" size=".1" mass=".1"/></body></body>
    <body name="b2" pos="1 1 1"><geom name="b2/g1" type="sphere" size=".1" mass=".1"/></body></body>
  <body name="b5" pos="1 1 1"><joint type="free"/><geom name="b5/g1" type="sphere" size=".1" mass=".1"/></body>
</worldbody></mujoco>"""


def test_get_subtree_body_ids():                                                 # :157-190
    model = mink.loads_mjcf(SUBTREE_BODIES)
    bid = lambda n: model.name2id("body", n)
    assert set(utils.get_subtree_body_ids(model, bid("b1"))) == {bid(n) for n in ("b1", "b2", "b3", "b4")}
    assert set(utils.get_subtree_body_ids(model, bid("b5"))) == {bid("b5")}
    assert set(utils.get_subtree_body_ids(model, 0)) == set(range(model.nbody))


def test_package_surface_matches_the_reference():
    """Every name the reference package really defines and exports (mink/__init__.py:1-84; its __all__ also lists
    `set_mocap_pose_from_frame` / `pose_from_mocap`, which it never defines) is exported here too."""
    names = ["ComTask", "Configuration", "build_ik", "solve_ik", "DampingTask", "FrameTask", "RelativeFrameTask",
             "PostureTask", "Task", "Objective", "ConfigurationLimit", "VelocityLimit", "CollisionAvoidanceLimit",
             "Constraint", "Limit", "SO3", "SE3", "MatrixLieGroup", "MinkError", "UnsupportedFrame", "InvalidFrame",
             "InvalidKeyframe", "NotWithinConfigurationLimits", "TargetNotSet", "InvalidMocapBody", "SUPPORTED_FRAMES",
             "FRAME_TO_ENUM", "FRAME_TO_JAC_FUNC", "FRAME_TO_POS_ATTR", "FRAME_TO_XMAT_ATTR",
             "custom_configuration_vector", "get_freejoint_dims", "move_mocap_to_frame", "get_subtree_geom_ids",
             "get_subtree_body_ids", "get_body_geom_ids"]
    for n in names:
        assert hasattr(mink, n) and n in mink.__all__, n
    assert isinstance(mink.SE3.identity(), mink.MatrixLieGroup) and isinstance(mink.SO3.identity(), mink.MatrixLieGroup)


def test_contact_struct():
    """mink/limits/collision_avoidance_limit.py:20-56."""
    from mink_amd.limits import Contact
    c = Contact(dist=0.2, fromto=np.array([0.0, 0, 0, 0, 0, 0.2]), geom1=1, geom2=2, distmax=0.5)
    np.testing.assert_allclose(c.normal, [0, 0, 1])
    assert not c.inactive
    far = Contact(dist=0.5, fromto=np.zeros(6), geom1=1, geom2=2, distmax=0.5)
    assert far.inactive
    np.testing.assert_allclose(far.normal, [1, 0, 0])             # mju_normalize3 of a zero vector
