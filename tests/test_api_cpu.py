"""Host-side API logic that needs no GPU: constructor validation and error strings
(cf. reference tests/test_posture_task.py:29-62, test_com_task.py:38-60,
test_velocity_limit.py:27-140, test_configuration_limit.py:36-121,
test_collision_avoidance_limit.py:30-63, test_frame_task.py:29-105), Lie value classes."""

import os

import numpy as np
import pytest

import mink_amd as mink
from mink_amd.flatmodel import mjMAXVAL

TINY = """
<mujoco>
  <compiler angle="radian"/>
  <worldbody>
    <body name="b1">
      <joint name="hinge" type="hinge" range="0 1.57"/>
      <geom name="g1" type="sphere" size=".1" mass=".1"/>
      <body name="b2" pos="0 0 .3">
        <joint name="ball" type="ball"/>
        <geom name="g2" type="sphere" size=".1" mass=".1"/>
        <body name="b3" pos="0 0 .3">
          <joint name="slide" type="slide" axis="1 0 0"/>
          <geom name="g3" type="capsule" size=".05 .1" mass=".1"/>
        </body>
      </body>
    </body>
    <body name="floating" pos="1 0 0"><freejoint name="free"/><geom name="g4" type="sphere" size=".1" mass=".1"/></body>
  </worldbody>
</mujoco>
"""


@pytest.fixture(scope="module")
def g1():
    return mink.load_robot("g1")


@pytest.fixture(scope="module")
def tiny():
    return mink.loads_mjcf(TINY)


def test_mjcf_reader_dimensions(tiny):
    assert (tiny.nq, tiny.nv, tiny.njnt) == (1 + 4 + 1 + 7, 1 + 3 + 1 + 6, 4)
    assert list(tiny.jnt_limited) == [1, 0, 0, 0]          # autolimits: range given ⇒ limited
    np.testing.assert_allclose(tiny.qpos0, [0, 1, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0])
    assert list(tiny.dof_parentid) == [-1, 0, 1, 2, 3, -1, 5, 6, 7, 8, 9]
    assert tiny.body_rootid.tolist() == [0, 1, 1, 1, 4]
    assert tiny.body_weldid.tolist() == [0, 1, 2, 3, 4]


def test_task_validation_strings(g1):
    with pytest.raises(mink.TaskDefinitionError, match=r"PostureTask cost must be a vector of shape \(1,\) "
                                                      r"\(aka identical cost for all dofs\) or \(43,\). Got \(2,\)"):
        mink.PostureTask(g1, cost=(0.5, 2.0))
    with pytest.raises(mink.TaskDefinitionError, match="PostureTask cost should be >= 0"):
        mink.PostureTask(g1, cost=-1.0)
    t = mink.PostureTask(g1, cost=1.0)
    with pytest.raises(mink.InvalidTarget, match=r"Expected target posture to have shape \(44,\) but got \(45,\)"):
        t.set_target(np.zeros(45))
    with pytest.raises(mink.TaskDefinitionError, match=r"ComTask cost must be a vector of shape \(1,\) "
                                                      r"\(aka identical cost for all coordinates\) or \(3,\). Got \(2,\)"):
        mink.ComTask(cost=(1, 2))
    with pytest.raises(mink.TaskDefinitionError, match="ComTask cost must be >= 0"):
        mink.ComTask(cost=(-1, -1, -1))
    with pytest.raises(mink.InvalidTarget, match=r"Expected target CoM to have shape \(3,\) but got \(5,\)"):
        mink.ComTask(cost=1.0).set_target(np.zeros(5))
    with pytest.raises(mink.TaskDefinitionError):
        mink.FrameTask("pelvis", "body", position_cost=[1.0, 2.0], orientation_cost=1.0)
    with pytest.raises(mink.TaskDefinitionError):
        mink.FrameTask("pelvis", "body", position_cost=1.0, orientation_cost=-1.0)
    with pytest.raises(mink.InvalidGain):
        mink.FrameTask("pelvis", "body", 1.0, 1.0, gain=1.5)
    with pytest.raises(mink.InvalidDamping):
        mink.FrameTask("pelvis", "body", 1.0, 1.0, lm_damping=-1.0)
    ft = mink.FrameTask("pelvis", "body", position_cost=[1.0, 2.0, 3.0], orientation_cost=5.0)
    np.testing.assert_array_equal(ft.cost, [1, 2, 3, 5, 5, 5])
    d = mink.DampingTask(g1, cost=1.0)
    assert d.gain == 0.0
    np.testing.assert_array_equal(d.target_q, g1.qpos0)


def test_frame_task_target_is_copied():
    """reference tests/test_frame_task.py:107-122"""
    task = mink.FrameTask("pelvis", "body", 1.0, 1.0)
    target = mink.SE3(np.array([1.0, 0, 0, 0, 0.1, 0.2, 0.3]))
    task.set_target(target)
    y = target.translation()[1]
    target.translation()[1] += 12.0
    assert task.transform_target_to_world.translation()[1] == y


def test_configuration_limit_constructor(g1, tiny):
    lim = mink.ConfigurationLimit(g1)
    np.testing.assert_array_equal(lim.indices, np.arange(6, g1.nv))      # free joint skipped
    assert lim.projection_matrix.shape == (g1.nv - 6, g1.nv)
    lt = mink.ConfigurationLimit(tiny, min_distance_from_limits=0.1)
    np.testing.assert_array_equal(lt.indices, [0])
    np.testing.assert_allclose(lt.lower, [0.1] + [-mjMAXVAL] * 12)
    np.testing.assert_allclose(lt.upper, [1.47] + [mjMAXVAL] * 12)
    with pytest.raises(mink.LimitDefinitionError, match=r"gain must be in the range \(0, 1\]"):
        mink.ConfigurationLimit(g1, gain=0.0)


def test_velocity_limit_constructor(tiny, g1):
    v = mink.VelocityLimit(tiny, {"hinge": 1.0, "ball": (1.0, 2.0, 3.0), "slide": 0.5})
    np.testing.assert_array_equal(v.indices, [0, 1, 2, 3, 4])
    np.testing.assert_allclose(v.limit, [1, 1, 2, 3, 0.5])
    assert v.projection_matrix.shape == (5, tiny.nv)
    with pytest.raises(mink.LimitDefinitionError, match="Free joint free is not supported"):
        mink.VelocityLimit(tiny, {"free": np.ones(6)})
    with pytest.raises(mink.LimitDefinitionError, match=r"Joint ball must have a limit of shape \(3,\). Got: \(1,\)"):
        mink.VelocityLimit(tiny, {"ball": 1.0})
    assert mink.VelocityLimit(g1).projection_matrix is None
    with pytest.raises(KeyError):
        mink.VelocityLimit(g1, {"no_such_joint": 1.0})


def test_collision_pair_filtering():
    m = mink.load_robot("shadow_left")
    f = ["thumb", "first", "middle", "ring", "little"]
    groups = [[f"{x}_1", f"{x}_2"] for x in f]
    col = mink.CollisionAvoidanceLimit(m, [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)])
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "shadow_c4_geom_pairs.npy"))
    assert sorted(col.geom_id_pairs) == sorted(map(tuple, ref.tolist()))   # recorded from the real mink
    # same-finger capsules are parent/child ⇒ filtered
    assert mink.CollisionAvoidanceLimit(m, [(["first_1"], ["first_2"])]).max_num_contacts == 0


def test_lie_value_classes(golden_dir):
    g = np.load(os.path.join(golden_dir, "lie.npz"))
    T = mink.SE3(g["se3_params"])
    np.testing.assert_allclose(T.log(), g["se3_log"], atol=1e-14)
    np.testing.assert_allclose(mink.SE3.exp(g["tangent"]).wxyz_xyz, g["se3_exp"], atol=1e-14)
    np.testing.assert_allclose(T.inverse().wxyz_xyz, g["se3_inverse"], atol=1e-14)
    np.testing.assert_allclose(T.as_matrix(), g["se3_as_matrix"], atol=1e-14)
    np.testing.assert_allclose(T.adjoint(), g["se3_adjoint"], atol=1e-14)
    one = mink.SE3(g["se3_params"][3])
    np.testing.assert_allclose(one.log(), g["se3_log"][3], atol=1e-14)
    np.testing.assert_allclose((one @ one.inverse()).wxyz_xyz, mink.SE3.identity().wxyz_xyz, atol=1e-15)
    with pytest.raises(ValueError):
        mink.SO3(np.zeros(3))


def test_exception_messages(g1):
    with pytest.raises(mink.InvalidKeyframe, match="Keyframe nope does not exist in the model"):
        cfg = mink.Configuration.__new__(mink.Configuration)
        cfg.model = g1
        cfg.update_from_keyframe("nope")
    e = mink.NotWithinConfigurationLimits(joint_id=1, value=9.0, lower=-1.0, upper=1.0, model=g1)
    assert "Joint 1 (left_hip_pitch_joint) violates configuration limits -1.0 <= 9.0 <= 1.0" in str(e)
    assert "No target set for FrameTask" in str(mink.TargetNotSet("FrameTask"))


def test_mesh_dependent_geometry_and_mass_are_refused():
    """Round-1 advisor findings: a primitive fitted to a mesh has no usable size/frame WITHOUT the mesh asset, and a
    body without <inertial> whose geoms are meshes has no usable mass: both must fail loudly, not compute from
    placeholders.  (Round 3: with the asset at hand the reader fits the primitive — the packaged Shadow hand's `*_3`
    fingertips are ordinary capsules now, tests/test_meshes_cpu.py.)"""
    import mink_amd as mink
    from mink_amd import workloads
    from mink_amd.mjcf import loads_mjcf

    m = workloads.load_robot("shadow_left")
    mink.CollisionAvoidanceLimit(m, [(["first_3"], ["thumb_2"])])            # fitted from the asset when the model was compiled
    with pytest.raises(mink.InvalidFrame, match="mesh asset"):               # ... but not a frame: axes only up to half turns
        mink.Configuration(m).get_transform_frame_to_world("first_3", "geom")
    xml = """<mujoco><asset><mesh name="link" file="link.stl"/></asset><worldbody>
      <body name="a"><joint type="hinge"/><geom type="mesh" mesh="link"/><geom name="fit" type="capsule" mesh="link"/>
        <body name="b" pos="0 0 1"><joint type="hinge"/><geom type="sphere" size=".1"/>
          <body name="c" pos="0 0 1"><joint type="hinge"/><geom name="ball" type="sphere" size=".1"/></body></body></body>
      </worldbody></mujoco>"""
    mm = loads_mjcf(xml)                                                     # link.stl does not exist
    assert mm.geom_valid.tolist() == [0, 0, 1, 1]
    with pytest.raises(mink.LimitDefinitionError, match="mesh asset"):
        mink.CollisionAvoidanceLimit(mm, [(["fit"], ["ball"])])
    assert mm.body_mass_valid.tolist() == [1, 0, 1, 1]
    with pytest.raises(ValueError, match="<inertial>"):
        mm.require_valid_masses("ComTask")
    workloads.load_robot("g1").require_valid_masses("ComTask")              # explicit <inertial> everywhere


def test_from_mjmodel_ingest(monkeypatch):
    """`Configuration(model: mujoco.MjModel)` (mink/configuration.py:37-51) → FlatModel.from_mjmodel, the production
    ingest: executed on an object that exposes exactly the real MjModel attribute names / dtypes (oracle/stubs's
    RawMjModel; the wheel itself is absent).  Every array the device library receives must come out identical to
    the MJCF reader's model the raw object was filled from."""
    import sys
    monkeypatch.syspath_prepend(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "stubs"))
    monkeypatch.delitem(sys.modules, "mujoco", raising=False)
    import mujoco
    from mink_amd import _native as nat
    from mink_amd import workloads
    from mink_amd.configuration import as_flat_model
    from mink_amd.flatmodel import FlatModel

    try:
        for robot in ("ur5e", "g1", "shadow_left"):
            src = workloads.load_robot(robot)
            raw = mujoco.RawMjModel(src)
            assert not isinstance(raw, FlatModel) and type(raw).__module__ == "mujoco"
            fm = as_flat_model(raw)                     # what Configuration.__init__ / PostureTask / the limits call
            assert isinstance(fm, FlatModel) and fm is not src
            for name, ctype in nat.MkhFlatModel._fields_:
                if name in ("nmesh", "nmeshvert"):      # (counts the binding derives from the arrays below)
                    continue
                a, b = getattr(fm, name), getattr(src, name)
                if isinstance(a, np.ndarray):
                    assert a.dtype == b.dtype and a.shape == b.shape, name
                    np.testing.assert_array_equal(a, b, err_msg=name)
                else:
                    assert a == b, name
            for name in ("body_names", "jnt_names", "site_names", "geom_names", "key_names"):
                assert getattr(fm, name) == getattr(src, name), name
            np.testing.assert_array_equal(fm.key_qpos, src.key_qpos)
            np.testing.assert_array_equal(fm.mocap_pos, src.mocap_pos)
            np.testing.assert_array_equal(fm.body_weldid, src.body_weldid)
            assert fm.geom_valid.all() and fm.body_mass_valid.all()       # a compiled MjModel has every field
            assert fm.name2id("site", src.site_names[0]) == 0
    finally:
        sys.modules.pop("mujoco", None)


def test_plugin_detection():
    """Built-in tasks/limits have device descriptors; a subclass that brings its own compute_error /
    compute_qp_inequalities is routed through the dense (plugin) rows."""
    import mink_amd as mink
    from mink_amd import workloads
    m = workloads.load_robot("ur5e")
    builtins_t = [mink.FrameTask("attachment_site", "site", 1.0, 1.0), mink.PostureTask(m, 1.0), mink.DampingTask(m, 1.0),
                  mink.ComTask(1.0), mink.RelativeFrameTask("attachment_site", "site", "base", "body", 1.0, 1.0)]
    assert not any(t._is_dense() for t in builtins_t)
    builtins_l = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {"elbow": 1.0}),
                  mink.CollisionAvoidanceLimit(m, [(["wrist_3_link"], ["floor"])])]
    assert not any(lim._is_dense() for lim in builtins_l)

    class Mine(mink.Task):
        def compute_error(self, configuration):
            return np.zeros(2)

        def compute_jacobian(self, configuration):
            return np.zeros((2, configuration.nv))

    class MyFrame(mink.FrameTask):             # overriding the error of a built-in also takes the plugin route
        def compute_error(self, configuration):
            return np.zeros(6)

    class MyLimit(mink.Limit):
        def compute_qp_inequalities(self, configuration, dt):
            return mink.Constraint()

    class MyVel(mink.VelocityLimit):
        def compute_qp_inequalities(self, configuration, dt):
            return mink.Constraint()

    assert Mine(cost=np.ones(2))._is_dense() and MyFrame("attachment_site", "site", 1.0, 1.0)._is_dense()
    assert MyLimit()._is_dense() and MyVel(m, {"elbow": 1.0})._is_dense()
    kind, desc = mink.Task._native_desc(Mine(cost=np.array([1.0, 2.0]), gain=0.5, lm_damping=0.1), None)
    assert kind == "dense" and desc["cost"].tolist() == [1.0, 2.0] and desc["gain"] == 0.5 and desc["lm_damping"] == 0.1


def test_fold_box_rows_of_caller_defined_limits():
    """solve_ik._fold_box_rows: single-entry rows of a plugin limit become per-dof bounds, the rest stay half-spaces."""
    from mink_amd.solve_ik import _fold_box_rows
    B, nv = 3, 5
    G = np.zeros((B, 6, nv)); h = np.zeros((B, 6))
    G[:, 0, 1] = 2.0; h[:, 0] = [1.0, 2.0, np.inf]            # x1 ≤ h/2 (inactive in the last instance)
    G[:, 1, 1] = -1.0; h[:, 1] = 0.25                          # x1 ≥ −0.25
    G[:, 2, :] = 1.0; h[:, 2] = 3.0                            # general row
    G[0, 3, 4] = 1.0; G[1, 3, 4] = -1.0; h[:, 3] = 0.5         # sign differs per instance; instance 2 has g = 0
    G[:, 4, 0] = 1.0; G[:, 4, 2] = 1.0; h[:, 4] = 1.0          # two entries: general
    h[:, 5] = [1.0, 1.0, -1.0]                                 # all-zero row: general (the solver reports 0 ≤ −1 infeasible)
    lo, hi, Gr, hr, single = _fold_box_rows(G, h)
    assert Gr.shape == (B, 3, nv) and hr.shape == (B, 3) and single
    np.testing.assert_array_equal(hi[:, 1], [0.5, 1.0, np.inf])
    np.testing.assert_array_equal(lo[:, 1], [-0.25] * 3)
    np.testing.assert_array_equal(hi[:, 4], [0.5, np.inf, np.inf])
    np.testing.assert_array_equal(lo[:, 4], [-np.inf, -0.5, -np.inf])
    assert np.isinf(lo[:, [0, 2, 3]]).all() and np.isinf(hi[:, [0, 2, 3]]).all()
    # g = 0 with h < 0 in one instance of a single-entry row: infeasible box for that instance only
    G2 = np.zeros((2, 1, nv)); G2[0, 0, 3] = 1.0; h2 = np.array([[1.0], [-1.0]])
    lo2, hi2, _, _, _ = _fold_box_rows(G2, h2)
    assert hi2[0, 3] == 1.0 and lo2[1, 3] > hi2[1, 3]
    # the layout key is structural: single-entry rows that are all inactive now (h = +inf) still announce the box — the handle
    # (and its warm-start state) must not be rebuilt when they switch on at the next call (round-3 advisor finding)
    h3 = np.full((2, 1), np.inf)
    lo3, hi3, G3r, _, single3 = _fold_box_rows(G2, h3)
    assert single3 and G3r.shape[1] == 0 and np.isinf(hi3).all() and np.isinf(lo3).all()
    assert _fold_box_rows(np.ones((2, 1, nv)), np.ones((2, 1)))[4] is False


def test_partial_override_detection():
    import mink_amd as mink

    class JOnly(mink.FrameTask):
        def compute_jacobian(self, configuration):
            return super().compute_jacobian(configuration)

    class ObjOnly(mink.PostureTask):
        def compute_qp_objective(self, configuration):
            return super().compute_qp_objective(configuration)

    class Plain(mink.FrameTask):                  # extra state / helpers only: still the built-in device path
        def helper(self):
            return 1

    from mink_amd import workloads
    m = workloads.load_robot("ur5e")
    assert JOnly("attachment_site", "site", 1.0, 1.0)._is_dense()
    assert ObjOnly(m, 1.0)._is_dense()
    p = Plain("attachment_site", "site", 1.0, 1.0)
    assert not p._is_dense() and p._builtin_class() is mink.FrameTask
    assert not mink.DampingTask(m, 1.0)._is_dense() and mink.DampingTask(m, 1.0)._builtin_class() is mink.PostureTask


def test_fingerprints_follow_what_the_device_descriptor_reads():
    """solve_ik._compile memoises on `_fingerprint()` (control loops): it must change with everything `_native_desc` reads —
    costs are arrays the reference's setters modify IN PLACE — stay equal otherwise (targets are per-call data, not part of
    the descriptor), differ between objects, and be None for caller-defined tasks / limits (their rows are per-call)."""
    m = mink.load_robot("ur5e")
    t = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    f0 = t._fingerprint()
    assert f0 is not None and f0 == t._fingerprint() and hash(f0) is not None
    t.set_target(mink.SE3.identity())
    assert t._fingerprint() == f0                                    # a target is not part of the descriptor
    t.set_position_cost(2.0)                                         # in place
    f1 = t._fingerprint()
    assert f1 != f0
    t.cost[5] = 0.25                                                 # even behind the setters' back
    assert t._fingerprint() != f1
    t2 = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    t3 = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    assert t2._fingerprint() != t3._fingerprint()                    # (layouts hold the objects: identity is part of it)
    for attr, val in (("gain", 0.5), ("lm_damping", 0.0), ("frame_name", "wrist_3_link"), ("frame_type", "body")):
        before = t2._fingerprint()
        setattr(t2, attr, val)
        assert t2._fingerprint() != before, attr
    p = mink.PostureTask(m, cost=1e-2)
    fp = p._fingerprint()
    p.set_target(m.key_qpos[0])
    assert p._fingerprint() == fp
    p.set_cost(np.linspace(0.1, 0.6, m.nv))
    assert p._fingerprint() != fp
    r = mink.RelativeFrameTask("attachment_site", "site", "base", "body", 1.0, 1.0)
    fr = r._fingerprint()
    r.root_name = "shoulder_link"
    assert r._fingerprint() != fr
    lim = mink.ConfigurationLimit(m)
    fl = lim._fingerprint()
    lim.gain = 0.5
    assert lim._fingerprint() != fl
    fl = lim._fingerprint()
    lim.upper[2] -= 0.1
    assert lim._fingerprint() != fl
    vl = mink.VelocityLimit(m, {n: 1.0 for n in m.jnt_names})
    fv = vl._fingerprint()
    vl.limit = np.full(m.nv, 2.0)                                    # (the array itself is read-only, as in the reference)
    assert vl._fingerprint() != fv

    class MyTask(mink.FrameTask):
        def compute_error(self, configuration):
            return super().compute_error(configuration)

    class MyLimit(mink.Limit):
        def compute_qp_inequalities(self, configuration, dt):
            return mink.Constraint()

    assert MyTask("attachment_site", "site", 1.0, 1.0)._fingerprint() is None
    assert MyLimit()._fingerprint() is None


def test_objective_to_rows_carries_any_psd_objective():
    """`compute_qp_objective` overrides reach the device as nv rows with JᵀJ = H and Jᵀe = c (mink_amd/tasks.py:
    objective_to_rows; the reference folds a task in through that method alone, mink/solve_ik.py:18-21)."""
    from mink_amd.tasks import objective_to_rows

    rng = np.random.default_rng(0)
    nv, B = 9, 17
    A = rng.normal(size=(B, 3, nv))
    H = np.einsum("bki,bkj->bij", A, A)                           # rank 3
    # (a) a least-squares objective: c in the range of H — no spare row is used
    c_in = np.einsum("bki,bk->bi", A, rng.normal(size=(B, 3)))
    e, J = objective_to_rows(H, c_in)
    assert e.shape == (B, nv) and J.shape == (B, nv, nv)
    np.testing.assert_allclose(np.einsum("bki,bkj->bij", J, J), H, rtol=0, atol=1e-13 * np.abs(H).max())
    np.testing.assert_allclose(np.einsum("bki,bk->bi", J, e), c_in, rtol=0, atol=1e-12 * np.abs(c_in).max())
    assert (np.abs(J[:, 3:]).max(axis=(1, 2)) == 0.0).all() and np.abs(e).max() < 1e3
    # (b) a generic linear term: the part outside the range rides on one tiny row, H moves below its rounding
    c_out = rng.normal(size=(B, nv))
    e, J = objective_to_rows(H, c_out)
    np.testing.assert_allclose(np.einsum("bki,bkj->bij", J, J), H, rtol=0, atol=1e-13 * np.abs(H).max())
    np.testing.assert_allclose(np.einsum("bki,bk->bi", J, e), c_out, rtol=0, atol=1e-12 * np.abs(c_out).max())
    # (c) full rank, unbatched, and a pure linear objective
    Hf = H[0] + np.eye(nv)
    e, J = objective_to_rows(Hf, c_out[0])
    assert e.shape == (nv,) and J.shape == (nv, nv)
    np.testing.assert_allclose(J.T @ J, Hf, rtol=0, atol=1e-13 * np.abs(Hf).max())
    np.testing.assert_allclose(J.T @ e, c_out[0], rtol=0, atol=1e-12)
    e, J = objective_to_rows(np.zeros((nv, nv)), c_out[0])
    assert np.abs(J.T @ J).max() < 1e-20
    np.testing.assert_allclose(J.T @ e, c_out[0], rtol=0, atol=1e-12)
    # (d) one H for the whole batch with per-instance c
    e, J = objective_to_rows(H[0], c_out)
    assert e.shape == (B, nv) and J.shape == (B, nv, nv)
    np.testing.assert_allclose(np.einsum("bki,bk->bi", J, e), c_out, rtol=0, atol=1e-12 * np.abs(c_out).max())
    # (e) what quadprog would refuse
    with pytest.raises(mink.TaskDefinitionError, match="positive semi-definite"):
        objective_to_rows(-np.eye(nv), np.zeros(nv))
    with pytest.raises(mink.TaskDefinitionError, match="symmetric"):
        objective_to_rows(np.triu(np.ones((nv, nv))), np.zeros(nv))
