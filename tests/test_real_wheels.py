"""The MuJoCo / quadprog layer under mink is restated, not linked (oracle/mjmath.py, oracle/qp_gi.py): the wheels are absent from
the build image.  These tests close that gap wherever the wheels ARE importable (tests/golden/real_wheels.py) and skip here —
so that "parity unpinned against MuJoCo"  becomes a number on the first box that has them, with no code to write.
Reference anchors: mink/configuration.py:53-64 (mj_kinematics / mj_comPos), :144-145 (mj_jac*), tasks/com_task.py:96
(mj_jacSubtreeCom), limits/collision_avoidance_limit.py:219 (mj_geomDistance), solve_ik.py:101 (qpsolvers → quadprog)."""

import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import real_wheels as rw  # noqa: E402

REFERENCE = os.environ.get("MINK_REFERENCE", "/root/reference")


def _real():
    w = rw.wheels(REFERENCE)
    if w is None:
        pytest.skip("the real mujoco / qpsolvers / quadprog wheels (and a mink checkout with examples/) are not available here")
    if not os.path.isdir(os.path.join(REFERENCE, "examples")):
        pytest.skip("no mink checkout with examples/ (set MINK_REFERENCE)")
    return w


def test_committed_fixtures_against_the_real_wheels():
    """Every array of the four BASELINE fixtures (task e / J, H, c, G, h, v), recomputed by mujoco + mink + quadprog from the
    committed inputs, within the tolerances the GPU path itself is held to (real_wheels.TOL = DESIGN.md §5)."""
    mj, mink, kind = _real()
    assert kind == "real"
    rep = rw.diff_fixtures(mj, mink, os.path.join(REFERENCE, "examples"))
    bad = {n: {k: v for k, v in r.items() if not v["ok"]} for n, r in rep.items()}
    assert not any(bad.values()), bad


def test_from_mjmodel_on_every_example_scene_against_the_mjcf_reader():
    """FlatModel.from_mjmodel on a REAL MjModel of each of the 18 example scenes equals what the repository's MJCF reader compiled
    (tests/golden/models/all): tree, joint frames, mesh-derived inertial frames and masses, primitives fitted to meshes, and the
    convex hulls mesh_graph holds against the reader's own hulls."""
    mj, mink, kind = _real()
    assert kind == "real"
    rep = rw.diff_models(mj, os.path.join(REFERENCE, "examples"))
    assert len(rep) >= 18
    bad = {n: [k for k, v in r.items() if isinstance(v, dict) and v.get("ok") is False] + ([] if r.get("sizes_ok") else ["sizes"])
           for n, r in rep.items()}
    assert not any(bad.values()), {n: b for n, b in bad.items() if b}


def test_the_checker_itself_reproduces_the_fixtures_over_the_stubs():
    """Build container only (needs the reference checkout): the configuration definitions of real_wheels.py are those of
    make_golden.py — over oracle/stubs the recomputed arrays equal the committed ones bit for bit.  Says nothing about MuJoCo."""
    if not os.path.isdir(os.path.join(REFERENCE, "mink")):
        pytest.skip("no reference checkout on this box")
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    try:
        mj, mink, kind = rw.wheels(REFERENCE, allow_stubs=True)
        if kind != "stubs":
            pytest.skip("real wheels present: the tests above are the ones that count")
        rep = rw.diff_fixtures(mj, mink, os.path.join(REFERENCE, "examples"))
        assert set(rep) == {"ur5e_c2", "g1_c3", "g1_full", "shadow_c4"}
        for name, r in rep.items():
            assert {"v", "H", "c", "task_e"} <= set(r), (name, list(r))
            assert all(v["worst"] == 0.0 for v in r.values()), (name, r)
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k not in saved_mods and k.split(".")[0] in ("mujoco", "qpsolvers", "mink")]:
            del sys.modules[k]
