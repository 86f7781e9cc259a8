"""Close the one boundary this repository cannot pin in its build container: the MuJoCo / quadprog layer under mink.

Every fixture in this directory was produced by the REAL `mink` package (imported from /root/reference) running over
`oracle/stubs` — numpy restatements of the `mujoco` and `qpsolvers` symbols mink calls (the wheels are not installed in the
build image and there is no network).  The mink layer is therefore pinned against the reference; what `mj_kinematics`,
`mj_jac*`, `mj_jacSubtreeCom`, `mj_geomDistance`, `mj_integratePos` and quadprog itself would have returned is the oracle's own
restatement (`oracle/mjmath.py`, `oracle/qp_gi.py`, written from the published algorithms).  On any machine where the real
wheels ARE importable this script turns "parity unpinned against MuJoCo" into numbers, in one command:

    python tests/golden/real_wheels.py [--reference /path/to/mink/checkout] [--out DIR]
    python tests/golden/make_golden.py --real-wheels            # the same

It (1) re-runs the four BASELINE configurations of make_golden.py with the real `mujoco.MjModel`, the real `mink` and
`solver="quadprog"` on the committed inputs (q, targets) and diffs every recorded array (task e / J, H, c, G, h, v) against the
committed fixture, and (2) compiles every example scene with `mujoco.MjModel.from_xml_path`, ingests it through the product's
`FlatModel.from_mjmodel`, and diffs it against the FlatModel the repository's own MJCF reader produced
(tests/golden/models/all/*.json: body frames, mesh-derived inertial frames and masses, fitted primitives, convex hulls).
Nothing is overwritten: results go to --out (default: a temporary directory) as JSON.  `tests/test_real_wheels.py` runs the
same functions under pytest and SKIPS when the wheels are absent (as they are in this image).

`--allow-stubs` runs the same code over oracle/stubs (self-check of this script in the build container: every diff is then 0
by construction and proves nothing about MuJoCo)."""

from __future__ import annotations

import argparse
import glob
import importlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
STUBS = os.path.join(REPO, "oracle", "stubs")

# tolerances: DESIGN.md §5 "Stated tolerance (fp64)" — what the GPU path is held to against these same fixtures
TOL = {"task_e": 1e-12, "task_J": 1e-9, "H": 1e-11, "c": 1e-11, "h": 1e-9, "G": 1e-9, "v": 1e-8}
# FlatModel fields compared between from_mjmodel(real MjModel) and the repository's MJCF reader (absolute tolerance; the XML
# files carry 6–8 significant digits, the mesh pipeline float32 vertices)
MODEL_FIELDS = {"body_pos": 1e-9, "body_quat": 1e-9, "body_ipos": 1e-6, "body_mass": 1e-6, "body_subtreemass": 1e-5,
                "jnt_pos": 1e-9, "jnt_axis": 1e-9, "jnt_range": 1e-9, "qpos0": 1e-9, "site_pos": 1e-9, "site_quat": 1e-9,
                "geom_size": 1e-6, "geom_pos": 1e-6, "geom_quat": 1e-6, "key_qpos": 1e-9}
MODEL_INT_FIELDS = ("body_parentid", "body_jntnum", "body_jntadr", "body_dofnum", "body_dofadr", "jnt_type", "jnt_qposadr",
                    "jnt_dofadr", "jnt_bodyid", "jnt_limited", "dof_bodyid", "dof_jntid", "dof_parentid", "site_bodyid",
                    "geom_bodyid", "geom_type")


def wheels(reference: str = "/root/reference", allow_stubs: bool = False):
    """(mujoco, mink, kind) with kind 'real' | 'stubs', or None when the real wheels are not importable (and stubs not allowed).
    'real' means: `mujoco` resolves outside oracle/stubs, exposes the compiled engine (mj_step) and `qpsolvers` + `quadprog` import."""
    for k in [k for k in sys.modules if k == "mujoco" or k.startswith("mujoco.") or k == "qpsolvers" or k.startswith("qpsolvers.")
              or k == "mink" or k.startswith("mink.")]:
        del sys.modules[k]
    path0 = list(sys.path)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p) != STUBS]
    kind = None
    try:
        mj = importlib.import_module("mujoco")
        importlib.import_module("qpsolvers")
        importlib.import_module("quadprog")
        if hasattr(mj, "mj_step") and STUBS not in os.path.abspath(getattr(mj, "__file__", "")):
            kind = "real"
    except ImportError:
        pass
    if kind is None:
        sys.path[:] = path0
        if not allow_stubs:
            return None
        for k in [k for k in sys.modules if k.split(".")[0] in ("mujoco", "qpsolvers")]:
            del sys.modules[k]
        sys.path[:0] = [STUBS, REPO]
        mj = importlib.import_module("mujoco")
        kind = "stubs"
    try:
        mink = importlib.import_module("mink")                 # a pip-installed mink, if any ...
    except ImportError:
        if not os.path.isdir(os.path.join(reference, "mink")):
            return None
        sys.path.insert(0, reference)                           # ... else the reference checkout
        mink = importlib.import_module("mink")
    return mj, mink, kind


def _names(mj, m, kind, n):
    return [mj.mj_id2name(m, kind, i) or "" for i in range(n)]


def _configs(mj, mink, examples: str):
    """The four BASELINE configurations exactly as make_golden.py::main builds them (kept in step by
    tests/test_real_wheels.py::test_the_checker_itself_reproduces_the_fixtures_over_the_stubs)."""
    out = {}
    m = mj.MjModel.from_xml_path(os.path.join(examples, "universal_robots_ur5e", "scene.xml"))
    ft = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    pt = mink.PostureTask(m, cost=1e-2)
    vel = {n: np.pi for n in _names(mj, m, mj.mjtObj.mjOBJ_JOINT, m.njnt)}
    out["ur5e_c2"] = dict(model=m, tasks=[ft, pt], frame_tasks=[ft], posture=pt, com=None,
                          limits=[mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)])
    m = mj.MjModel.from_xml_path(os.path.join(examples, "unitree_g1", "scene.xml"))
    feet = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0) for s in ("left_foot", "right_foot")]
    hands = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=0.0, lm_damping=1.0) for s in ("left_palm", "right_palm")]
    pt = mink.PostureTask(m, cost=1.0)
    jn = _names(mj, m, mj.mjtObj.mjOBJ_JOINT, m.njnt)
    vel = {jn[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] != 0}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)]
    out["g1_c3"] = dict(model=m, tasks=feet + hands + [pt], frame_tasks=feet + hands, posture=pt, com=None, limits=lims)
    pelvis = mink.FrameTask("pelvis", "body", position_cost=0.0, orientation_cost=10.0)
    com = mink.ComTask(cost=200.0)
    out["g1_full"] = dict(model=m, tasks=[pelvis, pt, com] + feet + hands, frame_tasks=[pelvis] + feet + hands, posture=pt, com=com,
                          limits=lims)
    m = mj.MjModel.from_xml_path(os.path.join(examples, "shadow_hand", "scene_left.xml"))
    fingers = ["thumb", "first", "middle", "ring", "little"]
    fts = [mink.FrameTask(f, "site", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0) for f in fingers]
    pt = mink.PostureTask(m, cost=1e-2)
    groups = [[f"{f}_1", f"{f}_2"] for f in fingers]
    pairs = [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)]
    col = mink.CollisionAvoidanceLimit(m, pairs, collision_detection_distance=0.03)
    out["shadow_c4"] = dict(model=m, tasks=[pt] + fts, frame_tasks=fts, posture=pt, com=None, limits=[mink.ConfigurationLimit(m), col])
    return out


def _worst(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return float("inf")
    if a.size == 0:
        return 0.0
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    with np.errstate(invalid="ignore"):
        d = np.where(both_inf, 0.0, np.abs(a - b))
    return float(np.nan_to_num(d, nan=np.inf).max() / max(1.0, float(np.abs(b[np.isfinite(b)]).max()) if np.isfinite(b).any() else 1.0))


def diff_fixtures(mj, mink, examples: str, names=("ur5e_c2", "g1_c3", "g1_full", "shadow_c4")) -> dict:
    """Re-run the committed inputs of ik_<name>.npz through (mujoco, mink, quadprog) and return, per array, the worst
    scaled difference max|new − committed| / max(1, max|committed|) over all instances, with its tolerance."""
    cfgs = _configs(mj, mink, examples)
    report = {}
    for name in names:
        c = cfgs[name]
        fx = np.load(os.path.join(HERE, f"ik_{name}.npz"))
        dt, damping = float(fx["dt"]), float(fx["damping"])
        if c["posture"] is not None:
            c["posture"].set_target(fx["posture_target"])
        rec = {k: [] for k in ("v", "H", "c", "h", "G", "task_e", "task_J")}
        nG = len(fx["G"]) if "G" in fx.files else 0
        nJ = len(fx["task_J"]) if "task_J" in fx.files else 0
        for i, q in enumerate(fx["q"]):
            for t, tg in zip(c["frame_tasks"], fx["frame_targets"][i]):
                t.set_target(mink.SE3(wxyz_xyz=np.asarray(tg, dtype=np.float64)))
            if c["com"] is not None:
                c["com"].set_target(fx["com_target"][i])
            cfg = mink.Configuration(c["model"], q)
            problem = mink.build_ik(cfg, c["tasks"], dt, damping, c["limits"])
            rec["v"].append(mink.solve_ik(cfg, c["tasks"], dt, "quadprog", damping, limits=c["limits"]))
            rec["H"].append(problem.P); rec["c"].append(problem.q)
            rec["h"].append(problem.h if problem.h is not None else np.zeros(0))
            if i < nG and problem.G is not None:
                rec["G"].append(problem.G)
            rec["task_e"].append(np.concatenate([t.compute_error(cfg) for t in c["tasks"]]))
            if i < nJ:
                rec["task_J"].append(np.vstack([t.compute_jacobian(cfg) for t in c["tasks"]]))
        rep = {}
        for k, vals in rec.items():
            if k in fx.files and len(vals):
                w = _worst(np.array(vals), fx[k])
                rep[k] = {"worst": w, "tol": TOL[k], "ok": bool(w <= TOL[k])}
        report[name] = rep
    return report


def diff_models(mj, examples: str) -> dict:
    """Every example scene: FlatModel.from_mjmodel(real MjModel) against the committed FlatModel of the repository's own MJCF
    reader (tests/golden/models/all).  Mesh geoms: the hull's vertex SET (order-free, float32 tolerance)."""
    sys.path.insert(0, REPO) if REPO not in sys.path else None
    from mink_amd.flatmodel import FlatModel
    report = {}
    for p in sorted(glob.glob(os.path.join(examples, "*", "scene*.xml"))):
        name = p.split(os.sep)[-2] + "__" + os.path.basename(p)[:-4]
        ref_path = os.path.join(HERE, "models", "all", name + ".json")
        if not os.path.exists(ref_path):
            continue
        mine = FlatModel.load(ref_path)
        real = FlatModel.from_mjmodel(mj.MjModel.from_xml_path(p))
        rep = {"sizes": {k: [int(getattr(real, k)), int(getattr(mine, k))] for k in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite")}}
        rep["sizes_ok"] = all(a == b for a, b in rep["sizes"].values())
        if rep["sizes_ok"]:
            for f in MODEL_INT_FIELDS:
                rep[f] = {"ok": bool(np.array_equal(np.asarray(getattr(real, f)), np.asarray(getattr(mine, f))))}
            for f, tol in MODEL_FIELDS.items():
                a, b = np.asarray(getattr(real, f), dtype=np.float64), np.asarray(getattr(mine, f), dtype=np.float64)
                if f.endswith("quat") and a.shape == b.shape and a.size:      # (q and −q are one rotation)
                    s = np.sign(np.sum(a * b, axis=-1, keepdims=True)); s[s == 0] = 1.0
                    a = a * s
                w = float(np.abs(a - b).max()) if a.shape == b.shape and a.size else (0.0 if a.shape == b.shape else float("inf"))
                rep[f] = {"worst_abs": w, "tol": tol, "ok": bool(w <= tol)}
            hull_worst = 0.0
            for g in range(real.ngeom):
                if int(real.geom_dataid[g]) < 0:
                    continue
                ha, hb = np.asarray(real.mesh_hull(g), dtype=np.float64), np.asarray(mine.mesh_hull(g), dtype=np.float64)
                d = np.sqrt(((ha[:, None, :] - hb[None, :, :]) ** 2).sum(-1))
                hull_worst = max(hull_worst, float(d.min(axis=1).max()), float(d.min(axis=0).max()))   # (Hausdorff, vertex sets)
            rep["mesh_hull_hausdorff"] = {"worst_abs": hull_worst, "tol": 1e-6, "ok": bool(hull_worst <= 1e-6)}
        report[name] = rep
    return report


def all_ok(report: dict) -> bool:
    def walk(x):
        if isinstance(x, dict):
            if "ok" in x and not x["ok"]:
                return False
            return all(walk(v) for v in x.values())
        return True
    return walk(report) and all(v.get("sizes_ok", True) for v in report.values() if isinstance(v, dict))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", default="/root/reference", help="mink checkout (examples/ with the robot scenes; mink/ unless pip-installed)")
    ap.add_argument("--out", default=None, help="directory for real_wheels_report.json (default: a temporary directory)")
    ap.add_argument("--allow-stubs", action="store_true", help="self-check of this script over oracle/stubs (proves nothing about MuJoCo)")
    a = ap.parse_args(argv)
    w = wheels(a.reference, a.allow_stubs)
    if w is None:
        print("real_wheels: `mujoco` + `qpsolvers` + `quadprog` (and mink) are not importable here — nothing checked.  "
              "Parity against MuJoCo / quadprog stays UNPINNED (DESIGN.md §5).")
        return 2
    mj, mink, kind = w
    examples = os.path.join(a.reference, "examples")
    report = {"wheels": kind, "mujoco_version": getattr(mj, "__version__", "stub"),
              "fixtures": diff_fixtures(mj, mink, examples)}
    if kind == "real":
        report["models"] = diff_models(mj, examples)
    out = a.out or tempfile.mkdtemp(prefix="mkh_real_wheels_")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "real_wheels_report.json")
    json.dump(report, open(path, "w"), indent=1)
    for name, rep in report["fixtures"].items():
        print(name, {k: "%.1e%s" % (v["worst"], "" if v["ok"] else " > %.0e FAIL" % v["tol"]) for k, v in rep.items()})
    for name, rep in report.get("models", {}).items():
        bad = [k for k, v in rep.items() if isinstance(v, dict) and v.get("ok") is False]
        print(name, "ok" if not bad and rep.get("sizes_ok") else ("DIFFERS: " + ", ".join(bad or ["sizes"])))
    ok = all_ok(report["fixtures"]) and all_ok(report.get("models", {}))
    print("real_wheels (%s): %s -> %s" % (kind, "ALL WITHIN TOLERANCE" if ok else "DIFFERENCES FOUND", path))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
