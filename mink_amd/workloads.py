"""Synthetic (q, target) batches for the BASELINE configs.

Host-side data generation only (numpy RNG); the forward kinematics that turns a
perturbed configuration into reachable frame targets runs on the device through the
same C ABI as the solve.
"""

from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np

from . import _native as nat
from .flatmodel import JNT_FREE, FlatModel

_ROBOTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robots")


def load_robot(name: str) -> FlatModel:
    """Packaged FlatModel of a benchmark robot: 'ur5e', 'g1', 'shadow_left', 'h1'."""
    return FlatModel.load(os.path.join(_ROBOTS, f"{name}.json"))


def _so3_exp(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w, axis=-1, keepdims=True)
    half = 0.5 * th
    k = np.where(th < 1e-8, 0.5 - th * th / 48.0, np.sin(half) / np.where(th < 1e-8, 1.0, th))
    return np.concatenate([np.cos(half), k * w], axis=-1)


def sample_q(model: FlatModel, rng: np.random.Generator, n: int, base_q=None) -> np.ndarray:
    """Hinge/slide joints uniform inside 90 % of their range, 10 % of the instances with 1–3
    joints within 1e-3·range of a bound (active box constraints); free joint near the keyframe."""
    q = np.tile(np.asarray(model.qpos0 if base_q is None else base_q, dtype=np.float64), (n, 1))
    limited = []
    for j in range(model.njnt):
        a = int(model.jnt_qposadr[j])
        if model.jnt_type[j] == JNT_FREE:
            q[:, a:a + 3] = np.array([0.0, 0.0, 0.75]) + rng.normal(scale=0.05, size=(n, 3))
            q[:, a + 3:a + 7] = _so3_exp(rng.normal(scale=0.2, size=(n, 3)))
        else:
            lo, hi = model.jnt_range[j] if model.jnt_limited[j] else (-np.pi, np.pi)
            w = hi - lo
            q[:, a] = rng.uniform(lo + 0.05 * w, hi - 0.05 * w, size=n)
            if model.jnt_limited[j]:
                limited.append(j)
    near = np.nonzero(rng.uniform(size=n) < 0.10)[0] if limited else []
    for i in near:
        for j in rng.choice(limited, size=min(len(limited), int(rng.integers(1, 4))), replace=False):
            lo, hi = model.jnt_range[j]
            eps = 1e-3 * (hi - lo) * rng.uniform()
            q[i, int(model.jnt_qposadr[j])] = (lo + eps) if rng.uniform() < 0.5 else (hi - eps)
    return q


def g1_config(model: FlatModel, nmodel: "nat.NativeModel", max_batch: int):
    """BASELINE config 3: G1, 4 FrameTasks (feet pos 200/ori 10, palms pos 200/ori 0, lm 1) +
    PostureTask(1) + ConfigurationLimit + VelocityLimit(π) — examples/humanoid_g1.py:28-52,80,88."""
    from .api_specs import configuration_limit_desc, velocity_limit_desc

    def ft(site, pos, ori):
        return {"frame_type": "site", "frame_id": model.name2id("site", site), "cost": [pos] * 3 + [ori] * 3,
                "gain": 1.0, "lm_damping": 1.0}

    fts = [ft("left_foot", 200.0, 10.0), ft("right_foot", 200.0, 10.0), ft("left_palm", 200.0, 0.0),
           ft("right_palm", 200.0, 0.0)]
    hinge = {model.jnt_names[j]: np.pi for j in range(model.njnt) if model.jnt_type[j] != JNT_FREE}
    prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}],
                             configuration_limits=[configuration_limit_desc(model)],
                             velocity_limits=[velocity_limit_desc(model, hinge)], max_batch=max_batch)
    return prob, 5e-3, 1e-1


def _frame_desc(model: FlatModel, name: str, ftype: str, pos: float, ori: float, lm: float = 0.0) -> dict:
    return {"frame_type": ftype, "frame_id": model.name2id(ftype, name), "cost": [pos] * 3 + [ori] * 3,
            "gain": 1.0, "lm_damping": lm}


def _hinge_velocities(model: FlatModel, vmax: float = np.pi) -> dict:
    return {model.jnt_names[j]: vmax for j in range(model.njnt) if model.jnt_type[j] != JNT_FREE}


SHADOW_FINGERS = ("thumb", "first", "middle", "ring", "little")
G1_HANDS_TIPS = ("ff_tip", "mf_tip", "rf_tip", "th_tip")

# BASELINE.json configs: robot, keyframe, default batch, algorithmic bytes per solve
# (q + frame targets + v + status, shared posture target; G1 full adds nothing per instance when the
# CoM target is shared)
BENCH_CONFIGS: Dict[str, dict] = {
    "ur5e_c2": {"robot": "ur5e", "key": "home", "batch": 4096, "bytes_per_solve": 6 * 8 + 7 * 8 + 6 * 8 + 4,
                "workload": "UR5e (nq=nv=6): FrameTask(attachment_site)+PostureTask+ConfigurationLimit+"
                            "VelocityLimit, dt=2e-3, damping=1e-3 (BASELINE configs[1])"},
    "g1_c3": {"robot": "g1", "key": "stand", "batch": 65536, "bytes_per_solve": 44 * 8 + 4 * 7 * 8 + 43 * 8 + 4,
              "workload": "Unitree G1 (nq=44,nv=43): 4 FrameTasks(feet+palms)+PostureTask+"
                          "ConfigurationLimit+VelocityLimit, dt=5e-3, damping=1e-1 (BASELINE configs[2])"},
    "g1_full": {"robot": "g1", "key": "stand", "batch": 65536,
                "bytes_per_solve": 44 * 8 + 5 * 7 * 8 + 3 * 8 + 43 * 8 + 4,
                "workload": "Unitree G1 full example: pelvis-orientation + 4 FrameTasks + PostureTask + "
                            "ComTask(per-instance target) + box limits (examples/humanoid_g1.py:22-52)"},
    "shadow_c4": {"robot": "shadow_left", "key": "grasp hard", "batch": 16384,
                  "bytes_per_solve": 24 * 8 + 5 * 7 * 8 + 24 * 8 + 4,
                  "workload": "Shadow Hand left (nq=nv=24): 5 fingertip FrameTasks+PostureTask+ConfigurationLimit+"
                              "CollisionAvoidanceLimit(40 capsule pairs, detect 30 mm), dt=2e-3, damping=1e-5 "
                              "(BASELINE configs[3])"},
    # The two general routes of the boundary (not BASELINE configs; measured so that no path behind the C ABI is untimed):
    # the plugin route of mink's Task / Limit API, and a collision pair that needs the general convex routine.
    "g1_plugin": {"robot": "g1", "key": "stand", "batch": 65536,
                  "bytes_per_solve": 44 * 8 + 4 * 7 * 8 + 43 * 8 + 4 + 8 * (3 + 3 * 43 + 2 * 43 + 2),
                  "workload": "G1 config 3 + one caller-defined Task (3 rows: head position, e and J handed over per instance) + "
                              "one caller-defined Limit (2 general rows G·Δq ≤ h per instance) through mkh_solve_dense "
                              "(mink/tasks/task.py:81-138, limits/limit.py:34-57)"},
    "ur5e_convex": {"robot": "ur5e", "key": "home", "batch": 4096, "bytes_per_solve": 6 * 8 + 7 * 8 + 6 * 8 + 4,
                    "workload": "UR5e, the collision set-up of examples/arm_ur5e.py:20-47 with the wrist_3_link geom as a "
                                "CYLINDER (menagerie's eef_collision class): FrameTask + ConfigurationLimit + "
                                "CollisionAvoidanceLimit(cylinder-plane floor, cylinder-box wall: general convex routine) + "
                                "VelocityLimit, dt=5e-2, damping=1e-3"},
    # G1 config 3 with a CollisionAvoidanceLimit over 46 pairs of the model's own primitive collision geoms (foot spheres, knee
    # and shoulder cylinders, finger boxes, the floor; every pair analytic).  nv + pairs = 89 > 64: the wavefront kernel holds
    # the 21 tightest contacts, the instances in which a dropped one is violated are solved again with every row by the
    # workgroup-per-problem kernel (wide_kernel.h) — the path that lifts the one-wavefront limit, driver-measured
    "g1_coll": {"robot": "g1", "key": "stand", "batch": 16384, "bytes_per_solve": 44 * 8 + 4 * 7 * 8 + 43 * 8 + 4,
                "workload": "Unitree G1 config 3 + CollisionAvoidanceLimit(46 analytic pairs of the model's primitive collision geoms: "
                            "foot spheres x floor / opposite foot / opposite knee cylinder / opposite finger box, cylinders and boxes x "
                            "floor; detect 0.25 m, d_min 5 mm), dt=5e-3, damping=1e-1"},
    # a mid-size robot (17–32 dofs: H1, Go1, Spot, Allegro — one wavefront per problem with most lanes idle): Unitree H1 under
    # the G1 task set, examples/humanoid_h1.py:22-52 without the pelvis / CoM tasks
    "h1_c3": {"robot": "h1", "key": "stand", "batch": 65536, "bytes_per_solve": 26 * 8 + 4 * 7 * 8 + 25 * 8 + 4,
              "workload": "Unitree H1 (nq=26,nv=25): 4 FrameTasks(feet pos 200/ori 10, wrists pos 200/ori 0, lm 1)+PostureTask(1)+"
                          "ConfigurationLimit+VelocityLimit(pi), dt=5e-3, damping=1e-1 (examples/humanoid_h1.py:22-52)"},
    # ... and the reference's H1 example as written (examples/humanoid_h1.py:22-52): + pelvis orientation + ComTask, per-instance CoM target
    "h1_full": {"robot": "h1", "key": "stand", "batch": 65536, "bytes_per_solve": 26 * 8 + 5 * 7 * 8 + 3 * 8 + 25 * 8 + 4,
                "workload": "Unitree H1 full example: pelvis-orientation + 4 FrameTasks + PostureTask + ComTask(per-instance target) + "
                            "box limits (examples/humanoid_h1.py:22-52), dt=5e-3, damping=1e-1"},
    # A model beyond one wavefront (round 5): the G1 with a 16-dof Allegro hand attached to each wrist (mink_amd/compose.py) —
    # 75 dofs, 86 bodies — under the G1 task set plus the arm + hand examples' fingertip tasks (one RelativeFrameTask per
    # fingertip, measured in its palm: examples/arm_hand_iiwa_allegro.py:77-94).  Every call runs on the workgroup-per-problem
    # kernel (wide_kernel.h)
    "g1_hands": {"robot": "g1_hands", "key": "stand", "batch": 8192, "bytes_per_solve": 76 * 8 + 12 * 7 * 8 + 75 * 8 + 4,
                 "workload": "Unitree G1 + two Allegro hands (nq=76,nv=75, 86 bodies): 4 FrameTasks(feet+palms) + 8 RelativeFrameTasks "
                             "(fingertips in their palm, pos 1) + PostureTask + ConfigurationLimit + VelocityLimit, dt=5e-3, damping=1e-1 "
                             "(examples/humanoid_g1.py:28-52 + examples/arm_hand_iiwa_allegro.py:62-94)"},
    # The reference's flagship collision example as written (examples/arm_aloha.py:76-121, 146-157): two FrameTasks on the grippers,
    # PostureTask(1e-4), ConfigurationLimit, VelocityLimit(π), CollisionAvoidanceLimit over 1 104 geom pairs (wrist subtree x wrist
    # subtree, both arms x metal frame + table; capsules fitted to meshes, spheres, the table box), d_min 5 cm, detection 10 cm,
    # dt = 1 / 200 s, damping 1e-5.  16 dofs: the wavefront kernel holds the 48 tightest contacts, the rest is checked at the solution
    "aloha_coll": {"robot": "aloha", "key": "neutral_pose", "batch": 16384, "bytes_per_solve": 16 * 8 + 2 * 7 * 8 + 16 * 8 + 4,
                   "workload": "ALOHA (nq=nv=16): 2 FrameTasks(grippers)+PostureTask(1e-4)+ConfigurationLimit+VelocityLimit(pi)+"
                               "CollisionAvoidanceLimit(1104 pairs: 521 capsule-capsule, 504 sphere-capsule, 49 sphere-sphere, 30 x box; "
                               "d_min 0.05, detect 0.1), dt=5e-3, damping=1e-5 (examples/arm_aloha.py:76-157)"},
    # the same set-up with the packaged model's CAPSULE wrist geom (analytic pairs only): the reference point of ur5e_convex
    "ur5e_coll": {"robot": "ur5e", "key": "home", "batch": 4096, "bytes_per_solve": 6 * 8 + 7 * 8 + 6 * 8 + 4,
                  "workload": "UR5e, the collision set-up of examples/arm_ur5e.py:20-47 (capsule-plane floor, capsule-box wall), "
                              "dt=5e-2, damping=1e-3"},
}


def g1_with_hands() -> FlatModel:
    """The G1 carrying a 16-dof Allegro hand on each wrist — `attach_site.attach(hand)` of the reference's arm + hand examples
    (examples/arm_hand_iiwa_allegro.py:32-42), twice: 43 + 32 = 75 dofs, 86 bodies (the model of the `g1_hands` workload: past
    one wavefront in both counts).  The left-hand model serves both sides (the packaged right hand carries no fingertip sites)."""
    from .compose import attach
    hand = load_robot("allegro_left")
    m = attach(load_robot("g1"), hand, site="left_palm", prefix="lh/", pos=(0.0, 0.0, 0.02))
    return attach(m, hand, site="right_palm", prefix="rh/", pos=(0.0, 0.0, 0.02))


def load_bench_robot(name: str) -> FlatModel:
    """FlatModel of a bench config (BENCH_CONFIGS key); `ur5e_convex` turns the wrist_3_link capsule into a cylinder."""
    if BENCH_CONFIGS[name]["robot"] == "g1_hands":
        return g1_with_hands()
    model = load_robot(BENCH_CONFIGS[name]["robot"])
    if name == "ur5e_convex":
        from .flatmodel import GEOM_CYLINDER
        model.geom_type = model.geom_type.copy()
        model.geom_type[model.name2id("geom", "wrist_3_link")] = GEOM_CYLINDER
    return model


def g1_collision_pairs(model: FlatModel):
    """46 geom-id pairs among G1's primitive collision geoms (bodies of the robot only), every one with an analytic distance
    routine: the eight foot spheres against the floor (8), left against right foot spheres (16), knee / shoulder cylinders and
    finger boxes against the floor (6), foot spheres against the opposite knee cylinder (8) and the opposite finger box (8)."""
    gt, gv, gb = np.asarray(model.geom_type), np.asarray(model.geom_valid), np.asarray(model.geom_bodyid)
    robot = [g for g in range(model.ngeom) if gv[g] == 1 and model.body_names[int(gb[g])].endswith("_link")]
    floor = [g for g in range(model.ngeom) if gt[g] == 0][0]
    side = lambda g: model.body_names[int(gb[g])].split("_")[0]
    sph = {s: [g for g in robot if gt[g] == 2 and side(g) == s] for s in ("left", "right")}
    knee = {s: [g for g in robot if gt[g] == 5 and side(g) == s and "knee" in model.body_names[int(gb[g])]] for s in ("left", "right")}
    cyl = [g for g in robot if gt[g] == 5]
    box = {s: [g for g in robot if gt[g] == 6 and side(g) == s] for s in ("left", "right")}
    pairs = [(g, floor) for g in sph["left"] + sph["right"]]
    pairs += [(a, b) for a in sph["left"] for b in sph["right"]]
    pairs += [(g, floor) for g in cyl + box["left"] + box["right"]]
    for s, o in (("left", "right"), ("right", "left")):
        pairs += [(a, b) for a in sph[s] for b in knee[o]] + [(a, b) for a in sph[s] for b in box[o]]
    return pairs


def bench_config(name: str, model: FlatModel, nmodel: "nat.NativeModel", max_batch: int):
    """NativeProblem, dt, damping of a BASELINE config (same descriptors as tests/native_configs.py, built from
    the product's own constructors so that bench.py does not touch tests/ or oracle/)."""
    from .api_specs import configuration_limit_desc, velocity_limit_desc

    cfg = [configuration_limit_desc(model)]
    if name == "g1_c3":
        return g1_config(model, nmodel, max_batch)
    if name in ("h1_c3", "h1_full"):
        fts = [_frame_desc(model, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
              [_frame_desc(model, s, "site", 200.0, 0.0, 1.0) for s in ("left_wrist", "right_wrist")]
        extra = {}
        if name == "h1_full":
            fts = [_frame_desc(model, "pelvis", "body", 0.0, 10.0)] + fts
            extra = {"com_tasks": [{"cost": 200.0}]}
        prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}], configuration_limits=cfg,
                                 velocity_limits=[velocity_limit_desc(model, _hinge_velocities(model))], max_batch=max_batch, **extra)
        return prob, 5e-3, 1e-1
    if name == "aloha_coll":
        from .limits import CollisionAvoidanceLimit
        from .utils import get_body_geom_ids, get_subtree_geom_ids

        sub = lambda b: get_subtree_geom_ids(model, model.name2id("body", b))
        frame = get_body_geom_ids(model, model.name2id("body", "metal_frame"))
        col = CollisionAvoidanceLimit(model, [(sub("left/wrist_link"), sub("right/wrist_link")),
                                              (sub("left/upper_arm_link") + sub("right/upper_arm_link"), frame + ["table"])],
                                      minimum_distance_from_collisions=0.05, collision_detection_distance=0.1)
        joints = ("waist", "shoulder", "elbow", "forearm_roll", "wrist_angle", "wrist_rotate")
        vel = velocity_limit_desc(model, {f"{p}/{n}": np.pi for p in ("left", "right") for n in joints})
        prob = nat.NativeProblem(nmodel, frame_tasks=[_frame_desc(model, f"{p}/gripper", "site", 1.0, 1.0, 1.0) for p in ("left", "right")],
                                 posture_tasks=[{"cost": 1e-4}], configuration_limits=cfg, velocity_limits=[vel],
                                 collision_limits=[col._native_desc()[1]], max_batch=max_batch)
        return prob, 5e-3, 1e-5
    if name == "g1_hands":
        fts = [_frame_desc(model, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
              [_frame_desc(model, s, "site", 200.0, 0.0, 1.0) for s in ("left_palm", "right_palm")]
        for side in ("lh", "rh"):
            for tip in G1_HANDS_TIPS:
                d = _frame_desc(model, f"{side}/{tip}", "site", 1.0, 0.0, 1.0)
                d.update(root_type="body", root_id=model.name2id("body", f"{side}/palm"))
                fts.append(d)
        prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}], configuration_limits=cfg,
                                 velocity_limits=[velocity_limit_desc(model, _hinge_velocities(model))], max_batch=max_batch)
        return prob, 5e-3, 1e-1
    if name == "g1_coll":
        fts = [_frame_desc(model, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
              [_frame_desc(model, s, "site", 200.0, 0.0, 1.0) for s in ("left_palm", "right_palm")]
        col = {"geom_id_pairs": np.array(g1_collision_pairs(model)), "gain": 0.85, "minimum_distance_from_collisions": 0.005,
               "collision_detection_distance": 0.25, "bound_relaxation": 0.0}
        prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}], configuration_limits=cfg,
                                 velocity_limits=[velocity_limit_desc(model, _hinge_velocities(model))],
                                 collision_limits=[col], max_batch=max_batch)
        return prob, 5e-3, 1e-1
    if name == "ur5e_c2":
        prob = nat.NativeProblem(nmodel, frame_tasks=[_frame_desc(model, "attachment_site", "site", 1.0, 1.0, 1.0)],
                                 posture_tasks=[{"cost": 1e-2}], configuration_limits=cfg,
                                 velocity_limits=[velocity_limit_desc(model, _hinge_velocities(model))],
                                 max_batch=max_batch)
        return prob, 2e-3, 1e-3
    if name == "g1_full":
        fts = [_frame_desc(model, "pelvis", "body", 0.0, 10.0)] + \
              [_frame_desc(model, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
              [_frame_desc(model, s, "site", 200.0, 0.0, 1.0) for s in ("left_palm", "right_palm")]
        prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}], com_tasks=[{"cost": 200.0}],
                                 configuration_limits=cfg,
                                 velocity_limits=[velocity_limit_desc(model, _hinge_velocities(model))],
                                 max_batch=max_batch)
        return prob, 5e-3, 1e-1
    if name == "shadow_c4":
        from .limits import CollisionAvoidanceLimit

        groups = [[f"{f}_1", f"{f}_2"] for f in SHADOW_FINGERS]
        pairs = [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)]
        col = CollisionAvoidanceLimit(model, pairs, collision_detection_distance=0.03)
        fts = [_frame_desc(model, f, "site", 1.0, 0.0, 1.0) for f in SHADOW_FINGERS]
        prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=cfg,
                                 collision_limits=[col._native_desc()[1]], max_batch=max_batch)
        return prob, 2e-3, 1e-5
    if name == "g1_plugin":
        fts = [_frame_desc(model, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
              [_frame_desc(model, s, "site", 200.0, 0.0, 1.0) for s in ("left_palm", "right_palm")]
        prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}], configuration_limits=cfg,
                                 velocity_limits=[velocity_limit_desc(model, _hinge_velocities(model))],
                                 dense_tasks=[{"cost": np.full(3, 50.0), "gain": 1.0, "lm_damping": 0.0}],
                                 dense_limit_rows=2, max_batch=max_batch)
        return prob, 5e-3, 1e-1
    if name in ("ur5e_convex", "ur5e_coll"):
        from .limits import CollisionAvoidanceLimit

        col = CollisionAvoidanceLimit(model, [(["wrist_3_link"], ["floor", "wall"])], collision_detection_distance=0.3)
        prob = nat.NativeProblem(nmodel, frame_tasks=[_frame_desc(model, "attachment_site", "site", 1.0, 1.0, 1.0)],
                                 configuration_limits=cfg, collision_limits=[col._native_desc()[1]],
                                 velocity_limits=[velocity_limit_desc(model, _hinge_velocities(model))],
                                 max_batch=max_batch)
        return prob, 5e-2, 1e-3
    raise KeyError(name)


def plugin_rows(model: FlatModel, nmodel, q: np.ndarray, rng: np.random.Generator) -> dict:
    """Per-call arrays of the `g1_plugin` workload, as a caller of mkh_solve_dense would hand them over: a user task
    "bring the head site 5 cm forward" (e, J = position rows of the head frame's task, evaluated by the caller at q — here
    through the taps of a one-task problem) and a user limit with two general rows (the summed joint steps of each arm)."""
    B = len(q)
    head = nat.NativeProblem(nmodel, frame_tasks=[_frame_desc(model, "head", "site", 1.0, 1.0)], max_batch=B)
    dummy = np.zeros((B, 1, 7)); dummy[:, :, 0] = 1.0
    _, _, t = head.solve(q, dummy, None, None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
    tgt = t["frame_pose"].copy()
    tgt[:, 0, 4] += 0.05
    _, _, t = head.solve(q, tgt, None, None, 1.0, 1.0, taps=["task_e", "task_J"], solve_qp=False)
    head.close()
    G = np.zeros((B, 2, model.nv))
    names = list(model.jnt_names)
    for r, side in enumerate(("left", "right")):
        for j, n in enumerate(names):
            if n.startswith(side) and ("shoulder" in n or "elbow" in n):
                G[:, r, int(model.jnt_dofadr[j])] = 1.0
    assert (G != 0).sum(axis=2).min() >= 2, "arm rows must be general (multi-entry) rows"
    h = np.full((B, 2), 8e-3) + 1e-3 * rng.uniform(size=(B, 2))
    return {"task_e": np.ascontiguousarray(t["task_e"][:, :3]), "task_J": np.ascontiguousarray(t["task_J"][:, :3]),
            "limit_G": G, "limit_h": h}


def bench_batch(name: str, model: FlatModel, nmodel, prob, rng: np.random.Generator, n: int, reachable: bool = False):
    """(q, frame_targets, posture_target, com_target) of a BASELINE config, host arrays (SURVEY §8d distributions);
    com_target is per instance (the instance's own CoM + 1 cm) for the G1 full example, else None."""
    c = BENCH_CONFIGS[name]
    base = model.key_qpos[model.name2id("key", c["key"])]
    q, tg = make_batch(model, nmodel, prob, rng, n, base_q=base, dense=name == "g1_plugin", reachable=reachable)
    if reachable:
        # (consistent targets: the posture target is the configuration the frame targets were taken from — per instance)
        tg, q2 = tg
        return q, tg, q2[:, None, :].copy(), None
    if c["robot"] == "shadow_left":
        q[::2] = 0.5 * (q[::2] + base)            # half of the samples near the grasp: fingers come close
    if name in ("ur5e_convex", "ur5e_coll"):
        q[::2] = base + rng.normal(scale=0.4, size=q[::2].shape)     # half of the samples around `home`: near wall and floor
    if name == "aloha_coll":
        # around the neutral pose (σ = 0.5 rad on half of the instances, 0.25 on the rest): arms near each other, the frame and the table
        sc = np.where(np.arange(n) % 2 == 0, 0.5, 0.25)[:, None]
        q = np.clip(base + rng.normal(size=q.shape) * sc, *_joint_box(model))
    com = None
    if prob.n_com:
        _, _, t = prob.solve(q, tg, base[None, :], np.zeros((1, 3)), 1.0, 1.0, taps=["subtree_com"], solve_qp=False)
        com = t["subtree_com"][:, None, :] + 0.01
    return q, tg, base[None, :].copy(), com


def _joint_box(model: FlatModel):
    """(lower, upper) per qpos of the hinge / slide joints, ±inf elsewhere."""
    lo, hi = np.full(model.nq, -np.inf), np.full(model.nq, np.inf)
    for j in range(model.njnt):
        if model.jnt_limited[j] and model.jnt_type[j] in (2, 3):
            a = int(model.jnt_qposadr[j])
            lo[a], hi[a] = model.jnt_range[j]
    return lo, hi


def bench_dense(name: str, model: FlatModel, nmodel, q: np.ndarray, rng: np.random.Generator):
    """Per-call plugin arrays of a bench config (None for every config but `g1_plugin`)."""
    return plugin_rows(model, nmodel, q, rng) if name == "g1_plugin" else None


def make_batch(model: FlatModel, nmodel, prob, rng: np.random.Generator, n: int, base_q=None,
               sigma: float = 0.15, dense: bool = False, reachable: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """q and frame targets = FK(q ⊕ δ), δ ~ N(0, σ²) per dof (device FK).  SURVEY §8(d)'s distribution does not keep q ⊕ δ
    inside the joint ranges (with 37 limited hinges some joint of almost every G1 instance ends outside: the target is then
    not exactly attainable under ConfigurationLimit); `reachable` clips the perturbed configuration into 96 % of every
    range — the targets of the convergence figure of bench.py."""
    q = sample_q(model, rng, n, base_q)
    delta = rng.normal(scale=sigma, size=(n, model.nv))
    q2 = nmodel.integrate(q, delta, 1.0)
    if reachable:
        for j in range(model.njnt):
            if model.jnt_type[j] != JNT_FREE and model.jnt_limited[j] and model.jnt_type[j] in (2, 3):
                a = int(model.jnt_qposadr[j])
                lo, hi = model.jnt_range[j]
                q2[:, a] = np.clip(q2[:, a], lo + 0.02 * (hi - lo), hi - 0.02 * (hi - lo))
    q2_keep = q2.copy()
    pt = np.zeros((prob.n_posture, model.nq)) if prob.n_posture else None
    ct = np.zeros((prob.n_com, 3)) if prob.n_com else None
    dummy = np.zeros((n, prob.n_frame, 7))
    dummy[:, :, 0] = 1.0
    if dense:        # (a problem with plugin rows only runs through mkh_solve_dense: FK of the targets on a plain twin)
        twin = nat.NativeProblem(nmodel, frame_tasks=[{"frame_type": "site", "frame_id": model.name2id("site", s),
                                                       "cost": [1.0] * 6} for s in ("left_foot", "right_foot", "left_palm", "right_palm")],
                                 max_batch=n)
        _, _, taps = twin.solve(q2, dummy, None, None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
        twin.close()
        return q, taps["frame_pose"]
    _, _, taps = prob.solve(q2, dummy, pt, ct, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
    if reachable:
        return q, (taps["frame_pose"], q2_keep)
    return q, taps["frame_pose"]
