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
    """Packaged FlatModel of a benchmark robot: 'ur5e', 'g1', 'shadow_left'."""
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
}


def bench_config(name: str, model: FlatModel, nmodel: "nat.NativeModel", max_batch: int):
    """NativeProblem, dt, damping of a BASELINE config (same descriptors as tests/native_configs.py, built from
    the product's own constructors so that bench.py does not touch tests/ or oracle/)."""
    from .api_specs import configuration_limit_desc, velocity_limit_desc

    cfg = [configuration_limit_desc(model)]
    if name == "g1_c3":
        return g1_config(model, nmodel, max_batch)
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
    raise KeyError(name)


def bench_batch(name: str, model: FlatModel, nmodel, prob, rng: np.random.Generator, n: int):
    """(q, frame_targets, posture_target, com_target) of a BASELINE config, host arrays (SURVEY §8d distributions);
    com_target is per instance (the instance's own CoM + 1 cm) for the G1 full example, else None."""
    c = BENCH_CONFIGS[name]
    base = model.key_qpos[model.name2id("key", c["key"])]
    q, tg = make_batch(model, nmodel, prob, rng, n, base_q=base)
    if c["robot"] == "shadow_left":
        q[::2] = 0.5 * (q[::2] + base)            # half of the samples near the grasp: fingers come close
    com = None
    if prob.n_com:
        _, _, t = prob.solve(q, tg, base[None, :], np.zeros((1, 3)), 1.0, 1.0, taps=["subtree_com"], solve_qp=False)
        com = t["subtree_com"][:, None, :] + 0.01
    return q, tg, base[None, :].copy(), com


def make_batch(model: FlatModel, nmodel, prob, rng: np.random.Generator, n: int, base_q=None,
               sigma: float = 0.15) -> Tuple[np.ndarray, np.ndarray]:
    """q and reachable frame targets = FK(q ⊕ δ), δ ~ N(0, σ²) per dof (device FK)."""
    q = sample_q(model, rng, n, base_q)
    delta = rng.normal(scale=sigma, size=(n, model.nv))
    q2 = nmodel.integrate(q, delta, 1.0)
    pt = np.zeros((prob.n_posture, model.nq)) if prob.n_posture else None
    ct = np.zeros((prob.n_com, 3)) if prob.n_com else None
    dummy = np.zeros((n, prob.n_frame, 7))
    dummy[:, :, 0] = 1.0
    _, _, taps = prob.solve(q2, dummy, pt, ct, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
    return q, taps["frame_pose"]
