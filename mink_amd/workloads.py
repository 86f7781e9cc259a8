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
    near = np.nonzero(rng.uniform(size=n) < 0.10)[0]
    for i in near:
        for j in rng.choice(limited, size=int(rng.integers(1, 4)), replace=False):
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
