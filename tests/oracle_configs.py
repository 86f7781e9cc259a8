"""BASELINE.json configs expressed as oracle specs.  Shared by the
oracle tests (CPU) and the GPU parity tests; robot models come from the committed
FlatModel fixtures tests/golden/models/*.json."""

import os

import numpy as np

from mink_amd.flatmodel import FlatModel
from oracle import ik

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_models = {}


def model(name: str) -> FlatModel:
    if name not in _models:
        _models[name] = FlatModel.load(os.path.join(GOLDEN, "models", f"{name}.json"))
    return _models[name]


def _cost6(pos, ori):
    return np.array([pos] * 3 + [ori] * 3, dtype=np.float64)


def _hinge_velocity_limit(m, vmax=np.pi):
    idx = [int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] != 0]
    return ik.VelocityLimitSpec(np.array(idx), np.full(len(idx), vmax))


def ur5e_c2(frame_targets, posture_target):
    m = model("ur5e")
    tasks = [
        ik.FrameTaskSpec(m.name2id("site", "attachment_site"), "site", _cost6(1.0, 1.0),
                         frame_targets[0], lm_damping=1.0),
        ik.PostureTaskSpec(np.full(m.nv, 1e-2), posture_target),
    ]
    limits = [ik.ConfigurationLimitSpec(), _hinge_velocity_limit(m)]
    return m, tasks, limits, 2e-3, 1e-3


def ur5e_c1(frame_targets, posture_target):
    m, tasks, _, dt, damping = ur5e_c2(frame_targets, posture_target)
    return m, tasks, None, dt, damping


def g1_c3(frame_targets, posture_target):
    m = model("g1")
    tasks = []
    for k, s in enumerate(("left_foot", "right_foot")):
        tasks.append(ik.FrameTaskSpec(m.name2id("site", s), "site", _cost6(200.0, 10.0),
                                      frame_targets[k], lm_damping=1.0))
    for k, s in enumerate(("left_palm", "right_palm")):
        tasks.append(ik.FrameTaskSpec(m.name2id("site", s), "site", _cost6(200.0, 0.0),
                                      frame_targets[2 + k], lm_damping=1.0))
    tasks.append(ik.PostureTaskSpec(np.full(m.nv, 1.0), posture_target))
    limits = [ik.ConfigurationLimitSpec(), _hinge_velocity_limit(m)]
    return m, tasks, limits, 5e-3, 1e-1


def g1_full(frame_targets, posture_target, com_target):
    m = model("g1")
    tasks = [
        ik.FrameTaskSpec(m.name2id("body", "pelvis"), "body", _cost6(0.0, 10.0), frame_targets[0]),
        ik.PostureTaskSpec(np.full(m.nv, 1.0), posture_target),
        ik.ComTaskSpec(np.full(3, 200.0), com_target),
    ]
    for k, s in enumerate(("left_foot", "right_foot")):
        tasks.append(ik.FrameTaskSpec(m.name2id("site", s), "site", _cost6(200.0, 10.0),
                                      frame_targets[1 + k], lm_damping=1.0))
    for k, s in enumerate(("left_palm", "right_palm")):
        tasks.append(ik.FrameTaskSpec(m.name2id("site", s), "site", _cost6(200.0, 0.0),
                                      frame_targets[3 + k], lm_damping=1.0))
    limits = [ik.ConfigurationLimitSpec(), _hinge_velocity_limit(m)]
    return m, tasks, limits, 5e-3, 1e-1


SHADOW_FINGERS = ("thumb", "first", "middle", "ring", "little")


def shadow_c4(frame_targets, posture_target):
    m = model("shadow_left")
    tasks = [ik.PostureTaskSpec(np.full(m.nv, 1e-2), posture_target)]
    for k, f in enumerate(SHADOW_FINGERS):
        tasks.append(ik.FrameTaskSpec(m.name2id("site", f), "site", _cost6(1.0, 0.0),
                                      frame_targets[k], lm_damping=1.0))
    pairs = [tuple(p) for p in np.load(os.path.join(GOLDEN, "shadow_c4_geom_pairs.npy"))]
    limits = [ik.ConfigurationLimitSpec(),
              ik.CollisionAvoidanceLimitSpec(pairs, collision_detection_distance=0.03)]
    return m, tasks, limits, 2e-3, 1e-5
