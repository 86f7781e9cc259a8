"""The BASELINE configs as NativeProblem descriptors (ctypes level), mirroring
tests/oracle_configs.py.  Used by the GPU parity tests."""

import numpy as np

import oracle_configs as oc
from mink_amd import _native as nat
from oracle import ik


def _cfg_limit(m, gain=0.95, min_distance=0.0):
    idx, lower, upper = ik.configuration_limit_arrays(m, ik.ConfigurationLimitSpec(gain, min_distance))
    return {"gain": gain, "lower": lower, "upper": upper, "indices": idx}


def _vel_limit(m, vmax=np.pi):
    idx = [int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] != 0]
    return {"indices": idx, "limit": np.full(len(idx), vmax)}


def _ft(m, name, ftype, pos, ori, lm=0.0):
    kind = {"site": "site", "body": "body", "geom": "geom"}[ftype]
    return {"frame_type": kind, "frame_id": m.name2id(kind, name), "cost": [pos] * 3 + [ori] * 3,
            "gain": 1.0, "lm_damping": lm}


def build(name, nmodel, max_batch):
    m = nmodel.model
    if name in ("ur5e_c2", "ur5e_c1"):
        lims = dict(configuration_limits=[_cfg_limit(m)])
        if name == "ur5e_c2":
            lims["velocity_limits"] = [_vel_limit(m)]
        return nat.NativeProblem(nmodel, frame_tasks=[_ft(m, "attachment_site", "site", 1.0, 1.0, 1.0)],
                                 posture_tasks=[{"cost": 1e-2}], max_batch=max_batch, **lims), 2e-3, 1e-3
    if name == "g1_c3":
        fts = [_ft(m, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
              [_ft(m, s, "site", 200.0, 0.0, 1.0) for s in ("left_palm", "right_palm")]
        return nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}],
                                 configuration_limits=[_cfg_limit(m)], velocity_limits=[_vel_limit(m)],
                                 max_batch=max_batch), 5e-3, 1e-1
    if name == "g1_full":
        fts = [_ft(m, "pelvis", "body", 0.0, 10.0)] + \
              [_ft(m, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
              [_ft(m, s, "site", 200.0, 0.0, 1.0) for s in ("left_palm", "right_palm")]
        return nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1.0}],
                                 com_tasks=[{"cost": 200.0}], configuration_limits=[_cfg_limit(m)],
                                 velocity_limits=[_vel_limit(m)], max_batch=max_batch), 5e-3, 1e-1
    if name == "shadow_c4":
        fts = [_ft(m, f, "site", 1.0, 0.0, 1.0) for f in oc.SHADOW_FINGERS]
        pairs = np.load(oc.GOLDEN + "/shadow_c4_geom_pairs.npy")
        col = {"geom_id_pairs": pairs, "gain": 0.85, "minimum_distance_from_collisions": 0.005,
               "collision_detection_distance": 0.03, "bound_relaxation": 0.0}
        return nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}],
                                 configuration_limits=[_cfg_limit(m)], collision_limits=[col],
                                 max_batch=max_batch), 2e-3, 1e-5
    raise KeyError(name)


ROBOT_OF = {"ur5e_c1": "ur5e", "ur5e_c2": "ur5e", "g1_c3": "g1", "g1_full": "g1", "shadow_c4": "shadow_left"}
# row order of the oracle task list → native tap row order (frame tasks, posture, com)
