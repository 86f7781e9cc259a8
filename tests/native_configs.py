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


# ------------------------------------------------------------- round-2 fixtures (tests/golden/make_golden_ext.py)
def build_ext(name, nmodel, d, max_batch):
    """NativeProblem + (frame_targets, posture_target, com_target) + golden-row → native-tap-row permutation."""
    m = nmodel.model
    nv = m.nv
    if name == "g1_ext":
        def rel(f, ft_, r, rt, cost, gain, lm):
            x = {"frame_type": ft_, "frame_id": m.name2id(ft_, f), "cost": cost, "gain": gain, "lm_damping": lm,
                 "root_type": rt, "root_id": m.name2id(rt, r)}
            return x
        fts = [rel("left_palm", "site", "right_palm", "site", [100.0] * 3 + [5.0] * 3, 0.8, 0.5),
               rel("left_foot", "site", "pelvis", "body", [50.0, 80.0, 120.0, 0, 0, 0], 1.0, 1.0),
               _ft(m, "torso_link", "body", 0.0, 4.0), _ft(m, "right_foot", "site", 200.0, 10.0, 1.0)]
        prob = nat.NativeProblem(nmodel, frame_tasks=fts,
                                 posture_tasks=[{"cost": 0.3, "gain": 0.0}, {"cost": d["posture_cost"]}],
                                 com_tasks=[{"cost": [200.0, 200.0, 50.0], "gain": 0.9}],
                                 configuration_limits=[_cfg_limit(m, 0.9, 0.01)], velocity_limits=[_vel_limit(m)],
                                 max_batch=max_batch)
        o = {"rel_hands": 0, "damp": 6, "rel_foot": 6 + nv, "post": 12 + nv, "torso": 12 + 2 * nv, "com": 18 + 2 * nv,
             "rfoot": 21 + 2 * nv}
        rows = (list(range(o["rel_hands"], o["rel_hands"] + 6)) + list(range(o["rel_foot"], o["rel_foot"] + 6)) +
                list(range(o["torso"], o["torso"] + 6)) + list(range(o["rfoot"], o["rfoot"] + 6)) +
                list(range(o["damp"], o["damp"] + nv)) + list(range(o["post"], o["post"] + nv)) +
                list(range(o["com"], o["com"] + 3)))
        return prob, (d["frame_targets"], d["posture_targets"], d["com_targets"]), np.array(rows), 5e-3, 1e-2
    if name == "ur5e_coll":
        g = {"frame_type": "geom", "frame_id": m.name2id("geom", "wrist_2_link"), "cost": [0.5, 0.5, 0.5, 0.1, 0.2, 0.3],
             "gain": 0.7, "lm_damping": 0.0}
        col = {"geom_id_pairs": d["geom_id_pairs"], "gain": 0.85, "minimum_distance_from_collisions": 0.005,
               "collision_detection_distance": 0.3, "bound_relaxation": 0.0}
        prob = nat.NativeProblem(nmodel, frame_tasks=[_ft(m, "attachment_site", "site", 1.0, 1.0, 1.0), g],
                                 configuration_limits=[_cfg_limit(m)], collision_limits=[col],
                                 velocity_limits=[_vel_limit(m)], max_batch=max_batch)
        return prob, (d["frame_targets"], None, None), None, 5e-2, 1e-3
    if name == "ballslide":
        fts = [_ft(m, "tip", "site", 2.0, 0.5, 0.1),
               {"frame_type": "body", "frame_id": m.name2id("body", "slider"), "cost": [1.0, 0.0, 0.3, 0, 0, 0],
                "gain": 1.0, "lm_damping": 0.0}]
        prob = nat.NativeProblem(nmodel, frame_tasks=fts, posture_tasks=[{"cost": d["posture_cost"]}],
                                 configuration_limits=[_cfg_limit(m)],
                                 velocity_limits=[{"indices": d["vel_indices"], "limit": d["vel_limit"]}],
                                 max_batch=max_batch)
        rows = list(range(0, 6)) + list(range(6 + nv, 12 + nv)) + list(range(6, 6 + nv))
        return prob, (d["frame_targets"], d["posture_target"][None, :], None), np.array(rows), 1e-2, 1e-4
    if name == "balllimit":
        prob = nat.NativeProblem(nmodel, frame_tasks=[_ft(m, "tip", "site", 2.0, 0.5, 0.1)], posture_tasks=[{"cost": 0.1}],
                                 configuration_limits=[_cfg_limit(m, 0.9)], max_batch=max_batch)
        return prob, (d["frame_targets"], d["posture_target"][None, :], None), None, 1e-2, 1e-4
    raise KeyError(name)


ROBOT_OF.update({"g1_ext": "g1", "ur5e_coll": "ur5e", "ballslide": "ballslide", "balllimit": "balllimit"})
