"""Every example robot of the reference (examples/*/scene*.xml: arms, hands, quadrupeds, humanoids, mobile
manipulators with slide joints and free bases) solved on the device and held against the plain-C oracle on every
instance and the numpy oracle on a sample — SURVEY §8(f)-4 "other example robots".  Models: tests/golden/models/all
(compiled from the reference's MJCF by tests/golden/make_models.py)."""

import glob
import os

import numpy as np
import pytest

import native_configs as nc
import oracle_configs as oc
from mink_amd import workloads
from mink_amd.flatmodel import FlatModel
from oracle import cport
from oracle import ik as oik

pytestmark = pytest.mark.gpu
MODELS = sorted(glob.glob(os.path.join(oc.GOLDEN, "models", "all", "*.json")))


def test_every_reference_scene_is_covered():
    assert len(MODELS) == 18


@pytest.mark.parametrize("path", MODELS, ids=[os.path.basename(p)[:-5] for p in MODELS])
def test_solve_matches_oracle(path):
    from mink_amd import _native as nat
    m = FlatModel.load(path)
    nm = nat.NativeModel(m)
    rng = np.random.default_rng(abs(hash(os.path.basename(path))) % 2**31)
    B = 32
    # frames: up to two named sites (else the two deepest bodies), one with orientation cost
    sites = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0][-2:]
    if sites:
        frames = [("site", i) for i in sites]
    else:
        frames = [("body", int(b)) for b in np.argsort(m.body_depth)[-2:]]
    fts, specs = [], []
    for k, (ft, fid) in enumerate(frames):
        cost = [1.0, 1.0, 1.0] + ([0.5, 0.5, 0.5] if k == 0 else [0.0, 0.0, 0.0])
        fts.append({"frame_type": ft, "frame_id": fid, "cost": cost, "gain": 1.0, "lm_damping": 1.0 if k == 0 else 0.0})
    vidx = [int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] in (2, 3)]
    vlim = np.where([m.jnt_type[m.dof_jntid[d]] == 2 for d in vidx], 0.5, np.pi)
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}],
                             configuration_limits=[nc._cfg_limit(m)],
                             velocity_limits=[{"indices": vidx, "limit": vlim}], max_batch=B)
    q, tg = workloads.make_batch(m, nm, prob, rng, B, base_q=m.qpos0)
    # (make_batch parks a free base at z = 0.75 with a random attitude; unlimited hinges get ±π)
    dt, damping = 1e-2, 1e-3
    v, st = prob.solve(q, tg, m.qpos0[None, :], None, dt, damping)
    assert (st & ~1 == 0).all(), st

    def tasks_for(i):
        ts = [oik.FrameTaskSpec(fid, ft, np.array(f["cost"]), tg[i, k], 1.0, f["lm_damping"])
              for k, ((ft, fid), f) in enumerate(zip(frames, fts))]
        return ts + [oik.PostureTaskSpec(np.full(m.nv, 1e-2), m.qpos0)]

    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.array(vidx), vlim)]
    cp = cport.CProblem(m, tasks_for(0), limits)
    v_c, st_c = cp.solve_batch(q, tg, m.qpos0[None, :], dt, damping)
    assert (st_c == 0).all()
    scale = np.maximum(1.0, np.abs(v_c).max(axis=1, keepdims=True))
    err = (np.abs(v - v_c) / scale).max()
    worst = 0.0
    for i in (0, B // 2, B - 1):
        v_ref = oik.solve_ik(m, q[i], tasks_for(i), dt, damping, limits)
        worst = max(worst, np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
    print("%-40s nv %2d  %-28s max rel err: C oracle %.1e, numpy oracle %.1e" %
          (os.path.basename(path)[:-5], m.nv, prob.last_kernel(), err, worst))
    assert err < 1e-8 and worst < 1e-8


def _solve_with_collisions(m, frame_sites, pairs_fn, q0, B, seed, col_kw, sigma=0.1, n_check=4):
    """Shared body of the two mesh-dependent set-ups below: frame tasks on sites + posture + ConfigurationLimit +
    CollisionAvoidanceLimit, device against the numpy oracle on a few instances (the oracle walks every pair in Python)."""
    import mink_amd as mink
    rng = np.random.default_rng(seed)
    q = q0 + rng.normal(scale=sigma, size=(B, m.nq))
    q = np.clip(q, np.where(np.isfinite(_qlo(m)), _qlo(m) + 1e-3, -np.inf), np.where(np.isfinite(_qhi(m)), _qhi(m) - 1e-3, np.inf))
    cfg = mink.Configuration(m, q)
    col = mink.CollisionAvoidanceLimit(m, pairs_fn(mink, m), **col_kw)
    tg_cfg = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.1, size=(B, m.nv)), 1.0))
    tasks = []
    for s in frame_sites:
        t = mink.FrameTask(s, "site", position_cost=1.0, orientation_cost=0.5, lm_damping=1.0)
        t.set_target(tg_cfg.get_transform_frame_to_world(s, "site"))
        tasks.append(t)
    post = mink.PostureTask(m, cost=1e-2); post.set_target(q0)
    dt, damping = 2e-2, 1e-4
    v, st = mink.solve_ik(cfg, tasks + [post], dt, "mi355x", damping, limits=[mink.ConfigurationLimit(m), col], return_status=True)
    G, h = col.compute_qp_inequalities(cfg, dt)
    spec = oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, **{k: v_ for k, v_ in col_kw.items()})
    worst, binding, active = 0.0, 0, int(np.isfinite(h).sum())
    print("detected contacts per instance: max %d (rows available: %d)" % (np.isfinite(h).sum(axis=1).max(), 64 - m.nv))
    for i in np.argsort(-np.isfinite(h).sum(axis=1))[:n_check]:       # the instances with the most contacts
        ots = [oik.FrameTaskSpec(m.name2id("site", s), "site", np.array([1, 1, 1, .5, .5, .5]), t.transform_target_to_world.wxyz_xyz[i], 1.0, 1.0)
               for s, t in zip(frame_sites, tasks)] + [oik.PostureTaskSpec(np.full(m.nv, 1e-2), q0)]
        G_ref, h_ref = oik.limit_inequalities(oik.Configuration(m, q[i]), spec, dt)
        fin = np.isfinite(h_ref)
        assert (np.isfinite(h[i]) == fin).all()
        np.testing.assert_allclose(h[i][fin], h_ref[fin], rtol=0, atol=1e-9 * max(1.0, np.abs(h_ref[fin]).max(initial=0)))
        np.testing.assert_allclose(G[i][fin], G_ref[fin], atol=2e-5)
        v_ref = oik.solve_ik(m, q[i], ots, dt, damping, [oik.ConfigurationLimitSpec(), spec])
        worst = max(worst, np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
        binding += int((np.abs(G_ref[fin] @ (v_ref * dt) - h_ref[fin]) < 1e-8).sum())
    return col, st, worst, binding, active


def _qlo(m):
    lo = np.full(m.nq, -np.inf)
    for j in range(m.njnt):
        if m.jnt_limited[j] and m.jnt_type[j] in (2, 3):
            lo[m.jnt_qposadr[j]] = m.jnt_range[j][0]
    return lo


def _qhi(m):
    hi = np.full(m.nq, np.inf)
    for j in range(m.njnt):
        if m.jnt_limited[j] and m.jnt_type[j] in (2, 3):
            hi[m.jnt_qposadr[j]] = m.jnt_range[j][1]
    return hi


def test_aloha_collision_setup_of_the_reference():
    """examples/arm_aloha.py:95-109: wrist subtree against wrist subtree, both arms against the metal frame and the table —
    1 104 geom pairs, every arm / frame collision geom a capsule FITTED to its mesh (aloha.xml:86-87, scene.xml:44-46: sizes
    and frames from the assets' inertia boxes, mink_amd/meshes.py).  More pairs than the 48 rows a 16-dof tableau leaves: the
    tightest are rows, the rest is checked at the solution."""
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", "aloha__scene.json"))
    assert m.nv == 16 and (m.geom_valid == 2).sum() >= 40

    def pairs(mink, m):
        lw = mink.get_subtree_geom_ids(m, m.body("left/wrist_link").id)
        rw = mink.get_subtree_geom_ids(m, m.body("right/wrist_link").id)
        lg = mink.get_subtree_geom_ids(m, m.body("left/upper_arm_link").id)
        rg = mink.get_subtree_geom_ids(m, m.body("right/upper_arm_link").id)
        fg = mink.get_body_geom_ids(m, m.body("metal_frame").id)
        return [(lw, rw), (lg + rg, fg + ["table"])]

    q0 = m.key_qpos[m.name2id("key", "neutral_pose")] if m.name2id("key", "neutral_pose") >= 0 else m.qpos0
    col, st, worst, binding, active = _solve_with_collisions(
        m, ["left/gripper", "right/gripper"], pairs, q0, 48, 5,
        dict(minimum_distance_from_collisions=0.05, collision_detection_distance=0.1), sigma=0.5, n_check=6)
    assert len(col.geom_id_pairs) == 1104
    print("aloha: %d pairs, %d detected contacts over the batch, %d binding rows in the checked instances, max rel err %.2e, "
          "status bits %s" % (len(col.geom_id_pairs), active, binding, worst, sorted(set(st.tolist()))))
    assert (st & ~1 == 0).all() and worst < 1e-7 and active > 0


def test_shadow_hand_with_mesh_fitted_fingertips_and_the_forearm_mesh():
    """The `*_3` fingertip capsules of the Shadow hand are fitted to f_distal_pst / th_distal_pst (left_hand.xml:149,175,
    201,232,263) and the forearm's collision geom is a MESH (left_hand.xml:101): fingertips against each other, against the
    other fingers' middle phalanges, and against the forearm hull (general convex routine with the hull's vertices)."""
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", "shadow_hand__scene_left.json"))
    fore = [g for g in range(m.ngeom) if m.geom_type[g] == 7 and m.geom_dataid[g] >= 0]
    assert len(fore) == 1

    def pairs(mink, m):
        tips = [f"{f}_3" for f in oc.SHADOW_FINGERS]
        mids = [f"{f}_2" for f in oc.SHADOW_FINGERS]
        return [(tips, tips), (tips, mids), (tips, fore)]

    grasp = m.key_qpos[m.name2id("key", "grasp hard")]
    col, st, worst, binding, active = _solve_with_collisions(
        m, list(oc.SHADOW_FINGERS), pairs, grasp, 32, 9,
        dict(minimum_distance_from_collisions=0.004, collision_detection_distance=0.06), sigma=0.15)
    types = {(int(m.geom_type[a]), int(m.geom_type[b])) for a, b in col.geom_id_pairs}
    assert (3, 7) in types or (7, 3) in types
    print("shadow *_3 + forearm mesh: %d pairs, %d detected contacts, %d binding rows, max rel err %.2e" %
          (len(col.geom_id_pairs), active, binding, worst))
    assert (st & ~1 == 0).all() and worst < 2e-5 and active > 0


def test_com_task_on_a_hand_whose_masses_come_from_its_meshes():
    """The Allegro hand's MJCF has no <inertial> elements: every body's mass is density x the volume of its visual meshes
    (`<geom density="800"/>`, wonik_allegro/left_hand.xml:10).  Since round 4 the MJCF reader derives them (it used to flag the
    bodies and ComTask refused the model); ComTask + fingertip tasks against the C oracle on the same FlatModel."""
    import mink_amd as mink
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", "wonik_allegro__scene_left.json"))
    assert (np.asarray(m.body_mass_valid) == 1).all() and 0.4 < m.body_subtreemass[1] < 0.8          # (the real hand: ≈ 0.6 kg)
    B = 64
    rng = np.random.default_rng(4)
    q = np.tile(m.qpos0, (B, 1)) + rng.uniform(0.05, 0.4, size=(B, m.nq))
    cfg = mink.Configuration(m, q)
    tips = [n for n in m.site_names if n][-4:]
    tgt = mink.Configuration(m, q + rng.normal(scale=0.1, size=q.shape))
    tasks = []
    for s in tips:
        t = mink.FrameTask(s, "site", position_cost=1.0, orientation_cost=0.0, lm_damping=1.0)
        t.set_target(tgt.get_transform_frame_to_world(s, "site"))
        tasks.append(t)
    com = mink.ComTask(cost=5.0)
    com_target = np.asarray(tgt.subtree_com())
    com.set_target(com_target)
    post = mink.PostureTask(m, cost=1e-2); post.set_target(m.qpos0)
    lims = [mink.ConfigurationLimit(m)]
    v = mink.solve_ik(cfg, tasks + [com, post], 1e-2, "quadprog", 1e-4, limits=lims)
    ftg = np.stack([t.transform_target_to_world.wxyz_xyz for t in tasks], axis=1)
    specs = [oik.FrameTaskSpec(m.name2id("site", s), "site", np.array([1.0, 1, 1, 0, 0, 0]), ftg[0, k], 1.0, 1.0) for k, s in enumerate(tips)]
    specs += [oik.ComTaskSpec(np.full(3, 5.0), None), oik.PostureTaskSpec(np.full(m.nv, 1e-2), m.qpos0)]
    v_c, st_c = cport.CProblem(m, specs, [oik.ConfigurationLimitSpec()]).solve_batch(q, ftg, m.qpos0[None, :], 1e-2, 1e-4,
                                                                                     com_target=np.asarray(com_target).reshape(B, 1, 3))
    assert (st_c == 0).all()
    err = (np.abs(v - v_c) / np.maximum(1.0, np.abs(v_c).max(axis=1, keepdims=True))).max()
    print("Allegro + ComTask (mesh-derived masses): max rel err vs C oracle %.1e" % err)
    assert err < 1e-8
