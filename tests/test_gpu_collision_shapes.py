"""CollisionAvoidanceLimit on box / cylinder pairs: device rows vs the numpy oracle.

examples/arm_ur5e.py:30-37 avoids collisions between the `wrist_3_link` capsule and the `floor` plane / `wall`
box; a synthetic scene covers every pair type the device routine knows."""

import copy
import os

import numpy as np
import pytest

import mink_amd as mink
import oracle_configs as oc
from oracle import ik as oik

pytestmark = pytest.mark.gpu
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCENE = """<mujoco><compiler angle="radian"/><worldbody>
  <geom name="floor" type="plane" size="1 1 .1" pos="0 0 -0.35" quat="0.99 0.05 -0.08 0"/>
  <geom name="crate" type="box" size=".12 .2 .15" pos="0.45 0.1 0.1" quat="0.9 0.1 0.3 0.2"/>
  <geom name="drum" type="cylinder" size=".1 .15" pos="-0.1 0.45 0.0" quat="0.8 0.5 0.1 0.2"/>
  <geom name="post" type="capsule" size=".03 .25" pos="0.1 -0.45 0.1" quat="0.9 0.3 0.2 0"/>
  <geom name="orb" type="sphere" size=".06" pos="-0.4 -0.2 0.2"/>
  <body name="l1" pos="0 0 0.1"><joint name="j1" type="hinge" axis="0 0 1" range="-3 3"/>
    <geom name="l1_cap" type="capsule" size=".03 .1" pos="0.1 0 0" quat="1 0 1 0"/>
    <body name="l2" pos="0.2 0 0"><joint name="j2" type="hinge" axis="0 1 0" range="-2 2"/>
      <geom name="l2_box" type="box" size=".08 .03 .04" pos="0.1 0 0"/>
      <body name="l3" pos="0.2 0 0"><joint name="j3" type="hinge" axis="0 1 0" range="-2 2"/>
        <joint name="j3s" type="slide" axis="1 0 0" range="-0.1 0.1"/>
        <geom name="l3_cap" type="capsule" size=".025 .08" pos="0.08 0 0" quat="1 0 1 0"/>
        <body name="l4" pos="0.16 0 0"><joint name="j4" type="ball"/>
          <geom name="l4_ball" type="sphere" size=".04" pos="0.05 0 0"/>
          <geom name="l4_can" type="cylinder" size=".03 .05" pos="0.12 0 0" quat="1 0 1 0"/>
          <site name="tip" pos="0.18 0 0"/>
        </body>
      </body>
    </body>
  </body>
</worldbody></mujoco>"""

PAIRS = [
    (["l1_cap", "l3_cap"], ["crate", "floor"]),          # capsule–box, plane–capsule
    (["l4_ball"], ["crate", "drum", "floor", "post"]),   # sphere–box, sphere–cylinder, plane–sphere, sphere–capsule
    (["l2_box"], ["floor", "orb", "post", "crate"]),     # plane–box, sphere–box, capsule–box, box–box (box on the robot)
    (["l4_can"], ["floor", "orb", "post"]),              # plane–cylinder, sphere–cylinder, capsule–cylinder (cylinder on the robot)
    (["l1_cap", "l3_cap"], ["drum"]),                    # capsule–cylinder (capsule on the robot)
]


def _rand_q(m, rng, n):
    q = np.tile(m.qpos0, (n, 1))
    q[:, 0] = rng.uniform(-3, 3, n)
    q[:, 1] = rng.uniform(-1.5, 1.5, n)
    q[:, 2] = rng.uniform(-1.5, 1.5, n)
    q[:, 3] = rng.uniform(-0.1, 0.1, n)
    quat = rng.normal(size=(n, 4))
    q[:, 4:8] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    return q


def test_every_box_and_cylinder_pair_type_against_the_oracle():
    m = mink.loads_mjcf(SCENE)
    rng = np.random.default_rng(3)
    B = 256
    q = _rand_q(m, rng, B)
    cfg = mink.Configuration(m, q)
    col = mink.CollisionAvoidanceLimit(m, PAIRS, collision_detection_distance=0.25, minimum_distance_from_collisions=0.01)
    assert len(col.geom_id_pairs) == 14
    dt = 0.1               # long step: h = gain·(d − d_min)/dt is small enough for the rows to bind
    G, h = col.compute_qp_inequalities(cfg, dt)
    assert G.shape == (B, 14, m.nv)
    spec = oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, collision_detection_distance=0.25,
                                           minimum_distance_from_collisions=0.01)
    active = np.zeros(14, dtype=int)
    for i in range(B):
        o = oik.Configuration(m, q[i])
        G_ref, h_ref = oik.limit_inequalities(o, spec, dt)
        fin = np.isfinite(h_ref)
        assert (np.isfinite(h[i]) == fin).all(), i
        active += fin
        np.testing.assert_allclose(h[i][fin], h_ref[fin], rtol=0, atol=1e-9 * max(1.0, np.abs(h_ref[fin]).max(initial=0.0)))
        np.testing.assert_allclose(G[i], G_ref, atol=1e-11)
    print("active instances per pair:", dict(zip([tuple(p) for p in col.geom_id_pairs], active)))
    assert (active > 0).all(), active                       # every pair type was exercised inside the cut-off

    # and the solve: tip task + posture + limits + these half-spaces — on the instances that start outside
    # minimum_distance_from_collisions for every pair (random poses put some geoms deep inside each other, where the
    # h = 0 rows of opposing contacts are inconsistent for the reference too)
    # ... drawn from a larger sample so that many start within a few mm of d_min: those rows bind
    q = _rand_q(m, rng, 8192)
    G, h = col.compute_qp_inequalities(mink.Configuration(m, q), dt)
    hmin = np.where(np.isfinite(h), h, np.inf).min(axis=1)
    ok = np.flatnonzero(hmin > 0)
    ok = ok[np.argsort(hmin[ok])][:192]
    q, G, h = q[ok], G[ok], h[ok]
    cfg = mink.Configuration(m, q)
    tgt_cfg = mink.Configuration(m, _rand_q(m, rng, len(ok)))
    ft = mink.FrameTask("tip", "site", position_cost=1.0, orientation_cost=0.2, lm_damping=0.0)
    ft.set_target(tgt_cfg.get_transform_frame_to_world("tip", "site"))
    post = mink.PostureTask(m, cost=1e-2)
    post.set_target(m.qpos0)
    lims = [mink.ConfigurationLimit(m), col]
    v = mink.solve_ik(cfg, [ft, post], dt, "mi355x", 1e-3, limits=lims)
    dq = v * dt
    fin = np.isfinite(h)
    Gx = np.einsum("bpj,bj->bp", G, dq)
    assert (Gx[fin] <= h[fin] + 1e-9).all()                 # every half-space respected
    binding = (np.abs(Gx - h) < 1e-9) & fin
    print("instances solved: %d, binding half-spaces: %d in %d instances" % (len(ok), binding.sum(), binding.any(axis=1).sum()))
    assert binding.any(axis=1).sum() >= 16
    worst = 0.0
    for i in np.concatenate([np.flatnonzero(binding.any(axis=1))[:32], np.arange(0, len(ok), 8)]):
        ts = [oik.FrameTaskSpec(m.name2id("site", "tip"), "site", ft.cost, ft.transform_target_to_world.wxyz_xyz[i], 1.0, 0.0),
              oik.PostureTaskSpec(post.cost, post.target_q, 1.0)]
        v_ref = oik.solve_ik(m, q[i], ts, dt, 1e-3, [oik.ConfigurationLimitSpec(), spec])
        worst = max(worst, np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
    print("solve with mixed-shape half-spaces vs oracle: max rel err %.2e" % worst)
    assert worst < 1e-7


def test_ur5e_example_wrist_against_wall_and_floor():
    """examples/arm_ur5e.py:20-47,74: FrameTask + ConfigurationLimit + CollisionAvoidanceLimit(wrist_3_link vs floor,
    wall) + VelocityLimit(π); the wrist capsule is driven towards the wall box and the floor plane."""
    from mink_amd import workloads
    m = workloads.load_robot("ur5e")
    om = oc.model("ur5e")
    rng = np.random.default_rng(11)
    B = 512
    home = m.key_qpos[m.name2id("key", "home")]
    col = mink.CollisionAvoidanceLimit(m, [(["wrist_3_link"], ["floor", "wall"])], collision_detection_distance=0.3)
    damping = 1e-3
    # half of the batch starts within millimetres of d_min of the wall or the floor (those rows bind), half anywhere
    pool = workloads.sample_q(m, rng, 16384, base_q=home)
    pool[::2] = home + rng.normal(scale=0.4, size=(8192, m.nq))
    _, hp = col.compute_qp_inequalities(mink.Configuration(m, pool), 2e-3)
    hmin = np.where(np.isfinite(hp), hp, np.inf).min(axis=1)
    near = np.flatnonzero(hmin > 0)
    near = near[np.argsort(hmin[near])][: B // 2]
    q = np.concatenate([pool[near], pool[rng.choice(np.flatnonzero(hmin > 0), size=B // 2, replace=False)]])
    cfg = mink.Configuration(m, q)
    vel = mink.VelocityLimit(m, {n: np.pi for n in ("shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3")})
    lims = [mink.ConfigurationLimit(m), col, vel]
    ft = mink.FrameTask("attachment_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
    # targets: inside / behind the wall and below the floor, so that the half-spaces bind
    tg = cfg.get_transform_frame_to_world("attachment_site", "site").wxyz_xyz.copy()
    wall = m.geom_names.index("wall")
    wall_pos = m.body_pos[m.geom_bodyid[wall]] + m.geom_pos[wall]
    tg[::2, 4:] = wall_pos + rng.normal(scale=0.05, size=(B // 2, 3))
    tg[1::2, 6] = -0.1
    ft.set_target(mink.SE3(tg))
    spec = oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, collision_detection_distance=0.3)
    vspec = oik.VelocityLimitSpec(vel.indices, vel.limit)
    # the example's 500 Hz step (rows rarely bind: h ∝ 1/dt) and a 20 Hz step (they do)
    for dt in (2e-3, 5e-2):
        G, h = col.compute_qp_inequalities(cfg, dt)
        fin = np.isfinite(h)
        print("UR5e: active rows  floor %d  wall %d  of %d" % (fin[:, 0].sum(), fin[:, 1].sum(), B))
        assert fin[:, 0].any() and fin[:, 1].any()
        v = mink.solve_ik(cfg, [ft], dt, "mi355x", damping, limits=lims)
        dq = v * dt
        Gx = np.einsum("bpj,bj->bp", G, dq)
        assert (Gx[fin] <= h[fin] + 1e-9).all()
        binding = (np.abs(Gx - h) < 1e-9) & fin
        print("UR5e dt=%g: binding half-spaces in %d instances" % (dt, binding.any(axis=1).sum()))
        assert dt < 1e-2 or binding.any(axis=1).sum() >= 8
        worst = 0.0
        idx = np.concatenate([np.flatnonzero(binding.any(axis=1))[:16], np.flatnonzero(fin.any(axis=1))[:16], np.arange(0, B, 64)])
        for i in idx:
            o = oik.Configuration(om, q[i])
            G_ref, h_ref = oik.limit_inequalities(o, spec, dt)
            assert (np.isfinite(h_ref) == fin[i]).all()
            np.testing.assert_allclose(h[i][fin[i]], h_ref[fin[i]], atol=1e-9 * max(1.0, np.abs(h_ref[fin[i]]).max(initial=0.0)))
            np.testing.assert_allclose(G[i], G_ref, atol=1e-11)
            ts = [oik.FrameTaskSpec(om.name2id("site", "attachment_site"), "site", ft.cost, tg[i], 1.0, 1.0)]
            v_ref = oik.solve_ik(om, q[i], ts, dt, damping, [oik.ConfigurationLimitSpec(), spec, vspec])
            worst = max(worst, np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
        print("UR5e example config, dt=%g, vs oracle: max rel err %.2e" % (dt, worst))
        assert worst < 1e-7


def test_more_contacts_than_tableau_rows_keeps_the_tightest():
    """More detected contacts than the tableau has rows (64 − nv): the reference hands every row to quadprog
    (collision_avoidance_limit.py:187-210).  The device keeps the max_rows tightest and checks the rest at the
    solution: unflagged instances equal the oracle's solve with ALL rows; a dropped row that does not hold is flagged."""
    from mink_amd import _native as nat, workloads
    import native_configs as nc
    m = workloads.load_robot("shadow_left")
    om = oc.model("shadow_left")
    nm = nat.NativeModel(m)
    fingers = oc.SHADOW_FINGERS
    groups = [[f"{f}_1", f"{f}_2"] for f in fingers]
    pairs = [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)] + [(sum(groups, []), ["floor"])]
    col = mink.CollisionAvoidanceLimit(m, pairs, collision_detection_distance=0.5, minimum_distance_from_collisions=0.004)
    assert len(col.geom_id_pairs) == 50 and m.nv == 24        # 50 contacts detected, 40 tableau rows
    fts = [nc._ft(m, f, "site", 1.0, 0.0, 1.0) for f in fingers]
    B = 256
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                             collision_limits=[col._native_desc()[1]], max_batch=B)
    rng = np.random.default_rng(5)
    dt, damping = 0.25, 1e-5   # a long step, so that h = gain·(d − d_min)/dt is small and many rows bind
    # instances that start outside minimum_distance_from_collisions for every pair (opposing h = 0 rows of geoms that
    # start inside each other are inconsistent for the reference too)
    prob0 = nat.NativeProblem(nm, frame_tasks=fts, max_batch=4096)
    qp, tgp = workloads.make_batch(m, nm, prob0, rng, 4096, base_q=m.qpos0, sigma=0.3)
    prob0.close()
    _, hp = col.compute_qp_inequalities(mink.Configuration(m, qp), dt)
    assert np.isfinite(hp).sum(axis=1).max() > 40
    hmin = np.where(np.isfinite(hp), hp, np.inf).min(axis=1)
    ok = np.flatnonzero(hmin > 0)
    ok = ok[np.argsort(hmin[ok])][:B]                         # ... and the closest of those: their rows bind
    assert len(ok) == B
    q, tg = qp[ok], tgp[ok]
    spec = oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, collision_detection_distance=0.5,
                                           minimum_distance_from_collisions=0.004)
    tasks_of = lambda i, tg: [oik.PostureTaskSpec(np.full(m.nv, 1e-2), m.qpos0)] + \
        [oik.FrameTaskSpec(om.name2id("site", f), "site", np.array([1.0, 1, 1, 0, 0, 0]), tg[i, k], lm_damping=1.0)
         for k, f in enumerate(fingers)]
    # (1) targets a short way off: the dropped rows hold at the solution, nothing is flagged
    # (2) targets far off (fingers sweep centimetres in one step): some dropped rows do not hold — flagged, and rightly so
    far = nm.integrate(q, rng.normal(scale=1.0, size=(B, m.nv)), 1.0)
    dummy = np.zeros((B, 5, 7)); dummy[:, :, 0] = 1.0
    prob1 = nat.NativeProblem(nm, frame_tasks=fts, max_batch=B)
    tg_far = prob1.solve(far, dummy, None, None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)[2]["frame_pose"]
    prob1.close()
    # the wavefront kernel by itself (a handle created with MKH_DIAG_NO_WIDE_REDO): what round 2 / 3 returned
    prob_nw = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                                collision_limits=[col._native_desc()[1]], max_batch=B, diag=nat.DIAG_NO_WIDE_REDO)
    for regime, tgr, dt in (("near", tg, dt), ("far", tg_far, 8 * dt)):   # (h ∝ 1/dt: the long step shrinks every slack)
        v_nw, st = prob_nw.solve(q, tgr, m.qpos0[None, :], None, dt, damping)
        assert not prob_nw.last_kernel().endswith("+wide")
        assert ((st & ~(1 | 16 | 32)) == 0).all(), st          # (32: internal — almost dependent active rows, re-solved by the dense iteration)
        flagged = (st & (16 | 32)) != 0
        print("%s targets: instances with a violated dropped row: %d of %d" % (regime, flagged.sum(), B))
        assert flagged.sum() == 0 if regime == "near" else 0 < flagged.sum() < B
        # round 4: the plain call solves the flagged instances again with EVERY row (wide_kernel.h) — nothing stays flagged,
        # the unflagged ones keep the wavefront kernel's answer bit for bit
        v, st_w = prob.solve(q, tgr, m.qpos0[None, :], None, dt, damping)
        assert prob.last_kernel().endswith("+wide") and ((st_w & ~1) == 0).all(), (prob.last_kernel(), np.unique(st_w))
        np.testing.assert_array_equal(v[~flagged], v_nw[~flagged])
        worst, worst_redo, n_over, n_bind = 0.0, 0.0, 0, 0
        for i in list(np.flatnonzero(~flagged)[:16]) + list(np.flatnonzero(flagged)[:6]):
            G_ref, h_ref = oik.limit_inequalities(oik.Configuration(om, q[i]), spec, dt)
            n_over += int(np.isfinite(h_ref).sum() > 40)
            v_ref = oik.solve_ik(om, q[i], tasks_of(i, tgr), dt, damping, [oik.ConfigurationLimitSpec(), spec])
            err = np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max())
            fin = np.isfinite(h_ref)
            n_bind += int((np.abs(G_ref[fin] @ (v_ref * dt) - h_ref[fin]) < 1e-9).sum())
            if flagged[i]:
                assert np.abs(v_nw[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()) > 1e-9    # (not a false alarm: the all-rows answer differs)
                worst_redo = max(worst_redo, err)
            else:
                worst = max(worst, err)
        print("%s targets: overflowing instances checked: %d, binding rows: %d, max rel err vs all-rows oracle: unflagged %.2e, re-solved %.2e"
              % (regime, n_over, n_bind, worst, worst_redo))
        assert n_over >= 16 and n_bind > 0
        assert worst < 1e-7 and worst_redo < 1e-7


CONVEX_SCENE = SCENE.replace('<geom name="orb"', '<geom name="egg" type="ellipsoid" size=".1 .06 .15" pos="0.35 -0.3 0.05" quat="0.7 0.2 0.5 0.1"/>\n  <geom name="orb"') \
    .replace('<geom name="l3_cap"', '<geom name="l3_egg" type="ellipsoid" size=".05 .03 .07" pos="0.02 0.05 0"/>\n        <geom name="l3_cap"')

CONVEX_PAIRS = [
    (["l4_can"], ["crate", "drum", "egg"]),                              # cylinder–box, cylinder–cylinder, ellipsoid–cylinder
    (["l3_egg"], ["floor", "crate", "drum", "post", "orb", "egg"]),      # plane– / box– / cylinder– / capsule– / sphere– / ellipsoid–ellipsoid
]


SEP_TOL = 1e-10     # rows of G of the general convex pairs (round 6: witness points polished onto the exact features; measured 6e-15)


def test_general_convex_pairs_against_the_oracle():
    """The pairs MuJoCo sends to its general convex collider (no native routine): cylinder–box, cylinder–cylinder and the
    ellipsoid against every primitive — device (GJK on support mappings, convex_dev.h) against the numpy statement
    (oracle/gjk.py, pinned against bounded minimisation and by the KKT checker in tests/test_oracle_gjk.py).  h to 1e-9; the rows
    of G to 1e-10 since round 6 (measured 6e-15 separated, 1.5e-14 overlapping): a GJK run leaves its witness points at ~1e-6 (a
    support gap ε leaves an angle √ε in the normal, thin simplices lose the rest) and the expanding polytope's on an ellipsoid at
    ~1e-2; both sides now move them onto the exact features with a certificate (convex_dev.h cvx_polish / oracle/gjk.py polish).
    (MuJoCo's own routine for these pairs, libccd MPR, runs to a tolerance of 1e-6.)"""
    m = mink.loads_mjcf(CONVEX_SCENE)
    rng = np.random.default_rng(4)
    B = 192
    q = _rand_q(m, rng, B)
    cfg = mink.Configuration(m, q)
    col = mink.CollisionAvoidanceLimit(m, CONVEX_PAIRS, collision_detection_distance=0.3, minimum_distance_from_collisions=0.01)
    assert len(col.geom_id_pairs) == 9
    dt = 0.1
    G, h = col.compute_qp_inequalities(cfg, dt)
    spec = oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, collision_detection_distance=0.3,
                                           minimum_distance_from_collisions=0.01)
    active = np.zeros(9, dtype=int)
    apart = np.zeros(9, dtype=int)
    worst_near, n_near = np.zeros(9), np.zeros(9, dtype=int)
    worst_sep = 0.0
    for i in range(B):
        o = oik.Configuration(m, q[i])
        G_ref, h_ref = oik.limit_inequalities(o, spec, dt)
        fin = np.isfinite(h_ref)
        assert (np.isfinite(h[i]) == fin).all(), i
        active += fin
        sep = fin & (h_ref > 0.0)                       # separated by more than d_min: the Euclidean distance, exactly
        apart += sep
        np.testing.assert_allclose(h[i][sep], h_ref[sep], rtol=0, atol=1e-9 * max(1.0, np.abs(h_ref[sep]).max(initial=0.0)))
        worst_sep = max(worst_sep, np.abs(G[i][sep] - G_ref[sep]).max(initial=0.0))
        np.testing.assert_array_equal(h[i][fin & ~sep], h_ref[fin & ~sep])      # closer than d_min (or overlapping): h = relaxation
        # ... and their rows: the direction of an OVERLAPPING pair is the smallest separating translation (expanding polytope,
        # round 4, polished in round 6: an ellipsoid against a curved shape used to stop at the vertex budget with the direction
        # good to ~1e-2, path-dependent)
        near = fin & ~sep
        worst_near = np.maximum(worst_near, np.where(near, np.abs(G[i] - G_ref).max(axis=1), 0.0))
        n_near += near
    print("active / separated instances per pair:", list(zip([tuple(p) for p in col.geom_id_pairs], active, apart)))
    print("rows of separated pairs: max |dG| %.2e" % worst_sep)
    assert worst_sep < SEP_TOL
    assert (apart > 0).all(), apart
    types = [(int(m.geom_type[a]), int(m.geom_type[b])) for a, b in col.geom_id_pairs]
    print("rows of pairs inside d_min / overlapping: count, max |dG| per pair:", list(zip(types, n_near, worst_near)))
    assert n_near.sum() > 0
    for (ta, tb), w in zip(types, worst_near):
        assert w < 1e-10, (ta, tb, w)
    # the solve on instances that start outside d_min for every pair (the lean collision variant with the convex routine)
    q = _rand_q(m, rng, 4096)
    G, h = col.compute_qp_inequalities(mink.Configuration(m, q), dt)
    hmin = np.where(np.isfinite(h), h, np.inf).min(axis=1)
    ok = np.flatnonzero(hmin > 0)
    ok = ok[np.argsort(hmin[ok])][:128]
    q, G, h = q[ok], G[ok], h[ok]
    cfg = mink.Configuration(m, q)
    ft = mink.FrameTask("tip", "site", position_cost=1.0, orientation_cost=0.2, lm_damping=0.0)
    ft.set_target(mink.Configuration(m, _rand_q(m, rng, len(ok))).get_transform_frame_to_world("tip", "site"))
    post = mink.PostureTask(m, cost=1e-2)
    post.set_target(m.qpos0)
    v = mink.solve_ik(cfg, [ft, post], dt, "mi355x", 1e-3, limits=[mink.ConfigurationLimit(m), col])
    last = list(cfg._problems.values())[-1].last_kernel()          # (most recently used descriptor of the cache)
    # (round 5: plain solves evaluate general convex pairs in a kernel in front of the ANALYTIC build — "convex_pre+…_8")
    assert last.startswith("convex_pre+") or last.removesuffix("+wide").endswith("_136"), last
    dq = v * dt
    fin = np.isfinite(h)
    Gx = np.einsum("bpj,bj->bp", G, dq)
    # (G here comes from the parity build's taps, v from the lean build: two compilations of the same routine)
    assert (Gx[fin] <= h[fin] + 1e-9).all()
    binding = (np.abs(Gx - h) < 1e-9) & fin
    print("binding convex half-spaces: %d in %d instances" % (binding.sum(), binding.any(axis=1).sum()))
    assert binding.any(axis=1).sum() >= 8
    worst = 0.0
    for i in np.flatnonzero(binding.any(axis=1))[:24]:
        ts = [oik.FrameTaskSpec(m.name2id("site", "tip"), "site", ft.cost, ft.transform_target_to_world.wxyz_xyz[i], 1.0, 0.0),
              oik.PostureTaskSpec(post.cost, post.target_q, 1.0)]
        v_ref = oik.solve_ik(m, q[i], ts, dt, 1e-3, [oik.ConfigurationLimitSpec(), spec])
        worst = max(worst, np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
    print("solve with general convex half-spaces vs oracle: max rel err %.2e" % worst)
    assert worst < 1e-9


def _with_mesh_geoms(m, rng):
    """CONVEX_SCENE with three geoms turned into MESH geoms (hull vertices in the geom frame, as a compiled model carries
    them): `crate` → the 8 corners of the same box (every answer must be the box's), `l4_can` → a 40-vertex hull of the same
    cylinder's rim points + noise, `egg` → a random 24-point hull.  The GPU box has no asset files: the hulls are made here."""
    from scipy.spatial import ConvexHull
    m = copy.deepcopy(m)
    hulls = []

    def to_mesh(name, pts):
        g = m.name2id("geom", name)
        pts = np.asarray(pts, dtype=np.float32).astype(np.float64)
        pts = pts[np.sort(ConvexHull(pts).vertices)]
        m.geom_type[g] = 7
        m.geom_dataid[g] = len(hulls)
        hulls.append(pts)
        return g

    g = m.name2id("geom", "crate")
    sb = m.geom_size[g].copy()
    to_mesh("crate", [[sx * sb[0], sy * sb[1], sz * sb[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    g = m.name2id("geom", "l4_can")
    r, hl = m.geom_size[g][:2]
    ang = 2 * np.pi * np.arange(20) / 20
    rim = np.concatenate([np.c_[r * np.cos(ang), r * np.sin(ang), np.full(20, s * hl)] for s in (-1, 1)])
    to_mesh("l4_can", rim * rng.uniform(0.9, 1.0, size=(40, 1)))
    to_mesh("egg", rng.normal(size=(24, 3)) * np.array([0.1, 0.06, 0.15]))
    num = np.array([len(h) for h in hulls], dtype=np.int32)
    m.mesh_vertnum, m.mesh_vertadr, m.mesh_vert = num, (np.cumsum(num) - num).astype(np.int32), np.concatenate(hulls)
    return m.finalize()


def test_mesh_geoms_against_the_oracle():
    """(f)-3, meshes: a mesh geom takes part in CollisionAvoidanceLimit through its convex hull, as in mj_geomDistance
    (/root/reference/mink/limits/collision_avoidance_limit.py:214-229) — plane–mesh by the hull's lowest vertex, mesh against
    primitives and meshes through the general convex routine with the hull's vertices as the support mapping.  Device
    against the numpy statement (oracle/gjk.py, pinned against bounded minimisation over hull points in
    tests/test_oracle_gjk.py); the box-corner mesh against the analytic box routines of the primitive model."""
    base = mink.loads_mjcf(CONVEX_SCENE)
    rng = np.random.default_rng(14)
    m = _with_mesh_geoms(base, rng)
    pairs = [(["l4_can"], ["crate", "drum", "egg", "floor"]),            # mesh–mesh, mesh–cylinder, mesh–mesh, plane–mesh
             (["l3_egg"], ["crate", "egg"]), (["l3_cap"], ["crate", "egg"])]   # ellipsoid–mesh, capsule–mesh
    col = mink.CollisionAvoidanceLimit(m, pairs, collision_detection_distance=0.3, minimum_distance_from_collisions=0.01)
    npair = len(col.geom_id_pairs)
    assert npair == 8
    B = 160
    q = _rand_q(m, rng, B)
    dt = 0.1
    G, h = col.compute_qp_inequalities(mink.Configuration(m, q), dt)
    spec = oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, collision_detection_distance=0.3, minimum_distance_from_collisions=0.01)
    apart = np.zeros(npair, dtype=int)
    for i in range(B):
        G_ref, h_ref = oik.limit_inequalities(oik.Configuration(m, q[i]), spec, dt)
        fin = np.isfinite(h_ref)
        assert (np.isfinite(h[i]) == fin).all(), i
        sep = fin & (h_ref > 0.0)
        apart += sep
        np.testing.assert_allclose(h[i][sep], h_ref[sep], rtol=0, atol=1e-9 * max(1.0, np.abs(h_ref[sep]).max(initial=0.0)))
        np.testing.assert_allclose(G[i][sep], G_ref[sep], atol=2e-5)
    print("separated instances per mesh pair:", list(zip([tuple(p) for p in col.geom_id_pairs], apart)))
    assert (apart > 0).all(), apart
    # the crate as a mesh of its corners = the crate as a box, pair by pair (capsule–box is analytic in the primitive model)
    col_m = mink.CollisionAvoidanceLimit(m, [(["l3_cap"], ["crate"])], collision_detection_distance=0.3)
    col_b = mink.CollisionAvoidanceLimit(base, [(["l3_cap"], ["crate"])], collision_detection_distance=0.3)
    Gm, hm = col_m.compute_qp_inequalities(mink.Configuration(m, q), dt)
    Gb, hb = col_b.compute_qp_inequalities(mink.Configuration(base, q), dt)
    fin = np.isfinite(hb[:, 0]) & (hb[:, 0] > 0)
    assert fin.sum() > 10 and (np.isfinite(hm[:, 0]) == np.isfinite(hb[:, 0])).all()
    np.testing.assert_allclose(hm[fin], hb[fin], rtol=0, atol=1e-6)     # (hull vertices are float32, as in a compiled model: 1e-8 in the sizes)
    # a solve with mesh half-spaces binding
    hmin = np.where(np.isfinite(h), h, np.inf).min(axis=1)
    ok = np.flatnonzero(hmin > 0)
    ok = ok[np.argsort(hmin[ok])][:64]
    cfg = mink.Configuration(m, q[ok])
    ft = mink.FrameTask("tip", "site", position_cost=1.0, orientation_cost=0.2)
    ft.set_target(mink.Configuration(m, _rand_q(m, rng, len(ok))).get_transform_frame_to_world("tip", "site"))
    post = mink.PostureTask(m, cost=1e-2); post.set_target(m.qpos0)
    v = mink.solve_ik(cfg, [ft, post], dt, "mi355x", 1e-3, limits=[mink.ConfigurationLimit(m), col])
    lk = list(cfg._problems.values())[-1].last_kernel()
    assert lk.startswith("convex_pre+") or lk.removesuffix("+wide").endswith("_136"), lk
    worst, binding = 0.0, 0
    for j, i in enumerate(ok[:24]):
        tasks = [oik.FrameTaskSpec(m.name2id("site", "tip"), "site", np.array([1.0, 1.0, 1.0, 0.2, 0.2, 0.2]), ft.transform_target_to_world.wxyz_xyz[j]),
                 oik.PostureTaskSpec(np.full(m.nv, 1e-2), m.qpos0)]
        v_ref = oik.solve_ik(m, q[i], tasks, dt, 1e-3, [oik.ConfigurationLimitSpec(), spec])
        worst = max(worst, np.abs(v[j] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
        fin = np.isfinite(h[i])
        binding += int((np.abs(G[i][fin] @ (v_ref * dt) - h[i][fin]) < 1e-7).sum())
    print("solve with mesh half-spaces: max rel err %.2e, binding rows %d" % (worst, binding))
    assert worst < 2e-5 and binding > 0
    # a model whose mesh geom carries no hull is refused at problem creation, with the reason
    bad = copy.deepcopy(m); bad.geom_dataid[bad.name2id("geom", "egg")] = -1; bad.finalize()
    with pytest.raises(Exception, match="hull"):
        mink.solve_ik(mink.Configuration(bad, q[:2]), [post], dt, "mi355x", 1e-3,
                      limits=[mink.CollisionAvoidanceLimit(bad, [(["l3_cap"], ["egg"])])])


TIE_SCENE = """<mujoco><worldbody>
  <geom name="slab" type="box" size=".3 .2 .05"/>
  <geom name="rail" type="capsule" size=".05 .2" pos=".1 0 0" quat="1 0 1 0"/>
  <body name="carrier"><joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 1 0"/><joint type="slide" axis="0 0 1"/>
    <inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>
    <body name="mover"><joint type="ball"/>
      <geom name="brick" type="box" size=".1 .1 .05"/>
      <geom name="rod" type="capsule" size=".04 .15" quat="1 0 1 0"/>
    </body>
  </body>
</worldbody></mujoco>"""


def test_tie_rule_is_pinned():
    """Where the closest pair of points is not unique, mujoco 3.1.6 keeps the first of several equal contacts and this
    library follows a rule of its own (include/minkhip.h at MkhCollisionLimitDesc).  The rule is pinned here so that it
    cannot drift: the witness point of three tie configurations, stated as numbers, on the oracle — and the device's rows
    of G against the oracle's on the same configurations (the witness point is the lever arm of the angular dofs)."""
    from oracle import mjmath as mj
    I = np.eye(3).reshape(-1)
    Rx = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], float).reshape(-1)          # capsule axis (local z) along world x
    # boxes face to face, partial overlap: the first closest (vertex, face) pair — the brick's (−, −) corner
    (d, pos, n), = mj._box_box(np.zeros(3), I, np.array([.3, .2, .05]), np.array([.25, .05, .2]), I, np.array([.1, .1, .05]), 1.0)
    assert abs(d - 0.1) < 1e-15 and np.allclose(pos, [0.15, -0.05, 0.1], atol=1e-15) and np.allclose(n, [0, 0, 1])
    # capsule parallel to a box face: midpoint of the stretch of the axis over the face ([.05, .35] ∩ [−.3, .3] → .175)
    (d, pos, n), = mj._capsule_box(np.array([.2, 0, .3]), Rx, np.array([.04, .15, 0]), np.zeros(3), I, np.array([.3, .2, .05]), 1.0)
    assert abs(d - 0.21) < 1e-15 and np.allclose(pos, [0.175, 0.0, 0.155], atol=1e-15)
    # parallel capsules: the two ends of the overlapping stretch ([−.15, .15] ∩ [−.1, .3]), the first kept
    cons = mj._capsule_capsule(np.array([0, 0, .3]), Rx, np.array([.04, .15, 0]), np.array([.1, 0, 0]), Rx, np.array([.05, .2, 0]), 1.0)
    assert len(cons) == 2 and np.allclose(cons[0][1], [0.15, 0, 0.155]) and np.allclose(cons[1][1], [-0.1, 0, 0.155])
    assert min(cons, key=lambda c: c[0]) is cons[0]
    # the device on the same three configurations (three slides + a ball joint: G's angular columns carry the witness point;
    # a free body directly under the world would be filtered out as parent and child, collision_avoidance_limit.py:85-106)
    m = mink.loads_mjcf(TIE_SCENE)
    q = np.array([[.25, .05, .2, 1, 0, 0, 0], [.2, 0, .3, 1, 0, 0, 0], [0, 0, .3, 1, 0, 0, 0]], dtype=np.float64)
    pairs = [(["brick"], ["slab"]), (["rod"], ["slab"]), (["rod"], ["rail"])]
    for i, pr in enumerate(pairs):
        col = mink.CollisionAvoidanceLimit(m, [pr], collision_detection_distance=0.5)
        G, h = col.compute_qp_inequalities(mink.Configuration(m, q[i:i + 1]), 0.1)
        spec = oik.CollisionAvoidanceLimitSpec(col.geom_id_pairs, collision_detection_distance=0.5)
        G_ref, h_ref = oik.limit_inequalities(oik.Configuration(m, q[i]), spec, 0.1)
        assert np.isfinite(h_ref).all() and np.abs(G_ref[0, 3:]).max() > 1e-3      # the lever arm is visible in the row
        np.testing.assert_allclose(h[0], h_ref, rtol=0, atol=1e-12)
        np.testing.assert_allclose(G[0], G_ref, rtol=0, atol=1e-12)


def test_tight_rows_then_redo_equals_full_rows():
    """Capsule-only collision sets whose pairs outnumber 48 − nv rows (the Shadow hand: 40 — here 50 — pairs, 24 rows) run a TIGHT-rows
    launch on the 48-row build first — the tightest contacts get the rows, dropped ones are checked at the solution — and the
    full-row build then re-solves the instances that launch flagged (SolveArgs::redo_mask).  Status and v equal the full-row
    solve's (MKH_FLAG_FULL_ROWS), in a regime where the tight launch alone leaves instances flagged (counted in a subprocess
    on a handle created with MKH_DIAG_NO_TIGHT_REDO) and in the benchmark's own."""
    import subprocess
    import sys
    from mink_amd import _native as nat, workloads
    import native_configs as nc
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r]
import native_configs as nc
import mink_amd as mink
from mink_amd import _native as nat, workloads
import oracle_configs as oc
m = workloads.load_robot("shadow_left"); nm = nat.NativeModel(m)
fingers = oc.SHADOW_FINGERS
groups = [[f"{f}_1", f"{f}_2"] for f in fingers]
pairs = [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)] + [(sum(groups, []), ["floor"])]
col = mink.CollisionAvoidanceLimit(m, pairs, collision_detection_distance=0.5, minimum_distance_from_collisions=0.004)
fts = [nc._ft(m, f, "site", 1.0, 0.0, 1.0) for f in fingers]
B = 1024
prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                         collision_limits=[col._native_desc()[1]], max_batch=B, diag=nat.DIAG_NO_TIGHT_REDO)
rng = np.random.default_rng(5)
p0 = nat.NativeProblem(nm, frame_tasks=fts, max_batch=8192)
qp, _ = workloads.make_batch(m, nm, p0, rng, 8192, base_q=m.qpos0, sigma=0.3)
_, hp = col.compute_qp_inequalities(mink.Configuration(m, qp), 0.25)
hmin = np.where(np.isfinite(hp), hp, np.inf).min(axis=1)
ok = np.flatnonzero(hmin > 0)
q = qp[ok[np.argsort(hmin[ok])][:B]]                  # the closest starts that are not inside d_min: their rows bind
far = nm.integrate(q, rng.normal(scale=1.0, size=(B, m.nv)), 1.0)
dummy = np.zeros((B, 5, 7)); dummy[:, :, 0] = 1.0
p1 = nat.NativeProblem(nm, frame_tasks=fts, max_batch=B)
tgf = p1.solve(far, dummy, None, None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)[2]["frame_pose"]
np.savez(sys.argv[1], q=q, tg=tgf)
v, st = prob.solve(q, tgf, m.qpos0[None, :], None, 2.0, 1e-5)
print("KERNEL", prob.last_kernel(), "FLAGGED", int(((st & 16) != 0).sum()))
""" % (REPO_ROOT, os.path.join(REPO_ROOT, "tests"))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "batch.npz")
        out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("KERNEL")][0].split()
        assert line[1] == "ik_solve_kernel_48_72", line
        flagged_by_tight = int(line[3])
        d = np.load(path)
        q, tgf = d["q"], d["tg"]
    m = workloads.load_robot("shadow_left")
    nm = nat.NativeModel(m)
    fingers = oc.SHADOW_FINGERS
    groups = [[f"{f}_1", f"{f}_2"] for f in fingers]
    pairs = [(groups[i], groups[j]) for i in range(5) for j in range(i + 1, 5)] + [(sum(groups, []), ["floor"])]
    col = mink.CollisionAvoidanceLimit(m, pairs, collision_detection_distance=0.5, minimum_distance_from_collisions=0.004)
    assert len(col.geom_id_pairs) == 50
    fts = [nc._ft(m, f, "site", 1.0, 0.0, 1.0) for f in fingers]
    B = len(q)
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                             collision_limits=[col._native_desc()[1]], max_batch=B)
    # (50 pairs against 40 rows: behind either launch sequence the workgroup-per-problem kernel re-solves what is still flagged;
    #  MKH_DIAG_NO_WIDE_REDO keeps this test on the two wavefront paths it compares)
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                             collision_limits=[col._native_desc()[1]], max_batch=B, diag=nat.DIAG_NO_WIDE_REDO)
    v, st = prob.solve(q, tgf, m.qpos0[None, :], None, 2.0, 1e-5)
    assert prob.last_kernel().removesuffix("+wide") == "ik_solve_kernel_48_72+redo_64", prob.last_kernel()
    vf, stf = prob.solve(q, tgf, m.qpos0[None, :], None, 2.0, 1e-5, full_rows=True)
    assert prob.last_kernel().removesuffix("+wide") == "ik_solve_kernel_64_72", prob.last_kernel()
    print("tight launch alone left %d of %d instances flagged; after the redo launch %d (full rows: %d)" % (
        flagged_by_tight, B, int(((st & 16) != 0).sum()), int(((stf & 16) != 0).sum())))
    assert flagged_by_tight >= 10
    np.testing.assert_array_equal(st, stf)
    # ... and the default handle: the 29 instances the wavefront kernels leave flagged (50 contacts, 40 rows) solved with every row
    prob_w = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                               collision_limits=[col._native_desc()[1]], max_batch=B)
    vw, stw = prob_w.solve(q, tgf, m.qpos0[None, :], None, 2.0, 1e-5)
    assert prob_w.last_kernel() == "ik_solve_kernel_48_72+redo_64+wide", prob_w.last_kernel()
    assert ((stw & 16) == 0).all() and ((stw & 14) == (st & 14)).sum() >= B - 29
    keep = (st & 16) == 0
    np.testing.assert_array_equal(vw[keep], v[keep])
    ok = (st & 14) == 0
    scale = np.maximum(1.0, np.abs(vf[ok]).max(axis=1, keepdims=True))
    assert (np.abs(v[ok] - vf[ok]) / scale).max() < 1e-9
    # the benchmark's regime (BASELINE configs[3]): the tightest 24 contacts always suffice there
    prob2, dt2, damping2 = nc.build("shadow_c4", nm, 512)
    key = m.key_qpos[m.name2id("key", "grasp hard")]
    q2, tg2 = workloads.make_batch(m, nm, prob2, np.random.default_rng(5), 512, base_q=key)
    q2[::2] = 0.5 * (q2[::2] + key)
    v2, st2 = prob2.solve(q2, tg2, key[None, :], None, dt2, damping2)
    assert prob2.last_kernel().removesuffix("+wide") == "ik_solve_kernel_48_72+redo_64"
    v2f, st2f = prob2.solve(q2, tg2, key[None, :], None, dt2, damping2, full_rows=True)
    np.testing.assert_array_equal(st2, st2f)
    assert (st2 & ~1 == 0).all() and (np.abs(v2 - v2f) / np.maximum(1.0, np.abs(v2f).max(axis=1, keepdims=True))).max() < 1e-9


def test_kernel_choice_does_not_depend_on_the_status_pointer():
    """A C caller may pass status_out = NULL.  On a handle with a tight-rows build that call used to fall back to the 64-row
    build (the redo launch reads the status): the same inputs ran different kernels depending on an OUTPUT pointer (round-3
    advisor finding).  Now the status of such a call lives in a buffer of the handle: same kernels, same v."""
    import ctypes as C
    from mink_amd import _native as nat, workloads
    import native_configs as nc
    model = workloads.load_robot("shadow_left")
    nm = nat.NativeModel(model)
    B = 2048
    prob, dt, damping = nc.build("shadow_c4", nm, B)
    base = model.key_qpos[model.name2id("key", "grasp hard")]
    q, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(9), B, base_q=base)
    q[::2] = 0.5 * (q[::2] + base)
    v_ref, st_ref = prob.solve(q, tg, base[None, :], None, dt, damping)
    assert prob.last_kernel().removesuffix("+wide") == "ik_solve_kernel_48_72+redo_64"
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    q_d, tg_d, pt_d = to(q), to(tg), to(base[None, :])
    v_d = torch.full((B, model.nv), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev).cuda_stream
    nat._check(nat.lib().mkh_solve(prob.handle, B, q_d.data_ptr(), tg_d.data_ptr(), pt_d.data_ptr(), None, float(dt), float(damping),
                                   v_d.data_ptr(), None, nat.FLAG_DEVICE_PTRS, stream))      # status_out = NULL, device pointers
    torch.cuda.synchronize()
    assert prob.last_kernel().removesuffix("+wide") == "ik_solve_kernel_48_72+redo_64", prob.last_kernel()
    np.testing.assert_array_equal(v_d.cpu().numpy(), v_ref)
