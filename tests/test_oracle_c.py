"""oracle/c (plain-C restatement) pinned against (a) the fixtures recorded from the real mink Python
(tests/golden/ik_*.npz) and (b) the numpy restatement oracle/*.py.  CPU only."""

import os

import numpy as np
import pytest

import oracle_configs as oc
from oracle import cport, ik, qp_gi


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"ik_{name}.npz"))


@pytest.mark.parametrize("name,cfgfn,extra", [("ur5e_c2", oc.ur5e_c2, None), ("g1_c3", oc.g1_c3, None),
                                              ("g1_full", oc.g1_full, "com_target")])
def test_c_oracle_vs_real_mink_fixtures(golden_dir, name, cfgfn, extra):
    d = _load(golden_dir, name)
    for i in range(len(d["q"])):
        args = [d["frame_targets"][i], d["posture_target"]]
        if extra is not None:
            args.append(d[extra][i])
        m, tasks, limits, dt, damping = cfgfn(*args)
        prob = cport.CProblem(m, tasks, limits)
        v, (H, c) = prob.solve(d["q"][i], dt, damping, return_problem=True)
        scale = max(1.0, np.abs(d["H"][i]).max())
        np.testing.assert_allclose(H, d["H"][i], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(c, d["c"][i], rtol=0, atol=1e-12 * max(1.0, np.abs(d["c"][i]).max()))
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][i]).max()))
        # and against the numpy restatement (same operation order): much tighter
        v_np = ik.solve_ik(m, d["q"][i], tasks, dt, damping, limits)
        np.testing.assert_allclose(v, v_np, rtol=0, atol=1e-11 * max(1.0, np.abs(v_np).max()))


def test_c_oracle_ur5e_c1_default_limits(golden_dir):
    """limits=None ⇒ a fresh ConfigurationLimit (solve_ik.py:28-29); every step of the recorded trajectory."""
    d = _load(golden_dir, "ur5e_c1")
    m, tasks, limits, dt, damping = oc.ur5e_c1([d["frame_target"]], d["posture_target"])
    assert limits is None
    prob = cport.CProblem(m, tasks, limits)
    for k in range(len(d["q"])):
        v = prob.solve(d["q"][k], dt, damping)
        np.testing.assert_allclose(v, d["v"][k], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][k]).max()))


def test_c_qp_vs_numpy_gi():
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(2, 20)); m = int(rng.integers(0, 30))
        A = rng.normal(size=(n + 2, n))
        P = A.T @ A + 0.1 * np.eye(n)
        q = rng.normal(size=n)
        G = rng.normal(size=(m, n)) if m else None
        h = (np.abs(rng.normal(size=m)) + 0.05) if m else None     # x = 0 feasible
        x_np = qp_gi.solve_qp(P, q, G, h)
        x_c = cport.solve_qp(P, q, G, h)
        np.testing.assert_allclose(x_c, x_np, rtol=0, atol=1e-10 * max(1.0, np.abs(x_np).max()))
        assert qp_gi.kkt_residual(P, q, G, h, x_c) < 1e-8 * max(1.0, np.abs(q).max())
    with pytest.raises(qp_gi.NotPositiveDefinite):
        cport.solve_qp(np.diag([1.0, -1.0]), np.zeros(2))
    with pytest.raises(qp_gi.Infeasible):
        cport.solve_qp(np.eye(2), np.zeros(2), np.array([[1.0, 0.0], [-1.0, 0.0]]), np.array([-1.0, -1.0]))


def test_c_batch_equals_single_and_threads(golden_dir):
    d = _load(golden_dir, "g1_c3")
    m, tasks, limits, dt, damping = oc.g1_c3(d["frame_targets"][0], d["posture_target"])
    prob = cport.CProblem(m, tasks, limits)
    v1, st1 = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping, nthreads=1)
    v2, st2 = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping, nthreads=2)
    assert (st1 == 0).all() and (st2 == 0).all()
    np.testing.assert_array_equal(v1, v2)
    np.testing.assert_allclose(v1, d["v"], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"]).max()))


def test_c_oracle_vs_2048_real_mink_instances(golden_dir):
    """The BASELINE-sized real-mink slice (tests/golden/make_golden_big.py) pins the checker itself at that size."""
    d = _load(golden_dir, "g1_c3_big")
    m, tasks, limits, dt, damping = oc.g1_c3(d["frame_targets"][0], d["posture_target"])
    prob = cport.CProblem(m, tasks, limits)
    v, st = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping)
    assert (st == 0).all()
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    err = np.abs(v - d["v"]) / vs
    main = np.ones(len(v), bool); main[7::8] = False
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5, (err[main].max(), err[~main].max())


def test_c_collision_rows_vs_real_mink_fixture(golden_dir):
    """CollisionAvoidanceLimit in the C restatement (plane / sphere / capsule pairs): every row of the Shadow config 4
    fixture recorded from the real mink — h of all 40 pairs on every instance (which pairs are inactive included), G on the
    instances that carry it — and v through the C Goldfarb–Idnani with the contact rows stacked behind the box rows."""
    d = _load(golden_dir, "shadow_c4")
    m, tasks, limits, dt, damping = oc.shadow_c4(d["frame_targets"][0], d["posture_target"])
    prob = cport.CProblem(m, tasks, limits)
    npair = len(limits[1].geom_id_pairs)
    for i in range(len(d["q"])):
        G, h = prob.collision_rows(d["q"][i], dt)
        h_ref = d["h"][i][-npair:]
        fin = np.isfinite(h_ref)
        np.testing.assert_array_equal(np.isfinite(h), fin)
        np.testing.assert_allclose(h[fin], h_ref[fin], rtol=0, atol=1e-12 * max(1.0, np.abs(h_ref[fin]).max()))
        if i < len(d["G"]):
            np.testing.assert_allclose(G, d["G"][i][-npair:], rtol=0, atol=1e-13)
    v, st = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping, nthreads=2)
    assert (st == 0).all()
    np.testing.assert_allclose(v, d["v"], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"]).max()))
    # the same rows from the numpy restatement, on a configuration with the fingers pulled together
    cfg = ik.Configuration(m, 0.5 * (d["q"][3] + m.key_qpos[m.name2id("key", "grasp hard")]))
    G_np, h_np = ik.limit_inequalities(cfg, limits[1], dt)
    G_c, h_c = prob.collision_rows(cfg.q, dt)
    np.testing.assert_array_equal(np.isfinite(h_c), np.isfinite(h_np))
    np.testing.assert_allclose(G_c, G_np, rtol=0, atol=1e-13)
    np.testing.assert_allclose(h_c[np.isfinite(h_c)], h_np[np.isfinite(h_np)], rtol=0, atol=1e-12)
    with pytest.raises(TypeError):          # box against box: outside the restated pair set, refused at construction
        mu, _, _, _, _ = oc.ur5e_c2([np.zeros(7)], np.zeros(6))
        box = [g for g in range(mu.ngeom) if int(mu.geom_type[g]) == 6]
        cport.CProblem(mu, [], [ik.CollisionAvoidanceLimitSpec([(box[0], box[0])])])


def _geom_soup(rng, n_each=3):
    """A model whose bodies are free-floating geoms of every type the C restatement knows: one plane on the world, then
    spheres, capsules, cylinders and boxes on free bodies (random sizes)."""
    from mink_amd import mjcf
    parts, names = ['<geom name="floor" type="plane" size="0 0 0.01"/>'], ["floor"]
    k = 0
    for ty, mk in (("sphere", lambda: "%.3f" % rng.uniform(0.02, 0.1)),
                   ("capsule", lambda: "%.3f %.3f" % (rng.uniform(0.02, 0.06), rng.uniform(0.05, 0.2))),
                   ("cylinder", lambda: "%.3f %.3f" % (rng.uniform(0.03, 0.1), rng.uniform(0.03, 0.2))),
                   ("box", lambda: "%.3f %.3f %.3f" % tuple(rng.uniform(0.03, 0.15, size=3)))):
        for _ in range(n_each):
            nm = "%s%d" % (ty, k); k += 1
            parts.append('<body name="b_%s"><freejoint/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>'
                         '<geom name="%s" type="%s" size="%s"/></body>' % (nm, nm, ty, mk()))
            names.append(nm)
    return mjcf.loads_mjcf("<mujoco><worldbody>" + "".join(parts) + "</worldbody></mujoco>"), names


@pytest.mark.parametrize("seed", range(6))
def test_c_box_and_cylinder_pairs_equal_the_numpy_restatement(seed):
    """Round 5: box against plane / sphere / capsule and cylinder against plane / sphere / capsule in the C restatement
    (what `g1_coll` and `ur5e_coll` need on every instance of their bench batch) — rows G, h of every such pair of a soup of
    free-floating geoms at random poses, separated AND penetrating, against oracle/mjmath.py's routines (which
    tests/test_oracle_collision_shapes.py holds against brute force)."""
    rng = np.random.default_rng(100 + seed)
    m, names = _geom_soup(rng)
    gt = np.asarray(m.geom_type)
    ok = cport._C_PAIR_TYPES
    pairs = [(a, b) for a in range(m.ngeom) for b in range(a + 1, m.ngeom)
             if tuple(sorted((int(gt[a]), int(gt[b])))) in ok]
    assert {tuple(sorted((int(gt[a]), int(gt[b])))) for a, b in pairs} == ok
    spec = ik.CollisionAvoidanceLimitSpec(pairs, collision_detection_distance=0.6, minimum_distance_from_collisions=0.01)
    prob = cport.CProblem(m, [], [spec])
    worst_G = worst_h = 0.0
    n_pen = n_act = 0
    for _ in range(6):
        q = np.array(m.qpos0, dtype=np.float64)
        for j in range(m.njnt):
            a = int(m.jnt_qposadr[j])
            q[a:a + 3] = rng.normal(scale=0.25, size=3) + [0, 0, 0.15]
            w = rng.normal(size=4); q[a + 3:a + 7] = w / np.linalg.norm(w)
        cfg = ik.Configuration(m, q)
        G_np, h_np = ik.limit_inequalities(cfg, spec, 0.01)
        G_c, h_c = prob.collision_rows(q, 0.01)
        np.testing.assert_array_equal(np.isfinite(h_c), np.isfinite(h_np))
        fin = np.isfinite(h_np)
        n_act += int(fin.sum()); n_pen += int((h_np[fin] == 0.0).sum())
        worst_h = max(worst_h, np.abs(h_c[fin] - h_np[fin]).max())
        worst_G = max(worst_G, np.abs(G_c - G_np).max())
    assert n_act > 50 and n_pen > 5, (n_act, n_pen)
    assert worst_h < 1e-11 and worst_G < 1e-12, (worst_h, worst_G)


def test_c_oracle_solves_g1_with_its_primitive_collision_pairs():
    """`g1_coll` (46 analytic pairs incl. cylinders and boxes, all rows stacked as the reference does: solve_ik.py:25-40):
    the C restatement against the numpy one on a few instances of the bench distribution."""
    from mink_amd import workloads
    m = workloads.load_bench_robot("g1_coll")
    pairs = workloads.g1_collision_pairs(m)
    stand = m.key_qpos[m.name2id("key", "stand")]
    rng = np.random.default_rng(5)
    q = workloads.sample_q(m, rng, 6, stand)
    tg = np.zeros((6, 4, 7)); tg[:, :, 0] = 1.0
    for i in range(6):
        cfg = ik.Configuration(m, workloads.sample_q(m, rng, 1, stand)[0])
        for k, s in enumerate(("left_foot", "right_foot", "left_palm", "right_palm")):
            tg[i, k] = cfg.get_transform_frame_to_world(m.name2id("site", s), "site")
    col = ik.CollisionAvoidanceLimitSpec([tuple(p) for p in pairs], gain=0.85, minimum_distance_from_collisions=0.005,
                                         collision_detection_distance=0.25)
    _, tasks, limits, dt, damping = oc.g1_c3(tg[0], stand)
    prob = cport.CProblem(m, tasks, limits + [col])
    v, st = prob.solve_batch(q, tg, stand[None, :], dt, damping)
    assert (st == 0).all()
    for i in range(6):
        _, t_i, _, _, _ = oc.g1_c3(tg[i], stand)
        v_ref = ik.solve_ik(m, q[i], t_i, dt, damping, limits + [col])
        np.testing.assert_allclose(v[i], v_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(v_ref).max()))


def test_c_batched_com_targets(golden_dir):
    d = _load(golden_dir, "g1_full")
    n = len(d["q"])
    m, tasks, limits, dt, damping = oc.g1_full(d["frame_targets"][0], d["posture_target"], d["com_target"][0])
    prob = cport.CProblem(m, tasks, limits)
    v, st = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping,
                             com_target=d["com_target"].reshape(n, 1, 3))
    assert (st == 0).all()
    # (the C side groups objectives by kind: H differs from mink's order at 1e-16, v at 1e-10 of its size)
    np.testing.assert_allclose(v, d["v"], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"]).max()))


def test_c_oracle_takes_more_than_64_dofs():
    """A 100-dof hinge / slide chain (the size the two-wavefront device path is checked at): C against numpy."""
    import random_models as rm
    from mink_amd import mjcf
    xml, sites = rm.chain_mjcf(100)
    m = mjcf.loads_mjcf(xml)
    assert m.nv == 100
    rng = np.random.default_rng(2)
    q = rm.rand_q(m, rng)
    tgt = ik.Configuration(m, rm.rand_q(m, rng))
    tasks = [ik.FrameTaskSpec(m.name2id("site", s), "site", np.array([1.0, 1.0, 1.0, 0.3, 0.3, 0.3]),
                              tgt.get_transform_frame_to_world(m.name2id("site", s), "site"), lm_damping=0.5) for s in sites]
    tasks.append(ik.PostureTaskSpec(np.full(m.nv, 0.05), np.array(m.qpos0)))
    limits = [ik.ConfigurationLimitSpec(), ik.VelocityLimitSpec(np.arange(m.nv), np.full(m.nv, 1.0))]
    v_np = ik.solve_ik(m, q, tasks, 0.02, 1e-4, limits)
    v_c = cport.CProblem(m, tasks, limits).solve(q, 0.02, 1e-4)
    np.testing.assert_allclose(v_c, v_np, rtol=0, atol=1e-10 * max(1.0, np.abs(v_np).max()))
    assert (np.abs(np.abs(v_np) - 1.0) < 1e-9).sum() > 3          # velocity bounds bind


def test_c_dense_rows_vs_numpy(golden_dir):
    """Caller-defined task / limit rows in the C restatement (mink/tasks/task.py:105-138, limits/limit.py:34-57) against the
    numpy restatement's DenseTaskSpec / DenseLimitSpec on G1 config 3 + 5 task rows + 3 limit rows (one inactive)."""
    d = _load(golden_dir, "g1_c3")
    m, tasks, limits, dt, damping = oc.g1_c3(d["frame_targets"][0], d["posture_target"])
    n, nv = len(d["q"]), m.nv
    rng = np.random.default_rng(11)
    e = rng.normal(scale=0.05, size=(n, 5)); J = rng.normal(size=(n, 5, nv))
    G = rng.normal(size=(n, 3, nv)); h = np.abs(rng.normal(scale=1e-3, size=(n, 3))) + 1e-4
    h[:, 1] = np.inf
    dts = [{"cost": np.array([3.0, 2.0, 1.0]), "gain": 0.8, "lm_damping": 0.5}, {"cost": np.array([4.0, 0.0])}]
    prob = cport.CProblem(m, tasks, limits, dense_tasks=dts, dense_limit_rows=3)
    v, st = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping,
                             dense={"task_e": e, "task_J": J, "limit_G": G, "limit_h": h})
    assert (st == 0).all()
    bound = 0
    for i in range(n):
        mm, t_i, l_i, _, _ = oc.g1_c3(d["frame_targets"][i], d["posture_target"])
        t_i = t_i + [ik.DenseTaskSpec(e[i, :3], J[i, :3], dts[0]["cost"], 0.8, 0.5), ik.DenseTaskSpec(e[i, 3:], J[i, 3:], dts[1]["cost"])]
        v_np = ik.solve_ik(mm, d["q"][i], t_i, dt, damping, l_i + [ik.DenseLimitSpec(G[i], h[i])])
        np.testing.assert_allclose(v[i], v_np, rtol=0, atol=1e-10 * max(1.0, np.abs(v_np).max()))
        bound += int((np.abs(G[i, [0, 2]] @ (v_np * dt) - h[i, [0, 2]]) < 1e-9).sum())
    assert bound > 0                        # the caller's rows bind somewhere


def test_c_oracle_vs_the_big_shadow_and_ur5e_real_mink_fixtures(golden_dir):
    """tests/golden/make_golden_big2.py: 2 048 Shadow instances with contact rows (up to 33 contacts in range) and 4 096 UR5e
    instances from the real mink pin the checker that the GPU tests hold whole bench batches against."""
    d = _load(golden_dir, "shadow_c4_big")
    m, tasks, limits, dt, damping = oc.shadow_c4(d["frame_targets"][0], d["posture_target"])
    prob = cport.CProblem(m, tasks, limits)
    v, st = prob.solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping, nthreads=4)
    assert (st == 0).all()
    main = np.ones(len(v), bool); main[7::8] = False
    err = np.abs(v - d["v"]) / np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5, (err[main].max(), err[~main].max())
    for i in range(0, len(v), 64):
        _, h = prob.collision_rows(d["q"][i], dt)
        fin = np.isfinite(d["coll_h"][i])
        np.testing.assert_array_equal(np.isfinite(h), fin)
        np.testing.assert_allclose(h[fin], d["coll_h"][i][fin], rtol=0, atol=1e-11 * max(1.0, np.abs(h[fin]).max()))
    d = _load(golden_dir, "ur5e_c2_big")
    m, tasks, limits, dt, damping = oc.ur5e_c2(d["frame_targets"][0], d["posture_target"])
    v, st = cport.CProblem(m, tasks, limits).solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping, nthreads=4)
    assert (st == 0).all()
    main = np.ones(len(v), bool); main[7::8] = False
    err = np.abs(v - d["v"]) / np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5, (err[main].max(), err[~main].max())


def test_c_relative_frame_task_vs_the_real_mink_fixture(golden_dir):
    """Round 5: RelativeFrameTask (mink/tasks/relative_frame_task.py:106-142) in the C restatement — the real-mink fixture of the
    reference's arm + hand task set (tests/golden/make_golden_mid.py::arm_hand): H, c of every instance through mko_solve_ik,
    v of the batch."""
    import os
    from mink_amd.flatmodel import FlatModel
    from test_oracle_ik import _arm_hand_specs
    d = _load(golden_dir, "arm_hand")
    m = FlatModel.load(os.path.join(golden_dir, "models", "arm_hand.json"))
    dt, damping = float(d["dt"]), float(d["damping"])
    for i in range(0, len(d["q"]), 5):
        tasks, limits = _arm_hand_specs(m, d, i)
        v, (H, c) = cport.CProblem(m, tasks, limits).solve(d["q"][i], dt, damping, return_problem=True)
        np.testing.assert_allclose(H, d["H"][i], rtol=0, atol=1e-11 * max(1.0, np.abs(d["H"][i]).max()))
        np.testing.assert_allclose(c, d["c"][i], rtol=0, atol=1e-11 * max(1.0, np.abs(d["c"][i]).max()))
        np.testing.assert_allclose(v, d["v"][i], rtol=0, atol=1e-9 * max(1.0, np.abs(d["v"][i]).max()))
    tasks, limits = _arm_hand_specs(m, d, 0)
    # (batch layout: frame-target slots in the caller's task order — FrameTask first, then the four RelativeFrameTasks)
    v, st = cport.CProblem(m, tasks, limits).solve_batch(d["q"], d["frame_targets"], d["posture_target"][None, :], dt, damping)
    assert (st == 0).all()
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    err = np.abs(v - d["v"]) / vs
    main = np.ones(len(v), bool); main[7::8] = False           # (every eighth instance: the small-angle sub-stream, 1e-5)
    assert err[main].max() < 1e-8 and err[~main].max() < 1e-5, (err[main].max(), err[~main].max())
