"""The workgroup-per-problem kernel (mink_amd/csrc/wide_kernel.h): what one wavefront cannot hold.

The reference stacks every row its limits return (mink/solve_ik.py:25-40) on any nv (:43-65); the wavefront kernels hold
nv + active rows ≤ 64 and nbody ≤ 64.  Beyond that the library (a) runs models with more bodies / dofs entirely on the wide
kernel, (b) re-solves the instances a wavefront kernel flagged MKH_ST_ROW_OVERFLOW with EVERY detected contact a row."""

import os
import subprocess
import sys

import numpy as np
import pytest

import native_configs as nc
import oracle_configs as oc
import random_models as rm
from oracle import cport
from oracle import ik as oik

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chain_problem(ndof, B, seed=3):
    from mink_amd import _native as nat
    from mink_amd import mjcf
    xml, sites = rm.chain_mjcf(ndof, seed=seed)
    m = mjcf.loads_mjcf(xml)
    rng = np.random.default_rng(seed)
    q = np.array([rm.rand_q(m, rng) for _ in range(B)])
    tg = np.empty((B, len(sites), 7))
    for i in range(B):
        cfg = oik.Configuration(m, rm.rand_q(m, rng))
        for k, s in enumerate(sites):
            tg[i, k] = cfg.get_transform_frame_to_world(m.name2id("site", s), "site")
    cost = [1.0, 1.0, 1.0, 0.3, 0.3, 0.3]
    fts = [{"frame_type": "site", "frame_id": m.name2id("site", s), "cost": cost, "gain": 1.0, "lm_damping": 0.5} for s in sites]
    nm = nat.NativeModel(m)
    idx, lower, upper = oik.configuration_limit_arrays(m, oik.ConfigurationLimitSpec())
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 0.05}],
                             configuration_limits=[{"gain": 0.95, "lower": lower, "upper": upper, "indices": idx}],
                             velocity_limits=[{"indices": np.arange(m.nv), "limit": np.full(m.nv, 1.0)}], max_batch=B)
    tasks = [oik.FrameTaskSpec(m.name2id("site", s), "site", np.array(cost), tg[0, k], lm_damping=0.5) for k, s in enumerate(sites)]
    tasks.append(oik.PostureTaskSpec(np.full(m.nv, 0.05), np.array(m.qpos0)))
    limits = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.arange(m.nv), np.full(m.nv, 1.0))]
    return m, prob, q, tg, tasks, limits


@pytest.mark.parametrize("ndof", [100, 70])
def test_chain_beyond_one_wavefront_against_the_c_oracle(ndof):
    """A serial chain of `ndof` one-dof bodies (ndof + 1 bodies: past both one-wavefront limits), frame tasks along it, a
    posture task, ConfigurationLimit and VelocityLimit: every instance against the plain-C restatement of the reference
    pipeline at the stated 1e-8·max(1, ‖v_ref‖∞)."""
    B = 96
    m, prob, q, tg, tasks, limits = _chain_problem(ndof, B)
    assert m.nv == ndof and m.nbody == ndof + 1
    dt, damping = 0.02, 1e-4
    v, st = prob.solve(q, tg, np.array(m.qpos0)[None, :], None, dt, damping)
    assert prob.last_kernel() == "ik_wide_kernel", prob.last_kernel()
    assert ((st & ~1) == 0).all(), np.unique(st, return_counts=True)
    v_ref, st_ref = cport.CProblem(m, tasks, limits).solve_batch(q, tg, np.array(m.qpos0)[None, :], dt, damping, nthreads=4)
    assert (st_ref == 0).all()
    err = np.abs(v - v_ref).max(axis=1) / np.maximum(1.0, np.abs(v_ref).max(axis=1))
    bound = (np.abs(np.abs(v_ref) - 1.0) < 1e-9).sum(axis=1)
    print("chain of %d dofs: max rel err %.2e; velocity bounds binding per instance: mean %.1f" % (ndof, err.max(), bound.mean()))
    assert err.max() < 1e-8 and bound.mean() > 2
    # round 5: per-task (e, J) taps and the iteration counts come from this kernel too — against the numpy oracle's
    # compute_error / compute_jacobian of every task (mink/tasks/task.py:81-103)
    _, _, taps = prob.solve(q[:6], tg[:6], np.array(m.qpos0)[None, :], None, dt, damping, taps=["task_e", "task_J", "qp_iters"])
    assert prob.last_kernel() == "ik_wide_kernel"
    for i in range(0, 6, 2):
        cfg = oik.Configuration(m, q[i])
        e_ref, J_ref = [], []
        for k, t in enumerate(tasks):
            if isinstance(t, oik.FrameTaskSpec):
                t = oik.FrameTaskSpec(t.frame_id, t.frame_type, t.cost, tg[i, k], lm_damping=t.lm_damping)
            e, J = oik.task_error_jacobian(cfg, t)
            e_ref.append(e); J_ref.append(J)
        e_ref, J_ref = np.concatenate(e_ref), np.vstack(J_ref)
        assert e_ref.shape == taps["task_e"][i].shape
        np.testing.assert_allclose(taps["task_e"][i], e_ref, rtol=0, atol=1e-12 * max(1.0, np.abs(e_ref).max()))
        np.testing.assert_allclose(taps["task_J"][i], J_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(J_ref).max()))
    assert (taps["qp_pivots"] > 0).any() and (taps["qp_iters"] >= taps["qp_loops"]).all()


def _oracle_loop(m, tasks, limits, q0, tg_i, dt, damping, max_iters, thresholds=None, posture=None):
    """The callers' loop (examples/arm_ur5e_actuators.py:88-97) on the numpy oracle: returns q, last v, iterations, converged."""
    cfg = oik.Configuration(m, q0)
    ts = []
    k = 0
    for t in tasks:
        if isinstance(t, oik.FrameTaskSpec):
            ts.append(oik.FrameTaskSpec(t.frame_id, t.frame_type, t.cost, tg_i[k], lm_damping=t.lm_damping)); k += 1
        elif posture is not None and isinstance(t, oik.PostureTaskSpec):
            ts.append(oik.PostureTaskSpec(t.cost, posture))
        else:
            ts.append(t)
    done, n, v_ref = False, 0, None
    for n in range(1, max_iters + 1):
        v_ref = oik.solve_ik(m, cfg, ts, dt, damping, limits)
        cfg.update(cfg.integrate(v_ref, dt))
        if thresholds is not None:
            ok = True
            for t in ts:
                if isinstance(t, oik.FrameTaskSpec):
                    err = oik.task_error_jacobian(cfg, t)[0]
                    ok = ok and np.linalg.norm(err[:3]) <= thresholds[0] and np.linalg.norm(err[3:]) <= thresholds[1]
            if ok:
                done = True
                break
    return cfg.q.copy(), v_ref, n, done


def test_fused_loops_beyond_one_wavefront():
    """mkh_solve_steps / mkh_solve_until on a 70-dof chain: the step loop runs inside the workgroup-per-problem kernel (round 5;
    round 4 returned MKH_E_INVALID).  Against the same loop of single launches, and against the callers' loop on the numpy
    oracle: final q, last v, per-instance iteration counts and converged flags."""
    B = 40
    m, prob, q, tg, tasks, limits = _chain_problem(70, B, seed=9)
    from mink_amd import _native as nat
    nm = prob.nmodel
    q0 = np.array(m.qpos0)[None, :]
    dt, damping = 0.05, 1e-4
    # reachable targets at very different distances: FK of q ⊕ δ
    rng = np.random.default_rng(4)
    scale = np.repeat([2e-4, 2e-3, 0.02, 0.2], B // 4)[:, None]
    qt = nm.integrate(q, rng.normal(size=(B, m.nv)) * scale, 1.0)
    dummy = np.zeros((B, prob.n_frame, 7)); dummy[:, :, 0] = 1
    _, _, t = prob.solve(qt, dummy, q0, None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
    tg = t["frame_pose"]
    K = 4
    qK, vK, st = prob.solve(q, tg, q0, None, dt, damping, n_steps=K)
    assert prob.last_kernel() == "ik_wide_kernel" and ((st & ~1) == 0).all()
    qs = q.copy()
    for _ in range(K):
        vs, _ = prob.solve(qs, tg, q0, None, dt, damping)
        qs = nm.integrate(qs, vs, dt)
    np.testing.assert_allclose(qK, qs, rtol=0, atol=1e-11)
    np.testing.assert_allclose(vK, vs, rtol=0, atol=1e-9 * max(1.0, np.abs(vs).max()))
    # threshold-terminated
    # (consistent targets: the posture target is the configuration the frame targets were taken from, per instance)
    max_iters, thr = 12, (2e-3, 2e-3)
    qU, vU, stU, it, cv = prob.solve(q, tg, qt[:, None, :].copy(), None, dt, damping, n_steps=max_iters, until=thr)
    assert ((stU & ~1) == 0).all()
    print("70-dof chain, until: iterations", np.bincount(it, minlength=max_iters + 1).tolist(), "converged", int(cv.sum()), "of", B)
    assert cv.sum() >= 4 and (cv == 0).sum() >= 4 and len(set(it[cv == 1].tolist())) >= 2
    for i in range(0, B, 3):
        q_ref, v_ref, n, done = _oracle_loop(m, tasks, limits, q[i], tg[i], dt, damping, max_iters, thr, posture=qt[i])
        assert (it[i], bool(cv[i])) == (n, done), (i, it[i], cv[i], n, done)
        np.testing.assert_allclose(qU[i], q_ref, rtol=0, atol=1e-9)
        np.testing.assert_allclose(vU[i], v_ref, rtol=0, atol=1e-7 * max(1.0, np.abs(v_ref).max()))
    # q_out may alias q (include/minkhip.h): in place
    q_in = q.copy()
    prob.solve(q_in, tg, q0, None, dt, damping, n_steps=K, q_out=q_in)
    np.testing.assert_array_equal(q_in, qK)


def test_wide_kernel_as_a_model_path_at_scale():
    """The workgroup-per-problem kernel as the ONLY kernel of a model, at a batch that puts several rounds on every workgroup
    and ends ragged (4 096 + 37 instances of the 100-dof chain): every instance against the plain-C restatement at the stated
    1e-8·max(1, ‖v_ref‖∞) (round 4 held this path on 96 instances)."""
    import os
    B = 4096 + 37
    m, prob, q, tg, tasks, limits = _chain_problem(100, B, seed=11)
    dt, damping = 0.02, 1e-4
    v, st = prob.solve(q, tg, np.array(m.qpos0)[None, :], None, dt, damping)
    assert prob.last_kernel() == "ik_wide_kernel"
    assert ((st & ~1) == 0).all(), np.unique(st, return_counts=True)
    info = prob.launch_info(B)
    assert info["block"] == 256 and 0 < info["grid"] < B, info
    v_ref, st_ref = cport.CProblem(m, tasks, limits).solve_batch(q, tg, np.array(m.qpos0)[None, :], dt, damping,
                                                                 nthreads=min(16, os.cpu_count() or 1))
    assert (st_ref == 0).all()
    err = np.abs(v - v_ref).max(axis=1) / np.maximum(1.0, np.abs(v_ref).max(axis=1))
    print("100-dof chain, %d instances on %d workgroups: max rel err vs C oracle %.2e" % (B, info["grid"], err.max()))
    assert err.max() < 1e-8


# every pair in range, and most of them closer than d_min: h = 0 rows — approaching is forbidden — far more than 21 at once
DETECT, SIGMA, DMIN = 2.0, 0.3, 0.35


def _g1_with_contacts(B):
    """G1 config 3 + a CollisionAvoidanceLimit over 48 pairs of the model's primitive collision geoms (foot spheres, leg and arm
    cylinders, hand boxes, the floor) with a detection distance that puts most of them in range at once."""
    from mink_amd import _native as nat
    from mink_amd import workloads
    model = workloads.load_robot("g1")
    gt, gv = np.asarray(model.geom_type), np.asarray(model.geom_valid)
    floor = [g for g in range(model.ngeom) if gt[g] == 0][0]
    prim = [g for g in range(model.ngeom) if gt[g] in (2, 5, 6) and gv[g] == 1 and int(model.geom_bodyid[g]) < 39]
    sph = [g for g in prim if gt[g] == 2]
    rest = [g for g in prim if gt[g] != 2]
    left, right = sph[:4], sph[4:8]
    pairs = [(g, floor) for g in sph] + [(a, b) for a in left for b in right]
    pairs += [(a, b) for i, a in enumerate(rest) for b in rest[i + 1:] if model.geom_bodyid[a] != model.geom_bodyid[b]]
    pairs += [(g, floor) for g in rest]
    pairs = pairs[:48]
    assert len(pairs) >= 44
    nm = nat.NativeModel(model)
    fts = [nc._ft(model, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + \
          [nc._ft(model, s, "site", 200.0, 0.0, 1.0) for s in ("left_palm", "right_palm")]
    col = {"geom_id_pairs": np.array(pairs), "gain": 0.85, "minimum_distance_from_collisions": DMIN,
           "collision_detection_distance": DETECT, "bound_relaxation": 0.0}
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1.0}], configuration_limits=[nc._cfg_limit(model)],
                             velocity_limits=[nc._vel_limit(model)], collision_limits=[col], max_batch=B)
    stand = model.key_qpos[model.name2id("key", "stand")]
    # (targets a whole radian away per joint: steps large enough that contacts far from the tightest 21 bind as well)
    q, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(17), B, base_q=stand, sigma=SIGMA)
    return model, prob, pairs, q, tg, stand


def test_g1_with_more_contacts_than_tableau_rows_gets_every_row():
    """G1 (nv 43) holds 21 half-space rows on the wavefront kernel.  With 40+ contacts in range at once the wavefront launch
    keeps the tightest 21, checks the rest at its solution and flags the instances where one of them is violated; the wide
    redo launch solves those with EVERY detected contact a row — mink's answer (solve_ik.py:25-40 stacks them all), held against
    the numpy oracle's all-rows solve.  No instance is left with MKH_ST_ROW_OVERFLOW."""
    B = 4096
    model, prob, pairs, q, tg, stand = _g1_with_contacts(B)
    dt, damping = 5e-2, 1e-1
    v, st = prob.solve(q, tg, stand[None, :], None, dt, damping)
    assert prob.last_kernel().endswith("+wide"), prob.last_kernel()
    assert ((st & ~1) == 0).all(), np.unique(st, return_counts=True)
    # what the wavefront launch alone leaves flagged (the regime must exercise the redo)
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r]
import test_gpu_wide as T
from mink_amd import _native as nat
with nat.diag_options(nat.DIAG_NO_WIDE_REDO):
    model, prob, pairs, q, tg, stand = T._g1_with_contacts(%d)
v, st = prob.solve(q, tg, stand[None, :], None, 5e-2, 1e-1)
print("FLAGGED", int(((st & 16) != 0).sum()), prob.last_kernel())
np.save(%r, st)
""" % (REPO, os.path.join(REPO, "tests"), B, "/tmp/mkh_wide_st.npy")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    flagged = np.flatnonzero(np.load("/tmp/mkh_wide_st.npy") & 16)
    print(r.stdout.strip().splitlines()[-1])
    assert len(flagged) >= 16, len(flagged)
    # EVERY re-solved instance (and a few others) against the numpy oracle with all rows, over the host's cores
    import multiprocessing as mp
    idx = [int(i) for i in flagged] + [int(j) for j in np.setdiff1d(np.arange(B), flagged)[:4]]
    ncpu = min(16, os.cpu_count() or 1)
    with mp.get_context("fork").Pool(ncpu) as pool:
        res = pool.map(_all_rows_oracle, [(pairs, q[i], tg[i], stand, dt, 1) for i in idx])
    worst = max(np.abs(v[i] - r[0][-1]).max() / max(1.0, np.abs(r[0][-1]).max()) for i, r in zip(idx, res))
    most = max(r[1] for r in res)
    print("G1 + %d pairs: %d instances re-solved on the wide kernel (all checked), up to %d contacts in range; max rel err vs all-rows oracle %.2e"
          % (len(pairs), len(flagged), most, worst))
    assert most > 21 and worst < 5e-6          # (cylinder pairs go through GJK: rows to ~1e-6)


def _all_rows_oracle(args):
    """n steps of solve + integrate on the numpy oracle with EVERY contact a row; returns ([v per step], most contacts in range, q)."""
    pairs, q_i, tg_i, stand, dt, n = args
    m = oc.model("g1")
    spec = oik.CollisionAvoidanceLimitSpec([tuple(p) for p in pairs], collision_detection_distance=DETECT, minimum_distance_from_collisions=DMIN)
    cfg = oik.Configuration(m, q_i)
    vs, most = [], 0
    for _ in range(n):
        mm, tasks, limits, _, damp_o = oc.g1_c3(tg_i, stand)
        v_ref, (_, _, G, h) = oik.solve_ik(m, cfg, tasks, dt, damp_o, limits + [spec], return_problem=True)
        most = max(most, int(np.isfinite(h[-len(pairs):]).sum()))
        vs.append(v_ref)
        cfg.update(cfg.integrate(v_ref, dt))
    return vs, most, cfg.q.copy()


def test_fused_loop_and_taps_with_more_contacts_than_tableau_rows():
    """Round 5: the fused loops and calls with taps no longer report MKH_ST_ROW_OVERFLOW (round 4: SolverError where the reference
    stacks every row, mink/solve_ik.py:25-40).  An instance the wavefront kernel flags inside mkh_solve_steps keeps its q
    (which q_out may alias) and runs its whole loop again on the workgroup-per-problem kernel; a call with taps gets the
    flagged instances' taps from that kernel."""
    import multiprocessing as mp
    B, K = 1024, 3
    model, prob, pairs, q, tg, stand = _g1_with_contacts(B)
    dt, damping = 5e-2, 1e-1
    q_in = q.copy()
    qK, vK, st = prob.solve(q_in, tg, stand[None, :], None, dt, damping, n_steps=K, q_out=q_in)       # in place
    assert prob.last_kernel().endswith("+wide"), prob.last_kernel()
    assert ((st & ~1) == 0).all(), np.unique(st, return_counts=True)
    # which instances overflowed inside the loop: the same call without the redo launch
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r]
import test_gpu_wide as T
from mink_amd import _native as nat
with nat.diag_options(nat.DIAG_NO_WIDE_REDO):
    model, prob, pairs, q, tg, stand = T._g1_with_contacts(%d)
qK, vK, st = prob.solve(q, tg, stand[None, :], None, 5e-2, 1e-1, n_steps=%d)
np.save(%r, st)
""" % (REPO, os.path.join(REPO, "tests"), B, K, "/tmp/mkh_wide_st_loop.npy")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    flagged = np.flatnonzero(np.load("/tmp/mkh_wide_st_loop.npy") & 16)
    assert len(flagged) >= 2, len(flagged)
    idx = [int(i) for i in flagged[:24]] + [int(j) for j in np.setdiff1d(np.arange(B), flagged)[:4]]
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:
        res = pool.map(_all_rows_oracle, [(pairs, q[i], tg[i], stand, dt, K) for i in idx])
    worst_q = max(np.abs(qK[i] - r[2]).max() for i, r in zip(idx, res))
    worst_v = max(np.abs(vK[i] - r[0][-1]).max() / max(1.0, np.abs(r[0][-1]).max()) for i, r in zip(idx, res))
    print("G1 + %d pairs, %d fused steps: %d instances overflowed inside the loop; max |q - oracle| %.2e, rel v %.2e"
          % (len(pairs), K, len(flagged), worst_q, worst_v))
    assert worst_q < 5e-6 and worst_v < 5e-5
    # threshold-terminated: iteration counts / flags of the re-run instances are the wide kernel's
    qU, vU, stU, it, cv = prob.solve(q, tg, stand[None, :], None, dt, damping, n_steps=K, until=(1e-3, 1e-3))
    assert ((stU & ~1) == 0).all() and (it >= 1).all() and (it <= K).all()
    # taps: every detected contact's row of a flagged instance, H and v from one call
    v1, st1, taps = prob.solve(q, tg, stand[None, :], None, dt, damping, taps=["coll_h", "coll_G", "H", "task_e"])
    assert prob.last_kernel().endswith("+wide") and ((st1 & ~1) == 0).all()
    v0, _ = prob.solve(q, tg, stand[None, :], None, dt, damping)
    # (plain solves take the contacts of general convex pairs from the kernel in front — other FK rounding, GJK witness points to
    #  ~1e-6 —, the call with taps from the routine inside the solve kernel: the tolerance of rows through GJK)
    np.testing.assert_allclose(v1, v0, rtol=0, atol=5e-6 * max(1.0, np.abs(v0).max()))
    i = idx[0]
    m = oc.model("g1")
    mm, tasks, limits, _, damp_o = oc.g1_c3(tg[i], stand)
    spec = oik.CollisionAvoidanceLimitSpec([tuple(p) for p in pairs], collision_detection_distance=DETECT, minimum_distance_from_collisions=DMIN)
    _, (H, c, G, h) = oik.solve_ik(m, q[i], tasks, dt, damp_o, limits + [spec], return_problem=True)
    h_ref = h[-len(pairs):]
    np.testing.assert_array_equal(np.isfinite(taps["coll_h"][i]), np.isfinite(h_ref))
    fin = np.isfinite(h_ref)
    np.testing.assert_allclose(taps["coll_h"][i][fin], h_ref[fin], rtol=0, atol=1e-6)
    np.testing.assert_allclose(taps["H"][i], H, rtol=0, atol=1e-10 * np.abs(H).max())


def test_public_api_on_a_model_beyond_one_wavefront():
    """mink's own call sequence on the 100-dof chain — Configuration, FrameTask.set_target from another configuration's frame
    pose, PostureTask, limits, build_ik, solve_ik, integrate_inplace — entirely on the wide kernel (frame poses, H, c through its
    taps), against the numpy oracle."""
    import mink_amd as mink
    xml, sites = rm.chain_mjcf(100, seed=5)
    m = mink.loads_mjcf(xml)
    rng = np.random.default_rng(8)
    B = 8
    q = np.array([rm.rand_q(m, rng) for _ in range(B)])
    q2 = np.array([rm.rand_q(m, rng) for _ in range(B)])
    cfg, goal = mink.Configuration(m, q), mink.Configuration(m, q2)
    tasks = []
    for s in sites:
        t = mink.FrameTask(s, "site", position_cost=1.0, orientation_cost=0.3, lm_damping=0.5)
        T = goal.get_transform_frame_to_world(s, "site")
        t.set_target(T)
        tasks.append(t)
        o = oik.Configuration(m, q2[0]).get_transform_frame_to_world(m.name2id("site", s), "site")
        got = T.wxyz_xyz[0].copy()
        if got[:4] @ o[:4] < 0:
            got[:4] = -got[:4]                                  # (q and −q are the same rotation)
        np.testing.assert_allclose(got, o, rtol=0, atol=1e-12)                    # FK of 100 links, device against oracle
    post = mink.PostureTask(m, cost=0.05); post.set_target(np.array(m.qpos0)); tasks.append(post)
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, {n: 1.0 for n in m.jnt_names})]
    dt, damping = 0.02, 1e-4
    prob = mink.build_ik(cfg, tasks, dt, damping, lims)
    v = mink.solve_ik(cfg, tasks, dt, "mi355x", damping, limits=lims)
    for i in range(0, B, 3):
        ots = [oik.FrameTaskSpec(m.name2id("site", s), "site", np.array([1.0, 1.0, 1.0, 0.3, 0.3, 0.3]), tasks[k].transform_target_to_world.wxyz_xyz[i], lm_damping=0.5)
               for k, s in enumerate(sites)] + [oik.PostureTaskSpec(np.full(m.nv, 0.05), np.array(m.qpos0))]
        olims = [oik.ConfigurationLimitSpec(), oik.VelocityLimitSpec(np.arange(m.nv), np.full(m.nv, 1.0))]
        v_ref, (H, c, G, h) = oik.solve_ik(m, q[i], ots, dt, damping, olims, return_problem=True)
        np.testing.assert_allclose(prob.P[i], H, rtol=0, atol=1e-11 * np.abs(H).max())
        np.testing.assert_allclose(prob.q[i], c, rtol=0, atol=1e-11 * max(1.0, np.abs(c).max()))
        np.testing.assert_allclose(v[i], v_ref, rtol=0, atol=1e-8 * max(1.0, np.abs(v_ref).max()))
    # round 5 — what round 4 refused on this model: Configuration.get_frame_jacobian (mink/configuration.py:112-155),
    # Task.compute_error / compute_jacobian (mink/tasks/task.py:81-103) and the callers' loop in one launch
    s0 = sites[0]
    Jb = cfg.get_frame_jacobian(s0, "site")
    e0, J0 = tasks[0].compute_error(cfg), tasks[0].compute_jacobian(cfg)
    for i in range(0, B, 3):
        ocfg = oik.Configuration(m, q[i])
        np.testing.assert_allclose(Jb[i], ocfg.get_frame_jacobian(m.name2id("site", s0), "site"), rtol=0, atol=1e-10)
        spec = oik.FrameTaskSpec(m.name2id("site", s0), "site", np.array([1.0, 1.0, 1.0, 0.3, 0.3, 0.3]),
                                 tasks[0].transform_target_to_world.wxyz_xyz[i], lm_damping=0.5)
        e_ref, J_ref = oik.task_error_jacobian(ocfg, spec)
        np.testing.assert_allclose(e0[i], e_ref, rtol=0, atol=1e-12)
        np.testing.assert_allclose(J0[i], J_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(J_ref).max()))
    cfg2 = mink.Configuration(m, q)
    q3, v3 = mink.solve_ik_steps(cfg2, tasks, dt, 3, "mi355x", damping=damping, limits=lims)
    qs = q.copy()
    for _ in range(3):
        c_ = mink.Configuration(m, qs)
        vs = mink.solve_ik(c_, tasks, dt, "mi355x", damping, limits=lims)
        c_.integrate_inplace(vs, dt)
        qs = c_.q
    np.testing.assert_allclose(q3, qs, rtol=0, atol=1e-11)
    np.testing.assert_allclose(v3, vs, rtol=0, atol=1e-9 * max(1.0, np.abs(vs).max()))
    cfg.integrate_inplace(v, dt)
    np.testing.assert_allclose(cfg.q[0], oik.Configuration(m, q[0]).integrate(v[0], dt), rtol=0, atol=1e-13)
