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
    # the per-task (e, J) taps and the fused loops are the wavefront kernels'
    from mink_amd import _native as nat
    with pytest.raises(nat.MinkHipError, match="beyond one wavefront"):
        prob.solve(q, tg, np.array(m.qpos0)[None, :], None, dt, damping, taps=["task_J"])
    with pytest.raises(nat.MinkHipError, match="beyond one wavefront"):
        prob.solve(q, tg, np.array(m.qpos0)[None, :], None, dt, damping, n_steps=3)


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
    B = 512
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
model, prob, pairs, q, tg, stand = T._g1_with_contacts(%d)
v, st = prob.solve(q, tg, stand[None, :], None, 5e-2, 1e-1)
print("FLAGGED", int(((st & 16) != 0).sum()), prob.last_kernel())
np.save(%r, st)
""" % (REPO, os.path.join(REPO, "tests"), B, "/tmp/mkh_wide_st.npy")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MKH_DEBUG_NO_WIDE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    flagged = np.flatnonzero(np.load("/tmp/mkh_wide_st.npy") & 16)
    print(r.stdout.strip().splitlines()[-1])
    assert len(flagged) >= 4, len(flagged)
    # the re-solved instances (and a few others) against the numpy oracle with all rows
    m = oc.model("g1")
    worst, most = 0.0, 0
    for i in list(flagged[:10]) + [int(j) for j in np.setdiff1d(np.arange(B), flagged)[:4]]:
        mm, tasks, limits, _, damp_o = oc.g1_c3(tg[i], stand)
        dt_o = dt
        spec = oik.CollisionAvoidanceLimitSpec([tuple(p) for p in pairs], collision_detection_distance=DETECT, minimum_distance_from_collisions=DMIN)
        v_ref, (_, _, G, h) = oik.solve_ik(m, q[i], tasks, dt_o, damp_o, limits + [spec], return_problem=True)
        most = max(most, int(np.isfinite(h[-len(pairs):]).sum()))
        worst = max(worst, np.abs(v[i] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
    print("G1 + %d pairs: %d instances re-solved on the wide kernel, up to %d contacts in range; max rel err vs all-rows oracle %.2e"
          % (len(pairs), len(flagged), most, worst))
    assert most > 21 and worst < 5e-6          # (cylinder pairs go through GJK: rows to ~1e-6)


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
    cfg.integrate_inplace(v, dt)
    np.testing.assert_allclose(cfg.q[0], oik.Configuration(m, q[0]).integrate(v[0], dt), rtol=0, atol=1e-13)
