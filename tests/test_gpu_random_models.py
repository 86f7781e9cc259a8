"""Randomised kinematic trees: GPU path vs the numpy oracle (intermediates) and the plain-C oracle (every
instance).  The hand-written robots share structure (serial limbs off a floating base); random branching,
joint types, several joints per body, frames on bodies/sites and random costs exercise what they do not —
ancestor/dof masks, pointer-jumping depth, qpos/dof address bookkeeping, the low-rank vs direct QP start."""

import os

import numpy as np
import pytest

import mink_amd as mink
from oracle import cport
from oracle import ik as oik
from oracle import lie as olie  # noqa: F401
from random_models import random_mjcf, rand_q as _rand_q

pytestmark = pytest.mark.gpu


# (MKH_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_random_models.py for a longer run)
@pytest.mark.parametrize("seed", range(int(os.environ.get("MKH_FUZZ_SEEDS", "12"))))
def test_random_tree(seed):
    rng = np.random.default_rng(1000 + seed)
    nbody = int(rng.integers(3, 40))
    xml, sites = random_mjcf(rng, nbody, free_root=bool(seed % 2))
    m = mink.loads_mjcf(xml)
    if m.nv == 0 or m.nv > 48:
        pytest.skip("degenerate draw")
    B = 32
    q = np.stack([_rand_q(m, rng) for _ in range(B)])
    cfg = mink.Configuration(m, q)
    tgt_cfg = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.15, size=(B, m.nv)), 1.0))
    frames = [(s, "site") for s in sites] + [(f"b{i}", "body") for i in range(nbody)]
    n_ft = int(rng.integers(1, 4))
    picks = [frames[i] for i in rng.choice(len(frames), size=min(n_ft, len(frames)), replace=False)]
    tasks, specs = [], []
    for name, typ in picks:
        pc = rng.uniform(0.5, 20.0) * (rng.uniform(size=3) < 0.85)
        oc_ = rng.uniform(0.1, 5.0) * (rng.uniform() < 0.6)
        if not pc.any() and oc_ == 0.0:
            pc = np.ones(3)
        gain, lm = float(rng.uniform(0.3, 1.0)), float(rng.uniform(0.0, 1.0))
        ft = mink.FrameTask(name, typ, position_cost=pc, orientation_cost=oc_, gain=gain, lm_damping=lm)
        ft.set_target(tgt_cfg.get_transform_frame_to_world(name, typ))
        tasks.append(ft)
        specs.append((m.name2id(typ, name), typ, ft.cost.copy(), gain, lm))
    post = mink.PostureTask(m, cost=rng.uniform(0.05, 1.0, size=m.nv), gain=float(rng.uniform(0.2, 1.0)))
    post.set_target(_rand_q(m, rng))
    use_com = rng.uniform() < 0.4
    extra = [post]
    if use_com:
        com = mink.ComTask(cost=rng.uniform(0.5, 5.0, size=3))
        com.set_target(rng.normal(scale=0.3, size=3))
        extra.append(com)
    vel = {}
    for j in range(m.njnt):
        if m.jnt_type[j] in (2, 3) and rng.uniform() < 0.7:
            vel[m.jnt_names[j]] = float(rng.uniform(0.5, 3.0))
    lims = [mink.ConfigurationLimit(m, gain=float(rng.uniform(0.5, 1.0)))]
    if vel:
        lims.append(mink.VelocityLimit(m, vel))
    dt, damping = float(rng.choice([2e-3, 1e-2, 5e-2])), float(rng.choice([1e-6, 1e-3, 1e-1]))
    v = mink.solve_ik(cfg, tasks + extra, dt, "mi355x", damping, limits=lims)
    assert np.isfinite(v).all()

    def spec_lists(i):
        ts = [oik.FrameTaskSpec(fid, typ, cost, ft.transform_target_to_world.wxyz_xyz[i], gain, lm)
              for (fid, typ, cost, gain, lm), ft in zip(specs, tasks)]
        ts.append(oik.PostureTaskSpec(post.cost, post.target_q, post.gain))
        if use_com:
            ts.append(oik.ComTaskSpec(com.cost, com.target_com))
        ls = [oik.ConfigurationLimitSpec(lims[0].gain)]
        if vel:
            ls.append(oik.VelocityLimitSpec(lims[1].indices, lims[1].limit))
        return ts, ls

    # every instance against the C oracle ...
    ts0, ls0 = spec_lists(0)
    cp = cport.CProblem(m, ts0, ls0)
    ftg = np.stack([ft.transform_target_to_world.wxyz_xyz for ft in tasks], axis=1)
    ctg = com.target_com[None, :] if use_com else None
    v_c, st_c = cp.solve_batch(q, ftg, post.target_q[None, :], dt, damping, com_target=ctg)
    assert (st_c == 0).all()
    err = np.abs(v - v_c).max(axis=1) / np.maximum(1.0, np.abs(v_c).max(axis=1))
    assert err.max() < 1e-7, (seed, err.max())
    # ... and two against the numpy oracle, with the task intermediates of one
    for i in (0, B - 1):
        ts, ls = spec_lists(i)
        o = oik.Configuration(m, q[i])
        v_ref = oik.solve_ik(m, o, ts, dt, damping, ls)
        np.testing.assert_allclose(v[i], v_ref, rtol=0, atol=1e-7 * max(1.0, np.abs(v_ref).max()))
    one = mink.Configuration(m, q[0])
    ts, _ = spec_lists(0)
    o = oik.Configuration(m, q[0])
    for task, ot in zip(tasks, ts):
        t1 = mink.FrameTask(task.frame_name, task.frame_type, task.cost[:3], task.cost[3:], task.gain, task.lm_damping)
        t1.set_target(mink.SE3(task.transform_target_to_world.wxyz_xyz[0]))
        e_ref, J_ref = oik.task_error_jacobian(o, ot)
        np.testing.assert_allclose(t1.compute_error(one), e_ref, atol=1e-11)
        np.testing.assert_allclose(t1.compute_jacobian(one), J_ref, atol=1e-9)
