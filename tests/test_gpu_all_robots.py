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
