"""The two CPU restatements (numpy, plain C) against each other on randomised kinematic trees: ball / slide /
hinge / free joints, several joints per body, frames on bodies and sites, ComTask, both limit types.  CPU only."""

import numpy as np
import pytest

from mink_amd.mjcf import loads_mjcf
from oracle import cport
from oracle import ik as oik
from random_models import rand_q, random_mjcf


@pytest.mark.parametrize("seed", range(8))
def test_c_vs_numpy_on_random_tree(seed):
    rng = np.random.default_rng(500 + seed)
    nbody = int(rng.integers(3, 30))
    xml, sites = random_mjcf(rng, nbody, free_root=bool(seed % 2))
    m = loads_mjcf(xml)
    if m.nv == 0 or m.nv > 48:
        pytest.skip("degenerate draw")
    frames = [(m.name2id("site", s), "site") for s in sites] + [(i + 1, "body") for i in range(nbody)]
    for trial in range(3):
        q = rand_q(m, rng)
        cfg = oik.Configuration(m, q)
        tgt = oik.Configuration(m, cfg.integrate(rng.normal(scale=0.15, size=m.nv), 1.0))
        tasks = []
        for k in rng.choice(len(frames), size=min(2, len(frames)), replace=False):
            fid, typ = frames[k]
            cost = np.concatenate([rng.uniform(0.5, 20.0, size=3), np.full(3, rng.uniform(0.0, 3.0))])
            tasks.append(oik.FrameTaskSpec(fid, typ, cost, tgt.get_transform_frame_to_world(fid, typ),
                                           float(rng.uniform(0.3, 1.0)), float(rng.uniform(0.0, 1.0))))
        tasks.append(oik.PostureTaskSpec(rng.uniform(0.05, 1.0, size=m.nv), rand_q(m, rng), 0.8))
        if trial == 1:
            tasks.append(oik.ComTaskSpec(rng.uniform(0.5, 5.0, size=3), rng.normal(scale=0.3, size=3)))
        idx = [int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] in (2, 3)]
        limits = [oik.ConfigurationLimitSpec(0.9)] + ([oik.VelocityLimitSpec(np.array(idx), np.full(len(idx), 2.0))] if idx else [])
        dt, damping = 1e-2, 1e-3
        v_np, (P, c, G, h) = oik.solve_ik(m, cfg, tasks, dt, damping, limits, return_problem=True)
        v_c, (H_c, c_c) = cport.CProblem(m, tasks, limits).solve(q, dt, damping, return_problem=True)
        np.testing.assert_allclose(H_c, P, rtol=0, atol=1e-12 * max(1.0, np.abs(P).max()))
        np.testing.assert_allclose(c_c, c, rtol=0, atol=1e-12 * max(1.0, np.abs(c).max()))
        np.testing.assert_allclose(v_c, v_np, rtol=0, atol=1e-9 * max(1.0, np.abs(v_np).max()))
