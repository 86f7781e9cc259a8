#!/usr/bin/env python
"""Round 6: `ur5e_convex` (cylinder–box through GJK / the expanding polytope + the witness-point polish) on every instance of its
bench batch against the numpy restatement: v, h and the rows of G, split by what the cylinder–wall pair is doing.  GPU only.

    python tools/convex_polish_check.py [B]"""
import multiprocessing as mp
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def _chunk(args):
    name, idx, q, tg, pt, dt, damping = args
    from mink_amd import workloads
    from oracle import ik
    model = workloads.load_bench_robot(name)
    site = model.name2id("site", "attachment_site")
    hinge = [int(model.jnt_dofadr[j]) for j in range(model.njnt) if model.jnt_type[j] != 0]
    g = lambda n: model.name2id("geom", n)
    pairs = [(g("wrist_3_link"), g("floor")), (g("wrist_3_link"), g("wall"))]
    spec = ik.CollisionAvoidanceLimitSpec(pairs, collision_detection_distance=0.3)
    vs, Gs, hs = [], [], []
    for i in idx:
        tasks = [ik.FrameTaskSpec(site, "site", np.ones(6), tg[i, 0], lm_damping=1.0)]
        limits = [ik.ConfigurationLimitSpec(), spec, ik.VelocityLimitSpec(np.array(hinge), np.full(len(hinge), np.pi))]
        vs.append(ik.solve_ik(model, q[i], tasks, dt, damping, limits))
        G, h = ik.limit_inequalities(ik.Configuration(model, q[i]), spec, dt)
        Gs.append(G); hs.append(h)
    return np.array(vs), np.array(Gs), np.array(hs)


def main():
    from mink_amd import _native as nat
    from mink_amd import workloads
    name = "ur5e_convex"
    B = int(sys.argv[1]) if len(sys.argv) > 1 else workloads.BENCH_CONFIGS[name]["batch"]
    model = workloads.load_bench_robot(name)
    nm = nat.NativeModel(model)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    q, tg, pt, _ = workloads.bench_batch(name, model, nm, prob, np.random.default_rng(2024), B)
    v, st = prob.solve(q, tg, pt, None, dt, damping)
    print("kernel", prob.last_kernel(), "status", np.unique(st, return_counts=True))
    _, _, taps = prob.solve(q, tg, pt, None, dt, damping, taps=["coll_h", "coll_G"], solve_qp=False)
    ncpu = min(16, os.cpu_count() or 1)
    chunks = np.array_split(np.arange(B), ncpu * 4)
    with mp.get_context("fork").Pool(ncpu) as pool:
        parts = pool.map(_chunk, [(name, c, q, tg, pt, dt, damping) for c in chunks])
    v_ref = np.concatenate([p[0] for p in parts]); G_ref = np.concatenate([p[1] for p in parts]); h_ref = np.concatenate([p[2] for p in parts])
    err = np.abs(v - v_ref).max(axis=1) / np.maximum(1.0, np.abs(v_ref).max(axis=1))
    h, G = taps["coll_h"], taps["coll_G"].reshape(B, -1, model.nv)
    fin = np.isfinite(h_ref[:, 1])
    pen = fin & (h_ref[:, 1] == 0.0)
    gerr = np.zeros(B)
    gerr[fin] = np.abs(G[fin, 1] - G_ref[fin, 1]).max(axis=1)
    for label, m in (("wall pair out of range", ~fin), ("in range, beyond d_min", fin & ~pen), ("at or inside d_min (h = 0)", pen)):
        if m.any():
            print(f"  {label:32s} {int(m.sum()):5d} instances: max rel |v - v_ref| {err[m].max():.2e}  max |G row - ref| {gerr[m].max():.2e}")
    w = np.argsort(-err)[:5]
    print("  worst:", [(int(i), f"{err[i]:.1e}", f"{gerr[i]:.1e}", float(h_ref[i, 1])) for i in w])
    print("  h max err", np.abs(h[np.isfinite(h_ref)] - h_ref[np.isfinite(h_ref)]).max())


if __name__ == "__main__":
    main()
