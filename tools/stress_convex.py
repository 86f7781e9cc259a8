#!/usr/bin/env python
"""Round 6: the general convex routine (loose GJK / expanding polytope + the witness-point polish) on many random pairs of primitives —
the device's mkh_geom_distance_eval against the numpy restatement, separated and overlapping: NaNs, the worst distance and witness-point
differences, and how many answers carry no certificate on the two sides (they must be the same pairs).  GPU + host cores.

    python tools/stress_convex.py [pairs per type, default 4096]"""
import multiprocessing as mp
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def _chunk(args):
    import test_gpu_geom_distance as T
    a, lo, hi = args
    return T._oracle(*[x[lo:hi] for x in a], 0.25)


def main():
    import test_gpu_geom_distance as T
    from mink_amd import _native as nat
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ncpu = min(16, os.cpu_count() or 1)
    for pair in T.CONVEX_PAIRS:
        for spread, label in ((0.35, "mostly separated"), (0.06, "mostly overlapping")):
            rng = np.random.default_rng(1000 * pair[0] + 10 * pair[1] + (1 if spread < 0.1 else 0))
            a = T._batch(rng, pair, n, spread)
            dist, fromto = nat.geom_distance_eval(*a, 0.25)
            cuts = np.linspace(0, n, ncpu * 2 + 1).astype(int)
            with mp.get_context("fork").Pool(ncpu) as pool:
                parts = pool.map(_chunk, [(a, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)])
            d_ref = np.concatenate([p[0] for p in parts]); ft_ref = np.concatenate([p[1] for p in parts])
            hit = d_ref != 0.25
            same_hit = ((dist != 0.25) == hit).all()
            ed = np.abs(dist - d_ref)[hit]; ep = np.abs(fromto - ft_ref).max(axis=1)[hit]
            print(f"{pair} {label:18s} n={n} in range {int(hit.sum()):5d}  NaN {int(np.isnan(dist).sum() + np.isnan(fromto).sum())}  same range {same_hit}  "
                  f"max |d dist| {ed.max():.1e}  max |d fromto| {ep.max():.1e}  beyond 1e-9: {int((ep > 1e-9).sum())}  beyond 1e-6: {int((ep > 1e-6).sum())}", flush=True)


if __name__ == "__main__":
    main()
