"""oracle/gjk.py (general convex pairs of mj_geomDistance: cylinder–box, cylinder–cylinder, ellipsoid–*) pinned against
bounded minimisation of |x − y| over both shapes (scipy SLSQP; the problem is convex), and against the analytic routines
of oracle/mjmath.py on the pairs both know."""

import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import gjk
from oracle import mjmath as mj


def _rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    R = np.zeros(9)
    mj.mju_quat2Mat(R, q)
    return R.reshape(3, 3)


def _inside(gtype, size):
    """g(x_local) >= 0 inside the (full) shape."""
    if gtype == gjk.GEOM_BOX:
        return lambda x: np.concatenate([size[:3] - x, size[:3] + x])
    if gtype == gjk.GEOM_CYLINDER:
        return lambda x: np.array([size[0] ** 2 - x[0] ** 2 - x[1] ** 2, size[1] - x[2], size[1] + x[2]])
    if gtype == gjk.GEOM_ELLIPSOID:
        return lambda x: np.array([1.0 - np.sum((x / size[:3]) ** 2)])
    if gtype == gjk.GEOM_SPHERE:
        return lambda x: np.array([size[0] ** 2 - x @ x])
    if gtype == gjk.GEOM_CAPSULE:
        def g(x):
            z = np.clip(x[2], -size[1], size[1])
            return np.array([size[0] ** 2 - x[0] ** 2 - x[1] ** 2 - (x[2] - z) ** 2])
        return g
    raise KeyError(gtype)


def _brute(t1, s1, p1, R1, t2, s2, p2, R2, rng):
    g1, g2 = _inside(t1, s1), _inside(t2, s2)
    cons = [{"type": "ineq", "fun": lambda z: g1(R1.T @ (z[:3] - p1))},
            {"type": "ineq", "fun": lambda z: g2(R2.T @ (z[3:] - p2))}]
    best = np.inf
    for _ in range(4):
        z0 = np.concatenate([p1 + 0.01 * rng.normal(size=3), p2 + 0.01 * rng.normal(size=3)])
        r = minimize(lambda z: (z[:3] - z[3:]) @ (z[:3] - z[3:]), z0, constraints=cons, method="SLSQP",
                     options={"ftol": 1e-15, "maxiter": 400})
        if r.success or r.status == 9:
            best = min(best, np.sqrt(max(r.fun, 0.0)))
    return best


SIZES = {gjk.GEOM_BOX: lambda rng: rng.uniform(0.05, 0.3, 3), gjk.GEOM_CYLINDER: lambda rng: np.append(rng.uniform(0.05, 0.25, 2), 0.0),
         gjk.GEOM_ELLIPSOID: lambda rng: rng.uniform(0.05, 0.3, 3), gjk.GEOM_SPHERE: lambda rng: np.append(rng.uniform(0.05, 0.2, 1), [0.0, 0.0]),
         gjk.GEOM_CAPSULE: lambda rng: np.append(rng.uniform(0.03, 0.2, 2), 0.0)}
NEW_PAIRS = [(gjk.GEOM_CYLINDER, gjk.GEOM_BOX), (gjk.GEOM_CYLINDER, gjk.GEOM_CYLINDER), (gjk.GEOM_ELLIPSOID, gjk.GEOM_BOX),
             (gjk.GEOM_ELLIPSOID, gjk.GEOM_CYLINDER), (gjk.GEOM_ELLIPSOID, gjk.GEOM_ELLIPSOID), (gjk.GEOM_SPHERE, gjk.GEOM_ELLIPSOID),
             (gjk.GEOM_CAPSULE, gjk.GEOM_ELLIPSOID)]


@pytest.mark.parametrize("pair", NEW_PAIRS)
def test_separated_pairs_are_the_euclidean_distance(pair):
    rng = np.random.default_rng(hash(pair) % 1000)
    t1, t2 = pair
    n = 0
    while n < 12:
        s1, s2 = SIZES[t1](rng), SIZES[t2](rng)
        p1, p2 = rng.uniform(-0.4, 0.4, 3), rng.uniform(-0.4, 0.4, 3)
        R1, R2 = _rand_rot(rng), _rand_rot(rng)
        c = gjk.convex_distance(t1, s1, p1, R1, t2, s2, p2, R2, 10.0)
        if c is None or c[0] <= 1e-3:
            continue
        n += 1
        dist, pos, nrm = c
        ref = _brute(t1, s1, p1, R1, t2, s2, p2, R2, rng)
        assert abs(dist - ref) < 2e-6 * max(1.0, ref), (pair, dist, ref)
        # witness points lie on the two surfaces and realise the distance
        a, b = pos - 0.5 * dist * nrm, pos + 0.5 * dist * nrm
        assert abs(np.linalg.norm(nrm) - 1.0) < 1e-9
        assert min(_inside(t1, s1)(R1.T @ (a - p1))) > -1e-8 and min(_inside(t2, s2)(R2.T @ (b - p2))) > -1e-8
        # optimality: the separating direction supports both shapes at the witness points
        assert abs(nrm @ (gjk.support(t1, s1, p1, R1, nrm) + gjk.core_radius(t1, s1) * nrm - a)) < 1e-9
        assert abs(nrm @ (gjk.support(t2, s2, p2, R2, -nrm) - gjk.core_radius(t2, s2) * nrm - b)) < 1e-9


def test_agrees_with_the_analytic_routines_where_both_exist():
    rng = np.random.default_rng(5)
    for _ in range(40):
        R1, R2 = _rand_rot(rng), _rand_rot(rng)
        p1, p2 = rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.3, 0.3, 3) + np.array([0.6, 0.0, 0.0])
        sb1, sb2 = rng.uniform(0.05, 0.2, 3), rng.uniform(0.05, 0.2, 3)
        cons = mj._box_box(p1, R1.reshape(-1), sb1, p2, R2.reshape(-1), sb2, 10.0)
        d_ref = min(c[0] for c in cons)
        d, _, _ = gjk.convex_distance(gjk.GEOM_BOX, sb1, p1, R1, gjk.GEOM_BOX, sb2, p2, R2, 10.0)
        if d_ref > 0.0:
            assert abs(d - d_ref) < 1e-10
        else:                                         # overlapping: a local minimum of the separating translation
            assert d < 0.0 and -d >= -d_ref - 1e-9
        sc, scy = np.array([0.05, 0.15, 0.0]), np.array([0.1, 0.12, 0.0])
        cons = mj._capsule_cylinder(p1, R1.reshape(-1), sc, p2, R2.reshape(-1), scy, 10.0)
        d, _, _ = gjk.convex_distance(gjk.GEOM_CAPSULE, sc, p1, R1, gjk.GEOM_CYLINDER, scy, p2, R2, 10.0)
        if min(c[0] for c in cons) > 0.0:
            assert abs(d - min(c[0] for c in cons)) < 1e-9


def test_overlapping_pairs_report_a_negative_distance_that_separates():
    rng = np.random.default_rng(11)
    for t1, t2 in NEW_PAIRS:
        for _ in range(6):
            s1, s2 = SIZES[t1](rng), SIZES[t2](rng)
            p1 = rng.uniform(-0.1, 0.1, 3)
            p2 = p1 + rng.uniform(-0.04, 0.04, 3)
            R1, R2 = _rand_rot(rng), _rand_rot(rng)
            dist, pos, n = gjk.convex_distance(t1, s1, p1, R1, t2, s2, p2, R2, 10.0)
            assert dist < 0.0
            # translating geom 2 by (−dist + ε)·n separates the shapes
            c = gjk.convex_distance(t1, s1, p1, R1, t2, s2, p2 + (-dist + 1e-6) * n, R2, 10.0)
            assert c is not None and c[0] > 0.0        # (an upper bound of the depth: the descent finds a local minimum)
