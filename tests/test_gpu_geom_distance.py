"""mj_geomDistance element-wise: the device's distance routines (collide_dev.h: 14 analytic pair types; convex_dev.h: GJK and,
for overlapping cores, the expanding polytope) through mkh_geom_distance_eval against the numpy restatement
(oracle/mjmath.py::mj_geomDistance, oracle/gjk.py) on random primitive pairs — separated, touching-range and OVERLAPPING.
The rows of CollisionAvoidanceLimit (collision_avoidance_limit.py:187-229) are built from exactly these outputs."""

from types import SimpleNamespace

import numpy as np
import pytest

from oracle import lie as olie
from oracle import mjmath

pytestmark = pytest.mark.gpu

PLANE, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX = 0, 2, 3, 4, 5, 6
SIZES = {
    PLANE: lambda r: np.array([1.0, 1.0, 0.1]),
    SPHERE: lambda r: np.array([r.uniform(0.03, 0.12), 0.0, 0.0]),
    CAPSULE: lambda r: np.array([r.uniform(0.02, 0.06), r.uniform(0.05, 0.2), 0.0]),
    ELLIPSOID: lambda r: r.uniform(0.04, 0.15, 3),
    CYLINDER: lambda r: np.array([r.uniform(0.03, 0.1), r.uniform(0.04, 0.15), 0.0]),
    BOX: lambda r: r.uniform(0.04, 0.15, 3),
}


def _quat2mat(q):
    m = np.empty(9)
    mjmath.mju_quat2Mat(m, q)
    return m


def _oracle(t1, s1, p1, q1, t2, s2, p2, q2, distmax):
    """mj_geomDistance of the restatement on a two-geom stand-in for (model, data)."""
    n = len(t1)
    dist, fromto = np.empty(n), np.empty((n, 6))
    for i in range(n):
        m = SimpleNamespace(geom_type=np.array([t1[i], t2[i]]), geom_size=np.array([s1[i], s2[i]]), geom_valid=np.ones(2, int))
        d = SimpleNamespace(geom_xpos=np.array([p1[i], p2[i]]), geom_xmat=np.array([_quat2mat(q1[i]), _quat2mat(q2[i])]))
        dist[i] = mjmath.mj_geomDistance(m, d, 0, 1, distmax, fromto[i])
    return dist, fromto


def _batch(rng, pair, n, spread):
    ta, tb = pair
    t1, t2 = np.full(n, ta), np.full(n, tb)
    s1 = np.array([SIZES[ta](rng) for _ in range(n)]); s2 = np.array([SIZES[tb](rng) for _ in range(n)])
    p1 = rng.uniform(-0.1, 0.1, (n, 3)); p2 = p1 + rng.uniform(-spread, spread, (n, 3))
    q1 = np.array([olie.so3_exp(rng.normal(size=3)) for _ in range(n)]); q2 = np.array([olie.so3_exp(rng.normal(size=3)) for _ in range(n)])
    return t1, s1, p1, q1, t2, s2, p2, q2


CONVEX_PAIRS = [(CYLINDER, BOX), (CYLINDER, CYLINDER), (ELLIPSOID, BOX), (ELLIPSOID, CYLINDER), (ELLIPSOID, ELLIPSOID),
                (SPHERE, ELLIPSOID), (CAPSULE, ELLIPSOID)]
ANALYTIC_PAIRS = [(SPHERE, SPHERE), (SPHERE, CAPSULE), (CAPSULE, CAPSULE), (PLANE, SPHERE), (PLANE, CAPSULE), (PLANE, BOX),
                  (PLANE, CYLINDER), (SPHERE, BOX), (SPHERE, CYLINDER), (CAPSULE, BOX), (CAPSULE, CYLINDER), (BOX, BOX),
                  (PLANE, ELLIPSOID)]


@pytest.mark.parametrize("pair", ANALYTIC_PAIRS + CONVEX_PAIRS)
def test_device_geom_distance_against_the_oracle(pair):
    from mink_amd import _native as nat
    rng = np.random.default_rng(100 * pair[0] + pair[1])
    n = 192
    args = _batch(rng, pair, n, 0.35)
    for swap in (False, True):          # both argument orders (mj_geomDistance sorts the pair by type and flips the segment back)
        a = args if not swap else args[4:] + args[:4]
        dist, fromto = nat.geom_distance_eval(*a, 0.25)
        d_ref, ft_ref = _oracle(*a, 0.25)
        hit = d_ref != 0.25
        assert ((dist != 0.25) == hit).all()
        sep = hit & (d_ref > 1e-6)
        assert sep.sum() > 10
        np.testing.assert_allclose(dist[sep], d_ref[sep], rtol=0, atol=1e-9)
        # (general convex pairs: 5e-6 before round 6 — GJK's witness points; polished onto the exact features they measure 7e-16)
        tol = 1e-9
        np.testing.assert_allclose(fromto[sep], ft_ref[sep], rtol=0, atol=tol)
        np.testing.assert_array_equal(fromto[~hit], 0.0)


@pytest.mark.parametrize("pair", CONVEX_PAIRS)
def test_overlapping_general_convex_pairs_against_the_oracle(pair):
    """Cores that overlap: depth, direction and deepest points of the smallest separating translation (expanding polytope) —
    the device's wave-cooperative routine against the sequential numpy statement of the same rules."""
    from mink_amd import _native as nat
    rng = np.random.default_rng(7 + 100 * pair[0] + pair[1])
    n = 256
    a = _batch(rng, pair, n, 0.05)
    dist, fromto = nat.geom_distance_eval(*a, 0.25)
    d_ref, ft_ref = _oracle(*a, 0.25)
    deep = d_ref < -1e-4
    assert deep.sum() > 100, deep.sum()
    flat = pair == (CYLINDER, BOX)       # every face of the Minkowski difference is flat or singly curved: the polytope converges
    err_d = np.abs(dist - d_ref)[deep]
    nrm = fromto[:, 3:] - fromto[:, :3]; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm_ref = ft_ref[:, 3:] - ft_ref[:, :3]; nrm_ref /= np.linalg.norm(nrm_ref, axis=1, keepdims=True)
    ang = np.arccos(np.clip(np.sum(nrm * nrm_ref, axis=1), -1, 1))[deep]
    err_p = np.abs(fromto - ft_ref).max(axis=1)[deep]
    print(pair, "overlapping:", int(deep.sum()), "max |d depth| %.2e, max angle %.2e, max |d fromto| %.2e; p90 angle %.2e"
          % (err_d.max(), ang.max(), err_p.max(), np.percentile(ang, 90)))
    # (before round 6: flat pairs 1e-8 / 2e-5 / 2e-5; two curved shapes — the polytope stops on its vertex budget or on the sliver
    #  guard, one iteration apart on the two sides — 1e-7 / 2e-3 / 2e-4.  With the witness points polished onto the exact features
    #  the 256 instances of every pair measure ≤ 3e-15 in depth and ≤ 1.5e-10 in the points)
    assert err_d.max() < 1e-12 and ang.max() < 1e-7 and err_p.max() < 1e-9
    # the witness points lie ON the two shapes and are the smallest separating translation apart — device side, by itself
    t1, s1, p1, q1, t2, s2, p2, q2 = a
    # (round 6: on the shapes to 1e-9 wherever the polish found a certificate — nearly every instance; where it did not — a face
    #  against a face or an edge, a stationary point that is not a strict minimum — the polytope's own witness stands, a chord off
    #  a curved patch: 1e-5 for the flat pair, 5e-3 otherwise, as before)
    exact = 0
    for i in np.flatnonzero(deep):
        R1, R2 = _quat2mat(q1[i]).reshape(3, 3), _quat2mat(q2[i]).reshape(3, 3)
        la, lb = R1.T @ (fromto[i, :3] - p1[i]), R2.T @ (fromto[i, 3:] - p2[i])
        out = max(_outside(t1[i], s1[i], la), _outside(t2[i], s2[i], lb))
        assert out < (1e-5 if flat else 5e-3), (i, la, lb)
        exact += out < 1e-9
        assert abs(np.linalg.norm(fromto[i, 3:] - fromto[i, :3]) + dist[i]) < 1e-9
    print(pair, "witness points on their shapes to 1e-9: %d of %d" % (exact, deep.sum()))
    assert exact >= 0.9 * deep.sum()


def _outside(t, s, x):
    """How far a point (geom frame) is outside a primitive (≤ 0: inside or on it)."""
    if t == SPHERE:
        return np.linalg.norm(x) - s[0]
    if t == CAPSULE:
        return np.linalg.norm(x - np.array([0, 0, np.clip(x[2], -s[1], s[1])])) - s[0]
    if t == CYLINDER:
        return max(np.hypot(x[0], x[1]) - s[0], abs(x[2]) - s[1])
    if t == BOX:
        return np.max(np.abs(x) - s)
    if t == ELLIPSOID:
        return (np.linalg.norm(x / s) - 1.0) * np.min(s)
    raise KeyError(t)
