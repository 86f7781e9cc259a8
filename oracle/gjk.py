"""TEST INFRASTRUCTURE — distance between two convex primitives by GJK on support mappings.

MuJoCo's mj_geomDistance (called at mink/limits/collision_avoidance_limit.py:214-229) sends every pair without a native
analytic routine — cylinder–box, cylinder–cylinder, ellipsoid against anything but a plane — to its general convex
collider (libccd in the pinned mujoco >= 3.1.6: MPR on the shapes inflated by half the margin each, tolerance 1e-6).  The
wheel is absent here, so this is not a restatement of libccd but of the quantity it approximates: the Euclidean distance
between the two convex sets and its witness points (Gilbert–Johnson–Keerthi 1988, closest-point sub-algorithm after
Ericson, Real-Time Collision Detection §5.1): the distance of separated shapes to ~1e-13 relative, witness points and
normal to ~1e-7 (the support-gap test |v|² − v·w ≤ 1e-14·|v|² bounds the angle of v by its square root).  Spheres and capsules enter as their
core (point / segment) plus a radius.  Overlapping shapes: the penetration depth min_|d|=1 h_{A⊖B}(d) by projected
descent on the sphere from the best of the centre-to-centre direction and the shapes' axes — a local minimum (libccd's MPR is an approximation there
too); mink only uses the sign of such a distance (h = bound_relaxation) and the direction.

tests/test_oracle_gjk.py pins the separated case against bounded minimisation over both shapes (scipy).
"""

import numpy as np

GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 0, 2, 3, 4, 5, 6, 7
MAX_ITERS = 128
POLISH = True            # witness points on the exact features after GJK / the expanding polytope (polish below); tests switch it off to look at the raw answers
GJK_PROGRESS = 1e-12     # an iteration that moves |v|² by less than this fraction ends the loop (convex_dev.h kGjkProgress)


def core_radius(gtype, size):
    """Radius of the spherical shell around the core the support mapping describes."""
    return float(np.asarray(size).reshape(-1)[0]) if gtype in (GEOM_SPHERE, GEOM_CAPSULE) else 0.0


def support_local(gtype, size, d):
    """Support point of the CORE of a primitive in its own frame: argmax_x d·x."""
    if gtype == GEOM_MESH:
        # a mesh geom is collided as its convex hull: `size` carries the hull's vertices (n, 3) in the geom frame;
        # the first maximiser of d·x
        verts = np.asarray(size, dtype=np.float64).reshape(-1, 3)
        return verts[int(np.argmax(verts @ d))].copy()
    if gtype == GEOM_SPHERE:
        return np.zeros(3)
    if gtype == GEOM_CAPSULE:
        return np.array([0.0, 0.0, size[1] if d[2] >= 0.0 else -size[1]])
    if gtype == GEOM_BOX:
        return np.array([size[i] if d[i] >= 0.0 else -size[i] for i in range(3)])
    if gtype == GEOM_CYLINDER:
        n = np.hypot(d[0], d[1])
        z = size[1] if d[2] >= 0.0 else -size[1]
        if n < 1e-300:
            return np.array([0.0, 0.0, z])
        return np.array([size[0] * d[0] / n, size[0] * d[1] / n, z])
    if gtype == GEOM_ELLIPSOID:
        e = np.array([size[0] * d[0], size[1] * d[1], size[2] * d[2]])
        n = np.sqrt(e @ e)
        if n < 1e-300:
            return np.array([size[0], 0.0, 0.0])
        return np.array([size[0] * e[0], size[1] * e[1], size[2] * e[2]]) / n
    raise NotImplementedError(gtype)


def support(gtype, size, pos, R, d):
    return pos + R @ support_local(gtype, size, R.T @ d)


def _closest_segment(P):
    a, b = P
    ab = b - a
    den = ab @ ab
    t = 0.0 if den <= 0.0 else -(a @ ab) / den
    if t <= 0.0:
        return [0], [1.0]
    if t >= 1.0:
        return [1], [1.0]
    return [0, 1], [1.0 - t, t]


def _closest_triangle(P):
    """Ericson 5.1.5 with the origin as the query point: (indices kept, barycentric weights)."""
    a, b, c = P
    ab, ac = b - a, c - a
    d1, d2 = -(ab @ a), -(ac @ a)
    if d1 <= 0.0 and d2 <= 0.0:
        return [0], [1.0]
    d3, d4 = -(ab @ b), -(ac @ b)
    if d3 >= 0.0 and d4 <= d3:
        return [1], [1.0]
    vc = d1 * d4 - d3 * d2
    if vc <= 0.0 and d1 >= 0.0 and d3 <= 0.0:
        v = d1 / (d1 - d3)
        return [0, 1], [1.0 - v, v]
    d5, d6 = -(ab @ c), -(ac @ c)
    if d6 >= 0.0 and d5 <= d6:
        return [2], [1.0]
    vb = d5 * d2 - d1 * d6
    if vb <= 0.0 and d2 >= 0.0 and d6 <= 0.0:
        w = d2 / (d2 - d6)
        return [0, 2], [1.0 - w, w]
    va = d3 * d6 - d5 * d4
    if va <= 0.0 and (d4 - d3) >= 0.0 and (d5 - d6) >= 0.0:
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        return [1, 2], [1.0 - w, w]
    den = 1.0 / (va + vb + vc)
    v, w = vb * den, vc * den
    return [0, 1, 2], [1.0 - v - w, v, w]


# (face, opposite vertex).  Vertex 3 is the one GJK has just added: the face (0, 1, 2) it stood on cannot hold a closer point
# than it held already (the loop stops on "no progress"), and the origin and vertex 3 lie on the same side of it.
_FACES = ((0, 2, 3, 1), (0, 3, 1, 2), (1, 3, 2, 0))


def _closest_tetrahedron(P):
    """Closest point of a tetrahedron to the origin; ([], []) when the origin is inside."""
    best, best_d2 = None, np.inf
    for i, j, k, o in _FACES:
        a, b, c, dv = P[i], P[j], P[k], P[o]
        n = np.cross(b - a, c - a)
        sp, sd = -(a @ n), (dv - a) @ n
        if sp * sd < 0.0 or sd == 0.0:                # the origin is on the far side of this face (or the tetrahedron is flat)
            idx, lam = _closest_triangle([a, b, c])
            glob = [(i, j, k)[t] for t in idx]
            pt = sum(l * P[g] for l, g in zip(lam, glob))
            d2 = pt @ pt
            if d2 < best_d2:
                best, best_d2 = (glob, lam), d2
    if best is None:
        return [], []
    return best


def gjk_cores(t1, s1, p1, R1, t2, s2, p2, R2, gap=1e-14, progress=None):
    """Closest points of the two cores: (distance, point on 1, point on 2, overlapping).  gap / progress: the support-gap and
    no-progress tolerances of the run (round 6: a LOOSE run — 1e-6 — in front of the witness-point polish, convex_distance)."""
    progress = GJK_PROGRESS if progress is None else progress
    d = p1 - p2
    if d @ d < 1e-30:
        d = np.array([1.0, 0.0, 0.0])
    a = support(t1, s1, p1, R1, -d)
    b = support(t2, s2, p2, R2, d)
    W, A = [a - b], [a]
    lam = lam_prev = [1.0]
    v = W[0]
    lb = 0.0
    for _ in range(MAX_ITERS):
        vv = v @ v
        scale = max(max(w @ w for w in W), 1e-300)
        if vv <= 1e-28 * scale:
            return 0.0, None, None, True
        a = support(t1, s1, p1, R1, -v)
        b = support(t2, s2, p2, R2, v)
        w = a - b
        if vv - v @ w <= gap * vv:                    # no support point is closer to the origin along v: converged
            break
        lb = max(lb, (v @ w) / np.sqrt(vv))           # every point of the difference is at least this far: a certified bound
        if any(((w - x) @ (w - x)) <= 1e-28 * scale for x in W):
            break                                     # the same vertex again (polytopes): converged
        W.append(w)
        A.append(a)
        if len(W) == 2:
            idx, lam = _closest_segment(W)
        elif len(W) == 3:
            idx, lam = _closest_triangle(W)
        else:
            idx, lam = _closest_tetrahedron(W)
            if not idx:
                if lb > 0.0:                          # "origin inside" against a certified separation: a flat tetrahedron
                    W, A, lam = W[:-1], A[:-1], lam_prev
                    break
                return 0.0, None, None, True
        Wn = [W[i] for i in idx]
        An = [A[i] for i in idx]
        v_new = sum(l * x for l, x in zip(lam, Wn))
        # no progress, or a point closer than the certified bound: a thin simplex misclassified by rounding —
        if v_new @ v_new >= vv or v_new @ v_new < lb * lb * (1.0 - 1e-10):
            W, A, lam = W[:-1], A[:-1], lam_prev      # the previous simplex is the answer
            break
        done = vv - v_new @ v_new <= progress * vv
        W, A, v = Wn, An, v_new
        lam_prev = lam
        if done:
            break
    pa = sum(l * x for l, x in zip(lam, A))
    pb = pa - v
    return float(np.sqrt(v @ v)), pa, pb, False


EPA_MAXV, EPA_MAXF, EPA_TOL = 48, 92, 1e-11        # vertices, faces (= 2·V − 4), relative gap at which a face counts as final


def _epa_face(W, i, j, k):
    """(i, j, k, unit normal, plane offset) of the triangle in its given winding; a triangle without area can never be the
    closest face (offset +inf) but stays in the list: the surface must remain closed."""
    n = np.cross(W[j] - W[i], W[k] - W[i])
    l = np.sqrt(n @ n)
    if l < 1e-150:
        return [i, j, k, np.zeros(3), np.inf]
    n = n / l
    return [i, j, k, n, float(n @ W[i])]


def penetration(t1, s1, p1, R1, r1, t2, s2, p2, R2, r2, tol=None):
    """Overlapping cores: depth, direction (from 1 to 2) and witness points (a on core 1, b on core 2) of the SMALLEST
    translation that separates them — the point of the boundary of the Minkowski difference D = {x₁ − x₂} closest to the
    origin — by the expanding polytope algorithm (van den Bergen 2001): an inner polytope of D grows by the support point in
    the direction of the face whose plane is nearest to the origin until that face is a face of D (gap ≤ EPA_TOL).  Plane
    offsets are SIGNED: the start polytope (two support points along ±x, the farthest point from their line, both sides of
    that triangle) need not contain the origin yet; every face with a negative offset is expanded first.  What MuJoCo's own
    collider returns for such a pair is an approximation of this quantity (libccd MPR, tolerance 1e-6, in the pinned
    mujoco ≥ 3.1.6; GJK + EPA since its native convex collider).  The device runs the same rules (slot order, ties) one face
    per lane: mink_amd/csrc/convex_dev.h cvx_epa."""
    W, A = [], []

    def add(d):
        a = support(t1, s1, p1, R1, d)
        W.append(a - support(t2, s2, p2, R2, -d))
        A.append(a)
        return len(W) - 1

    ex = np.array([1.0, 0.0, 0.0])
    add(ex); add(-ex)
    u = W[1] - W[0]
    uu = u @ u
    k = int(np.argmin(np.abs(u)))
    e = np.zeros(3); e[k] = 1.0
    v = e - ((e @ u) / uu) * u if uu > 0.0 else e
    v = v / np.sqrt(v @ v)
    add(v)
    if abs((W[2] - W[0]) @ v) <= 1e-12 * max(1.0, np.sqrt(uu)):      # nothing on that side of the line: the other side
        W.pop(); A.pop(); add(-v)
    n = np.cross(W[1] - W[0], W[2] - W[0])
    n = n / max(np.sqrt(n @ n), 1e-300)
    add(n); add(-n)
    c = (W[0] + W[1] + W[2] + W[3] + W[4]) / 5.0                     # an interior point: orientation of the first faces
    faces = []
    for (i, j, k) in ((0, 1, 3), (1, 2, 3), (2, 0, 3), (0, 1, 4), (1, 2, 4), (2, 0, 4)):
        f = _epa_face(W, i, j, k)
        if f[3] @ (W[i] - c) < 0.0:
            f = _epa_face(W, i, k, j)
        faces.append(f)                                              # slots 0..5; a removed face leaves its slot free (None)
    answer, intact = None, True
    while True:
        alive = [s for s, f in enumerate(faces) if f is not None]
        best = min(alive, key=lambda s: (faces[s][4], s))            # nearest plane, lowest slot on ties
        i, j, k, nb, off = faces[best]
        # the nearest plane can only move outwards as the polytope grows; when it jumps back in, a sliver face (three nearly
        # collinear support points near convergence: a normal without digits) has been created — the previous face stands
        if answer is not None and answer[4] >= 0.0 and off < answer[4] - 1e-9 * max(1.0, answer[4]):
            intact = False
            break
        answer = (i, j, k, nb, off)
        if len(W) >= EPA_MAXV:
            break
        ip = add(nb)
        gap = nb @ W[ip] - off
        if gap <= (EPA_TOL if tol is None else tol) * max(1.0, abs(off)):   # the face is (within the gap) a face of D itself
            W.pop(); A.pop()
            break
        p = W[ip]
        vis = [s for s in alive if faces[s][3] @ p - faces[s][4] > 1e-13 * max(1.0, abs(faces[s][4])) or s == best]
        visset = set(vis)
        owner = {}
        for s in alive:
            f = faces[s]
            for e in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
                owner[e] = s in visset
        horizon = [e for s in vis for e in ((faces[s][0], faces[s][1]), (faces[s][1], faces[s][2]), (faces[s][2], faces[s][0]))
                   if not owner.get((e[1], e[0]), False)]            # in (slot, edge) order
        if len(faces) - len(vis) + len(horizon) > EPA_MAXF:          # out of slots: the nearest face so far is the answer
            W.pop(); A.pop()
            break
        free = list(vis) + list(range(len(faces), len(faces) + max(0, len(horizon) - len(vis))))
        for s in vis:
            faces[s] = None
        for q, (a_, b_) in enumerate(horizon):                       # new face q takes the q-th free slot
            while len(faces) <= free[q]:
                faces.append(None)
            faces[free[q]] = _epa_face(W, a_, b_, ip)
    if intact:
        # a face of D is usually covered by several coplanar triangles: of those in the answer's plane (offsets within 1e-9)
        # the one NEAREST to the origin as a triangle — the one that holds the foot of the perpendicular — gives the witness
        # points (a clamped foot on a neighbouring triangle would put them off the shapes)
        thr = answer[4] + 1e-9 * max(1.0, abs(answer[4]))
        cand = []
        for s, f in enumerate(faces):
            if f is None or not f[4] <= thr:
                continue
            idx, lam = _closest_triangle([W[f[0]], W[f[1]], W[f[2]]])
            w = sum(l * W[(f[0], f[1], f[2])[t]] for t, l in zip(idx, lam))
            cand.append((float(w @ w), s))
        answer = faces[min(cand)[1]]
    i, j, k, nb, off = answer
    idx, lam = _closest_triangle([W[i], W[j], W[k]])                 # the face's point nearest to the origin, barycentric
    g = (i, j, k)
    a = sum(l * A[g[t]] for t, l in zip(idx, lam))
    w = sum(l * W[g[t]] for t, l in zip(idx, lam))
    # (geom 2 translated by t overlaps geom 1 iff t ∈ D: the shortest separating translation of geom 2 is depth·n, n the
    #  outward normal of the nearest face — mj_geomDistance's direction "from 1 to 2")
    depth = float(nb @ support(t1, s1, p1, R1, nb) - nb @ support(t2, s2, p2, R2, -nb))
    return depth + r1 + r2, nb, a, a - w


# ---------------------------------------------------------------------------------------------------------------------
# Witness points on the exact features (round 6).  GJK and the expanding polytope approximate a curved rim by chords: the
# distance converges to ~1e-13, the direction and the witness points only to ~1e-6 (thin simplices), and a row of G inherits
# that.  Both answers are minimisers of the support function of the difference D = A ⊖ B over unit directions,
#     h_D(n) = h_A(n) + h_B(−n),   signed distance = −min h_D   (n from shape 1 to shape 2; a local minimum when the cores overlap),
# with witnesses a ∈ F_A(n), b ∈ F_B(−n) (the support SETS) and a − b = h_D(n)·n.  Given the approximate direction n0:
#   * h_D is smooth except on KINKS: the planes n·k = 0 (k a box axis, the axis of a cylinder or of a capsule's segment) and the
#     poles n = ±u of a cylinder.  The kinks within KINK_TOL of n0 are candidates to hold the minimiser exactly;
#   * on a set of active kinks the direction is either determined (a pole; two planes: n = ±k₁ × k₂) or found by Newton in the
#     remaining tangent space T with the reduced Hessian Tᵀ(∇²h_A(n) + ∇²h_B(−n))T − h_D(n)·I (the second term is the curvature
#     of the unit sphere under a 1-homogeneous function; ∇²h of a cylinder is (r/ρ)·t·tᵀ along the rim, of an ellipsoid
#     (S² − (S²d)(S²d)ᵀ/h²)/h, zero for polytopes, segments and points);
#   * a candidate is ACCEPTED only with an exact certificate: witnesses built from the support sets (a unique support point of
#     one shape fixes the other's witness through a − b = h_D·n; a segment against a segment is a 2 × 2 solve) must lie in their
#     shapes to POLISH_TOL.  For separated shapes that is the optimality condition of a convex problem, so a wrongly snapped kink
#     cannot pass.  No certificate (mesh hulls: their kinks are not enumerated; a face against an edge or a face: the witness is
#     not unique) → the GJK / expanding-polytope answer stands.
# mink_amd/csrc/convex_dev.h cvx_polish is the device statement.
KINK_TOL, POLISH_TOL = 1e-3, 1e-10


def _kinks(gtype, R):
    """Unit normals k of the planes n·k = 0 on which the support point of the shape jumps."""
    if gtype == GEOM_BOX:
        return [R[:, 0], R[:, 1], R[:, 2]]
    if gtype in (GEOM_CYLINDER, GEOM_CAPSULE):
        return [R[:, 2]]
    return []


def _support_hessian(gtype, size, R, d):
    """∇²h(d) of the core's support function at the unit direction d (world frame)."""
    if gtype == GEOM_CYLINDER:
        u = R[:, 2]
        m = d - (d @ u) * u
        rho = np.sqrt(m @ m)
        if rho < 1e-12:
            return None                                # (a pole: not smooth)
        t = np.cross(u, m / rho)
        return (size[0] / rho) * np.outer(t, t)
    if gtype == GEOM_ELLIPSOID:
        dl = R.T @ d
        s2 = np.asarray(size[:3]) ** 2
        h = np.sqrt(np.sum(s2 * dl * dl))
        e = s2 * dl
        return R @ ((np.diag(s2) - np.outer(e, e) / (h * h)) / h) @ R.T
    return np.zeros((3, 3))


def _contains(gtype, size, p, R, x, tol):
    """x in the CORE of the shape, to the absolute tolerance tol."""
    l = R.T @ (x - p)
    if gtype == GEOM_BOX:
        return bool(np.all(np.abs(l) <= np.asarray(size[:3]) + tol))
    if gtype == GEOM_CYLINDER:
        return bool(np.hypot(l[0], l[1]) <= size[0] + tol and abs(l[2]) <= size[1] + tol)
    if gtype == GEOM_ELLIPSOID:
        return bool(np.sum((l / np.asarray(size[:3])) ** 2) <= 1.0 + 2.0 * tol / min(size[:3]))
    if gtype == GEOM_CAPSULE:
        return bool(np.hypot(l[0], l[1]) <= tol and abs(l[2]) <= size[1] + tol)
    if gtype == GEOM_SPHERE:
        return bool(np.sqrt(l @ l) <= tol)
    return False


def _support_set(gtype, size, p, R, d, active, pole):
    """The support set of the core along d: ('point', x) | ('segment', x0, x1) | ('face',) — `active`: the shape's kink planes d
    lies on, `pole`: d is ± the axis of a cylinder."""
    if gtype == GEOM_CYLINDER and pole:
        return ("face",)
    if not active:
        return ("point", support(gtype, size, p, R, d))
    if len(active) > 1:
        return ("face",)
    k = active[0]
    # the two ends: support points a hair to either side of the plane
    x0, x1 = support(gtype, size, p, R, d - 1e-6 * k), support(gtype, size, p, R, d + 1e-6 * k)
    if gtype == GEOM_CYLINDER:                         # (the generator of d's radial direction, exactly)
        c = support(gtype, size, p, R, d)
        mid = c - ((c - p) @ k) * k
        x0, x1 = mid - size[1] * k, mid + size[1] * k
    return ("segment", x0, x1)


def polish(t1, s1, p1, R1, t2, s2, p2, R2, n0):
    """(signed core distance, a, b, n) with the witnesses on the exact features, or None without a certificate (see above)."""
    if GEOM_MESH in (t1, t2):
        return None
    scale = max(float(np.max(np.abs(np.asarray(s1, dtype=np.float64).reshape(-1)[:3]))), float(np.max(np.abs(np.asarray(s2, dtype=np.float64).reshape(-1)[:3]))))
    tol = POLISH_TOL * max(scale, 1e-3)
    n0 = n0 / np.sqrt(n0 @ n0)
    kinks = [(1, k) for k in _kinks(t1, R1)] + [(2, k) for k in _kinks(t2, R2)]
    near = [(w, k) for (w, k) in kinks if abs(n0 @ k) < KINK_TOL]
    poles = []
    if t1 == GEOM_CYLINDER and np.linalg.norm(np.cross(n0, R1[:, 2])) < KINK_TOL:
        poles.append((1, R1[:, 2] * (1.0 if n0 @ R1[:, 2] >= 0.0 else -1.0)))
    if t2 == GEOM_CYLINDER and np.linalg.norm(np.cross(n0, R2[:, 2])) < KINK_TOL:
        poles.append((2, R2[:, 2] * (1.0 if n0 @ R2[:, 2] >= 0.0 else -1.0)))

    def h_of(n):
        a, b = support(t1, s1, p1, R1, n), support(t2, s2, p2, R2, -n)
        return a, b, float(n @ (a - b))

    def certify(n, act, pole_of):
        a_s, b_s, h = h_of(n)
        A = _support_set(t1, s1, p1, R1, n, [k for (w, k) in act if w == 1], pole_of == 1)
        Bs = _support_set(t2, s2, p2, R2, -n, [k for (w, k) in act if w == 2], pole_of == 2)
        if A[0] == "point" and Bs[0] == "point":
            a, b = A[1], Bs[1]
            if np.linalg.norm((a - b) - h * n) > tol:
                return None
            b = a - h * n
        elif A[0] == "point":
            a = A[1]
            b = a - h * n
            if not _contains(t2, s2, p2, R2, b, tol):
                return None
        elif Bs[0] == "point":
            b = Bs[1]
            a = b + h * n
            if not _contains(t1, s1, p1, R1, a, tol):
                return None
        elif A[0] == "segment" and Bs[0] == "segment":
            ea, eb = A[2] - A[1], Bs[2] - Bs[1]
            M = np.array([[ea @ ea, -(ea @ eb)], [-(ea @ eb), eb @ eb]])
            if abs(np.linalg.det(M)) < 1e-12 * (ea @ ea) * (eb @ eb):
                return None                            # parallel: the witness is not unique
            r = Bs[1] + h * n - A[1]
            al, be = np.linalg.solve(M, np.array([ea @ r, -(eb @ r)]))
            a, b = A[1] + al * ea, Bs[1] + be * eb
            if not (-1e-9 <= al <= 1.0 + 1e-9 and -1e-9 <= be <= 1.0 + 1e-9) or np.linalg.norm((a - b) - h * n) > tol:
                return None
            b = a - h * n
        else:
            return None
        return -h, a, b, n

    def newton(n, act):
        ks = [k for (_, k) in act]
        for k in ks:
            n = n - (n @ k) * k
        n = n / np.sqrt(n @ n)
        for _ in range(8):
            a, b, h = h_of(n)
            if ks:
                T = np.cross(n, ks[0])[:, None]
                T = T / np.linalg.norm(T)
            else:
                e = np.zeros(3); e[int(np.argmin(np.abs(n)))] = 1.0
                t1_ = np.cross(n, e); t1_ /= np.linalg.norm(t1_)
                T = np.stack([t1_, np.cross(n, t1_)], axis=1)
            HA, HB = _support_hessian(t1, s1, R1, n), _support_hessian(t2, s2, R2, -n)
            if HA is None or HB is None:
                return None
            M = T.T @ (HA + HB) @ T - h * np.eye(T.shape[1])
            if np.any(np.linalg.eigvalsh(M) <= 1e-12 * max(1.0, abs(h))):
                return None                            # not a (strict) local minimum of the smooth piece
            delta = T @ np.linalg.solve(M, -(T.T @ (a - b)))
            n = n + delta
            for k in ks:
                n = n - (n @ k) * k
            n = n / np.sqrt(n @ n)
            if np.linalg.norm(delta) < 1e-9:           # (quadratic convergence: the error left is below 1e-18)
                break
        else:
            return None                                # no convergence in eight steps
        return n

    # candidates in the device's order: the poles, the first two near kink planes together (slot order: the axes of shape 1, then
    # those of shape 2; a third near plane is a degenerate pose and is ignored), each of them alone, none
    near = near[:2]
    cands = []
    for (w, u) in poles:
        cands.append((u, [], w))
    if len(near) == 2:
        c = np.cross(near[0][1], near[1][1])
        if np.linalg.norm(c) > 1e-6:
            c = c / np.linalg.norm(c)
            cands.append((c * (1.0 if c @ n0 >= 0.0 else -1.0), [near[0], near[1]], 0))
    for nk in near:
        cands.append((None, [nk], 0))
    cands.append((None, [], 0))
    for n, act, pole_of in cands:
        if n is None:
            n = newton(n0.copy(), act)
            if n is None:
                continue
        r = certify(n, act, pole_of)
        if r is not None:
            return r
    return None


LOOSE_GAP, LOOSE_EPA = 1e-6, 1e-6   # tolerances of the runs in front of the polish (convex_dev.h kLooseGap / kLooseEpa)


def convex_distance(t1, s1, p1, R1, t2, s2, p2, R2, margin):
    """One contact (dist, pos, n) in mj_geomDistance's convention — n from geom 1 to geom 2, pos the midpoint of the
    witness points — or None beyond `margin`.
    Round 6: GJK and the expanding polytope first run LOOSE (support gap 1e-6 instead of 1e-14 / 1e-11: half the support
    evaluations) — the polish finishes their answer exactly, and its certificate says so; without one (mesh hulls, non-unique
    witnesses: a few per cent of overlapping pairs) the tight run follows, and its answer stands if the polish has none for it either."""
    r1, r2 = core_radius(t1, s1), core_radius(t2, s2)
    loose = POLISH and GEOM_MESH not in (t1, t2)
    for tight in ((False, True) if loose else (True,)):
        dist_c, pa, pb, overlap = gjk_cores(t1, s1, p1, R1, t2, s2, p2, R2) if tight else \
            gjk_cores(t1, s1, p1, R1, t2, s2, p2, R2, LOOSE_GAP, LOOSE_GAP)
        if overlap or not dist_c > 1e-9:
            break
        # cores apart (also when only the spherical shells overlap)
        if dist_c - r1 - r2 > margin * (1.0 + 1e-3) + 1e-6:           # (beyond the margin by more than a loose run can be off)
            return None
        n = (pb - pa) / dist_c
        pol = polish(t1, s1, p1, R1, t2, s2, p2, R2, n) if POLISH else None
        if pol is not None and pol[0] > 0.0 and abs(pol[0] - dist_c) <= (1e-6 if tight else 1e-3) * max(dist_c, 1e-3):
            dist_c, pa, pb, n = pol
        elif not tight:
            continue                                                  # no certificate for the loose answer: the tight run
        dist = dist_c - r1 - r2
        if dist > margin:
            return None
        a, b = pa + r1 * n, pb - r2 * n
        return dist, 0.5 * (a + b), n
    for tight in ((False, True) if loose else (True,)):
        depth, n, a, b = penetration(t1, s1, p1, R1, r1, t2, s2, p2, R2, r2, None if tight else LOOSE_EPA)
        pol = polish(t1, s1, p1, R1, t2, s2, p2, R2, n) if POLISH else None
        # (overlapping cores: a certified stationary point in the polytope's basin — a direction within 0.14 rad of its answer and a depth
        #  no larger than its own, which stops on a vertex budget a few 1e-4 above the minimum for doubly curved pairs, and within 1 %)
        dc_, ds_ = depth - r1 - r2, max(depth - r1 - r2, 1e-3)
        if pol is not None and pol[3] @ n >= 0.99 and -pol[0] <= dc_ + (1e-9 if tight else 1e-5) * ds_ and -pol[0] >= dc_ - 1e-2 * ds_:
            depth, a, b, n = -pol[0] + r1 + r2, pol[1], pol[2], pol[3]
            break
    a, b = a + r1 * n, b - r2 * n                     # the deepest points: a − b = depth·n
    return -depth, 0.5 * (a + b), n
