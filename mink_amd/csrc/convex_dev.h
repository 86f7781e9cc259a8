// Distance between two convex shapes by GJK on support mappings: the pairs of mj_geomDistance
// (mink/limits/collision_avoidance_limit.py:214-229) that have no native analytic routine in MuJoCo either and go to its
// general convex collider — cylinder–box, cylinder–cylinder, ellipsoid against anything but a plane, and every pair with a
// MESH geom (MuJoCo collides the convex hull of a mesh; here: the hull's vertices as the support mapping).
//
// What is computed is the Euclidean distance of the two convex sets with its witness points (Gilbert–Johnson–Keerthi
// 1988; closest point of a simplex after Ericson, Real-Time Collision Detection §5.1): distance to ~1e-13 relative,
// witness points / normal to ~1e-7 (the support-gap test |v|² − v·w ≤ 1e-14·|v|² bounds the angle of v by its square
// root).  MuJoCo's own answer for these pairs (libccd MPR on shapes inflated by half the margin, tolerance 1e-6) is an
// approximation of the same quantity.  Spheres and capsules enter as their core (point / segment) plus a radius.
// Overlapping shapes: an upper bound of the penetration depth, min over unit d of h₁(d) + h₂(−d) by projected descent from
// the best of the centre-to-centre direction and the shapes' axes (mink only uses the sign of such a distance —
// h = bound_relaxation — and the direction).  oracle/gjk.py is the CPU statement of the same algorithm.
//
// Register discipline (round 3).  The first version kept the simplex in arrays indexed by the sub-algorithm's results
// (`W[idx[i]]`, `P[faces[f][li[t]]]`): hipcc sends every runtime-indexed private array to SCRATCH, so a GJK run was a
// chain of scratch round trips (528–1 120 B of scratch per lane, 4 096 UR5e problems with ONE cylinder–box pair took
// 0.39 ms against 0.036 ms without it).  Here every simplex vertex is a named value: the sub-algorithms return barycentric
// weights on FIXED slots plus a keep mask, appending and compacting the simplex are select networks, the faces of the
// tetrahedron are four static instantiations.  Nothing is indexed at run time; the routine compiles without scratch.
#pragma once
#include "lie_dev.h"

namespace mkh {

constexpr int kGeomSphere = 2, kGeomCapsule = 3, kGeomEllipsoid = 4, kGeomCylinder = 5, kGeomBox = 6, kGeomMesh = 7;
constexpr int kGjkMaxIters = 128;
// Overlapping shapes (cvx_penetration): mink uses the SIGN of such a distance (h = bound_relaxation) and the direction of the
// row; a projected-gradient descent converges linearly, so every digit of the direction costs iterations — and a launch
// waits for its slowest wavefront: with the first version's 128 × 20 support evaluations at 1e-12 a UR5e batch with one
// cylinder–box pair spent 0.19 of its 0.25 ms on the 3 % of instances that start inside the wall.  1e-7 on the gradient
// (the direction to ~1e-7 rad), 40 steps of at most 10 halvings.
constexpr int kPenMaxIters = 40, kPenMaxHalvings = 10;
constexpr double kPenTol = 1e-7;

// vert / nvert: convex-hull vertices of a mesh geom in the geom's frame (global memory, 3 doubles each); nullptr else
struct ConvexGeom { int type; V3 size; V3 pos; M3 R; const double* vert; int nvert; };

__device__ __forceinline__ double cvx_core_radius(const ConvexGeom& g) {
  return (g.type == kGeomSphere || g.type == kGeomCapsule) ? g.size.x : 0.0;
}

// support point of the CORE in the world: argmax_x d·x
__device__ __forceinline__ V3 cvx_support(const ConvexGeom& g, V3 d) {
  const V3 dl = mulT(g.R, d);
  V3 s{0.0, 0.0, 0.0};
  if (g.type == kGeomCapsule) {
    s.z = dl.z >= 0.0 ? g.size.y : -g.size.y;
  } else if (g.type == kGeomBox) {
    s = {dl.x >= 0.0 ? g.size.x : -g.size.x, dl.y >= 0.0 ? g.size.y : -g.size.y, dl.z >= 0.0 ? g.size.z : -g.size.z};
  } else if (g.type == kGeomCylinder) {
    const double n = sqrt(dl.x * dl.x + dl.y * dl.y);
    s.z = dl.z >= 0.0 ? g.size.y : -g.size.y;
    if (n >= 1e-300) { const double k = g.size.x * fast_rcp(n); s.x = k * dl.x; s.y = k * dl.y; }
  } else if (g.type == kGeomEllipsoid) {
    const V3 e{g.size.x * dl.x, g.size.y * dl.y, g.size.z * dl.z};
    const double n = sqrt(dot(e, e));
    if (n < 1e-300) s = {g.size.x, 0.0, 0.0};
    else { const double k = fast_rcp(n); s = {g.size.x * e.x * k, g.size.y * e.y * k, g.size.z * e.z * k}; }
  } else if (g.type == kGeomMesh) {
    // hull vertices: the first maximiser of d·x (numpy argmax order, oracle/gjk.py).  Four vertices per trip: the loads of
    // a trip are independent, the running maximum is the only chain.
    double best = -__builtin_huge_val();
    const double* v = g.vert;
    int i = 0;
    for (; i + 4 <= g.nvert; i += 4) {
      const double* p = v + 3 * i;
      const double x0 = p[0], y0 = p[1], z0 = p[2], x1 = p[3], y1 = p[4], z1 = p[5];
      const double x2 = p[6], y2 = p[7], z2 = p[8], x3 = p[9], y3 = p[10], z3 = p[11];
      const double d0 = dl.x * x0 + dl.y * y0 + dl.z * z0, d1 = dl.x * x1 + dl.y * y1 + dl.z * z1;
      const double d2 = dl.x * x2 + dl.y * y2 + dl.z * z2, d3 = dl.x * x3 + dl.y * y3 + dl.z * z3;
      if (d0 > best) { best = d0; s = {x0, y0, z0}; }
      if (d1 > best) { best = d1; s = {x1, y1, z1}; }
      if (d2 > best) { best = d2; s = {x2, y2, z2}; }
      if (d3 > best) { best = d3; s = {x3, y3, z3}; }
    }
    for (; i < g.nvert; ++i) {
      const double* p = v + 3 * i;
      const double x0 = p[0], y0 = p[1], z0 = p[2];
      const double d0 = dl.x * x0 + dl.y * y0 + dl.z * z0;
      if (d0 > best) { best = d0; s = {x0, y0, z0}; }
    }
  }
  return g.pos + mul(g.R, s);
}

// Closest point of a simplex to the origin as barycentric weights on the simplex's own vertices + the vertices kept
// (bit i of keep).  keep = 0 from the tetrahedron: the origin is inside.
struct CvxW2 { double l0, l1; int keep; };
struct CvxW3 { double l0, l1, l2; int keep; };
struct CvxW4 { double l0, l1, l2, l3; int keep; };

__device__ __forceinline__ CvxW2 cvx_closest_segment(V3 a, V3 b) {
  const V3 ab = b - a;
  const double den = dot(ab, ab);
  const double t = den <= 0.0 ? 0.0 : -dot(a, ab) / den;
  if (t <= 0.0) return {1.0, 0.0, 1};
  if (t >= 1.0) return {0.0, 1.0, 2};
  return {1.0 - t, t, 3};
}

__device__ __forceinline__ CvxW3 cvx_closest_triangle(V3 a, V3 b, V3 c) {
  const V3 ab = b - a, ac = c - a;
  const double d1 = -dot(ab, a), d2 = -dot(ac, a);
  if (d1 <= 0.0 && d2 <= 0.0) return {1.0, 0.0, 0.0, 1};
  const double d3 = -dot(ab, b), d4 = -dot(ac, b);
  if (d3 >= 0.0 && d4 <= d3) return {0.0, 1.0, 0.0, 2};
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    const double v = d1 / (d1 - d3);
    return {1.0 - v, v, 0.0, 3};
  }
  const double d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6) return {0.0, 0.0, 1.0, 4};
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    const double w = d2 / (d2 - d6);
    return {1.0 - w, 0.0, w, 5};
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    return {0.0, 1.0 - w, w, 6};
  }
  const double den = 1.0 / (va + vb + vc);
  const double v = vb * den, w = vc * den;
  return {1.0 - v - w, v, w, 7};
}

// one face (a, b, c) of the tetrahedron, opposite vertex dv; SA / SB / SC: the tetrahedron slots of a, b, c
template <int SA, int SB, int SC>
__device__ __forceinline__ void cvx_tet_face(V3 a, V3 b, V3 c, V3 dv, double& best_d2, CvxW4& best) {
  const V3 n = cross(b - a, c - a);
  const double sp = -dot(a, n), sd = dot(dv - a, n);
  if (sp * sd < 0.0 || sd == 0.0) {               // the origin is on the far side of this face (or the tetrahedron is flat)
    const CvxW3 t = cvx_closest_triangle(a, b, c);
    const V3 pt = t.l0 * a + (t.l1 * b + t.l2 * c);
    const double d2 = dot(pt, pt);
    if (d2 < best_d2) {
      best_d2 = d2;
      auto slot = [&](int s) -> double { return (SA == s) ? t.l0 : ((SB == s) ? t.l1 : ((SC == s) ? t.l2 : 0.0)); };
      auto bit = [&](int s) -> int {
        return ((SA == s) ? (t.keep & 1) : ((SB == s) ? ((t.keep >> 1) & 1) : ((SC == s) ? ((t.keep >> 2) & 1) : 0))) << s;
      };
      best = {slot(0), slot(1), slot(2), slot(3), bit(0) | bit(1) | bit(2) | bit(3)};
    }
  }
}

__device__ __forceinline__ CvxW4 cvx_closest_tetrahedron(V3 p0, V3 p1, V3 p2, V3 p3) {
  CvxW4 best{0.0, 0.0, 0.0, 0.0, 0};
  double best_d2 = __builtin_huge_val();
  cvx_tet_face<0, 1, 2>(p0, p1, p2, p3, best_d2, best);       // (face, opposite vertex) as in oracle/gjk.py _FACES
  cvx_tet_face<0, 2, 3>(p0, p2, p3, p1, best_d2, best);
  cvx_tet_face<0, 3, 1>(p0, p3, p1, p2, best_d2, best);
  cvx_tet_face<1, 3, 2>(p1, p3, p2, p0, best_d2, best);
  return best;
}

// Closest points of the two cores: distance, point on 1, point on 2; returns false when the cores overlap.
// `cutoff`: the caller discards the pair when the cores are farther apart than this — the loop stops as soon as its
// certified lower bound says so (dist = that bound, no witness points): two or three iterations instead of ~9 for the
// pairs of a batch that are nowhere near each other.
__device__ __forceinline__ bool cvx_gjk(const ConvexGeom& g1, const ConvexGeom& g2, const double cutoff, double& dist, V3& pa, V3& pb) {
  V3 d = g1.pos - g2.pos;
  if (dot(d, d) < 1e-30) d = {1.0, 0.0, 0.0};
  const V3 zero{0.0, 0.0, 0.0};
  V3 W0, W1 = zero, W2 = zero, W3 = zero, A0, A1 = zero, A2 = zero, A3 = zero;     // simplex of the difference / points on shape 1
  double lam0 = 1.0, lam1 = 0.0, lam2 = 0.0;       // weights of the last accepted simplex (n ≤ 3 vertices)
  int n = 1;
  A0 = cvx_support(g1, -1.0 * d);
  W0 = A0 - cvx_support(g2, d);
  V3 v = W0;
  double lb = 0.0;
#pragma nounroll
  for (int it = 0; it < kGjkMaxIters; ++it) {
    const double vv = dot(v, v);
    double scale = fmax(1e-300, dot(W0, W0));
    if (n > 1) scale = fmax(scale, dot(W1, W1));
    if (n > 2) scale = fmax(scale, dot(W2, W2));
    if (vv <= 1e-28 * scale) return false;
    const V3 a = cvx_support(g1, -1.0 * v);
    const V3 w = a - cvx_support(g2, v);
    const double vw = dot(v, w);
    if (vv - vw <= 1e-14 * vv) break;             // no support point is closer to the origin along v: converged
    lb = fmax(lb, vw / sqrt(vv));                 // every point of the difference is at least this far: a certified bound
    if (lb > cutoff) { dist = lb; return true; }
    const double tol = 1e-28 * scale;
    bool same = dot(w - W0, w - W0) <= tol;
    same = same || (n > 1 && dot(w - W1, w - W1) <= tol);
    same = same || (n > 2 && dot(w - W2, w - W2) <= tol);
    if (same) break;                              // the same vertex again (polytopes): converged
    // append at slot n, closest point of the new simplex: weights on the slots + keep mask
    double l0, l1, l2 = 0.0, l3 = 0.0;
    int keep;
    if (n == 1) {
      W1 = w; A1 = a;
      const CvxW2 r = cvx_closest_segment(W0, W1);
      l0 = r.l0; l1 = r.l1; keep = r.keep;
    } else if (n == 2) {
      W2 = w; A2 = a;
      const CvxW3 r = cvx_closest_triangle(W0, W1, W2);
      l0 = r.l0; l1 = r.l1; l2 = r.l2; keep = r.keep;
    } else {
      W3 = w; A3 = a;
      const CvxW4 r = cvx_closest_tetrahedron(W0, W1, W2, W3);
      l0 = r.l0; l1 = r.l1; l2 = r.l2; l3 = r.l3; keep = r.keep;
      if (keep == 0) {
        if (lb > 0.0) break;                      // "origin inside" against a certified separation: a flat tetrahedron
        return false;
      }
    }
    const V3 vn = (l0 * W0 + l1 * W1) + (l2 * W2 + l3 * W3);
    // no progress, or a point closer than the certified bound — both are a thin simplex misclassified (or a barycentric
    // denominator lost) to rounding: the previous simplex (slots [0, n), weights lam) is the answer
    const double vnn = dot(vn, vn);
    if (vnn >= vv || vnn < lb * lb * (1.0 - 1e-10)) break;
    // compact the kept vertices to the front (slot order): a select network, no indexed storage
    V3 nW0 = zero, nW1 = zero, nW2 = zero, nA0 = zero, nA1 = zero, nA2 = zero;
    double nl0 = 0.0, nl1 = 0.0, nl2 = 0.0;
    int c = 0;
#define MKH_CVX_PUT(Ws, As, ls, s)                                         \
    if ((keep >> s) & 1) {                                                 \
      if (c == 0) { nW0 = Ws; nA0 = As; nl0 = ls; }                        \
      else if (c == 1) { nW1 = Ws; nA1 = As; nl1 = ls; }                   \
      else { nW2 = Ws; nA2 = As; nl2 = ls; }                               \
      ++c;                                                                 \
    }
    MKH_CVX_PUT(W0, A0, l0, 0)
    MKH_CVX_PUT(W1, A1, l1, 1)
    MKH_CVX_PUT(W2, A2, l2, 2)
    MKH_CVX_PUT(W3, A3, l3, 3)
#undef MKH_CVX_PUT
    W0 = nW0; W1 = nW1; W2 = nW2; A0 = nA0; A1 = nA1; A2 = nA2;
    lam0 = nl0; lam1 = nl1; lam2 = nl2;
    n = c; v = vn;
  }
  pa = lam0 * A0;
  if (n > 1) pa = pa + lam1 * A1;
  if (n > 2) pa = pa + lam2 * A2;
  pb = pa - v;
  dist = sqrt(dot(v, v));
  return true;
}

// depth (> 0) and direction (from 1 to 2) of a separating translation of two overlapping shapes
__device__ __forceinline__ double cvx_penetration(const ConvexGeom& g1, double r1, const ConvexGeom& g2, double r2, V3& dir) {
  auto hs = [&](V3 d, V3& s) -> double {
    s = cvx_support(g1, d) - cvx_support(g2, -1.0 * d);
    return dot(d, s) + r1 + r2;
  };
  V3 d0 = g2.pos - g1.pos;
  const double n0 = sqrt(dot(d0, d0));
  d0 = n0 > 1e-12 ? (1.0 / n0) * d0 : V3{1.0, 0.0, 0.0};
  V3 s, d = d0;
  double h = hs(d0, s);
#pragma nounroll
  for (int k = 0; k < 12; ++k) {                  // ± axes of both shapes (column c of R1 / R2, by selects)
    const bool first = k < 6;
    const int c = (k % 6) >> 1;
    const double m0 = first ? g1.R.m[0] : g2.R.m[0], m1 = first ? g1.R.m[1] : g2.R.m[1], m2 = first ? g1.R.m[2] : g2.R.m[2];
    const double m3 = first ? g1.R.m[3] : g2.R.m[3], m4 = first ? g1.R.m[4] : g2.R.m[4], m5 = first ? g1.R.m[5] : g2.R.m[5];
    const double m6 = first ? g1.R.m[6] : g2.R.m[6], m7 = first ? g1.R.m[7] : g2.R.m[7], m8 = first ? g1.R.m[8] : g2.R.m[8];
    V3 cd{c == 0 ? m0 : (c == 1 ? m1 : m2), c == 0 ? m3 : (c == 1 ? m4 : m5), c == 0 ? m6 : (c == 1 ? m7 : m8)};
    if (k & 1) cd = -1.0 * cd;
    V3 sc;
    const double hc = hs(cd, sc);
    if (hc < h) { h = hc; s = sc; d = cd; }
  }
  double step = 1.0;
#pragma nounroll
  for (int it = 0; it < kPenMaxIters; ++it) {
    const V3 g = s - dot(s, d) * d;               // gradient of d·s(d) on the sphere
    const double gn = sqrt(dot(g, g));
    if (gn < kPenTol * fmax(1.0, fabs(h))) break;
    bool ok = false;
#pragma nounroll
    for (int ls = 0; ls < kPenMaxHalvings; ++ls) {
      V3 dn = d - (step / fmax(sqrt(dot(s, s)), 1e-300)) * g;
      dn = (1.0 / sqrt(dot(dn, dn))) * dn;
      V3 sn;
      const double hn = hs(dn, sn);
      if (hn < h) { d = dn; h = hn; s = sn; ok = true; step = fmin(step * 1.5, 4.0); break; }
      step *= 0.5;
    }
    if (!ok) break;
  }
  dir = d;
  return h;
}

// One contact in mj_geomDistance's convention: n from geom 1 to geom 2, pos the midpoint of the witness points.
__device__ __forceinline__ bool cvx_distance(const ConvexGeom& g1, const ConvexGeom& g2, double margin, double& dist, V3& pos, V3& nrm) {
  const double r1 = cvx_core_radius(g1), r2 = cvx_core_radius(g2);
  double dc = 0.0;
  V3 pa{0, 0, 0}, pb{0, 0, 0};
  const bool apart = cvx_gjk(g1, g2, margin + r1 + r2, dc, pa, pb);
  if (apart && dc > 1e-9) {                       // (cores apart: also when only the spherical shells overlap)
    dist = dc - r1 - r2;
    if (dist > margin) return false;
    nrm = (1.0 / dc) * (pb - pa);
    pos = 0.5 * ((pa + r1 * nrm) + (pb - r2 * nrm));
    return true;
  }
  V3 n;
  const double depth = cvx_penetration(g1, r1, g2, r2, n);
  const V3 a = cvx_support(g1, n) + r1 * n;       // deepest point of 1 along n
  const V3 b = cvx_support(g2, -1.0 * n) - r2 * n;
  dist = -depth; nrm = n; pos = 0.5 * (a + b);
  return true;
}

}  // namespace mkh
