// Distance between two convex primitives by GJK on support mappings: the pairs of mj_geomDistance
// (mink/limits/collision_avoidance_limit.py:214-229) that have no native analytic routine in MuJoCo either and go to its
// general convex collider — cylinder–box, cylinder–cylinder, ellipsoid against anything but a plane.
//
// What is computed is the Euclidean distance of the two convex sets with its witness points (Gilbert–Johnson–Keerthi
// 1988; closest point of a simplex after Ericson, Real-Time Collision Detection §5.1): distance to ~1e-13 relative,
// witness points / normal to ~1e-7 (the support-gap test |v|² − v·w ≤ 1e-14·|v|² bounds the angle of v by its square
// root).  MuJoCo's own answer for these pairs (libccd MPR on shapes inflated by half the margin, tolerance 1e-6) is an
// approximation of the same quantity.  Spheres and capsules enter as their core (point / segment) plus a radius.
// Overlapping shapes: an upper bound of the penetration depth, min over unit d of h₁(d) + h₂(−d) by projected descent from
// the best of the centre-to-centre direction and the shapes' axes (mink only uses the sign of such a distance —
// h = bound_relaxation — and the direction).  oracle/gjk.py is the CPU statement of the same algorithm.
#pragma once
#include "lie_dev.h"

namespace mkh {

constexpr int kGeomSphere = 2, kGeomCapsule = 3, kGeomEllipsoid = 4, kGeomCylinder = 5, kGeomBox = 6;
constexpr int kGjkMaxIters = 128;

struct ConvexGeom { int type; V3 size; V3 pos; M3 R; };

__device__ __forceinline__ double cvx_core_radius(const ConvexGeom& g) {
  return (g.type == kGeomSphere || g.type == kGeomCapsule) ? g.size.x : 0.0;
}

// support point of the CORE in the world: argmax_x d·x
__device__ inline V3 cvx_support(const ConvexGeom& g, V3 d) {
  const V3 dl = mulT(g.R, d);
  V3 s{0.0, 0.0, 0.0};
  if (g.type == kGeomCapsule) {
    s.z = dl.z >= 0.0 ? g.size.y : -g.size.y;
  } else if (g.type == kGeomBox) {
    s = {dl.x >= 0.0 ? g.size.x : -g.size.x, dl.y >= 0.0 ? g.size.y : -g.size.y, dl.z >= 0.0 ? g.size.z : -g.size.z};
  } else if (g.type == kGeomCylinder) {
    const double n = sqrt(dl.x * dl.x + dl.y * dl.y);
    s.z = dl.z >= 0.0 ? g.size.y : -g.size.y;
    if (n >= 1e-300) { s.x = g.size.x * dl.x / n; s.y = g.size.x * dl.y / n; }
  } else if (g.type == kGeomEllipsoid) {
    const V3 e{g.size.x * dl.x, g.size.y * dl.y, g.size.z * dl.z};
    const double n = sqrt(dot(e, e));
    if (n < 1e-300) s = {g.size.x, 0.0, 0.0};
    else s = {g.size.x * e.x / n, g.size.y * e.y / n, g.size.z * e.z / n};
  }
  return g.pos + mul(g.R, s);
}

// Closest point of a simplex to the origin: indices kept (idx[0..n)) and barycentric weights.  Returns n (0: inside).
__device__ inline int cvx_closest_segment(const V3* P, int* idx, double* lam) {
  const V3 ab = P[1] - P[0];
  const double den = dot(ab, ab);
  const double t = den <= 0.0 ? 0.0 : -dot(P[0], ab) / den;
  if (t <= 0.0) { idx[0] = 0; lam[0] = 1.0; return 1; }
  if (t >= 1.0) { idx[0] = 1; lam[0] = 1.0; return 1; }
  idx[0] = 0; idx[1] = 1; lam[0] = 1.0 - t; lam[1] = t;
  return 2;
}

__device__ inline int cvx_closest_triangle(V3 a, V3 b, V3 c, int* idx, double* lam) {
  const V3 ab = b - a, ac = c - a;
  const double d1 = -dot(ab, a), d2 = -dot(ac, a);
  if (d1 <= 0.0 && d2 <= 0.0) { idx[0] = 0; lam[0] = 1.0; return 1; }
  const double d3 = -dot(ab, b), d4 = -dot(ac, b);
  if (d3 >= 0.0 && d4 <= d3) { idx[0] = 1; lam[0] = 1.0; return 1; }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    const double v = d1 / (d1 - d3);
    idx[0] = 0; idx[1] = 1; lam[0] = 1.0 - v; lam[1] = v;
    return 2;
  }
  const double d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6) { idx[0] = 2; lam[0] = 1.0; return 1; }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    const double w = d2 / (d2 - d6);
    idx[0] = 0; idx[1] = 2; lam[0] = 1.0 - w; lam[1] = w;
    return 2;
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    idx[0] = 1; idx[1] = 2; lam[0] = 1.0 - w; lam[1] = w;
    return 2;
  }
  const double den = 1.0 / (va + vb + vc);
  const double v = vb * den, w = vc * den;
  idx[0] = 0; idx[1] = 1; idx[2] = 2; lam[0] = 1.0 - v - w; lam[1] = v; lam[2] = w;
  return 3;
}

__device__ inline int cvx_closest_tetrahedron(const V3* P, int* idx, double* lam) {
  const int faces[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};   // (face, opposite vertex)
  int best_n = 0;
  double best_d2 = __builtin_huge_val();
  for (int f = 0; f < 4; ++f) {
    const V3 a = P[faces[f][0]], b = P[faces[f][1]], c = P[faces[f][2]], dv = P[faces[f][3]];
    const V3 n = cross(b - a, c - a);
    const double sp = -dot(a, n), sd = dot(dv - a, n);
    if (sp * sd < 0.0 || sd == 0.0) {             // the origin is on the far side of this face (or the tetrahedron is flat)
      int li[3];
      double ll[3];
      const int ln = cvx_closest_triangle(a, b, c, li, ll);
      V3 pt{0.0, 0.0, 0.0};
      for (int t = 0; t < ln; ++t) pt = pt + ll[t] * P[faces[f][li[t]]];
      const double d2 = dot(pt, pt);
      if (d2 < best_d2) {
        best_d2 = d2; best_n = ln;
        for (int t = 0; t < ln; ++t) { idx[t] = faces[f][li[t]]; lam[t] = ll[t]; }
      }
    }
  }
  return best_n;
}

// Closest points of the two cores: distance, point on 1, point on 2; returns false when the cores overlap.
__device__ inline bool cvx_gjk(const ConvexGeom& g1, const ConvexGeom& g2, double& dist, V3& pa, V3& pb) {
  V3 d = g1.pos - g2.pos;
  if (dot(d, d) < 1e-30) d = {1.0, 0.0, 0.0};
  V3 W[4], A[4];
  double lam[4] = {1.0, 0.0, 0.0, 0.0}, lam_prev[4] = {1.0, 0.0, 0.0, 0.0};
  int n = 1, n_prev = 1;
  A[0] = cvx_support(g1, -1.0 * d);
  W[0] = A[0] - cvx_support(g2, d);
  V3 v = W[0];
  double lb = 0.0;
  for (int it = 0; it < kGjkMaxIters; ++it) {
    const double vv = dot(v, v);
    double scale = 1e-300;
    for (int i = 0; i < n; ++i) scale = fmax(scale, dot(W[i], W[i]));
    if (vv <= 1e-28 * scale) return false;
    const V3 a = cvx_support(g1, -1.0 * v);
    const V3 w = a - cvx_support(g2, v);
    if (vv - dot(v, w) <= 1e-14 * vv) break;      // no support point is closer to the origin along v: converged
    const double lb_now = dot(v, w) / sqrt(vv);   // every point of the difference is at least this far: a certified bound
    lb = fmax(lb, lb_now);
    bool same = false;
    for (int i = 0; i < n; ++i) same = same || dot(w - W[i], w - W[i]) <= 1e-28 * scale;
    if (same) break;                              // the same vertex again (polytopes): converged
    W[n] = w; A[n] = a;
    int idx[4];
    double ln[4];
    int m;
    if (n == 1) m = cvx_closest_segment(W, idx, ln);
    else if (n == 2) m = cvx_closest_triangle(W[0], W[1], W[2], idx, ln);
    else {
      m = cvx_closest_tetrahedron(W, idx, ln);
      if (m == 0) {
        if (lb > 0.0) break;                      // "origin inside" against a certified separation: a flat tetrahedron
        return false;
      }
    }
    V3 Wn[4], An[4];
    V3 vn{0.0, 0.0, 0.0};
    for (int i = 0; i < m; ++i) { Wn[i] = W[idx[i]]; An[i] = A[idx[i]]; vn = vn + ln[i] * Wn[i]; }
    // no progress, or a point closer than the certified bound — both are a thin simplex misclassified (or a barycentric
    // denominator lost) to rounding: the
    if (dot(vn, vn) >= vv || dot(vn, vn) < lb * lb * (1.0 - 1e-10)) {
      for (int i = 0; i < n_prev; ++i) lam[i] = lam_prev[i];   // previous simplex is the answer
      break;
    }
    for (int i = 0; i < m; ++i) { W[i] = Wn[i]; A[i] = An[i]; lam[i] = ln[i]; lam_prev[i] = ln[i]; }
    n = m; n_prev = m; v = vn;
  }
  pa = {0.0, 0.0, 0.0};
  for (int i = 0; i < n; ++i) pa = pa + lam[i] * A[i];
  pb = pa - v;
  dist = sqrt(dot(v, v));
  return true;
}

// depth (> 0) and direction (from 1 to 2) of a separating translation of two overlapping shapes
__device__ inline double cvx_penetration(const ConvexGeom& g1, double r1, const ConvexGeom& g2, double r2, V3& dir) {
  auto hs = [&](V3 d, V3& s) -> double {
    s = cvx_support(g1, d) - cvx_support(g2, -1.0 * d);
    return dot(d, s) + r1 + r2;
  };
  V3 d0 = g2.pos - g1.pos;
  const double n0 = sqrt(dot(d0, d0));
  d0 = n0 > 1e-12 ? (1.0 / n0) * d0 : V3{1.0, 0.0, 0.0};
  V3 s, d = d0;
  double h = hs(d0, s);
  for (int k = 0; k < 12; ++k) {                  // ± axes of both shapes
    const M3& R = (k < 6) ? g1.R : g2.R;
    const int c = (k % 6) >> 1;
    V3 cd{R.m[c], R.m[3 + c], R.m[6 + c]};
    if (k & 1) cd = -1.0 * cd;
    V3 sc;
    const double hc = hs(cd, sc);
    if (hc < h) { h = hc; s = sc; d = cd; }
  }
  double step = 1.0;
  for (int it = 0; it < kGjkMaxIters; ++it) {
    const V3 g = s - dot(s, d) * d;               // gradient of d·s(d) on the sphere
    const double gn = sqrt(dot(g, g));
    if (gn < 1e-12 * fmax(1.0, fabs(h))) break;
    bool ok = false;
    for (int ls = 0; ls < 20; ++ls) {
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
__device__ inline bool cvx_distance(const ConvexGeom& g1, const ConvexGeom& g2, double margin, double& dist, V3& pos, V3& nrm) {
  const double r1 = cvx_core_radius(g1), r2 = cvx_core_radius(g2);
  double dc = 0.0;
  V3 pa{0, 0, 0}, pb{0, 0, 0};
  const bool apart = cvx_gjk(g1, g2, dc, pa, pb);
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
