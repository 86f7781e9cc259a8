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
// Overlapping shapes: the smallest separating translation (depth, direction, deepest points) by the expanding polytope
// algorithm, the wavefront cooperating on one pair at a time (cvx_epa below; round 4 — rounds 2 and 3 ran a projected
// descent of h₁(d) + h₂(−d), which stalls on the kinks of polytope support functions: device and oracle stopped at different
// points of the same kink, found by the whole-batch parity test of the `ur5e_convex` workload).  mink uses the sign of such
// a distance (h = bound_relaxation) and the direction.  oracle/gjk.py is the CPU statement of the same algorithm.
//
// Register discipline (round 3).  The first version kept the simplex in arrays indexed by the sub-algorithm's results
// (`W[idx[i]]`, `P[faces[f][li[t]]]`): hipcc sends every runtime-indexed private array to SCRATCH, so a GJK run was a
// chain of scratch round trips (528–1 120 B of scratch per lane, 4 096 UR5e problems with ONE cylinder–box pair took
// 0.39 ms against 0.036 ms without it).  Here every simplex vertex is a named value: the sub-algorithms return barycentric
// weights on FIXED slots plus a keep mask, appending and compacting the simplex are select networks, the faces of the
// tetrahedron are four static instantiations.  Nothing is indexed at run time; the routine compiles without scratch.
#pragma once
#include "lie_dev.h"
#include "wave_ops.h"

namespace mkh {

constexpr int kGeomSphere = 2, kGeomCapsule = 3, kGeomEllipsoid = 4, kGeomCylinder = 5, kGeomBox = 6, kGeomMesh = 7;
constexpr int kGjkMaxIters = 128;
// On a curved rim GJK converges linearly (the squared distance gains a digit every three or four iterations, the last ones
// below anything a row of G can see): once an iteration moves |v|² by less than this fraction the simplex stands.  The
// distance is then good to ~1e-12 relative, the witness points to ~1e-6 of the shapes' size; oracle/gjk.py: GJK_PROGRESS.
constexpr double kGjkProgress = 1e-12;

// vert / nvert: convex-hull vertices of a mesh geom in the geom's frame (global memory, 3 doubles each); nullptr else
struct ConvexGeom { int type; V3 size; V3 pos; M3 R; const double* vert; int nvert; };

__device__ __forceinline__ double cvx_core_radius(const ConvexGeom& g) {
  return (g.type == kGeomSphere || g.type == kGeomCapsule) ? g.size.x : 0.0;
}

// support point of the CORE in the geom's own frame: argmax_x dl·x
__device__ __forceinline__ V3 cvx_support_local(const int type, const V3 size, const double* vert, const int nvert, const V3 dl) {
  V3 s{0.0, 0.0, 0.0};
  if (type == kGeomCapsule) {
    s.z = dl.z >= 0.0 ? size.y : -size.y;
  } else if (type == kGeomBox) {
    s = {dl.x >= 0.0 ? size.x : -size.x, dl.y >= 0.0 ? size.y : -size.y, dl.z >= 0.0 ? size.z : -size.z};
  } else if (type == kGeomCylinder) {
    const double n = sqrt(dl.x * dl.x + dl.y * dl.y);
    s.z = dl.z >= 0.0 ? size.y : -size.y;
    if (n >= 1e-300) { const double k = size.x * fast_rcp(n); s.x = k * dl.x; s.y = k * dl.y; }
  } else if (type == kGeomEllipsoid) {
    const V3 e{size.x * dl.x, size.y * dl.y, size.z * dl.z};
    const double n = sqrt(dot(e, e));
    if (n < 1e-300) s = {size.x, 0.0, 0.0};
    else { const double k = fast_rcp(n); s = {size.x * e.x * k, size.y * e.y * k, size.z * e.z * k}; }
  } else if (type == kGeomMesh) {
    // hull vertices: the first maximiser of d·x (numpy argmax order, oracle/gjk.py).  Four vertices per trip: the loads of
    // a trip are independent, the running maximum is the only chain.
    double best = -__builtin_huge_val();
    const double* v = vert;
    int i = 0;
    for (; i + 4 <= nvert; i += 4) {
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
    for (; i < nvert; ++i) {
      const double* p = v + 3 * i;
      const double x0 = p[0], y0 = p[1], z0 = p[2];
      const double d0 = dl.x * x0 + dl.y * y0 + dl.z * z0;
      if (d0 > best) { best = d0; s = {x0, y0, z0}; }
    }
  }
  return s;
}

// support point of the CORE in the world: argmax_x d·x
__device__ __forceinline__ V3 cvx_support(const ConvexGeom& g, V3 d) {
  return g.pos + mul(g.R, cvx_support_local(g.type, g.size, g.vert, g.nvert, mulT(g.R, d)));
}

// The pair in the frame of shape 1 (the separated case, cvx_gjk): shape 1 needs no rotation at all and shape 2 one
// quaternion — 13 doubles of geometry instead of 30 live across the loop, a third fewer rotations per iteration.
struct ConvexRel { int t1; V3 s1; const double* vert1; int nvert1; int t2; V3 s2; const double* vert2; int nvert2; Q4 q21; V3 p21; };
__device__ __forceinline__ V3 cvx_support1(const ConvexRel& g, V3 d) { return cvx_support_local(g.t1, g.s1, g.vert1, g.nvert1, d); }
__device__ __forceinline__ V3 cvx_support2(const ConvexRel& g, V3 d) {
  return g.p21 + qrot(g.q21, cvx_support_local(g.t2, g.s2, g.vert2, g.nvert2, qrot(qconj(g.q21), d)));
}


// Closest point of a simplex to the origin as barycentric weights on the simplex's own vertices + the vertices kept
// (bit i of keep).  keep = 0 from the tetrahedron: the origin is inside.
struct CvxW2 { double l0, l1; int keep; };
struct CvxW3 { double l0, l1, l2; int keep; };
struct CvxW4 { double l0, l1, l2, l3; int keep; };

__device__ __forceinline__ CvxW2 cvx_closest_segment(V3 a, V3 b) {
  const V3 ab = b - a;
  const double den = dot(ab, ab);
  const double t = den <= 0.0 ? 0.0 : -dot(a, ab) * fast_rcp(den);
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
    const double v = d1 * fast_rcp(d1 - d3);
    return {1.0 - v, v, 0.0, 3};
  }
  const double d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6) return {0.0, 0.0, 1.0, 4};
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    const double w = d2 * fast_rcp(d2 - d6);
    return {1.0 - w, 0.0, w, 5};
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    const double w = (d4 - d3) * fast_rcp((d4 - d3) + (d5 - d6));
    return {0.0, 1.0 - w, w, 6};
  }
  const double den = fast_rcp(va + vb + vc);
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
  // (face, opposite vertex) as in oracle/gjk.py _FACES.  p3 is the vertex GJK has just added, (p0, p1, p2) the simplex it
  // stood on: that face cannot hold a point closer than the one it held already (the caller stops on "no progress"), and the
  // origin and p3 lie on the same side of it — only the three faces at p3 are looked at.
  cvx_tet_face<0, 2, 3>(p0, p2, p3, p1, best_d2, best);
  cvx_tet_face<0, 3, 1>(p0, p3, p1, p2, best_d2, best);
  cvx_tet_face<1, 3, 2>(p1, p3, p2, p0, best_d2, best);
  return best;
}

// Closest points of the two cores: distance, point on 1, point on 2; returns false when the cores overlap.
// `cutoff`: the caller discards the pair when the cores are farther apart than this — the loop stops as soon as its
// certified lower bound says so (dist = that bound, no witness points): two or three iterations instead of ~9 for the
// pairs of a batch that are nowhere near each other.
//
// The simplex lives in LDS (round 4): vertex k of this lane at S[(6k + c)·kGjkSlots], c = 0..2 its point of the difference,
// 3..5 its point on shape 1 — lane slots side by side, so the lanes of a wavefront never share a bank.  Rounds 2 and 3 kept
// it in named registers with select networks for every append / compaction (runtime-indexed private arrays go to scratch):
// 196 VGPRs, which as a callee saved 82 registers per problem (209 MB of scratch traffic per launch of the `ur5e_convex`
// workload) and inlined spilled inside the loop.  LDS takes runtime indices: the loop keeps the logical order of the vertices as
// four 2-bit slot numbers, loads the ≤ 3 standing vertices per iteration and the points on shape 1 only at the end.
constexpr int kGjkSlots = 32;                       // lanes of a wavefront that run GJK at the same time (the caller batches)
constexpr int kGjkWsDoubles = 24 * kGjkSlots;
// (in the frame of shape 1: pa / pb come out in that frame)
// gap / progress: the support-gap and no-progress tolerances of the run (1e-14 / kGjkProgress; round 6: a LOOSE run — kLooseGap — in
// front of the witness-point polish, cvx_distance)
__device__ __forceinline__ bool cvx_gjk(const ConvexRel& g, const double cutoff, double& dist, V3& pa, V3& pb, double* S,
                                        const double gap = 1e-14, const double progress = kGjkProgress) {
  auto ldW = [&](int k) -> V3 { const double* p = S + 6 * k * kGjkSlots; return V3{p[0], p[kGjkSlots], p[2 * kGjkSlots]}; };
  auto ldA = [&](int k) -> V3 { const double* p = S + (6 * k + 3) * kGjkSlots; return V3{p[0], p[kGjkSlots], p[2 * kGjkSlots]}; };
  auto st = [&](int k, V3 w, V3 a) {
    double* p = S + 6 * k * kGjkSlots;
    p[0] = w.x; p[kGjkSlots] = w.y; p[2 * kGjkSlots] = w.z; p[3 * kGjkSlots] = a.x; p[4 * kGjkSlots] = a.y; p[5 * kGjkSlots] = a.z;
  };
  V3 d = -1.0 * g.p21;
  if (dot(d, d) < 1e-30) d = {1.0, 0.0, 0.0};
  const V3 zero{0.0, 0.0, 0.0};
  int ord = 0;                                     // physical slot of logical vertex i: bits [2i, 2i + 2)
  int n = 1;
  double lam0 = 1.0, lam1 = 0.0, lam2 = 0.0;       // weights of the last accepted simplex (n ≤ 3 vertices)
  {
    const V3 a0 = cvx_support1(g, -1.0 * d);
    const V3 w0 = a0 - cvx_support2(g, d);
    st(0, w0, a0);
    pa = w0;                                       // (v below)
  }
  V3 v = pa;
  double lb = 0.0;
#pragma nounroll
  for (int it = 0; it < kGjkMaxIters; ++it) {
    const V3 W0 = ldW(ord & 3), W1 = n > 1 ? ldW((ord >> 2) & 3) : zero, W2 = n > 2 ? ldW((ord >> 4) & 3) : zero;
    const double vv = dot(v, v);
    double scale = fmax(1e-300, dot(W0, W0));
    if (n > 1) scale = fmax(scale, dot(W1, W1));
    if (n > 2) scale = fmax(scale, dot(W2, W2));
    if (vv <= 1e-28 * scale) return false;
    const V3 a = cvx_support1(g, -1.0 * v);
    const V3 w = a - cvx_support2(g, v);
    const double vw = dot(v, w);
    if (vv - vw <= gap * vv) break;               // no support point is closer to the origin along v: converged
    lb = fmax(lb, vw / sqrt(vv));                 // every point of the difference is at least this far: a certified bound
    if (lb > cutoff) { dist = lb; return true; }
    const double tol = 1e-28 * scale;
    bool same = dot(w - W0, w - W0) <= tol;
    same = same || (n > 1 && dot(w - W1, w - W1) <= tol);
    same = same || (n > 2 && dot(w - W2, w - W2) <= tol);
    if (same) break;                              // the same vertex again (polytopes): converged
    // the free physical slot takes the new vertex (logical index n)
    int used = 1 << (ord & 3);
    if (n > 1) used |= 1 << ((ord >> 2) & 3);
    if (n > 2) used |= 1 << ((ord >> 4) & 3);
    const int fr = __builtin_ctz(~used);
    st(fr, w, a);
    // closest point of the new simplex: weights on the logical vertices + keep mask
    double l0, l1, l2 = 0.0, l3 = 0.0;
    int keep;
    if (n == 1) {
      const CvxW2 r = cvx_closest_segment(W0, w);
      l0 = r.l0; l1 = r.l1; keep = r.keep;
    } else if (n == 2) {
      const CvxW3 r = cvx_closest_triangle(W0, W1, w);
      l0 = r.l0; l1 = r.l1; l2 = r.l2; keep = r.keep;
    } else {
      const CvxW4 r = cvx_closest_tetrahedron(W0, W1, W2, w);
      l0 = r.l0; l1 = r.l1; l2 = r.l2; l3 = r.l3; keep = r.keep;
      if (keep == 0) {
        if (lb > 0.0) break;                      // "origin inside" against a certified separation: a flat tetrahedron
        return false;
      }
    }
    const V3 Wn1 = n == 1 ? w : W1, Wn2 = n == 2 ? w : W2, Wn3 = n == 3 ? w : zero;      // logical vertices 1..3 of the new simplex
    const V3 vn = (l0 * W0 + l1 * Wn1) + (l2 * Wn2 + l3 * Wn3);
    // no progress, or a point closer than the certified bound — both are a thin simplex misclassified (or a barycentric
    // denominator lost) to rounding: the previous simplex (logical [0, n), weights lam) is the answer
    const double vnn = dot(vn, vn);
    if (vnn >= vv || vnn < lb * lb * (1.0 - 1e-10)) break;
    const bool done = vv - vnn <= progress * vv;              // the distance has stopped moving: this simplex is the answer
    // keep the kept vertices in their order: new slot numbers and weights
    const int full = ord | (fr << (2 * n));        // logical → physical of the simplex with the new vertex
    int nord = 0, c = 0;
    double nl0 = 0.0, nl1 = 0.0, nl2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((keep >> i) & 1) {
        const double li = i == 0 ? l0 : (i == 1 ? l1 : (i == 2 ? l2 : l3));
        nord |= ((full >> (2 * i)) & 3) << (2 * c);
        if (c == 0) nl0 = li; else if (c == 1) nl1 = li; else nl2 = li;
        ++c;
      }
    }
    ord = nord; lam0 = nl0; lam1 = nl1; lam2 = nl2;
    n = c; v = vn;
    if (done) break;
  }
  pa = lam0 * ldA(ord & 3);
  if (n > 1) pa = pa + lam1 * ldA((ord >> 2) & 3);
  if (n > 2) pa = pa + lam2 * ldA((ord >> 4) & 3);
  pb = pa - v;
  dist = sqrt(dot(v, v));
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Overlapping cores: the SMALLEST separating translation — the point of the boundary of the Minkowski difference
// D = {x₁ − x₂} nearest to the origin — by the expanding polytope algorithm (van den Bergen 2001), ONE pair per call with
// the whole wavefront cooperating: the faces of the inner polytope are spread over the lanes (slot s on lane s mod 64), so
// "the face whose plane is nearest to the origin", "the faces that see the new support point" and the horizon of those are
// wave reductions / ballots instead of loops; vertices, faces and one bit per directed edge live in `ws` (LDS,
// kEpaWsDoubles).  Plane offsets are SIGNED: the start polytope (two support points along ±x, the farthest point from
// their line, both sides of that triangle) need not contain the origin yet; faces with a negative offset are expanded
// first.  A face is final when the support point along its normal lies within kEpaTol of its plane.  oracle/gjk.py
// `penetration` states the same rules (slot order of new faces, ties by slot) sequentially.  What MuJoCo's collider
// returns for such a pair approximates this quantity (libccd MPR to 1e-6 in mujoco 3.1.6; its native GJK + EPA later).
// Arguments arrive wave-uniform; every lane executes every statement (no divergent call).
// (the workgroup-per-problem kernel, wide_kernel.h, runs it in one of four wavefronts: a wave-local fence instead of the barrier)
#ifndef MKH_EPA_SYNC
#define MKH_EPA_SYNC() wave_sync()
#endif
constexpr int kEpaMaxV = 48, kEpaMaxF = 92;
constexpr double kEpaTol = 1e-11;
constexpr int kEpaOffV = 0, kEpaOffF = 6 * kEpaMaxV, kEpaOffI = kEpaOffF + 4 * kEpaMaxF, kEpaOffE = kEpaOffI + (kEpaMaxF + 1) / 2,
              kEpaOffL = kEpaOffE + kEpaMaxV, kEpaWsDoubles = kEpaOffL + (kEpaMaxF + 7) / 8 + 1;
struct CvxEpa { double depth; V3 n, a, b; };

__device__ __forceinline__ double wave_min_f64(double x) {
  x = fmin(x, dpp_f64<0xB1>(x));
  x = fmin(x, dpp_f64<0x4E>(x));
  x = fmin(x, dpp_f64<0x141>(x));
  x = fmin(x, dpp_f64<0x140>(x));
  return fmin(fmin(readlane_f64(x, 0), readlane_f64(x, 16)), fmin(readlane_f64(x, 32), readlane_f64(x, 48)));
}

__device__ __forceinline__ CvxEpa cvx_epa(const ConvexGeom& g1, const ConvexGeom& g2, double* ws, const double tol = kEpaTol) {
  const int lane = lane_id();
  double* const V = ws + kEpaOffV;                                   // vertex k: W at V[6k], A (its point on shape 1) at V[6k + 3]
  double* const F = ws + kEpaOffF;                                   // face s: unit normal F[4s..4s+2], plane offset F[4s+3]
  int* const Fi = reinterpret_cast<int*>(ws + kEpaOffI);             // face s: i | j << 8 | k << 16, −1 = free slot
  unsigned long long* const E = reinterpret_cast<unsigned long long*>(ws + kEpaOffE);   // bit j of E[i]: edge i→j of a visible face
  unsigned char* const freel = reinterpret_cast<unsigned char*>(ws + kEpaOffL);         // freed slots in ascending order
  const double kInf = __builtin_huge_val();
  int nvert = 0;
  auto add = [&](V3 d) -> V3 {                                       // support point of D along d (uniform): vertex nvert
    const V3 a = cvx_support(g1, d);
    const V3 w = a - cvx_support(g2, -1.0 * d);
    if (lane == 0) {
      double* o = V + 6 * nvert;
      o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = a.x; o[4] = a.y; o[5] = a.z;
    }
    ++nvert;
    return w;
  };
  auto vtx = [&](int k) -> V3 { return V3{V[6 * k], V[6 * k + 1], V[6 * k + 2]}; };
  // face (i, j, k) in its given winding → slot s (a triangle without area can never be the nearest face, but stays: the
  // surface must remain closed)
  auto put_face = [&](int s, int i, int j, int k, V3 wi, V3 wj, V3 wk) {
    const V3 n = cross(wj - wi, wk - wi);
    const double l = sqrt(dot(n, n));
    double* o = F + 4 * s;
    if (l < 1e-150) { o[0] = 0.0; o[1] = 0.0; o[2] = 0.0; o[3] = kInf; }
    else {
      const V3 u{n.x / l, n.y / l, n.z / l};
      o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = dot(u, wi);
    }
    Fi[s] = i | (j << 8) | (k << 16);
  };
  const V3 w0 = add(V3{1.0, 0.0, 0.0}), w1 = add(V3{-1.0, 0.0, 0.0});
  const V3 u = w1 - w0;
  const double uu = dot(u, u);
  const double ax = fabs(u.x), ay = fabs(u.y), az = fabs(u.z);
  V3 e{0.0, 0.0, 0.0};
  if (ax <= ay && ax <= az) e.x = 1.0; else if (ay <= az) e.y = 1.0; else e.z = 1.0;     // (numpy argmin: the first minimum)
  V3 v = uu > 0.0 ? e - (dot(e, u) / uu) * u : e;
  v = (1.0 / sqrt(dot(v, v))) * v;
  V3 w2 = add(v);
  if (fabs(dot(w2 - w0, v)) <= 1e-12 * fmax(1.0, sqrt(uu))) { --nvert; w2 = add(-1.0 * v); }
  V3 n0 = cross(w1 - w0, w2 - w0);
  n0 = (1.0 / fmax(sqrt(dot(n0, n0)), 1e-300)) * n0;
  const V3 w3 = add(n0), w4 = add(-1.0 * n0);
  const V3 sum5 = (((w0 + w1) + w2) + w3) + w4;
  const V3 cen{sum5.x / 5.0, sum5.y / 5.0, sum5.z / 5.0};
  MKH_EPA_SYNC();
  if (lane < 6) {
    const int i = lane % 3, j = (lane + 1) % 3, k = lane < 3 ? 3 : 4;
    const V3 wi = i == 0 ? w0 : (i == 1 ? w1 : w2), wj = j == 0 ? w0 : (j == 1 ? w1 : w2), wk = lane < 3 ? w3 : w4;
    put_face(lane, i, j, k, wi, wj, wk);
    if (F[4 * lane] * (wi.x - cen.x) + F[4 * lane + 1] * (wi.y - cen.y) + F[4 * lane + 2] * (wi.z - cen.z) < 0.0)
      put_face(lane, i, k, j, wi, wk, wj);
  }
  MKH_EPA_SYNC();
  int nslots = 6, best = 0;
  bool intact = true;
  int ans_id = -1;                                                   // the answer so far: vertices, normal, offset of the nearest face
  V3 ans_n{1.0, 0.0, 0.0};
  double ans_off = 0.0;
#pragma nounroll
  for (;;) {
    // the face whose plane is nearest to the origin (signed), lowest slot on ties
    double my_off = kInf;
    unsigned my_slot = 0xffffffffu;
    for (int s = lane; s < nslots; s += 64)
      if (Fi[s] >= 0) { const double off = F[4 * s + 3]; if (off < my_off || my_slot == 0xffffffffu) { my_off = off; my_slot = (unsigned)s; } }
    const double mo = wave_min_f64(my_off);
    best = (int)wave_min_u32((my_slot != 0xffffffffu && my_off == mo) ? my_slot : 0xffffffffu);
    const V3 nb{F[4 * best], F[4 * best + 1], F[4 * best + 2]};
    const double off = F[4 * best + 3];
    // the nearest plane can only move outwards as the polytope grows; when it jumps back in, a sliver face (three nearly
    // collinear support points near convergence: a normal without digits) has been created — the previous face stands
    if (ans_id >= 0 && ans_off >= 0.0 && off < ans_off - 1e-9 * fmax(1.0, ans_off)) { intact = false; break; }
    ans_id = Fi[best]; ans_n = nb; ans_off = off;
    if (nvert >= kEpaMaxV) break;
    const int ip = nvert;
    const V3 p = add(nb);
    const double gap = dot(nb, p) - off;
    if (gap <= tol * fmax(1.0, fabs(off))) { --nvert; break; }              // the face is (within the gap) a face of D itself
    if (lane < kEpaMaxV) E[lane] = 0ull;
    MKH_EPA_SYNC();
    // faces that see p, and one bit per directed edge of those
    bool vis0 = false, vis1 = false;
    int id0 = -1, id1 = -1;
    {
      const int s = lane;
      if (s < nslots && (id0 = Fi[s]) >= 0) {
        const double o = F[4 * s + 3];
        vis0 = (F[4 * s] * p.x + F[4 * s + 1] * p.y + F[4 * s + 2] * p.z) - o > 1e-13 * fmax(1.0, fabs(o)) || s == best;
      }
      const int t = lane + 64;
      if (t < nslots && (id1 = Fi[t]) >= 0) {
        const double o = F[4 * t + 3];
        vis1 = (F[4 * t] * p.x + F[4 * t + 1] * p.y + F[4 * t + 2] * p.z) - o > 1e-13 * fmax(1.0, fabs(o)) || t == best;
      }
    }
    auto mark = [&](int id) {
      const int i = id & 255, j = (id >> 8) & 255, k = (id >> 16) & 255;
      atomicOr(&E[i], 1ull << j); atomicOr(&E[j], 1ull << k); atomicOr(&E[k], 1ull << i);
    };
    if (vis0) mark(id0);
    if (vis1) mark(id1);
    MKH_EPA_SYNC();
    // horizon: edges of visible faces whose twin belongs to a face that does not see p
    auto horizon = [&](int id) -> int {
      const int i = id & 255, j = (id >> 8) & 255, k = (id >> 16) & 255;
      return (int)(((E[j] >> i) & 1ull) ^ 1ull) | ((int)(((E[k] >> j) & 1ull) ^ 1ull) << 1) | ((int)(((E[i] >> k) & 1ull) ^ 1ull) << 2);
    };
    const int hz0 = vis0 ? horizon(id0) : 0, hz1 = vis1 ? horizon(id1) : 0;
    const int c0 = __popc(hz0), c1 = __popc(hz1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long v0m = __ballot(vis0), v1m = __ballot(vis1);
    const unsigned long long a0 = __ballot(c0 & 1), b0 = __ballot(c0 & 2), a1 = __ballot(c1 & 1), b1 = __ballot(c1 & 2);
    const int tot0 = __popcll(a0) + 2 * __popcll(b0), tot1 = __popcll(a1) + 2 * __popcll(b1);
    const int nvis = __popcll(v0m) + __popcll(v1m), nh = tot0 + tot1;
    if (nslots - nvis + nh > kEpaMaxF) { --nvert; break; }                    // out of slots: the nearest face so far is the answer
    int q0 = __popcll(a0 & lt) + 2 * __popcll(b0 & lt);                        // index of this lane's first horizon edge, (slot, edge) order
    int q1 = tot0 + __popcll(a1 & lt) + 2 * __popcll(b1 & lt);
    if (vis0) { freel[__popcll(v0m & lt)] = (unsigned char)lane; Fi[lane] = -1; }
    if (vis1) { freel[__popcll(v0m) + __popcll(v1m & lt)] = (unsigned char)(lane + 64); Fi[lane + 64] = -1; }
    MKH_EPA_SYNC();
    auto emit = [&](int id, int hz, int q) {
      const int i = id & 255, j = (id >> 8) & 255, k = (id >> 16) & 255;
#pragma unroll
      for (int ed = 0; ed < 3; ++ed) {
        if (!((hz >> ed) & 1)) continue;
        const int a_ = ed == 0 ? i : (ed == 1 ? j : k), b_ = ed == 0 ? j : (ed == 1 ? k : i);
        const int slot = q < nvis ? (int)freel[q] : nslots + (q - nvis);
        put_face(slot, a_, b_, ip, vtx(a_), vtx(b_), p);
        ++q;
      }
    };
    if (hz0) emit(id0, hz0, q0);
    if (hz1) emit(id1, hz1, q1);
    if (nh > nvis) nslots += nh - nvis;
    MKH_EPA_SYNC();
  }
  MKH_EPA_SYNC();
  if (intact) {
    // a face of D is usually covered by several coplanar triangles: of those in the answer's plane (offsets within 1e-9)
    // the one NEAREST to the origin as a triangle — the one that holds the foot of the perpendicular — gives the witness
    // points (a clamped foot on a neighbouring triangle would put them off the shapes)
    const double thr = ans_off + 1e-9 * fmax(1.0, fabs(ans_off));
    double my_d2 = kInf;
    unsigned my_slot = 0xffffffffu;
    for (int sl = lane; sl < nslots; sl += 64) {
      const int fid = Fi[sl];
      if (fid < 0 || !(F[4 * sl + 3] <= thr)) continue;
      const V3 a_ = vtx(fid & 255), b_ = vtx((fid >> 8) & 255), c_ = vtx((fid >> 16) & 255);
      const CvxW3 t = cvx_closest_triangle(a_, b_, c_);
      const V3 w = t.l0 * a_ + (t.l1 * b_ + t.l2 * c_);
      const double d2 = dot(w, w);
      if (d2 < my_d2 || my_slot == 0xffffffffu) { my_d2 = d2; my_slot = (unsigned)sl; }
    }
    const double md = wave_min_f64(my_d2);
    const int win = (int)wave_min_u32((my_slot != 0xffffffffu && my_d2 == md) ? my_slot : 0xffffffffu);
    ans_id = Fi[win];
    ans_n = V3{F[4 * win], F[4 * win + 1], F[4 * win + 2]};
  }
  const V3 nb = ans_n;
  const int id = ans_id;
  const int i = id & 255, j = (id >> 8) & 255, k = (id >> 16) & 255;
  const V3 wi = vtx(i), wj = vtx(j), wk = vtx(k);
  const CvxW3 t = cvx_closest_triangle(wi, wj, wk);                          // the face's point nearest to the origin, barycentric
  const V3 ai{V[6 * i + 3], V[6 * i + 4], V[6 * i + 5]}, aj{V[6 * j + 3], V[6 * j + 4], V[6 * j + 5]}, ak{V[6 * k + 3], V[6 * k + 4], V[6 * k + 5]};
  CvxEpa r;
  r.a = t.l0 * ai + (t.l1 * aj + t.l2 * ak);
  r.b = r.a - (t.l0 * wi + (t.l1 * wj + t.l2 * wk));
  r.n = nb;
  // (geom 2 translated by t overlaps geom 1 iff t ∈ D: the shortest separating translation of geom 2 is depth·n)
  r.depth = dot(nb, cvx_support(g1, nb)) - dot(nb, cvx_support(g2, -1.0 * nb));
  MKH_EPA_SYNC();
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Witness points on the EXACT features (round 6; oracle/gjk.py `polish` is the numpy statement and carries the derivation).
// GJK and the expanding polytope approximate a curved rim by chords: the distance converges to ~1e-13, the direction and the
// witness points only to ~1e-6, and a row of G inherits that (`ur5e_convex`: v to 4e-6).  Both answers minimise the support
// function of the difference over unit directions, h_D(n) = h_1(n) + h_2(−n) (signed distance −min h_D, n from shape 1 to
// shape 2), with a ∈ F_1(n), b ∈ F_2(−n), a − b = h_D·n.  h_D is smooth except on the planes n·k = 0 (k a box axis, the axis of
// a cylinder / a capsule's segment) and at the poles ±u of a cylinder.  Candidates, in this order: a pole within kKinkTol of
// the given direction n0; the first two near planes together (n = ±k_a × k_b); each alone and none — Newton in the remaining
// tangent space with the reduced Hessian Tᵀ(∇²h_1 + ∇²h_2)T − h_D·I (cylinder: (r/ρ)·t·tᵀ, ellipsoid: (S² − (S²d)(S²d)ᵀ/h²)/h,
// zero otherwise).  A candidate is accepted only with a certificate: witnesses built from the support sets (a unique support
// point fixes the other's through a − b = h_D·n; edge against edge is a 2 × 2 solve) lie in their shapes to kPolishTol.
// Mesh hulls (kinks not enumerated) and non-unique witnesses (face against edge / face) keep the caller's answer.
// Everything in the frame of shape 1: shape 2 sits at p21 with rotation R21.
constexpr double kKinkTol = 1e-3, kPolishTol = 1e-10;


// 1/√x: hardware estimate + two Newton steps (the routine normalises a dozen vectors per contact; the IEEE sqrt + division pair is
// ≈ 30 instructions of one dependent chain each, on a single busy lane)
__device__ __forceinline__ double cvx_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

__device__ __forceinline__ bool cvx_core_contains(int type, V3 size, V3 l, double tol) {
  if (type == kGeomBox) return fabs(l.x) <= size.x + tol && fabs(l.y) <= size.y + tol && fabs(l.z) <= size.z + tol;
  if (type == kGeomCylinder) return l.x * l.x + l.y * l.y <= (size.x + tol) * (size.x + tol) && fabs(l.z) <= size.y + tol;
  if (type == kGeomEllipsoid) {
    const double x = l.x / size.x, y = l.y / size.y, z = l.z / size.z;
    return x * x + y * y + z * z <= 1.0 + 2.0 * tol / fmin(size.x, fmin(size.y, size.z));
  }
  if (type == kGeomCapsule) return l.x * l.x + l.y * l.y <= tol * tol && fabs(l.z) <= size.y + tol;
  if (type == kGeomSphere) return dot(l, l) <= tol * tol;
  return false;
}

// τᵀ·∇²h(d)·τ' of the core's support function, everything in the shape's own frame (d unit); ok = false at a cylinder's pole
__device__ __forceinline__ void cvx_hess_form(int type, V3 size, V3 d, V3 t1, V3 t2, double& m11, double& m12, double& m22, bool& ok) {
  if (type == kGeomCylinder) {
    const double rho2 = d.x * d.x + d.y * d.y;
    if (rho2 < 1e-24) { ok = false; return; }
    const double ir = cvx_rsqrt(rho2);
    const V3 t{-d.y * ir, d.x * ir, 0.0};
    const double a = dot(t, t1), b = dot(t, t2), c = size.x * ir;
    m11 += c * a * a; m12 += c * a * b; m22 += c * b * b;
  } else if (type == kGeomEllipsoid) {
    const V3 s2{size.x * size.x, size.y * size.y, size.z * size.z};
    const V3 e{s2.x * d.x, s2.y * d.y, s2.z * d.z};
    const double h2 = e.x * d.x + e.y * d.y + e.z * d.z, ih = cvx_rsqrt(h2), ih2 = ih * ih;
    auto form = [&](V3 u, V3 w) { return ((s2.x * u.x * w.x + s2.y * u.y * w.y + s2.z * u.z * w.z) - dot(e, u) * dot(e, w) * ih2) * ih; };
    m11 += form(t1, t1); m12 += form(t1, t2); m22 += form(t2, t2);
  }
}

struct CvxPolish { bool ok; double h; V3 a, b, n; };
#ifndef MKH_POLISH_ATTR
#define MKH_POLISH_ATTR __forceinline__
#endif
// Inlined, with ONE evaluation site of the two support mappings (the Newton loop's; its last pass is the certificate's).  Measured
// on `ur5e_convex` (phase clocks, collision phase per problem): as a real call every problem paid 68 k cycles for the callee-saved
// blocks of a non-leaf collision phase, pair in range or not; with the support mapping as a callee of its own every evaluation
// spilled the routine's live values around the call (≈ 4 k cycles each).
__device__ MKH_POLISH_ATTR CvxPolish cvx_polish(const int t1, const V3 s1, const int t2, const V3 s2, const Q4 q21, const V3 p21, V3 n0) {
  CvxPolish out{false, 0.0, {0, 0, 0}, {0, 0, 0}, {1, 0, 0}};
  if (t1 == kGeomMesh || t2 == kGeomMesh) return out;
  const M3 R21 = qmat(q21);
  const double scale = fmax(fmax(fabs(s1.x), fmax(fabs(s1.y), fabs(s1.z))), fmax(fabs(s2.x), fmax(fabs(s2.y), fabs(s2.z))));
  const double tol = kPolishTol * fmax(scale, 1e-3);
  n0 = cvx_rsqrt(dot(n0, n0)) * n0;
  // kink planes, slots 0-2: the axes of shape 1, 3-5: those of shape 2 (a box has three, a cylinder / capsule its z axis)
  // (select chains on named values: a runtime index into R21.m would send the matrix to scratch)
  const V3 c0{R21.m[0], R21.m[3], R21.m[6]}, c1{R21.m[1], R21.m[4], R21.m[7]}, c2{R21.m[2], R21.m[5], R21.m[8]};
  auto kink = [&](int i) -> V3 {
    const V3 r = i == 3 ? c0 : (i == 4 ? c1 : c2);
    return i < 3 ? V3{i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0} : r;
  };
  const bool ax1 = t1 == kGeomBox, z1 = ax1 || t1 == kGeomCylinder || t1 == kGeomCapsule;
  const bool ax2 = t2 == kGeomBox, z2 = ax2 || t2 == kGeomCylinder || t2 == kGeomCapsule;
  int ka = -1, kb = -1;
  {
    const double d0 = fabs(n0.x), d1 = fabs(n0.y), d2 = fabs(n0.z), d3 = fabs(dot(n0, c0)), d4 = fabs(dot(n0, c1)), d5 = fabs(dot(n0, c2));
    auto take = [&](bool has, double d, int i) { if (has && d < kKinkTol) { if (ka < 0) ka = i; else if (kb < 0) kb = i; } };
    take(ax1, d0, 0); take(ax1, d1, 1); take(z1, d2, 2); take(ax2, d3, 3); take(ax2, d4, 4); take(z2, d5, 5);
  }
  const bool pole1 = t1 == kGeomCylinder && n0.x * n0.x + n0.y * n0.y < kKinkTol * kKinkTol;
  const V3 cr2 = cross(n0, c2);
  const bool pole2 = t2 == kGeomCylinder && dot(cr2, cr2) < kKinkTol * kKinkTol;
  // candidates: 0 pole of 1, 1 pole of 2, 2 both near planes, 3 plane a, 4 plane b, 5 none
#pragma nounroll
  for (int c = 0; c < 6; ++c) {
    int act_a = -1, act_b = -1, pole_of = 0;
    V3 n = n0;
    bool done = true;                                         // the direction is determined (no Newton)
    if (c == 0) { if (!pole1) continue; n = V3{0.0, 0.0, n0.z >= 0.0 ? 1.0 : -1.0}; pole_of = 1; }
    else if (c == 1) { if (!pole2) continue; n = (dot(n0, c2) >= 0.0 ? 1.0 : -1.0) * c2; pole_of = 2; }
    else if (c == 2) {
      if (kb < 0) continue;
      const V3 x = cross(kink(ka), kink(kb));
      const double l2 = dot(x, x);
      if (!(l2 > 1e-12)) continue;
      n = ((dot(x, n0) >= 0.0 ? 1.0 : -1.0) * cvx_rsqrt(l2)) * x;
      act_a = ka; act_b = kb;
    } else {
      if (c == 3) { if (ka < 0) continue; act_a = ka; }
      if (c == 4) { if (kb < 0) continue; act_a = kb; }
      done = false;
    }
    const bool one = !done && act_a >= 0;
    const V3 k = one ? kink(act_a) : V3{0, 0, 0};
    if (one) n = n - dot(n, k) * k;
    if (!done) n = cvx_rsqrt(dot(n, n)) * n;
    bool good = true;
    V3 a_s{0, 0, 0}, b_l{0, 0, 0}, b_s{0, 0, 0};
    double h = 0.0;
#pragma nounroll
    for (int it = 0;; ++it) {
      // the ONE evaluation site of the support mappings (b_l: the point of shape 2 in its own frame)
      a_s = cvx_support_local(t1, s1, nullptr, 0, n);
      b_l = cvx_support_local(t2, s2, nullptr, 0, mulT(R21, -1.0 * n));
      b_s = p21 + mul(R21, b_l);
      const V3 g = a_s - b_s;
      h = dot(n, g);
      if (done) break;
      if (it == 8) { good = false; break; }
      // Newton on the smooth piece: tangent basis, reduced Hessian
      V3 ta, tb;
      if (one) { ta = cross(n, k); ta = cvx_rsqrt(dot(ta, ta)) * ta; tb = ta; }
      else {
        const double ax = fabs(n.x), ay = fabs(n.y), az = fabs(n.z);
        const bool ex = ax <= ay && ax <= az, ey = !ex && ay <= az;
        const V3 e{ex ? 1.0 : 0.0, ey ? 1.0 : 0.0, (!ex && !ey) ? 1.0 : 0.0};
        ta = cross(n, e); ta = cvx_rsqrt(dot(ta, ta)) * ta; tb = cross(n, ta);
      }
      double m11 = 0.0, m12 = 0.0, m22 = 0.0;
      bool okh = true;
      cvx_hess_form(t1, s1, n, ta, tb, m11, m12, m22, okh);
      cvx_hess_form(t2, s2, mulT(R21, -1.0 * n), mulT(R21, ta), mulT(R21, tb), m11, m12, m22, okh);
      if (!okh) { good = false; break; }
      m11 -= h; m22 -= h;
      const double thr = 1e-12 * fmax(1.0, fabs(h));
      V3 delta;
      if (one) {
        if (!(m11 > thr)) { good = false; break; }
        delta = (-dot(ta, g) * fast_rcp(m11)) * ta;
      } else {
        // (strict local minimum: both eigenvalues of the 2 × 2 form above the threshold — det > thr·(tr − thr), tr > 2·thr)
        const double tr = m11 + m22, det = m11 * m22 - m12 * m12;
        if (!(tr > 2.0 * thr && det > thr * (tr - thr))) { good = false; break; }
        const double g1 = -dot(ta, g), g2 = -dot(tb, g), id = fast_rcp(det);
        delta = ((m22 * g1 - m12 * g2) * id) * ta + ((m11 * g2 - m12 * g1) * id) * tb;
      }
      n = n + delta;
      if (one) n = n - dot(n, k) * k;
      n = cvx_rsqrt(dot(n, n)) * n;
      // (quadratic convergence: a step below 1e-9 leaves an error below 1e-18 — the next pass only evaluates the certificate's points)
      if (dot(delta, delta) < 1e-18) done = true;
    }
    if (!good) continue;
    // ---- certificate: witnesses from the support sets
    const int n1 = (act_a >= 0 && act_a < 3 ? 1 : 0) + (act_b >= 0 && act_b < 3 ? 1 : 0);
    const int n2 = (act_a >= 3 ? 1 : 0) + (act_b >= 3 ? 1 : 0);
    // 0 point, 1 segment, 2 face
    const int set1 = (pole_of == 1 || n1 == 2) ? 2 : n1, set2 = (pole_of == 2 || n2 == 2) ? 2 : n2;
    V3 a, b;
    if (set1 == 0 && set2 == 0) {
      const V3 r = (a_s - b_s) - h * n;
      if (dot(r, r) > tol * tol) continue;
      a = a_s; b = a - h * n;
    } else if (set1 == 0) {
      a = a_s; b = a - h * n;
      if (!cvx_core_contains(t2, s2, mulT(R21, b - p21), tol)) continue;
    } else if (set2 == 0) {
      b = b_s; a = b + h * n;
      if (!cvx_core_contains(t1, s1, a, tol)) continue;
    } else if (set1 == 1 && set2 == 1) {
      // edge / generator / core segment against the same: the support point with the coordinate along the shape's own active axis
      // at either end (a box edge: ∓ its half-size there; a cylinder's generator or a capsule's core: ∓ the half-length)
      const int j1 = (act_a >= 0 && act_a < 3) ? act_a : act_b, j2 = ((act_a >= 3) ? act_a : act_b) - 3;
      const double e1 = ax1 ? (j1 == 0 ? s1.x : (j1 == 1 ? s1.y : s1.z)) : s1.y, e2 = ax2 ? (j2 == 0 ? s2.x : (j2 == 1 ? s2.y : s2.z)) : s2.y;
      V3 a0 = a_s, bl0 = b_l;
      if (j1 == 0) a0.x = -e1; else if (j1 == 1) a0.y = -e1; else a0.z = -e1;
      if (j2 == 0) bl0.x = -e2; else if (j2 == 1) bl0.y = -e2; else bl0.z = -e2;
      const V3 b0 = p21 + mul(R21, bl0);
      const V3 ea = (2.0 * e1) * kink(j1), eb = (2.0 * e2) * kink(j2 + 3);
      const double maa = dot(ea, ea), mab = -dot(ea, eb), mbb = dot(eb, eb);
      const double det = maa * mbb - mab * mab;
      if (fabs(det) < 1e-12 * maa * mbb) continue;            // parallel: the witness is not unique
      const V3 r = (b0 + h * n) - a0;
      const double r1 = dot(ea, r), r2 = -dot(eb, r), id = fast_rcp(det);
      const double al = (mbb * r1 - mab * r2) * id, be = (maa * r2 - mab * r1) * id;
      a = a0 + al * ea; b = b0 + be * eb;
      const V3 rr = (a - b) - h * n;
      if (!(al >= -1e-9 && al <= 1.0 + 1e-9 && be >= -1e-9 && be <= 1.0 + 1e-9) || dot(rr, rr) > tol * tol) continue;
      b = a - h * n;
    } else {
      continue;
    }
    out.ok = true; out.h = h; out.a = a; out.b = b; out.n = n;
    return out;
  }
  return out;
}

// One contact in mj_geomDistance's convention: n from geom 1 to geom 2, pos the midpoint of the witness points — in the frame
// of geom 1 (the caller rotates pos / nrm back).
// need_epa: the cores overlap — the caller runs cvx_epa for this pair at wave level (cvx_overlap_contact finishes the contact).
// Round 6: GJK first runs LOOSE (support gap kLooseGap instead of 1e-14: half the support evaluations on a curved rim, where it
// converges linearly) — the polish finishes its answer exactly and its certificate says so; without one (mesh hulls, a witness that
// is not unique) the tight run follows, and its answer stands if the polish has none for it either.  oracle/gjk.py convex_distance.
constexpr double kLooseGap = 1e-6, kLooseEpa = 1e-6;
__device__ __forceinline__ bool cvx_distance(const ConvexRel& g, double margin, double& dist, V3& pos, V3& nrm, bool& need_epa,
                                             double* gjk_slot) {
  const double r1 = (g.t1 == kGeomSphere || g.t1 == kGeomCapsule) ? g.s1.x : 0.0, r2 = (g.t2 == kGeomSphere || g.t2 == kGeomCapsule) ? g.s2.x : 0.0;
#ifdef MKH_NO_POLISH
  const bool loose = false;                         // (A/B builds: the raw GJK / expanding-polytope answers)
#else
  const bool loose = g.t1 != kGeomMesh && g.t2 != kGeomMesh;
#endif
#pragma nounroll
  for (int pass = loose ? 0 : 1; pass < 2; ++pass) {
    const bool tight = pass != 0;
    double dc = 0.0;
    V3 pa{0, 0, 0}, pb{0, 0, 0};
    const bool apart = cvx_gjk(g, margin + r1 + r2, dc, pa, pb, gjk_slot, tight ? 1e-14 : kLooseGap, tight ? kGjkProgress : kLooseGap);
    if (!(apart && dc > 1e-9)) break;               // (cores apart: also when only the spherical shells overlap)
    if (dc - r1 - r2 > margin * (1.0 + 1e-3) + 1e-6) return false;     // (beyond the margin by more than a loose run can be off)
    nrm = (1.0 / dc) * (pb - pa);
#ifdef MKH_NO_POLISH
    const CvxPolish pl{false, 0.0, {0, 0, 0}, {0, 0, 0}, {1, 0, 0}};
#else
    // witness points on the exact features (cvx_polish above): same distance, certified
    const CvxPolish pl = cvx_polish(g.t1, g.s1, g.t2, g.s2, g.q21, g.p21, nrm);
#endif
    if (pl.ok && -pl.h > 0.0 && fabs(-pl.h - dc) <= (tight ? 1e-6 : 1e-3) * fmax(dc, 1e-3)) {
      dc = -pl.h; pa = pl.a; pb = pl.b; nrm = pl.n;
    } else if (!tight) {
      continue;                                     // no certificate for the loose answer: the tight run
    }
    dist = dc - r1 - r2;
    if (dist > margin) return false;
    pos = 0.5 * ((pa + r1 * nrm) + (pb - r2 * nrm));
    return true;
  }
  need_epa = true;
  dist = 0.0; nrm = {1.0, 0.0, 0.0}; pos = {0.0, 0.0, 0.0};
  return true;
}

// the contact of an overlapping pair from the expanding polytope's answer: the deepest points a − b = depth·n
// (the witness-point polish of such a pair runs on the pair's own lane afterwards — collide_dev.h geom_overlap_polish: inlined here,
//  where all 64 lanes execute, it cost the `ur5e_convex` launch 0.1 ms of tail: the callee around it saves every register it touches)
__device__ __forceinline__ void cvx_overlap_contact(const ConvexGeom& g1, const ConvexGeom& g2, const CvxEpa& e, double& dist, V3& pos, V3& nrm) {
  const double r1 = cvx_core_radius(g1), r2 = cvx_core_radius(g2);
  dist = -(e.depth + r1 + r2);
  nrm = e.n;
  pos = 0.5 * ((e.a + r1 * e.n) + (e.b - r2 * e.n));
}

}  // namespace mkh
