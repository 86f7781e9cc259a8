// Narrow-phase signed distance for CollisionAvoidanceLimit half-spaces.
//
// Replaces mujoco.mj_geomDistance as called at
// mink/limits/collision_avoidance_limit.py:214-229 for the analytic primitive pairs
// (plane/sphere/capsule, box against plane/sphere/capsule/box, cylinder against plane/sphere/capsule);
// semantics per MuJoCo's mj_geomDistance: geoms are ordered so that type1 <= type2, the contact
// with the smallest distance within `distmax` wins, and fromto = pos ∓ ½·dist·n is returned in
// the caller's geom order.  The box / cylinder routines state the exact Euclidean distance between
// the two convex shapes (what MuJoCo's native routines compute for separated geoms); where the closest
// pair is not unique (an edge parallel to a face) the tie rule is ours and is documented at the routine.
#pragma once
#include "lie_dev.h"
#include "convex_dev.h"

namespace mkh {

enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_ELLIPSOID = 4, GEOM_CYLINDER = 5, GEOM_BOX = 6, GEOM_MESH = 7 };

struct Contact { double dist; V3 pos; V3 n; bool hit; };

__device__ __forceinline__ Contact sphere_sphere(V3 p1, double r1, V3 p2, double r2, double margin) {
  Contact c; c.hit = false; c.dist = margin; c.pos = p1; c.n = {1, 0, 0};
  V3 dif = p2 - p1;
  double cdist = sqrt(dot(dif, dif));
  double dist = cdist - r1 - r2;
  if (dist > margin) return c;
  V3 n = (cdist < 1e-15) ? V3{1.0, 0.0, 0.0} : (1.0 / cdist) * dif;
  c.hit = true; c.dist = dist; c.n = n; c.pos = p1 + (r1 + 0.5 * dist) * n;
  return c;
}

__device__ __forceinline__ Contact better(Contact a, Contact b) {
  // field-wise selects: returning one of two structs by value makes hipcc select between two
  // stack addresses, i.e. a real scratch array
  const bool tb = b.hit && (!a.hit || b.dist < a.dist);
  Contact r;
  r.hit = a.hit || b.hit;
  r.dist = tb ? b.dist : a.dist;
  r.pos = {tb ? b.pos.x : a.pos.x, tb ? b.pos.y : a.pos.y, tb ? b.pos.z : a.pos.z};
  r.n = {tb ? b.n.x : a.n.x, tb ? b.n.y : a.n.y, tb ? b.n.z : a.n.z};
  return r;
}

__device__ __forceinline__ Contact capsule_capsule(V3 pos1, V3 axis1, double r1, double l1,
                                                   V3 pos2, V3 axis2, double r2, double l2, double margin) {
  V3 dif = pos1 - pos2;
  double ma = dot(axis1, axis1), mb = -dot(axis1, axis2), mc = dot(axis2, axis2);
  double u = -dot(axis1, dif), v = dot(axis2, dif);
  double det = ma * mc - mb * mb;
  Contact best; best.hit = false; best.dist = margin; best.pos = pos1; best.n = {1, 0, 0};
  if (fabs(det) >= 1e-15) {
    double x1 = (mc * u - mb * v) / det;
    double x2 = (ma * v - mb * u) / det;
    if (x1 > l1) { x1 = l1; x2 = (v - mb * l1) / mc; }
    else if (x1 < -l1) { x1 = -l1; x2 = (v + mb * l1) / mc; }
    if (x2 > l2) { x2 = l2; x1 = (u - mb * l2) / ma; }
    else if (x2 < -l2) { x2 = -l2; x1 = (u + mb * l2) / ma; }
    x1 = fmin(fmax(x1, -l1), l1);
    x2 = fmin(fmax(x2, -l2), l2);
    best = sphere_sphere(pos1 + x1 * axis1, r1, pos2 + x2 * axis2, r2, margin);
  } else {
    int n = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double x1 = k ? -l1 : l1;
      double x2 = (v - mb * x1) / mc;
      if (x2 >= -l2 && x2 <= l2 && n < 2) { best = better(best, sphere_sphere(pos1 + x1 * axis1, r1, pos2 + x2 * axis2, r2, margin)); ++n; }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double x2 = k ? -l2 : l2;
      double x1 = (u - mb * x2) / ma;
      if (x1 >= -l1 && x1 <= l1 && n < 2) { best = better(best, sphere_sphere(pos1 + x1 * axis1, r1, pos2 + x2 * axis2, r2, margin)); ++n; }
    }
  }
  return best;
}

__device__ __forceinline__ Contact plane_sphere(V3 ppos, V3 pn, V3 c, double r, double margin) {
  Contact k; k.hit = false; k.dist = margin; k.pos = c; k.n = pn;
  double dist = dot(pn, c - ppos) - r;
  if (dist > margin) return k;
  k.hit = true; k.dist = dist; k.pos = c - (r + 0.5 * dist) * pn;
  return k;
}


// ---- box / cylinder routines: everything in the frame of the box (cylinder), mapped back with its rotation R
__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// plane–box (mjc_PlaneBox): the corner furthest below the centre carries the smallest distance;
// a face/edge parallel to the plane ties towards the −size corner (MuJoCo's enumeration order)
__device__ __forceinline__ Contact plane_box(V3 ppos, V3 pn, V3 bpos, const M3& R, V3 s, double margin) {
  Contact k; k.hit = false; k.dist = margin; k.pos = bpos; k.n = pn;
  const V3 nb = mulT(R, pn);
  const V3 vec{nb.x < 0.0 ? s.x : -s.x, nb.y < 0.0 ? s.y : -s.y, nb.z < 0.0 ? s.z : -s.z};
  const double dist = dot(pn, bpos - ppos) + dot(nb, vec);
  if (dist > margin) return k;
  k.hit = true; k.dist = dist; k.pos = bpos + mul(R, vec) - (0.5 * dist) * pn;
  return k;
}

// plane–cylinder (mjc_PlaneCylinder): lowest rim point; a cap parallel to the plane ties to the cap centre
__device__ __forceinline__ Contact plane_cylinder(V3 ppos, V3 pn, V3 cpos, V3 axis, double rad, double half,
                                                  double margin) {
  Contact k; k.hit = false; k.dist = margin; k.pos = cpos; k.n = pn;
  const double c = dot(pn, axis);
  V3 radial = pn - c * axis;                               // plane normal projected on the cap plane
  const double rl = sqrt(dot(radial, radial));
  V3 pt = cpos - ((c < 0.0 ? -half : half)) * axis;
  if (rl > 1e-15) pt = pt - (rad / rl) * radial;
  const double dist = dot(pn, pt - ppos);
  if (dist > margin) return k;
  k.hit = true; k.dist = dist; k.pos = pt - (0.5 * dist) * pn;
  return k;
}

// a point with a radius (sphere centre / capsule axis point) against a box: p is in the box frame
// (mjc_SphereBox: clamp onto the box; centre inside ⇒ leave through the nearest face)
__device__ __forceinline__ Contact ball_box_local(V3 p, double r, V3 s, double margin) {
  Contact k; k.hit = false; k.dist = margin; k.pos = p; k.n = {1, 0, 0};
  const V3 cl{clampd(p.x, -s.x, s.x), clampd(p.y, -s.y, s.y), clampd(p.z, -s.z, s.z)};
  const V3 d = cl - p;                                     // from the ball towards the box
  const double dl = sqrt(dot(d, d));
  if (dl - r > margin) return k;
  k.hit = true;
  if (dl > 1e-15) {
    k.dist = dl - r;
    k.n = (1.0 / dl) * d;
    k.pos = p + (r + 0.5 * k.dist) * k.n;
    return k;
  }
  const double fx = s.x - fabs(p.x), fy = s.y - fabs(p.y), fz = s.z - fabs(p.z);
  // nearest face in MuJoCo's order (−x, +x, −y, +y, −z, +z; strict <)
  double closest = fx; V3 n{p.x < 0.0 ? 1.0 : -1.0, 0.0, 0.0};
  if (p.x == 0.0) n.x = 1.0;
  if (fy < closest) { closest = fy; n = {0.0, p.y <= 0.0 ? 1.0 : -1.0, 0.0}; }
  if (fz < closest) { closest = fz; n = {0.0, 0.0, p.z <= 0.0 ? 1.0 : -1.0}; }
  k.dist = -closest - r;
  k.n = n;
  k.pos = p + (0.5 * (r - closest)) * n;
  return k;
}

// capsule–box: exact closest point of the segment c + t·a (|t| ≤ l) to the box.  ½·dist² is convex in t with
// derivative g(t) = Σ aᵢ·eᵢ(t) (eᵢ = excess of coordinate i over the slab ±sᵢ), piecewise linear and
// non-decreasing with breakpoints where the axis crosses a slab plane: bracket the root between the
// breakpoints and interpolate.  A flat stretch (axis parallel to a face, or through the box) takes its midpoint.
// (`on` = the axis whose slab plane t lies on: its excess is 0 by construction, not t·a + c − s up to rounding —
// inside the box g must be exactly 0 at the entry and exit points, or rounding noise picks the point of the chord)
__device__ __forceinline__ double seg_box_slope(V3 c, V3 a, V3 s, double t, int on = -1) {
  const double x = fma(t, a.x, c.x), y = fma(t, a.y, c.y), z = fma(t, a.z, c.z);
  const double ex = (on == 0) ? 0.0 : x - clampd(x, -s.x, s.x);
  const double ey = (on == 1) ? 0.0 : y - clampd(y, -s.y, s.y);
  const double ez = (on == 2) ? 0.0 : z - clampd(z, -s.z, s.z);
  return a.x * ex + a.y * ey + a.z * ez;
}
// A slope below 1e-13 of the problem's size counts as zero: an axis "parallel" to a face up to rounding (a component of
// 1e-16 from the quaternion products) must land on the flat-stretch rule, not on whichever end the noise favours
// (tests/test_gpu_collision_shapes.py::test_tie_rule_is_pinned).
__device__ __forceinline__ double seg_box_flat(double g, double tol) { return fabs(g) <= tol ? 0.0 : g; }
__device__ __forceinline__ Contact capsule_box_local(V3 c, V3 a, double r, double l, V3 s, double margin) {
  const double tol = 1e-13 * (l + fmax(fmax(fabs(c.x), fabs(c.y)), fabs(c.z)) + fmax(fmax(s.x, s.y), s.z));
  const double g0 = seg_box_flat(seg_box_slope(c, a, s, -l), tol), g1 = seg_box_flat(seg_box_slope(c, a, s, l), tol);
  double t;
  if (g0 > 0.0) t = -l;
  else if (g1 < 0.0) t = l;
  else {
    // tL = last candidate with g ≤ 0, tR = first with g ≥ 0 (they cross over on a flat stretch)
    double tL = (g1 <= 0.0) ? l : -l, gL = (g1 <= 0.0) ? g1 : g0;
    double tR = (g0 >= 0.0) ? -l : l, gR = (g0 >= 0.0) ? g0 : g1;
    const double av[3] = {a.x, a.y, a.z}, cv[3] = {c.x, c.y, c.z}, sv[3] = {s.x, s.y, s.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (fabs(av[i]) < 1e-15) continue;
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        const double tb = ((sg ? sv[i] : -sv[i]) - cv[i]) / av[i];
        if (!(tb > -l && tb < l)) continue;
        const double gb = seg_box_flat(seg_box_slope(c, a, s, tb, i), tol);
        if (gb <= 0.0 && tb > tL) { tL = tb; gL = gb; }
        if (gb >= 0.0 && tb < tR) { tR = tb; gR = gb; }
      }
    }
    const double dg = gR - gL;
    t = (dg > 0.0) ? tL + (tR - tL) * (-gL / dg) : 0.5 * (tL + tR);
  }
  return ball_box_local(c + t * a, r, s, margin);
}

// sphere–cylinder (mjc_SphereCylinder): p in the cylinder frame; side / cap / rim, centre inside ⇒ nearest exit
__device__ __forceinline__ Contact ball_cylinder_local(V3 p, double r, double rad, double half, double margin) {
  Contact k; k.hit = false; k.dist = margin; k.pos = p; k.n = {1, 0, 0};
  const double rho = sqrt(p.x * p.x + p.y * p.y);
  const double sc = (rho > rad) ? rad / rho : 1.0;
  const V3 cl{p.x * sc, p.y * sc, clampd(p.z, -half, half)};
  const V3 d = cl - p;
  const double dl = sqrt(dot(d, d));
  if (dl - r > margin) return k;
  k.hit = true;
  if (dl > 1e-15) {
    k.dist = dl - r;
    k.n = (1.0 / dl) * d;
    k.pos = p + (r + 0.5 * k.dist) * k.n;
    return k;
  }
  const double fr = rad - rho, fz = half - fabs(p.z);
  double closest; V3 n;
  if (fz < fr) { closest = fz; n = {0.0, 0.0, p.z <= 0.0 ? 1.0 : -1.0}; }
  else { closest = fr; n = (rho > 1e-15) ? V3{-p.x / rho, -p.y / rho, 0.0} : V3{-1.0, 0.0, 0.0}; }
  k.dist = -closest - r;
  k.n = n;
  k.pos = p + (0.5 * (r - closest)) * n;
  return k;
}

// ---- box–box (mjc_BoxBox).  Separated boxes: the exact Euclidean distance as the minimum over vertex–box (both
// ways) and edge–edge pairs; overlapping boxes (no separating axis among the 15 of the SAT): the smallest overlap
// and its axis, one contact midway (MuJoCo clips faces and reports several — documented approximation).
// Everything in B's frame: A has centre c, axes = columns of R, half sizes sa.  Loops are kept rolled: 144 edge
// pairs as straight-line code would double the size of every collision-capable variant.
__device__ __forceinline__ double comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
__device__ __forceinline__ V3 col(const M3& R, int i) { return {R.m[i], R.m[3 + i], R.m[6 + i]}; }
__device__ __forceinline__ V3 unit(int i) { return {i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0}; }
__device__ __forceinline__ V3 vabs(V3 v) { return {fabs(v.x), fabs(v.y), fabs(v.z)}; }
__device__ __forceinline__ V3 vmulc(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 vclamp(V3 v, V3 s) { return {clampd(v.x, -s.x, s.x), clampd(v.y, -s.y, s.y), clampd(v.z, -s.z, s.z)}; }
// endpoints of edge number e (0..11) of the box ±s: axis e/4, the sign choices of the two other axes from e%4
__device__ __forceinline__ void box_edge(V3 s, int e, V3& a, V3& b) {
  const int ax = e >> 2, j = (ax + 1) % 3, k = (ax + 2) % 3;
  const double sj = (e & 1) ? comp(s, j) : -comp(s, j), sk = (e & 2) ? comp(s, k) : -comp(s, k);
  double av[3], bv[3];
  av[ax] = -comp(s, ax); bv[ax] = comp(s, ax);
  av[j] = bv[j] = sj;
  av[k] = bv[k] = sk;
  a = {av[0], av[1], av[2]};
  b = {bv[0], bv[1], bv[2]};
}
__device__ __forceinline__ Contact box_box_local(V3 c, M3 R, V3 sa, V3 sb, double margin) {
  Contact k; k.hit = false; k.dist = margin; k.pos = c; k.n = {1, 0, 0};
  // ---- separating-axis test
  double best_sep = -1e300;
  V3 best_axis{1, 0, 0};
#pragma unroll 1
  for (int i = 0; i < 15; ++i) {
    V3 L;
    if (i < 3) L = unit(i);
    else if (i < 6) L = col(R, i - 3);
    else {
      L = cross(col(R, (i - 6) / 3), unit((i - 6) % 3));
      const double n2 = dot(L, L);
      if (n2 < 1e-18) continue;
      L = (1.0 / sqrt(n2)) * L;
    }
    const double ra = dot(vabs(mulT(R, L)), sa), rb = dot(vabs(L), sb), cl = dot(c, L);
    const double sep = fabs(cl) - (ra + rb);
    if (sep > best_sep) { best_sep = sep; best_axis = (cl <= 0.0) ? L : -1.0 * L; }   // from A towards B
  }
  if (best_sep <= 0.0) {
    if (best_sep > margin) return k;
    const V3 n = best_axis;
    const double cn = dot(c, n);
    const double ca = cn + dot(vabs(mulT(R, n)), sa), cb = -dot(vabs(n), sb);
    k.hit = true; k.dist = best_sep; k.n = n;
    k.pos = 0.5 * c + (0.5 * (ca + cb) - 0.5 * cn) * n;
    return k;
  }
  // ---- separated: closest features
  double best = 1e300;
  V3 pa{0, 0, 0}, pb{0, 0, 0};
#pragma unroll 1
  for (int i = 0; i < 8; ++i) {
    const V3 sg{(i & 1) ? 1.0 : -1.0, (i & 2) ? 1.0 : -1.0, (i & 4) ? 1.0 : -1.0};
    const V3 va = c + mul(R, vmulc(sg, sa));                  // vertex of A against box B
    V3 cl = vclamp(va, sb);
    double d2 = dot(va - cl, va - cl);
    if (d2 < best) { best = d2; pa = va; pb = cl; }
    const V3 ub = vmulc(sg, sb);                              // vertex of B against box A (A's frame)
    const V3 ul = mulT(R, ub - c);
    cl = vclamp(ul, sa);
    d2 = dot(ul - cl, ul - cl);
    if (d2 < best) { best = d2; pa = c + mul(R, cl); pb = ub; }
  }
#pragma unroll 1
  for (int ea = 0; ea < 12; ++ea) {
    V3 a0, a1;
    box_edge(sa, ea, a0, a1);
    const V3 p1 = c + mul(R, a0), d1 = mul(R, a1 - a0);
    const double a = dot(d1, d1);
#pragma unroll 1
    for (int eb = 0; eb < 12; ++eb) {
      V3 p2, q2;
      box_edge(sb, eb, p2, q2);
      const V3 d2v = q2 - p2, r = p1 - p2;
      const double e = dot(d2v, d2v), f = dot(d2v, r), cc = dot(d1, r), b = dot(d1, d2v);
      const double den = a * e - b * b;
      double sp = (den > 1e-15 * a * e) ? clampd((b * f - cc * e) / den, 0.0, 1.0) : 0.0;
      double t = (b * sp + f) / e;
      if (t < 0.0) { t = 0.0; sp = clampd(-cc / a, 0.0, 1.0); }
      else if (t > 1.0) { t = 1.0; sp = clampd((b - cc) / a, 0.0, 1.0); }
      const V3 x1 = p1 + sp * d1, x2 = p2 + t * d2v;
      const double dd = dot(x1 - x2, x1 - x2);
      if (dd < best) { best = dd; pa = x1; pb = x2; }
    }
  }
  const double dist = sqrt(best);
  if (dist > margin) return k;
  k.hit = true; k.dist = dist;
  k.n = (dist > 1e-15) ? (1.0 / dist) * (pb - pa) : best_axis;
  k.pos = 0.5 * (pa + pb);
  return k;
}

// capsule–cylinder (MuJoCo: libccd).  The squared distance from the capsule axis c + t·a to the solid cylinder is
// convex in t, so g(t) = a·(p(t) − closest(p(t))) is non-decreasing: 64 bisection steps on g give the closest axis
// point to the last bit, then ball-against-cylinder.  Axis through the cylinder: lower end of the g = 0 interval.
__device__ __forceinline__ double seg_cyl_slope(V3 c, V3 a, double rad, double half, double t) {
  const V3 p = c + t * a;
  const double rho = sqrt(p.x * p.x + p.y * p.y);
  const double sc = (rho > rad) ? rad / rho : 1.0;
  return a.x * (p.x - p.x * sc) + a.y * (p.y - p.y * sc) + a.z * (p.z - clampd(p.z, -half, half));
}
__device__ __forceinline__ Contact capsule_cylinder_local(V3 c, V3 a, double r, double l, double rad, double half,
                                                         double margin) {
  double lo = -l, hi = l, t;
  if (seg_cyl_slope(c, a, rad, half, lo) >= 0.0) t = lo;
  else if (seg_cyl_slope(c, a, rad, half, hi) <= 0.0) t = hi;
  else {
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
      const double mid = 0.5 * (lo + hi);
      if (seg_cyl_slope(c, a, rad, half, mid) < 0.0) lo = mid; else hi = mid;
    }
    t = hi;
  }
  return ball_cylinder_local(c + t * a, r, rad, half, margin);
}

// map a contact found in the frame of geom 2 (rotation R, origin o) back to the world
__device__ __forceinline__ Contact to_world(Contact k, const M3& R, V3 o) {
  if (k.hit) { k.pos = o + mul(R, k.pos); k.n = mul(R, k.n); }
  return k;
}

// dist / fromto for one geom pair in the CALLER's order (g1 -> g2).
// type/size/pose are those of g1 and g2 as given; returns false when the pair type
// is not one of the analytic routines above.
// SIMPLE: only planes, spheres and capsules can occur (checked on the host for the whole pair list) — the box and
// cylinder routines are not even compiled in: the capsule-only Shadow-hand variant needs a fraction of the registers
// (62 spilled VGPRs → 0) and a seventh of the instructions of the general collision phase.
// CONVEX: the pair list contains pairs without an analytic routine (cylinder–box, cylinder–cylinder, ellipsoid–*): the
// general convex routine of convex_dev.h is compiled in (its own kernel variants: it costs registers and scratch).
template <bool SIMPLE = false, bool CONVEX = false>
__device__ __forceinline__ bool geom_distance(int t1, V3 s1, V3 p1, Q4 q1, int t2, V3 s2, V3 p2, Q4 q2,
                                              double distmax, double& dist, V3& from, V3& to,
                                              const double* hv1 = nullptr, int hn1 = 0, const double* hv2 = nullptr, int hn2 = 0,
                                              bool* need_epa = nullptr, double* gjk_slot = nullptr) {
  // hv / hn: hull vertices (geom frame) of a mesh geom
  // gjk_slot (general convex pairs): this lane's simplex storage in LDS (convex_dev.h cvx_gjk: kGjkSlots lanes at a time)
  // need_epa (general convex pairs): set when the pair's cores OVERLAP — dist / from / to are then placeholders and the
  // caller finishes the pair at wave level with geom_overlap_distance (the expanding polytope is a wave-cooperative routine)
  const bool flip = t1 > t2;
  if (flip) {
    int ti = t1; t1 = t2; t2 = ti;
    V3 tv = s1; s1 = s2; s2 = tv; tv = p1; p1 = p2; p2 = tv;
    Q4 tq = q1; q1 = q2; q2 = tq;
    const double* th = hv1; hv1 = hv2; hv2 = th; ti = hn1; hn1 = hn2; hn2 = ti;
  }
  M3 R1 = qmat(q1), R2 = qmat(q2);
  V3 z1{R1.m[2], R1.m[5], R1.m[8]}, z2{R2.m[2], R2.m[5], R2.m[8]};
  Contact c;
  if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) {
    c = capsule_capsule(p1, z1, s1.x, s1.y, p2, z2, s2.x, s2.y, distmax);
  } else if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) {
    c = sphere_sphere(p1, s1.x, p2, s2.x, distmax);
  } else if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) {
    double x = fmin(fmax(dot(z2, p1 - p2), -s2.y), s2.y);
    c = sphere_sphere(p1, s1.x, p2 + x * z2, s2.x, distmax);
  } else if (t1 == GEOM_PLANE && t2 == GEOM_SPHERE) {
    c = plane_sphere(p1, z1, p2, s2.x, distmax);
  } else if (t1 == GEOM_PLANE && t2 == GEOM_CAPSULE) {
    c = better(plane_sphere(p1, z1, p2 + s2.y * z2, s2.x, distmax),
               plane_sphere(p1, z1, p2 - s2.y * z2, s2.x, distmax));
  } else if (SIMPLE) {
    dist = distmax; from = {0, 0, 0}; to = {0, 0, 0};
    return false;
  } else if (t1 == GEOM_PLANE && t2 == GEOM_BOX) {
    c = plane_box(p1, z1, p2, R2, s2, distmax);
  } else if (t1 == GEOM_PLANE && t2 == GEOM_CYLINDER) {
    c = plane_cylinder(p1, z1, p2, z2, s2.x, s2.y, distmax);
  } else if (t1 == GEOM_SPHERE && t2 == GEOM_BOX) {
    c = to_world(ball_box_local(mulT(R2, p1 - p2), s1.x, s2, distmax), R2, p2);
  } else if (t1 == GEOM_SPHERE && t2 == GEOM_CYLINDER) {
    c = to_world(ball_cylinder_local(mulT(R2, p1 - p2), s1.x, s2.x, s2.y, distmax), R2, p2);
  } else if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX) {
    c = to_world(capsule_box_local(mulT(R2, p1 - p2), mulT(R2, z1), s1.x, s1.y, s2, distmax), R2, p2);
  } else if (t1 == GEOM_CAPSULE && t2 == GEOM_CYLINDER) {
    c = to_world(capsule_cylinder_local(mulT(R2, p1 - p2), mulT(R2, z1), s1.x, s1.y, s2.x, s2.y, distmax), R2, p2);
  } else if (t1 == GEOM_BOX && t2 == GEOM_BOX) {
    M3 Rab;                                                   // A's axes in B's frame: R2ᵀ·R1
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        Rab.m[3 * i + j] = R2.m[i] * R1.m[j] + R2.m[3 + i] * R1.m[3 + j] + R2.m[6 + i] * R1.m[6 + j];
    c = to_world(box_box_local(mulT(R2, p1 - p2), Rab, s1, s2, distmax), R2, p2);
  } else if (CONVEX && t1 == GEOM_PLANE && t2 == GEOM_ELLIPSOID) {
    // lowest point of the ellipsoid: its support point against the plane normal
    const V3 e = vmulc(s2, mulT(R2, z1));
    const V3 pt = p2 - (1.0 / sqrt(dot(e, e))) * mul(R2, vmulc(s2, e));
    c.dist = dot(z1, pt - p1); c.hit = c.dist <= distmax; c.n = z1; c.pos = pt - (0.5 * c.dist) * z1;
  } else if (CONVEX && t1 == GEOM_PLANE && t2 == GEOM_MESH) {
    // lowest point of the hull: its support point against the plane normal (mjc_PlaneConvex keeps the deepest vertex)
    const ConvexGeom g2{t2, s2, p2, R2, hv2, hn2};
    const V3 pt = cvx_support(g2, -1.0 * z1);
    c.dist = dot(z1, pt - p1); c.hit = c.dist <= distmax; c.n = z1; c.pos = pt - (0.5 * c.dist) * z1;
  } else if (CONVEX && t1 >= GEOM_SPHERE && t1 <= GEOM_MESH && t2 >= GEOM_SPHERE && t2 <= GEOM_MESH) {
    // in the frame of geom 1: pose of geom 2 there, contact rotated back
    const Q4 q1c = qconj(q1);
    const ConvexRel g{t1, s1, hv1, hn1, t2, s2, hv2, hn2, qmul(q1c, q2), qrot(q1c, p2 - p1)};
    bool ne = false;
    c.hit = cvx_distance(g, distmax, c.dist, c.pos, c.n, ne, gjk_slot);
    c.pos = p1 + qrot(q1, c.pos);
    c.n = qrot(q1, c.n);
    if (need_epa) *need_epa = ne;
  } else {
    dist = distmax; from = {0, 0, 0}; to = {0, 0, 0};
    return false;
  }
  if (!c.hit) { dist = distmax; from = {0, 0, 0}; to = {0, 0, 0}; return true; }
  const double sgn = flip ? -1.0 : 1.0;
  dist = c.dist;
  from = c.pos - (0.5 * sgn * c.dist) * c.n;
  to = c.pos + (0.5 * sgn * c.dist) * c.n;
  return true;
}

// Does geom_distance send this pair of types through GJK (no analytic routine; plane–ellipsoid / plane–mesh are analytic)?
__device__ __forceinline__ bool geom_pair_runs_gjk(int a, int b) {
  if (a > b) { const int x = a; a = b; b = x; }
  const bool analytic = (a == GEOM_CAPSULE && b == GEOM_CAPSULE) || (a == GEOM_SPHERE && b == GEOM_SPHERE) ||
                        (a == GEOM_SPHERE && b == GEOM_CAPSULE) || a == GEOM_PLANE ||
                        (b == GEOM_BOX && (a == GEOM_SPHERE || a == GEOM_CAPSULE || a == GEOM_BOX)) ||
                        (b == GEOM_CYLINDER && (a == GEOM_SPHERE || a == GEOM_CAPSULE));
  return !analytic && a >= GEOM_SPHERE && b <= GEOM_MESH;
}

// Second half of geom_distance for a general convex pair whose cores overlap: every lane of the wavefront calls this with
// the SAME (wave-uniform) pair; ws = the expanding polytope's LDS workspace (kEpaWsDoubles).
// tol: the polytope's gap tolerance — kEpaTol, or kLooseEpa for the run in front of the polish (geom_overlap_loose below says when)
__device__ __forceinline__ void geom_overlap_distance(int t1, V3 s1, V3 p1, Q4 q1, int t2, V3 s2, V3 p2, Q4 q2, double& dist, V3& from,
                                                      V3& to, const double* hv1, int hn1, const double* hv2, int hn2, double* ws,
                                                      const double tol = kEpaTol) {
  const bool flip = t1 > t2;
  if (flip) {
    int ti = t1; t1 = t2; t2 = ti;
    V3 tv = s1; s1 = s2; s2 = tv; tv = p1; p1 = p2; p2 = tv;
    Q4 tq = q1; q1 = q2; q2 = tq;
    const double* th = hv1; hv1 = hv2; hv2 = th; ti = hn1; hn1 = hn2; hn2 = ti;
  }
  const ConvexGeom g1{t1, s1, p1, qmat(q1), hv1, hn1}, g2{t2, s2, p2, qmat(q2), hv2, hn2};
  const CvxEpa e = cvx_epa(g1, g2, ws, tol);
  Contact c;
  cvx_overlap_contact(g1, g2, e, c.dist, c.pos, c.n);
  const double sgn = flip ? -1.0 : 1.0;
  dist = c.dist;
  from = c.pos - (0.5 * sgn * c.dist) * c.n;
  to = c.pos + (0.5 * sgn * c.dist) * c.n;
}

// Third part, on the pair's OWN lane (`if (lane == l)` behind the wave-level call): the expanding polytope's witness points onto
// the exact features (convex_dev.h cvx_polish) — a certified stationary point next to its answer, same depth to its tolerance; the
// answer stands when there is no certificate.  dist / from / to as geom_overlap_distance left them.
// Returns whether the answer is certified.  The callers run the polytope LOOSE first (geom_overlap_loose: not for mesh hulls, which
// the polish never certifies) and once more at kEpaTol when this returns false — `tight` says which run's answer comes in.
__device__ __forceinline__ bool geom_overlap_loose(int t1, int t2) {
#ifdef MKH_NO_POLISH
  return false;
#else
  return t1 != GEOM_MESH && t2 != GEOM_MESH;
#endif
}
__device__ __forceinline__ bool geom_overlap_polish(int t1, V3 s1, V3 p1, Q4 q1, int t2, V3 s2, V3 p2, Q4 q2, double& dist, V3& from, V3& to,
                                                    const bool tight = true) {
#ifndef MKH_NO_POLISH
  const bool flip = t1 > t2;
  if (flip) {
    int ti = t1; t1 = t2; t2 = ti;
    V3 tv = s1; s1 = s2; s2 = tv; tv = p1; p1 = p2; p2 = tv;
    Q4 tq = q1; q1 = q2; q2 = tq;
  }
  if (t1 == GEOM_MESH || t2 == GEOM_MESH) return false;
  const double r1 = (t1 == GEOM_SPHERE || t1 == GEOM_CAPSULE) ? s1.x : 0.0, r2 = (t2 == GEOM_SPHERE || t2 == GEOM_CAPSULE) ? s2.x : 0.0;
  const double sgn = flip ? -1.0 : 1.0;
  const V3 dv = to - from;                                   // = sgn·dist·n, n from geom 1 to geom 2 in the sorted order
  const double l2 = dot(dv, dv);
  if (!(l2 > 0.0) || !(dist < 0.0)) return false;
  const V3 n0 = (-sgn * cvx_rsqrt(l2)) * dv;
  const double depth = -dist - r1 - r2;                      // of the cores
  const Q4 q1c = qconj(q1);
  const CvxPolish pl = cvx_polish(t1, s1, t2, s2, qmul(q1c, q2), qrot(q1c, p2 - p1), qrot(q1c, n0));
  // (the same basin: a direction within 0.14 rad of the polytope's and a depth no larger than the polytope's own — which stops on a
  //  vertex budget for doubly curved pairs, a few 1e-4 above the minimum — and within 1 % of it)
  const double dscale = fmax(depth, 1e-3);
  if (pl.ok && dot(pl.n, qrot(q1c, n0)) >= 0.99 && pl.h <= depth + (tight ? 1e-9 : 1e-5) * dscale && pl.h >= depth - 1e-2 * dscale) {
    const V3 n = qrot(q1, pl.n), a = p1 + qrot(q1, pl.a), b = p1 + qrot(q1, pl.b);
    const double d = -(pl.h + r1 + r2);
    const V3 pos = 0.5 * ((a + r1 * n) + (b - r2 * n));
    dist = d;
    from = pos - (0.5 * sgn * d) * n;
    to = pos + (0.5 * sgn * d) * n;
    return true;
  }
#endif
  return false;
}

}  // namespace mkh
