// Narrow-phase signed distance for CollisionAvoidanceLimit half-spaces.
//
// Replaces mujoco.mj_geomDistance as called at
// mink/limits/collision_avoidance_limit.py:214-229 for the analytic primitive pairs
// (plane/sphere/capsule); semantics per MuJoCo's mj_geomDistance: geoms are ordered
// so that type1 <= type2, the contact with the smallest distance within `distmax`
// wins, and fromto = pos ∓ ½·dist·n is returned in the caller's geom order.
#pragma once
#include "lie_dev.h"

namespace mkh {

enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3 };

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

// dist / fromto for one geom pair in the CALLER's order (g1 -> g2).
// type/size/pose are those of g1 and g2 as given; returns false when the pair type
// is not one of the analytic routines above.
__device__ __forceinline__ bool geom_distance(int t1, V3 s1, V3 p1, Q4 q1, int t2, V3 s2, V3 p2, Q4 q2,
                                              double distmax, double& dist, V3& from, V3& to) {
  const bool flip = t1 > t2;
  if (flip) {
    int ti = t1; t1 = t2; t2 = ti;
    V3 tv = s1; s1 = s2; s2 = tv; tv = p1; p1 = p2; p2 = tv;
    Q4 tq = q1; q1 = q2; q2 = tq;
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

}  // namespace mkh
