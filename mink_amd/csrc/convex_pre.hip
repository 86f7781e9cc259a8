// convex_contacts_kernel — the general convex distance routine in a kernel of its own, in front of a plain solve.
//
// mj_geomDistance (mink/limits/collision_avoidance_limit.py:219) of the geom pairs that have no analytic routine — cylinder–box,
// cylinder–cylinder, ellipsoid–*, every pair with a mesh hull — goes through GJK and, for overlapping shapes, the expanding polytope
// (convex_dev.h).  Inside the solve kernel that routine was a callee of a kernel with a pinned tableau: its prologue wrote
// callee-saved VGPR blocks to scratch for every problem — 182 MB of HBM traffic per 4 096-instance launch of `ur5e_convex`
// against 0.64 MB of algorithmic bytes (round-4 review).  Here one LANE takes one (instance, convex pair): the poses of the two
// bodies by a walk down their kinematic chains (mj_kinematics restricted to the chain), the distance, and (dist, from, to) to
// device memory — 56 B per pair and instance —, which the ANALYTIC collision build of the solve kernel picks up as a
// pre-evaluated contact (ik_kernel.h contact_of, CollisionPairDev::cv_slot).  Fused loops and calls with taps keep the
// in-kernel routine: their contacts move with q inside the launch.
#include <hip/hip_runtime.h>

#include "mkh_types.h"
#include "wave_ops.h"
#include "collide_dev.h"
#include "wide_types.h"

namespace mkh {

struct CvPre {
  int32_t n_cv;
  const int32_t* pair;       // [n_cv] index into WideProblem::pairs
  const int32_t* chain_adr;  // [2·n_cv + 1]: bodies root → body1 of convex pair k at [chain_adr[2k], chain_adr[2k+1]), → body2 behind
  const int32_t* chain;      // body ids
};

// pose of the last body of a chain (root first) at qrow: mj_kinematics along one branch (the arithmetic of wide_kernel.h's FK)
__device__ __forceinline__ void cv_chain_pose(const WideProblem& P, const double* qrow, const int32_t* ch, int len, V3& xp, Q4& xq) {
  xp = V3{0, 0, 0};
  xq = Q4{1, 0, 0, 0};
  for (int c = 0; c < len; ++c) {
    const int b = ch[c];
    xp = xp + qrot(xq, V3{P.body_pos[3 * b], P.body_pos[3 * b + 1], P.body_pos[3 * b + 2]});
    xq = qmul(xq, Q4{P.body_quat[4 * b], P.body_quat[4 * b + 1], P.body_quat[4 * b + 2], P.body_quat[4 * b + 3]});
    const int jadr = P.body_jntadr[b], jnum = P.body_jntnum[b];
    for (int jn = 0; jn < jnum; ++jn) {
      const int j = jadr + jn, jt = P.jnt_type[j], qa = P.jnt_qadr[j];
      const V3 axl{P.jnt_axis[3 * j], P.jnt_axis[3 * j + 1], P.jnt_axis[3 * j + 2]};
      const V3 jp{P.jnt_pos[3 * j], P.jnt_pos[3 * j + 1], P.jnt_pos[3 * j + 2]};
      if (jt == JNT_FREE) {
        xp = {qrow[qa], qrow[qa + 1], qrow[qa + 2]};
        xq = qnormalize(Q4{qrow[qa + 3], qrow[qa + 4], qrow[qa + 5], qrow[qa + 6]});
      }
      const V3 ax = qrot(xq, axl), an = xp + qrot(xq, jp);
      if (jt == JNT_SLIDE) {
        xp = xp + (qrow[qa] - P.jnt_qpos0[j]) * ax;
      } else if (jt == JNT_HINGE || jt == JNT_BALL) {
        const Q4 qloc = (jt == JNT_HINGE) ? axis_angle(axl, qrow[qa] - P.jnt_qpos0[j])
                                          : qnormalize(Q4{qrow[qa], qrow[qa + 1], qrow[qa + 2], qrow[qa + 3]});
        xq = qmul(xq, qloc);
        xp = an - qrot(xq, jp);
      }
    }
    xq = qnormalize(xq);
  }
}

// `ipw` items per wavefront (lanes [0, ipw) work): the routine is one long dependent chain per lane and a wavefront lasts as long as its
// slowest lane (and runs its overlapping pairs one after the other), so a small batch is spread over MANY thin wavefronts —
// 4 096 items on 64 full wavefronts took 0.33 ms, on 2 048 wavefronts of two lanes 0.03 ms
__global__ __launch_bounds__(64) void convex_contacts_kernel(const WideProblem* __restrict__ Pg, CvPre C, int B, int ipw, const double* __restrict__ q,
                                                             double* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) double ws[kEpaWsDoubles > 2 * kGjkWsDoubles ? kEpaWsDoubles : 2 * kGjkWsDoubles];
  const WideProblem& P = *Pg;
  const int lane = lane_id();
  const long long base = (long long)blockIdx.x * ipw, total = (long long)B * C.n_cv;
  auto poses = [&](long long item, const CollisionPairDev*& cpp, V3& gp1, Q4& gq1, V3& gp2, Q4& gq2) {
    const int b = (int)(item / C.n_cv), k = (int)(item - (long long)b * C.n_cv);
    const CollisionPairDev& cp = P.pairs[C.pair[k]];
    cpp = &cp;
    const double* qrow = q + (size_t)b * P.nq;
    V3 xp1, xp2; Q4 xq1, xq2;
    cv_chain_pose(P, qrow, C.chain + C.chain_adr[2 * k], C.chain_adr[2 * k + 1] - C.chain_adr[2 * k], xp1, xq1);
    cv_chain_pose(P, qrow, C.chain + C.chain_adr[2 * k + 1], C.chain_adr[2 * k + 2] - C.chain_adr[2 * k + 1], xp2, xq2);
    gp1 = xp1 + qrot(xq1, V3{cp.lpos1[0], cp.lpos1[1], cp.lpos1[2]});
    gp2 = xp2 + qrot(xq2, V3{cp.lpos2[0], cp.lpos2[1], cp.lpos2[2]});
    gq1 = qmul(xq1, Q4{cp.lquat1[0], cp.lquat1[1], cp.lquat1[2], cp.lquat1[3]});
    gq2 = qmul(xq2, Q4{cp.lquat2[0], cp.lquat2[1], cp.lquat2[2], cp.lquat2[3]});
  };
  const long long item = base + lane;
  const bool want = lane < ipw && item < total;
  double dist = 0.0;
  V3 from{0, 0, 0}, to{0, 0, 0};
  bool need_epa = false;
  if (want) {
    const CollisionPairDev* cp; V3 gp1, gp2; Q4 gq1, gq2;
    poses(item, cp, gp1, gq1, gp2, gq2);
    dist = cp->ddetect;
    // (two GJK banks: every lane of this kernel holds a convex pair)
    geom_distance<false, true>(cp->type1, V3{cp->size1[0], cp->size1[1], cp->size1[2]}, gp1, gq1, cp->type2,
                               V3{cp->size2[0], cp->size2[1], cp->size2[2]}, gp2, gq2, cp->ddetect, dist, from, to,
                               cp->vert1, cp->nvert1, cp->vert2, cp->nvert2, &need_epa, ws + (lane >> 5) * kGjkWsDoubles + (lane & 31));
  }
  // pairs whose cores overlap: one at a time, the wavefront cooperating on the expanding polytope
  for (unsigned long long em = __ballot(want && need_epa); em; em &= em - 1) {
    const int l = (int)__builtin_ctzll(em);
    const CollisionPairDev* cp; V3 gp1, gp2; Q4 gq1, gq2;
    poses(base + l, cp, gp1, gq1, gp2, gq2);
    // (loose polytope + polish on the pair's lane; without a certificate the tight polytope — collide_dev.h geom_overlap_polish)
    const V3 sz1{cp->size1[0], cp->size1[1], cp->size1[2]}, sz2{cp->size2[0], cp->size2[1], cp->size2[2]};
    bool certified = false;
#pragma nounroll
    for (int pass = geom_overlap_loose(cp->type1, cp->type2) ? 0 : 1; pass < 2 && !certified; ++pass) {
      double d_e; V3 f_e, t_e;
      geom_overlap_distance(cp->type1, sz1, gp1, gq1, cp->type2, sz2, gp2, gq2, d_e, f_e, t_e, cp->vert1, cp->nvert1, cp->vert2, cp->nvert2, ws,
                            pass ? kEpaTol : kLooseEpa);
      bool ok = false;
      if (lane == l) {
        dist = d_e; from = f_e; to = t_e;
        ok = geom_overlap_polish(cp->type1, sz1, gp1, gq1, cp->type2, sz2, gp2, gq2, dist, from, to, pass != 0);
      }
      certified = __ballot(ok) != 0;
    }
  }
  if (want) {
    double* o = out + (size_t)item * 7;
    o[0] = dist; o[1] = from.x; o[2] = from.y; o[3] = from.z; o[4] = to.x; o[5] = to.y; o[6] = to.z;
  }
}

// returns 0 or the HIP error of the launch
int launch_convex_pre(hipStream_t stream, const WideProblem* P, const CvPre& C, int B, const double* q, double* out) {
  const long long total = (long long)B * C.n_cv;
  // ≈ 4 096 wavefronts when the batch allows (2 resident per SIMD x 1 024 SIMDs, twice over), at most 64 items each
  long long ipw = (total + 4095) / 4096;
  if (ipw < 1) ipw = 1;
  if (ipw > 64) ipw = 64;
  hipLaunchKernelGGL(convex_contacts_kernel, dim3((unsigned)((total + ipw - 1) / ipw)), dim3(64), 0, stream, P, C, B, (int)ipw, q, out);
  return (int)hipGetLastError();
}

}  // namespace mkh
