// One-LANE-per-problem differential-IK kernel for small arms (nv ≤ 8, hinge/slide joints) on gfx950.
//
// The wavefront-per-problem kernel (ik_kernel.h) is built for G1-sized problems: its lanes are tableau columns,
// bodies, dofs.  A 6-dof arm uses 6-10 of the 64 lanes in every phase — the machine runs three-quarters empty
// (round-1 review, "lane utilisation on small robots").  Below nv ≈ 8 the whole problem fits in ONE lane's
// registers: H is 21-36 doubles, one task Jacobian 36-48.  So here lane l of wave w solves problem 64·w + l
// start to finish — forward kinematics down the chain, frame-task error and Jacobian (the same lie_dev.h
// functions the wavefront kernel's task lanes run), H = Σ JᵀW²J + (λ + Σμ)I, box limits, and the box-constrained QP —
// with no cross-lane traffic at all: 64 problems per wavefront, every lane busy in every instruction, divergence only
// in the QP's iteration count.  Same reference path: mink/solve_ik.py:68-105 and what it calls (see ik_kernel.h).
//
// Kinematics are wave-uniform data (every lane has the same robot), so loops over links / dofs / tasks are scalar
// loops with uniform addresses; per-lane link poses and joint axes live in LDS as [item][component][lane] (bank
// conflict free), everything else in registers (loops over dofs are unrolled: NV is a template parameter).
//
// QP: min ½xᵀHx + cᵀx, lo ≤ x ≤ hi with H ≻ 0 — block principal pivoting (Júdice & Pires 1994) on the partition
// free / at-lower / at-upper: solve the free block by a masked Cholesky factorisation (bound rows/columns replaced
// by the identity, so the loops are static), flip every infeasible index while that keeps reducing their number,
// otherwise only the last one (Murty's rule: finite for P-matrices).  The optimum of a strictly convex QP is unique,
// so this gives quadprog's answer (tests/test_gpu_lane_kernel.py: against both oracles and the wavefront kernel).
#pragma once
#include "lie_dev.h"
#include "mkh_types.h"

namespace mkh {

constexpr int kLaneMaxLinks = 16;       // descriptor of the lane kernel and of the row kernel's one-row builds
constexpr int kLaneMaxDofs = 8;         // the lane kernel's own limit (its register arrays)
constexpr int kLaneDescDofs = 16;       // (the row kernel takes up to 16 dofs on one DPP row)
constexpr int kLaneMaxLinks2 = 32, kLaneDescDofs2 = 32;   // descriptor of the row kernel's two-row build (17 … 32 dofs or links)
constexpr int kLaneMaxFrames = 8;       // frame tasks of a problem (round 3: 4 → 8, a hand's fingertips + its palm)

struct LaneLink {
  int32_t parent;        // link index of the parent body, −1 = world
  int32_t jtype;         // −1 none, JNT_SLIDE, JNT_HINGE; JNT_BALL: the rotation of a free joint (two-row build of the row kernel only —
                         // a free joint is three slide links along the world axes with `pos` = 0 and this link on top of them)
  int32_t dof;           // dof / register index of the joint coordinate
  int32_t qadr;          // JNT_BALL: address of the quaternion in q
  double pos[3], quat[4];
  double axis[3], jpos[3], qpos0;
};

struct LaneFrame {
  int32_t link;          // link the frame is attached to (−1: the world body)
  uint32_t chain;        // dofs on the chain world → frame
  int32_t rowmask, pad;
  double lpos[3], lquat[4], cost[6], gain, lm_damping;
};

// ML links, MD dofs of capacity: two instantiations, so that the two-row build's larger tables leave the layout (and with it
// the register allocation) of the lane kernel and of the one-row builds exactly as it was
template <int ML, int MD>
struct LaneProblemT {
  int32_t nq, nv, nlink, n_frame, n_posture, n_cfg, n_vel, pad;
  LaneLink link[ML];
  int32_t dof_link[MD];   // link whose joint moves dof d (−1: not on any task chain)
  int32_t dof_qadr[MD];
  double range_lo[MD], range_hi[MD];    // joint range for check_limits (±inf)
  LaneFrame frame[kLaneMaxFrames];
  double posture_cost[kMaxPostureTasks][MD], posture_gain[kMaxPostureTasks], posture_lm[kMaxPostureTasks];
  double cfg_gain[kMaxBoxTerms], cfg_lower[kMaxBoxTerms][MD], cfg_upper[kMaxBoxTerms][MD];
  double vel_limit[kMaxBoxTerms][MD];
  // per dof, for the row kernel (quad_kernel.h: one load level less than dof_link → link): the joint's axis and anchor in its
  // body frame, and whether it is a slide joint
  double dof_axis[MD][3], dof_jpos[MD][3];
  int32_t dof_slide[MD];
  // ComTask (two-row build of the row kernel only; appended so that the offsets above stay what they were): every body of the robot
  // is a link then; a link carries its own mass (jointless children folded in) at link_ipos, its subtree is the contiguous range
  // [link, link_last] (links are in body order), link_stmass that range's mass; bodies fixed to the world enter as a constant
  int32_t n_com, com_rowmask;
  double com_cost[3], com_gain, com_lm, com_minv;       // com_minv = 1 / mass of the subtree of body 1
  double com_static[3];                                  // Σ m·p over the bodies of that subtree that are fixed to the world
  double link_mass[ML], link_ipos[ML][3], link_stmass[ML];
  int32_t link_last[ML];
  // RelativeFrameTask (two-row build only; appended for the same reason): frame task t measures its frame in the frame `root`
  struct Rel { int32_t relative, root_link; uint32_t rchain; int32_t pad; double rlpos[3], rlquat[4]; } rel[kLaneMaxFrames];
};
using LaneProblem = LaneProblemT<kLaneMaxLinks, kLaneDescDofs>;
using LaneProblem2 = LaneProblemT<kLaneMaxLinks2, kLaneDescDofs2>;

// the same problem in the smaller descriptor (host side; the builder fills the large one)
template <int ML, int MD, int ML2, int MD2>
inline void lane_problem_narrow(const LaneProblemT<ML2, MD2>& b, LaneProblemT<ML, MD>& a) {
  a.nq = b.nq; a.nv = b.nv; a.nlink = b.nlink; a.n_frame = b.n_frame; a.n_posture = b.n_posture; a.n_cfg = b.n_cfg; a.n_vel = b.n_vel; a.pad = 0;
  for (int i = 0; i < ML; ++i) a.link[i] = b.link[i];
  for (int i = 0; i < kLaneMaxFrames; ++i) a.frame[i] = b.frame[i];
  for (int d = 0; d < MD; ++d) {
    a.dof_link[d] = b.dof_link[d]; a.dof_qadr[d] = b.dof_qadr[d]; a.range_lo[d] = b.range_lo[d]; a.range_hi[d] = b.range_hi[d];
    a.dof_slide[d] = b.dof_slide[d];
    for (int c = 0; c < 3; ++c) { a.dof_axis[d][c] = b.dof_axis[d][c]; a.dof_jpos[d][c] = b.dof_jpos[d][c]; }
    for (int t = 0; t < kMaxPostureTasks; ++t) a.posture_cost[t][d] = b.posture_cost[t][d];
    for (int t = 0; t < kMaxBoxTerms; ++t) { a.cfg_lower[t][d] = b.cfg_lower[t][d]; a.cfg_upper[t][d] = b.cfg_upper[t][d]; a.vel_limit[t][d] = b.vel_limit[t][d]; }
  }
  for (int t = 0; t < kMaxPostureTasks; ++t) { a.posture_gain[t] = b.posture_gain[t]; a.posture_lm[t] = b.posture_lm[t]; }
  for (int t = 0; t < kMaxBoxTerms; ++t) a.cfg_gain[t] = b.cfg_gain[t];
  a.n_com = 0; a.com_rowmask = 0;                        // (no ComTask, no RelativeFrameTask on the small descriptor)
  for (int t = 0; t < kLaneMaxFrames; ++t) a.rel[t].relative = 0;
}

// The sizes of a LaneProblem, by value in the row kernel's arguments (SGPRs at wave start instead of a dependent load).
// `qadr_identity`: dof d reads q[d] (always the case for nq = nv with hinge / slide joints only; checked on the host).
struct LaneDims { int32_t nq, nv, nlink, n_frame, n_posture, n_cfg, n_vel, qadr_identity; };

// bytes of LDS per wavefront: link poses [nlink][7][64] (joint axes and anchors are recomputed from the pose where a
// Jacobian column needs them: ~40 flops instead of 6 more doubles of LDS per link and lane, i.e. residency)
__host__ __device__ inline int lane_lds_bytes(int nlink) { return nlink * 7 * kWave * (int)sizeof(double); }

// LOOP: the fused caller loop (mkh_solve_steps / mkh_solve_until, ik_kernel.h "Fused outer loop"): every lane iterates
// (solve, q ← q + Δq) on its own problem — hinge / slide joints only, so the integration is an addition — until its
// frame tasks are within the thresholds (until) or the iteration budget is spent; lanes that are finished idle through
// the remaining iterations of their wavefront (masked commits).  A separate instantiation: the single-solve kernel
// keeps its register budget.
template <int NV, bool LOOP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LOOP ? 1 : 2, 2))) void ik_lane_kernel(const LaneProblem* __restrict__ Pg, const SolveArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const LaneProblem& P = *Pg;
  const int lane = (int)threadIdx.x;
  const int pb_raw = (int)blockIdx.x * kWave + lane;
  const bool live = pb_raw < A.B;
  const int pb = live ? pb_raw : A.B - 1;            // idle lanes of the last wave redo the last problem, store nothing
  const int nv = P.nv, nq = P.nq, nlink = P.nlink;
  const double kInf = __builtin_huge_val();
  double* const sX = smem + lane;                     // sX[(l·7 + c)·64]
  int status_all = 0;

  // ------------------------------------------------------------------ q
  double q[NV];
#pragma unroll
  for (int d = 0; d < NV; ++d) q[d] = d < nv ? A.q[(size_t)pb * nq + P.dof_qadr[d]] : 0.0;
  const bool until = LOOP && A.pos_threshold >= 0.0;
  const int n_steps = LOOP ? A.n_steps : 1;
  bool fin = false;                                  // this lane's loop is over (converged / failed / budget spent)
  int it_done = 0, conv_flag = 0;
  double vlast[NV];
#pragma unroll
  for (int d = 0; d < NV; ++d) vlast[d] = 0.0;
  double x[NV];
  int st[NV];                                        // QP partition: 0 free, 1 at lower, 2 at upper (kept across fused steps)
#pragma unroll
  for (int d = 0; d < NV; ++d) st[d] = 0;
  int status = 0;
  for (int step = 0; step < n_steps + (until ? 1 : 0); ++step) {
  if (LOOP && !__ballot(!fin)) break;
  status = 0;
  bool conv_now = true;                              // every frame task within the thresholds at the current q
  // Configuration.check_limits (mink/configuration.py:77-110), tol = 1e-6
#pragma unroll
  for (int d = 0; d < NV; ++d)
    if (d < nv && (q[d] < P.range_lo[d] - 1e-6 || q[d] > P.range_hi[d] + 1e-6)) status |= 1;

  // ----------------------------------------------------------------- FK
  // mj_kinematics down the (uniform) list of links: parents come first
  for (int l = 0; l < nlink; ++l) {
    const LaneLink& L = P.link[l];
    V3 xp{L.pos[0], L.pos[1], L.pos[2]};
    Q4 xq{L.quat[0], L.quat[1], L.quat[2], L.quat[3]};
    if (L.jtype >= 0) {
      double qv = 0.0;
#pragma unroll
      for (int d = 0; d < NV; ++d) qv = (d == L.dof) ? q[d] : qv;       // (uniform select: scalar compares)
      qv -= L.qpos0;
      const V3 ax{L.axis[0], L.axis[1], L.axis[2]};
      if (L.jtype == JNT_SLIDE) {
        xp = xp + qv * qrot(xq, ax);
      } else {
        const V3 jp{L.jpos[0], L.jpos[1], L.jpos[2]};
        const V3 anchor = xp + qrot(xq, jp);
        xq = qmul(xq, axis_angle(ax, qv));
        xp = anchor - qrot(xq, jp);
      }
    }
    if (L.parent >= 0) {
      const double* a = sX + L.parent * 7 * kWave;
      const V3 ap{a[0], a[kWave], a[2 * kWave]};
      const Q4 aq{a[3 * kWave], a[4 * kWave], a[5 * kWave], a[6 * kWave]};
      xp = ap + qrot(aq, xp);
      xq = qmul(aq, xq);
    }
    xq = qnormalize(xq);
    double* o = sX + l * 7 * kWave;
    o[0] = xp.x; o[kWave] = xp.y; o[2 * kWave] = xp.z;
    o[3 * kWave] = xq.w; o[4 * kWave] = xq.x; o[5 * kWave] = xq.y; o[6 * kWave] = xq.z;
  }

  // ------------------------------------------------- objective: H (upper triangle), c
  // (The Lie-algebra part of a frame task — log, jlog — is the register peak of the kernel.  The first frame task is
  // peeled out of the loop so that H and c only come alive after it: 288 → 2xx VGPRs for a 6-dof arm, i.e. two
  // resident wavefronts per SIMD instead of one.)
  double H[NV][NV], c[NV];
  double mu_total = A.damping;
  // frame tasks (frame_task.py:95-146)
  auto frame_task = [&](const int f, const bool first) __attribute__((always_inline)) {
    const LaneFrame& ft = P.frame[f];
    SE3 F;
    {
      V3 bp{0, 0, 0};
      Q4 bq{1, 0, 0, 0};
      if (ft.link >= 0) {
        const double* a = sX + ft.link * 7 * kWave;
        bp = V3{a[0], a[kWave], a[2 * kWave]};
        bq = Q4{a[3 * kWave], a[4 * kWave], a[5 * kWave], a[6 * kWave]};
      }
      F.p = bp + qrot(bq, V3{ft.lpos[0], ft.lpos[1], ft.lpos[2]});
      F.q = qmul(bq, Q4{ft.lquat[0], ft.lquat[1], ft.lquat[2], ft.lquat[3]});
    }
    const double* tg = A.frame_targets + ((size_t)pb * P.n_frame + f) * 7;
    const SE3 Tt{Q4{tg[0], tg[1], tg[2], tg[3]}, V3{tg[4], tg[5], tg[6]}};
    V3 ev, ew;
    double Jm[9], Qm[9];
    bool ident;
    se3_log(se3_mul(se3_inv(F), Tt), ev, ew);        // e = target.minus(frame)
    se3_ljacinv(ev, ew, Jm, Qm, ident);              // jlog(T_tb) = ljacinv(e)
    const double e6[6] = {ev.x, ev.y, ev.z, ew.x, ew.y, ew.z};
    double we[6], ss = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      we[r] = ft.cost[r] * (-ft.gain * e6[r]);
      ss += we[r] * we[r];
    }
    mu_total += ft.lm_damping * ss;
    if (LOOP && until) {
      const double pt = A.pos_threshold, ot = A.ori_threshold;
      conv_now = conv_now && (!(ft.rowmask & 7) || dot(ev, ev) <= pt * pt) && (!(ft.rowmask & 56) || dot(ew, ew) <= ot * ot);
    }
    // Weighted task Jacobian column of a dof with world motion axis (lin, ang) at the frame:
    //   ᴮJ = [Rfᵀ·lin; Rfᵀ·ang]  (configuration.py:148-153),  J = −jlog·ᴮJ,  jlog = [[J, −J·Q·J],[0, J]]
    //   ⇒ rows 0-2 = A1·lin + A2·ang,  rows 3-5 = A1·ang   with  A1 = −J·Rfᵀ,  A2 = J·Q·J·Rfᵀ
    // (two 3×3 matrices per task instead of J, Q, Rf per column), rows scaled by the cost: weighted_jacobian.
    double A1[9], A2[9];
    {
      // staged so that few 3×3 temporaries are live at once: A1 = −J·Rfᵀ, then JQ = J·Q, then A2 = −JQ·A1
      {
        const M3 Rf = qmat(F.q);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)    // (M·Rfᵀ)[i][j] = Σ_k M[i][k]·Rf[j][k]
            A1[3 * i + j] = -(Jm[3 * i] * Rf.m[3 * j] + Jm[3 * i + 1] * Rf.m[3 * j + 1] + Jm[3 * i + 2] * Rf.m[3 * j + 2]);
      }
      __builtin_amdgcn_sched_barrier(0);
      double JQ[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          JQ[3 * i + j] = Jm[3 * i] * Qm[j] + Jm[3 * i + 1] * Qm[3 + j] + Jm[3 * i + 2] * Qm[6 + j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          A2[3 * i + j] = -(JQ[3 * i] * A1[j] + JQ[3 * i + 1] * A1[3 + j] + JQ[3 * i + 2] * A1[6 + j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    double Jw[6][NV];
#pragma unroll
    for (int d = 0; d < NV; ++d) {
      double Jt[6] = {0, 0, 0, 0, 0, 0};
      if (d < nv && ((ft.chain >> d) & 1u)) {        // (uniform branch)
        // a joint's axis and anchor are invariant under its own motion: the final body frame gives them
        const int l = P.dof_link[d];
        const LaneLink& L = P.link[l];
        const double* a7 = sX + l * 7 * kWave;
        const Q4 lq{a7[3 * kWave], a7[4 * kWave], a7[5 * kWave], a7[6 * kWave]};
        const V3 axw = qrot(lq, V3{L.axis[0], L.axis[1], L.axis[2]});
        if (L.jtype == JNT_SLIDE) {
#pragma unroll
          for (int r = 0; r < 3; ++r) Jt[r] = A1[3 * r] * axw.x + A1[3 * r + 1] * axw.y + A1[3 * r + 2] * axw.z;
        } else {
          V3 an{a7[0], a7[kWave], a7[2 * kWave]};
          if (L.jpos[0] != 0.0 || L.jpos[1] != 0.0 || L.jpos[2] != 0.0) an = an + qrot(lq, V3{L.jpos[0], L.jpos[1], L.jpos[2]});
          const V3 jp = cross(axw, F.p - an);
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            Jt[r] = A1[3 * r] * jp.x + A1[3 * r + 1] * jp.y + A1[3 * r + 2] * jp.z +
                    (A2[3 * r] * axw.x + A2[3 * r + 1] * axw.y + A2[3 * r + 2] * axw.z);
            Jt[3 + r] = A1[3 * r] * axw.x + A1[3 * r + 1] * axw.y + A1[3 * r + 2] * axw.z;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) Jw[r][d] = ft.cost[r] * Jt[r];          // weighted_jacobian (task.py:129)
    }
    if (first) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        c[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) H[i][j] = 0.0;
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if ((ft.rowmask >> r) & 1) {                   // (uniform: rows with zero cost contribute nothing)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          c[i] -= we[r] * Jw[r][i];
#pragma unroll
          for (int j = i; j < NV; ++j) H[i][j] = fma(Jw[r][i], Jw[r][j], H[i][j]);
        }
      }
    }
  };
  frame_task(0, true);                               // (n_frame ≥ 1: checked on the host)
  for (int f = 1; f < P.n_frame; ++f) frame_task(f, false);
  // posture tasks (posture_task.py:87-142): e = target − q, J = −I  (hinge / slide dofs)
  for (int t = 0; t < P.n_posture; ++t) {
    const double* tq = A.posture_target + (A.posture_batched ? ((size_t)pb * P.n_posture + t) * nq : (size_t)t * nq);
    double ss = 0.0;
#pragma unroll
    for (int d = 0; d < NV; ++d) {
      if (d < nv) {
        const double cost = P.posture_cost[t][d];
        const double we = cost * (-P.posture_gain[t] * (tq[P.dof_qadr[d]] - q[d]));
        const double wj = -cost;
        H[d][d] += wj * wj;
        c[d] -= we * wj;
        ss += we * we;
      }
    }
    mu_total += P.posture_lm[t] * ss;
  }
#pragma unroll
  for (int d = 0; d < NV; ++d) H[d][d] += (d < nv) ? mu_total : 1.0;     // padded dofs: identity, x = 0

  // ------------------------------------------------------------ box limits
  double lo[NV], hi[NV];
#pragma unroll
  for (int d = 0; d < NV; ++d) {
    lo[d] = -kInf; hi[d] = kInf;
    if (d < nv) {
      for (int t = 0; t < P.n_cfg; ++t) {            // configuration_limit.py:94-124
        const double lw = P.cfg_lower[t][d], up = P.cfg_upper[t][d];
        if (up < kInf) hi[d] = fmin(hi[d], P.cfg_gain[t] * (up - q[d]));
        if (lw > -kInf) lo[d] = fmax(lo[d], -(P.cfg_gain[t] * (q[d] - lw)));
      }
      for (int t = 0; t < P.n_vel; ++t) {            // velocity_limit.py:96-101
        const double vm = P.vel_limit[t][d];
        if (vm < kInf) { hi[d] = fmin(hi[d], A.dt * vm); lo[d] = fmax(lo[d], -(A.dt * vm)); }
      }
      if (lo[d] > hi[d] + 1e-12) status |= 2;        // quadprog: "constraints are inconsistent"
    } else {
      lo[d] = 0.0; hi[d] = 0.0;
    }
  }

  if (LOOP && until && step > 0 && !fin) {
    it_done = step;
    if (conv_now) { conv_flag = 1; fin = true; status_all |= status; }        // the callers' break (after the integration)
    else if (step == n_steps) { fin = true; status_all |= status; }           // budget spent: this pass only tested
  }

  // ------------------------------------------------------------------- QP
  // (fused loop, from the third step on: the partition starts where the previous step's QP ended — along an IK loop the
  //  active set changes little from step to step; see the warm start of phase 1a in ik_kernel.h)
  // Cold start: the partition of the diagonal estimate x_d ≈ −c_d / H_dd — a dof it puts outside its box starts AT that
  // bound (any partition is a valid start).  On saturated problems (the benchmark's velocity limits: 5.8 of 6 dofs on a
  // bound) block principal pivoting then needs 1.2 factorisations on average instead of 3.4 from the all-free start
  // (counted on the tapped H, c, box of the benchmark batch; quad_kernel.h uses the same start).
  // (the same across CALLS on one handle — MKH_FLAG_WARM_START, SolveArgs::warm, round 6: the partition this instance's previous solve
  //  ended with, from the handle's third call on, like the wavefront and row kernels; single solves only — a fused loop carries it)
  const bool warm_in = !LOOP && A.warm != nullptr && A.warm_age >= 2;
#pragma unroll
  for (int d = 0; d < NV; ++d) {
    x[d] = 0.0;
    if (warm_in) {
      int w = (d < nv) ? (int)A.warm[(size_t)pb * nv + d] : 0;
      if ((w == 2 && !(hi[d] < kInf)) || (w == 1 && !(lo[d] > -kInf))) w = 0;    // (a bound that is not there any more)
      st[d] = w;
    } else if (!(LOOP && step >= 2)) {
      const double xd = -c[d] * fast_rcp(H[d][d]);
      st[d] = (d < nv) ? (xd > hi[d] ? 2 : (xd < lo[d] ? 1 : 0)) : 0;
    }
  }
  double hmax = 0.0;
#pragma unroll
  for (int d = 0; d < NV; ++d) hmax = fmax(hmax, H[d][d]);
  const double tolw = 1e-16 * hmax;         // (the wavefront kernel's multiplier threshold, ik_kernel.h phase 1a)
  int best = NV + 1, budget = 3;
  bool done = (status & 2) != 0;
  for (int it = 0; it < 10 * NV + 10; ++it) {
    if (!__ballot(!done)) break;                     // every lane of the wave has its optimum
    // masked system: bound indices → identity rows with the bound as right-hand side
    double L[NV][NV], rhs[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double xb = st[i] == 1 ? lo[i] : hi[i];
      double r = st[i] ? xb : -c[i];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const double hij = (i <= j) ? H[i][j] : H[j][i];
        const double xbj = st[j] == 1 ? lo[j] : hi[j];
        if (!st[i] && st[j]) r -= hij * xbj;         // free row: move the bound columns to the right-hand side
        if (j >= i) L[i][j] = (st[i] || st[j]) ? ((i == j) ? 1.0 : 0.0) : hij;
      }
      rhs[i] = r;
    }
    // Cholesky (upper triangle in place: L[i][j], j ≥ i, holds Uᵀ), forward + back substitution
    bool pd = true;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double dk = L[k][k];
      pd = pd && (dk > 0.0);
      const double inv = fast_rcp(sqrt(dk > 0.0 ? dk : 1.0));
      L[k][k] = inv;                                 // store 1/u_kk
#pragma unroll
      for (int j = k + 1; j < NV; ++j) L[k][j] *= inv;
#pragma unroll
      for (int i = k + 1; i < NV; ++i)
#pragma unroll
        for (int j = i; j < NV; ++j) L[i][j] = fma(-L[k][i], L[k][j], L[i][j]);
    }
    if (!pd && !done) { status |= 4; done = true; }
    double* const xn = rhs;                          // forward and back substitution in place
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double s = rhs[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(-L[k][i], xn[k], s);
      xn[i] = s * L[i][i];
    }
#pragma unroll
    for (int i = NV - 1; i >= 0; --i) {
      double s = xn[i];
#pragma unroll
      for (int k = i + 1; k < NV; ++k) s = fma(-L[i][k], xn[k], s);
      xn[i] = s * L[i][i];
    }
    // gradient w = H·x + c (multiplier of a bound index) and the infeasibilities
    int cnt = 0, last = -1;
    int flip[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double w = c[i];
#pragma unroll
      for (int j = 0; j < NV; ++j) w = fma((i <= j) ? H[i][j] : H[j][i], xn[j], w);
      int f = 0;                                     // 0 keep, 1 → lower, 2 → upper, 3 → free
      if (!st[i]) {
        if (lo[i] - xn[i] > 1e-12) f = 1;
        else if (xn[i] - hi[i] > 1e-12) f = 2;
      } else if ((st[i] == 1 && w < -tolw) || (st[i] == 2 && w > tolw)) {
        f = 3;
      }
      flip[i] = f;
      if (f) { ++cnt; last = i; }
    }
    if (!done) {
#pragma unroll
      for (int i = 0; i < NV; ++i) x[i] = xn[i];
      if (cnt == 0) {
        done = true;
      } else {
        bool all = true;
        if (cnt < best) { best = cnt; budget = 3; }
        else if (budget > 0) --budget;
        else all = false;                            // Murty: only the infeasible index with the largest number
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (flip[i] && (all || i == last)) st[i] = (flip[i] == 3) ? 0 : flip[i];
      }
    }
  }
  if (!done) status |= 8;
  if (!LOOP) { status_all = status; break; }
  if (!fin) {                                        // commit this iteration
    status_all |= status;
    if (status & 14) {
      fin = true;                                    // an instance stops at the first step whose QP fails
    } else {
#pragma unroll
      for (int d = 0; d < NV; ++d) {
        vlast[d] = x[d] / A.dt;
        q[d] += x[d];                                // mj_integratePos for hinge / slide joints (configuration.py:228-236)
      }
      if (!until) { it_done = step + 1; fin = step + 1 == n_steps; }
    }
  }
  }  // step loop

  // ------------------------------------------------------------------ out
  if (live) {
    const double bad = __builtin_nan("");
#pragma unroll
    for (int d = 0; d < NV; ++d) {
      if (d < nv) {
        const double vd = LOOP ? vlast[d] : x[d] / A.dt;                              // v = Δq / dt (solve_ik.py:104)
        A.v_out[(size_t)pb * nv + d] = (status_all & 14) ? bad : vd;
        if (LOOP && A.q_out) A.q_out[(size_t)pb * nq + P.dof_qadr[d]] = q[d];
      }
    }
    if (A.status_out) A.status_out[pb] = status_all;
    if (!LOOP && A.warm != nullptr) {
#pragma unroll
      for (int d = 0; d < NV; ++d)
        if (d < nv) A.warm[(size_t)pb * nv + d] = (int8_t)((status_all & 14) ? 0 : st[d]);
    }
    if (LOOP && until) {
      if (A.iters_out) A.iters_out[pb] = it_done;
      if (A.converged_out) A.converged_out[pb] = conv_flag;
    }
  }
}

}  // namespace mkh
