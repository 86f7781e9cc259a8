// ik_wide_kernel — one WORKGROUP (256 threads) per problem: the path for what one wavefront cannot hold.
//
// The wavefront kernels (ik_kernel.h) keep a problem's tableau column in the registers of 64 lanes: nv + active half-space
// rows ≤ 64, nbody ≤ 64.  The reference has no such limit — `solve_ik` stacks every row its limits return
// (mink/solve_ik.py:25-40: np.vstack(G_list)) on any nv (:43-65), and CollisionAvoidanceLimit returns one row per pair
// (mink/limits/collision_avoidance_limit.py:187-210).  This kernel takes
//   * models with more than 64 bodies or dofs (every call), and
//   * the instances of a collision / plugin-row problem in which more rows are ACTIVE than the wavefront kernel has tableau
//     rows and a row it dropped is violated at its solution (MKH_ST_ROW_OVERFLOW): a redo launch behind the normal one
//     (SolveArgs::redo_mask) that solves exactly those, every detected contact a row.
// It has to return mink's answer first; the structure is still the device's: body poses level by
// level of the kinematic tree, (task, dof) pairs / (i, j) entries of H / tableau entries spread over the 256 threads, the
// dual active-set iteration of the wavefront kernels (Goldfarb–Idnani on a symmetric sweep tableau, tools/proto_tableau_qp.py)
// with the tableau in LDS when (nv + rows)² doubles fit next to the per-problem state, else in a slice of device memory.
//
// Covered: FrameTask / RelativeFrameTask (body, geom, site frames), PostureTask (any number: DampingTask is one), ComTask,
// caller-defined task rows, ConfigurationLimit, VelocityLimit, CollisionAvoidanceLimit (every pair type of collide_dev.h /
// convex_dev.h), caller-defined limit rows and box rows; every parity tap of MkhTaps except the cycle counters (round 5: per-task
// (e, J) and the iteration counts too, so that Configuration.get_frame_jacobian / Task.compute_error / compute_jacobian work on
// big models — mink/configuration.py:112-155, mink/tasks/task.py:81-103); the fused caller loops mkh_solve_steps /
// mkh_solve_until (round 5: the step loop runs inside this kernel, q in LDS, per-instance break on the callers' thresholds —
// examples/arm_ur5e_actuators.py:88-97), also as the redo of a fused loop of a wavefront kernel that met more contacts than it
// has rows (the host keeps a copy of q for that launch: q_out may alias q).
#pragma once
#include <hip/hip_runtime.h>

#include "mkh_types.h"
#include "wave_ops.h"
// (a workgroup of four wavefronts: the wave-cooperative expanding polytope must not use the workgroup barrier)
#define MKH_EPA_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#include "collide_dev.h"
#include "wide_types.h"

namespace mkh {

__device__ __forceinline__ bool wide_on_chain(const WideProblem& P, int body, int k) {
  return (P.chain[(size_t)body * P.chain_words + (k >> 6)] >> (k & 63)) & 1ull;
}

// Column k of frame task t (6 rows, unweighted) — frame_column_fn of ik_kernel.h with the chain tests handed in.
__device__ __forceinline__ void wide_frame_column(const double* o, const double* kd, bool on_frame, bool rel, bool on_root, double (&Jt)[6]) {
  const V3 d_ang{kd[0], kd[1], kd[2]}, d_lin{kd[3], kd[4], kd[5]}, d_anchor{kd[6], kd[7], kd[8]};
  V3 a{0, 0, 0}, w{0, 0, 0};
  if (on_frame) {
    const V3 pf{o[27], o[28], o[29]};
    const V3 jp = d_lin + cross(d_ang, pf - d_anchor);
    M3 Rf;
#pragma unroll
    for (int i = 0; i < 9; ++i) Rf.m[i] = o[18 + i];
    a = mulT(Rf, jp); w = mulT(Rf, d_ang);              // body-frame Jacobian (configuration.py:148-153)
  }
  double sign = -1.0;                                    // FrameTask: J = −jlog(T_tb)·ᴮJ  (frame_task.py:144-146)
  if (rel) {
    sign = 1.0;                                          // RelativeFrameTask: J = +jlog(T_tf)·(ᶠJ − Ad·ʳJ)  (relative_frame_task.py:131-142)
    if (on_root) {
      const V3 pr{o[45], o[46], o[47]};
      const V3 jp = d_lin + cross(d_ang, pr - d_anchor);
      M3 Rr, Rrf;
#pragma unroll
      for (int i = 0; i < 9; ++i) { Rr.m[i] = o[36 + i]; Rrf.m[i] = o[48 + i]; }
      const V3 ar = mulT(Rr, jp), wr = mulT(Rr, d_ang);
      const V3 Rw = mul(Rrf, wr);                        // Ad(T_fr⁻¹) = [[R, [t]×R],[0, R]]
      a = a - (mul(Rrf, ar) + cross(V3{o[57], o[58], o[59]}, Rw));
      w = w - Rw;
    }
  }
  // jlog = [[J, −J·Q·J],[0, J]]:  y = J·w;  rows 0-2 = ±J·(a − Q·y), rows 3-5 = ±y
  const double wv[3] = {w.x, w.y, w.z}, av[3] = {a.x, a.y, a.z};
  double y[3], z3[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) y[r] = o[3 * r] * wv[0] + o[3 * r + 1] * wv[1] + o[3 * r + 2] * wv[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) z3[r] = av[r] - (o[9 + 3 * r] * y[0] + o[9 + 3 * r + 1] * y[1] + o[9 + 3 * r + 2] * y[2]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Jt[r] = sign * (o[3 * r] * z3[0] + o[3 * r + 1] * z3[1] + o[3 * r + 2] * z3[2]);
    Jt[3 + r] = sign * y[r];
  }
}

// index states of the active-set iteration (tools/proto_tableau_qp.py)
enum { WS_FREE = 0, WS_AT_LO = 1, WS_AT_HI = 2, WS_ROW_OFF = 3, WS_ROW_ON = 4, WS_ZERO = 5 };

typedef __attribute__((address_space(3))) double lds_f64;

// T[i][j] += α·u[i]·u[j] (kTwo: + β·v[i]·v[j]) on the leading n × n block of the row-major tableau (row stride N): THE inner loop of
// this kernel — a sweep is one such update with the row / column fix-up folded in, H is accumulated by them.  Each wavefront takes
// every fourth row, lanes take columns (consecutive addresses: no bank conflicts), eight rows per trip so that sixteen LDS reads
// are in flight; u[i]·u[j] is formed before α enters, so the update is bitwise symmetric and a tableau column can be read as
// a row.  kfix ≥ 0 (a sweep on index kfix, α = −1/d): row and column kfix get ±u/d (`sgi`) and the diagonal entry −1/d in the
// same pass — every entry has one owner (wave = row mod 4, lane = column mod 64), so no barrier separates update and fix-up.
// (Round 4 walked the flat index with an integer division per element and one dependent LDS round trip after the other:
//  12.6 k cycles per 75 x 75 sweep.)
template <bool kTwo, class TP>
__device__ __forceinline__ void wide_rank1(TP* Tp, int n, int N, const lds_f64* u, double alpha, const lds_f64* v2, double beta,
                                           int kfix, double sgi, int wave, int lane) {
  constexpr int NW = kWideThreads / 64, RB = 8;
  for (int jb = 0; jb < n; jb += 64) {
    const int j = jb + lane;
    if (j >= n) continue;
    const bool colk = j == kfix;
    const double uj = u[j], vj = kTwo ? v2[j] : 0.0;
    TP* const c = Tp + j;
    for (int i = wave; i < n; i += RB * NW) {
      // (rows past the end: the loads go to the last row — a valid address, no divergent guard —, only the stores are predicated)
      double t[RB], ui[RB], vi[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int ii = i + r * NW < n ? i + r * NW : n - 1;
        t[r] = c[(size_t)ii * N];
        ui[r] = u[ii];
        vi[r] = kTwo ? v2[ii] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        double x = fma(ui[r] * uj, alpha, t[r]);
        if (kTwo) x = fma(vi[r] * vj, beta, x);
        if (!kTwo) x = colk ? sgi * ui[r] : x;
        if (i + r * NW < n) c[(size_t)(i + r * NW) * N] = x;
      }
    }
    // row kfix, by the wave that owns it (its lanes wrote these entries above: program order, no barrier)
    if (!kTwo && kfix >= 0 && wave == (kfix & (NW - 1))) c[(size_t)kfix * N] = colk ? alpha : sgi * uj;
  }
}

// T[i][j] −= Σ_c V[i][c]·U[c][j], c < 4, on the leading n × n block: four pivots (or four Jacobian rows) per pass over the tableau —
// a quarter of the tableau traffic and of the barriers of four rank-1 passes.  U: [4][stride] (lane-contiguous), Vt: [i][4] (two
// 16-byte broadcast reads per row).  k0 ≥ 0 (a BLOCK SWEEP on the indices k0 … k0+3, k0 a multiple of 4, Vt = U·E with E the
// inverse of the 4 × 4 pivot block): columns k0 … k0+3 are left alone by the update, and rows / columns k0+c are then written by
// wave c — the owner of row k0+c — as E·Uᵀ (the new rows of a swept tableau) and −E on the block itself.
// (Measured and dropped: the row factors through the lanes and v_readlane instead of broadcast LDS reads — 5.76 → 6.5 ms on the
//  G1 with two hands: with two workgroups per CU the passes are bound by VALU issue, not by the LDS return path.)
template <class TP>
__device__ __forceinline__ void wide_rank4(TP* Tp, int n, int N, const lds_f64* U, int stride, const lds_f64* Vt, int k0, const double (&E)[4][4],
                                           int wave, int lane) {
  constexpr int NW = kWideThreads / 64, RB = 6;
  for (int jb = 0; jb < n; jb += 64) {
    const int j = jb + lane;
    const bool jin = k0 >= 0 && (unsigned)(j - k0) < 4u;
    if (j < n && !jin) {
      const double u0 = U[j], u1 = U[stride + j], u2 = U[2 * stride + j], u3 = U[3 * stride + j];
      TP* const c = Tp + j;
      for (int i = wave; i < n; i += RB * NW) {
        double t[RB], v[RB][4];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int ii = i + r * NW < n ? i + r * NW : n - 1;
          t[r] = c[(size_t)ii * N];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[r][q] = Vt[4 * ii + q];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const double x = fma(-v[r][3], u3, fma(-v[r][2], u2, fma(-v[r][1], u1, fma(-v[r][0], u0, t[r]))));
          if (i + r * NW < n) c[(size_t)(i + r * NW) * N] = x;
        }
      }
    }
    if (k0 >= 0 && j < n) {
      // rows / columns of the block, by the wave that owns row k0 + wave (its lanes' writes above precede these in program order;
      // nobody else writes the columns: the update skipped them)
      const int kr = k0 + wave;
      double val = Vt[4 * j + wave];
      if (jin) val = -E[wave][(j - k0) & 3];
      Tp[(size_t)kr * N + j] = val;
      if (!jin) Tp[(size_t)j * N + kr] = val;
    }
  }
}

// What the tableau phases need of the kernel's state: a REAL call (noinline) — the kernel around them carries the inlined distance
// routines and sits at its register ceiling (256 VGPRs, 470 spilled SGPRs); as callees the two hot loops get a register allocation of
// their own (round 5: the same inner loop compiled inline took 11 k cycles per sweep, spilling scalars inside it).
struct WideQpCtx {
  double* T;                    // tableau (LDS or the workgroup's slice of device memory)
  int in_lds, N, nv;
  int o_z, o_w, o_lo, o_hi, o_rown, o_ref, o_col, o_red, o_state;      // LDS offsets (doubles)
  int o_blk, blk_stride;        // staging of the block pivots / rank-4 accumulation (−1: none)
};

// H = λI + Σ JwᵀJw (+ the posture tasks' diagonal, added by the caller): two weighted Jacobian rows per pass, staged in LDS (sCol,
// sRef — both free until the QP), T += u·uᵀ + v·vᵀ  (round 4: every entry walked all rows of Jw in device memory — 185 k cycles
// on the G1 with two hands, an eighth of the solve).  The next pair of rows is requested before the current update runs.
__device__ __attribute__((noinline)) void wide_accumulate_h(WideQpCtx X, const double* Jw, int R_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, N = X.N, nv = X.nv;
  lds_f64* const sCol = (lds_f64*)(smem + X.o_col);
  lds_f64* const sRef = (lds_f64*)(smem + X.o_ref);
  double* const T = X.T;
  for (int e = tid; e < N * N; e += kWideThreads) T[e] = 0.0;
  if (X.o_blk >= 0) {
    // four rows per pass: U[c][k] = Jw[r + c][k], Vt[k][c] = −U[c][k]
    lds_f64* const U = (lds_f64*)(smem + X.o_blk);
    lds_f64* const Vt = U + 4 * X.blk_stride;
    const double none[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int r = 0; r < R_all; r += 4) {
      __syncthreads();
      for (int c = 0; c < 4; ++c)
        for (int k = tid; k < nv; k += kWideThreads) {
          const double x = (r + c < R_all) ? Jw[(size_t)(r + c) * nv + k] : 0.0;
          U[c * X.blk_stride + k] = x; Vt[4 * k + c] = -x;
        }
      __syncthreads();
      if (X.in_lds) wide_rank4((lds_f64*)T, nv, N, U, X.blk_stride, Vt, -1, none, wave, lane);
      else wide_rank4(T, nv, N, U, X.blk_stride, Vt, -1, none, wave, lane);
    }
    __syncthreads();
    return;
  }
  // (rows of nv ≤ 4·256 doubles: up to four entries of each of the two rows per thread)
  double a0[4], a1[4];
  auto fetch = [&](int r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = tid + c * kWideThreads;
      a0[c] = (k < nv && r < R_all) ? Jw[(size_t)r * nv + k] : 0.0;
      a1[c] = (k < nv && r + 1 < R_all) ? Jw[(size_t)(r + 1) * nv + k] : 0.0;
    }
  };
  fetch(0);
  for (int r = 0; r < R_all; r += 2) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = tid + c * kWideThreads;
      if (k < nv) { sCol[k] = a0[c]; sRef[k] = a1[c]; }
    }
    __syncthreads();
    fetch(r + 2);
    if (X.in_lds) wide_rank1<true>((lds_f64*)T, nv, N, sCol, 1.0, sRef, 1.0, -1, 0.0, wave, lane);
    else wide_rank1<true>(T, nv, N, sCol, 1.0, sRef, 1.0, -1, 0.0, wave, lane);
  }
  __syncthreads();
}


// Contacts of one problem (collision_avoidance_limit.py:187-229): every pair's distance and witness points → rec[pair] = {h, n, from, to}
// (h = +inf: not in range).  Each wavefront takes kGjkSlots pairs per trip (GJK keeps its simplex in the wave's LDS workspace).
// CVX: the pair list holds pairs without an analytic routine (the general convex routine — GJK, the expanding polytope, the witness-point
// polish — is compiled in).  Two builds of the kernel (round 6): the routine's frame doubled the scratch of EVERY launch of the one kernel
// there was — `g1_hands`, which has no collision pair at all, moved 4.3 GB per launch instead of 3.0 (816 → 1 728 B per lane).
template <bool CVX>
__device__ __attribute__((noinline)) void wide_contacts(const WideProblem* Pg, double dt, double* rec, const double* sX, int XS, double* sCwsAll) {
  const WideProblem& P = *Pg;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double kInf = __builtin_huge_val();
  double* const sCws = sCwsAll + wave * (kEpaWsDoubles > kGjkWsDoubles ? kEpaWsDoubles : kGjkWsDoubles);   // this wave's GJK / EPA workspace
  // More than a wavefront of pairs: bounding spheres first (mkh_types.h PairCull, the rule of ik_kernel.h's collision_phase) —
  // the distance routines run over the pairs that are left (ALOHA: a few dozen of 1 104).  The list's order is of no
  // consequence here: every pair has its own record.
  extern __shared__ __attribute__((aligned(16))) double smem[];
  unsigned short* const sList = reinterpret_cast<unsigned short*>(smem + P.o_rown);
  int* const sCount = reinterpret_cast<int*>(smem + P.o_red);
  const bool use_cull = P.cull != nullptr;
  int n_cand = P.n_pairs;
  if (use_cull) {
    __syncthreads();
    if (tid == 0) *sCount = 0;
    __syncthreads();
    for (int pi = tid; pi < P.n_pairs; pi += kWideThreads) {
      const PairCull& c = P.cull[pi];
      const int b1 = c.body1, b2 = c.body2;
      const Q4 bq1{sX[3 * XS + b1], sX[4 * XS + b1], sX[5 * XS + b1], sX[6 * XS + b1]}, bq2{sX[3 * XS + b2], sX[4 * XS + b2], sX[5 * XS + b2], sX[6 * XS + b2]};
      const V3 d = (V3{sX[b2], sX[XS + b2], sX[2 * XS + b2]} + qrot(bq2, V3{c.lpos2[0], c.lpos2[1], c.lpos2[2]})) -
                   (V3{sX[b1], sX[XS + b1], sX[2 * XS + b1]} + qrot(bq1, V3{c.lpos1[0], c.lpos1[1], c.lpos1[2]}));
      if (!(dot(d, d) > c.reach2)) sList[atomicAdd(sCount, 1)] = (unsigned short)pi;
      else rec[(size_t)pi * 10] = kInf;
    }
    __syncthreads();
    n_cand = *sCount;
  }
  for (int base = 0; base < n_cand; base += 4 * kGjkSlots) {
      const int kc = base + wave * kGjkSlots + lane;
      const bool want = lane < kGjkSlots && kc < n_cand;
      const int pi = !want ? 0 : (use_cull ? (int)sList[kc] : kc);
      double dist = 0.0;
      V3 from{0, 0, 0}, to{0, 0, 0};
      bool need_epa = false;
      auto poses = [&](const CollisionPairDev& cp, V3& gp1, Q4& gq1, V3& gp2, Q4& gq2) {
        const int b1 = cp.body1, b2 = cp.body2;
        const Q4 bq1{sX[3 * XS + b1], sX[4 * XS + b1], sX[5 * XS + b1], sX[6 * XS + b1]}, bq2{sX[3 * XS + b2], sX[4 * XS + b2], sX[5 * XS + b2], sX[6 * XS + b2]};
        gp1 = V3{sX[b1], sX[XS + b1], sX[2 * XS + b1]} + qrot(bq1, V3{cp.lpos1[0], cp.lpos1[1], cp.lpos1[2]});
        gp2 = V3{sX[b2], sX[XS + b2], sX[2 * XS + b2]} + qrot(bq2, V3{cp.lpos2[0], cp.lpos2[1], cp.lpos2[2]});
        gq1 = qmul(bq1, Q4{cp.lquat1[0], cp.lquat1[1], cp.lquat1[2], cp.lquat1[3]});
        gq2 = qmul(bq2, Q4{cp.lquat2[0], cp.lquat2[1], cp.lquat2[2], cp.lquat2[3]});
      };
      if (want) {
        const CollisionPairDev& cp = P.pairs[pi];
        V3 gp1, gp2; Q4 gq1, gq2;
        poses(cp, gp1, gq1, gp2, gq2);
        geom_distance<false, CVX>(cp.type1, V3{cp.size1[0], cp.size1[1], cp.size1[2]}, gp1, gq1, cp.type2,
                                   V3{cp.size2[0], cp.size2[1], cp.size2[2]}, gp2, gq2, cp.ddetect, dist, from, to,
                                   cp.vert1, cp.nvert1, cp.vert2, cp.nvert2, &need_epa, sCws + lane);
      }
      // pairs whose cores overlap: one at a time, this wavefront cooperating on the expanding polytope
      if constexpr (CVX)
      for (unsigned long long em = __ballot(want && need_epa); em; em &= em - 1) {
        const int l = (int)__builtin_ctzll(em);
        const CollisionPairDev& cp = P.pairs[use_cull ? (int)sList[base + wave * kGjkSlots + l] : base + wave * kGjkSlots + l];
        V3 gp1, gp2; Q4 gq1, gq2;
        poses(cp, gp1, gq1, gp2, gq2);
        // (loose polytope + polish on the pair's lane; without a certificate the tight polytope — collide_dev.h geom_overlap_polish)
        const V3 sz1{cp.size1[0], cp.size1[1], cp.size1[2]}, sz2{cp.size2[0], cp.size2[1], cp.size2[2]};
        bool certified = false;
#pragma nounroll
        for (int pass = geom_overlap_loose(cp.type1, cp.type2) ? 0 : 1; pass < 2 && !certified; ++pass) {
          double d_e; V3 f_e, t_e;
          geom_overlap_distance(cp.type1, sz1, gp1, gq1, cp.type2, sz2, gp2, gq2, d_e, f_e, t_e, cp.vert1, cp.nvert1, cp.vert2, cp.nvert2, sCws,
                                pass ? kEpaTol : kLooseEpa);
          bool ok = false;
          if (lane == l) {
            dist = d_e; from = f_e; to = t_e;
            ok = geom_overlap_polish(cp.type1, sz1, gp1, gq1, cp.type2, sz2, gp2, gq2, dist, from, to, pass != 0);
          }
          certified = __ballot(ok) != 0;
        }
      }
      if (want) {
        const CollisionPairDev& cp = P.pairs[pi];
        double* o = rec + (size_t)pi * 10;
        double hk = kInf;
        if (dist != cp.ddetect) {                            // Contact.inactive (:52-56)
          hk = (dist > cp.dmin) ? (cp.gain * (dist - cp.dmin) / dt) + cp.relax : cp.relax;      // :200-205
          V3 nrm = to - from;                                // Contact.normal (:46-50)
          const double nn = sqrt(dot(nrm, nrm));
          nrm = (nn < 1e-15) ? V3{1.0, 0.0, 0.0} : fast_rcp(nn) * nrm;
          o[1] = nrm.x; o[2] = nrm.y; o[3] = nrm.z; o[4] = from.x; o[5] = from.y; o[6] = from.z; o[7] = to.x; o[8] = to.y; o[9] = to.z;
        }
        o[0] = hk;
      }
  }
}

struct WideQpOut { int status, iters, n_outer, n_piv; };

// The QP of one problem: dual active set (Goldfarb–Idnani) on the symmetric sweep tableau K = [[H, Aᵀ],[A, 0]] (tools/proto_tableau_qp.py),
// N = nv + rows.  One pivot = stage column p (barrier), the step on z / w by the owners of the indices, the sweep (rank-1 update
// with the row / column fix-up folded in), barrier: two barriers (round 4: seven, and a second pass over row and column).
__device__ __attribute__((noinline)) WideQpOut wide_qp(WideQpCtx X, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, N = X.N, nv = X.nv;
  constexpr int NT_ = kWideThreads;
  const double kInf = __builtin_huge_val();
  lds_f64* const sZ = (lds_f64*)(smem + X.o_z);
  lds_f64* const sW = (lds_f64*)(smem + X.o_w);
  const lds_f64* const sLo = (const lds_f64*)(smem + X.o_lo);
  const lds_f64* const sHi = (const lds_f64*)(smem + X.o_hi);
  const lds_f64* const sRown = (const lds_f64*)(smem + X.o_rown);
  lds_f64* const sRef = (lds_f64*)(smem + X.o_ref);
  lds_f64* const sCol = (lds_f64*)(smem + X.o_col);
  lds_f64* const sRed = (lds_f64*)(smem + X.o_red);
  typedef __attribute__((address_space(3))) int lds_i32;
  lds_i32* const sRedI = (lds_i32*)(smem + X.o_red + NT_);
  lds_i32* const sState = (lds_i32*)(smem + X.o_state);
  double* const T = X.T;
  int status = 0;
  // arg-max / arg-min over the candidates of all threads: largest (smallest) value, lowest index on ties; idx −1 when none.
  // One DPP reduction per wavefront, then four entries through LDS (round 4: a 256-wide tree with nine barriers).
  auto block_arg = [&](double val, int idx, bool want_max, double& best, int& besti) {
    const double key = idx < 0 ? -kInf : (want_max ? val : -val);       // (arg-min = arg-max of the negated values)
    const double mk = wave_max(key);
    const unsigned cand = (idx >= 0 && key == mk) ? (unsigned)idx : 0xffffffffu;
    const unsigned mi = wave_min_u32(cand);
    __syncthreads();
    if (lane == 0) { sRed[wave] = mk; sRedI[wave] = (int)mi; }
    __syncthreads();
    double b = -kInf; int bi = -1;
#pragma unroll
    for (int w = 0; w < NT_ / 64; ++w) {
      const double kw = sRed[w]; const int iw = sRedI[w];
      if (iw >= 0 && (bi < 0 || kw > b || (kw == b && iw < bi))) { b = kw; bi = iw; }
    }
    best = want_max ? b : -b; besti = bi;
    __syncthreads();
  };
  auto take_column = [&](int p) {                          // (the tableau is bitwise symmetric: column p = row p, consecutive addresses)
    for (int i = tid; i < N; i += NT_) sCol[i] = T[(size_t)p * N + i];
    __syncthreads();
  };
  // step(p, α): z −= α·τ on the basic indices, w += α·τ on the others (τ = column p); p itself: w_p += α when basic, z_p += α when not
  auto take_step = [&](int p, double alpha, bool p_basic) {     // (sCol holds column p; every index is updated by its owner thread)
    for (int i = tid; i < N; i += NT_) {
      const int st = sState[i];
      if (st == WS_FREE || st == WS_ROW_ON) sZ[i] -= alpha * sCol[i]; else sW[i] += alpha * sCol[i];
      if (i == p) { if (p_basic) sW[p] += alpha; else sZ[p] += alpha; }
    }
  };
  auto sweep = [&](int k, bool reverse) {                  // (sCol holds column k)
    const double d = sCol[k], inv = 1.0 / d, sgi = reverse ? -inv : inv;
    if (X.in_lds) wide_rank1<false>((lds_f64*)T, N, N, sCol, -inv, sCol, 0.0, k, sgi, wave, lane);
    else wide_rank1<false>(T, N, N, sCol, -inv, sCol, 0.0, k, sgi, wave, lane);
    __syncthreads();
  };
  int iters = 0, n_outer = 0, n_piv = 0;
  const int max_iters = 20 * (N + 4);
  // phase 0: bring every dof into the basis (x0 = −H⁻¹c), no ratio tests.  Four dofs per pass where the staging vectors exist:
  // with K the block, U = T[:, K], D = T[K, K], E = D⁻¹ the four sweeps in a row are  T ← T − U·E·Uᵀ off the block,
  // T[K, :] ← E·Uᵀ, T[K, K] ← −E, and the four steps are z_K ← −E·w_K with every other index moved along U·z_K
  // (the state after the block is the one the four single pivots reach — K basic, w_K = 0 — so only rounding differs).
  int k_first = 0;
  if (X.o_blk >= 0) {
    lds_f64* const U = (lds_f64*)(smem + X.o_blk);
    lds_f64* const Vt = U + 4 * X.blk_stride;
    const int bs = X.blk_stride;
    for (; k_first + 4 <= nv; k_first += 4) {
      const int k0 = k_first;
      double wK[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) wK[c] = sW[k0 + c];
      for (int c = 0; c < 4; ++c)
        for (int i = tid; i < N; i += NT_) U[c * bs + i] = T[(size_t)(k0 + c) * N + i];
      __syncthreads();
      // E = D⁻¹: the four sweeps on the 4 × 4 block itself (every thread, redundantly — the values are uniform), which leave −D⁻¹;
      // D is a principal block of the Schur complement of an SPD H, so every pivot is positive unless H is not
      double M[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) M[a][b] = U[a * bs + k0 + b];
      bool pd = true;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double d = M[q][q];
        pd = pd && d > 0.0;
        const double inv = 1.0 / d;
        double rq[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) rq[b] = M[q][b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (a != q && b != q) M[a][b] = fma(-(rq[a] * rq[b]), inv, M[a][b]);
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (b != q) { M[q][b] = rq[b] * inv; M[b][q] = rq[b] * inv; }
        M[q][q] = -inv;
      }
      if (!pd) { status |= 4; break; }
      double E[4][4], a4[4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) E[a][b] = -M[a][b];
#pragma unroll
      for (int a = 0; a < 4; ++a) a4[a] = -(E[a][0] * wK[0] + E[a][1] * wK[1] + E[a][2] * wK[2] + E[a][3] * wK[3]);
      // the step and V = U·E, by the owner of every index
      for (int i = tid; i < N; i += NT_) {
        const double ui[4] = {U[i], U[bs + i], U[2 * bs + i], U[3 * bs + i]};
        const double sstep = a4[0] * ui[0] + a4[1] * ui[1] + a4[2] * ui[2] + a4[3] * ui[3];
        const int st = sState[i];
        if (st == WS_FREE || st == WS_ROW_ON) sZ[i] -= sstep; else sW[i] += sstep;
        if ((unsigned)(i - k0) < 4u) { sZ[i] += a4[(i - k0) & 3]; sW[i] = 0.0; sState[i] = WS_FREE; }
#pragma unroll
        for (int c = 0; c < 4; ++c) Vt[4 * i + c] = E[c][0] * ui[0] + E[c][1] * ui[1] + E[c][2] * ui[2] + E[c][3] * ui[3];
      }
      __syncthreads();
      if (X.in_lds) wide_rank4((lds_f64*)T, N, N, U, bs, Vt, k0, E, wave, lane);
      else wide_rank4(T, N, N, U, bs, Vt, k0, E, wave, lane);
      __syncthreads();
    }
  }
  for (int k = k_first; k < nv && !(status & 14); ++k) {
    const double wk = sW[k];
    take_column(k);
    const double d = sCol[k];
    if (!(d > 0.0)) { status |= 4; break; }
    take_step(k, -wk / d, false);
    if (tid == (k & (NT_ - 1))) { sW[k] = 0.0; sState[k] = WS_FREE; }      // (k's owner, behind its own step)
    sweep(k, false);
  }
  __syncthreads();
  for (int i = tid; i < N; i += NT_) { const double r = fabs(T[(size_t)i * N + i]); sRef[i] = r == 0.0 ? 1.0 : r; }
  __syncthreads();
  if (clk && tid == 0) clk[10] = (long long)__builtin_readcyclecounter();
  while (!(status & 14)) {
    // most violated primal condition among basic dofs / inactive rows
    double bv = 0.0; int bi = -1;
    for (int i = tid; i < N; i += NT_) {
      double v = -kInf;
      if (sState[i] == WS_FREE) v = fmax(sZ[i] - sHi[i], sLo[i] - sZ[i]);
      else if (sState[i] == WS_ROW_OFF) v = sW[i] / sRown[i];
      if (v > 1e-12 && (bi < 0 || v > bv)) { bv = v; bi = i; }
    }
    double pv; int p;
    block_arg(bv, bi, true, pv, p);
    if (p < 0) break;                                    // optimal
    ++n_outer;
    const bool p_basic = sState[p] == WS_FREE;
    const bool upper = p_basic && (sZ[p] - sHi[p] > sLo[p] - sZ[p]);
    const double beta = p_basic ? (upper ? sHi[p] : sLo[p]) : 0.0;
    const double sgn = p_basic ? (upper ? -1.0 : 1.0) : 1.0;
    for (;;) {
      if (++iters > max_iters) { status |= 8; break; }
      take_column(p);
      const double tpp = sCol[p];
      if (!p_basic && !(-tpp > 1e-6 * sRef[p])) status |= 32;       // (MKH_ST_DEGENERATE: ik_kernel.h has the reasoning)
      double full = kInf;
      if (p_basic) { if (fabs(tpp) > 1e-12 * sRef[p]) full = (sZ[p] - beta) / tpp; }
      else if (-tpp > 1e-12 * sRef[p]) full = -sW[p] / tpp;
      const double t2 = fabs(full);
      // ratio test on dual feasibility, α = sgn·t, t ≥ 0
      double t1v = kInf; int t1i = -1;
      for (int i = tid; i < N; i += NT_) {
        if (i == p) continue;
        const double r = sgn * sCol[i];
        const int st = sState[i];
        double t = kInf;
        if (st == WS_ROW_ON) { if (r > 0.0) t = sZ[i] / r; }
        else if (st == WS_AT_HI) { if (r > 0.0) t = -sW[i] / r; }
        else if (st == WS_AT_LO) { if (r < 0.0) t = sW[i] / -r; }
        if (t < kInf && (t1i < 0 || t < t1v)) { t1v = t; t1i = i; }
      }
      double t1; int l;
      block_arg(t1v, t1i, false, t1, l);
      if (l < 0) t1 = kInf;
      if (!(fmin(t1, t2) < kInf)) { status |= 2; break; }            // no step possible: infeasible
      if (t2 <= t1) {
        take_step(p, sgn * t2, p_basic);
        if (tid == (p & (NT_ - 1))) {
          if (p_basic) { sZ[p] = beta; sState[p] = upper ? WS_AT_HI : WS_AT_LO; }
          else { sW[p] = 0.0; sState[p] = WS_ROW_ON; }
        }
        sweep(p, p_basic);
        ++n_piv;
        break;
      }
      const bool l_row = sState[l] == WS_ROW_ON;
      __syncthreads();                                   // (every thread has read l's state)
      take_step(p, sgn * t1, p_basic);
      if (tid == (l & (NT_ - 1))) {
        if (l_row) { sZ[l] = 0.0; sState[l] = WS_ROW_OFF; }
        else { sW[l] = 0.0; sState[l] = WS_FREE; }
      }
      __syncthreads();                                   // (column p is consumed: sCol is staged again)
      take_column(l);
      sweep(l, l_row);
      ++n_piv;
    }
  }
  return WideQpOut{status, iters, n_outer, n_piv};
}

// The dense Goldfarb–Idnani iteration with orthogonal factors — the algorithm of quadprog, the reference's QP backend
// (mink/solve_ik.py:101; Goldfarb & Idnani 1983: H = L·Lᵀ, J = L⁻ᵀ·Q, the active normals N = J⁻ᵀ·[R; 0]) — for the instances
// the sweep tableau flags MKH_ST_DEGENERATE (or fails on): active rows that are almost linearly dependent, where the tableau's
// explicit inverse Schur complement has lost its digits and a QR-updated factorisation has not.  Box bounds are constraints
// ±e_i like any other.  Rare path (5 % of the ALOHA workload, none elsewhere): parallel over the workgroup in the O(n²) steps,
// one thread for the O(n) bookkeeping; J and R live in the workgroup's slice of device memory.
// In: T = K = [[H, Aᵀ],[A, 0]] as built (no pivots yet), sW = (c, −h), sLo / sHi, sRown (row norms).  Out: sZ[0, nv) = Δq.
__device__ __attribute__((noinline)) WideQpOut wide_qp_dense(WideQpCtx X, double* ws, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, N = X.N, n = X.nv, m = X.N - X.nv;
  constexpr int NT_ = kWideThreads;
  const double kInf = __builtin_huge_val();
  double* const xv = smem + X.o_z;                      // x
  const double* const cw = smem + X.o_w;               // (c, −h)
  const double* const lo = smem + X.o_lo;
  const double* const hi = smem + X.o_hi;
  const double* const rown = smem + X.o_rown;
  double* const dv = smem + X.o_col;                    // d = Jᵀn⁺
  double* const zv = smem + X.o_ref;                    // z = J₂·d₂
  double* const rv = smem + X.o_rown;                   // r = R⁻¹d₁ in [0, nv): the row norms sit behind, in [nv, N)
  double* const sRed = smem + X.o_red;
  int* const sRedI = reinterpret_cast<int*>(sRed + NT_);
  int* const act = reinterpret_cast<int*>(smem + X.o_state);      // dof i: bit 0 lower bound active, 1 upper active, 2 / 3 lower / upper set aside; row s (at nv + s): 1 active, 2 set aside
  const double* const T = X.T;
  const int ld = n | 1;                                 // row stride of J and R: odd, so that a column walks all LDS banks
  double* const J = ws;                                 // n × n, row-major (stride ld)
  double* const R = ws + (size_t)n * ld;                // upper triangular, R[i·ld + k], i ≤ k
  double* const u = R + (size_t)n * ld;                 // multipliers of the active set (+ u⁺)
  int* const A = reinterpret_cast<int*>(u + n + 2);     // active constraints, ordered: code 0 … m−1 rows, m + i upper of dof i, m + n + i lower
  int status = 0, iters = 0, n_outer = 0, n_piv = 0;
  // (clock stamps of the sections, MKH_DEBUG_CLOCKS: slots 17 … 22 of the problem's row — factors, violated-constraint search,
  //  directions, step lengths, add, drop)
  long long t_sec[6] = {0, 0, 0, 0, 0, 0}, t_last = clk ? (long long)__builtin_readcyclecounter() : 0;
#define MKH_DSEC(k) do { if (clk) { const long long now_ = (long long)__builtin_readcyclecounter(); t_sec[k] += now_ - t_last; t_last = now_; } } while (0)
  auto block_arg_min = [&](double val, int idx, double& best, int& besti) {
    const double key = idx < 0 ? -kInf : -val;
    const double mk = wave_max(key);
    const unsigned cand = (idx >= 0 && key == mk) ? (unsigned)idx : 0xffffffffu;
    const unsigned mi = wave_min_u32(cand);
    __syncthreads();
    if (lane == 0) { sRed[wave] = mk; sRedI[wave] = (int)mi; }
    __syncthreads();
    double b = -kInf; int bi = -1;
#pragma unroll
    for (int w = 0; w < NT_ / 64; ++w) {
      const double kw = sRed[w]; const int iw = sRedI[w];
      if (iw >= 0 && (bi < 0 || kw > b || (kw == b && iw < bi))) { b = kw; bi = iw; }
    }
    best = -b; besti = bi;
    __syncthreads();
  };
  // normal of constraint `code` (n⁺ᵀx ≥ b⁺ form), entry k, and its right-hand side
  auto normal = [&](int code, int k) -> double {
    if (code < m) return -T[(size_t)(n + code) * N + k];
    if (code < m + n) return (k == code - m) ? -1.0 : 0.0;
    return (k == code - m - n) ? 1.0 : 0.0;
  };
  auto rhs = [&](int code) -> double {
    if (code < m) return cw[n + code];                    // −h
    if (code < m + n) return -hi[code - m];
    return lo[code - m - n];
  };
  // ---- H = L·Lᵀ (L in the R buffer), J = L⁻ᵀ
  for (int e = tid; e < n * n; e += NT_) { const int i_ = e / n, j_ = e % n; R[(size_t)i_ * ld + j_] = T[(size_t)i_ * N + j_]; J[(size_t)i_ * ld + j_] = 0.0; }
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const double dkk = R[(size_t)k * ld + k];
    if (!(dkk > 0.0)) { status |= 4; break; }             // (uniform: every thread reads the same entry)
    const double lkk = sqrt(dkk);
    __syncthreads();
    for (int i = k + tid; i < n; i += NT_) R[(size_t)i * ld + k] = (i == k) ? lkk : R[(size_t)i * ld + k] / lkk;
    __syncthreads();
    for (int e = tid; e < (n - k - 1) * (n - k - 1); e += NT_) {
      const int i = k + 1 + e / (n - k - 1), j = k + 1 + e % (n - k - 1);
      if (j <= i) R[(size_t)i * ld + j] -= R[(size_t)i * ld + k] * R[(size_t)j * ld + k];
    }
    __syncthreads();
  }
  if (status) return WideQpOut{status, 0, 0, 0};
  // X = L⁻¹ column by column (one thread per column), J = Xᵀ
  for (int j = tid; j < n; j += NT_) {
    J[(size_t)j * ld + j] = 1.0 / R[(size_t)j * ld + j];
    for (int i = j + 1; i < n; ++i) {
      double sacc = 0.0;
      for (int k = j; k < i; ++k) sacc += R[(size_t)i * ld + k] * J[(size_t)j * ld + k];      // X[k][j] is stored as J[j][k]
      J[(size_t)j * ld + i] = -sacc / R[(size_t)i * ld + i];
    }
  }
  __syncthreads();
  for (int e = tid; e < n * ld; e += NT_) R[e] = 0.0;
  // x = −J·Jᵀ·c
  for (int j = tid; j < n; j += NT_) { double sacc = 0.0; for (int i = 0; i < n; ++i) sacc += J[(size_t)i * ld + j] * cw[i]; dv[j] = sacc; }
  __syncthreads();
  for (int i = tid; i < n; i += NT_) { double sacc = 0.0; for (int k = 0; k < n; ++k) sacc += J[(size_t)i * ld + k] * dv[k]; xv[i] = -sacc; }
  for (int i = tid; i < N; i += NT_) act[i] = 0;
  __syncthreads();
  MKH_DSEC(0);
  if (clk && tid == 0) clk[10] = (long long)__builtin_readcyclecounter();
  int nact = 0;
  const int mc = m + 2 * n, max_iters = 50 * (n + mc);
  while (!(status & 14)) {
    // ---- step 1: the most violated constraint (normalised slack; quadprog's rule)
    double bv = 0.0; int bi = -1;
    for (int code = tid; code < mc; code += NT_) {
      double sl, nrm = 1.0, bb;
      if (code < m) {
        if (act[n + code]) continue;
        bb = cw[n + code];
        if (!(bb > -kInf)) continue;                      // (h = +inf: an inactive row)
        nrm = rown[n + code];
        double sacc = 0.0;
        for (int k = 0; k < n; ++k) sacc += T[(size_t)(n + code) * N + k] * xv[k];
        sl = -sacc - bb;
      } else if (code < m + n) {
        const int i = code - m;
        if ((act[i] & (2 | 8)) || !(hi[i] < kInf)) continue;
        bb = -hi[i]; sl = -xv[i] - bb;
      } else {
        const int i = code - m - n;
        if ((act[i] & (1 | 4)) || !(lo[i] > -kInf)) continue;
        bb = lo[i]; sl = xv[i] - bb;
      }
      const double v = sl / nrm;
      if (v < -1e-12 * fmax(1.0, fabs(bb) / nrm) && (bi < 0 || v < bv)) { bv = v; bi = code; }
    }
    double pv; int p;
    block_arg_min(bv, bi, pv, p);
    MKH_DSEC(1);
    if (p < 0) break;                                     // optimal
    ++n_outer;
    if (tid == 0) u[nact] = 0.0;
    for (;;) {
      if (++iters > max_iters) { status |= 8; break; }
      // ---- step 2a: d = Jᵀn⁺, z = J₂·d₂, r = R⁻¹·d₁
      __syncthreads();
      for (int j = tid; j < n; j += NT_) {
        double sacc = 0.0;
        if (p < m) { for (int i = 0; i < n; ++i) sacc -= J[(size_t)i * ld + j] * T[(size_t)(n + p) * N + i]; }
        else if (p < m + n) sacc = -J[(size_t)(p - m) * ld + j];
        else sacc = J[(size_t)(p - m - n) * ld + j];
        dv[j] = sacc;
      }
      __syncthreads();
      for (int i = tid; i < n; i += NT_) {
        double sacc = 0.0;
        for (int k = nact; k < n; ++k) sacc += J[(size_t)i * ld + k] * dv[k];
        zv[i] = sacc;
        if (i < nact) rv[i] = dv[i];
      }
      __syncthreads();
      if (nact <= 64) {
        // back-substitution inside ONE wavefront: lane i carries r_i in a register, the pivot travels by v_readlane — no barrier per
        // column (two 4-wave barriers per column were most of an iteration's synchronisation: 2·nact of ≈ 2·nact + 10)
        if (wave == 0) {
          // (the diagonal's reciprocals by their lanes at once, and the column of step k − 1 on its way while step k computes:
          //  a division and an LDS round trip per step were half of this loop)
          double ri = lane < nact ? rv[lane] : 0.0;
          const double invd = lane < nact ? 1.0 / R[(size_t)lane * ld + lane] : 0.0;
          double rnext = lane < nact ? R[(size_t)lane * ld + nact - 1] : 0.0;
          for (int k = nact - 1; k >= 0; --k) {
            const double rcol = rnext;
            if (k > 0) rnext = lane < nact ? R[(size_t)lane * ld + k - 1] : 0.0;
            const double rk = readlane_f64(ri, k) * readlane_f64(invd, k);
            if (lane < k) ri -= rk * rcol;
            if (lane == k) ri = rk;
          }
          if (lane < nact) rv[lane] = ri;
        }
        __syncthreads();
      } else {
      for (int k = nact - 1; k >= 0; --k) {               // back-substitution, column-oriented
        if (tid == 0) rv[k] = rv[k] / R[(size_t)k * ld + k];
        __syncthreads();
        for (int i = tid; i < k; i += NT_) rv[i] -= rv[k] * R[(size_t)i * ld + k];
        __syncthreads();
      }
      }
      MKH_DSEC(2);
      // ---- step 2b: step lengths — wave 0, one constraint / one dof per lane (one thread walking them: 8.7 k cycles per iteration,
      //      a third of this function on ALOHA)
      if (wave == 0) {
        double t1l = kInf; int ll = -1;
        for (int k = lane; k < nact; k += 64) {
          const double rk = rv[k];
          if (rk > 0.0) { const double tk = u[k] / rk; if (tk < t1l) { t1l = tk; ll = k; } }
        }
        const double t1w = -wave_max(-t1l);
        const unsigned lw = wave_min_u32((ll >= 0 && t1l == t1w) ? (unsigned)ll : 0xffffffffu);       // (lowest index on ties)
        double dd2 = 0.0, dd = 0.0, sp = 0.0;
        for (int k = lane; k < n; k += 64) { const double dk = dv[k]; dd += dk * dk; if (k >= nact) dd2 += dk * dk; sp += normal(p, k) * xv[k]; }
        dd = wave_sum(dd); dd2 = wave_sum(dd2); sp = wave_sum(sp) - rhs(p);
        const double t2 = (dd2 > 1e-24 * dd) ? -sp / dd2 : kInf;
        if (lane == 0) { sRed[0] = t1w; sRed[1] = t2; sRed[2] = sqrt(dd2); sRedI[8] = (int)lw; }
      }
      __syncthreads();
      const double t1 = sRed[0], t2 = sRed[1], nd2 = sRed[2];
      const int l = sRedI[8];
      const double t = fmin(t1, t2);
      __syncthreads();
      const bool dual_only = !(t2 < kInf), full = !dual_only && t2 <= t1;
      // A constraint whose normal lies in the span of the active ones and which is violated only at rounding level (≤ 1e-9 of a
      // joint step, normalised) is CONSISTENT with them — exact arithmetic would show slack 0 — and is set aside: pursuing it means
      // dual steps of 1e8 and more along directions of length 1e-17 (seen on ALOHA: two geom pairs of one body pair with opposite
      // normals and h = 0, a third almost parallel to them), which quadprog's own factors only survive by luck.
      if (dual_only && -pv <= 1e-9) {
        if (tid == 0) { if (p < m) act[n + p] = 2; else if (p < m + n) act[p - m] |= 8; else act[p - m - n] |= 4; }
        __syncthreads();
        break;
      }
      if (!(t < kInf)) { status |= 2; break; }            // constraints are inconsistent
      if (!dual_only) for (int i = tid; i < n; i += NT_) xv[i] += t * zv[i];
      if (wave == 0) { for (int k = lane; k < nact; k += 64) u[k] -= t * rv[k]; if (lane == 0) u[nact] += t; }
      __syncthreads();
      MKH_DSEC(3);
      if (full) {
        // ---- add p: one Householder reflection on the columns nact … n−1 of J takes d₂ to ±‖d₂‖·e₁ (the Givens sequence of the
        // textbook in one parallel step); R gets the column (d₁, ∓‖d₂‖)
        const double d0 = dv[nact], beta = d0 >= 0.0 ? -nd2 : nd2;      // v = d₂ − β·e₁, no cancellation
        const double vtv = 2.0 * (nd2 * nd2 - d0 * beta);
        if (vtv > 0.0) {
          for (int i = tid; i < n; i += NT_) {
            double wacc = J[(size_t)i * ld + nact] * (d0 - beta);
            for (int k = nact + 1; k < n; ++k) wacc += J[(size_t)i * ld + k] * dv[k];
            const double f = 2.0 * wacc / vtv;
            J[(size_t)i * ld + nact] -= f * (d0 - beta);
            for (int k = nact + 1; k < n; ++k) J[(size_t)i * ld + k] -= f * dv[k];
          }
        }
        for (int i = tid; i <= nact; i += NT_) R[(size_t)i * ld + nact] = (i < nact) ? dv[i] : beta;
        if (tid == 0) { A[nact] = p; if (p < m) act[n + p] = 1; else if (p < m + n) act[p - m] |= 2; else act[p - m - n] |= 1; }
        ++nact; ++n_piv;
        __syncthreads();
        MKH_DSEC(4);
        break;
      }
      // ---- drop the blocking constraint l (partial step, or a step in the dual space only), then try p again
      if (wave == 0) {                                    // (the lists close up: a lane per entry, loads before stores)
        if (lane == 0) {
          const int code = A[l];
          if (code < m) act[n + code] = 0; else if (code < m + n) act[code - m] &= ~2; else act[code - m - n] &= ~1;
        }
        for (int k0 = l; k0 < nact; k0 += 64) {
          const int k = k0 + lane;
          const bool on = k < nact;
          const double uu = on ? u[k + 1] : 0.0;
          const int aa = (on && k < nact - 1) ? A[k + 1] : 0;
          if (on) u[k] = uu;
          if (on && k < nact - 1) A[k] = aa;
        }
      }
      // columns shift left: upper Hessenberg from l on — every entry at once when the block fits the workgroup
      if ((nact - 1 - l) * nact <= NT_) {
        const int kk = l + tid / nact, ii = tid % nact;
        const bool on = tid < (nact - 1 - l) * nact && ii <= kk + 1;
        const double val = on ? R[(size_t)ii * ld + kk + 1] : 0.0;
        __syncthreads();
        if (on) R[(size_t)ii * ld + kk] = val;
      } else {
        for (int k = l; k < nact - 1; ++k) {
          __syncthreads();
          for (int i = tid; i <= k + 1; i += NT_) R[(size_t)i * ld + k] = R[(size_t)i * ld + k + 1];
        }
      }
      __syncthreads();
      for (int i = tid; i < n; i += NT_) R[(size_t)i * ld + nact - 1] = 0.0;
      --nact; ++n_piv;
      for (int k = l; k < nact; ++k) {                    // Givens on rows (k, k+1) of R, columns (k, k+1) of J
        __syncthreads();
        const double a = R[(size_t)k * ld + k], b = R[(size_t)(k + 1) * ld + k];
        const double hh = hypot(a, b);
        const double cs = b == 0.0 ? 1.0 : a / hh, sn = b == 0.0 ? 0.0 : b / hh;
        __syncthreads();
        if (b != 0.0) {
          for (int j = k + tid; j < nact; j += NT_) {
            const double ra = R[(size_t)k * ld + j], rb = R[(size_t)(k + 1) * ld + j];
            R[(size_t)k * ld + j] = (j == k) ? hh : cs * ra + sn * rb;
            R[(size_t)(k + 1) * ld + j] = (j == k) ? 0.0 : -sn * ra + cs * rb;
          }
          for (int i = tid; i < n; i += NT_) {
            const double ja = J[(size_t)i * ld + k], jb = J[(size_t)i * ld + k + 1];
            J[(size_t)i * ld + k] = cs * ja + sn * jb;
            J[(size_t)i * ld + k + 1] = -sn * ja + cs * jb;
          }
        }
      }
      __syncthreads();
      MKH_DSEC(5);
    }
  }
  __syncthreads();
  if (clk && tid == 0) for (int k = 0; k < 6; ++k) clk[17 + k] = t_sec[k];
#undef MKH_DSEC
  return WideQpOut{status, iters, n_outer, n_piv};
}

// (two workgroups per CU: every phase of this kernel is latency-bound — one wavefront per SIMD waits out each LDS round trip and
//  barrier alone — so the second resident workgroup is worth more than the registers it costs)
template <bool CVX>
__device__ __forceinline__ void wide_kernel_body(const WideProblem* __restrict__ Pg, SolveArgs& A, const TapArgs* __restrict__ tp) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // A redo launch whose slice of the batch carries no flagged status — the usual case — ends HERE, before the layout is read and
  // before the first spilled register is stored: the stores of the body's prologue made an idle launch write 66 KB of scratch per
  // workgroup (34 MB per launch in the WRITE_SIZE counter, r05 profiles of shadow_c4 / g1_coll / ur5e_coll).
  // (A workgroup's share of the statuses: chunks of 16 — one 64-byte line — dealt round-robin, so that a cluster of flagged
  //  instances spreads over the workgroups.)
  if (A.redo_mask) {
    const int nck = (A.B + 15) >> 4;
    int any = 0;
    for (int c = (int)blockIdx.x + ((int)threadIdx.x >> 4) * (int)gridDim.x; c < nck; c += (kWideThreads / 16) * (int)gridDim.x) {
      const int i = c * 16 + ((int)threadIdx.x & 15);
      if (i < A.B) any |= A.status_out[i] & A.redo_mask;
    }
    if (!__syncthreads_or(any)) return;
  }
  const WideProblem& P = *Pg;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, NT_ = kWideThreads;
  const int nq = P.nq, nv = P.nv, nbody = P.nbody, XS = nbody;
  const double kInf = __builtin_huge_val();
  double* const sq = smem + P.o_q;
  double* const sX = smem + P.o_X;                       // body poses, component-major: X[c][body]
  double* const sJnt = smem + P.o_jnt;                   // per joint: axis, anchor (world)
  double* const sDof = smem + P.o_dof;                   // per dof: ang, lin, anchor, q
  double* const sTask = smem + P.o_task;                 // per frame task: the 64-double block of ik_kernel.h's task lanes
  double* const sCom = smem + P.o_com;                   // per body: subtree CoM, subtree mass
  double* const sWe = smem + P.o_we;                     // weighted error of every Jacobian row (compact rows + dense rows)
  double* const sC = smem + P.o_c;                       // c
  double* const sHd = smem + P.o_hd;                     // diagonal part of H owned by the posture tasks
  double* const sZ = smem + P.o_z;
  double* const sW = smem + P.o_w;
  double* const sLo = smem + P.o_lo;
  double* const sHi = smem + P.o_hi;
  double* const sRown = smem + P.o_rown;
  double* const sRef = smem + P.o_ref;
  double* const sCol = smem + P.o_col;
  double* const sRed = smem + P.o_red;                   // 2 × 256: reductions
  int* const sRedI = reinterpret_cast<int*>(sRed + NT_);
  int* const sState = reinterpret_cast<int*>(smem + P.o_state);
  double* const wsb = P.ws + (size_t)blockIdx.x * P.ws_stride;
  double* const Jw = wsb + P.ws_jw;                      // weighted Jacobian rows [R][nv]
  double* const rec = wsb + P.ws_rec;                    // per pair: h, n, from, to (10 doubles)
  int* const rowpair = reinterpret_cast<int*>(wsb + P.ws_rowpair);
  int* const rowrank = reinterpret_cast<int*>(wsb + P.ws_rank);
  double* const T_lds = smem + P.o_T;
  double* const T_mem = wsb + P.ws_T;
  const int R_task = P.n_jrows, R_all = P.n_jrows + P.n_dense_rows;
  // parity taps (mkh_eval): body poses, frame poses, subtree CoM, H, c, the box, the contact rows — what the reference's
  // Configuration / build_ik expose; the per-task (e, J) taps are the wavefront kernels'
#define MKH_WTAP(f) (tp ? tp->f : nullptr)
  // phase stamps (SolveArgs::clk, MKH_DEBUG_CLOCKS=<file>: tools/wide_phase_clocks.py): slot k of row pb = shader clock at phase
  // boundary k of the problem's LAST step — 0 start, 1 FK, 2 dof axes / CoM, 3 task lanes, 4 posture / LM, 5 Jacobian rows, 6 c / box,
  // 7 contacts, 8 rows selected, 9 tableau built, 10 phase 0 done, 11 active set done, 12 end
#define MKH_WSTAMP(k) do { if (A.clk && tid == 0) A.clk[(size_t)pb * 24 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)

  auto block_sum = [&](double x) -> double {
    x = wave_sum(x);
    __syncthreads();
    if (lane == 0) sRed[wave] = x;
    __syncthreads();
    return (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
  };
  // A redo launch (redo_mask != 0) solves only the instances whose status carries one of the mask's bits.  Every workgroup reads
  // its share of the statuses 256 at a time (round 4 walked them one dependent load after the other: 29 µs for 16 384 instances)
  // and PUSHES what it finds into one queue of the launch; then the workgroups that found anything take instances from that
  // queue until it is empty.  (A workgroup solving a contiguous slice of its own: ALOHA's 519 flagged instances of 16 384 come
  // in clusters — the launch took six dense solves one after the other where two would do.  Workgroups with an empty slice never get here (the early exit above), so a launch that finds nothing, the usual
  // case, pays for none of this; a ticket counter over ALL workgroups instead cost such launches +8 µs.)
  // The queue is a ring that is never reset: head ≤ tail always (an entry is taken by compare-and-swap on head, never past
  // tail), a workgroup leaves when head = tail, and one that pushes later drains what it pushed itself if nobody else does.
  int* const sList = reinterpret_cast<int*>(sRed + 16);  // flagged instances of the current chunk (≤ 256; sRed[0, 16) carry the reductions)
  if (A.redo_mask) {
    const int nck = (A.B + 15) >> 4;
    for (int c0 = (int)blockIdx.x; c0 < nck; c0 += (NT_ / 16) * (int)gridDim.x) {
      __syncthreads();
      if (tid == 0) sRedI[15] = 0;
      __syncthreads();
      const int c = c0 + (tid >> 4) * (int)gridDim.x;
      const int i = c * 16 + (tid & 15);
      if (c < nck && i < A.B && (A.status_out[i] & A.redo_mask)) sList[atomicAdd(&sRedI[15], 1)] = i;
      __syncthreads();
      const int n = sRedI[15];
      if (n > 0) {
        if (tid == 0) sRedI[14] = (int)__hip_atomic_fetch_add(&P.redo_ctr[1], (uint32_t)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t base = (uint32_t)sRedI[14];
        if (tid < n) __hip_atomic_store(&P.redo_queue[(base + (uint32_t)tid) & (P.redo_cap - 1u)], sList[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  for (;;) {
  {
    int pb;
    if (A.redo_mask) {
      __syncthreads();
      if (tid == 0) {
        int item = -1;
        for (;;) {
          const uint32_t h = __hip_atomic_load(&P.redo_ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t t = __hip_atomic_load(&P.redo_ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int32_t)(t - h) <= 0) break;
          uint32_t expect = h;
          if (!__hip_atomic_compare_exchange_strong(&P.redo_ctr[0], &expect, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
          int32_t* const slot = &P.redo_queue[h & (P.redo_cap - 1u)];
          while ((item = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) __builtin_amdgcn_s_sleep(2);   // (reserved, on its way)
          __hip_atomic_store(slot, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        sRedI[14] = item;
      }
      __syncthreads();
      pb = sRedI[14];
      if (pb < 0) break;
    } else {
      // As THE path of a model (redo_mask = 0): every problem from the ticket counter — QP work varies by ±9 % per problem, a workgroup does 16 of an 8 192-instance batch, and with static
      // strides the launch waited for the unluckiest of 512 sums.  Self-resetting like the wavefront kernels' counter: every
      // workgroup ends on exactly one rejected draw, the last of them zeroes it.
      __syncthreads();
      if (tid == 0) {
        const unsigned tk = atomicAdd(A.work_counter, 1u);
        if (tk == (unsigned)A.B + gridDim.x - 1u) atomicExch(A.work_counter, 0u);
        sRedI[14] = (int)tk;
      }
      __syncthreads();
      pb = sRedI[14];
      if (pb >= A.B) break;
    }
    __syncthreads();
    // ------------------------------------------------------------ inputs
    for (int i = tid; i < nq; i += NT_) sq[i] = A.q[(size_t)pb * nq + i];
    __syncthreads();
    // Fused outer loop (mkh_solve_steps / mkh_solve_until; the semantics of ik_kernel.h's loop): q stays in LDS between steps,
    // status is the OR over the steps, an instance stops at the first step whose QP fails; with thresholds the test sits
    // behind the frame-task errors of the NEXT step (the error at the integrated q), one check-only pass after the last
    // allowed iteration.
    const int n_steps = A.n_steps > 1 ? A.n_steps : 1;
    const bool until = A.pos_threshold >= 0.0;
    const bool integrate = A.n_steps > 1 || A.q_out != nullptr;
    int status_all = 0, it_done = 0, conv_flag = 0;
    for (int step = 0; step < n_steps + (until ? 1 : 0); ++step) {
    int status = 0;
    MKH_WSTAMP(0);
    // ------------------------------------------------------------ FK, level by level (mj_kinematics)
    for (int lv = 0; lv < P.nlevels; ++lv) {
      for (int idx = P.level_start[lv] + tid; idx < P.level_start[lv + 1]; idx += NT_) {
        const int b = P.level_body[idx];
        V3 xp{0, 0, 0};
        Q4 xq{1, 0, 0, 0};
        if (b > 0) {
          const int pa = P.body_parent[b];
          const Q4 pq{sX[3 * XS + pa], sX[4 * XS + pa], sX[5 * XS + pa], sX[6 * XS + pa]};
          xp = V3{sX[pa], sX[XS + pa], sX[2 * XS + pa]} + qrot(pq, V3{P.body_pos[3 * b], P.body_pos[3 * b + 1], P.body_pos[3 * b + 2]});
          xq = qmul(pq, Q4{P.body_quat[4 * b], P.body_quat[4 * b + 1], P.body_quat[4 * b + 2], P.body_quat[4 * b + 3]});
          const int jadr = P.body_jntadr[b], jnum = P.body_jntnum[b];
          for (int jn = 0; jn < jnum; ++jn) {
            const int j = jadr + jn, jt = P.jnt_type[j], qa = P.jnt_qadr[j];
            const V3 axl{P.jnt_axis[3 * j], P.jnt_axis[3 * j + 1], P.jnt_axis[3 * j + 2]};
            const V3 jp{P.jnt_pos[3 * j], P.jnt_pos[3 * j + 1], P.jnt_pos[3 * j + 2]};
            if (jt == JNT_FREE) {
              xp = {sq[qa], sq[qa + 1], sq[qa + 2]};
              xq = qnormalize(Q4{sq[qa + 3], sq[qa + 4], sq[qa + 5], sq[qa + 6]});
            }
            // joint axis / anchor in the world, in the frame the joint acts in (before it is applied)
            const V3 ax = qrot(xq, axl), an = xp + qrot(xq, jp);
            double* o = sJnt + j * 6;
            o[0] = ax.x; o[1] = ax.y; o[2] = ax.z; o[3] = an.x; o[4] = an.y; o[5] = an.z;
            if (jt == JNT_SLIDE) {
              xp = xp + (sq[qa] - P.jnt_qpos0[j]) * ax;
            } else if (jt == JNT_HINGE || jt == JNT_BALL) {
              const Q4 qloc = (jt == JNT_HINGE) ? axis_angle(axl, sq[qa] - P.jnt_qpos0[j])
                                                : qnormalize(Q4{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]});
              xq = qmul(xq, qloc);
              xp = an - qrot(xq, jp);
            }
          }
          xq = qnormalize(xq);
        }
        sX[b] = xp.x; sX[XS + b] = xp.y; sX[2 * XS + b] = xp.z;
        sX[3 * XS + b] = xq.w; sX[4 * XS + b] = xq.x; sX[5 * XS + b] = xq.y; sX[6 * XS + b] = xq.z;
      }
      __syncthreads();
    }
    if (MKH_WTAP(t_xpos))
      for (int e = tid; e < nbody * 3; e += NT_) MKH_WTAP(t_xpos)[(size_t)pb * nbody * 3 + e] = sX[(e % 3) * XS + e / 3];
    if (MKH_WTAP(t_xquat))
      for (int e = tid; e < nbody * 4; e += NT_) MKH_WTAP(t_xquat)[(size_t)pb * nbody * 4 + e] = sX[(3 + e % 4) * XS + e / 4];
    MKH_WSTAMP(1);
    // ------------------------------------------------------------ dof axes (cdof): jacp(p) = lin + ang × (p − anchor), jacr = ang
    bool viol = false;
    for (int d = tid; d < nv; d += NT_) {
      const int kind = P.dof_kind[d], dk = P.dof_k[d], j = P.dof_jnt[d], body = P.dof_body[d], qa = P.dof_qadr[d];
      V3 ang{0, 0, 0}, lin{0, 0, 0}, anchor{0, 0, 0};
      double qd = 0.0;
      const double* o = sJnt + j * 6;
      if (kind == DOF_HINGE) { ang = {o[0], o[1], o[2]}; anchor = {o[3], o[4], o[5]}; qd = sq[qa]; }
      else if (kind == DOF_SLIDE) { lin = {o[0], o[1], o[2]}; qd = sq[qa]; }
      else if (kind == DOF_FREE_LIN) { lin = {dk == 0 ? 1.0 : 0.0, dk == 1 ? 1.0 : 0.0, dk == 2 ? 1.0 : 0.0}; }
      else {                                               // ball / free rotational dof: body-frame axis k about the joint anchor
        const M3 Rm = qmat(Q4{sX[3 * XS + body], sX[4 * XS + body], sX[5 * XS + body], sX[6 * XS + body]});
        ang = (dk == 0) ? V3{Rm.m[0], Rm.m[3], Rm.m[6]} : ((dk == 1) ? V3{Rm.m[1], Rm.m[4], Rm.m[7]} : V3{Rm.m[2], Rm.m[5], Rm.m[8]});
        anchor = (kind == DOF_FREE_ANG) ? V3{sX[body], sX[XS + body], sX[2 * XS + body]} : V3{o[3], o[4], o[5]};
      }
      // Configuration.check_limits (mink/configuration.py:77-110), tol = 1e-6 (a limited ball joint: the quaternion's w, first dof)
      if (kind == DOF_HINGE || kind == DOF_SLIDE) viol = viol || qd < P.dof_lo[d] - 1e-6 || qd > P.dof_hi[d] + 1e-6;
      if (kind == DOF_BALL && dk == 0) viol = viol || sq[qa] < P.dof_lo[d] - 1e-6 || sq[qa] > P.dof_hi[d] + 1e-6;
      double* s = sDof + d * 10;
      s[0] = ang.x; s[1] = ang.y; s[2] = ang.z; s[3] = lin.x; s[4] = lin.y; s[5] = lin.z;
      s[6] = anchor.x; s[7] = anchor.y; s[8] = anchor.z; s[9] = qd;
    }
    if (__syncthreads_or(viol ? 1 : 0)) status |= 1;
    // ------------------------------------------------------------ subtree CoM (mj_comPos) for ComTask
    if (P.n_com > 0) {
      for (int b = tid; b < nbody; b += NT_) {             // (a subtree is a contiguous id range in MuJoCo's depth-first order)
        double sx = 0, sy = 0, sz = 0;
        for (int c = b; c <= P.body_last[b]; ++c) {
          if (!P.body_inrobot[c]) continue;
          const Q4 cq{sX[3 * XS + c], sX[4 * XS + c], sX[5 * XS + c], sX[6 * XS + c]};
          const V3 xi = V3{sX[c], sX[XS + c], sX[2 * XS + c]} + qrot(cq, V3{P.body_ipos[3 * c], P.body_ipos[3 * c + 1], P.body_ipos[3 * c + 2]});
          sx += P.body_mass[c] * xi.x; sy += P.body_mass[c] * xi.y; sz += P.body_mass[c] * xi.z;
        }
        const double stm = P.body_stmass[b];
        V3 cs;
        if (stm >= 1e-15) { const double im = fast_rcp(stm); cs = {sx * im, sy * im, sz * im}; }
        else {
          const Q4 bq{sX[3 * XS + b], sX[4 * XS + b], sX[5 * XS + b], sX[6 * XS + b]};
          cs = V3{sX[b], sX[XS + b], sX[2 * XS + b]} + qrot(bq, V3{P.body_ipos[3 * b], P.body_ipos[3 * b + 1], P.body_ipos[3 * b + 2]});
        }
        sCom[4 * b] = cs.x; sCom[4 * b + 1] = cs.y; sCom[4 * b + 2] = cs.z; sCom[4 * b + 3] = stm;
      }
      __syncthreads();
    }
    MKH_WSTAMP(2);
    // ------------------------------------------------------------ frame tasks: pose, error, jlog (the task lanes of ik_kernel.h)
    double mu_part = 0.0;                                  // Levenberg–Marquardt terms owned by this thread
    bool conv_mine = true;                                 // this thread's frame tasks are within the thresholds (rows with a nonzero cost only)
    for (int t = tid; t < P.n_frame; t += NT_) {
      const FrameTaskDev& ft = P.frame[t];
      SE3 F;
      const Q4 bq{sX[3 * XS + ft.body], sX[4 * XS + ft.body], sX[5 * XS + ft.body], sX[6 * XS + ft.body]};
      F.p = V3{sX[ft.body], sX[XS + ft.body], sX[2 * XS + ft.body]} + qrot(bq, V3{ft.lpos[0], ft.lpos[1], ft.lpos[2]});
      F.q = qmul(bq, Q4{ft.lquat[0], ft.lquat[1], ft.lquat[2], ft.lquat[3]});
      const double* tg = A.frame_targets + ((size_t)pb * P.n_frame + t) * 7;
      const SE3 Tt{Q4{tg[0], tg[1], tg[2], tg[3]}, V3{tg[4], tg[5], tg[6]}};
      double* o = sTask + t * 64;
      V3 ev, ew;
      double Jm[9], Qm[9];
      bool ident;
      if (!ft.relative) {
        se3_log(se3_mul(se3_inv(F), Tt), ev, ew);          // e = target.minus(frame)  (frame_task.py:119-122)
        se3_ljacinv(ev, ew, Jm, Qm, ident);                // jlog(T_tb) = ljacinv(e)   (frame_task.py:144-146)
      } else {
        const int rb = ft.root_body;
        const Q4 rq0{sX[3 * XS + rb], sX[4 * XS + rb], sX[5 * XS + rb], sX[6 * XS + rb]};
        SE3 Rt;
        Rt.p = V3{sX[rb], sX[XS + rb], sX[2 * XS + rb]} + qrot(rq0, V3{ft.root_lpos[0], ft.root_lpos[1], ft.root_lpos[2]});
        Rt.q = qmul(rq0, Q4{ft.root_lquat[0], ft.root_lquat[1], ft.root_lquat[2], ft.root_lquat[3]});
        const SE3 Tfr = se3_mul(se3_inv(Rt), F);
        se3_log(se3_mul(se3_inv(Tt), Tfr), ev, ew);        // e = T_fr.rminus(target)  (relative_frame_task.py:106-129)
        se3_ljacinv(-1.0 * ev, -1.0 * ew, Jm, Qm, ident);
        const SE3 Trf = se3_inv(Tfr);
        const M3 Rr = qmat(Rt.q), Rrf = qmat(Trf.q);
#pragma unroll
        for (int i = 0; i < 9; ++i) { o[36 + i] = Rr.m[i]; o[48 + i] = Rrf.m[i]; }
        o[45] = Rt.p.x; o[46] = Rt.p.y; o[47] = Rt.p.z;
        o[57] = Trf.p.x; o[58] = Trf.p.y; o[59] = Trf.p.z;
      }
      const M3 Rf = qmat(F.q);
#pragma unroll
      for (int i = 0; i < 9; ++i) { o[i] = Jm[i]; o[9 + i] = Qm[i]; o[18 + i] = Rf.m[i]; }
      o[27] = F.p.x; o[28] = F.p.y; o[29] = F.p.z;
      const double e6[6] = {ev.x, ev.y, ev.z, ew.x, ew.y, ew.z};
      double ss = 0.0;
      int c = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double we = ft.cost[r] * (-ft.gain * e6[r]);  // weighted_error (task.py:129-130)
        ss += we * we;
        if ((ft.rowmask >> r) & 1) { sWe[ft.jrow0 + c] = we; ++c; }
        if (MKH_WTAP(t_task_e)) MKH_WTAP(t_task_e)[(size_t)pb * P.n_rows_tap + ft.row0 + r] = e6[r];
      }
      mu_part += ft.lm_damping * ss;                        // task.py:131
      if (until)
        conv_mine = conv_mine && (!(ft.rowmask & 7) || dot(ev, ev) <= A.pos_threshold * A.pos_threshold) &&
                    (!(ft.rowmask & 56) || dot(ew, ew) <= A.ori_threshold * A.ori_threshold);
      if (MKH_WTAP(t_frame_pose)) {                          // pose of the frame in the world (RelativeFrameTask: in its root frame)
        SE3 Fo = F;
        if (ft.relative) {
          const int rb = ft.root_body;
          const Q4 rq0{sX[3 * XS + rb], sX[4 * XS + rb], sX[5 * XS + rb], sX[6 * XS + rb]};
          SE3 Rt;
          Rt.p = V3{sX[rb], sX[XS + rb], sX[2 * XS + rb]} + qrot(rq0, V3{ft.root_lpos[0], ft.root_lpos[1], ft.root_lpos[2]});
          Rt.q = qmul(rq0, Q4{ft.root_lquat[0], ft.root_lquat[1], ft.root_lquat[2], ft.root_lquat[3]});
          Fo = se3_mul(se3_inv(Rt), F);
        }
        double* t7 = MKH_WTAP(t_frame_pose) + ((size_t)pb * P.n_frame + t) * 7;
        t7[0] = Fo.q.w; t7[1] = Fo.q.x; t7[2] = Fo.q.y; t7[3] = Fo.q.z; t7[4] = Fo.p.x; t7[5] = Fo.p.y; t7[6] = Fo.p.z;
      }
    }
    if (until && step > 0) {                                 // (workgroup-uniform: every thread sees the same vote)
      it_done = step;
      if (!__syncthreads_or(conv_mine ? 0 : 1)) { conv_flag = 1; status_all |= status; break; }   // every frame task achieved
      if (step == n_steps) { status_all |= status; break; }                                        // iteration budget spent
    }
    if (MKH_WTAP(t_subtree_com) && P.n_com > 0 && tid < 3) MKH_WTAP(t_subtree_com)[(size_t)pb * 3 + tid] = sCom[P.robot_root * 4 + tid];
    // ComTask error & LM term (com_task.py:71-82)
    if (tid == 0)
      for (int t = 0; t < P.n_com; ++t) {
        const double* tg = A.com_target + (A.com_batched ? (size_t)pb * P.n_com * 3 : 0) + t * 3;
        const double* cr = sCom + P.robot_root * 4;
        double ss = 0.0;
        int c = 0;
        for (int r = 0; r < 3; ++r) {
          const double we = P.com_cost[t][r] * (-P.com_gain[t] * (cr[r] - tg[r]));
          ss += we * we;
          if (MKH_WTAP(t_task_e)) MKH_WTAP(t_task_e)[(size_t)pb * P.n_rows_tap + P.com_row0[t] + r] = cr[r] - tg[r];
          if ((P.com_rowmask[t] >> r) & 1) { sWe[P.com_jrow0[t] + c] = we; ++c; }
        }
        mu_part += P.com_lm[t] * ss;
      }
    // caller-defined tasks: weighted errors and LM terms (task.py:129-131)
    if (tid == 0)
      for (int t = 0; t < P.n_dense_tasks; ++t) {
        const double* de = A.dense_e + (size_t)pb * P.n_dense_rows + P.dense_row0[t];
        double ss = 0.0;
        for (int r = 0; r < P.dense_k[t]; ++r) {
          const double we = P.dense_wgain[P.dense_row0[t] + r] * de[r];
          sWe[R_task + P.dense_row0[t] + r] = we;
          ss += we * we;
          if (MKH_WTAP(t_task_e)) MKH_WTAP(t_task_e)[(size_t)pb * P.n_rows_tap + P.dense_tap_row0 + P.dense_row0[t] + r] = de[r];
        }
        mu_part += P.dense_lm[t] * ss;
      }
    MKH_WSTAMP(3);
    // ------------------------------------------------------------ posture tasks (diagonal; posture_task.py:87-142)
    for (int d = tid; d < nv; d += NT_) { sC[d] = 0.0; sHd[d] = 0.0; }
    __syncthreads();
    for (int t = 0; t < P.n_posture; ++t) {
      const double* tq = A.posture_target + (A.posture_batched ? ((size_t)pb * P.n_posture + t) * nq : (size_t)t * nq);
      double ssw = 0.0;
      for (int d = tid; d < nv; d += NT_) {
        const int kind = P.dof_kind[d], dk = P.dof_k[d], qa = P.dof_qadr[d];
        double e = 0.0, jd = 0.0;
        if (kind == DOF_HINGE || kind == DOF_SLIDE) { e = tq[qa] - sDof[d * 10 + 9]; jd = -1.0; }     // mj_differentiatePos, dt = 1
        else if (kind == DOF_BALL) {
          const Q4 q1{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]}, q2{tq[qa], tq[qa + 1], tq[qa + 2], tq[qa + 3]};
          const V3 dv = quat2vel(qmul(qconj(q1), q2));
          e = (dk == 0) ? dv.x : ((dk == 1) ? dv.y : dv.z);
          jd = -1.0;
        }                                                  // free-joint dofs: error and column zeroed (posture_task.py:115-116,139-141)
        const double cost = P.posture_cost[(size_t)t * nv + d];
        const double we = cost * (-P.posture_gain[t] * e), wj = cost * jd;
        if (MKH_WTAP(t_task_e)) MKH_WTAP(t_task_e)[(size_t)pb * P.n_rows_tap + P.posture_row0[t] + d] = e;
        if (MKH_WTAP(t_task_J)) MKH_WTAP(t_task_J)[((size_t)pb * P.n_rows_tap + P.posture_row0[t] + d) * nv + d] = jd;   // (the buffer arrives zeroed)
        sHd[d] += wj * wj;
        sC[d] -= we * wj;
        ssw += we * we;
      }
      if (P.posture_lm[t] != 0.0) { const double s = block_sum(ssw); if (tid == 0) mu_part += P.posture_lm[t] * s; }
    }
    const double mu_total = A.damping + block_sum(mu_part);  // solve_ik.py:16 + Σ μ_t
    MKH_WSTAMP(4);
    // ------------------------------------------------------------ weighted Jacobian rows Jw[r][k] (weighted_jacobian, task.py:129)
    for (int e = tid; e < P.n_frame * nv; e += NT_) {
      const int t = e / nv, k = e - t * nv;
      const FrameTaskDev& ft = P.frame[t];
      const bool on_f = wide_on_chain(P, ft.body, k), rel = ft.relative != 0, on_r = rel && wide_on_chain(P, ft.root_body, k);
      double Jt[6] = {0, 0, 0, 0, 0, 0};
      if (on_f || on_r) wide_frame_column(sTask + t * 64, sDof + k * 10, on_f, rel, on_r, Jt);
      if (MKH_WTAP(t_task_J)) {
#pragma unroll
        for (int r = 0; r < 6; ++r) MKH_WTAP(t_task_J)[((size_t)pb * P.n_rows_tap + ft.row0 + r) * nv + k] = Jt[r];
      }
      int c = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
        if ((ft.rowmask >> r) & 1) { Jw[(size_t)(ft.jrow0 + c) * nv + k] = ft.cost[r] * Jt[r]; ++c; }
    }
    for (int e = tid; e < P.n_com * nv; e += NT_) {        // CoM Jacobian column (mj_jacSubtreeCom closed form)
      const int t = e / nv, k = e - t * nv;
      const int body = P.dof_body[k];
      double Jt[3] = {0, 0, 0};
      if (P.body_inrobot[body]) {
        const double* cd = sCom + body * 4;
        const double* kd = sDof + k * 10;
        const double fac = cd[3] * fast_rcp(sCom[P.robot_root * 4 + 3]);
        const V3 jc = fac * (V3{kd[3], kd[4], kd[5]} + cross(V3{kd[0], kd[1], kd[2]}, V3{cd[0], cd[1], cd[2]} - V3{kd[6], kd[7], kd[8]}));
        Jt[0] = jc.x; Jt[1] = jc.y; Jt[2] = jc.z;
      }
      if (MKH_WTAP(t_task_J))
        for (int r = 0; r < 3; ++r) MKH_WTAP(t_task_J)[((size_t)pb * P.n_rows_tap + P.com_row0[t] + r) * nv + k] = Jt[r];
      int c = 0;
      for (int r = 0; r < 3; ++r)
        if ((P.com_rowmask[t] >> r) & 1) { Jw[(size_t)(P.com_jrow0[t] + c) * nv + k] = P.com_cost[t][r] * Jt[r]; ++c; }
    }
    for (int e = tid; e < P.n_dense_rows * nv; e += NT_) {  // caller-defined rows: W·J straight from memory
      const int r = e / nv, k = e - r * nv;
      const double jr = A.dense_J[((size_t)pb * P.n_dense_rows + r) * nv + k];
      Jw[(size_t)(R_task + r) * nv + k] = P.dense_cost[r] * jr;
      if (MKH_WTAP(t_task_J)) MKH_WTAP(t_task_J)[((size_t)pb * P.n_rows_tap + P.dense_tap_row0 + r) * nv + k] = jr;
    }
    __syncthreads();
    MKH_WSTAMP(5);
    // c = −weighted_errorᵀ·weighted_jacobian (task.py:133-134)
    for (int k = tid; k < nv; k += NT_) {
      double c = sC[k];
      for (int r = 0; r < R_all; ++r) c -= sWe[r] * Jw[(size_t)r * nv + k];
      sC[k] = c;
    }
    // ------------------------------------------------------------ box limits (configuration_limit.py:94-124, velocity_limit.py:96-101)
    bool box_bad = false;
    for (int d = tid; d < nv; d += NT_) {
      double lo = -kInf, hi = kInf;
      const int kind = P.dof_kind[d], dk = P.dof_k[d], qa = P.dof_qadr[d];
      const double qd = sDof[d * 10 + 9];
      for (int t = 0; t < P.n_cfg; ++t) {
        const double lw = P.cfg_lower[(size_t)t * nv + d], up = P.cfg_upper[(size_t)t * nv + d];
        if (kind == DOF_BALL) {                            // (a limited ball joint: the reference differentiates quaternions, see ik_kernel.h)
          const Q4 qc{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]};
          if (up < kInf) {
            const V3 dv = quat2vel(qmul(qconj(qc), Q4{up, up, up, up}));
            hi = fmin(hi, P.cfg_gain[t] * ((dk == 0) ? dv.x : ((dk == 1) ? dv.y : dv.z)));
          }
          if (lw > -kInf) {
            const V3 dv = quat2vel(qmul(qconj(Q4{lw, lw, lw, lw}), qc));
            lo = fmax(lo, -(P.cfg_gain[t] * ((dk == 0) ? dv.x : ((dk == 1) ? dv.y : dv.z))));
          }
          continue;
        }
        if (up < kInf) hi = fmin(hi, P.cfg_gain[t] * (up - qd));
        if (lw > -kInf) lo = fmax(lo, -(P.cfg_gain[t] * (qd - lw)));
      }
      for (int t = 0; t < P.n_vel; ++t) {
        const double vm = P.vel_limit[(size_t)t * nv + d];
        if (vm < kInf) { hi = fmin(hi, A.dt * vm); lo = fmax(lo, -(A.dt * vm)); }
      }
      if (A.dense_lo) lo = fmax(lo, A.dense_lo[(size_t)pb * nv + d]);
      if (A.dense_hi) hi = fmin(hi, A.dense_hi[(size_t)pb * nv + d]);
      sLo[d] = lo; sHi[d] = hi;
      if (MKH_WTAP(t_box_lo)) MKH_WTAP(t_box_lo)[(size_t)pb * nv + d] = lo;
      if (MKH_WTAP(t_box_hi)) MKH_WTAP(t_box_hi)[(size_t)pb * nv + d] = hi;
      box_bad = box_bad || lo > hi + 1e-12;
    }
    if (__syncthreads_or(box_bad ? 1 : 0)) status |= 2;              // inconsistent box ⇒ quadprog "constraints are inconsistent"
    MKH_WSTAMP(6);
    // ------------------------------------------------------------ contacts (collision_avoidance_limit.py:187-229): every one a row
    // (a REAL call: the distance routines — GJK, the expanding polytope — want ≈150 registers of their own; inlined, they cost the
    //  two-workgroups-per-CU build of this kernel 98 spilled VGPRs)
    if (P.n_pairs > 0) wide_contacts<CVX>(Pg, A.dt, rec, smem + P.o_X, XS, smem + P.o_cws);
    __syncthreads();
    MKH_WSTAMP(7);
    // Rows: the detected contacts in pair order, then the caller's rows with a finite bound.  More contacts in range than the
    // workspace has rows (kWideMaxRows): the rule of the wavefront kernels (collision_phase) — the max_rows TIGHTEST (smallest
    // h, ties by pair index) become rows, the rest are checked at the solution below; a dropped one that holds there was
    // inactive, so the result is still the reference's (mink/solve_ik.py:25-40 stacks every row).
    int n_det = 0;
    for (int pi = tid; pi < P.n_pairs; pi += NT_) n_det += rec[(size_t)pi * 10] < kInf ? 1 : 0;
    n_det = (int)(block_sum((double)n_det) + 0.5);
    const bool select = n_det > P.max_rows;
    if (select) {
      for (int pi = tid; pi < P.n_pairs; pi += NT_) {
        const double hk = rec[(size_t)pi * 10];
        int rank = 0;
        if (hk < kInf)
          for (int j = 0; j < P.n_pairs; ++j) { const double hj = rec[(size_t)j * 10]; rank += (hj < hk || (hj == hk && j < pi)) ? 1 : 0; }
        rowrank[pi] = rank;
      }
      __syncthreads();
    }
    // (in pair order, 256 pairs per trip: ballot + prefix — one thread walking 1 104 records in device memory cost the ALOHA redo
    //  350 k cycles per instance, a third of it)
    if (tid == 0) sRedI[2] = 0;
    __syncthreads();
    for (int base = 0; base < P.n_pairs; base += NT_) {
      const int pi = base + tid;
      const bool on = pi < P.n_pairs && rec[(size_t)pi * 10] < kInf && (!select || rowrank[pi] < P.max_rows);
      const unsigned long long bm = __ballot(on);
      if (lane == 0) sRedI[4 + wave] = __popcll(bm);
      __syncthreads();
      int off = sRedI[2];
      for (int w = 0; w < wave; ++w) off += sRedI[4 + w];
      if (on) rowpair[off + __popcll(bm & ((1ull << lane) - 1ull))] = pi;
      __syncthreads();
      if (tid == 0) sRedI[2] += sRedI[4] + sRedI[5] + sRedI[6] + sRedI[7];
      __syncthreads();
    }
    if (tid == 0) {
      int m = sRedI[2], over = 0;
      for (int r = 0; r < P.n_dense_limit_rows; ++r)       // (the caller's rows are not ranked: one that finds no place is reported)
        if (A.dense_h[(size_t)pb * P.n_dense_limit_rows + r] < kInf) { if (m < P.max_rows) rowpair[m++] = -1 - r; else over = 1; }
      sRedI[0] = m; sRedI[1] = over;
    }
    __syncthreads();
    const int m = sRedI[0];
    if (sRedI[1]) status |= 16;
    const int N = nv + m;
    __syncthreads();
    MKH_WSTAMP(8);
    // ------------------------------------------------------------ tableau K = [[H, Aᵀ],[A, 0]], z, w, states
    WideQpCtx X;
    const bool t_in_lds = P.tableau_in_lds || (long long)N * N <= (long long)P.t_lds_doubles;      // (this instance's tableau)
    double* const T = t_in_lds ? T_lds : T_mem;
    X.T = T; X.in_lds = t_in_lds ? 1 : 0; X.N = N; X.nv = nv;
    X.o_z = P.o_z; X.o_w = P.o_w; X.o_lo = P.o_lo; X.o_hi = P.o_hi; X.o_rown = P.o_rown; X.o_ref = P.o_ref; X.o_col = P.o_col;
    X.o_red = P.o_red; X.o_state = P.o_state; X.o_blk = P.o_blk; X.blk_stride = P.blk_stride;
    auto build_tableau = [&]() {
    wide_accumulate_h(X, Jw, R_all);
    for (int d = tid; d < nv; d += NT_) T[(size_t)d * N + d] += mu_total + sHd[d];
    for (int e = tid; e < m * nv; e += NT_) {                // A: G[s][k] = −nᵀ(jacp₂(to) − jacp₁(from))   (compute_contact_normal_jacobian :59-72)
      const int s = e / nv, k = e - s * nv;
      const int rp = rowpair[s];
      double a;
      if (rp >= 0) {
        const CollisionPairDev& cp = P.pairs[rp];
        const double* o = rec + (size_t)rp * 10;
        const double* kd = sDof + k * 10;
        const V3 d_ang{kd[0], kd[1], kd[2]}, d_lin{kd[3], kd[4], kd[5]}, d_anchor{kd[6], kd[7], kd[8]};
        V3 dj{0, 0, 0};
        if (wide_on_chain(P, cp.body2, k)) dj = dj + d_lin + cross(d_ang, V3{o[7], o[8], o[9]} - d_anchor);
        if (wide_on_chain(P, cp.body1, k)) dj = dj - (d_lin + cross(d_ang, V3{o[4], o[5], o[6]} - d_anchor));
        a = -dot(V3{o[1], o[2], o[3]}, dj);
      } else {
        a = A.dense_G[((size_t)pb * P.n_dense_limit_rows + (-1 - rp)) * nv + k];
      }
      T[(size_t)(nv + s) * N + k] = a; T[(size_t)k * N + nv + s] = a;
    }
    for (int i = tid; i < N; i += NT_) {
      sZ[i] = 0.0;
      if (i < nv) { sW[i] = sC[i]; sState[i] = WS_ZERO; sRown[i] = 1.0; }
      else {
        const int rp = rowpair[i - nv];
        sW[i] = -(rp >= 0 ? rec[(size_t)rp * 10] : A.dense_h[(size_t)pb * P.n_dense_limit_rows + (-1 - rp)]);     // w = A·0 − h
        sState[i] = WS_ROW_OFF;
      }
    }
    __syncthreads();
    for (int s = tid; s < m; s += NT_) {
      double nn = 0.0;
      for (int k = 0; k < nv; ++k) { const double a = T[(size_t)(nv + s) * N + k]; nn += a * a; }
      sRown[nv + s] = nn > 0.0 ? sqrt(nn) : 1.0;
    }
    __syncthreads();
    };
    build_tableau();
    if (tp) {
      if (tp->t_H) for (int e = tid; e < nv * nv; e += NT_) tp->t_H[(size_t)pb * nv * nv + e] = T[(size_t)(e / nv) * N + e % nv];
      if (tp->t_c) for (int k = tid; k < nv; k += NT_) tp->t_c[(size_t)pb * nv + k] = sC[k];
      if (tp->t_coll_h) for (int pi = tid; pi < P.n_pairs; pi += NT_) tp->t_coll_h[(size_t)pb * P.n_pairs + pi] = rec[(size_t)pi * 10];
      if (tp->t_coll_G)                                       // the row of EVERY detected contact (build_ik returns them all)
        for (int e = tid; e < m * nv; e += NT_) {
          const int rp = rowpair[e / nv];
          if (rp >= 0) tp->t_coll_G[((size_t)pb * P.n_pairs + rp) * nv + e % nv] = T[(size_t)(nv + e / nv) * N + e % nv];
        }
      __syncthreads();
    }
    MKH_WSTAMP(9);
    // ------------------------------------------------------------ the QP: dual active set on the sweep tableau (wide_qp)
    int iters = 0, n_outer = 0, n_piv = 0;                  // (qp_iters tap: ratio-test rounds, violated conditions handled, sweeps after phase 0)
    if (!(status & 14) && A.do_qp) {
      // The sweep tableau first; the instances it flags MKH_ST_DEGENERATE (almost dependent active rows) or fails on, and — in a
      // redo launch — the ones a wavefront kernel flagged that way, go through the dense Goldfarb–Idnani iteration with orthogonal
      // factors on the tableau as built (wide_qp_dense)
      bool dense = m > 0 && A.redo_mask != 0 && (A.status_out[pb] & (2 | 8 | 32)) != 0;
      WideQpOut qo{0, 0, 0, 0};
      if (!dense) {
        qo = wide_qp(X, A.clk ? A.clk + (size_t)pb * 24 : nullptr);
        if (m > 0 && (qo.status & (2 | 8 | 32))) { dense = true; build_tableau(); }
      }
      if (dense) qo = wide_qp_dense(X, P.gi_in_lds ? smem + P.o_gi : wsb + P.ws_gi, A.clk ? A.clk + (size_t)pb * 24 : nullptr);
      status |= qo.status & ~32; iters = qo.iters; n_outer = qo.n_outer; n_piv = qo.n_piv;
    }
    MKH_WSTAMP(11);
    if (A.clk && tid == 0) { A.clk[(size_t)pb * 24 + 13] = iters; A.clk[(size_t)pb * 24 + 14] = n_piv; A.clk[(size_t)pb * 24 + 16] = N; }
    if (MKH_WTAP(t_qp_iters) && tid == 0) MKH_WTAP(t_qp_iters)[pb] = (iters & 1023) | ((n_outer & 1023) << 10) | ((n_piv & 1023) << 20);
    // the contacts that found no row: G·Δq ≤ h at the solution?  (the wavefront kernels' collision_phase, mode 1)
    if (select && A.do_qp && !(status & 14)) {
      __syncthreads();
      bool viol = false;
      for (int pi = tid; pi < P.n_pairs; pi += NT_) {
        const double* o = rec + (size_t)pi * 10;
        if (!(o[0] < kInf) || rowrank[pi] < P.max_rows) continue;
        const CollisionPairDev& cp = P.pairs[pi];
        V3 vel{0, 0, 0};
        for (int k = 0; k < nv; ++k) {
          const bool c2 = wide_on_chain(P, cp.body2, k), c1 = wide_on_chain(P, cp.body1, k);
          if (!c1 && !c2) continue;
          const double* kd = sDof + k * 10;
          const V3 d_ang{kd[0], kd[1], kd[2]}, d_lin{kd[3], kd[4], kd[5]}, d_anchor{kd[6], kd[7], kd[8]};
          const double dq = sZ[k];
          if (c2) vel = vel + dq * (d_lin + cross(d_ang, V3{o[7], o[8], o[9]} - d_anchor));
          if (c1) vel = vel - dq * (d_lin + cross(d_ang, V3{o[4], o[5], o[6]} - d_anchor));
        }
        viol = viol || (-dot(V3{o[1], o[2], o[3]}, vel) > o[0] + 1e-9 * (1.0 + fabs(o[0])));
      }
      if (__syncthreads_or(viol ? 1 : 0)) status |= 16;
    }
    status_all |= status;
    // ------------------------------------------------------------ v = Δq / dt (solve_ik.py:104)
    const bool last = until || (step + 1 == n_steps) || (status & 14);     // (until: v of every step — the loop may end at the next check)
    if (last) {
      if (A.v_out) {
        const bool ok = !(status & 14);
        for (int d = tid; d < nv; d += NT_) A.v_out[(size_t)pb * nv + d] = ok ? sZ[d] / A.dt : __builtin_nan("");
      }
      if (status & 14) break;
    }
    if (integrate) {
      // q ← q ⊕ Δq (mj_integratePos; Configuration.integrate_inplace, mink/configuration.py:228-236)
      __syncthreads();
      for (int j = tid; j < P.njnt; j += NT_) {
        const int jt = P.jnt_type[j];
        int qa = P.jnt_qadr[j], va = P.jnt_dadr[j];
        if (jt == JNT_HINGE || jt == JNT_SLIDE) { sq[qa] += sZ[va]; continue; }
        if (jt == JNT_FREE) {
          for (int i = 0; i < 3; ++i) sq[qa + i] += sZ[va + i];
          qa += 3; va += 3;
        }
        const V3 w{sZ[va], sZ[va + 1], sZ[va + 2]};
        const double n = sqrt(dot(w, w));
        const V3 ax = (n < 1e-15) ? V3{1.0, 0.0, 0.0} : (1.0 / n) * w;
        const Q4 qr = (n == 0.0) ? Q4{1, 0, 0, 0} : axis_angle(ax, n);
        const Q4 r = qmul(qnormalize(Q4{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]}), qr);
        sq[qa] = r.w; sq[qa + 1] = r.x; sq[qa + 2] = r.y; sq[qa + 3] = r.z;
      }
      __syncthreads();
    }
    }  // step loop
    __syncthreads();
    if (A.q_out) for (int i = tid; i < nq; i += NT_) A.q_out[(size_t)pb * nq + i] = sq[i];
    if (until && tid == 0) {
      if (A.iters_out) A.iters_out[pb] = it_done;
      if (A.converged_out) A.converged_out[pb] = conv_flag;
    }
    if (A.status_out && tid == 0) A.status_out[pb] = status_all;
    MKH_WSTAMP(12);
  }
  }
}

// the two builds: pair lists of analytic pairs only (every model workload without general convex pairs, every redo launch behind an
// analytic collision build) / with pairs that need the general convex routine
__global__ __launch_bounds__(kWideThreads, 2) void ik_wide_kernel(const WideProblem* __restrict__ Pg, SolveArgs A, const TapArgs* __restrict__ tp) {
  wide_kernel_body<false>(Pg, A, tp);
}
__global__ __launch_bounds__(kWideThreads, 2) void ik_wide_kernel_cvx(const WideProblem* __restrict__ Pg, SolveArgs A, const TapArgs* __restrict__ tp) {
  wide_kernel_body<true>(Pg, A, tp);
}

}  // namespace mkh
