// One-wavefront-per-problem differential-IK kernel for gfx950 (MI355X, wave64).
//
// One launch = one batched mink.solve_ik (mink/solve_ik.py:68-105).  A workgroup is
// exactly one wavefront; it loops over the problems of its XCD's contiguous slice of the batch
// (workgroups are dealt round-robin to the 8 XCDs, each with its own L2: keeping neighbouring rows of
// q / targets / v in ONE L2 lets partial cache lines at row boundaries merge before they reach HBM).
// (The host picks the launch shape — minkhip.hip::launch: persistent wavefronts with a static share each and a ticket tail,
//  or one workgroup per problem, for which the loop below runs once and the XCD ranges are the same.)
// Lanes change role by phase:
//   body lane   (l < nbody)   forward kinematics by pointer jumping over the tree
//                             (replaces mj_kinematics, mink/configuration.py:63)
//   task lane   (l < n_frame) frame pose, e = log(T_bt), jlog blocks
//                             (mink/tasks/frame_task.py:95-146, mink/lie/se3.py)
//   dof lane    (l < nv)      world Jacobian column of dof l (replaces mj_jacSite /
//                             mj_jac / mj_jacSubtreeCom), task Jacobian column, box limits,
//                             column l of H = Σ JᵀW²J + (λ+Σμ)I (mink/tasks/task.py:125-138)
//   pair lane   (l < n_pairs) capsule/sphere/plane signed distance + half-space row
//                             (mink/limits/collision_avoidance_limit.py:187-229)
//   tableau lane(l < ntab)    one column of the symmetric sweep tableau of
//                             K = [[H, Aᵀ],[A, 0]] held in pinned VGPRs (tab_asm.inc); the dual
//                             active-set QP (replaces qpsolvers→quadprog, mink/solve_ik.py:101) is a
//                             sequence of rank-1 sweeps: the pivot column goes through LDS once
//                             (indexed row read + one ds_write), comes back as 16-lane planes and is
//                             broadcast by the DPP operand network (v_fmac_f64_dpp row_newbcast).
//                             Low-rank variants (F_WOOD) never sweep a dof: H = Dg + JwᵀJw, the task
//                             residuals are eliminated outside the tableau — one column of [S | Jh]
//                             per lane — and the dof block −H⁻¹ is built by rank-1 updates that do not
//                             depend on each other (wood_start); a cold G1-size solve first
//                             re-eliminates for the bounds its unconstrained minimiser violates, so that the
//                             tableau starts with those dofs on their bounds (wood_eliminate).
// Per-problem J rows, task blocks, poses and half-space rows are staged in LDS.
// Kernel builds with one more resident wave per SIMD (_w3: 168 / 128 registers) run the phases that need many
// registers — kinematics, Lie algebra, Jacobian rows, the elimination — as real function calls (pre_phases,
// wood_start, direct_pairs), which are not bound by the kernel's register cap.
#pragma once
#include "collide_dev.h"
#include "lie_dev.h"
#include "mkh_types.h"
#include "wave_ops.h"
#include "tab_asm.inc"
#include <utility>

// Register map of the tableau: Tab<NT> (2 resident waves per SIMD for NT > 24) or, in the variant TUs that define
// MKH_W3, TabW3<NT> — the same primitives pinned below 168 VGPRs, i.e. 3 resident waves per SIMD.
#if defined(MKH_W4)     // (a W4 translation unit also defines MKH_W3: the same call structure and compact LDS layout)
#define MKH_TAB TabW4
#elif defined(MKH_W3)
#define MKH_TAB TabW3
#else
#define MKH_TAB Tab
#endif

// Phases as REAL CALLS (MKH_CALLS): the phases that run while the tableau is dead — kinematics and task lanes (pre_phases), the
// (task, dof) pair lanes (direct_pairs), the collision rows (collision_phase), the low-rank elimination (wood_start) — are
// noinline functions.  The kernel's own register cap (amdgpu_num_vgpr: everything below the pinned tableau) does not bind a
// callee, which may use the whole file of the occupancy; the kernel keeps only what is live across the call.  Round 2 built
// this for the one-more-wave register maps (MKH_W3); round 3 turns it on for every build whose phases do not fit next to
// the kernel's own values either: the general collision builds (FEAT 8, FEAT 136 with the convex routine) and the all-feature
// one (FEAT 30).  Measured and kept inlined: the plane / sphere / capsule builds (FEAT 72 / 88: the Shadow hand, whose phases
// fit — as calls 0.402 → 0.472 ms, prologues and callee-saved registers); FEAT 31, every feature + taps, is the parity build
// and its cycle stamps sit inside the phases.
// Round 4, the builds of the general convex routine (FEAT 136): GJK keeps its simplex in LDS and works in the frame of shape 1
// (143 VGPRs instead of 196), the overlapping case left the per-lane path (the expanding polytope is a wave-level routine, a
// real and rarely taken call of its own: overlap_pair).  Measured on `ur5e_convex` (4 096 instances, one cylinder–box pair):
// the phase INLINED on the 32-row map 0.178 ms / 27 MB of HBM traffic; as a call on the 32-row map 0.186 ms / 88 MB (its
// prologue still saves ≈50 callee-saved VGPRs per problem); as a call on the 16-row three-waves map 0.156 ms — kept
// (minkhip.hip launch(): calls_nt_min).  -DMKH_INLINE_136 builds the inlined variant for A/B runs.
#if defined(MKH_W3) || (defined(MKH_FEAT) && (MKH_FEAT & 8) && !(MKH_FEAT & 1) && !(MKH_FEAT & 64) && (!(MKH_FEAT & 128) || !defined(MKH_INLINE_136)))
#define MKH_CALLS 1
#endif

namespace mkh {

constexpr int kNumXcd = 8;   // MI355X: 8 XCDs × 32 CUs, one L2 each

// ------------------------------------------------------------------ LDS layout
// Row stride of the staged half-space rows A[s][:]: the smallest column-load size ≥ nv (tab_asm.inc load_lo_*),
// not the wave width — 40 rows × 64 doubles were 20 KB of Shadow's 37 KB and cost it a third of its occupancy.
__host__ __device__ inline int a_stride_for(int nv) {
  return nv <= 16 ? 16 : (nv <= 24 ? 24 : (nv <= 32 ? 32 : (nv <= 44 ? 44 : (nv <= 48 ? 48 : 64))));
}

// Row stride of the staged Jacobian rows of the direct start: only the dof rows are read (rank1_leading_rows)
__host__ __device__ inline int j_stride_direct(int nv, int nt) { const int a = a_stride_for(nv); return a < nt ? a : nt; }

constexpr int kPivBuf = kWave + 8;   // doubles per pivot broadcast buffer
constexpr int kMu = 18, kMuBig = 24; // low-rank start: compiled capacities of task-residual rows of one problem (WoodElim in tab_asm.inc)
constexpr int kWoodRow = 64;         // low-rank start: doubles of the published-row buffer — (S[r][c], w_c) pairs — in the pivot buffers, followed by 1/d_r
struct LdsLayout {
  int q, X, jnt, tgt, task, J, dof, com, col, A, piv, S, q2, tgt2, hsel, total;  // offsets in doubles
};
__host__ __device__ inline int lds_even(int x) { return (x + 1) & ~1; }
__host__ __device__ inline LdsLayout lds_layout(int nq, int nv, int nbody, int njnt, int n_frame,
                                                int n_posture, int n_com, int max_rows, int j_rows, int j_stride,
                                                int s_doubles = 0, bool prefetch = false, bool compact = false,
                                                bool wood = false, int n_hsel = 0, bool piv_small = false, bool wood_rows = false) {
  LdsLayout L;
  int o = 0;
  const int x_sz = 7 * lds_even(nbody), jnt_sz = lds_even(njnt * 6);
  const int j_sz = lds_even(j_rows * j_stride);
  L.q = o;    o += lds_even(nq);
  if (wood_rows) {
    // low-rank start WITH half-space rows (round 6; F_WOOD | F_COLL builds): the rows [r][NT] of Jh — the dof part AND, behind it,
    // the half-space rows' columns of the elimination — overwrite the joint axes and the task blocks as in the compact layout (one
    // pass of (task, dof) pair lanes: the host checks n_jpairs <= 64), but the body poses, the dof axes and h of every pair stay:
    // collision_phase reads them again after the QP (dropped contacts at the solution).  One pivot buffer (no look-ahead publishing).
    L.tgt = o;  o += lds_even(n_frame * 7 + n_com * 3);
    L.X = o;    o += x_sz;
    const int u0 = o;
    L.jnt = o;  o += jnt_sz;
    L.task = o; o += n_frame * 64;
    if (o - u0 < j_sz) o = u0 + j_sz;
    L.J = u0;
    L.dof = o;  o += lds_even(nv * 10);
    L.com = o;  o += (n_com > 0 ? nbody * 4 : 0);
    L.col = o;  o += max_rows * 16;
    L.A = o;    o += max_rows * a_stride_for(nv);
    L.piv = o;  o += kWoodRow + 2 * kMuBig;
    L.S = o;    o += lds_even(s_doubles);
    L.q2 = prefetch ? o : L.q;     o += prefetch ? lds_even(nq) : 0;
    L.tgt2 = prefetch ? o : L.tgt; o += prefetch ? lds_even(n_frame * 7 + n_com * 3) : 0;
    L.hsel = o; o += lds_even(n_hsel);
    L.total = o;
    return L;
  }
  // compact (3-waves-per-SIMD variants, no collision rows): ranges that are not alive at the same time share storage.
  //   direct start:   {body poses, joint axes}              | {staged Jacobian rows of one task, pivot buffers}
  //   low-rank start: {body poses, joint axes, task blocks} | {all Jacobian rows} — the pair lanes hold their entries in
  //                   registers until every lane has read its task block (wood_start); the pivot buffers stay apart
  //                   (they carry the published rows and 1/d_r of the elimination)
  const int task_sz = n_frame * 64;
  if (compact && wood) { L.tgt = o; o += lds_even(n_frame * 7 + n_com * 3); }
  const int u0 = o;
  L.X = o;    o += x_sz;                             // body poses, component-major: X[c][body] (c = x y z qw qx qy qz)
  L.jnt = o;  o += jnt_sz;
  if (compact && wood) {
    L.task = o; o += task_sz;
    if (o - u0 < j_sz) o = u0 + j_sz;
  } else {
    if (compact) { const int need = j_sz + 2 * kPivBuf; o = u0 + (x_sz + jnt_sz > need ? x_sz + jnt_sz : need); }
    L.tgt = o;  o += lds_even(n_frame * 7 + n_com * 3);
    L.task = o; o += task_sz;
  }
  // weighted Jacobian rows [r][j_stride]: the 6 rows of ONE task at a time (direct start, stride NT), or
  // every task row + one vector (low-rank start, stride NR)
  L.J = compact ? u0 : o;    o += compact ? 0 : j_sz;
  L.dof = o;  o += lds_even(nv * 10);
  L.com = o;  o += (n_com > 0 ? nbody * 4 : 0);
  L.col = o;  o += max_rows * 16;
  L.A = o;    o += max_rows * a_stride_for(nv);       // half-space rows A[s][0..stride)
  const bool piv_apart = !compact || wood;
  // two pivot column broadcast buffers (64 entries + 8 scalar slots of the pivot lane): look-ahead publishing.  (piv_small, the
  // F_COM builds with one more resident wave: a low-rank start never publishes ahead — one buffer for the QP, and the
  // elimination's published row + 1/d_r + ω_r = kWoodRow + 2·kMuBig doubles: 32 doubles less, the difference between 9 and 10
  // wavefronts per CU for the G1 full example — LDS is handed out in 1 280-byte granules on gfx950)
  L.piv = piv_apart ? o : u0 + j_sz;  o += piv_apart ? (piv_small ? kWoodRow + 2 * kMuBig : 2 * kPivBuf) : 0;
  L.S = o;    o += lds_even(s_doubles);   // low-rank start: columns of −Jh·Jhᵀ + right-hand sides
  // second buffers of the per-problem inputs: the next problem's q / targets are fetched straight into LDS
  // (global_load_lds) while the current problem is being solved
  // (only when the host found that they do not cost a resident wave: DeviceProblem::prefetch)
  L.q2 = prefetch ? o : L.q;     o += prefetch ? lds_even(nq) : 0;
  L.tgt2 = prefetch ? o : L.tgt; o += prefetch ? lds_even(n_frame * 7 + n_com * 3) : 0;
  // more collision pairs than tableau rows: h of EVERY pair, for the selection of the tightest rows and the check of the
  // dropped ones at the solution (collision_phase) — the ALOHA pair set of the reference is 1 104 pairs for 48 rows
  L.hsel = o; o += lds_even(n_hsel);
  L.total = o;
  return L;
}

// Low-rank start: the n_mu × (sp + 1) block of −Jh·Jhᵀ and right-hand sides fits in the dof stash?
// (n_com_bodies: the subtree-CoM array that follows the dof stash — nbody when the problem has ComTasks, else 0)
__host__ __device__ inline bool wood_s_aliases_dof(int nv, int n_mu, int sp, int n_com_bodies = 0) {
  return n_mu * (sp + 1) <= lds_even(nv * 10) + n_com_bodies * 4;
}

// Compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, N).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS byte address of a pointer into the dynamic shared segment (low half of the flat address).
__device__ __forceinline__ unsigned lds_addr(const double* p) { return (unsigned)(size_t)p; }

// Per-lane QP bookkeeping.  Lane j owns column j of the symmetric tableau, stored unscaled in
// R[i] with a lazy SYMMETRIC scale:  true T[i][j] = σ_i·σ_j·R[i][j]  (i ≠ j),  true T[j][j] = D.
// A sweep multiplies row k and column k by ±1/d — that is one scalar update of σ_k, so no register
// whose index depends on the runtime pivot is ever written (hipcc would send it to scratch, and a
// scalar switch over statically named registers costs ~500 cycles of branches per access).
struct QpLane {
  double x;        // the FREE value of the index: z (dof value / multiplier) when basic, w (gradient / slack) when not.
                   // The other one is implied: w = 0 when basic; z = bound (dof) or 0 (row) when not.
  double lo, hi, sg, D;
  // Roles as sign masks / small flags so that the per-iteration arithmetic is XORs instead of compare+select
  // pairs (the active-set loop is VALU-issue bound):
  int usign;       // kSign when basic: the step update is x += (α ^ usign)·τ   (basic: z −= α·τ, else w += α·τ)
  int ysign;       // kSign when the index is a dof sitting at its UPPER bound (multiplier that must stay ≥ 0 is −w)
  int rsign;       // kSign when the index is a dof sitting at its LOWER bound (its rate of decrease is −r)
  int sel;         // selectable by the primal test: 1 basic dof (bounds), 2 inactive half-space row (slack), 0 no
  int elig;        // takes part in the ratio test (dof at a bound / active row)
};
constexpr int kSign = (int)0x80000000;
__device__ __forceinline__ double xor_sign(double v, int mask) {
  return __hiloint2double(__double2hiint(v) ^ mask, __double2loint(v));
}

// Lane `col` publishes its raw tableau column R[0..NT) in LDS (the only cross-lane transport of
// the QP): every lane then reads its own entry (ratio test / multiplier) and streams the whole
// vector back with broadcast reads for the rank-1 update.  Entry `col` is published as 0 so that
// row `col` of every column is left alone by the update (its change is carried by σ_col).
// D, σ, w, z of the pivot lane — and, for the active-set phase, its bounds, row norm and basic flag —
// broadcast through LDS in the same round trip as the column (a v_readlane chain costs ~45-90 cycles
// per value and sits in the pivot's dependent chain).
struct PivotScalars { double d, sg, x, rn, lo, hi; };

template <int NT, bool FULL = false>
__device__ __forceinline__ double publish_column(typename MKH_TAB<NT>::Regs& ts, const QpLane& s, int col, int lane, double* sPiv,
                                                 PivotScalars& ps, int nact = kWave, double rown = 1.0) {
  // Column `col` equals row `col` (R is symmetric): lane i holds R[col][i] in tableau register
  // `col`.  One indexed register read (VGPR index mode on the pinned base) + ONE ds_write_b64 for
  // the whole wave.  Having lane `col` dump its 48 registers itself costs 48 single-lane LDS
  // writes = 750-1850 cycles (measured, tools/ubench) — half of the whole pivot.
  const double rowv = MKH_TAB<NT>::get_dyn(ts, col);
  // lanes ≥ nact hold dropped indices (task residuals of the low-rank start): they publish 0, so
  // their columns stop changing and the rows they own leave every other column alone
  const double own = (lane == col || lane >= nact) ? 0.0 : rowv;
  wave_sync();                                   // earlier readers of sPiv are done
  sPiv[lane] = own;
  if (lane == col) {
    double2* o = reinterpret_cast<double2*>(sPiv + kWave);
    o[0] = double2{s.D, s.sg};
    o[1] = double2{s.x, rown};
    if (FULL) o[2] = double2{s.lo, s.hi};
  }
  wave_sync();
  const double2* o = reinterpret_cast<const double2*>(sPiv + kWave);
  const double2 a = o[0], b = o[1];
  ps.d = a.x; ps.sg = a.y; ps.x = b.x; ps.rn = b.y;
  if (FULL) {
    const double2 c = o[2];
    ps.lo = c.x; ps.hi = c.y;
  }
  return own;                                    // raw R[lane][col] (0 for lane col)
}

// The two halves of publish_column for look-ahead publishing: while the rank-1 update of pivot j streams through
// the FMA pipe, the column of pivot j+1 (whose entries after update j are one FMA away) is already on its way
// through LDS, so the write → read round trip leaves the pivot's dependent chain.
template <bool FULL>
__device__ __forceinline__ void stage_column(const QpLane& s, int col, int lane, double* buf, double own, double rown = 1.0) {
  buf[lane] = own;
  if (lane == col) {
    double2* o = reinterpret_cast<double2*>(buf + kWave);
    o[0] = double2{s.D, s.sg};
    o[1] = double2{s.x, rown};
    if (FULL) o[2] = double2{s.lo, s.hi};
  }
}
template <bool FULL>
__device__ __forceinline__ void read_pivot_scalars(const double* buf, PivotScalars& ps) {
  const double2* o = reinterpret_cast<const double2*>(buf + kWave);
  const double2 a = o[0], b = o[1];
  ps.d = a.x; ps.sg = a.y; ps.x = b.x; ps.rn = b.y;
  if (FULL) {
    const double2 c = o[2];
    ps.lo = c.x; ps.hi = c.y;
  }
}

// Symmetric sweep (reverse = un-sweep) on index k (wave-uniform); sPiv holds column k.
template <int NT, int ROWS>
__device__ __forceinline__ void rank1_rows(typename MKH_TAB<NT>::Regs& ts, unsigned addr, double g) {
  if constexpr (ROWS >= NT) MKH_TAB<NT>::rank1_body(ts, addr, g);
  else if constexpr (ROWS == 16) MKH_TAB<NT>::rank1_body_16(ts, addr, g);
  else if constexpr (ROWS == 24) MKH_TAB<NT>::rank1_body_24(ts, addr, g);
  else if constexpr (ROWS == 32) MKH_TAB<NT>::rank1_body_32(ts, addr, g);
  else if constexpr (ROWS == 44) MKH_TAB<NT>::rank1_body_44(ts, addr, g);
  else if constexpr (ROWS == 48) MKH_TAB<NT>::rank1_body_48(ts, addr, g);
}
// T[i] += lds[i]·g for i < n (n from a_stride_for ≥ nv): the H accumulation only touches the dof rows, the rows
// of the half-space block stay zero — 24 instead of 64 FMAs per staged Jacobian row for the Shadow hand
template <int NT>
__device__ __forceinline__ void rank1_leading_rows(typename MKH_TAB<NT>::Regs& ts, unsigned addr, double g, int n) {
  MKH_TAB<NT>::rank1_prefetch(ts, addr);
  if (n >= NT) { MKH_TAB<NT>::rank1_body(ts, addr, g); return; }
  if constexpr (NT > 16) { if (n == 16) MKH_TAB<NT>::rank1_body_16(ts, addr, g); }
  if constexpr (NT > 24) { if (n == 24) MKH_TAB<NT>::rank1_body_24(ts, addr, g); }
  if constexpr (NT > 32) { if (n == 32) MKH_TAB<NT>::rank1_body_32(ts, addr, g); }
  if constexpr (NT > 44) { if (n == 44) MKH_TAB<NT>::rank1_body_44(ts, addr, g); }
  if constexpr (NT > 48) { if (n == 48) MKH_TAB<NT>::rank1_body_48(ts, addr, g); }
}

// streamed update of the rows [0, hb) (rounded up to a generated prefix); see gen_tab_asm.py rank1_stream_*
template <int NT>
__device__ __forceinline__ void rank1_stream_rows(typename MKH_TAB<NT>::Regs& ts, unsigned next, double g, int hb) {
  if constexpr (NT > 16) { if (hb <= 16) { MKH_TAB<NT>::rank1_stream_16(ts, next, g); return; } }
  if constexpr (NT > 24) { if (hb <= 24) { MKH_TAB<NT>::rank1_stream_24(ts, next, g); return; } }
  if constexpr (NT > 32) { if (hb <= 32) { MKH_TAB<NT>::rank1_stream_32(ts, next, g); return; } }
  if constexpr (NT == 8) MKH_TAB<NT>::rank1_stream_8(ts, next, g);
  else if constexpr (NT == 16) MKH_TAB<NT>::rank1_stream_16(ts, next, g);
  else if constexpr (NT == 24) MKH_TAB<NT>::rank1_stream_24(ts, next, g);
  else if constexpr (NT == 32) MKH_TAB<NT>::rank1_stream_32(ts, next, g);
  else if constexpr (NT == 44) MKH_TAB<NT>::rank1_stream_44(ts, next, g);
  else if constexpr (NT == 48) MKH_TAB<NT>::rank1_stream_48(ts, next, g);
  else MKH_TAB<NT>::rank1_stream_64(ts, next, g);
}

// T[i] = lds[i] for i < n (n from a_stride_for: wave-uniform, one of the generated sizes)
template <int NT>
__device__ __forceinline__ void load_leading_rows(typename MKH_TAB<NT>::Regs& ts, unsigned addr, int n) {
  if (n >= NT) { MKH_TAB<NT>::load_all(ts, addr); return; }
  if constexpr (NT > 16) { if (n == 16) MKH_TAB<NT>::load_lo_16(ts, addr); }
  if constexpr (NT > 24) { if (n == 24) MKH_TAB<NT>::load_lo_24(ts, addr); }
  if constexpr (NT > 32) { if (n == 32) MKH_TAB<NT>::load_lo_32(ts, addr); }
  if constexpr (NT > 44) { if (n == 44) MKH_TAB<NT>::load_lo_44(ts, addr); }
  if constexpr (NT > 48) { if (n == 48) MKH_TAB<NT>::load_lo_48(ts, addr); }
}

template <int NT, int ROWS = NT>
__device__ __forceinline__ void pivot(typename MKH_TAB<NT>::Regs& ts, QpLane& s, int k, bool reverse, int lane, const double* sPiv,
                                      double own, const PivotScalars& ps, double inv) {
  const double sk = ps.sg;
  const double ck = s.sg * sk * own;             // true T[lane][k]
  const double g = (sk * sk) * own * inv;        // R-units multiplier of this lane's column
  rank1_rows<NT, ROWS>(ts, lds_addr(sPiv), -g);      // R[i][lane] −= R[i][k]·g   (row k: published 0)
  if (lane == k) {
    s.D = -inv;                                  // T[k][k] = −1/d
    s.sg = (reverse ? -sk : sk) * inv;           // row/column k scaled by ±1/d
  } else {
    s.D = fma(-ck * inv, ck, s.D);               // T[j][j] −= T[j][k]²/d
  }
}

// ----------------------------------------------------------------- the kernel
// Compile-time feature set of a kernel variant.  The hot production variant (FEAT = 0) carries no
// tap code, no RelativeFrameTask / CoM / collision branches and no fused step loop: fewer live values
// for the compiler's 128-VGPR budget and a smaller instruction footprint.
// F_WOOD: low-rank ("Woodbury") start of the QP — H = Dg + JwᵀJw is never formed and no dof index is ever swept on the
// tableau: with S = I + Jh·Jhᵀ = L·D·Lᵀ and Z = L⁻¹Jh the dof block after phase 0 is −σσᵀ∘(I − ZᵀD⁻¹Z) (wood_start below;
// tools/proto_woodbury.py states the algebra in numpy).  F_WOOD | F_COM: + ComTask rows, up to 24 task rows.
// taps that exist in the low-rank variants (profiling only: H is never formed there)
constexpr bool kTapIsProf_t_xpos = false, kTapIsProf_t_xquat = false, kTapIsProf_t_frame_pose = false,
               kTapIsProf_t_subtree_com = false, kTapIsProf_t_task_e = false, kTapIsProf_t_task_J = false,
               kTapIsProf_t_H = false, kTapIsProf_t_c = false, kTapIsProf_t_box_lo = false, kTapIsProf_t_box_hi = false,
               kTapIsProf_t_coll_G = false, kTapIsProf_t_coll_h = false, kTapIsProf_t_qp_iters = true,
               kTapIsProf_t_cycles = true;
enum : int { F_TAPS = 1, F_REL = 2, F_COM = 4, F_COLL = 8, F_STEPS = 16, F_ALL = 31, F_WOOD = 32, F_SIMPLE_COLL = 64, F_CONVEX_COLL = 128,
              F_DENSE = 256 };   // F_DENSE alone: the lean build of the plugin route (dense task / limit rows next to frame + posture tasks and box limits)

// P lives in device memory (not in the kernarg segment): hipcc materialises every by-value kernel
// argument field in SGPRs at kernel entry and keeps it there, which starved the QP loop of SGPRs
// (580 SGPR spills, v_readlane results serialised through one SGPR pair).
// One kernel per translation unit: the variant TU defines MKH_NT (tableau rows per lane) and
// MKH_FEAT before including this header.  Not a template because `amdgpu_num_vgpr` — the cap that
// keeps the compiler out of the pinned tableau registers — only accepts an integer literal.
// NOTE the value: on gfx90a+ LLVM doubles the attribute (it budgets the unified VGPR+AGPR file:
// GCNSubtarget::getBaseMaxNumVGPRs) and silently DROPS it when the doubled value exceeds what the
// waves-per-EU bound allows (256 at 2 waves/SIMD).  So a hard limit of C architectural VGPRs is
// requested as C/2; tools/check_vgpr_cap.py verifies on the ISA that no compiler-generated
// instruction touches a register at or above the cap.
#ifdef MKH_NT
#ifdef MKH_DEBUG_ALL_TAPS
#define MKH_TAP(f) ((kTaps && tp) ? tp->f : nullptr)
#else
#define MKH_TAP(f) ((kTaps && tp && (!kWood || kTapIsProf_##f)) ? tp->f : nullptr)
#endif
#define MKH_CAT2(a, b, c) a##b##_##c
#define MKH_CAT(a, b, c) MKH_CAT2(a, b, c)
#ifndef MKH_KERNEL_NAME   // low-rank variants (MKH_NR defined) are named by their translation unit
#define MKH_KERNEL_NAME MKH_CAT(ik_solve_kernel_, MKH_NT, MKH_FEAT)
#endif

// ISA markers for static instruction counting (tools/isa_census.py): comments only, and only with -DMKH_MARKERS
#ifdef MKH_MARKERS
#define MKH_MARK(name) asm volatile("; MKH_MARK " name)
#else
#define MKH_MARK(name) do {} while (0)
#endif

// ---------------------------------------------------------------- kinematics and task lanes (per problem and step)
// Forward kinematics, joint axes / dof lanes, subtree CoM and the frame-task lanes: everything up to the first use of the
// tableau, handed over through LDS plus the few per-lane values of PreOut.  In the 3-waves-per-SIMD variants (MKH_W3) this is
// a real function call: the kernel itself is capped below the pinned tableau range (72 VGPRs for 44 rows) all the way
// through, which these phases do not fit in (83–236 spilled VGPRs when inlined); as a callee they get the whole 168-register
// file — the tableau is dead while they run — and the kernel keeps only what is live across the call.
struct PreOut { int status, d_kind, d_k, d_body, d_qadr, conv; double mu_lane; V3 com_root; };
#ifdef MKH_CALLS
#ifdef MKH_ONE_SHOT
#define MKH_PRE_ATTR __attribute__((noinline, internal_linkage))   // (lets the compiler prove that the callee reads no work-item id: wave_ops.h lane_id)
#else
#define MKH_PRE_ATTR __attribute__((noinline))
#endif
#ifdef MKH_CLOCKS     // experiment builds: the callee stamps slots 20.. of the problem's (B, 24) clock row itself
#define MKH_PRE_TC_PARAMS , long long* clk_row
#define MKH_PRE_TC_ARGS , (A.clk ? A.clk + (size_t)pb * 24 + 20 : nullptr)
#define MKH_PRE_TICK() do { if (clk_row) { *clk_row = __builtin_readcyclecounter(); ++clk_row; } asm volatile("" : "+v"(lane)); } while (0)
// phase-stop (tools/phase_census.sh): slot 15 of the problem's clock row holds the boundary after which the solve is abandoned,
// so that the PMC counters of launches with stop = k and k + 1 differ by phase k + 1's instructions.  0 = run to the end.
#define MKH_PRE_STOP(id) do { if (clk_row_stop == (id)) return PreOut{}; } while (0)
#else
#define MKH_PRE_TC_PARAMS
#define MKH_PRE_TC_ARGS
#define MKH_PRE_TICK() do { asm volatile("" : "+v"(lane)); } while (0)
#define MKH_PRE_STOP(id) do {} while (0)
#endif
#else
#define MKH_PRE_ATTR __forceinline__
#define MKH_PRE_TC_PARAMS , long long (&tc)[8], int& tci
#define MKH_PRE_TC_ARGS , tc, tci
#ifdef MKH_CLOCKS
#define MKH_PRE_TICK() do { tc[tci] = __builtin_readcyclecounter(); ++tci; asm volatile("" : "+v"(lane)); } while (0)
#else
#define MKH_PRE_TICK() do { if (MKH_TAP(t_cycles)) tc[tci] = __builtin_readcyclecounter(); ++tci; asm volatile("" : "+v"(lane)); } while (0)
#endif
#define MKH_PRE_STOP(id) do {} while (0)
#endif
// The descriptor fields pre_phases reads, loaded through the constant address space (scalar loads) with the table pointers
// typed as global memory: inside a real function the compiler has to treat pointers that arrive as arguments or come out
// of memory as flat, i.e. per-lane flat_load instructions even for wave-uniform fields (83 of them in the first build).
#define MKH_GLOBAL __attribute__((address_space(1)))
#define MKH_CONSTANT __attribute__((address_space(4)))
struct PreView {
  const MKH_GLOBAL double *body_f, *jnt_f, *dof_f;
  const MKH_GLOBAL int32_t *body_i, *jnt_i, *dof_i;
  const MKH_GLOBAL FrameTaskDev* frame;
  int nbody, nv, nq, njnt, n_frame, n_posture, n_com, max_rows, n_jrows, prefetch, prefetch_w3, prefetch_w3w, wood_compact, prefetch_wc, n_rows_tap, nrounds, robot_root, n_hsel;
  __device__ __forceinline__ explicit PreView(const DeviceProblem* Pq) {
    const MKH_CONSTANT DeviceProblem* c = (const MKH_CONSTANT DeviceProblem*)Pq;
    body_f = (const MKH_GLOBAL double*)c->body_f; jnt_f = (const MKH_GLOBAL double*)c->jnt_f; dof_f = (const MKH_GLOBAL double*)c->dof_f;
    body_i = (const MKH_GLOBAL int32_t*)c->body_i; jnt_i = (const MKH_GLOBAL int32_t*)c->jnt_i; dof_i = (const MKH_GLOBAL int32_t*)c->dof_i;
    frame = (const MKH_GLOBAL FrameTaskDev*)c->frame;
    nbody = c->nbody; nv = c->nv; nq = c->nq; njnt = c->njnt; n_frame = c->n_frame; n_posture = c->n_posture; n_com = c->n_com;
    max_rows = c->max_rows; n_jrows = c->n_jrows; prefetch = c->prefetch; prefetch_w3 = c->prefetch_w3; prefetch_w3w = c->prefetch_w3w;
    wood_compact = c->wood_compact; prefetch_wc = c->prefetch_wc;
    n_rows_tap = c->n_rows_tap; nrounds = c->nrounds; robot_root = c->robot_root; n_hsel = c->n_hsel;
  }
};
// second LDS buffers for the next problem's inputs: decided on the host per LDS layout (only where they cost no resident wave)
template <class PT>
__device__ __forceinline__ bool kernel_prefetch(const PT& P0) {
  constexpr bool kWood = (MKH_FEAT & F_WOOD) != 0;
  constexpr bool kWoodRows = kWood && (MKH_FEAT & (F_COLL | F_DENSE)) != 0;    // low-rank start with half-space rows: its own layout
#ifdef MKH_W3
  return (kWood ? P0.prefetch_w3w : P0.prefetch_w3) != 0;
#else
  return ((kWoodRows || (kWood && P0.wood_compact != 0)) ? P0.prefetch_wc : P0.prefetch) != 0;
#endif
}
template <class PT>
__device__ __forceinline__ LdsLayout kernel_lds_layout(const PT& P0) {
  constexpr int NT = MKH_NT, FEAT = MKH_FEAT;
  constexpr bool kWood = (FEAT & F_WOOD) != 0;
#ifdef MKH_NR
  constexpr int NR = MKH_NR;
#else
  constexpr int NR = NT;
#endif
#ifdef MKH_W3
  constexpr bool kCompact = true;            // LDS ranges aliased by phase (lds_layout): 12 waves per CU need ≤ 13.3 KB each
#else
  constexpr bool kCompact = false;
#endif
  constexpr bool kWoodRows = kWood && (FEAT & (F_COLL | F_DENSE)) != 0;
  if constexpr (kWoodRows)     // (S never shares the dof stash here: collision_phase reads the axes again after the QP)
    return lds_layout(P0.nq, P0.nv, P0.nbody, P0.njnt, P0.n_frame, P0.n_posture, P0.n_com, P0.max_rows, P0.n_jrows + 1, NR,
                      P0.n_jrows * (lds_even(P0.n_jrows) + 1), kernel_prefetch(P0), false, true, (FEAT & F_COLL) ? P0.n_hsel : 0, true, true);
  return lds_layout(P0.nq, P0.nv, P0.nbody, P0.njnt, P0.n_frame, P0.n_posture, P0.n_com, P0.max_rows,
                    kWood ? P0.n_jrows + 1 : 6, kWood ? NR : j_stride_direct(P0.nv, NT),
                    (kWood && !wood_s_aliases_dof(P0.nv, P0.n_jrows, lds_even(P0.n_jrows), P0.n_com > 0 ? P0.nbody : 0))
                        ? P0.n_jrows * (lds_even(P0.n_jrows) + 1) : 0,
                    kernel_prefetch(P0), kCompact || (kWood && P0.wood_compact != 0), kWood,
                    (FEAT & F_COLL) ? P0.n_hsel : 0,        // (a compile-time 0 without collision rows: one value less to keep)
#if defined(MKH_W3) && (MKH_FEAT & 4) && (MKH_FEAT & 32)
                    true
#else
                    false
#endif
                    );
}
__device__ MKH_PRE_ATTR PreOut pre_phases(const DeviceProblem* Pq, const TapArgs* tp, int pb, int oz, int off_q, int off_tgt,
                                          bool until, double pos_thr, double ori_thr MKH_PRE_TC_PARAMS) {
  constexpr int FEAT = MKH_FEAT;
  constexpr bool kTaps = (FEAT & F_TAPS) != 0, kRel = (FEAT & F_REL) != 0, kCom = (FEAT & F_COM) != 0;
  constexpr bool kSteps = (FEAT & F_STEPS) != 0, kWood = (FEAT & F_WOOD) != 0;
  extern __shared__ __attribute__((aligned(16))) double smem[];
#if defined(MKH_CALLS) && defined(MKH_CLOCKS)
  const int clk_row_stop = clk_row ? (int)clk_row[-5] : 0;      // (clk_row = slot 20 of the problem's row)
#endif
#ifdef MKH_CALLS
  // arguments of a real call arrive in VGPRs: make the wave-uniform ones scalar again
  pb = uni(pb); oz = uni(oz); off_q = uni(off_q); off_tgt = uni(off_tgt);
  until = uni((int)until) != 0; pos_thr = uni(pos_thr); ori_thr = uni(ori_thr);
  Pq = reinterpret_cast<const DeviceProblem*>(uni((unsigned long long)reinterpret_cast<size_t>(Pq)));
  tp = reinterpret_cast<const TapArgs*>(uni((unsigned long long)reinterpret_cast<size_t>(tp)));
#endif
#ifdef MKH_CALLS
  const PreView P(Pq);          // (a callee cannot assume that pointers loaded from the descriptor are global memory)
#else
  const DeviceProblem& P = *Pq;
#endif
  int lane = lane_id();
  const int ol = lane + oz;
  const int nbody = P.nbody, nv = P.nv;
  const LdsLayout L = kernel_lds_layout(P);
  double* const sq = smem + off_q;
  double* const sX = smem + L.X;
  const int XS = lds_even(nbody);
  double* const sJnt = smem + L.jnt;
  double* const sTgt = smem + off_tgt;
  double* const sTask = smem + L.task;
  double* const sDof = smem + L.dof;
  double* const sCom = smem + L.com;
  const bool is_body = lane < nbody;
  const bool is_dof = lane < nv;
  int status = 0;
    // ------------------------------------------------- FK: local transforms
    // X = pose of body `lane` relative to its parent, joints applied
    // (mj_kinematics).
    V3 xp{0, 0, 0};
    Q4 xq{1, 0, 0, 0};
    int b_jadr = 0, b_jnum = 0, anc_lo = 0, anc_hi = 0;
    if (is_body) {
      anc_lo = P.body_i[BI_ANCPACK0 * 64 + ol];
      anc_hi = P.body_i[BI_ANCPACK1 * 64 + ol];
      const auto* bf = P.body_f + ol;
      xp = {bf[(BF_POS + 0) * 64], bf[(BF_POS + 1) * 64], bf[(BF_POS + 2) * 64]};
      xq = {bf[(BF_QUAT + 0) * 64], bf[(BF_QUAT + 1) * 64], bf[(BF_QUAT + 2) * 64], bf[(BF_QUAT + 3) * 64]};
      b_jadr = P.body_i[BI_JNTADR * 64 + ol];
      b_jnum = P.body_i[BI_JNTNUM * 64 + ol];
      for (int jn = 0; jn < b_jnum; ++jn) {
        const int j = b_jadr + jn;
        const int jt = P.jnt_i[j * JI_COUNT + JI_TYPE];
        const int qa = P.jnt_i[j * JI_COUNT + JI_QADR];
        // (the joint's constants with its type, not behind it: inside the branches they were a third dependent L2 round trip)
        const auto* jfp = P.jnt_f + j * JF_COUNT;
        const double jf[JF_COUNT] = {jfp[0], jfp[1], jfp[2], jfp[3], jfp[4], jfp[5], jfp[6]};
        if (jt == JNT_FREE) {
          xp = {sq[qa], sq[qa + 1], sq[qa + 2]};
          xq = qnormalize(Q4{sq[qa + 3], sq[qa + 4], sq[qa + 5], sq[qa + 6]});
        } else if (jt == JNT_SLIDE) {
          V3 ax{jf[JF_AXIS], jf[JF_AXIS + 1], jf[JF_AXIS + 2]};
          xp = xp + (sq[qa] - jf[JF_QPOS0]) * qrot(xq, ax);
        } else {
          V3 jp{jf[JF_POS], jf[JF_POS + 1], jf[JF_POS + 2]};
          Q4 qloc;
          if (jt == JNT_HINGE) {
            V3 ax{jf[JF_AXIS], jf[JF_AXIS + 1], jf[JF_AXIS + 2]};
            qloc = axis_angle(ax, sq[qa] - jf[JF_QPOS0]);
          } else {
            qloc = qnormalize(Q4{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]});
          }
          V3 anchor = xp + qrot(xq, jp);
          xq = qmul(xq, qloc);
          xp = anchor - qrot(xq, jp);
        }
      }
    }
    // ---------------------------------------- FK: pointer jumping to the root
    for (int r = 0; r < P.nrounds; ++r) {
      wave_sync();
      if (is_body) {
        double* o = sX + lane;
        o[0] = xp.x; o[XS] = xp.y; o[2 * XS] = xp.z; o[3 * XS] = xq.w; o[4 * XS] = xq.x; o[5 * XS] = xq.y; o[6 * XS] = xq.z;
      }
      wave_sync();
      if (is_body) {
        const double* a = sX + (((r < 5) ? (anc_lo >> (6 * r)) : anc_hi) & 63);
        V3 ap{a[0], a[XS], a[2 * XS]};
        Q4 aq{a[3 * XS], a[4 * XS], a[5 * XS], a[6 * XS]};
        xp = ap + qrot(aq, xp);
        xq = qmul(aq, xq);
      }
    }
    xq = qnormalize(xq);
    wave_sync();
    if (is_body) {
      double* o = sX + lane;
      o[0] = xp.x; o[XS] = xp.y; o[2 * XS] = xp.z; o[3 * XS] = xq.w; o[4 * XS] = xq.x; o[5 * XS] = xq.y; o[6 * XS] = xq.z;
      if (MKH_TAP(t_xpos)) {
        double* t = MKH_TAP(t_xpos) + ((size_t)pb * nbody + lane) * 3;
        t[0] = xp.x; t[1] = xp.y; t[2] = xp.z;
      }
      if (MKH_TAP(t_xquat)) {
        double* t = MKH_TAP(t_xquat) + ((size_t)pb * nbody + lane) * 4;
        t[0] = xq.w; t[1] = xq.x; t[2] = xq.y; t[3] = xq.z;
      }
    }
    wave_sync();
    MKH_MARK("fk_done");
    MKH_PRE_TICK();   // 1: FK done
    MKH_PRE_STOP(1);
    // --------------------- joint anchors / axes in the world (xanchor, xaxis)
    if (is_body && b_jnum > 0) {
      if (b_jnum == 1) {
        // single joint: rotation about its own axis/anchor leaves both invariant, so the
        // final body frame gives them directly.
        const int j = b_jadr;
        const auto* jf = P.jnt_f + j * JF_COUNT;
        V3 ax = qrot(xq, V3{jf[JF_AXIS], jf[JF_AXIS + 1], jf[JF_AXIS + 2]});
        V3 an = xp + qrot(xq, V3{jf[JF_POS], jf[JF_POS + 1], jf[JF_POS + 2]});
        double* o = sJnt + j * 6;
        o[0] = ax.x; o[1] = ax.y; o[2] = ax.z; o[3] = an.x; o[4] = an.y; o[5] = an.z;
      } else {
        // several joints in one body: replay them from the parent's world pose.
        const double* a = sX + P.body_i[BI_PARENT * 64 + ol];
        const auto* bf = P.body_f + ol;
        V3 fp = V3{a[0], a[XS], a[2 * XS]} + qrot(Q4{a[3 * XS], a[4 * XS], a[5 * XS], a[6 * XS]},
                                            V3{bf[(BF_POS + 0) * 64], bf[(BF_POS + 1) * 64], bf[(BF_POS + 2) * 64]});
        Q4 fq = qmul(Q4{a[3 * XS], a[4 * XS], a[5 * XS], a[6 * XS]}, Q4{bf[(BF_QUAT + 0) * 64], bf[(BF_QUAT + 1) * 64],
                                                    bf[(BF_QUAT + 2) * 64], bf[(BF_QUAT + 3) * 64]});
        for (int jn = 0; jn < b_jnum; ++jn) {
          const int j = b_jadr + jn;
          const int jt = P.jnt_i[j * JI_COUNT + JI_TYPE];
          const int qa = P.jnt_i[j * JI_COUNT + JI_QADR];
          const auto* jf = P.jnt_f + j * JF_COUNT;
          V3 axl{jf[JF_AXIS], jf[JF_AXIS + 1], jf[JF_AXIS + 2]};
          V3 jp{jf[JF_POS], jf[JF_POS + 1], jf[JF_POS + 2]};
          V3 ax = qrot(fq, axl);
          V3 an = fp + qrot(fq, jp);
          double* o = sJnt + j * 6;
          o[0] = ax.x; o[1] = ax.y; o[2] = ax.z; o[3] = an.x; o[4] = an.y; o[5] = an.z;
          if (jt == JNT_SLIDE) {
            fp = fp + (sq[qa] - jf[JF_QPOS0]) * ax;
          } else if (jt == JNT_HINGE || jt == JNT_BALL) {
            Q4 qloc = (jt == JNT_HINGE) ? axis_angle(axl, sq[qa] - jf[JF_QPOS0])
                                        : qnormalize(Q4{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]});
            fq = qmul(fq, qloc);
            fp = an - qrot(fq, jp);
          }
        }
      }
    }
    wave_sync();
    // --------------------------- dof lane: motion axis of dof `lane` (cdof)
    // jacp(p) = lin + ang × (p − anchor), jacr = ang  (mj_jac)
    int d_kind = DOF_HINGE, d_k = 0, d_body = 0, d_qadr = -1;
    {
    V3 d_ang{0, 0, 0}, d_lin{0, 0, 0}, d_anchor{0, 0, 0};
    double q_dof = 0.0;  // joint coordinate for hinge/slide dofs
    // (the joint range of check_limits below, requested with the dof's other table entries: as `lo || hi` inside the two
    //  conditionals it was four loads one after the other, each behind its own full wait — 3 k of this phase's 7 k cycles)
    const double r_lo = P.dof_f[DF_RANGE_LO * 64 + ol], r_hi = P.dof_f[DF_RANGE_HI * 64 + ol];
    if (is_dof) {
      const auto* di = P.dof_i + ol;
      const int d_jnt = di[DI_JNT * 64];
      d_kind = di[DI_KIND * 64];
      d_k = di[DI_K * 64];
      d_body = di[DI_BODY * 64];
      d_qadr = di[DI_QADR * 64];
      if (d_kind == DOF_HINGE) {
        const double* o = sJnt + d_jnt * 6;
        d_ang = {o[0], o[1], o[2]};
        d_anchor = {o[3], o[4], o[5]};
        q_dof = sq[d_qadr];
      } else if (d_kind == DOF_SLIDE) {
        const double* o = sJnt + d_jnt * 6;
        d_lin = {o[0], o[1], o[2]};
        q_dof = sq[d_qadr];
      } else if (d_kind == DOF_FREE_LIN) {
        d_lin = {d_k == 0 ? 1.0 : 0.0, d_k == 1 ? 1.0 : 0.0, d_k == 2 ? 1.0 : 0.0};
      } else {  // ball / free rotational dof: body-frame axis k, about the joint anchor
        const double* xb = sX + d_body;
        M3 R = qmat(Q4{xb[3 * XS], xb[4 * XS], xb[5 * XS], xb[6 * XS]});
        d_ang = (d_k == 0) ? V3{R.m[0], R.m[3], R.m[6]}
                           : ((d_k == 1) ? V3{R.m[1], R.m[4], R.m[7]} : V3{R.m[2], R.m[5], R.m[8]});
        if (d_kind == DOF_FREE_ANG) {
          d_anchor = {xb[0], xb[XS], xb[2 * XS]};
        } else {
          const double* o = sJnt + d_jnt * 6;
          d_anchor = {o[3], o[4], o[5]};
        }
      }
    }
    // Configuration.check_limits (mink/configuration.py:77-110), tol = 1e-6
    {
      // a limited ball joint: the reference's loop compares q[jnt_qposadr] — the quaternion's w — with the range
      // (configuration.py:92-99); the range sits on the joint's first dof lane only
      const bool ball0 = is_dof && d_kind == DOF_BALL && d_k == 0;
      const bool ranged = (is_dof && (d_kind == DOF_HINGE || d_kind == DOF_SLIDE)) || ball0;
      const double qc = ball0 ? sq[ball0 ? d_qadr : 0] : q_dof;
      const bool viol = ranged & ((qc < r_lo - 1e-6) | (qc > r_hi + 1e-6));
      if (__ballot(viol)) status |= 1;
    }
    // stash the dof's motion axis in LDS; phases below reload it instead of keeping 20 VGPRs live
    if (is_dof) {
      double* o = sDof + lane * 10;
      o[0] = d_ang.x; o[1] = d_ang.y; o[2] = d_ang.z; o[3] = d_lin.x; o[4] = d_lin.y; o[5] = d_lin.z;
      o[6] = d_anchor.x; o[7] = d_anchor.y; o[8] = d_anchor.z; o[9] = q_dof;
    }
    }

    // -------------------------------------- subtree CoM (mj_comPos) for ComTask
    V3 com_root{0, 0, 0};
    if (kCom && P.n_com > 0) {
      V3 b_ipos{0, 0, 0};
      double b_mass = 0.0, b_stmass = 0.0;
      int b_last = 0, b_inrobot = 0;
      if (is_body) {
        const auto* bf = P.body_f + ol;
        b_ipos = {bf[(BF_IPOS + 0) * 64], bf[(BF_IPOS + 1) * 64], bf[(BF_IPOS + 2) * 64]};
        b_mass = bf[BF_MASS * 64];
        b_stmass = bf[BF_SUBTREEMASS * 64];
        b_last = P.body_i[BI_SUBTREE_LAST * 64 + ol];
        b_inrobot = P.body_i[BI_IN_ROBOT * 64 + ol];
      }
      V3 xi = xp + qrot(xq, b_ipos);
      const double m = (is_body && b_inrobot) ? b_mass : 0.0;
      double sx = m * xi.x, sy = m * xi.y, sz = m * xi.z;
      // inclusive scan over body ids (a subtree is a contiguous id range in MuJoCo's DFS order)
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        double tx = __shfl_up(sx, o), ty = __shfl_up(sy, o), tz = __shfl_up(sz, o);
        if (lane >= o) { sx += tx; sy += ty; sz += tz; }
      }
      const int hi_l = is_body ? b_last : 0, lo_l = lane - 1;
      double ex = __shfl(sx, hi_l), ey = __shfl(sy, hi_l), ez = __shfl(sz, hi_l);
      double bx = __shfl(sx, lo_l < 0 ? 0 : lo_l), by = __shfl(sy, lo_l < 0 ? 0 : lo_l),
             bz = __shfl(sz, lo_l < 0 ? 0 : lo_l);
      if (lane == 0) { bx = 0; by = 0; bz = 0; }
      V3 cs = xi;
      if (b_stmass >= 1e-15) {
        const double im = fast_rcp(b_stmass);
        cs = {(ex - bx) * im, (ey - by) * im, (ez - bz) * im};
      }
      if (is_body) {
        double* o = sCom + lane * 4;
        o[0] = cs.x; o[1] = cs.y; o[2] = cs.z; o[3] = b_stmass;
      }
      wave_sync();
      const double* cr = sCom + P.robot_root * 4;
      com_root = {cr[0], cr[1], cr[2]};
      if (MKH_TAP(t_subtree_com) && lane < 3) MKH_TAP(t_subtree_com)[(size_t)pb * 3 + lane] = cr[lane];
    }

    MKH_MARK("axes_done");
    MKH_PRE_TICK();   // 2: joint axes / dof lanes / com done
    MKH_PRE_STOP(2);
    // ------------------------------------------- task lanes: pose, error, jlog
    double mu_lane = 0.0;  // Levenberg–Marquardt term of the task owned by this lane
    bool conv_lane = true; // this lane's frame task is within the thresholds (rows with a nonzero cost only)
    if (lane < P.n_frame) {
      const auto& ft = P.frame[lane];
      const double* xb = sX + ft.body;
      SE3 F;
      Q4 bq{xb[3 * XS], xb[4 * XS], xb[5 * XS], xb[6 * XS]};
      F.p = V3{xb[0], xb[XS], xb[2 * XS]} + qrot(bq, V3{ft.lpos[0], ft.lpos[1], ft.lpos[2]});
      F.q = qmul(bq, Q4{ft.lquat[0], ft.lquat[1], ft.lquat[2], ft.lquat[3]});
      const double* tg = sTgt + lane * 7;
      SE3 Tt{Q4{tg[0], tg[1], tg[2], tg[3]}, V3{tg[4], tg[5], tg[6]}};
      double* o = sTask + lane * 64;
      V3 ev, ew;
      double Jm[9], Qm[9];
      bool ident;
      if (!(kRel && ft.relative)) {
        // e = target.minus(frame) = log(T_frame⁻¹ · T_target)          (frame_task.py:119-122)
        se3_log(se3_mul(se3_inv(F), Tt), ev, ew);
        // jlog(T_tb) = ljacinv(−log(T_tb)) = ljacinv(e)   since T_tb = T_bt⁻¹   (frame_task.py:144-146)
        se3_ljacinv(ev, ew, Jm, Qm, ident);
      } else {
        // RelativeFrameTask (relative_frame_task.py:106-142): T_fr = T_root⁻¹·T_frame,
        // e = T_fr.rminus(target) = log(target⁻¹·T_fr),  J = jlog(T_tf)·(ᶠJ − Ad(T_fr⁻¹)·ʳJ)
        const double* xr = sX + ft.root_body;
        Q4 rq0{xr[3 * XS], xr[4 * XS], xr[5 * XS], xr[6 * XS]};
        SE3 Rt;
        Rt.p = V3{xr[0], xr[XS], xr[2 * XS]} + qrot(rq0, V3{ft.root_lpos[0], ft.root_lpos[1], ft.root_lpos[2]});
        Rt.q = qmul(rq0, Q4{ft.root_lquat[0], ft.root_lquat[1], ft.root_lquat[2], ft.root_lquat[3]});
        const SE3 Tfr = se3_mul(se3_inv(Rt), F);
        se3_log(se3_mul(se3_inv(Tt), Tfr), ev, ew);
        se3_ljacinv(-1.0 * ev, -1.0 * ew, Jm, Qm, ident);       // jlog(T) = ljacinv(−log T)
        const SE3 Trf = se3_inv(Tfr);
        const M3 Rr = qmat(Rt.q), Rrf = qmat(Trf.q);
#pragma unroll
        for (int i = 0; i < 9; ++i) { o[36 + i] = Rr.m[i]; o[48 + i] = Rrf.m[i]; }
        o[45] = Rt.p.x; o[46] = Rt.p.y; o[47] = Rt.p.z;
        o[57] = Trf.p.x; o[58] = Trf.p.y; o[59] = Trf.p.z;
        if (MKH_TAP(t_frame_pose)) {                            // tap: pose of the frame in the root
          double* t = MKH_TAP(t_frame_pose) + ((size_t)pb * P.n_frame + lane) * 7;
          t[0] = Tfr.q.w; t[1] = Tfr.q.x; t[2] = Tfr.q.y; t[3] = Tfr.q.z; t[4] = Tfr.p.x; t[5] = Tfr.p.y; t[6] = Tfr.p.z;
        }
      }
      M3 Rf = qmat(F.q);
#pragma unroll
      for (int i = 0; i < 9; ++i) { o[i] = Jm[i]; o[9 + i] = Qm[i]; o[18 + i] = Rf.m[i]; }
      o[27] = F.p.x; o[28] = F.p.y; o[29] = F.p.z;
      const double e6[6] = {ev.x, ev.y, ev.z, ew.x, ew.y, ew.z};
      double ss = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double we = ft.cost[r] * (-ft.gain * e6[r]);  // weighted_error (task.py:129-130)
        o[30 + r] = we;
        ss += we * we;
        if (MKH_TAP(t_task_e)) MKH_TAP(t_task_e)[(size_t)pb * P.n_rows_tap + ft.row0 + r] = e6[r];
      }
      mu_lane = ft.lm_damping * ss;                          // task.py:131
      if (kSteps && until) {
        const double pt = pos_thr, ot = ori_thr;
        conv_lane = (!(ft.rowmask & 7) || dot(ev, ev) <= pt * pt) && (!(ft.rowmask & 56) || dot(ew, ew) <= ot * ot);
      }
      if (MKH_TAP(t_frame_pose) && !(kRel && ft.relative)) {
        double* t = MKH_TAP(t_frame_pose) + ((size_t)pb * P.n_frame + lane) * 7;
        t[0] = F.q.w; t[1] = F.q.x; t[2] = F.q.y; t[3] = F.q.z; t[4] = F.p.x; t[5] = F.p.y; t[6] = F.p.z;
      }
    }
  PreOut po;
  po.status = status; po.d_kind = d_kind; po.d_k = d_k; po.d_body = d_body; po.d_qadr = d_qadr;
  po.conv = conv_lane ? 1 : 0; po.mu_lane = mu_lane; po.com_root = com_root;
  return po;
}

// Column `k` of frame task t's Jacobian (6 rows, unweighted): dof k must be on the chain of the frame (or of the
// root frame of a RelativeFrameTask).  sTask / sDof: the task blocks and dof axes in LDS.
template <bool kRel>
__device__ __forceinline__ void frame_column_fn(const double* sTask, const double* sDof, int t, int k, uint64_t mask,
                                                uint64_t rmask, bool rel, double (&Jt)[6]) {
    const double* o = sTask + t * 64;
    const double* kd = sDof + k * 10;
    const V3 d_ang{kd[0], kd[1], kd[2]}, d_lin{kd[3], kd[4], kd[5]}, d_anchor{kd[6], kd[7], kd[8]};
    V3 a{0, 0, 0}, w{0, 0, 0};
    if ((mask >> k) & 1) {
      V3 pf{o[27], o[28], o[29]};
      V3 jp = d_lin + cross(d_ang, pf - d_anchor);
      M3 Rf;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rf.m[i] = o[18 + i];
      a = mulT(Rf, jp); w = mulT(Rf, d_ang);             // body-frame Jacobian (configuration.py:148-153)
    }
    double sign = -1.0;                                   // FrameTask: J = −jlog(T_tb)·ᴮJ
    if (kRel && rel) {
      sign = 1.0;                                         // RelativeFrameTask: J = +jlog(T_tf)·(ᶠJ − Ad·ʳJ)
      if ((rmask >> k) & 1) {
        V3 pr{o[45], o[46], o[47]};
        V3 jp = d_lin + cross(d_ang, pr - d_anchor);
        M3 Rr, Rrf;
#pragma unroll
        for (int i = 0; i < 9; ++i) { Rr.m[i] = o[36 + i]; Rrf.m[i] = o[48 + i]; }
        V3 ar = mulT(Rr, jp), wr = mulT(Rr, d_ang);       // root's body-frame column
        V3 Rw = mul(Rrf, wr);                             // Ad(T_fr⁻¹) = [[R, [t]×R],[0, R]]
        a = a - (mul(Rrf, ar) + cross(V3{o[57], o[58], o[59]}, Rw));
        w = w - Rw;
      }
    }
    // jlog = [[J, −J·Q·J],[0, J]]:  y = J·w;  rows 0-2 = ±J·(a − Q·y), rows 3-5 = ±y
    const double wv[3] = {w.x, w.y, w.z};
    double y[3], z3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) y[r] = o[3 * r] * wv[0] + o[3 * r + 1] * wv[1] + o[3 * r + 2] * wv[2];
    const double av[3] = {a.x, a.y, a.z};
#pragma unroll
    for (int r = 0; r < 3; ++r)
      z3[r] = av[r] - (o[9 + 3 * r] * y[0] + o[9 + 3 * r + 1] * y[1] + o[9 + 3 * r + 2] * y[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      Jt[r] = sign * (o[3 * r] * z3[0] + o[3 * r + 1] * z3[1] + o[3 * r + 2] * z3[2]);
      Jt[3 + r] = sign * y[r];
    }
  }

// Direct start: the Jacobian columns of EVERY frame task at once, one (task, dof) pair per lane — weighted rows of this
// lane's pair.  A real call in the 3-waves-per-SIMD variants (see pre_phases).
struct DirectPairs { double Jp[6]; int p_task, p_dof; };
__device__ MKH_PRE_ATTR DirectPairs direct_pairs(const DeviceProblem* Pq) {
  constexpr bool kRel = (MKH_FEAT & F_REL) != 0;
  extern __shared__ __attribute__((aligned(16))) double smem[];
#ifdef MKH_CALLS
  Pq = reinterpret_cast<const DeviceProblem*>(uni((unsigned long long)reinterpret_cast<size_t>(Pq)));
  const MKH_CONSTANT DeviceProblem& P = *(const MKH_CONSTANT DeviceProblem*)Pq;
  const MKH_GLOBAL FrameTaskDev* const frames = (const MKH_GLOBAL FrameTaskDev*)P.frame;
#else
  const DeviceProblem& P = *Pq;
  const FrameTaskDev* const frames = P.frame;
#endif
  const int lane = lane_id();
  const LdsLayout L = kernel_lds_layout(P);
  DirectPairs dp{{0, 0, 0, 0, 0, 0}, -1, 0};
  if (lane < P.n_dpairs) {
    dp.p_task = P.dpair_task[lane]; dp.p_dof = P.dpair_dof[lane];
    const auto& ft = frames[dp.p_task];
    const bool rel = kRel && ft.relative != 0;
    double Jt[6];
    frame_column_fn<kRel>(smem + L.task, smem + L.dof, dp.p_task, dp.p_dof, ft.dof_mask, rel ? ft.root_mask : 0ull, rel, Jt);
#pragma unroll
    for (int r = 0; r < 6; ++r) dp.Jp[r] = ft.cost[r] * Jt[r];   // weighted_jacobian (task.py:129)
  }
  return dp;
}


#if (MKH_FEAT & 32)
// ---------------------------------------------------------------- low-rank start of the QP (F_WOOD)
// H = Dg + JwᵀJw with Dg diagonal (damping + Σ LM terms + posture tasks) and Jw the n_μ weighted rows of the frame tasks
// (and of the ComTasks).  With σ = 1/√Dg, Jh = Jw·σ and S = I + Jh·Jhᵀ = L·D·Lᵀ (n_μ × n_μ, SPD, diagonal ≥ 1 — no pivoting
// needed):
//     −H⁻¹ = −σσᵀ∘(I − JhᵀS⁻¹Jh),   JhᵀS⁻¹Jh = ZᵀD⁻¹Z  with  Z = L⁻¹Jh.
// One COLUMN of the augmented matrix [S | Jh] per lane in K compiler-allocated registers: dof lanes [0, NR) carry a column
// of Jh, n_μ lanes a column of S together with their entry of the right-hand side w — lanes [NR, NR + n_μ) in the same
// registers, or, when NR + n_μ exceeds the wavefront (the G1 full example: 44 + 24), lanes [0, n_μ) in a second set (DUAL).
// Step r publishes row r (the dof part is row r of Z, written to its final place in LDS), every lane reads the
// n_μ − r − 1 multipliers S[i][r] as two 16-lane planes and updates its rows below r with DPP-broadcast FMAs (tab_asm.inc
// WoodElim).  The chain per step is one LDS round trip + one reciprocal + ≤ K − 1 FMAs and never touches the tableau:
// the kernel then builds the dof block with n_μ rank-1 updates R[i][j] += Z[r][i]·Z[r][j]/d_r that do not depend on each
// other.  (Round 2's first version swept the residual indices on the tableau itself — every pivot a publish /
// reciprocal / multiplier chain in front of a 62-row update, later six at a time with a per-lane 6 × 6 LDLᵀ: 30 % of a
// G1 solve.)
// Outputs per dof lane: D = −H⁻¹[j][j] = σ²(Σ_r z_r²/d_r − 1),  x0 = −H⁻¹c = z − σ·Σ_r z_r·ω_r/d_r  (ω = L⁻¹w, w = Jw·z − r).
struct WoodOut { double hdiag, dsq, x, D, rown; int status, clamp; };

// cold-start refinement of wood_eliminate (single-pass lane layout): bounds, the dof's closed-form point and scales
struct WoodRefine { bool on; int nv; double lo, hi, zfree, dsq, sqdg, zsq; };
template <int K, bool DUAL>
__device__ __forceinline__ void wood_eliminate(int n_mu, int NR, int lane, double* sJ, double* sS, int SP, const double* sW,
                                               double* sRow, double* sDinv, double& ssq, double& quad, double& zw, int& status,
                                               const WoodRefine& rf, int& clamp, double& beta) {
  const unsigned plane_off = (unsigned)(lane & 15);
  if constexpr (!DUAL) {
    // ONE pass: lanes [0, NR) carry a column of Jh, lanes [NR, NR + n_μ) a column of S and their entry w_c
    const int my_c = lane - NR;                                          // S column of this lane (if 0 ≤ my_c < n_μ)
    const bool is_s = my_c >= 0 && my_c < n_mu;
    double z[K];
    {
      const double* src = sJ + lane;
      int stride = NR;
      bool live = lane < NR;
      if (is_s) { src = sS + my_c * SP; stride = 1; live = true; }
#pragma unroll
      for (int r = 0; r < K; ++r) z[r] = (live && r < n_mu) ? src[r * stride] : 0.0;
    }
    double ws = is_s ? sW[my_c] : 0.0;
    {
      double q = 0.0;
#pragma unroll
      for (int r = 0; r < K; ++r) q = fma(z[r], z[r], q);                // Σ Jh² of a dof lane (before the elimination):
      if (lane < NR) sJ[n_mu * NR + lane] = q;                           // parked in LDS (the right-hand-side row of the product
    }                                                                    // is dead), not in two registers through the loop
    double* const sOm = sDinv + kMuBig;
#pragma nounroll
    for (int pass = 0;; ++pass) {
      quad = 0.0; zw = 0.0;
      static_for<K>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if (r < n_mu) {
          // row r of S straight from the S lanes' registers: d_r and ω_r by v_readlane (the reciprocal starts at once),
          // the two 16-lane planes through the LDS crossbar (a store → wait → load round trip per step doubled the chain)
          const double d = readlane_f64(z[r], NR + r), om = readlane_f64(ws, NR + r);
          const double p0 = bperm_f64(z[r], NR + (int)plane_off);
          // (unconditional: a permute under a lane condition would not see the source lanes the condition switches off)
          const double p1 = (K > 16) ? bperm_f64(z[r], (NR + 16 + (int)plane_off) & 63) : 0.0;   // rows ≥ n_μ: never used
          if (lane < NR) sJ[r * NR + lane] = z[r];                       // row r of Z, final (read after the loop)
          if (!(d > 0.0)) status |= 4;
          const double inv = fast_rcp(d);
          if (lane == 0) { sDinv[r] = inv; sOm[r] = om; }
          const double zi = z[r] * inv;
          quad = fma(z[r], zi, quad);
          zw = fma(om, zi, zw);
          WoodElim<r, K>::step(z, p0, p1, -zi);                          // z[i] −= S[i][r]·z[r]/d   (i > r)
          ws = fma(-zi, om, ws);                                         // w_c −= S[c][r]·ω_r/d  (S lanes; harmless on dof lanes)
        }
      });
      if (pass == 1 || !rf.on) break;
      // ---- cold start: the bounds the unconstrained minimiser x⁰ violates are (nearly) the optimal active set — block
      // principal pivoting would put exactly these dofs on their bounds, one un-sweep pivot of the whole tableau each.
      // Cheaper here, where the tableau does not exist yet: with A the violated set, S_F = S − Jh_A·Jh_Aᵀ = L·(D − Z_A·Z_Aᵀ)·Lᵀ,
      // so a second elimination of the n_μ × n_μ matrix M = D − Z_A·Z_Aᵀ = L_M·D_M·L_Mᵀ — columns on the S lanes again, summed
      // over the |A| violated dofs only — carries Z to L_M⁻¹·Z and ω to L_M⁻¹·(ω + Σ_A Z[·][k]·(β_k − z_k)·√Dg_k): the state of
      // the predicted-set start (wood_start), from what the first pass left in registers and LDS.
      const double x0 = rf.zfree - rf.dsq * zw;
      const int viol = (lane < rf.nv) ? ((x0 > rf.hi) ? 2 : ((x0 < rf.lo) ? 1 : 0)) : 0;
      const unsigned long long am = __ballot(viol != 0);
      if (!am) break;
      clamp = viol;
      beta = (viol == 2) ? rf.hi : ((viol == 1) ? rf.lo : 0.0);
      double dself = 0.0;
#pragma unroll
      for (int r = 0; r < K; ++r) dself = (r == my_c) ? z[r] : dself;    // d_c: entry c of an S column is final after step c
      double om_c = is_s ? sOm[my_c] : 0.0;
      wave_sync();                                                       // the last row's readers are done
      if (viol) sRow[lane] = beta * rf.sqdg - rf.zsq;                    // (β_k − z_k)·√Dg_k
      if (is_s) {
#pragma unroll
        for (int r = 0; r < K; ++r) z[r] = (r == my_c) ? dself : 0.0;
      }
      wave_sync();
      {
        const int rows0 = (int)plane_off, rows1 = 16 + (int)plane_off;
        const bool has0 = rows0 < n_mu, has1 = (K > 16) && rows1 < n_mu;
        const double* c0 = sJ + (has0 ? rows0 : 0) * NR;
        const double* c1 = sJ + (has1 ? rows1 : 0) * NR;
        const double* cm = sJ + (is_s ? my_c : 0) * NR;
        for (unsigned long long mk = am; mk; mk &= mk - 1) {
          const int k = __ffsll(mk) - 1;
          const double p0 = has0 ? c0[k] : 0.0, p1 = has1 ? c1[k] : 0.0;
          const double g = is_s ? cm[k] : 0.0;
          WoodAll<K>::step(z, p0, p1, -g);                               // M[r][c] −= Z[r][k]·Z[c][k]  (dof lanes: g = 0)
          om_c = fma(g, sRow[k], om_c);
        }
      }
      ws = om_c;
    }
  } else {
    // TWO passes (NR + n_μ lanes do not exist): first the S columns on lanes [0, n_μ) — the chained part; row r, 1/d_r and ω_r
    // stay in LDS (S is loaded into registers first, its storage takes the rows) — then the Jh columns on the dof lanes, which
    // only read multipliers: no publish / read-back chain at all.
    const bool is_s = lane < n_mu;
    double* const sOm = sDinv + kMuBig;
    {
      double z[K];
#pragma unroll
      for (int r = 0; r < K; ++r) z[r] = (is_s && r < n_mu) ? sS[lane * SP + r] : 0.0;
      double ws = is_s ? sW[lane] : 0.0;
      wave_sync();                                                       // every column is in registers: sS takes the rows
      static_for<K>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if (r < n_mu) {
          const double d = readlane_f64(z[r], r), om = readlane_f64(ws, r);      // (as in the one-pass layout)
          const double p0 = bperm_f64(z[r], (int)plane_off);
          const double p1 = (K > 16) ? bperm_f64(z[r], 16 + (int)plane_off) : 0.0;
          if (is_s) sS[r * SP + lane] = z[r];                                    // row r for the dof lanes' pass
          if (!(d > 0.0)) status |= 4;
          const double inv = fast_rcp(d);
          if (lane == 0) { sDinv[r] = inv; sOm[r] = om; }
          const double zi = z[r] * inv;
          WoodElim<r, K>::step(z, p0, p1, -zi);
          ws = fma(-zi, om, ws);
        }
      });
    }
    wave_sync();
    {
      double z[K];
      const bool live = lane < NR;
#pragma unroll
      for (int r = 0; r < K; ++r) z[r] = (live && r < n_mu) ? sJ[r * NR + lane] : 0.0;
      {
        double q = 0.0;
#pragma unroll
        for (int r = 0; r < K; ++r) q = fma(z[r], z[r], q);
        wave_sync();
        if (live) sJ[n_mu * NR + lane] = q;
      }
      static_for<K>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if (r < n_mu) {
          const double inv = sDinv[r], om = sOm[r];
          const double p0 = sS[r * SP + plane_off], p1 = (K > 16) ? sS[r * SP + 16 + plane_off] : 0.0;
          if (live) sJ[r * NR + lane] = z[r];                            // row r of Z, final
          const double zi = z[r] * inv;
          quad = fma(z[r], zi, quad);
          zw = fma(om, zi, zw);
          WoodElim<r, K>::step(z, p0, p1, -zi);
        }
      });
    }
  }
  wave_sync();
  ssq = lane < NR ? sJ[n_mu * NR + lane] : 0.0;
}

// One more resident wave per SIMD for the F_COM builds (MKH_W3 with ComTask rows: the reference's humanoid example as written):
// wood_start as ONE callee needs ≈150 VGPRs there — the dense Jh·Jhᵀ product keeps 24 accumulators, the elimination a 24-row
// column, and hipcc lets both ranges overlap with the pair lanes' values — and everything beyond the 104 caller-saved registers
// below v168 is a callee-saved block (v40-47, v56-63, …) that the prologue stores to scratch and the epilogue reloads: 45
// registers, 360 B per lane and call, the reason this build measured 2.27 ms against 1.48 ms on two waves in round 4.  The two
// heavy phases are callees of their own here, each inside the caller-saved set; they talk through LDS offsets (a pointer
// argument would arrive as a flat address).
#if defined(MKH_W3) && (MKH_FEAT & 4)
#define MKH_WOOD_SPLIT 1
struct WoodElimOut { double ssq, quad, zw; int status; };
template <int K, bool DUAL>
__device__ __attribute__((noinline)) WoodElimOut wood_eliminate_call(int n_mu, int off_J, int off_S, int SP, int off_Row) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  n_mu = uni(n_mu); off_J = uni(off_J); off_S = uni(off_S); SP = uni(SP); off_Row = uni(off_Row);
  const int lane = lane_id();
  double* const sS = smem + off_S;
  double* const sRow = smem + off_Row;
  WoodElimOut o{0.0, 0.0, 0.0, 0};
  int clamp = 0;
  double beta = 0.0;
  const WoodRefine rf{false, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};          // (no cold-start refinement in the F_COM builds)
  wood_eliminate<K, DUAL>(n_mu, MKH_NT, lane, smem + off_J, sS, SP, sS + n_mu * SP, sRow, sRow + kWoodRow, o.ssq, o.quad, o.zw, o.status,
                          rf, clamp, beta);
  return o;
}
// the dense product S = I + Jh·Jhᵀ and Jw·z̃ of the F_COM builds (see wood_start)
__device__ __attribute__((noinline)) void wood_s_dense_call(int n_mu, int nv, int off_J, int off_S, int SP, unsigned a_lo, unsigned a_hi) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NR = MKH_NT;
  n_mu = uni(n_mu); nv = uni(nv); off_J = uni(off_J); off_S = uni(off_S); SP = uni(SP);
  const unsigned long long a_mask = ((unsigned long long)(unsigned)uni((int)a_hi) << 32) | (unsigned)uni((int)a_lo);
  const int lane = lane_id();
  const double* const sJ = smem + off_J;
  double* const sS = smem + off_S;
  double* const sW = sS + n_mu * SP;
  double acc[kMuBig];
#pragma unroll
  for (int r = 0; r < kMuBig; ++r) acc[r] = 0.0;
  double wacc = 0.0;
  const int rows0 = (int)(lane & 15), rows1 = 16 + (int)(lane & 15);
  const bool has0 = rows0 < n_mu, has1 = rows1 < n_mu, mine = lane < n_mu;
  const double* c0 = sJ + (has0 ? rows0 : 0) * NR;
  const double* c1 = sJ + (has1 ? rows1 : 0) * NR;
  const double* cm = sJ + (mine ? lane : 0) * NR;
  const double* zs = sJ + n_mu * NR;
  for (int k = 0; k < nv; ++k) {
    const double p0 = has0 ? c0[k] : 0.0, p1 = has1 ? c1[k] : 0.0;
    const double g = mine ? cm[k] : 0.0;
    WoodAll<kMuBig>::step(acc, p0, p1, ((a_mask >> k) & 1) ? 0.0 : g);   // acc[r] += Jh[r][k]·Jh[c][k], free dofs only
    wacc = fma(g, zs[k], wacc);
  }
  if (mine) {
#pragma unroll
    for (int r = 0; r < kMuBig; ++r)
      if (r < n_mu) sS[lane * SP + r] = acc[r] + (r == lane ? 1.0 : 0.0);
    sW[lane] = wacc;
  }
}
#endif

// A real call in the 3-waves variants (see pre_phases) and in the F_COM builds, whose 24-row / two-pass instantiations do not
// fit next to the kernel's own live values (86 spilled VGPRs when inlined).
#if defined(MKH_CALLS) || (MKH_FEAT & 4)
#define MKH_WOOD_CALL 1
#ifdef MKH_ONE_SHOT
#define MKH_WOOD_ATTR __attribute__((noinline, internal_linkage))
#else
#define MKH_WOOD_ATTR __attribute__((noinline))
#endif
#else
#define MKH_WOOD_ATTR __forceinline__
#endif
// clamp / beta: the PREDICTED active set — this dof is expected on a bound (1 lower, 2 upper; 0 free) of value beta.  The
// tableau is then built directly in the state "free dofs swept, predicted dofs not": with S summed over the FREE columns
// only, Z = L⁻¹Jh over ALL columns and the lazy scale 1/σ instead of σ on a predicted dof, every off-diagonal entry is
// still R[i][j] = Σ_r Z[r][i]·Z[r][j]/d_r (T_FF = −H_FF⁻¹, T_AF = H_AF·H_FF⁻¹, T_AA = H_AA − H_AF·H_FF⁻¹·H_FA all reduce to
// it through (I − Jh_FᵀS⁻¹Jh_F)·Jh_Fᵀ = Jh_FᵀS⁻¹ and I − Jh_F·H̃_FF⁻¹·Jh_Fᵀ = S⁻¹); a predicted dof gets the diagonal
// Dg·(1 + Σ z_r²/d_r) and its gradient Dg·β + c + √Dg·Σ z_r·ω_r/d_r, the right-hand side takes z̃ = β on predicted dofs.
// No un-sweep pivot at all for a correct prediction: the warm starts of the fused loop and of MKH_FLAG_WARM_START.
// nrows (builds with half-space rows, round 6): the rows A[s][:] staged in LDS by the collision phase take part — see "half-space
// rows" below; their lanes [nv, nv + nrows) get D = T_rr[s][s], x = the slack A·x0 − h and rown = ‖A[s]‖ back.
__device__ MKH_WOOD_ATTR WoodOut wood_start(const DeviceProblem* Pq, int oz, int off_q, int off_tgt, double c_lane, double hdiag_base,
                                             int clamp, double beta, double lo, double hi, int nrows,
                                             long long* prof = nullptr) {   // prof (inlined profiling build only): cycle stamps
  constexpr int NR = MKH_NT;
  constexpr bool kCom = (MKH_FEAT & F_COM) != 0;
  constexpr bool kRowsW = (MKH_FEAT & (F_COLL | F_DENSE)) != 0;         // half-space rows next to the low-rank start
  static_assert(MKH_NR == MKH_NT, "low-rank start: the residual rows are not tableau rows any more");
  extern __shared__ __attribute__((aligned(16))) double smem[];
#ifdef MKH_WOOD_CALL
  oz = uni(oz); off_q = uni(off_q); off_tgt = uni(off_tgt); nrows = uni(nrows);
  Pq = reinterpret_cast<const DeviceProblem*>(uni((unsigned long long)reinterpret_cast<size_t>(Pq)));
  const MKH_CONSTANT DeviceProblem& P = *(const MKH_CONSTANT DeviceProblem*)Pq;
  const MKH_GLOBAL FrameTaskDev* const frames = (const MKH_GLOBAL FrameTaskDev*)P.frame;
  const MKH_GLOBAL int32_t* const dof_i = (const MKH_GLOBAL int32_t*)P.dof_i;
  const MKH_GLOBAL int32_t* const body_i = (const MKH_GLOBAL int32_t*)P.body_i;
#else
  const DeviceProblem& P = *Pq;
  const FrameTaskDev* const frames = P.frame;
  const int32_t* const dof_i = P.dof_i;
  const int32_t* const body_i = P.body_i;
#endif
  int lane = lane_id();
  const int ol = lane + oz;
  const int nv = P.nv, n_mu = P.n_jrows;
  const int SP = lds_even(n_mu);                      // stride of a column of S
  const LdsLayout L = kernel_lds_layout(P);
  double* const sTask = smem + L.task;
  double* const sTgt = smem + off_tgt;
  double* const sJ = smem + L.J;
  double* const sDof = smem + L.dof;
  double* const sCom = smem + L.com;
  double* const sRow = smem + L.piv;                  // published row: pairs (S[r][c], w_c) for c < n_μ
  double* const sDinv = sRow + kWoodRow;              // 1/d_r for the kernel's rank-1 updates
  const bool is_dof = lane < nv;
  const bool dual = NR + n_mu > kWave;
  const int my_c = dual ? lane : lane - NR;
  const bool is_s = my_c >= 0 && my_c < n_mu;
  int status = 0;
  // 1/√Dg of this dof (Dg = damping + Σμ + posture diagonal > 0, checked on the host)
  const double dsq = is_dof ? fast_rcp(sqrt(hdiag_base)) : 0.0;
  const bool clamped = is_dof && clamp != 0;
  const unsigned long long a_mask = __ballot(clamped);                 // predicted active set (wave-uniform)
  // weighted error of residual row my_c (read before the Jacobian rows may overwrite the task blocks):
  // a frame-task row, or (−1 − 3·t − r) row r of ComTask t: cost·(−gain·(com − target))   (com_task.py:71-82)
  // (per-lane entries of the descriptor's tables, requested together here — each used to be fetched where it is needed, one L2
  //  round trip after the other: the residual's source, the first pass of (task, dof) pairs, the lane's column of S)
  // (humanoid-size builds only: in the 16- / 24- / 32-row builds the six values cost this function spilled registers)
  constexpr bool kPre = NR >= 44;
  int pre_src = 0, pre_t = 0, pre_k = 0, pre_wc = 0, pre_wr0 = 0;
  uint64_t pre_chain = 0;
  if constexpr (kPre) {
    pre_src = P.mu_src[is_s ? my_c : 0];
    pre_t = P.jpair_task[lane]; pre_k = P.jpair_dof[lane];
    pre_wc = P.wood_col[ol]; pre_wr0 = P.wood_row0[ol];
    pre_chain = P.wood_mask[ol];
  }
  double we_mu = 0.0;
  if (is_s) {
    const int src = kPre ? pre_src : P.mu_src[my_c];
    if (src >= 0) we_mu = sTask[src];
    else if (kCom) {
      const int t = (-1 - src) / 3, r = (-1 - src) % 3;
      we_mu = P.com_cost[t][r] * (-P.com_gain[t] * (sCom[P.robot_root * 4 + r] - sTgt[P.n_frame * 7 + t * 3 + r]));
    }
  }
  if (is_dof) sDof[lane * 10 + 9] = dsq;
  wave_sync();
  // ---- Jacobian columns by (task, dof) PAIR lanes — 56 pairs for G1's four tasks, one pass — into the row-major array
  // Jh[r][k] = weighted_jacobian/√Dg (stride NR).  A pass keeps its entries in registers until every lane has read its
  // task block and dof axes: in the compact layout the rows overwrite both.
  const int n_jp = P.n_jpairs;
  for (int base = 0; base < n_jp || base == 0; base += kWave) {
    const int pi = base + lane;
    double Jo[6] = {0, 0, 0, 0, 0, 0};
    int rowmask = 0, o0 = 0;
    if (pi < n_jp) {
      const int t = (kPre && base == 0) ? pre_t : P.jpair_task[pi], k = (kPre && base == 0) ? pre_k : P.jpair_dof[pi];
      const auto& ft = frames[t];
      const double* o = sTask + t * 64;
      const double* dd = sDof + k * 10;
      const V3 d_ang{dd[0], dd[1], dd[2]}, d_lin{dd[3], dd[4], dd[5]}, d_anchor{dd[6], dd[7], dd[8]};
      const double dsk = dd[9];
      const V3 pf{o[27], o[28], o[29]};
      const V3 jp = d_lin + cross(d_ang, pf - d_anchor);
      M3 Rf;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rf.m[i] = o[18 + i];
      const V3 a = mulT(Rf, jp), w = mulT(Rf, d_ang);       // body-frame Jacobian (configuration.py:148-153)
      // J = −jlog(T_tb)·ᴮJ with jlog = [[J, −J·Q·J],[0, J]]:  y = J·w;  rows 0-2 = −J·(a − Q·y), rows 3-5 = −y
      const double wv[3] = {w.x, w.y, w.z}, av[3] = {a.x, a.y, a.z};
      double y[3], z3[3], Jt[6];
#pragma unroll
      for (int r = 0; r < 3; ++r) y[r] = o[3 * r] * wv[0] + o[3 * r + 1] * wv[1] + o[3 * r + 2] * wv[2];
#pragma unroll
      for (int r = 0; r < 3; ++r)
        z3[r] = av[r] - (o[9 + 3 * r] * y[0] + o[9 + 3 * r + 1] * y[1] + o[9 + 3 * r + 2] * y[2]);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        Jt[r] = -(o[3 * r] * z3[0] + o[3 * r + 1] * z3[1] + o[3 * r + 2] * z3[2]);
        Jt[3 + r] = -y[r];
      }
      rowmask = ft.rowmask;
      o0 = ft.jrow0 * NR + k;
#pragma unroll
      for (int r = 0; r < 6; ++r) Jo[r] = (ft.cost[r] * Jt[r]) * dsk;                // weighted_jacobian/√Dg
    }
    if (base == 0) {
      wave_sync();                                               // every pair of a one-pass problem has read its inputs
      for (int i = lane; i < n_mu * NR; i += kWave) sJ[i] = 0.0;   // dofs off a task's chain
      // x after the closed-form dof sweeps: z_k = −c_k/Dg_k (posture part of c only), staged as z_k·√Dg_k in row n_μ of
      // the array so that the right-hand side Jw·z is one more row of the product below
      // (a predicted dof contributes its bound: z̃_k = β_k)
      if (lane < NR) sJ[n_mu * NR + lane] = is_dof ? (clamped ? beta * hdiag_base * dsq : -c_lane * dsq) : 0.0;
      wave_sync();
    }
    {
      int c = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
        if ((rowmask >> r) & 1) { sJ[o0 + c * NR] = Jo[r]; ++c; }                     // (nonzero-cost rows only)
    }
  }
  // ComTask rows: CoM Jacobian column of this dof (mj_jacSubtreeCom closed form), weighted / √Dg
  if (kCom && is_dof) {
    const int d_body = dof_i[DI_BODY * 64 + ol];
    const double* cd = sCom + d_body * 4;
    V3 jc{0, 0, 0};
    if (body_i[BI_IN_ROBOT * 64 + d_body]) {
      const double* dd = sDof + lane * 10;
      const V3 d_ang{dd[0], dd[1], dd[2]}, d_lin{dd[3], dd[4], dd[5]}, d_anchor{dd[6], dd[7], dd[8]};
      const double fac = cd[3] * fast_rcp(sCom[P.robot_root * 4 + 3]);
      jc = fac * (d_lin + cross(d_ang, V3{cd[0], cd[1], cd[2]} - d_anchor));
    }
    const double jv[3] = {jc.x, jc.y, jc.z};
    for (int t = 0; t < P.n_com; ++t) {
      int c = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if ((P.com_rowmask[t] >> r) & 1) { sJ[(P.com_jrow0[t] + c) * NR + lane] = (P.com_cost[t][r] * jv[r]) * dsq; ++c; }
    }
  }
  wave_sync();
  if (prof) prof[0] = __builtin_readcyclecounter();                    // Jacobian rows staged
#if defined(MKH_WOOD_CALL) && defined(MKH_CLOCKS)
  if (prof && (int)prof[-2] == 6) return WoodOut{0.0, 0.0, 0.0, 0.0, 1.0, 0, 0};   // phase-stop 6 (prof = slot 17 of the problem's row)
#endif
  // ---- half-space rows (round 6).  With all dofs swept the tableau holds T_rd = A·H⁻¹ and T_rr = −A·H⁻¹·Aᵀ in the rows' block; the
  // direct start gets there by nv single pivots over nv + rows tableau rows (43 for a G1: 40–47 % of the collision / plugin
  // workloads).  With Ah = A·σ (σ = 1/√Dg, column scaling) and Y = Ah·Zᵀ:
  //     T_rd[s][j] = σ_j·(Ah[s][j] − Σ_r Y[s][r]·Z[r][j]/d_r),      T_rr[s][t] = −Ah[s]·Ah[t] + Σ_r Y[s][r]·Y[t][r]/d_r,
  // i.e. with the rows of Z extended by Z̃[r][nv + s] = −Y[s][r] the SAME n_μ rank-1 updates R[i][j] += Z̃[r][i]·Z̃[r][j]/d_r that
  // build the dof block build the whole tableau on top of R_rd = Ah, R_rr = −Ah·Ahᵀ (lazy scale 1 on a row index).  And
  // Y[s][·] = L⁻¹·(Jh·Ah[s]ᵀ): a half-space row is one more COLUMN of the elimination — lane nv + s carries −Jh·Ah[s]ᵀ next to
  // the dof lanes' columns of Jh and gets −Y back, its Σ z²/d (the diagonal) and Σ z·ω/d (the slack: A·x0 = A·z⁰ + Σ z̃·ω/d).
  // Lane nv + s walks the dofs of its row (the kinematic chains of the contact's two bodies): Jh·Ah[s]ᵀ and the right-hand side
  // A·z⁰ against the rows of Jh, Ah[s]·Ah[t] against the other rows; the Gram entries wait in the row's slot of the contact
  // table (normal and witness points are dead once the rows exist), at most 8 rows.
  double row_a0 = 0.0, row_gs = 0.0, row_h = 0.0, row_n = 1.0;
  const bool is_row = kRowsW && lane >= nv && lane < nv + nrows;
  if constexpr (kRowsW) {
    if (nrows > 0) {
      const int AS = a_stride_for(nv);
      double* const sA = smem + L.A;
      double* const sCol = smem + L.col;
      if (is_row) {
        const int sr = lane - nv;
        double* const col = sCol + sr * 16;
        row_h = col[9];
        uint64_t m = (MKH_FEAT & F_COLL) ? ((uint64_t)__double_as_longlong(col[10]) | (uint64_t)__double_as_longlong(col[11])) : ~0ull;
        if (nv < 64) m &= (1ull << nv) - 1ull;
        double accC[kMu + 1], accG[8], nn = 0.0;
#pragma unroll
        for (int r = 0; r <= kMu; ++r) accC[r] = 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t) accG[t] = 0.0;
        const double* const a_row = sA + sr * AS;
        for (; m; m &= m - 1) {
          const int k = __ffsll((unsigned long long)m) - 1;
          const double a = a_row[k], dk = sDof[k * 10 + 9];
          const double ah = a * dk, ahk = ah * dk;
          nn = fma(a, a, nn);
          row_gs = fma(ah, ah, row_gs);
          // (rows past the right-hand side / past the last half-space row: whatever lies behind them in this wave's LDS, never used)
          const double* const jk = sJ + k;
#pragma unroll
          for (int r = 0; r <= kMu; ++r) accC[r] = fma(ah, jk[r * NR], accC[r]);
          const double* const ak = sA + k;
#pragma unroll
          for (int t = 0; t < 8; ++t) accG[t] = fma(ahk, ak[t * AS], accG[t]);
        }
        row_n = sqrt(nn);
#pragma unroll
        for (int r = 0; r <= kMu; ++r) {
          if (r < n_mu) sJ[r * NR + lane] = -accC[r];
          row_a0 = (r == n_mu) ? accC[r] : row_a0;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) col[t] = accG[t];
      }
      wave_sync();
      if (is_dof)                                                      // the tableau's entries are Ah: scale the staged rows in place
        for (int sr = 0; sr < nrows; ++sr) sA[sr * AS + lane] *= dsq;
      wave_sync();
    }
  }
  // ---- S = I + Jh·Jhᵀ and Jw·z by (column, row-chunk) lanes: 64/n_μ chunks of rows per column, each lane a handful of
  // dot products on its own LDS addresses.  A row of Jh is nonzero only on the kinematic chain of its task (12–16 of the
  // 43 dofs on G1): the dot products walk the set bits of the column's chain mask instead of all NR dofs.
  // [column c][SP]: S[·][c], then [c]: (Jw·z)[c] − weighted error.  Lives in the dof stash (+ the subtree CoMs behind it:
  // axes / anchors / CoMs of the Jacobian columns are dead by now) when it fits, so that the low-rank start costs no LDS.
  double* const sS = (!kRowsW && wood_s_aliases_dof(nv, n_mu, SP, P.n_com > 0 ? P.nbody : 0)) ? sDof : smem + L.S;
  double* const sW = sS + n_mu * SP;
  for (int i = lane; i < n_mu * SP; i += kWave) sS[i] = 0.0;
  wave_sync();
  if constexpr (kCom) {
    // Dense product (F_COM builds: ComTask rows reach every dof, and with 24 rows the chain walk below — a dependent
    // ffs → address → LDS read per bit, 43 bits × 2 passes for a CoM column — took 39 k cycles): lane c accumulates column c
    // of Jh·Jhᵀ and (Jw·z)[c] over the dofs; column k of Jh arrives as two 16-lane planes, broadcast by the DPP network.
#ifdef MKH_WOOD_SPLIT
    wood_s_dense_call(n_mu, nv, (int)(sJ - smem), (int)(sS - smem), SP, (unsigned)a_mask, (unsigned)(a_mask >> 32));
    asm volatile("" : "+v"(lane));
#else
    double acc[kMuBig];
#pragma unroll
    for (int r = 0; r < kMuBig; ++r) acc[r] = 0.0;
    double wacc = 0.0;
    const int rows0 = (int)(lane & 15), rows1 = 16 + (int)(lane & 15);
    const bool has0 = rows0 < n_mu, has1 = rows1 < n_mu, mine = lane < n_mu;
    const double* c0 = sJ + (has0 ? rows0 : 0) * NR;
    const double* c1 = sJ + (has1 ? rows1 : 0) * NR;
    const double* cm = sJ + (mine ? lane : 0) * NR;
    const double* zs = sJ + n_mu * NR;
    for (int k = 0; k < nv; ++k) {
      const double p0 = has0 ? c0[k] : 0.0, p1 = has1 ? c1[k] : 0.0;
      const double g = mine ? cm[k] : 0.0;
      WoodAll<kMuBig>::step(acc, p0, p1, ((a_mask >> k) & 1) ? 0.0 : g);   // acc[r] += Jh[r][k]·Jh[c][k], free dofs only
      wacc = fma(g, zs[k], wacc);
    }
    if (mine) {
#pragma unroll
      for (int r = 0; r < kMuBig; ++r)
        if (r < n_mu) sS[lane * SP + r] = acc[r] + (r == lane ? 1.0 : 0.0);
      sW[lane] = wacc;
    }
#endif
  } else {
    const int wc = kPre ? pre_wc : P.wood_col[ol], wr0 = kPre ? pre_wr0 : P.wood_row0[ol];
    if (wc >= 0) {
      const double* a = sJ + wc * NR;
      const uint64_t chain = kPre ? pre_chain : P.wood_mask[ol];
      const int rpc = P.wood_rpc;
      // eight rows per pass (one pass for G1's 7 rows per lane): the lane's own column entry is read once per pass, and
      // the walk over the chain bits — a dependent ffs → address → LDS read → FMA chain per bit — runs once
      for (int i0 = 0; i0 < rpc; i0 += 8) {
        const int row0 = wr0 + i0;
        if (row0 > n_mu) break;
        // (the eight rows through ONE address and compile-time offsets j·NR: rows past the right-hand side are read — whatever
        //  lies behind the rows in this wave's LDS, or 0 past its end — and never stored; eight clamped row pointers cost
        //  eight address additions per chain bit, 261 of a G1 solve's 504 VALU instructions in this phase were such arithmetic)
        const double* const b0 = sJ + row0 * NR;
        double acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0;
        for (uint64_t mk = chain & ~a_mask; mk; mk &= mk - 1) {        // S: the free dofs of the chain
          const int k = __ffsll((unsigned long long)mk) - 1;
          const double av = a[k];
          const double* const bk = b0 + k;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fma(av, bk[j * NR], acc[j]);
        }
        if (chain & a_mask) {                                          // Jw·z̃ also takes the predicted dofs (z̃ = β)
          const int jr = n_mu - row0;                                  // the right-hand-side row of this pass, if any
          double extra = 0.0;
          for (uint64_t mk = chain & a_mask; mk; mk &= mk - 1) {
            const int k = __ffsll((unsigned long long)mk) - 1;
            extra = fma(a[k], sJ[n_mu * NR + k], extra);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (j == jr) ? extra : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = row0 + j;
          if (i0 + j < rpc && row <= n_mu) {
            if (row < n_mu) sS[wc * SP + row] = acc[j] + (row == wc ? 1.0 : 0.0); else sW[wc] = acc[j];
          }
        }
      }
    }
  }
  wave_sync();
  if (is_s) sW[my_c] -= we_mu;                                         // w = Jw·z − r
  wave_sync();
  if (prof) prof[1] = __builtin_readcyclecounter();                    // S and w ready
#if defined(MKH_WOOD_CALL) && defined(MKH_CLOCKS)
  if (prof && (int)prof[-2] == 7) return WoodOut{0.0, 0.0, 0.0, 0.0, 1.0, 0, 0};   // phase-stop 7
#endif
  // ---- elimination (K = the smallest compiled row capacity that holds n_μ)
  double ssq = 0.0, quad = 0.0, zw = 0.0;
  // (the 24-row and two-register-set instantiations only exist in the F_COM variants, which the host also picks for
  //  problems without a ComTask that need them: 48 + 48 column registers cost the lean variant its spill-free build)
  // (cold start only: a warm start brings its own prediction)
  // (compiled into the lean 44- / 48-row builds only: its code costs the others — register pressure of this function, scalar
  //  spills — more than the pivots it saves: G1 +1.5 %, but H1 / hands / quadrupeds −10 % and the G1 full example −4.5 %)
  constexpr bool kRefine = !kCom && NR >= 44 && !(MKH_FEAT & F_STEPS) && !kRowsW;
  const WoodRefine rf{kRefine && a_mask == 0 && !dual && P.wood_refine != 0, nv, lo, hi, -c_lane * (dsq * dsq), dsq, hdiag_base * dsq, -c_lane * dsq};
  if constexpr (kCom) {
#ifdef MKH_WOOD_SPLIT
    {
      const int oJ = (int)(sJ - smem), oS = (int)(sS - smem), oR = (int)(sRow - smem);
      const WoodElimOut eo = dual ? wood_eliminate_call<kMuBig, true>(n_mu, oJ, oS, SP, oR)
                                  : (n_mu <= kMu ? wood_eliminate_call<kMu, false>(n_mu, oJ, oS, SP, oR)
                                                 : wood_eliminate_call<kMuBig, false>(n_mu, oJ, oS, SP, oR));
      asm volatile("" : "+v"(lane));
      ssq = eo.ssq; quad = eo.quad; zw = eo.zw; status |= eo.status;
    }
#else
    if (dual) wood_eliminate<kMuBig, true>(n_mu, NR, lane, sJ, sS, SP, sW, sRow, sDinv, ssq, quad, zw, status, rf, clamp, beta);
    else if (n_mu <= kMu) wood_eliminate<kMu, false>(n_mu, NR, lane, sJ, sS, SP, sW, sRow, sDinv, ssq, quad, zw, status, rf, clamp, beta);
    else wood_eliminate<kMuBig, false>(n_mu, NR, lane, sJ, sS, SP, sW, sRow, sDinv, ssq, quad, zw, status, rf, clamp, beta);
#endif
  } else {
    if constexpr (kRowsW) {      // (tableau lanes [0, NR) + n_μ S lanes exceed the wavefront from NR = 48, n_μ = 17 on: two passes)
      if (dual) wood_eliminate<kMu, true>(n_mu, NR, lane, sJ, sS, SP, sW, sRow, sDinv, ssq, quad, zw, status, rf, clamp, beta);
      else wood_eliminate<kMu, false>(n_mu, NR, lane, sJ, sS, SP, sW, sRow, sDinv, ssq, quad, zw, status, rf, clamp, beta);
    } else {
      wood_eliminate<kMu, false>(n_mu, NR, lane, sJ, sS, SP, sW, sRow, sDinv, ssq, quad, zw, status, rf, clamp, beta);
    }
  }
  wave_sync();
  WoodOut wo;
  wo.hdiag = hdiag_base * (1.0 + ssq);                                 // H[k][k] = Dg·(1 + Σ Jh²)  (only scales thresholds)
  wo.dsq = dsq; wo.status = __ballot(status != 0) ? 4 : 0;
  // free dof: −H⁻¹[j][j] and x;  predicted dof (not swept): its diagonal of the Schur complement and its gradient
  const bool on_bound = is_dof && clamp != 0;                          // (predicted by the caller, or by the refinement pass)
  wo.clamp = on_bound ? clamp : 0;
  wo.D = on_bound ? hdiag_base * (1.0 + quad) : (dsq * dsq) * (quad - 1.0);
  wo.x = on_bound ? fma(hdiag_base, beta, c_lane) + hdiag_base * dsq * zw : -c_lane * (dsq * dsq) - dsq * zw;
  wo.rown = 1.0;
  if constexpr (kRowsW) {
    if (is_row) { wo.D = quad - row_gs; wo.x = (row_a0 + zw) - row_h; wo.rown = row_n; }   // T_rr[s][s], the slack A·x0 − h, ‖A[s]‖
  }
  return wo;
}
#else
struct WoodOut { double hdiag, dsq, x, D, rown; int status, clamp; };
__device__ __forceinline__ WoodOut wood_start(const DeviceProblem*, int, int, int, double, double, int, double, double, double, int, long long* = nullptr) { return WoodOut{0.0, 0.0, 0.0, 0.0, 1.0, 0, 0}; }
#endif
// ---------------------------------------------------------------- collision half-space rows (CollisionAvoidanceLimit)
// mode 0: the contacts of this problem's geom pairs (mj_geomDistance, collision_avoidance_limit.py:187-229) become half-space
// rows: sCol[slot] = {n, from, to, h, dof chains, pair}, sA[slot][·] = −nᵀ(jacp₂(to) − jacp₁(from)); returns
// nrows | status bits << 8 | rows_dropped << 16.  mode 1 (after the QP, only when rows were dropped): does a contact that
// found no tableau row hold at the solution Δq (sDof slot 9)?  Returns 1 when one does not (MKH_ST_ROW_OVERFLOW).
// Both calls run while the tableau is dead — the rows are evaluated BEFORE the H accumulation — so in the builds where the
// distance routines do not fit next to the kernel's own values (general convex pairs, the all-feature builds, the
// one-more-wave register maps) this is a real call and owns the whole register file (see pre_phases).
#if (MKH_FEAT & 8)
#if defined(MKH_CALLS) || defined(MKH_FORCE_COLL_CALL)
#define MKH_COLL_CALL 1
#define MKH_COLL_ATTR __attribute__((noinline))
#else
#define MKH_COLL_ATTR __forceinline__
#endif
// One general convex pair whose cores overlap, the wavefront cooperating (geom_overlap_distance → cvx_epa).  A REAL call in
// every build: the path is rare (a pair in penetration) and its registers must not count against the phase around it; the
// pair's descriptor and the body poses arrive as generic pointers.
struct OverlapOut { double dist; V3 from, to; };
__device__ __attribute__((noinline)) OverlapOut overlap_pair(const CollisionPairDev* cpp, const double* sX, int XS, double* ws, double tol) {
  tol = uni(tol);
  cpp = reinterpret_cast<const CollisionPairDev*>(uni((unsigned long long)reinterpret_cast<size_t>(cpp)));
  sX = reinterpret_cast<const double*>(uni((unsigned long long)reinterpret_cast<size_t>(sX)));
  ws = reinterpret_cast<double*>(uni((unsigned long long)reinterpret_cast<size_t>(ws)));
  XS = uni(XS);
  const CollisionPairDev& cp = *cpp;
  const double* x1 = sX + cp.body1;
  const double* x2 = sX + cp.body2;
  const Q4 bq1{x1[3 * XS], x1[4 * XS], x1[5 * XS], x1[6 * XS]}, bq2{x2[3 * XS], x2[4 * XS], x2[5 * XS], x2[6 * XS]};
  const V3 gp1 = V3{x1[0], x1[XS], x1[2 * XS]} + qrot(bq1, V3{cp.lpos1[0], cp.lpos1[1], cp.lpos1[2]});
  const V3 gp2 = V3{x2[0], x2[XS], x2[2 * XS]} + qrot(bq2, V3{cp.lpos2[0], cp.lpos2[1], cp.lpos2[2]});
  const Q4 gq1 = qmul(bq1, Q4{cp.lquat1[0], cp.lquat1[1], cp.lquat1[2], cp.lquat1[3]});
  const Q4 gq2 = qmul(bq2, Q4{cp.lquat2[0], cp.lquat2[1], cp.lquat2[2], cp.lquat2[3]});
  OverlapOut o;
  geom_overlap_distance(cp.type1, V3{cp.size1[0], cp.size1[1], cp.size1[2]}, gp1, gq1, cp.type2,
                        V3{cp.size2[0], cp.size2[1], cp.size2[2]}, gp2, gq2, o.dist, o.from, o.to, cp.vert1, cp.nvert1, cp.vert2, cp.nvert2, ws, tol);
  return o;
}
__device__ MKH_COLL_ATTR int collision_phase(const DeviceProblem* Pq, const TapArgs* tp, int pb, double dt, int mode,
                                              const LdsLayout& Lk) {
  constexpr int NT = MKH_NT, FEAT = MKH_FEAT;
  constexpr bool kTaps = (FEAT & F_TAPS) != 0, kWood = false;
  constexpr bool kSimpleColl = (FEAT & F_SIMPLE_COLL) != 0;     // every collision pair is plane / sphere / capsule
  // pairs without an analytic routine (convex_dev.h): their own collision variants, and the all-feature ones
  constexpr bool kConvexColl = (FEAT & F_CONVEX_COLL) != 0 || (FEAT & (F_ALL & ~F_TAPS)) == (F_ALL & ~F_TAPS);
  extern __shared__ __attribute__((aligned(16))) double smem[];
#ifdef MKH_COLL_CALL
  pb = uni(pb); mode = uni(mode); dt = uni(dt);
  Pq = reinterpret_cast<const DeviceProblem*>(uni((unsigned long long)reinterpret_cast<size_t>(Pq)));
  tp = reinterpret_cast<const TapArgs*>(uni((unsigned long long)reinterpret_cast<size_t>(tp)));
  const MKH_CONSTANT DeviceProblem& P = *(const MKH_CONSTANT DeviceProblem*)Pq;
  const MKH_GLOBAL CollisionPairDev* const pairs = (const MKH_GLOBAL CollisionPairDev*)P.pairs;
  const MKH_GLOBAL PairCull* const cull = (const MKH_GLOBAL PairCull*)P.cull;
#else
  const DeviceProblem& P = *Pq;
  const CollisionPairDev* const pairs = P.pairs;
  const PairCull* const cull = P.cull;
#endif
  const int lane = lane_id();
  const int nv = P.nv, n_pairs = P.n_pairs, max_rows = P.max_rows;
#ifdef MKH_COLL_CALL
  const LdsLayout L = kernel_lds_layout(P);                  // (a reference argument would travel through scratch)
  (void)Lk;
#else
  const LdsLayout& L = Lk;
#endif
  const double* const sX = smem + L.X;
  const int XS = lds_even(P.nbody);
  const double* const sDof = smem + L.dof;
  double* const sCol = smem + L.col;
  double* const sA = smem + L.A;
  const int AS = a_stride_for(nv);
  const bool is_dof = lane < nv;
  const double kInf = __builtin_huge_val();
  // More detected contacts than tableau rows (64 − nv): the reference hands every row to quadprog
  // (collision_avoidance_limit.py:187-210); here the max_rows TIGHTEST (smallest h, ties by pair index) become rows and
  // the solution is checked against the rest after the QP — a dropped row that holds at the solution was inactive, so
  // the result is the reference's; one that does not hold sets MKH_ST_ROW_OVERFLOW.
  double* const sH = smem + L.hsel;                         // h of every candidate pair (only laid out when n_pairs > max_rows)
  const bool can_select = n_pairs > max_rows;
  // More than one wavefront of pairs (the reference's ALOHA example: 1 104): a cull pass over bounding spheres first — a pair whose
  // centres are farther apart than rbound₁ + rbound₂ + detection distance is beyond that distance, and that is all mj_geomDistance
  // says about it — and the distance routines run over the COMPACTED list of the pairs that are left, 64 at a time (17 trips of the
  // routines with a few busy lanes each were 48 % of an ALOHA solve).  The list is ascending in the pair index, so rows, ranks
  // and ties are what they were.  (The capsule-only builds keep the plain loop: their pair sets fit one trip.)
  constexpr bool kCull = !kSimpleColl;
  const bool use_cull = kCull && cull != nullptr;
  unsigned short* const sList = reinterpret_cast<unsigned short*>(smem + L.hsel + (can_select ? lds_even(n_pairs) : 0));
  // the expanding polytope's workspace (general convex pairs whose cores overlap): behind the h of every pair and the list
  double* const sEpa = smem + L.hsel + (can_select ? lds_even(n_pairs) : 0) + (use_cull ? lds_even((n_pairs + 3) / 4) : 0);
  int n_cand = n_pairs;
  if (use_cull) {
    n_cand = 0;
    for (int base = 0; base < n_pairs; base += 64) {
      const int pi = base + lane;
      bool keep = false;
      if (pi < n_pairs) {
        const auto& c = cull[pi];
        const double* x1 = sX + c.body1;
        const double* x2 = sX + c.body2;
        const Q4 bq1{x1[3 * XS], x1[4 * XS], x1[5 * XS], x1[6 * XS]}, bq2{x2[3 * XS], x2[4 * XS], x2[5 * XS], x2[6 * XS]};
        const V3 d = (V3{x2[0], x2[XS], x2[2 * XS]} + qrot(bq2, V3{c.lpos2[0], c.lpos2[1], c.lpos2[2]})) -
                     (V3{x1[0], x1[XS], x1[2 * XS]} + qrot(bq1, V3{c.lpos1[0], c.lpos1[1], c.lpos1[2]}));
        keep = !(dot(d, d) > c.reach2);
        if (!keep && mode == 0 && MKH_TAP(t_coll_h)) MKH_TAP(t_coll_h)[(size_t)pb * n_pairs + pi] = kInf;
      }
      const unsigned long long km = __ballot(keep);
      if (keep) sList[n_cand + __popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)pi;
      n_cand += __popcll(km);
    }
    wave_sync();
  }
  // world poses of the two geoms of pair pi
  auto pair_poses = [&](const auto& cp, V3& gp1, Q4& gq1, V3& gp2, Q4& gq2) {
    const double* x1 = sX + cp.body1;
    const double* x2 = sX + cp.body2;
    const Q4 bq1{x1[3 * XS], x1[4 * XS], x1[5 * XS], x1[6 * XS]}, bq2{x2[3 * XS], x2[4 * XS], x2[5 * XS], x2[6 * XS]};
    gp1 = V3{x1[0], x1[XS], x1[2 * XS]} + qrot(bq1, V3{cp.lpos1[0], cp.lpos1[1], cp.lpos1[2]});
    gp2 = V3{x2[0], x2[XS], x2[2 * XS]} + qrot(bq2, V3{cp.lpos2[0], cp.lpos2[1], cp.lpos2[2]});
    gq1 = qmul(bq1, Q4{cp.lquat1[0], cp.lquat1[1], cp.lquat1[2], cp.lquat1[3]});
    gq2 = qmul(bq2, Q4{cp.lquat2[0], cp.lquat2[1], cp.lquat2[2], cp.lquat2[3]});
  };
  // a pair's distance and witness points → active, h, unit normal, dof chains
  auto finish_contact = [&](const auto& cp, double dist, V3 from, V3 to, double& hk, V3& nrm, uint64_t& m1, uint64_t& m2) -> bool {
    const bool active = dist != cp.ddetect;                  // Contact.inactive (:52-56)
    hk = kInf;
    if (active) {
      hk = (dist > cp.dmin) ? (cp.gain * (dist - cp.dmin) / dt) + cp.relax : cp.relax;  // :200-205
      nrm = to - from;                                       // Contact.normal (:46-50)
      const double nn = sqrt(dot(nrm, nrm));
      nrm = (nn < 1e-15) ? V3{1.0, 0.0, 0.0} : fast_rcp(nn) * nrm;
      m1 = cp.mask1;
      m2 = cp.mask2;
    }
    return active;
  };
  // contact of pair pi at the current poses.  need_epa: a general convex pair whose cores overlap — finished at wave level
  // below (overlap_of), the values returned here are placeholders
  auto contact_of = [&](int pi, double& hk, V3& nrm, V3& from, V3& to, uint64_t& m1, uint64_t& m2, bool& need_epa, double* gjk_slot) -> bool {
    const auto& cp = pairs[pi];
    if constexpr (!kSimpleColl && !kConvexColl) {
      // a pair without an analytic routine in the ANALYTIC build: its contact was evaluated by convex_contacts_kernel in front of
      // this launch (the general convex routine — GJK, the expanding polytope — owns a kernel of its own there: no callee-saved
      // register blocks in scratch, which were 284 x the algorithmic bytes of `ur5e_convex`)
      if (cp.cv_slot >= 0) {
        const double* r = P.cv_contacts + ((size_t)pb * P.n_cv + cp.cv_slot) * 7;
        from = V3{r[1], r[2], r[3]}; to = V3{r[4], r[5], r[6]};
        return finish_contact(cp, r[0], from, to, hk, nrm, m1, m2);
      }
    }
    V3 gp1, gp2;
    Q4 gq1, gq2;
    pair_poses(cp, gp1, gq1, gp2, gq2);
    double dist;
    geom_distance<kSimpleColl, kConvexColl>(cp.type1, V3{cp.size1[0], cp.size1[1], cp.size1[2]}, gp1, gq1, cp.type2,
                  V3{cp.size2[0], cp.size2[1], cp.size2[2]}, gp2, gq2, cp.ddetect, dist, from, to,
                  cp.vert1, cp.nvert1, cp.vert2, cp.nvert2, &need_epa, gjk_slot);
    return finish_contact(cp, dist, from, to, hk, nrm, m1, m2);
  };
  // the same for ONE wave-uniform pair whose cores overlap, every lane cooperating (expanding polytope: overlap_pair above)
  // (mine: this lane is the pair's — it alone polishes the witness points, collide_dev.h geom_overlap_polish)
  auto overlap_of = [&](int pi, bool mine, double& hk, V3& nrm, V3& from, V3& to, uint64_t& m1, uint64_t& m2) -> bool {
    // (loose polytope + polish; without a certificate the tight polytope, whose answer stands if the polish has none for it either)
    double dist = 0.0;
    bool certified = false;
#pragma nounroll
    for (int pass = geom_overlap_loose(pairs[pi].type1, pairs[pi].type2) ? 0 : 1; pass < 2 && !certified; ++pass) {
      const OverlapOut o = overlap_pair((const CollisionPairDev*)(pairs + pi), sX, XS, sEpa, pass ? kEpaTol : kLooseEpa);
      from = o.from; to = o.to;
      dist = o.dist;
      bool ok = false;
      if (mine) {
        const auto& cp = pairs[pi];
        V3 gp1, gp2;
        Q4 gq1, gq2;
        pair_poses(cp, gp1, gq1, gp2, gq2);
        ok = geom_overlap_polish(cp.type1, V3{cp.size1[0], cp.size1[1], cp.size1[2]}, gp1, gq1, cp.type2, V3{cp.size2[0], cp.size2[1], cp.size2[2]}, gp2, gq2, dist, from, to, pass != 0);
      }
      certified = __ballot(ok) != 0;
    }
    return finish_contact(pairs[pi], dist, from, to, hk, nrm, m1, m2);
  };
  // position of pair pi in the order (h, index) among all pairs (h = +inf: not detected)
  auto rank_of = [&](int k, double hk) -> int {            // (k: position in the candidate list, which is ascending in the pair index)
    int rank = 0;
    for (int j = 0; j < n_cand; ++j) {
      const double hj = sH[j];
      rank += (hj < hk || (hj == hk && j < k)) ? 1 : 0;
    }
    return rank;
  };
  // ONE call site of contact_of for both modes (the distance routines — the general convex one above all — are inlined once)
  int nrows = 0, status = 0;
  bool rows_dropped = false, viol = false;
  for (int pass = 0; pass < 2; ++pass) {
    nrows = 0;
    for (int base = 0; base < n_cand; base += 64) {
      const int kc = base + lane;                            // candidate, and the pair it stands for
      const bool in_list = kc < n_cand;
      const int pi = !in_list ? n_pairs : (use_cull ? (int)sList[kc] : kc);
      bool active = false;
      double hk = kInf;
      V3 nrm{1, 0, 0}, from{0, 0, 0}, to{0, 0, 0};
      uint64_t m1 = 0, m2 = 0;
      bool want = in_list;
      if (mode != 0 && want) {                               // mode 1: only the contacts that found no tableau row
        const double hs = sH[kc];
        want = hs < kInf && rank_of(kc, hs) >= max_rows;
      }
      bool need_epa = false;
      if constexpr (kConvexColl) {
        // GJK keeps its simplex in LDS (the front of the expanding polytope's workspace): kGjkSlots lanes at a time
        const bool gjk = want && geom_pair_runs_gjk(pairs[pi].type1, pairs[pi].type2);
        const unsigned long long gm = __ballot(gjk);
        const int rank = __popcll(gm & ((1ull << lane) - 1ull));
        for (int b0 = 0; b0 == 0 || b0 < __popcll(gm); b0 += kGjkSlots) {
          const bool mine = want && (gjk ? (rank >= b0 && rank < b0 + kGjkSlots) : b0 == 0);
          if (mine) active = contact_of(pi, hk, nrm, from, to, m1, m2, need_epa, sEpa + (rank - b0));
        }
        wave_sync();
      } else {
        if (want) active = contact_of(pi, hk, nrm, from, to, m1, m2, need_epa, nullptr);
      }
      if constexpr (kConvexColl) {
        // pairs whose cores overlap: one at a time, the wavefront cooperating on the expanding polytope
        for (unsigned long long em = __ballot(want && need_epa); em; em &= em - 1) {
          const int l = (int)__builtin_ctzll(em);
          double hk_e = kInf;
          V3 n_e{1, 0, 0}, f_e{0, 0, 0}, t_e{0, 0, 0};
          uint64_t m1_e = 0, m2_e = 0;
          const bool act_e = overlap_of(use_cull ? (int)sList[base + l] : base + l, lane == l, hk_e, n_e, f_e, t_e, m1_e, m2_e);
          if (lane == l) { active = act_e; hk = hk_e; nrm = n_e; from = f_e; to = t_e; m1 = m1_e; m2 = m2_e; }
        }
      }
      if (mode != 0) {
        // G·Δq ≤ h at the solution?  (G·Δq = −nᵀ(ṗ₂(to) − ṗ₁(from)) for joint displacements Δq, the dofs of each geom's chain)
        if (active) {
          V3 vel{0, 0, 0};
          for (uint64_t mm = m1 | m2; mm; mm &= mm - 1) {
            const int d = __builtin_ctzll(mm);
            const double* dd = sDof + d * 10;
            const V3 a_ang{dd[0], dd[1], dd[2]}, a_lin{dd[3], dd[4], dd[5]}, a_anchor{dd[6], dd[7], dd[8]};
            const double dq = dd[9];
            if ((m2 >> d) & 1) vel = vel + dq * (a_lin + cross(a_ang, to - a_anchor));
            if ((m1 >> d) & 1) vel = vel - dq * (a_lin + cross(a_ang, from - a_anchor));
          }
          viol = viol || (-dot(nrm, vel) > hk + 1e-9 * (1.0 + fabs(hk)));
        }
        continue;
      }
      if (in_list) {
        if (pass == 0) {
          if (can_select) sH[kc] = hk;
          if (MKH_TAP(t_coll_h)) MKH_TAP(t_coll_h)[(size_t)pb * n_pairs + pi] = hk;
          if (MKH_TAP(t_coll_G) && active) {
            // the tap holds the row of EVERY detected contact (Limit.compute_qp_inequalities / build_ik return them all),
            // also of those that find no tableau row below: the pair lane walks its two dof chains itself
            double* g = MKH_TAP(t_coll_G) + ((size_t)pb * n_pairs + pi) * nv;
            for (uint64_t mm = m1 | m2; mm; mm &= mm - 1) {
              const int d = __builtin_ctzll(mm);
              const double* dd = sDof + d * 10;
              const V3 a_ang{dd[0], dd[1], dd[2]}, a_lin{dd[3], dd[4], dd[5]}, a_anchor{dd[6], dd[7], dd[8]};
              V3 dj{0, 0, 0};
              if ((m2 >> d) & 1) dj = dj + a_lin + cross(a_ang, to - a_anchor);
              if ((m1 >> d) & 1) dj = dj - (a_lin + cross(a_ang, from - a_anchor));
              g[d] = -dot(nrm, dj);
            }
          }
        } else if (active) {
          active = rank_of(kc, hk) < max_rows;
        }
      }
      const unsigned long long am = __ballot(active);
      const int slot = nrows + __popcll(am & ((1ull << lane) - 1ull));
      if (active && slot < max_rows) {
        double* o = sCol + slot * 16;
        o[0] = nrm.x; o[1] = nrm.y; o[2] = nrm.z; o[3] = from.x; o[4] = from.y; o[5] = from.z;
        o[6] = to.x; o[7] = to.y; o[8] = to.z; o[9] = hk;
        o[10] = __longlong_as_double((long long)m1);
        o[11] = __longlong_as_double((long long)m2);
        o[12] = (double)pi;
      }
      nrows += __popcll(am);
    }
    if (mode != 0) return __ballot(viol) ? 1 : 0;
    if (pass == 0 && can_select && nrows > max_rows) { rows_dropped = true; wave_sync(); continue; }   // select, then refill
    break;
  }
  if (nrows > max_rows) { status |= 16; nrows = max_rows; }
  wave_sync();
  // G[s][lane] = −nᵀ(jacp₂(to) − jacp₁(from))   (compute_contact_normal_jacobian :59-72)
  // (two rows per trip: the rows are independent chains of ≈40 dependent fp64 operations)
  // (robots up to 32 dofs — the Shadow hand of BASELINE config 4: 24 — put a row on each HALF of the wavefront, four rows per trip:
  //  lane l works on dof l % 32 of row s + l / 32; round 6, the ≈ 17 rows of a Shadow instance took nine trips with 24 busy lanes)
  {
    const bool halves = AS <= 32;
    const int dl = halves ? (lane & 31) : lane, half = halves ? (lane >> 5) : 0, stride = halves ? 2 : 1;
    const bool dof_l = dl < nv;
    const double* ax = sDof + (dof_l ? dl : 0) * 10;
    const V3 d_ang{ax[0], ax[1], ax[2]}, d_lin{ax[3], ax[4], ax[5]}, d_anchor{ax[6], ax[7], ax[8]};
    auto row_entry = [&](int s) -> double {
      const double* o = sCol + s * 16;
      const uint64_t m1 = (uint64_t)__double_as_longlong(o[10]), m2 = (uint64_t)__double_as_longlong(o[11]);
      V3 n{o[0], o[1], o[2]};
      V3 dj{0, 0, 0};
      if ((m2 >> dl) & 1) dj = dj + d_lin + cross(d_ang, V3{o[6], o[7], o[8]} - d_anchor);
      if ((m1 >> dl) & 1) dj = dj - (d_lin + cross(d_ang, V3{o[3], o[4], o[5]} - d_anchor));
      const double a = dof_l ? -dot(n, dj) : 0.0;
      return a;
    };
    for (int s = 0; s < nrows; s += 2 * stride) {
      // (a row past the last one: evaluated on the last row's record, not stored)
      const int r0 = s + half, r1 = s + stride + half;
      const double a0 = row_entry(r0 < nrows ? r0 : nrows - 1), a1 = row_entry(r1 < nrows ? r1 : nrows - 1);
      if (dl < AS) {
        if (r0 < nrows) sA[r0 * AS + dl] = a0;
        if (r1 < nrows) sA[r1 * AS + dl] = a1;
      }
    }
  }
  (void)NT;
  return nrows | (status << 8) | ((rows_dropped ? 1 : 0) << 16);
}
#else
__device__ __forceinline__ int collision_phase(const DeviceProblem*, const TapArgs*, int, double, int, const LdsLayout&) { return 0; }
#endif

#ifdef MKH_W3
#define MKH_STAGE (2 * ((MKH_NT + 15) / 16))   // 3-waves maps: only the planes the column has (gen_tab_asm.py)
#else
#define MKH_STAGE 8   // staging VGPRs of the rank-1 update: four 16-lane planes of the pivot column (gen_tab_asm.py ntmp_for)
#endif
#if defined(MKH_W4)
#define MKH_WAVES 4
#define MKH_TOP 128
#elif defined(MKH_W3)
#define MKH_WAVES 3
#define MKH_TOP 168
#else
#define MKH_WAVES (MKH_NT <= 8 ? 4 : (MKH_NT <= 24 ? 3 : 2))   // resident waves per SIMD the register map is built for
#define MKH_TOP (MKH_NT <= 8 ? 128 : (MKH_NT <= 24 ? 168 : 256))  // VGPRs per lane at that occupancy (gen_tab_asm.py total_for)
#endif
#define MKH_CAP (MKH_TOP - 2 * MKH_NT - MKH_STAGE)
#ifdef MKH_CAP_PROBE     // (pressure probing only: the code is wrong when the cap reaches into the pinned range)
#undef MKH_CAP
#define MKH_CAP MKH_CAP_PROBE
#else
static_assert(MKH_TAB<MKH_NT>::kCompilerVgprs == MKH_CAP, "register map of tab_asm.inc changed");
#endif
__global__ __launch_bounds__(64, MKH_WAVES) __attribute__((amdgpu_num_vgpr(MKH_CAP / 2)))
void MKH_KERNEL_NAME(const DeviceProblem* __restrict__ Pg, const SolveArgs A_k, const TapArgs* __restrict__ tp) {
  // The call arguments are read from the kernarg segment WHERE they are used (as the descriptor is, see Pq below): as a
  // by-value struct hipcc loads all of it into s[12:27] at entry, spills that 16-register tuple, and reloads the whole
  // tuple — 16 v_readlane, VALU instructions — for every field it touches: ≈ 300 VALU instructions per problem.
  // (Low-rank variants only: G1 config 3 0.831 → 0.822 ms.  In the direct-start and all-feature builds, whose compiler
  //  budgets are tighter, the same change trades the scalar spills for vector ones — `8_0`: 0 → 20 spilled VGPRs, UR5e at
  //  4 096 instances −5 %; `64_30`: 447 → 534, the Shadow hand's fused loop 9.4 → 13.9 ms — so they keep the by-value struct.)
#if (MKH_FEAT & 32)
  struct KernArgs { const DeviceProblem* P; SolveArgs A; const TapArgs* tp; };
  const MKH_CONSTANT SolveArgs* A_p = (const MKH_CONSTANT SolveArgs*)((const MKH_CONSTANT char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                                      offsetof(KernArgs, A));
#ifndef MKH_ONE_SHOT
  asm volatile("" : "+s"(A_p));
#endif
  const MKH_CONSTANT SolveArgs& A = *A_p;
  (void)A_k;
#ifndef MKH_ONE_SHOT
#define MKH_ARGS_AT_USE() asm volatile("" : "+s"(A_p)); const MKH_CONSTANT SolveArgs& A = *A_p
#else
#define MKH_ARGS_AT_USE() do {} while (0)
#endif
#else
  const SolveArgs& A = A_k;
#define MKH_ARGS_AT_USE() do {} while (0)
#endif
  constexpr int NT = MKH_NT, FEAT = MKH_FEAT;
  constexpr bool kTaps = (FEAT & F_TAPS) != 0, kRel = (FEAT & F_REL) != 0, kCom = (FEAT & F_COM) != 0;
  constexpr bool kColl = (FEAT & F_COLL) != 0, kSteps = (FEAT & F_STEPS) != 0, kWood = (FEAT & F_WOOD) != 0;
  // plugin route (caller-defined Task / Limit subclasses as dense rows): only in the variants that carry every
  // feature (FEAT 30 / 31) — it is the general path, not a tuned one
  constexpr bool kDense = ((FEAT & F_DENSE) != 0 || (FEAT & (F_ALL & ~F_TAPS)) == (F_ALL & ~F_TAPS)) && !kWood;
  // half-space rows in the tableau: collision contacts and / or general rows of caller-defined limits
  constexpr bool kRows = kColl || (FEAT & F_DENSE) != 0;
  constexpr bool kSimpleColl = (FEAT & F_SIMPLE_COLL) != 0;     // every collision pair is plane / sphere / capsule
  // pairs without an analytic routine (convex_dev.h): their own collision variants, and the all-feature ones
  constexpr bool kConvexColl = (FEAT & F_CONVEX_COLL) != 0 || (FEAT & (F_ALL & ~F_TAPS)) == (F_ALL & ~F_TAPS);
#ifdef MKH_NR
  constexpr int NR = MKH_NR;                 // dof rows of the tableau (low-rank start: NR ≥ nv, NT ≥ nv + n_μ)
#else
  constexpr int NR = NT;
#endif
  static_assert(!kWood || !kRel, "low-rank start: frame / posture / CoM tasks, box limits and (round 6) half-space rows");
  constexpr bool kWoodRows = kWood && kRows;       // low-rank start with half-space rows (wood_start "half-space rows")
  static_assert(!kWoodRows || NR == NT, "the rows' columns of the elimination live in the tableau-wide rows of Jh");
#ifdef MKH_W3
  constexpr bool kCompact = true;            // LDS ranges aliased by phase (lds_layout): 12 waves per CU need ≤ 13.3 KB each
  static_assert(!kColl, "compact LDS layout: the collision phase reads the body poses after the Jacobian rows");
#else
  constexpr bool kCompact = false;
#endif
#ifdef MKH_ONE_SHOT
  // (the descriptor through the CONSTANT address space from the start: as a plain reference its sizes arrive by vector loads and the
  //  whole LDS layout is computed in VGPRs — per problem in this build, ≈ 100 VALU instructions and two addresses that end up spilled.
  //  The persistent builds compute the layout once per wavefront; there the scalar version only adds spilled SGPRs — measured.)
  const MKH_CONSTANT DeviceProblem& P0 = *(const MKH_CONSTANT DeviceProblem*)Pg;
#else
  const DeviceProblem& P0 = *Pg;
#endif
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int lane = lane_id();     // (re-laundered at every phase boundary, see MKH_TICK)
  const int nq = P0.nq, nv = P0.nv, nbody = P0.nbody;
  const LdsLayout L = kernel_lds_layout(P0);
  const bool prefetch = kernel_prefetch(P0);
  double* sq = smem + L.q;            // (sq / sTgt alternate between two buffers, see "load inputs")
  double* const sX = smem + L.X;
  const int XS = lds_even(nbody);                      // component stride of sX (consecutive lanes hit consecutive banks;
                                                       // body-major with stride 8 put 16 lanes on each bank)
  double* const sJnt = smem + L.jnt;
  double* sTgt = smem + L.tgt;
  double* const sTask = smem + L.task;
  double* const sJ = smem + L.J;
  double* const sDof = smem + L.dof;
  double* const sCom = smem + L.com;
  double* const sCol = smem + L.col;
  double* const sA = smem + L.A;
  const int AS = a_stride_for(nv);                     // row stride of sA
  const int JS = j_stride_direct(nv, NT);              // row stride of sJ (direct start)
  double* const sPiv = smem + L.piv;
  const double kInf = __builtin_huge_val();

  const bool is_body = lane < nbody;
  const bool is_dof = lane < nv;
  const int ntab = nv + P0.max_rows;                   // tableau indices in use (upper bound)

  // Problem distribution.  Workgroup g runs on XCD g % 8, and rows are not cache-line aligned (352 B of q, 224 B of
  // targets, 344 B of v per G1 problem): neighbouring problems share lines, and a line should not be fetched into two
  // L2s (a round-robin mapping measured 77 MB per launch instead of 61 MB).  So the static part of the batch — the
  // first A.static_rounds rounds of every wave — is one contiguous row range per XCD.  The rest is drawn one problem
  // at a time from a device-wide ticket counter: QP work varies by ±9 % per problem (active-set pivots), and with a
  // static 32 problems per wave the slowest wave of a 65 536 batch ran 5.5 % longer than the average one.  (One
  // counter for the whole device, not one per XCD: per-XCD tails measured 1.3 % slower — an XCD that runs behind
  // gets no help.)  The problem after this one is fixed at the top of the loop, so the atomic's latency hides behind
  // the current problem.  Same-address atomics retire at ≈80 M/s, which is why short problems — UR5e — stay fully
  // static, and why every wave opening with one costs more than the balance returns.
  const bool xcd_map = (gridDim.x & (kNumXcd - 1)) == 0;
  const int n_share = xcd_map ? (int)gridDim.x / kNumXcd : (int)gridDim.x;     // waves that share a row range
  const int xcd = xcd_map ? (int)blockIdx.x & (kNumXcd - 1) : 0;
  const int w_local = xcd_map ? (int)blockIdx.x / kNumXcd : (int)blockIdx.x;
  const bool tickets = A.static_rounds != 0x7fffffff;
  // rows per XCD range: the static part only (tickets), or the whole batch in eighths (static only)
  const int per = !xcd_map ? A.B : (tickets ? A.static_rounds * n_share : (A.B + kNumXcd - 1) / kNumXcd);
  const int pb_first = xcd * per;
  const int n_own = max(0, min(A.B, pb_first + per) - pb_first);
  int round = 0;
  auto draw_any = [&]() -> int {                                               // −1: nothing left for this wave
    if (round < A.static_rounds) {
      const int local = w_local + (round++) * n_share;
      return local < n_own ? pb_first + local : -1;
    }
    unsigned tk = 0;
    if (lane == 0) {
      tk = atomicAdd(A.work_counter, 1u);
      // the launch's last ticket (every wave ends on exactly one rejected draw): nobody draws after it
      if (tk == (unsigned)(A.B - A.static_rounds * (int)gridDim.x) + gridDim.x - 1u) atomicExch(A.work_counter, 0u);
    }
    const int pb = A.static_rounds * (int)gridDim.x + __builtin_amdgcn_readfirstlane((int)tk);
    return (unsigned)pb < (unsigned)A.B ? pb : -1;
  };
  // (collision builds, redo launch — SolveArgs::redo_mask: the problems the tight-rows launch before this one flagged.  Static
  //  distribution only (the host sets static_rounds = INT32_MAX): the wave reads the status of its next 64 candidates at once,
  //  one per lane, and walks the flagged ones — one load latency for a wave that has nothing to do, which is nearly all of them)
  unsigned long long redo_flags = 0;
  int redo_next = 0, redo_cur = 0;
  auto draw = [&]() -> int {
    if constexpr (!kColl) {
      return draw_any();
    } else {
      if (!A.redo_mask) return draw_any();
      for (;;) {
        if (!redo_flags) {
          const int local = w_local + (redo_next + lane) * n_share;
          const bool in = local < n_own;
          const int st = in ? A.status_out[pb_first + local] : 0;
          redo_flags = __ballot(in && (st & A.redo_mask) != 0);
          const bool any_in = __ballot(in) != 0;
          redo_cur = redo_next;
          redo_next += kWave;
          if (!redo_flags) { if (!any_in) return -1; continue; }
        }
        const int k = redo_cur + (int)__builtin_ctzll(redo_flags);
        redo_flags &= redo_flags - 1;
        return pb_first + w_local + k * n_share;
      }
    }
  };
#ifdef MKH_ONE_SHOT
  // One problem per workgroup, compiled as such (round 6; build.py W3_WOOD_ONE_SHOT — the host launches these builds with grid = B):
  // the loop below runs once, and without the draws, the double-buffered inputs and the scalars they carry around it the kernel
  // body of `44_32_r44_w3o` spills 26 SGPRs instead of 78 (109 `v_readlane` / 26 `v_writelane` instead of 206 / 78, 138 fewer VALU
  // and 304 fewer SALU instructions of its 1 867 / 1 185).  Same XCD ranges as the persistent shape: workgroup g is row g / 8 of XCD g % 8.
  (void)draw;
  int pb_next = xcd_map ? xcd * ((A.B + kNumXcd - 1) / kNumXcd) + w_local : (int)blockIdx.x;
  if (pb_next >= A.B) pb_next = -1;
#else
  int pb_next = draw();
#endif
  bool have_inputs = false;                            // the rows of `pb` are already on their way into (sq, sTgt)
#ifdef MKH_ONE_SHOT
  if (pb_next >= 0) {                                   // (no loop at all)
    const int pb = pb_next;
    pb_next = -1;
#else
  for (;;) {
    if (pb_next < 0) break;
    const int pb = pb_next;
    pb_next = draw();
#endif
    int status_all = 0;
    typename MKH_TAB<NT>::Regs ts;   // the tableau column (operand map: compiler-visible register tuples; else empty)
    long long tc[8];
    long long ta[6] = {0, 0, 0, 0, 0, 0}, tl = 0;   // QP sub-phase cycle sums (profiling)
    int tci = 0;
// Phase boundary: besides the profiling stamp, `lane` goes through an empty asm so that the compiler
// cannot keep per-lane LDS addresses (sX + 8·lane, sPiv + lane, …) alive across phases: it computed them
// once per kernel and then SPILLED them (9 of the 24 spills of the production variant), although each
// is one v_lshl_add away.
// MKH_CLK: the (B, 16) sink of the cycle stamps — the t_cycles tap of the builds with taps, or, in a -DMKH_CLOCKS experiment
// build of ANY variant (tools/phase_clocks.py: the production kernels cannot be tapped), SolveArgs::clk
#ifdef MKH_CLOCKS
#define MKH_CLK (A.clk)
#else
#define MKH_CLK MKH_TAP(t_cycles)
#endif
#define MKH_TICK() do { if (MKH_CLK) tc[tci] = __builtin_readcyclecounter(); ++tci; asm volatile("" : "+v"(lane)); } while (0)
#ifdef MKH_CLOCKS
    const int phase_stop = A.clk ? (int)A.clk[(size_t)pb * 24 + 15] : 0;   // (see MKH_PRE_STOP)
#define MKH_STOP(id) if (phase_stop && phase_stop <= (id)) break
#else
#define MKH_STOP(id) do {} while (0)
#endif
#define MKH_LAP0() do { if (MKH_CLK) tl = __builtin_readcyclecounter(); } while (0)
#define MKH_LAP(i) do { if (MKH_CLK) { const long long n_ = __builtin_readcyclecounter(); ta[i] += n_ - tl; tl = n_; } } while (0)
    MKH_MARK("problem_begin");
    MKH_TICK();   // 0: start
    // Opaque per-iteration zero: table loads below are indexed with it so that LICM cannot hoist
    // them out of the problem loop and keep ~60 VGPRs of lane constants live through the QP.
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const int ol = lane + oz;
    const DeviceProblem* Pq = Pg;
#ifndef MKH_ONE_SHOT
    asm volatile("" : "+s"(Pq));          // opaque: descriptor fields are (re)loaded where they are used
#endif                                    // (no loop to hoist them out of in the one-problem-per-workgroup build — and as kernel arguments, not opaque
                                          //  copies, the two pointers are re-read from the kernarg segment instead of spilled: 157 → 109 `v_readlane`)
    // ... through the CONSTANT address space: scalar loads.  As a plain reference the fields — wave-uniform, but read inside
    // lane-conditional code — came by per-lane flat_load, each a full s_waitcnt in front of the table load that depends on it:
    // the posture / box-limit phase alone was ten dependent L2 round trips (10.6 k of a G1 solve's 104 k wave cycles for ≈150
    // instructions; round 4, found with the phase × class census)
    const MKH_CONSTANT DeviceProblem& P = *(const MKH_CONSTANT DeviceProblem*)Pq;
    MKH_ARGS_AT_USE();                     // ... and so are the call arguments (low-rank variants)
    wave_sync();  // previous problem's LDS readers are done
    // ------------------------------------------------------------ load inputs
    // Double-buffered in LDS.  The rows of the NEXT problem (known since the top of the loop) are requested now with
    // global_load_lds — memory → LDS without passing through registers, so nothing stays live across the QP — into the
    // buffer this problem does not use; a wave's first problem fetches its own rows the same way.  The ≈3 k cycles of HBM
    // latency per problem (tools/phase_profile.py) disappear behind the previous problem's work.
    {
      typedef const __attribute__((address_space(1))) void* gptr_t;
      typedef __attribute__((address_space(3))) void* lptr_t;
      const int ndq = 2 * nq, ndt = 2 * P.n_frame * 7;                 // dwords per row
      auto fetch = [&](int pbn, double* dq, double* dt) {
        const char* gq = reinterpret_cast<const char*>(A.q + (size_t)pbn * nq);
        for (int b0 = 0; b0 < ndq; b0 += kWave)
          if (b0 + lane < ndq)
            __builtin_amdgcn_global_load_lds((gptr_t)(gq + 4 * (b0 + lane)), (lptr_t)(reinterpret_cast<char*>(dq) + 4 * b0), 4, 0, 0);
        const char* gt = reinterpret_cast<const char*>(A.frame_targets + (size_t)pbn * (ndt / 2));
        for (int b0 = 0; b0 < ndt; b0 += kWave)
          if (b0 + lane < ndt)
            __builtin_amdgcn_global_load_lds((gptr_t)(gt + 4 * (b0 + lane)), (lptr_t)(reinterpret_cast<char*>(dt) + 4 * b0), 4, 0, 0);
      };
      if (!have_inputs) { fetch(pb, sq, sTgt); have_inputs = prefetch; }
      double* const nq_buf = (sq == smem + L.q) ? smem + L.q2 : smem + L.q;
      double* const nt_buf = (sTgt == smem + L.tgt) ? smem + L.tgt2 : smem + L.tgt;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this problem's rows are in LDS
      if (prefetch && pb_next >= 0) fetch(pb_next, nq_buf, nt_buf);
      const int nt = P.n_frame * 7;
      const int nc = P.n_com * 3;
      if (lane < nc) sTgt[nt + lane] = A.com_target[(A.com_batched ? (size_t)pb * nc : 0) + lane];
    }
    wave_sync();
    // Fused outer loop (mink's callers iterate solve_ik + integrate_inplace, e.g.
    // examples/arm_ur5e_actuators.py:88-97): q stays in LDS between steps.
    const int n_steps = kSteps ? A.n_steps : 1;
    // Threshold-terminated loop (mkh_solve_until): the callers' loop breaks as soon as every end-effector error is
    // below (pos_threshold, ori_threshold) AFTER the integration (examples/arm_ur5e_actuators.py:88-97,
    // examples/arm_aloha.py:146-169).  Here the error at the integrated q is the first thing the next step computes
    // (FK + task lanes), so the test sits there; one extra check-only pass follows the last allowed iteration.
    const bool until = kSteps && A.pos_threshold >= 0.0;
    int it_done = 0, conv_flag = 0;
    // fused loop: where this lane's dof ended the previous step's QP (0 free, 1 at its lower bound, 2 at its upper): the next
    // step's active set is almost the same, so its first block step starts from there (phase 1a)
    int prev_bound = 0;
    // ... or the previous CALL on this handle (closed-loop callers: MKH_FLAG_WARM_START, SolveArgs::warm)
    // (variants with half-space rows run Goldfarb–Idnani only: no block step to seed)
    const bool warm_in = !kRows && A.warm != nullptr && A.warm_age >= 2;
    if (!kRows && warm_in && lane < nv) prev_bound = A.warm[(size_t)pb * nv + lane];
#if !(MKH_FEAT & 16)
    // (no fused loop in this build: ONE step, and no loop for the compiler to carry values around — as a single-trip `for` it still was
    //  one to the compiler: a dozen to twenty spilled SGPRs less in every build; plugin workload 1.643 → 1.603 ms, G1 full example 1.231
    //  → 1.224, the headline's twin 0.694 → 0.691; round 6)
    do {
    const int step = 0;
#else
    for (int step = 0; step < n_steps + (until ? 1 : 0); ++step) {
#endif
    int status = 0;
    tci = 1;                                                 // phase stamps 1..7 belong to the current step

    // FK, joint axes / dof lanes, subtree CoM, frame-task lanes (pre_phases above)
    const PreOut po = pre_phases(Pq, tp, pb, oz, (int)(sq - smem), (int)(sTgt - smem), until, A.pos_threshold, A.ori_threshold MKH_PRE_TC_ARGS);
#ifdef MKH_CALLS
    tci = 3;                                                  // (a callee has no stamps: phases 1 and 2 read as 0)
#endif
    asm volatile("" : "+v"(lane));
    status |= po.status;
    const int d_kind = po.d_kind, d_k = po.d_k, d_body = po.d_body, d_qadr = po.d_qadr;
    const bool conv_lane = po.conv != 0;
    const double mu_lane = po.mu_lane;
    const V3 com_root = po.com_root;
    const double* const my_dof = sDof + lane * 10;   // {ang, lin, anchor, q} of dof `lane`
#define MKH_LOAD_DOF_AXES()                                                        \
  const V3 d_ang{my_dof[0], my_dof[1], my_dof[2]}, d_lin{my_dof[3], my_dof[4], my_dof[5]}, \
      d_anchor{my_dof[6], my_dof[7], my_dof[8]}
    if (kSteps && until && step > 0) {
      it_done = step;
      if (!__ballot(!conv_lane)) { conv_flag = 1; status_all |= status; break; }   // every frame task achieved
      if (step == n_steps) { status_all |= status; break; }                        // iteration budget spent
    }
    double mu_total = A.damping + wave_sum(mu_lane);         // solve_ik.py:16 + Σ μ_t

    MKH_MARK("tasklanes_done");
    MKH_TICK();   // 3: task lanes done
    MKH_STOP(3);
    // ------------------------------------------- posture tasks (diagonal)
    double c_lane = 0.0;   // c[lane]
    double hdiag = 0.0;    // H[lane][lane]
    for (int t = 0; t < P.n_posture; ++t) {
      const double* tq = A.posture_target + (A.posture_batched ? ((size_t)pb * P.n_posture + t) * nq : (size_t)t * nq);
      double e = 0.0, jd = 0.0;
      const double cost = is_dof ? P.posture_cost[t * 64 + lane] : 0.0;
      if (is_dof) {
        if (d_kind == DOF_HINGE || d_kind == DOF_SLIDE) {
          e = tq[d_qadr] - my_dof[9];                        // mj_differentiatePos, dt = 1
          jd = -1.0;
        } else if (d_kind == DOF_BALL) {
          // ball joint: qvel = quat2Vel(conj(q1) ⊗ q2)      (posture_task.py:107)
          const int qa = d_qadr;
          Q4 q1{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]}, q2{tq[qa], tq[qa + 1], tq[qa + 2], tq[qa + 3]};
          const V3 dv = quat2vel(qmul(qconj(q1), q2));
          e = (d_k == 0) ? dv.x : ((d_k == 1) ? dv.y : dv.z);
          jd = -1.0;
        }  // free-joint dofs: error and Jacobian column zeroed (posture_task.py:115-116,139-141)
      }
      const double we = cost * (-P.posture_gain[t] * e);
      const double wj = cost * jd;
      hdiag += wj * wj;
      c_lane -= we * wj;
      if (P.posture_lm[t] != 0.0) mu_total += P.posture_lm[t] * wave_sum(we * we);
      if (MKH_TAP(t_task_e) && is_dof) MKH_TAP(t_task_e)[(size_t)pb * P.n_rows_tap + P.posture_row0[t] + lane] = e;
      if (MKH_TAP(t_task_J) && is_dof) {
        for (int r = 0; r < nv; ++r)
          MKH_TAP(t_task_J)[((size_t)pb * P.n_rows_tap + P.posture_row0[t] + r) * nv + lane] = (r == lane) ? jd : 0.0;
      }
    }
    // ComTask error & LM term (com_task.py:71-82)
    for (int t = 0; t < (kCom ? P.n_com : 0); ++t) {
      const double* tg = sTgt + P.n_frame * 7 + t * 3;
      const double e3[3] = {com_root.x - tg[0], com_root.y - tg[1], com_root.z - tg[2]};
      double ss = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double we = P.com_cost[t][r] * (-P.com_gain[t] * e3[r]);
        ss += we * we;
        if (MKH_TAP(t_task_e) && lane == 0) MKH_TAP(t_task_e)[(size_t)pb * P.n_rows_tap + P.com_row0[t] + r] = e3[r];
      }
      mu_total += P.com_lm[t] * ss;
    }

    // caller-defined tasks (dense rows): Levenberg–Marquardt term lm·‖W·(−gain·e)‖² of each (task.py:129-131)
    if (kDense && P.n_dense_rows > 0) {
      const double* de = A.dense_e + (size_t)pb * P.n_dense_rows;
      for (int t = 0; t < P.n_dense_tasks; ++t) {
        const int k0 = P.dense_row0[t], kt = P.dense_k[t];
        double ss = 0.0;
        for (int r = lane; r < kt; r += kWave) {
          const double we = P.dense_wgain[k0 + r] * de[k0 + r];
          ss += we * we;
        }
        mu_total += P.dense_lm[t] * wave_sum(ss);
      }
    }

    hdiag += mu_total;
#if defined(MKH_WOOD_CALL) && defined(MKH_CLOCKS)
    if (A.clk && lane == 0) A.clk[(size_t)pb * 24 + 22] = __builtin_readcyclecounter();            // [22] posture / damping / dense tasks done
    MKH_STOP(4);
#endif
    const double hdiag_base = hdiag;   // damping + Σμ + posture diagonal: the explicit diagonal of H

    // ------------------------------------------- collision half-space rows
    // BEFORE the tableau comes alive (round 3; they used to follow the H accumulation): the distance routines — the general
    // convex one above all — want registers, and while the pinned range is dead a real callee may use the whole file
    // (collision_phase above).  Rows go to sCol / sA, which nothing touches until the QP.
    int nrows = 0;
    bool rows_dropped = false;
    auto collision_rows = [&]() {
      if (kColl && P.n_pairs > 0) {
        const int r = collision_phase(Pq, tp, pb, A.dt, 0, L);
        asm volatile("" : "+v"(lane));
        nrows = r & 0xff; status |= (r >> 8) & 0xff; rows_dropped = ((r >> 16) & 1) != 0;
      }
    // caller-defined limits (Limit.compute_qp_inequalities, limits/limit.py:34-57): rows G·Δq ≤ h of this instance,
    // appended to the half-space rows; h = +inf marks an inactive row
    if (kDense && kRows && P.n_dense_limit_rows > 0) {
      const int M = P.n_dense_limit_rows;
      const double* dh = A.dense_h + (size_t)pb * M;
      const double* dG = A.dense_G + (size_t)pb * M * nv;
      const int first = nrows;
      for (int base = 0; base < M; base += kWave) {
        const int r = base + lane;
        const double hr = r < M ? dh[r] : kInf;
        const bool active = hr < kInf;
        const unsigned long long am = __ballot(active);
        const int slot = nrows + __popcll(am & ((1ull << lane) - 1ull));
        if (active && slot < P.max_rows) {
          double* o = sCol + slot * 16;
          o[9] = hr;
          o[12] = (double)r;
        }
        nrows += __popcll(am);
      }
      if (nrows > P.max_rows) { status |= 16; nrows = P.max_rows; }
      wave_sync();
      for (int s = first; s < nrows; ++s) {
        const int r = (int)sCol[s * 16 + 12];
        if (lane < AS) sA[s * AS + lane] = is_dof ? dG[(size_t)r * nv + lane] : 0.0;
      }
    }
    };
#if defined(MKH_COLL_CALL) || (MKH_FEAT & 128) || ((MKH_FEAT & 30) == 30) || (MKH_FEAT & 32)
    // (round 5: the parity build FEAT 31 — every feature + taps, phases inlined — carries the general convex routine too, and with
    //  it overlap_pair, a REAL call that owns the whole register file: behind the H accumulation it overwrote the pinned tableau
    //  of the rare instance with a PENETRATING general convex pair — a silently wrong H, hence v, found by comparing a call with
    //  taps against the plain call on 1 024 G1 instances with 45 contacts each)
    constexpr bool kCollFirst = true;      // a real callee (or, general convex pairs: overlap_pair inside the phase): while the tableau is dead
#else
    constexpr bool kCollFirst = false;     // inlined (plane / sphere / capsule builds): after the H accumulation, as in round 2
#endif
    if constexpr (kCollFirst) collision_rows();
    // ------------------------------ frame + CoM tasks: Jacobian columns, H, c
    const int n_jt = P.n_frame + (kCom ? P.n_com : 0);
    // The tableau column lives in pinned VGPRs (tab_asm.inc), outside the compiler's budget, so H is
    // accumulated right here, task by task, while the lane still holds its own weighted column.
    // (zeroed as late as possible: in the operand map the column's registers belong to the compiler until then)
    if constexpr (!kWood) { if (!(!kTaps && P.n_dpairs > 0)) MKH_TAB<NT>::zero(ts); }
    const double q_dof_stash = is_dof ? my_dof[9] : 0.0;    // (slot 9 of the dof stash is reused below)
    // ------------------------------------------------------------ box limits
    // lo ≤ Δq ≤ hi from ConfigurationLimit rows (configuration_limit.py:94-124) and
    // VelocityLimit rows (velocity_limit.py:96-101); never materialise the dense G.
    // (direct start: evaluated after the collision rows — lo / hi are not live while the distance routines run; low-rank
    //  start: before wood_start, which builds the tableau with the predicted active set already on its bounds)
    double lo = -kInf, hi = kInf;
    auto box_limits = [&]() {
    if (is_dof) {
        const double q_dof = q_dof_stash;
        for (int t = 0; t < P.n_cfg; ++t) {
          const double lw = P.cfg_lower[t * 64 + lane], up = P.cfg_upper[t * 64 + lane];
          if (d_kind == DOF_BALL) {
            // A limited ball joint.  The reference fills the joint's four qpos slots of `lower` / `upper` with the scalar
            // range ends and differentiates quaternions (configuration_limit.py:46-52, 94-112: mj_differentiatePos →
            // mju_subQuat → mju_quat2Vel, nothing normalised): Δq_max = quat2vel(q̄ ⊗ (u,u,u,u)), Δq_min = quat2vel((l,l,l,l)‾ ⊗ q).
            const Q4 qc{sq[d_qadr], sq[d_qadr + 1], sq[d_qadr + 2], sq[d_qadr + 3]};
            if (up < kInf) {
              const V3 dv = quat2vel(qmul(qconj(qc), Q4{up, up, up, up}));
              hi = fmin(hi, P.cfg_gain[t] * ((d_k == 0) ? dv.x : ((d_k == 1) ? dv.y : dv.z)));
            }
            if (lw > -kInf) {
              const V3 dv = quat2vel(qmul(qconj(Q4{lw, lw, lw, lw}), qc));
              lo = fmax(lo, -(P.cfg_gain[t] * ((d_k == 0) ? dv.x : ((d_k == 1) ? dv.y : dv.z))));
            }
            continue;
          }
          if (up < kInf) hi = fmin(hi, P.cfg_gain[t] * (up - q_dof));
          if (lw > -kInf) lo = fmax(lo, -(P.cfg_gain[t] * (q_dof - lw)));
        }
        for (int t = 0; t < P.n_vel; ++t) {
          const double vm = P.vel_limit[t * 64 + lane];
          if (vm < kInf) { hi = fmin(hi, A.dt * vm); lo = fmax(lo, -(A.dt * vm)); }
        }
        if constexpr (kDense) {     // single-entry rows of caller-defined limits (e.g. an acceleration limit [I; −I]): no tableau row
          if (A.dense_lo) lo = fmax(lo, A.dense_lo[(size_t)pb * nv + lane]);
          if (A.dense_hi) hi = fmin(hi, A.dense_hi[(size_t)pb * nv + lane]);
        }
        if (MKH_TAP(t_box_lo)) MKH_TAP(t_box_lo)[(size_t)pb * nv + lane] = lo;
        if (MKH_TAP(t_box_hi)) MKH_TAP(t_box_hi)[(size_t)pb * nv + lane] = hi;
      }
    };
    if constexpr (kWood) box_limits();
#if defined(MKH_WOOD_CALL) && defined(MKH_CLOCKS)
    if (A.clk && lane == 0) A.clk[(size_t)pb * 24 + 23] = __builtin_readcyclecounter();            // [23] box limits done
    MKH_STOP(5);
#endif
    // ---- low-rank start: Jacobian rows, S = I + Jh·Jhᵀ, LDLᵀ elimination (wood_start above), then the dof block of the
    // tableau by n_μ rank-1 updates that do not depend on each other:  R[i][j] = Σ_r Z[r][i]·Z[r][j]/d_r
    WoodOut wo{0.0, 0.0, 0.0, 0.0, 1.0, 0, 0};
    // predicted active set (warm starts: the previous fused step from the third step on / the previous call): the low-rank
    // start builds the tableau with these dofs already on their bounds
    int pred = 0;
    double pred_beta = 0.0;
    if constexpr (kWood) {
      if ((kSteps && step >= 2) || (warm_in && step == 0)) pred = prev_bound;
      if ((pred == 2 && !(hi < kInf)) || (pred == 1 && !(lo > -kInf)) || !is_dof) pred = 0;
      pred_beta = (pred == 2) ? hi : ((pred == 1) ? lo : 0.0);
    }
    if constexpr (kWood) {
#if defined(MKH_WOOD_CALL) && defined(MKH_CLOCKS)
      if (A.clk && lane == 0) A.clk[(size_t)pb * 24 + 16] = __builtin_readcyclecounter();          // [16] entry, [17] J rows, [18] S / w
      wo = wood_start(Pq, oz, (int)(sq - smem), (int)(sTgt - smem), c_lane, hdiag_base, pred, pred_beta, lo, hi, nrows,
                      A.clk ? A.clk + (size_t)pb * 24 + 17 : nullptr);
      if (A.clk && lane == 0) A.clk[(size_t)pb * 24 + 19] = __builtin_readcyclecounter();          // [19] back in the kernel
      MKH_STOP(8);
#elif defined(MKH_WOOD_CALL)
      wo = wood_start(Pq, oz, (int)(sq - smem), (int)(sTgt - smem), c_lane, hdiag_base, pred, pred_beta, lo, hi, nrows);
#else
      long long wprof[2] = {0, 0};
      const long long wt0 = MKH_TAP(t_cycles) ? __builtin_readcyclecounter() : 0;
      wo = wood_start(Pq, oz, (int)(sq - smem), (int)(sTgt - smem), c_lane, hdiag_base, pred, pred_beta, lo, hi, nrows, MKH_TAP(t_cycles) ? wprof : nullptr);
      if (MKH_TAP(t_cycles)) {                 // phase_profile.py: Jacobian rows | S and w | elimination  (slots 0, 1, 2)
        const long long wt1 = __builtin_readcyclecounter();
        ta[0] += wprof[0] - wt0; ta[1] += wprof[1] - wprof[0]; ta[2] += wt1 - wprof[1];
      }
#endif
      asm volatile("" : "+v"(lane));
      status |= wo.status;
      pred = wo.clamp;                         // (cold start: the bounds the unconstrained minimiser violates)
      hdiag = wo.hdiag;
      if (MKH_CLK) ta[3] -= __builtin_readcyclecounter();
      MKH_TAB<NT>::zero(ts);
      if constexpr (kWoodRows) {
        // the rows' block before the rank-1 updates: column nv + s (lane nv + s) = (Ah[s][·] | −Ah·Ah[s]ᵀ), rows nv + s of the dof
        // columns = Ah[s][lane] (wood_start scaled the staged rows in place and left the Gram entries in the contact table)
        if (nrows > 0) {
          const bool row_lane = lane >= nv && lane < nv + nrows;
          if (row_lane) load_leading_rows<NT>(ts, lds_addr(sA + (lane - nv) * AS), AS);   // (entries ≥ nv of a staged row are zero)
          for (int sr = 0; sr < nrows; ++sr)
            MKH_TAB<NT>::set_dyn(ts, nv + sr, is_dof ? sA[sr * AS + lane] : (row_lane ? -sCol[(lane - nv) * 16 + sr] : 0.0));
        }
      }
      const int n_mu = P.n_jrows;
      const double* const sDinv = sPiv + kWoodRow;
      // (streamed: the statement of row r requests the planes of row r + 1 behind its own FMAs, Z[r + 1][lane] is read one
      //  trip ahead — 18 updates × (LDS round trip + 44 FMAs) back to back were 14 % of a G1 solve)
      const unsigned lane_off = (unsigned)(lane & 15) << 3;
      double zr = (lane < NT && n_mu > 0) ? sJ[lane] : 0.0;
      MKH_TAB<NT>::rank1_prefetch(ts, lds_addr(sJ));
#pragma nounroll
      for (int r = 0; r < n_mu; ++r) {
        const double g = zr * sDinv[r];
        // dof rows the row reaches: [0, hb) (zero outside the kinematic chains eliminated so far)
        const unsigned long long nzd = __ballot(zr != 0.0);
        const int hb = nzd ? 64 - __builtin_clzll(nzd) : 0;
        zr = (lane < NT) ? sJ[(r + 1) * NT + lane] : 0.0;          // (row n_μ exists: the parked Σ Jh²)
        rank1_stream_rows<NT>(ts, lds_addr(sJ + (r + 1) * NT) + lane_off, g, hb);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the planes requested by the last statement are not used
      if (MKH_CLK) { asm volatile("s_waitcnt lgkmcnt(0)"); ta[3] += __builtin_readcyclecounter(); }   // (− start below)
    }
    auto frame_column = [&](int t, int k, uint64_t mask, uint64_t rmask, bool rel, double (&Jt)[6]) {
      frame_column_fn<kRel>(sTask, sDof, t, k, mask, rmask, rel, Jt);
    };
    // Direct start, production variants: the columns of EVERY frame task at once, one (task, dof) pair per lane (the
    // per-task loop below used to compute them task after task on the dof lanes: n_frame dependent passes of ≈150
    // fp64 operations each); the loop only stages the pair lanes' weighted rows and runs the rank-1 updates.
    const bool pair_path = !kWood && !kTaps && P.n_dpairs > 0;
    int p_task = -1, p_dof = 0;
    double Jp[6] = {0, 0, 0, 0, 0, 0};                      // weighted rows of this lane's pair
    if (pair_path) {
      const DirectPairs dp = direct_pairs(Pq);
      asm volatile("" : "+v"(lane));
      p_task = dp.p_task; p_dof = dp.p_dof;
#pragma unroll
      for (int r = 0; r < 6; ++r) Jp[r] = dp.Jp[r];
    }
    if constexpr (!kWood) { if (pair_path) MKH_TAB<NT>::zero(ts); }
    for (int t = 0; t < (kWood ? 0 : n_jt); ++t) {
      double Jt[6] = {0, 0, 0, 0, 0, 0}, cw[6] = {0, 0, 0, 0, 0, 0}, we6[6] = {0, 0, 0, 0, 0, 0};
      uint64_t mask;
      int nrow, row0, rowmask, jrow0;
      bool second_half;
      if (t < P.n_frame && pair_path) {
        const FrameTaskDev& ft = P.frame[t];
        const uint64_t cmask = uni((unsigned long long)ft.dof_mask) |
                               ((kRel && uni(ft.relative) != 0) ? uni((unsigned long long)ft.root_mask) : 0ull);
        rowmask = uni(ft.rowmask);
        const double* o = sTask + t * 64;
#pragma unroll
        for (int r = 0; r < 6; ++r) we6[r] = uni(o[30 + r]);
        wave_sync();                                           // previous task's rows are consumed
        {
          // compact rows of this task, sJ[c][0..JS): the pair lanes of the task write their dof's entries, dof lanes
          // off the chain write zeros (disjoint addresses)
          const bool off = lane < JS && !((cmask >> lane) & 1);
          int c = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r)
            if ((rowmask >> r) & 1) {
              if (off) sJ[c * JS + lane] = 0.0;
              if (p_task == t) sJ[c * JS + p_dof] = Jp[r];
              ++c;
            }
        }
        wave_sync();
        double Jw[6];
        {
          int c = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            Jw[r] = 0.0;
            if ((rowmask >> r) & 1) { if (is_dof) Jw[r] = sJ[c * JS + lane]; ++c; }
            c_lane -= we6[r] * Jw[r];                          // c = −weighted_errorᵀ·weighted_jacobian
            hdiag += Jw[r] * Jw[r];
          }
        }
        {
          // (streamed: each statement requests the planes of the task's next row behind its own FMAs)
          const unsigned lane_off = (unsigned)(lane & 15) << 3;
          MKH_TAB<NT>::rank1_prefetch(ts, lds_addr(sJ));
          int c = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r)
            if ((rowmask >> r) & 1) { rank1_stream_rows<NT>(ts, lds_addr(sJ + (c + 1) * JS) + lane_off, Jw[r], AS); ++c; }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the planes requested last are not used)
        }
                continue;
      }
      if (t < P.n_frame) {
        const FrameTaskDev& ft = P.frame[t];
        mask = uni((unsigned long long)ft.dof_mask);
        nrow = 6;
        row0 = uni(ft.row0);
        second_half = uni(ft.any_ori) != 0;
        rowmask = uni(ft.rowmask);
        jrow0 = uni(ft.jrow0);
        const double* o = sTask + t * 64;
#pragma unroll
        for (int r = 0; r < 6; ++r) { cw[r] = uni(ft.cost[r]); we6[r] = uni(o[30 + r]); }
        const bool rel = kRel && uni(ft.relative) != 0;
        const uint64_t rmask = rel ? uni((unsigned long long)ft.root_mask) : 0ull;
        if (is_dof && (((mask | rmask) >> lane) & 1)) frame_column(t, lane, mask, rmask, rel, Jt);
      } else {
        // CoM Jacobian column (mj_jacSubtreeCom closed form)
        const int tc = t - P.n_frame;
        mask = ~0ull;
        nrow = 3;
        row0 = P.com_row0[tc];
        second_half = false;
        rowmask = P.com_rowmask[tc];
        jrow0 = P.com_jrow0[tc];
        {
          const double* tg = sTgt + P.n_frame * 7 + tc * 3;
          const double e3[3] = {com_root.x - tg[0], com_root.y - tg[1], com_root.z - tg[2]};
#pragma unroll
          for (int r = 0; r < 3; ++r) { cw[r] = P.com_cost[tc][r]; we6[r] = cw[r] * (-P.com_gain[tc] * e3[r]); }
        }
        if (is_dof) {
          const double* cd = sCom + d_body * 4;
          const int inrobot = P.body_i[BI_IN_ROBOT * 64 + d_body];
          if (inrobot) {
            const double fac = cd[3] * fast_rcp(sCom[P.robot_root * 4 + 3]);
            MKH_LOAD_DOF_AXES();
            V3 jc = fac * (d_lin + cross(d_ang, V3{cd[0], cd[1], cd[2]} - d_anchor));
            Jt[0] = jc.x; Jt[1] = jc.y; Jt[2] = jc.z;
          }
        }
      }
      if (MKH_TAP(t_task_J) && is_dof) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if (r < nrow) MKH_TAP(t_task_J)[((size_t)pb * P.n_rows_tap + row0 + r) * nv + lane] = Jt[r];
      }
      double Jw[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        Jw[r] = cw[r] * Jt[r];                               // weighted_jacobian (task.py:129)
        c_lane -= we6[r] * Jw[r];                            // c = −weighted_errorᵀ·weighted_jacobian
        hdiag += Jw[r] * Jw[r];
      }
      wave_sync();                                             // previous task's rows are consumed
      if (lane < JS) {
        double* o = sJ + lane;                                 // compact rows of this task: sJ[c][0..JS)
        int c = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if ((rowmask >> r) & 1) { o[c * JS] = is_dof ? Jw[r] : 0.0; ++c; }
      }
      wave_sync();
      // H[:, lane] += Σ_r Jw_r · Jw_r[lane]: one rank-1 update per staged (nonzero-cost) row
      {
        int c = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if ((rowmask >> r) & 1) { rank1_leading_rows<NT>(ts, lds_addr(sJ + c * JS), is_dof ? Jw[r] : 0.0, AS); ++c; }
      }
    }
    // caller-defined tasks: rows of W·J straight from memory (row r is contiguous over the dofs: one coalesced load
    // per row), six at a time through the same staged rank-1 path as the built-in tasks
    if (kDense && !kWood) {
      const int K = P.n_dense_rows;
      const double* de = A.dense_e + (size_t)pb * K;
      const double* dJ = A.dense_J + (size_t)pb * K * nv;
      for (int r0 = 0; r0 < K; r0 += 6) {
        const int nr = min(6, K - r0);
        double Jw[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          Jw[r] = 0.0;
          if (r < nr) {
            const double cw = P.dense_cost[r0 + r], we = P.dense_wgain[r0 + r] * de[r0 + r];
            const double jr = is_dof ? dJ[(size_t)(r0 + r) * nv + lane] : 0.0;
            Jw[r] = cw * jr;                                   // weighted_jacobian (task.py:129)
            c_lane -= we * Jw[r];                              // c = −weighted_errorᵀ·weighted_jacobian
            hdiag += Jw[r] * Jw[r];
            if (MKH_TAP(t_task_J) && is_dof)
              MKH_TAP(t_task_J)[((size_t)pb * P.n_rows_tap + P.dense_tap_row0 + r0 + r) * nv + lane] = jr;
            if (MKH_TAP(t_task_e) && lane == 0)
              MKH_TAP(t_task_e)[(size_t)pb * P.n_rows_tap + P.dense_tap_row0 + r0 + r] = de[r0 + r];
          }
        }
        wave_sync();                                           // previous rows are consumed
        if (lane < JS) {
#pragma unroll
          for (int r = 0; r < 6; ++r) sJ[r * JS + lane] = Jw[r];
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if (r < nr) rank1_leading_rows<NT>(ts, lds_addr(sJ + r * JS), Jw[r], AS);
      }
    }
    MKH_MARK("jcols_done");
    long long tj = 0;
    if (MKH_CLK) tj = __builtin_readcyclecounter();     // profiling: end of the Jacobian-column loop
    if (MKH_TAP(t_c) && is_dof) MKH_TAP(t_c)[(size_t)pb * nv + lane] = c_lane;
    MKH_MARK("jcols_s_done");
    MKH_TICK();   // 4: posture + task Jacobian columns done
    if constexpr (!kCollFirst) collision_rows();
    if constexpr (!kWood) box_limits();
    wave_sync();

    MKH_MARK("limits_done");
    MKH_TICK();   // 5: limits + collision rows done
    if (!A.do_qp && !MKH_TAP(t_H)) break;

    // ------------------------------------------------- build the tableau column
    // lane j holds column j of K = [[H, Aᵀ],[A, 0]].  Built only now so that the 2·NT tableau
    // VGPRs are not live during FK / task / collision phases.
    // (the diagonal of K is carried separately in s.D; the diagonal register of the column is never read)
    if (MKH_TAP(t_H) && is_dof) {
      double* hrow = MKH_TAP(t_H) + (size_t)pb * nv * nv + lane;
      static_for<NT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i < nv) hrow[(size_t)i * nv] = (i == lane) ? hdiag : MKH_TAB<NT>::template get<i>(ts);
      });
    }
    if (!A.do_qp) break;
    if (kRows && !kWood && nrows > 0) {
      // rows nv+s of the dof columns: one indexed register write per active row (a static_for over all NT
      // rows with a runtime range test cost 860 VALU instructions and 278 spilled SGPRs) ...
      for (int sr = 0; sr < nrows; ++sr) MKH_TAB<NT>::set_dyn(ts, nv + sr, is_dof ? sA[sr * AS + lane] : 0.0);
      // ... and column nv+s (owned by lane nv+s) = A[s][:]  (entries ≥ nv of the staged row are zero)
      if (lane >= nv && lane < nv + nrows) {
        const unsigned addr = lds_addr(sA + (lane - nv) * AS);
        load_leading_rows<NT>(ts, addr, AS);                     // (rows ≥ AS keep the zeros of Tab::zero())
      }
    }

    // ====================================================================== QP
    // Dual active set (Goldfarb–Idnani) on the sweep tableau; see tools/proto_tableau_qp.py
    // for the numpy statement of the same algorithm.
    QpLane s;
    s.sg = 1.0; s.usign = 0; s.ysign = 0; s.rsign = 0; s.sel = 0; s.elig = 0;
    s.D = is_dof ? hdiag : ((lane >= ntab) ? 1.0 : 0.0);   // true diagonal of K
    s.x = 0.0; s.lo = -kInf; s.hi = kInf;
    double rown = 1.0;
    if (kWood) {
      // state after phase 0 (wood_start + the rank-1 updates above): every dof swept, the dof block is −H⁻¹
      // free dofs are basic (swept); the predicted ones sit on their bound, not swept: scale 1/σ, x = their gradient
      s.sg = is_dof ? (pred ? hdiag_base * wo.dsq : wo.dsq) : 1.0;
      s.D = is_dof ? wo.D : 1.0;                             // −H_FF⁻¹[j][j]  /  Schur complement diagonal
      s.usign = (is_dof && !pred) ? kSign : 0;
      s.sel = (is_dof && !pred) ? 1 : 0;
      s.elig = pred ? 1 : 0;
      s.ysign = (pred == 2) ? kSign : 0;
      s.rsign = (pred == 1) ? kSign : 0;
      s.x = is_dof ? wo.x : 0.0;                             // x_F = −H_FF⁻¹(c_F + H_FA·β)  /  w_A
      if (is_dof) { s.lo = lo; s.hi = hi; }
      if constexpr (kWoodRows) {
        if (lane >= nv && lane < nv + nrows) {               // an inactive half-space row: slack A·x0 − h, diagonal −A·H⁻¹·Aᵀ
          s.sel = 2; s.lo = 0.0; s.x = wo.x; s.D = wo.D; rown = wo.rown;
        }
      }
    } else if (is_dof) {
      s.x = c_lane; s.lo = lo; s.hi = hi;                     // nonbasic at z = 0: w = c   (sel = 1 once swept)
    } else if (kRows && lane < nv + nrows) {
      s.sel = 2; s.lo = 0.0;
      s.x = -sCol[(lane - nv) * 16 + 9];                      // w = A·0 − h
      const double* o = sA + (lane - nv) * AS;
      double nn = 0.0;
      for (int i = 0; i < nv; ++i) nn += o[i] * o[i];
      rown = sqrt(nn);
    }
    // inconsistent box ⇒ quadprog "constraints are inconsistent"
    if (__ballot(is_dof && lo > hi + 1e-12)) status |= 2;
    const double hmax = wave_max(is_dof ? hdiag : 0.0);
    const double thr_dof = 1e-13 * fast_rcp(hmax * (double)nv);

    int iters = 0;
    const int max_iters = 8 * (ntab + 8);
    // ---- phase 0: bring every dof into the basis, x0 = −H⁻¹c (Gauss–Jordan, no ratio tests).
    // Tight loop: publish row k → (LDS loads of the rank-1 update already in flight) → 1/d,
    // multipliers, z/w update → rank-1 update.
    // (low-rank start: the dofs are already in; the n_μ residual indices NR.. take the pivots, d < 0)
    const int k_begin = 0, k_end = kWood ? 0 : nv;           // (low-rank start: phase 0 is done)
    // The pivot order is known, so pivot k+1 is published BEFORE the rank-1 update of pivot k runs: its column
    // entries after update k are one FMA on row k+1 away (stored to the second buffer by the first instruction of
    // the update's asm statement), and D, σ, x of lane k+1 are final once the cheap per-lane updates of pivot k are
    // done (read with v_readlane into SGPRs).  The LDS round trip and the readlane latency hide under the FMA stream.
    double* bufc = sPiv;
    double* bufn = sPiv + kPivBuf;
    double own = 0.0;
    double pd = 1.0, psg = 1.0, px = 0.0;            // D, σ, x of the pivot lane (wave-uniform)
    if (!kWood && k_begin < k_end) {
      const double rowv = MKH_TAB<NT>::get_dyn(ts, k_begin);
      own = (lane == k_begin) ? 0.0 : rowv;
      wave_sync();                                   // earlier readers of the buffer are done
      bufc[lane] = own;
      pd = readlane_f64(s.D, k_begin); psg = readlane_f64(s.sg, k_begin); px = readlane_f64(s.x, k_begin);
    }
    for (int k = k_begin; !kWood && k < k_end; ++k) {
      MKH_MARK("p0_iter_begin");
            MKH_LAP0();
      wave_sync();
      MKH_TAB<NT>::rank1_prefetch(ts, lds_addr(bufc));
      MKH_LAP(0);
      if (!((kWood ? -pd : pd) > 0.0)) { status |= 4; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break; }
      const double inv = fast_rcp(pd);
      const double sk = psg;
      const double ck = s.sg * sk * own;                               // true T[lane][k]
      const double tau = (lane == k) ? pd : ck;                        // column k of the tableau
      const double alpha = -px * inv;                                  // drives w_k to 0
      s.x = fma(xor_sign(alpha, s.usign), tau, s.x);                   // basic: z −= α·τ, nonbasic: w += α·τ
      const double g = (sk * sk) * own * inv;                          // R-units multiplier of this lane's column
      if (lane == k) {
        s.x = alpha; s.usign = kSign; s.sel = kWood ? 0 : 1;           // z_k = 0 + α; now basic
        s.D = -inv;                                                    // T[k][k] = −1/d
        s.sg = sk * inv;                                               // row/column k scaled by 1/d
      } else {
        s.D = fma(-ck * inv, ck, s.D);                                 // T[j][j] −= T[j][k]²/d
      }
      double own_next = 0.0;
      const int kn = (k + 1 < k_end) ? k + 1 : k;                      // (last pivot: a harmless re-publication of column k)
      {
        const double rn = MKH_TAB<NT>::get_dyn(ts, kn);                        // R[kn][lane] before update k
        // R[kn][k]: lane kn's entry of the published column, by v_readlane (round 5; as a broadcast read of bufc[kn] it was a
        // full LDS round trip in front of the FMA that depends on it, in every pivot: Shadow config 4 0.342 -> 0.333 ms,
        // g1_coll 0.596 -> 0.580, the plugin workload 1.689 -> 1.680)
        const double cn = readlane_f64(own, kn);
        own_next = (lane == kn) ? 0.0 : fma(cn, -g, rn);               // the same FMA the update applies to row kn
        pd = readlane_f64(s.D, kn); psg = readlane_f64(s.sg, kn); px = readlane_f64(s.x, kn);
      }
      const unsigned pub_addr = lds_addr(bufn + lane);
      {
        MKH_TAB<NT>::rank1_body_pub(ts, lds_addr(bufc), -g, pub_addr, own_next);   // R[i][lane] −= R[i][k]·g   (row k: published 0)
      }
      MKH_LAP(1);
      double* t = bufc; bufc = bufn; bufn = t;
      own = own_next;
    }
    MKH_MARK("phase0_done");
    MKH_TICK();   // 6: tableau built, phase 0 done
    MKH_STOP(9);
    const int nact = (kWood && !kRows) ? nv : kWave;         // lanes that still own a live index
    int n_loop = 0, n_piv = 0;   // profiling (qp_iters tap): loop iterations / rank-1 pivots after x0
    // ---- phase 1a (box limits only): block principal pivoting.  lo ≤ x ≤ hi with H ≻ 0 is a bound-constrained
    // LCP with a P-matrix; Júdice & Pires (1994): flip ALL infeasible indices at once — free dofs outside their
    // bounds onto the violated bound, bound dofs whose multiplier has the wrong sign back into the basis.  On
    // the sweep tableau a flip is one rank-1 pivot WITHOUT arg-max reduction and ratio test (≈ a third of the
    // cost of an active-set iteration), and the first block step alone does what Goldfarb–Idnani needs ≈12
    // iterations for on the benchmark (every dof whose unconstrained step exceeds the velocity limit).
    // Block steps continue only while each at least halves the number of infeasibilities (≤ 3 steps): on
    // ill-conditioned, heavily saturated problems the block method flip-flops, and every extra pivot costs
    // accuracy (measured: 1e-6 instead of 1e-12 after ~100 pivots at cond(H) ≈ 1e5).  Then wrong-signed
    // multipliers are released one by one — the state becomes dual feasible — and Goldfarb–Idnani finishes from
    // there (tools/proto_bpp.py is the numpy statement and has the statistics).
    // With half-space rows (round 3): the same block steps on the BOX first, the rows passive — their slacks ride along in the
    // updates like any nonbasic index — and Goldfarb–Idnani then adds the violated rows from a state that is dual feasible
    // for the box (it used to do the ≈13 bound activations of a G1 solve itself: select, publish, ratio test, pivot each).
    bool need_gi = kRows;        // half-space rows: always finish with Goldfarb–Idnani
    // (in the builds of the plugin route: the collision builds' boxes are ConfigurationLimit rows that seldom bind, and the
    //  block-step code costs them registers — 64_72: 0 → 38 spilled VGPRs)
    constexpr bool kBoxSteps = !kRows || kDense || kWood;    // (round 6: and the low-rank builds with rows — a humanoid's velocity limits bind)
    if (kBoxSteps && !(status & 14)) {
      // Multipliers are sums of terms ≲ hmax·|Δq| ≈ hmax·1e-2: rounding noise ≈ 1e-16·hmax.  A bound dof whose
      // wrong-signed multiplier is below the threshold stays put (Δq error ≤ tolw / λ_min(H) ≈ 1e-12).
      const double tolw = 1e-16 * hmax;
      // one flip: clamp a basic dof k onto its violated bound (kb) / release a bound dof k into the basis
      auto flip = [&](int k, bool kb, bool up) {
        PivotScalars ps;
                MKH_LAP0();
        const double own = publish_column<NT, true>(ts, s, k, lane, sPiv, ps, nact, 1.0);
        MKH_TAB<NT>::rank1_prefetch(ts, lds_addr(sPiv));
        MKH_LAP(3);
        if (__ballot(!((kb ? -ps.d : ps.d) > 0.0))) { status |= 4; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
        const double inv = fast_rcp(ps.d);
        const double tau = (lane == k) ? ps.d : s.sg * ps.sg * own;
        const double beta = up ? ps.hi : ps.lo;
        const double alpha = kb ? (ps.x - beta) * inv : -ps.x * inv;   // z_k → β   /   w_k → 0
        s.x = fma(xor_sign(alpha, s.usign), tau, s.x);
        if (lane == k) {
          if (kb) { s.x = alpha; s.usign = 0; s.sel = 0; s.elig = 1; s.ysign = up ? kSign : 0; s.rsign = up ? 0 : kSign; }
          else { s.x = (s.ysign ? s.hi : s.lo) + alpha; s.usign = kSign; s.sel = 1; s.elig = 0; s.ysign = 0; s.rsign = 0; }
        }
        ++n_piv;
        MKH_LAP(4);
        pivot<NT, NR>(ts, s, k, kb, lane, sPiv, own, ps, inv);
        MKH_LAP(5);
      };
      int best = 4 * kWave;
      for (int outer = 0;; ++outer) {
        ++n_loop;
        const bool is_b = s.sel == 1;                                     // basic dof
        const unsigned long long m_over = __ballot(is_b && (s.x - s.hi > 1e-12));
        const unsigned long long m_under = __ballot(is_b && (s.lo - s.x > 1e-12));
        // a bound dof's multiplier y (−w at the upper bound, w at the lower) must stay ≥ 0
        const unsigned long long m_wrong = __ballot(s.elig != 0 && xor_sign(s.x, s.ysign) < -tolw);
        unsigned long long todo = m_over | m_under | m_wrong;
        if (!todo) break;
        const int cnt = __builtin_popcountll(todo);
        if (2 * cnt > best || outer >= 3) { need_gi = true; break; }
        best = cnt;
        unsigned long long m_basic = m_over | m_under, m_up = m_over;
        if (!kWood && outer == 0 && ((kSteps && step >= 2) || (warm_in && step == 0))) {
          // Warm start of a fused step.  Along an IK loop the active set grows to ≈26 of G1's 43 dofs and then changes by
          // ≈4 dofs per step, while the unconstrained step violates more bounds than end up active (28 against 21 on a
          // typical problem): cold, block pivoting over-clamps, releases, re-clamps — ≈54 pivots per solve from step 10 on,
          // against 13 on the first step.  So from the third step on the FIRST block step clamps exactly the dofs that sat on
          // a bound at the end of the previous step, onto that bound, violated or not; what that guess misses shows up as an
          // infeasibility or a wrong-signed multiplier in the next block step (numpy replay of the kernel's rules on
          // dumped problems: 54 → 34 pivots at step 10, Goldfarb–Idnani hand-overs 63 → 42 of 64; worse on step 1, where
          // the set still changes by ≈10 dofs).
          const unsigned long long w_any = __ballot(is_b && prev_bound != 0);
          if (w_any) { m_basic = w_any; m_up = __ballot(is_b && prev_bound == 2); todo = w_any; }
        }
        while (todo && !(status & 14)) {
          const int k = (int)__builtin_ctzll(todo);
          todo &= todo - 1;
          flip(k, ((m_basic >> k) & 1) != 0, ((m_up >> k) & 1) != 0);
        }
        if (status & 14) break;
      }
      // hand-over: release wrong-signed multipliers until the state is dual feasible (what Goldfarb–Idnani needs)
      while (need_gi && !(status & 14)) {
        const unsigned long long m_wrong = __ballot(s.elig != 0 && xor_sign(s.x, s.ysign) < -tolw);
        if (!m_wrong) break;
        if (++iters > max_iters) { status |= 8; break; }
        flip((int)__builtin_ctzll(m_wrong), false, false);
      }
    }
    // ---- phase 1b: Goldfarb–Idnani.  Each iteration publishes ONE column `col`; a blocking
    // constraint found by the ratio test becomes the column of the next iteration (`pend`).
    int p = -1;          // index being driven (−1 ⇒ select a new one)
    bool p_basic = true, upper = false;
    double sgn = 1.0;
    double acc = 0.0;    // step accumulated by the driven index: multiplier of a dof going to its bound / of a row coming in
    int pend = -1;       // pending sweep of a blocking index (reverse flag in pend_rev)
    bool pend_rev = false;
    const double inv_rown = (kRows && rown > 0.0) ? fast_rcp(rown) : 0.0;
    while (need_gi && !(status & 14)) {
      MKH_MARK("gi_iter_begin");
            ++n_loop;
      MKH_LAP0();
      int col;
      if (pend >= 0) col = pend;
      else {
        if (p < 0) {
          // ---- most violated primal condition (GI step 1).  "Most" only steers the path (the optimum
          // is unique), so the arg-max compares the high words of the violations: one 32-bit DPP
          // reduction instead of a 64-bit one, no readlane chain (p's scalars come with the column).
          const double over = s.x - s.hi, under = s.lo - s.x;
          double viol = fmax(over, under);
          if (kRows) viol = (s.sel == 2) ? s.x * inv_rown : viol;
          const bool cand = s.sel != 0 && viol > 1e-12;
          if (!__ballot(cand)) break;
          const unsigned vh = cand ? (unsigned)__double2hiint(viol) : 0u;
          const unsigned mh = wave_max_u32(vh);
          p = first_lane(cand && vh == mh);
          p_basic = kRows ? (((__ballot(s.usign != 0) >> p) & 1) != 0) : true;   // (A) basic dof / (B) inactive row
          upper = p_basic && (((__ballot(over > under) >> p) & 1) != 0);
          sgn = upper ? -1.0 : 1.0;
          acc = 0.0;
        }
        if (++iters > max_iters) { status |= 8; break; }
        col = p;
      }
      MKH_MARK("gi_publish");
      PivotScalars ps;
      MKH_LAP(2);
      const double own = publish_column<NT, true>(ts, s, col, lane, sPiv, ps, nact, rown);
      MKH_TAB<NT>::rank1_prefetch(ts, lds_addr(sPiv));
      MKH_LAP(3);
      const double inv = fast_rcp(ps.d);                         // 1 / T[col][col]
      bool rev = false;
      if (pend >= 0) {
        rev = pend_rev;
        pend = -1;
      } else {
        const double tau = (lane == col) ? ps.d : s.sg * ps.sg * own;   // column `col` of the tableau
        const double tpp = ps.d;
        const double beta = p_basic ? (upper ? ps.hi : ps.lo) : 0.0;
        const double thr = (!kRows || p_basic) ? thr_dof : thr_dof * ps.rn * ps.rn;
        // A row whose normal lies (almost) in the span of the active ones — |projected normal|² below 1e-6 of its unprojected
        // bound ‖a‖²/hmax: the sweep tableau carries the INVERSE Schur complement of the active rows explicitly and loses its
        // digits exactly there (many geom pairs of one body pair: ALOHA's 1 104-pair limit), where quadprog's orthogonal
        // factors do not.  The instance is flagged and solved again by the dense Goldfarb–Idnani of the workgroup-per-problem
        // kernel (MKH_ST_DEGENERATE is internal: the redo launch clears it).
        // (A row that has lost all its free dofs to their BOUNDS — every arm joint of a plugin row on its velocity limit — is
        //  exactly dependent and harmless: only dependence that runs through other rows counts.)
        if (kRows && !p_basic && !(-tpp > thr * (1e7 * (double)nv))) {
          if (__ballot(is_dof && s.usign != 0 && sA[(p - nv) * AS + lane] != 0.0)) status |= 32;
        }
        // full step length t2 (GI step 2b): z_p reaches its bound / the slack w_p reaches 0
        double t2 = kInf;
        if (p_basic ? (fabs(tpp) > thr) : (-tpp > thr)) t2 = fabs((ps.x - beta) * inv);
        // partial step length t1: keep the multipliers of the active set dual feasible.  Branch-free:
        // y = the multiplier that must stay ≥ 0 (z of an active row, −w at an upper bound, w at a lower
        // one), rr = its rate of decrease; one reciprocal instead of three divergent divisions.
        const double r = sgn * tau;
        const double y = xor_sign(s.x, s.ysign);
        const double rr = xor_sign(r, s.rsign);
        const bool cnd = s.elig != 0 && rr > 0.0;               // (p itself is never eligible)
        double t = fmax(y, 0.0) * fast_rcp(cnd ? rr : 1.0);
        t = (cnd && t == t) ? t : kInf;                          // (0·∞ from a denormal direction: no block)
        const double t1 = wave_min_nonneg(t, __ballot(cnd));
        if (__ballot(!(fmin(t1, t2) < kInf))) { status |= 2; break; }     // no step possible: infeasible
        const bool full = __ballot(t2 <= t1) != 0;              // wave-uniform: scalar branch
        const double alpha = sgn * (full ? t2 : t1);
        s.x = fma(xor_sign(alpha, s.usign), tau, s.x);          // basic: z −= α·τ, nonbasic: w += α·τ  (p included)
        acc += alpha;
        if (full) {
          if (lane == p) {
            // dof p lands on its bound with multiplier acc  /  row p enters the active set with λ = acc
            s.x = acc;
            s.usign = p_basic ? 0 : kSign;
            s.sel = 0;
            s.elig = 1;
            s.ysign = (p_basic && upper) ? kSign : 0;
            s.rsign = (p_basic && !upper) ? kSign : 0;
          }
          rev = p_basic;
          p = -1;
        } else {
          pend = first_lane(cnd && t == t1);                    // blocking index: sweep it next iteration
          pend_rev = kRows ? (((__ballot(s.usign != 0) >> pend) & 1) != 0) : false;
          if (lane == pend) {
            // an active row leaves with λ = 0 (its slack w = 0 right now); a dof leaves its bound with
            // w = 0 and z = the bound
            s.x = pend_rev ? 0.0 : (s.ysign ? s.hi : s.lo);
            s.usign = pend_rev ? 0 : kSign;
            s.sel = pend_rev ? 2 : 1;
            s.elig = 0; s.ysign = 0; s.rsign = 0;
          }
          // the prefetched loads are simply abandoned: drain them before LDS is reused
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          continue;
        }
      }
      MKH_MARK("gi_pivot");
      ++n_piv;
      MKH_LAP(4);
      pivot<NT, NR>(ts, s, col, rev, lane, sPiv, own, ps, inv);
      MKH_LAP(5);
    }
    // Δq of this dof: the free value when basic, else the bound it sits on
    const double zfin = s.usign ? s.x : (s.ysign ? s.hi : s.lo);
    if (kRows && (status & 32)) {
      // ... and only where it did damage: the digits go with the growth of the multipliers (λ·‖a‖ of an active row against the scale of
      // H).  Measured on 5 × 16 384 ALOHA instances: every instance beyond 1e-8 that the pivot test finds keeps its flag, 35 % of
      // the flagged ones — and the one Shadow instance in 65 536 — lose it.
      const double lam = (lane >= nv && s.elig != 0 && s.usign != 0) ? s.x * rown : 0.0;
      if (!(wave_max(lam) > hmax)) status &= ~32;
    }
    if (!kRows && (kSteps || A.warm)) prev_bound = (is_dof && !s.usign) ? (s.ysign ? 2 : 1) : 0;
    MKH_MARK("qp_done");
    MKH_TICK();   // 7: QP done
    if (MKH_CLK && lane < 16) {
      long long x = tc[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) x = (lane == i) ? tc[i] : x;
#pragma unroll
      for (int i = 0; i < 6; ++i) x = (lane == 8 + i) ? ta[i] : x;
      if (lane == 14) x = tj;
      if (lane == 15) x = 0;
#ifdef MKH_CLOCKS
      MKH_CLK[(size_t)pb * 24 + lane] = x;
#else
      MKH_CLK[(size_t)pb * 16 + lane] = x;
#endif
    }
    if (MKH_TAP(t_qp_iters) && lane == 0) MKH_TAP(t_qp_iters)[pb] = (kRows ? iters : n_piv) | (n_loop << 10) | (n_piv << 20);
    if (kColl && rows_dropped && !(status & 14)) {
      // the contacts that found no tableau row: G·Δq ≤ h at the solution?  (collision_phase, mode 1)
      wave_sync();
      if (is_dof) sDof[lane * 10 + 9] = zfin;                  // (slot 9 = q is dead now)
      wave_sync();
      if (collision_phase(Pq, tp, pb, A.dt, 1, L)) status |= 16;
      asm volatile("" : "+v"(lane));
    }
    // (builds without the fused loop: ONE step — an assignment, so that the 0 above is a constant of the early exits and not a value
    //  carried through the QP: the one-problem-per-workgroup twin spilled it)
    if constexpr (kSteps) status_all |= status; else status_all = status;
    const bool last = until || (step + 1 == n_steps) || (status & 14);   // (until: v of every step — the loop may end at the next check)
    if (last) {
      if (A.v_out && is_dof) {
        const double bad = __builtin_nan("");
        int pb_o = pb;
#ifdef MKH_ONE_SHOT
        asm volatile("" : "+s"(pb_o));       // (the row's address is computed HERE: in straight-line code it was hoisted to the kernel's entry and spilled)
#endif
        A.v_out[(size_t)pb_o * nv + lane] = (status & 14) ? bad : zfin / A.dt;   // v = dq / dt (solve_ik.py:104)
      }
      if (status & 14) break;
    }
    if (kSteps && (n_steps > 1 || A.q_out)) {
      // q ← q ⊕ Δq (mj_integratePos, Configuration.integrate_inplace, mink/configuration.py:228-236)
      wave_sync();
      if (is_dof) sDof[lane * 10 + 9] = zfin;                   // Δq of dof `lane` (slot 9 = q is dead now)
      wave_sync();
      if (lane < P.njnt) {
        const int jt = P.jnt_i[lane * JI_COUNT + JI_TYPE];
        int qa = P.jnt_i[lane * JI_COUNT + JI_QADR];
        int va = P.jnt_i[lane * JI_COUNT + JI_DADR];
        if (jt == JNT_HINGE || jt == JNT_SLIDE) {
          sq[qa] += sDof[va * 10 + 9];
        } else {
          if (jt == JNT_FREE) {
            for (int i = 0; i < 3; ++i) sq[qa + i] += sDof[(va + i) * 10 + 9];
            qa += 3; va += 3;
          }
          V3 w{sDof[va * 10 + 9], sDof[(va + 1) * 10 + 9], sDof[(va + 2) * 10 + 9]};
          const double n = sqrt(dot(w, w));
          V3 ax = (n < 1e-15) ? V3{1.0, 0.0, 0.0} : (1.0 / n) * w;
          Q4 qr = (n == 0.0) ? Q4{1, 0, 0, 0} : axis_angle(ax, n);
          Q4 r = qmul(qnormalize(Q4{sq[qa], sq[qa + 1], sq[qa + 2], sq[qa + 3]}), qr);
          sq[qa] = r.w; sq[qa + 1] = r.x; sq[qa + 2] = r.y; sq[qa + 3] = r.z;
        }
      }
      wave_sync();
    }
#if !(MKH_FEAT & 16)
    } while (0);
#else
    }  // step loop
#endif
    if (kSteps && A.q_out) {
      for (int i = lane; i < nq; i += 64) A.q_out[(size_t)pb * nq + i] = sq[i];
    }
    if (kSteps && until && lane == 0) {
      if (A.iters_out) A.iters_out[pb] = it_done;
      if (A.converged_out) A.converged_out[pb] = conv_flag;
    }
    int pb_w = pb;
#ifdef MKH_ONE_SHOT
    asm volatile("" : "+s"(pb_w));
#endif
    if (!kRows && A.warm && lane < nv) A.warm[(size_t)pb_w * nv + lane] = (int8_t)((status_all & 14) ? 0 : prev_bound);
    if (A.status_out && lane == 0) A.status_out[pb_w] = status_all;
    sq = (sq == smem + L.q) ? smem + L.q2 : smem + L.q;           // the next problem's rows are (being) fetched there
    sTgt = (sTgt == smem + L.tgt) ? smem + L.tgt2 : smem + L.tgt;
  }
}

#endif  // MKH_NT

#ifndef MKH_NT   // the host translation unit (minkhip.hip) owns the small streaming kernel
// q_out = q ⊕ v·dt (mj_integratePos; Configuration.integrate, mink/configuration.py:214-226).
// One thread per (problem, joint).
__global__ __launch_bounds__(256) void integrate_kernel(const DeviceProblem P, int B, const double* __restrict__ q,
                                                        const double* __restrict__ v, double dt,
                                                        double* __restrict__ q_out) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * P.njnt;
  if (tid >= total) return;
  const int b = (int)(tid / P.njnt), j = (int)(tid % P.njnt);
  const int jt = P.jnt_i[j * JI_COUNT + JI_TYPE];
  int qa = P.jnt_i[j * JI_COUNT + JI_QADR];
  int va = P.jnt_i[j * JI_COUNT + JI_DADR];
  const double* qi = q + (size_t)b * P.nq;
  const double* vi = v + (size_t)b * P.nv;
  double* qo = q_out + (size_t)b * P.nq;
  if (jt == JNT_HINGE || jt == JNT_SLIDE) { qo[qa] = qi[qa] + dt * vi[va]; return; }
  if (jt == JNT_FREE) {
    for (int i = 0; i < 3; ++i) qo[qa + i] = qi[qa + i] + dt * vi[va + i];
    qa += 3; va += 3;
  }
  // mju_quatIntegrate: q ← normalize(q) ⊗ axisangle(v̂, dt·|v|)
  V3 w{vi[va], vi[va + 1], vi[va + 2]};
  const double n = sqrt(dot(w, w));
  V3 ax = (n < 1e-15) ? V3{1.0, 0.0, 0.0} : (1.0 / n) * w;
  const double ang = dt * n;
  Q4 qr = (ang == 0.0) ? Q4{1, 0, 0, 0} : axis_angle(ax, ang);
  Q4 q0 = qnormalize(Q4{qi[qa], qi[qa + 1], qi[qa + 2], qi[qa + 3]});
  Q4 r = qmul(q0, qr);
  qo[qa] = r.w; qo[qa + 1] = r.x; qo[qa + 2] = r.y; qo[qa + 3] = r.z;
}

#endif  // !MKH_NT

}  // namespace mkh
