// One 16-LANE ROW per problem: differential IK for small robots (nv ≤ 16, hinge / slide joints): arms at small and mid-size
// batches, hands and mobile arms (9 … 16 dofs) at every batch size.  Round 4: the same source on TWO rows per problem (LP = 32,
// two problems per wavefront) for 17 … 32 dofs or links — Unitree H1 / Go1, Spot, Allegro, arms carrying hands — with a floating
// base (a free joint = three slide links + a quaternion link), ComTask (com_task.py:71-97) and RelativeFrameTask
// (relative_frame_task.py:106-142); see "two DPP rows per problem" below and DESIGN.md §3.4.
//
// The wavefront kernel (ik_kernel.h) gives a 6-dof arm 64 lanes of which 6-10 work; the lane kernel (lane_kernel.h)
// gives it one lane, i.e. one ≈48 µs dependent instruction stream — right for ≥ 8 192 problems, where every SIMD
// has several wavefronts to interleave, wrong for the 1 024-4 096 instances of a typical parallel-environment batch
// (BASELINE.json configs[2] at its quoted batch): 4 096 problems are 64 lane-kernel wavefronts on 1 024 SIMDs.
// Here a problem owns one DPP row (16 lanes) of a wavefront, four problems per wavefront, so that 4 096 problems put one
// wavefront on every SIMD of the chip and each problem still spreads its work over its lanes:
//
//   lanes as LINKS   forward kinematics: local joint transforms, then world poses by pointer jumping up the parent
//                    chain (⌈log₂ depth⌉ rounds through LDS instead of `depth` dependent compositions);
//   lanes as TASKS   frame-task error, log / jlog and the two 3×3 blocks every Jacobian column needs (lie_dev.h — the same
//                    functions the other two kernels run);
//   lanes as DOFS    Jacobian columns, H = Σ JᵀW²J + (λ + Σμ)I one COLUMN per lane in NT = 8 or 16 registers, box limits, the QP.
//
// The 16-lane row is the unit the DPP operand network broadcasts in: `v_fmac_f64_dpp T[i], u, g row_newbcast:i` adds
// u(lane i of MY row)·g to T[i] — a rank-1 update of four NT×NT matrices, one per row, in NT instructions without a
// byte of LDS traffic.  H assembly, the pivots of the QP and its matrix-vector products are all that instruction.
//
// QP: block principal pivoting from the partition of the diagonal estimate; at its first stall the primal active-set
// iteration from the clipped point (monotone), Murty's rule behind it as the finite last resort (see the loop) — on
// a SWEEP tableau instead of a fresh factorisation per iteration: sweeping index k in or out of the free set is one
// principal pivot of the symmetric matrix [[H, c], [cᵀ, ·]]; with the free set swept, column j holds −x_j (free) or the
// multiplier w_j (bound) as  ĉ_j + Σ_b T[b][j]·x_b  over the bound indices b.  Unique optimum of a strictly convex QP ⇒
// quadprog's answer (tests/test_gpu_quad_kernel.py: against the wavefront kernel, the lane kernel and both oracles).
// Same reference path: mink/solve_ik.py:68-105 and what it calls (see ik_kernel.h).
#pragma once
#include "lane_kernel.h"
#include "wave_ops.h"
#include <utility>

namespace mkh {

// LP = lanes per problem: 16 (one DPP row, nv ≤ 16) or 32 (TWO rows, 17 … 32 dofs or links, floating bases: Unitree H1 / Go1, Spot,
// Allegro — round 4).  On two rows the DPP broadcast still works row by row, so a value u that every lane of the problem needs
// from lane i is taken from `ua` = (own u on row 0, the other row's u on row 1) for i < 16 and from `ub` (the other way round) for
// i ≥ 16: one ds_swizzle exchange of the rows (lane ⊕ 16) per broadcast vector, then 16 + 16 `row_newbcast` FMAs.
constexpr int kQuadTaskDoubles = 28;                          // per frame task: A1 (9), A2 (9), frame position (3), W·e (6), μ
// poses [component][link], ancestors, task blocks; two-row build: + m·com of every link and the subtree centres of mass (ComTask)
__host__ __device__ constexpr int quad_row_doubles(int lp) { return 7 * lp + lp / 2 + kLaneMaxFrames * kQuadTaskDoubles + (lp == 32 ? 6 * lp : 0); }
static_assert(kLaneMaxLinks <= 16 && kLaneDescDofs <= 16 && kLaneMaxLinks2 <= 32 && kLaneDescDofs2 <= 32, "a problem must fit one / two DPP rows");
__host__ __device__ inline int quad_lds_bytes(int lp = 16) { return (kWave / lp) * quad_row_doubles(lp) * (int)sizeof(double); }

// ---------------------------------------------------------------- DPP row primitives (all 64 lanes must be active)
// s_nop 4: a DPP operand must not be read within 5 wait states of an EXEC write (and 2 of a VALU write of that register)
// by the compiler's code before the statement.
#define MKH_QFMAC(D, I) "v_fmac_f64_dpp " D ", %[u], %[g] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
// T[i] += u(lane i of this row) · g   (8 or 16 column registers)
__device__ __forceinline__ void quad_rank1(double (&T)[8], const double u, const double g) {
  asm volatile("s_nop 4\n\t" MKH_QFMAC("%[t0]", 0) MKH_QFMAC("%[t1]", 1) MKH_QFMAC("%[t2]", 2) MKH_QFMAC("%[t3]", 3)
               MKH_QFMAC("%[t4]", 4) MKH_QFMAC("%[t5]", 5) MKH_QFMAC("%[t6]", 6) MKH_QFMAC("%[t7]", 7)
               : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [t4] "+v"(T[4]), [t5] "+v"(T[5]),
                 [t6] "+v"(T[6]), [t7] "+v"(T[7])
               : [u] "v"(u), [g] "v"(g));
}
__device__ __forceinline__ void quad_rank1(double (&T)[16], const double u, const double g) {
  asm volatile("s_nop 4\n\t" MKH_QFMAC("%[t0]", 0) MKH_QFMAC("%[t1]", 1) MKH_QFMAC("%[t2]", 2) MKH_QFMAC("%[t3]", 3)
               MKH_QFMAC("%[t4]", 4) MKH_QFMAC("%[t5]", 5) MKH_QFMAC("%[t6]", 6) MKH_QFMAC("%[t7]", 7)
               MKH_QFMAC("%[t8]", 8) MKH_QFMAC("%[t9]", 9) MKH_QFMAC("%[t10]", 10) MKH_QFMAC("%[t11]", 11)
               MKH_QFMAC("%[t12]", 12) MKH_QFMAC("%[t13]", 13) MKH_QFMAC("%[t14]", 14) MKH_QFMAC("%[t15]", 15)
               : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [t4] "+v"(T[4]), [t5] "+v"(T[5]),
                 [t6] "+v"(T[6]), [t7] "+v"(T[7]), [t8] "+v"(T[8]), [t9] "+v"(T[9]), [t10] "+v"(T[10]), [t11] "+v"(T[11]),
                 [t12] "+v"(T[12]), [t13] "+v"(T[13]), [t14] "+v"(T[14]), [t15] "+v"(T[15])
               : [u] "v"(u), [g] "v"(g));
}
#undef MKH_QFMAC
// acc + Σ_i u(lane i of this row) · T[i]   (two accumulators: half the dependent chain)
#define MKH_QDOT(A, I) "v_fmac_f64_dpp " A ", %[u], %[t" #I "] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ double quad_dot(const double (&T)[8], const double u, const double acc) {
  double a = acc, b = 0.0;
  asm volatile("s_nop 4\n\t" MKH_QDOT("%[a]", 0) MKH_QDOT("%[b]", 1) MKH_QDOT("%[a]", 2) MKH_QDOT("%[b]", 3)
               MKH_QDOT("%[a]", 4) MKH_QDOT("%[b]", 5) MKH_QDOT("%[a]", 6) MKH_QDOT("%[b]", 7)
               : [a] "+v"(a), [b] "+v"(b)
               : [u] "v"(u), [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [t3] "v"(T[3]), [t4] "v"(T[4]), [t5] "v"(T[5]),
                 [t6] "v"(T[6]), [t7] "v"(T[7]));
  return a + b;
}
__device__ __forceinline__ double quad_dot(const double (&T)[16], const double u, const double acc) {
  double a = acc, b = 0.0;
  asm volatile("s_nop 4\n\t" MKH_QDOT("%[a]", 0) MKH_QDOT("%[b]", 1) MKH_QDOT("%[a]", 2) MKH_QDOT("%[b]", 3)
               MKH_QDOT("%[a]", 4) MKH_QDOT("%[b]", 5) MKH_QDOT("%[a]", 6) MKH_QDOT("%[b]", 7)
               MKH_QDOT("%[a]", 8) MKH_QDOT("%[b]", 9) MKH_QDOT("%[a]", 10) MKH_QDOT("%[b]", 11)
               MKH_QDOT("%[a]", 12) MKH_QDOT("%[b]", 13) MKH_QDOT("%[a]", 14) MKH_QDOT("%[b]", 15)
               : [a] "+v"(a), [b] "+v"(b)
               : [u] "v"(u), [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [t3] "v"(T[3]), [t4] "v"(T[4]), [t5] "v"(T[5]),
                 [t6] "v"(T[6]), [t7] "v"(T[7]), [t8] "v"(T[8]), [t9] "v"(T[9]), [t10] "v"(T[10]), [t11] "v"(T[11]),
                 [t12] "v"(T[12]), [t13] "v"(T[13]), [t14] "v"(T[14]), [t15] "v"(T[15]));
  return a + b;
}
#undef MKH_QDOT
// u of lane K of this row
template <int K> __device__ __forceinline__ double quad_bcast(double u);
#define MKH_QBCAST(K)                                                                                              \
  template <> __device__ __forceinline__ double quad_bcast<K>(const double u) {                                     \
    double r = 0.0;                                                                                                 \
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %[r], %[u], %[one] row_newbcast:" #K " row_mask:0xf bank_mask:0xf"     \
                 : [r] "+v"(r) : [u] "v"(u), [one] "v"(1.0));                                                      \
    return r;                                                                                                       \
  }
MKH_QBCAST(0) MKH_QBCAST(1) MKH_QBCAST(2) MKH_QBCAST(3) MKH_QBCAST(4) MKH_QBCAST(5) MKH_QBCAST(6) MKH_QBCAST(7)
MKH_QBCAST(8) MKH_QBCAST(9) MKH_QBCAST(10) MKH_QBCAST(11) MKH_QBCAST(12) MKH_QBCAST(13) MKH_QBCAST(14) MKH_QBCAST(15)
#undef MKH_QBCAST
template <int N, typename F, int... I>
__device__ __forceinline__ void quad_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void quad_for(F&& f) { quad_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }
// Σ / max over the 16 lanes of each row (every lane gets its row's value)
__device__ __forceinline__ double quad_sum(double x) {
  x += dpp_f64<0xB1>(x); x += dpp_f64<0x4E>(x); x += dpp_f64<0x141>(x); x += dpp_f64<0x140>(x);
  return x;
}
__device__ __forceinline__ double quad_min(double x) {
  x = fmin(x, dpp_f64<0xB1>(x)); x = fmin(x, dpp_f64<0x4E>(x)); x = fmin(x, dpp_f64<0x141>(x)); x = fmin(x, dpp_f64<0x140>(x));
  return x;
}
__device__ __forceinline__ double quad_max(double x) {
  x = fmax(x, dpp_f64<0xB1>(x)); x = fmax(x, dpp_f64<0x4E>(x)); x = fmax(x, dpp_f64<0x141>(x)); x = fmax(x, dpp_f64<0x140>(x));
  return x;
}


// ---------------------------------------------------------------- two DPP rows per problem
// the other row's value (lane ⊕ 16): ds_swizzle, bit mode, xor mask 0x10
__device__ __forceinline__ double quad_swap16(const double x) {
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x401F), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x401F);
  return __hiloint2double(hi, lo);
}
struct QuadAB { double a, b; };                      // u as seen through row_newbcast:i for i < 16 (a) and for i ≥ 16 (b)
__device__ __forceinline__ QuadAB quad_ab(const double u, const bool upper_row) {
  const double sw = quad_swap16(u);
  return QuadAB{upper_row ? sw : u, upper_row ? u : sw};
}
#define MKH_QFMAC2(D, U, I) "v_fmac_f64_dpp " D ", " U ", %[g] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void quad_rank1_ab(double (&T)[32], const QuadAB u, const double g) {
  asm volatile("s_nop 4\n\t" MKH_QFMAC2("%[t0]", "%[u]", 0) MKH_QFMAC2("%[t1]", "%[u]", 1) MKH_QFMAC2("%[t2]", "%[u]", 2) MKH_QFMAC2("%[t3]", "%[u]", 3)
               MKH_QFMAC2("%[t4]", "%[u]", 4) MKH_QFMAC2("%[t5]", "%[u]", 5) MKH_QFMAC2("%[t6]", "%[u]", 6) MKH_QFMAC2("%[t7]", "%[u]", 7)
               MKH_QFMAC2("%[t8]", "%[u]", 8) MKH_QFMAC2("%[t9]", "%[u]", 9) MKH_QFMAC2("%[t10]", "%[u]", 10) MKH_QFMAC2("%[t11]", "%[u]", 11)
               MKH_QFMAC2("%[t12]", "%[u]", 12) MKH_QFMAC2("%[t13]", "%[u]", 13) MKH_QFMAC2("%[t14]", "%[u]", 14) MKH_QFMAC2("%[t15]", "%[u]", 15)
               : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [t4] "+v"(T[4]), [t5] "+v"(T[5]),
                 [t6] "+v"(T[6]), [t7] "+v"(T[7]), [t8] "+v"(T[8]), [t9] "+v"(T[9]), [t10] "+v"(T[10]), [t11] "+v"(T[11]),
                 [t12] "+v"(T[12]), [t13] "+v"(T[13]), [t14] "+v"(T[14]), [t15] "+v"(T[15])
               : [u] "v"(u.a), [g] "v"(g));
  asm volatile("s_nop 4\n\t" MKH_QFMAC2("%[t0]", "%[u]", 0) MKH_QFMAC2("%[t1]", "%[u]", 1) MKH_QFMAC2("%[t2]", "%[u]", 2) MKH_QFMAC2("%[t3]", "%[u]", 3)
               MKH_QFMAC2("%[t4]", "%[u]", 4) MKH_QFMAC2("%[t5]", "%[u]", 5) MKH_QFMAC2("%[t6]", "%[u]", 6) MKH_QFMAC2("%[t7]", "%[u]", 7)
               MKH_QFMAC2("%[t8]", "%[u]", 8) MKH_QFMAC2("%[t9]", "%[u]", 9) MKH_QFMAC2("%[t10]", "%[u]", 10) MKH_QFMAC2("%[t11]", "%[u]", 11)
               MKH_QFMAC2("%[t12]", "%[u]", 12) MKH_QFMAC2("%[t13]", "%[u]", 13) MKH_QFMAC2("%[t14]", "%[u]", 14) MKH_QFMAC2("%[t15]", "%[u]", 15)
               : [t0] "+v"(T[16]), [t1] "+v"(T[17]), [t2] "+v"(T[18]), [t3] "+v"(T[19]), [t4] "+v"(T[20]), [t5] "+v"(T[21]),
                 [t6] "+v"(T[22]), [t7] "+v"(T[23]), [t8] "+v"(T[24]), [t9] "+v"(T[25]), [t10] "+v"(T[26]), [t11] "+v"(T[27]),
                 [t12] "+v"(T[28]), [t13] "+v"(T[29]), [t14] "+v"(T[30]), [t15] "+v"(T[31])
               : [u] "v"(u.b), [g] "v"(g));
}
#undef MKH_QFMAC2
#define MKH_QDOT2(A, I) "v_fmac_f64_dpp " A ", %[u], %[t" #I "] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ double quad_dot_ab(const double (&T)[32], const QuadAB u, const double acc) {
  double a = acc, b = 0.0;
  asm volatile("s_nop 4\n\t" MKH_QDOT2("%[a]", 0) MKH_QDOT2("%[b]", 1) MKH_QDOT2("%[a]", 2) MKH_QDOT2("%[b]", 3)
               MKH_QDOT2("%[a]", 4) MKH_QDOT2("%[b]", 5) MKH_QDOT2("%[a]", 6) MKH_QDOT2("%[b]", 7)
               MKH_QDOT2("%[a]", 8) MKH_QDOT2("%[b]", 9) MKH_QDOT2("%[a]", 10) MKH_QDOT2("%[b]", 11)
               MKH_QDOT2("%[a]", 12) MKH_QDOT2("%[b]", 13) MKH_QDOT2("%[a]", 14) MKH_QDOT2("%[b]", 15)
               : [a] "+v"(a), [b] "+v"(b)
               : [u] "v"(u.a), [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [t3] "v"(T[3]), [t4] "v"(T[4]), [t5] "v"(T[5]),
                 [t6] "v"(T[6]), [t7] "v"(T[7]), [t8] "v"(T[8]), [t9] "v"(T[9]), [t10] "v"(T[10]), [t11] "v"(T[11]),
                 [t12] "v"(T[12]), [t13] "v"(T[13]), [t14] "v"(T[14]), [t15] "v"(T[15]));
  asm volatile("s_nop 4\n\t" MKH_QDOT2("%[a]", 0) MKH_QDOT2("%[b]", 1) MKH_QDOT2("%[a]", 2) MKH_QDOT2("%[b]", 3)
               MKH_QDOT2("%[a]", 4) MKH_QDOT2("%[b]", 5) MKH_QDOT2("%[a]", 6) MKH_QDOT2("%[b]", 7)
               MKH_QDOT2("%[a]", 8) MKH_QDOT2("%[b]", 9) MKH_QDOT2("%[a]", 10) MKH_QDOT2("%[b]", 11)
               MKH_QDOT2("%[a]", 12) MKH_QDOT2("%[b]", 13) MKH_QDOT2("%[a]", 14) MKH_QDOT2("%[b]", 15)
               : [a] "+v"(a), [b] "+v"(b)
               : [u] "v"(u.b), [t0] "v"(T[16]), [t1] "v"(T[17]), [t2] "v"(T[18]), [t3] "v"(T[19]), [t4] "v"(T[20]), [t5] "v"(T[21]),
                 [t6] "v"(T[22]), [t7] "v"(T[23]), [t8] "v"(T[24]), [t9] "v"(T[25]), [t10] "v"(T[26]), [t11] "v"(T[27]),
                 [t12] "v"(T[28]), [t13] "v"(T[29]), [t14] "v"(T[30]), [t15] "v"(T[31]));
  return a + b;
}
#undef MKH_QDOT2
// the operations of the kernel body by lanes per problem
template <int LP, int NT> __device__ __forceinline__ void qrank1(double (&T)[NT], const double u, const double g, const bool upper_row) {
  if constexpr (LP == 32) quad_rank1_ab(T, quad_ab(u, upper_row), g); else quad_rank1(T, u, g);
}
template <int LP, int NT> __device__ __forceinline__ double qdot(const double (&T)[NT], const double u, const double acc, const bool upper_row) {
  if constexpr (LP == 32) return quad_dot_ab(T, quad_ab(u, upper_row), acc); else return quad_dot(T, u, acc);
}
template <int LP> __device__ __forceinline__ double qsum(double x) {
  x = quad_sum(x);
  if constexpr (LP == 32) x += quad_swap16(x);
  return x;
}
template <int LP> __device__ __forceinline__ double qmin(double x) {
  x = quad_min(x);
  if constexpr (LP == 32) x = fmin(x, quad_swap16(x));
  return x;
}
template <int LP> __device__ __forceinline__ double qmax(double x) {
  x = quad_max(x);
  if constexpr (LP == 32) x = fmax(x, quad_swap16(x));
  return x;
}

// Principal pivot on index K of the rows whose `act` is set: s = +1 sweeps K INTO the free set (pivot element: a Schur
// complement diagonal of H, > 0), s = −1 sweeps it OUT (−(H_FF⁻¹)_KK < 0).  Lane j holds column j: T[i] = a_ij, cc = ĉ_j.
//   a_ij ← a_ij − a_iK·a_Kj / d (i, j ≠ K),   a_Kj ← s·a_Kj / d,   a_KK ← −1/d;   the matrix stays symmetric, so row K is
// register K of every lane, and the new column K (lane K) is that row again: lane K clears its column and takes part in
// the same rank-1 update with the multiplier −s/d.  Returns "the pivot element had the wrong sign" (H is not ≻ 0).
template <int K, int NT, int LP = 16>
__device__ __forceinline__ bool quad_pivot(double (&T)[NT], double& cc, const int l, const bool act, const double s, const bool upper_row = false) {
  const double rowk = T[K];
  double d, ck;
  QuadAB rk{0.0, 0.0};
  if constexpr (LP == 32) {
    rk = quad_ab(rowk, upper_row);
    const QuadAB c2 = quad_ab(cc, upper_row);
    d = quad_bcast<K % 16>(K < 16 ? rk.a : rk.b); ck = quad_bcast<K % 16>(K < 16 ? c2.a : c2.b);
  } else {
    d = quad_bcast<K>(rowk); ck = quad_bcast<K>(cc);
  }
  const bool ok = s * d > 0.0, go = act && ok, isk = l == K;
  const double inv = go ? fast_rcp(go ? d : 1.0) : 0.0;
  const double g = rowk * inv;
  const double gg = isk ? -s * inv : g;
  if (go && isk) {
#pragma unroll
    for (int i = 0; i < NT; ++i) T[i] = 0.0;
    cc = 0.0;
  }
  if constexpr (LP == 32) quad_rank1_ab(T, rk, -gg); else quad_rank1(T, rowk, -gg);
  cc = fma(-ck, gg, cc);
  T[K] = go ? (isk ? -inv : s * g) : T[K];
  return act && !ok;
}

// LOOP: the fused caller loop (mkh_solve_steps / mkh_solve_until; lane_kernel.h, same semantics): every ROW iterates
// (solve, q ← q + Δq) on its own problem until its frame tasks are within the thresholds or the budget is spent; rows
// that are finished idle through the remaining iterations of their wavefront (masked commits).
// NT: column registers per lane = the largest nv the instantiation takes (8: arms; 16: hands, mobile arms).
template <int NT, bool LOOP, int LP = 16>
__global__ __launch_bounds__(64) void ik_quad_kernel(const LaneProblemT<LP, LP>* __restrict__ Pg, const LaneDims D, const SolveArgs A) {
  static_assert((LP == 16 && NT <= 16) || (LP == 32 && NT == 32 && !LOOP), "one DPP row: ≤ 16 column registers; two rows: 32, single solves");
  constexpr int kQuadRow = LP, kQuadPerWave = kWave / LP, kQuadPoseDoubles = 7 * LP, kQuadAncDoubles = LP / 2, kQuadRowDoubles = quad_row_doubles(LP);
  constexpr unsigned long long kRowBits = LP == 32 ? 0xffffffffull : 0xffffull;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const LaneProblemT<LP, LP>& P = *Pg;
  const int lane = (int)threadIdx.x, row = lane / LP, l = lane & (LP - 1), rbase = lane & (64 - LP);
  const bool upper_row = (lane & 16) != 0;           // (two rows per problem: which of the two this lane sits on)
  // a set of lanes as bits of THIS problem / of any problem of the wavefront
  auto own_bits = [&](const unsigned long long m) -> unsigned { return (unsigned)((m >> rbase) & kRowBits); };
  auto any_bits = [&](const unsigned long long m) -> unsigned {
    return LP == 32 ? (unsigned)((m | (m >> 32)) & kRowBits) : (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & kRowBits);
  };
  const int pb_raw = (int)blockIdx.x * kQuadPerWave + row;
  const bool live = pb_raw < A.B;
  const int pb = live ? pb_raw : A.B - 1;            // idle rows of the last wave redo the last problem, store nothing
  const int nv = D.nv, nq = D.nq, nlink = D.nlink, nf = D.n_frame;
  const double kInf = __builtin_huge_val();
  double* const S = smem + row * kQuadRowDoubles;
  double* const sX = S;                               // sX[c·16 + link]
  int* const sAnc = (int*)(S + kQuadPoseDoubles);
  double* const sT = S + kQuadPoseDoubles + kQuadAncDoubles;
  auto row_mask = [&](const bool p) -> unsigned { return own_bits(__ballot(p)); };
  int status = 0;
#ifdef MKH_CLOCKS   // experiment builds (tools/phase_clocks.py): cycle stamps at the phase boundaries, row 24·pb of SolveArgs::clk
  long long tc[8];
  int tci = 0;
#define MKH_QTICK() do { if (!LOOP) { tc[tci++] = __builtin_readcyclecounter(); asm volatile("" : "+v"(status)); } } while (0)
#else
#define MKH_QTICK() do {} while (0)
#endif
  MKH_QTICK();

  // ------------------------------------------------------------------ q (lane = dof)
  const bool dv = l < nv;
  const int ld = dv ? l : 0;
  const int qadr = D.qadr_identity ? ld : P.dof_qadr[ld];
  double qd = dv ? A.q[(size_t)pb * nq + qadr] : 0.0;
  // per-call inputs that do not change over a fused loop: posture targets (lane = dof), frame target (lane = task)
  double ptq[kMaxPostureTasks], pcost[kMaxPostureTasks];
#pragma unroll
  for (int t = 0; t < kMaxPostureTasks; ++t) {
    ptq[t] = 0.0; pcost[t] = 0.0;
    if (t < D.n_posture) {
      const double* tq = A.posture_target + (A.posture_batched ? ((size_t)pb * D.n_posture + t) * nq : (size_t)t * nq);
      ptq[t] = tq[qadr];
      pcost[t] = dv ? P.posture_cost[t][ld] : 0.0;
    }
  }
  const double* const tgp = A.frame_targets + ((size_t)pb * nf + (l < nf ? l : 0)) * 7;
  const SE3 Tt{Q4{tgp[0], tgp[1], tgp[2], tgp[3]}, V3{tgp[4], tgp[5], tgp[6]}};
  const bool until = LOOP && A.pos_threshold >= 0.0;
  const int n_steps = LOOP ? A.n_steps : 1;
  bool fin = false;                                  // this row's loop is over (converged / failed / budget spent)
  int it_done = 0, conv_flag = 0, status_all = 0;
  double vlast = 0.0, x = 0.0;
  int st = 1;                                        // QP partition: 0 free, 1 at lower, 2 at upper (padded lanes: bound at 0, never flip)
  for (int step = 0; step < n_steps + (until ? 1 : 0); ++step) {
  if (LOOP && !__ballot(!fin)) break;
  status = 0;
  // Configuration.check_limits (mink/configuration.py:77-110), tol = 1e-6
  if (row_mask(dv && (qd < P.range_lo[ld] - 1e-6 || qd > P.range_hi[ld] + 1e-6))) status |= 1;
  // (everything below that needs q only comes first: its descriptor loads are in flight together with the links')
  // the link whose joint moves this lane's dof: axis and anchor of the Jacobian column
  const int dl = dv ? P.dof_link[ld] : -1;
  const bool has = dl >= 0;
  const int dla = has ? dl : 0;
  const V3 d_axis{P.dof_axis[ld][0], P.dof_axis[ld][1], P.dof_axis[ld][2]};
  const V3 d_jpos{P.dof_jpos[ld][0], P.dof_jpos[ld][1], P.dof_jpos[ld][2]};
  const bool slide = P.dof_slide[ld] != 0;
  // posture tasks (posture_task.py:87-142): e = target − q, J = −I  (hinge / slide dofs)
  double diag = 0.0, cc = 0.0, mu_total = A.damping;
#pragma unroll
  for (int t = 0; t < kMaxPostureTasks; ++t) {
    if (t < D.n_posture) {
      const double cost = pcost[t];
      const double we = cost * (-P.posture_gain[t] * (ptq[t] - qd));
      diag = fma(cost, cost, diag);
      cc = fma(we, cost, cc);                        // c −= (W·e)·(−cost)
      mu_total += P.posture_lm[t] * qsum<LP>(we * we);
    }
  }
  // box limits
  double lo = 0.0, hi = 0.0;
  if (dv) {
    lo = -kInf; hi = kInf;
    for (int t = 0; t < D.n_cfg; ++t) {              // configuration_limit.py:94-124
      const double lw = P.cfg_lower[t][l], up = P.cfg_upper[t][l];
      if (up < kInf) hi = fmin(hi, P.cfg_gain[t] * (up - qd));
      if (lw > -kInf) lo = fmax(lo, -(P.cfg_gain[t] * (qd - lw)));
    }
    for (int t = 0; t < D.n_vel; ++t) {              // velocity_limit.py:96-101
      const double vm = P.vel_limit[t][l];
      if (vm < kInf) { hi = fmin(hi, A.dt * vm); lo = fmax(lo, -(A.dt * vm)); }
    }
  }
  if (row_mask(dv && lo > hi + 1e-12)) status |= 2;  // quadprog: "constraints are inconsistent"

  MKH_QTICK();
  // ----------------------------------------------------------------- FK (lane = link)
  {
    const bool lv = l < nlink;
    const LaneLink& L = P.link[lv ? l : 0];
    const int jt = L.jtype;
    const double qv = bperm_f64(qd, rbase + (jt >= 0 ? L.dof : 0)) - L.qpos0;
    V3 xp{L.pos[0], L.pos[1], L.pos[2]};
    Q4 xq{L.quat[0], L.quat[1], L.quat[2], L.quat[3]};
    if (LP == 32 && jt == JNT_BALL) {
      // the rotation of a free joint: the body's orientation is the (normalised) quaternion of q (mj_kinematics)
      const double* qq = A.q + (size_t)pb * nq + L.qadr;
      xq = Q4{qq[0], qq[1], qq[2], qq[3]};
    } else if (jt >= 0) {
      const V3 ax{L.axis[0], L.axis[1], L.axis[2]};
      if (jt == JNT_SLIDE) {
        xp = xp + qv * qrot(xq, ax);
      } else {
        const V3 jp{L.jpos[0], L.jpos[1], L.jpos[2]};
        const V3 anchor = xp + qrot(xq, jp);
        xq = qmul(xq, axis_angle(ax, qv));
        xp = anchor - qrot(xq, jp);
      }
    }
    xq = qnormalize(xq);
    // pose relative to the ancestor `anc` (−1: the world); every round composes with that ancestor's pose and moves
    // the pointer to the ancestor's ancestor: depth d → ⌈log₂ d⌉ rounds
    int anc = lv ? L.parent : -1;
    auto put = [&]() {
      sX[0 * kQuadRow + l] = xp.x; sX[1 * kQuadRow + l] = xp.y; sX[2 * kQuadRow + l] = xp.z;
      sX[3 * kQuadRow + l] = xq.w; sX[4 * kQuadRow + l] = xq.x; sX[5 * kQuadRow + l] = xq.y; sX[6 * kQuadRow + l] = xq.z;
      sAnc[l] = anc;
    };
    if (lv) put();
    wave_sync();
    while (__ballot(anc >= 0)) {
      const int a = anc >= 0 ? anc : 0;
      const V3 ap{sX[a], sX[kQuadRow + a], sX[2 * kQuadRow + a]};
      const Q4 aq{sX[3 * kQuadRow + a], sX[4 * kQuadRow + a], sX[5 * kQuadRow + a], sX[6 * kQuadRow + a]};
      const int aanc = sAnc[a];
      wave_sync();                                   // every read of this round before any write
      if (anc >= 0) {
        xp = ap + qrot(aq, xp);
        xq = qmul(aq, xq);                           // (unit × unit: the local quaternions were normalised above)
        anc = aanc;
        put();
      }
      wave_sync();
    }
  }

  MKH_QTICK();
  // ------------------------------------------------- frame tasks (lane = task; frame_task.py:95-146)
  bool far_lane = false;                             // this task lane's frame is outside the callers' thresholds
  if (l < nf) {
    const LaneFrame& ft = P.frame[l];
    SE3 F;
    {
      V3 bp{0, 0, 0};
      Q4 bq{1, 0, 0, 0};
      if (ft.link >= 0) {
        const int a = ft.link;
        bp = V3{sX[a], sX[kQuadRow + a], sX[2 * kQuadRow + a]};
        bq = Q4{sX[3 * kQuadRow + a], sX[4 * kQuadRow + a], sX[5 * kQuadRow + a], sX[6 * kQuadRow + a]};
      }
      F.p = bp + qrot(bq, V3{ft.lpos[0], ft.lpos[1], ft.lpos[2]});
      F.q = qmul(bq, Q4{ft.lquat[0], ft.lquat[1], ft.lquat[2], ft.lquat[3]});
    }
    V3 ev, ew;
    double Jm[9], Qm[9];
    bool ident;
    bool rel = false;
    if constexpr (LP == 32) rel = P.rel[l].relative != 0;
    if (rel) {
      // RelativeFrameTask (relative_frame_task.py:106-142): e = log(T_target⁻¹·T_fr), T_fr = the frame in the root frame;
      // J = jlog(T_tf)·(ᶠJ − Ad(T_fr⁻¹)·ʳJ) — a dof on the chains of both frames drops out, one on the root's chain only enters
      // with the opposite sign, and the column is the FrameTask's formula at the frame's position (see the dof lanes): only
      // the sign of the first block and of jlog's argument differ
      const auto& rr = P.rel[l];
      V3 bp{0, 0, 0};
      Q4 bq{1, 0, 0, 0};
      if (rr.root_link >= 0) {
        const int a = rr.root_link;
        bp = V3{sX[a], sX[kQuadRow + a], sX[2 * kQuadRow + a]};
        bq = Q4{sX[3 * kQuadRow + a], sX[4 * kQuadRow + a], sX[5 * kQuadRow + a], sX[6 * kQuadRow + a]};
      }
      const SE3 R{qmul(bq, Q4{rr.rlquat[0], rr.rlquat[1], rr.rlquat[2], rr.rlquat[3]}), bp + qrot(bq, V3{rr.rlpos[0], rr.rlpos[1], rr.rlpos[2]})};
      se3_log(se3_mul(se3_inv(Tt), se3_mul(se3_inv(R), F)), ev, ew);
      se3_ljacinv(-1.0 * ev, -1.0 * ew, Jm, Qm, ident);            // jlog(T_tf) = ljacinv(−log T_tf)
    } else {
      se3_log(se3_mul(se3_inv(F), Tt), ev, ew);      // e = target.minus(frame)
      se3_ljacinv(ev, ew, Jm, Qm, ident);            // jlog(T_tb) = ljacinv(e)
    }
    const double e6[6] = {ev.x, ev.y, ev.z, ew.x, ew.y, ew.z};
    if (LOOP && until) {
      const double pt = A.pos_threshold, ot = A.ori_threshold;
      far_lane = !((!(ft.rowmask & 7) || dot(ev, ev) <= pt * pt) && (!(ft.rowmask & 56) || dot(ew, ew) <= ot * ot));
    }
    double* const t = sT + l * kQuadTaskDoubles;
    double ss = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double we = ft.cost[r] * (-ft.gain * e6[r]);
      t[21 + r] = we;
      ss += we * we;
    }
    t[27] = ft.lm_damping * ss;
    t[18] = F.p.x; t[19] = F.p.y; t[20] = F.p.z;
    // J = −jlog·ᴮJ with jlog = [[J, −J·Q·J], [0, J]] and ᴮJ = [Rfᵀ·lin; Rfᵀ·ang] (configuration.py:148-153):
    // rows 0-2 = A1·lin + A2·ang, rows 3-5 = A1·ang,  A1 = −J·Rfᵀ,  A2 = J·Q·J·Rfᵀ   (lane_kernel.h, same blocks)
    double A1[9];
    {
      const M3 Rf = qmat(F.q);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double jr = Jm[3 * i] * Rf.m[3 * j] + Jm[3 * i + 1] * Rf.m[3 * j + 1] + Jm[3 * i + 2] * Rf.m[3 * j + 2];
          A1[3 * i + j] = rel ? jr : -jr;
          t[3 * i + j] = A1[3 * i + j];
        }
    }
    double JQ[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) JQ[3 * i + j] = Jm[3 * i] * Qm[j] + Jm[3 * i + 1] * Qm[3 + j] + Jm[3 * i + 2] * Qm[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) t[9 + 3 * i + j] = -(JQ[3 * i] * A1[j] + JQ[3 * i + 1] * A1[3 + j] + JQ[3 * i + 2] * A1[6 + j]);
  }
  wave_sync();
  MKH_QTICK();

  // ------------------------------------------------- objective (lane = dof: column l of H in T[0..8), c_l in cc)
  double T[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) T[i] = 0.0;
  {
    // a joint's axis and anchor are invariant under its own motion: the final body frame gives them
    const int a = dla;
    const Q4 lq{sX[3 * kQuadRow + a], sX[4 * kQuadRow + a], sX[5 * kQuadRow + a], sX[6 * kQuadRow + a]};
    const V3 axw = qrot(lq, d_axis);
    const V3 an = V3{sX[a], sX[kQuadRow + a], sX[2 * kQuadRow + a]} + qrot(lq, d_jpos);
    for (int f = 0; f < nf; ++f) {
      const LaneFrame& ft = P.frame[f];
      const double* const t = sT + f * kQuadTaskDoubles;
      bool on = has && ((ft.chain >> l) & 1u);
      double sgn = 1.0;
      if constexpr (LP == 32) {
        if (P.rel[f].relative) {                     // (uniform) signed membership: frame chain − root chain
          const bool onr = has && ((P.rel[f].rchain >> l) & 1u);
          sgn = (on && !onr) ? 1.0 : -1.0;
          on = on != onr;
        }
      }
      const V3 Fp{t[18], t[19], t[20]};
      const V3 lin = slide ? axw : cross(axw, Fp - an);
      double Jw[6];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double a1l = t[3 * r] * lin.x + t[3 * r + 1] * lin.y + t[3 * r + 2] * lin.z;
        const double a1a = t[3 * r] * axw.x + t[3 * r + 1] * axw.y + t[3 * r + 2] * axw.z;
        const double a2a = t[9 + 3 * r] * axw.x + t[9 + 3 * r + 1] * axw.y + t[9 + 3 * r + 2] * axw.z;
        Jw[r] = on ? (sgn * ft.cost[r]) * (slide ? a1l : a1l + a2a) : 0.0;   // weighted_jacobian (task.py:129)
        Jw[3 + r] = on && !slide ? (sgn * ft.cost[3 + r]) * a1a : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        if ((ft.rowmask >> r) & 1) {                 // (uniform: rows with zero cost contribute nothing)
          cc = fma(-t[21 + r], Jw[r], cc);
          qrank1<LP>(T, Jw[r], Jw[r], upper_row);
        }
      }
      mu_total += t[27];
    }
  }
  if constexpr (LP == 32) {
    if (P.n_com) {
      // ------------------------------------------- ComTask (com_task.py:71-97): e = com − target, J = mj_jacSubtreeCom(body 1)
      // lane = link: m·(world centre of mass) of the link's own mass; then the sums over its subtree — links are in body order,
      // a subtree is a contiguous range of them
      double* const sW = sT + kLaneMaxFrames * kQuadTaskDoubles;       // [3][LP]
      double* const sC = sW + 3 * LP;                                   // [3][LP] subtree centres of mass
      V3 mw{0.0, 0.0, 0.0};
      if (l < nlink) {
        const V3 lp{sX[l], sX[kQuadRow + l], sX[2 * kQuadRow + l]};
        const Q4 lq{sX[3 * kQuadRow + l], sX[4 * kQuadRow + l], sX[5 * kQuadRow + l], sX[6 * kQuadRow + l]};
        mw = P.link_mass[l] * (lp + qrot(lq, V3{P.link_ipos[l][0], P.link_ipos[l][1], P.link_ipos[l][2]}));
      }
      sW[l] = mw.x; sW[LP + l] = mw.y; sW[2 * LP + l] = mw.z;
      const V3 all{qsum<LP>(mw.x) + P.com_static[0], qsum<LP>(mw.y) + P.com_static[1], qsum<LP>(mw.z) + P.com_static[2]};
      wave_sync();
      if (l < nlink) {
        V3 acc{0.0, 0.0, 0.0};
        const int last = P.link_last[l];
        for (int j = l; j <= last; ++j) acc = acc + V3{sW[j], sW[LP + j], sW[2 * LP + j]};
        const double ms = P.link_stmass[l];
        const double inv = ms > 0.0 ? 1.0 / ms : 0.0;
        sC[l] = acc.x * inv; sC[LP + l] = acc.y * inv; sC[2 * LP + l] = acc.z * inv;
      }
      wave_sync();
      const double* ctp = A.com_target + (A.com_batched ? (size_t)pb * 3 : 0);
      const V3 e = P.com_minv * all - V3{ctp[0], ctp[1], ctp[2]};
      // lane = dof: ∂com/∂q_k = (m_subtree / M)·(axis × (c_subtree − anchor)) for a rotation, (m_subtree / M)·axis for a translation
      const int a = dla;
      const Q4 lq{sX[3 * kQuadRow + a], sX[4 * kQuadRow + a], sX[5 * kQuadRow + a], sX[6 * kQuadRow + a]};
      const V3 axw = qrot(lq, d_axis);
      const V3 an = V3{sX[a], sX[kQuadRow + a], sX[2 * kQuadRow + a]} + qrot(lq, d_jpos);
      const V3 cs{sC[a], sC[LP + a], sC[2 * LP + a]};
      const V3 lin = slide ? axw : cross(axw, cs - an);
      const double w = has ? P.link_stmass[a] * P.com_minv : 0.0;
      const double Jc[3] = {w * lin.x, w * lin.y, w * lin.z}, e3[3] = {e.x, e.y, e.z};
      double ss = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if ((P.com_rowmask >> r) & 1) {                  // (uniform)
          const double we = P.com_cost[r] * (-P.com_gain * e3[r]);
          const double Jw = P.com_cost[r] * Jc[r];
          cc = fma(-we, Jw, cc);
          qrank1<LP>(T, Jw, Jw, upper_row);
          ss += we * we;
        }
      }
      mu_total += P.com_lm * ss;
    }
  }
  diag += dv ? mu_total : 1.0;                       // padded dofs: identity, x = 0
#pragma unroll
  for (int i = 0; i < NT; ++i) T[i] += (l == i) ? diag : 0.0;
  double hdiag = 1.0;
#pragma unroll
  for (int i = 0; i < NT; ++i) hdiag = (l == i) ? T[i] : hdiag;

  if (LOOP && until && step > 0 && !fin) {
    it_done = step;
    if (!row_mask(far_lane)) { conv_flag = 1; fin = true; status_all |= status; }     // the callers' break (after the integration)
    else if (step == n_steps) { fin = true; status_all |= status; }                    // budget spent: this pass only tested
  }

  MKH_QTICK();
  // ------------------------------------------------------------------- QP
  const double tolw = 1e-16 * qmax<LP>(dv ? hdiag : 0.0);         // (the wavefront kernel's multiplier threshold)
  bool done = (status & 2) != 0 || (LOOP && fin);
  // Starting partition from the diagonal estimate x_l ≈ −c_l / H_ll: dofs it puts outside the box start AT that bound
  // (any partition is a valid start of block principal pivoting; the benchmark's velocity limits saturate most dofs of
  // most instances, and a dof that starts at its bound saves the pivot in and the pivot out again)
  // (fused loop, from the third step on: the partition the previous step's QP ended with — along an IK loop the active
  //  set changes little from step to step, the diagonal estimate is at its worst when only some dofs saturate:
  //  block principal pivoting then takes 3 iterations on average instead of 1.2, and a wavefront waits for the slowest
  //  of its four rows)
  // (the same across CALLS on one handle — MKH_FLAG_WARM_START, SolveArgs::warm: closed-loop callers solve the same
  //  instances again and again; from the handle's third call on, like the wavefront kernels)
  const bool warm_in = !LOOP && A.warm != nullptr && A.warm_age >= 2;
  if (warm_in) {
    st = dv ? (int)A.warm[(size_t)pb * nv + l] : 1;
    if ((st == 2 && !(hi < kInf)) || (st == 1 && !(lo > -kInf))) st = 0;       // (a bound that is not there any more)
  } else if (!(LOOP && step >= 2)) {
    st = 1;
    if (dv) {
      const double xd = -cc * fast_rcp(hdiag);
      st = xd > hi ? 2 : (xd < lo ? 1 : 0);
    }
  }
  {
    const unsigned long long fr = __ballot(st == 0 && !done);
    const unsigned mine = own_bits(fr), any = any_bits(fr);
    quad_for<NT>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      if ((any >> K) & 1u) { if (quad_pivot<K, NT, LP>(T, cc, l, ((mine >> K) & 1u) != 0, 1.0, upper_row)) { status |= 4; done = true; } }
    });
  }
  MKH_QTICK();
  // Block principal pivoting (mode 0) settles 99.9 % of the rows in a handful of iterations — and leaves a tail: rows on which
  // the block steps stall and Murty's single flips then wander for 20 … 50 iterations (counted on tapped H, c, box of 4 096
  // mobile-arm instances: worst 47), at a batch that fits the chip once the launch waits for that row.  So the first stall
  // hands the row to a MONOTONE method (mode 1): the primal active-set iteration from the clipped point — step towards
  // the free set's Newton point (the tableau's own evaluation), stop at the first bound in the way and put that dof on it
  // (one pivot out); at the Newton point release the bound dof with the worst multiplier (one pivot in), or finish.  The
  // same instances: worst 15 iterations in total.  Murty's rule (mode 2) stays behind it as the finite last resort.
  int best = NT + 1, budget = 3, mode = 0, primal_left = 3 * NT;
  double xc = 0.0;                                   // mode 1: this dof's coordinate of the current feasible point
  for (int it = 0; it < 10 * NT + 10; ++it) {
    if (!__ballot(!done)) break;                     // every row of the wave has its optimum
    const double xb = st == 1 ? lo : (st == 2 ? hi : 0.0);
    const double val = qdot<LP>(T, xb, cc, upper_row);          // −x_l of a free index, the multiplier w_l of a bound one
    const double xn = st ? xb : -val;
    int f = 0;                                       // 0 keep, 1 → lower, 2 → upper, 3 → free
    if (!st) {
      if (lo - xn > 1e-12) f = 1;
      else if (xn - hi > 1e-12) f = 2;
    } else if (dv && ((st == 1 && val < -tolw) || (st == 2 && val > tolw))) {
      f = 3;
    }
    const unsigned fm = row_mask(f != 0);
    // (mode 1: evaluated by every row of a wavefront that has a row in it — the DPP reductions need all lanes — under a
    //  wave-uniform branch, so the common case pays nothing)
    const double dstep = st ? 0.0 : xn - xc;
    double amin = 1.0;
    unsigned blockm = 0, relm = 0;
    if (__ballot(mode == 1 && !done)) {
      const double aj = (f == 2) ? (hi - xc) / dstep : ((f == 1) ? (lo - xc) / dstep : 1.0);   // (|dstep| > 1e-12 where f is 1 or 2)
      amin = qmin<LP>((mode == 1 && !st) ? aj : 1.0);
      const double viol = (dv && st == 1) ? -val : ((dv && st == 2) ? val : 0.0);
      const double vmax = qmax<LP>(viol);
      blockm = row_mask(mode == 1 && !st && f != 0 && aj == amin);
      relm = row_mask(st != 0 && viol == vmax && vmax > tolw);
    }
    bool flip = false;
    if (!done) {
      if (mode != 1) {
        x = xn;
        if (!fm) {
          done = true;
        } else {
          const int cnt = __builtin_popcount(fm), last = 31 - __builtin_clz(fm);
          bool all = true;
          if (cnt < best) { best = cnt; budget = 3; }
          else if (budget > 0) --budget;
          else all = false;
          if (!all && mode == 0) {
            // first stall: the clipped point is feasible; the free dofs it moved onto a bound leave the free set
            mode = 1;
            xc = st ? xb : fmin(fmax(xn, lo), hi);
            flip = f == 1 || f == 2;
          } else {
            flip = f && (all || l == last);          // (mode 2 — Murty: only the infeasible index with the largest number)
          }
        }
      } else {
        if (--primal_left < 0) mode = 2;             // (degenerate zero-length steps could cycle: hand over to Murty)
        if (amin < 1.0) {
          xc = st ? xc : fma(amin, dstep, xc);
          if (l == __builtin_ctz(blockm)) { flip = true; xc = (f == 2) ? hi : lo; }     // the first bound in the way
        } else {
          xc = st ? xc : xn;                         // at the Newton point of this free set
          x = xc;
          if (!relm) done = true;
          else if (l == __builtin_ctz(relm)) { flip = true; f = 3; }                   // release the worst multiplier
        }
      }
    }
    const unsigned long long flips = __ballot(flip);
    const unsigned mine = own_bits(flips), freem = row_mask(st == 0), any = any_bits(flips);
    quad_for<NT>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      if ((any >> K) & 1u) {
        if (quad_pivot<K, NT, LP>(T, cc, l, ((mine >> K) & 1u) != 0, ((freem >> K) & 1u) ? -1.0 : 1.0, upper_row)) { status |= 4; done = true; }
      }
    });
    if (flip) st = (f == 3) ? 0 : f;
  }
  if (!done) status |= 8;
  if (!LOOP && A.warm != nullptr && live && dv) A.warm[(size_t)pb * nv + l] = (int8_t)((status & 14) ? 0 : st);
  if (!LOOP) { status_all = status; break; }
  if (!fin) {                                        // commit this iteration
    status_all |= status;
    if (status & 14) {
      fin = true;                                    // an instance stops at the first step whose QP fails
    } else {
      vlast = x / A.dt;
      qd += x;                                       // mj_integratePos for hinge / slide joints (configuration.py:228-236)
      if (!until) { it_done = step + 1; fin = step + 1 == n_steps; }
    }
  }
  }  // step loop
  MKH_QTICK();
#ifdef MKH_CLOCKS
  if (A.clk && live && l == 0) {
    for (int k = 0; k < 7; ++k) A.clk[(size_t)pb * 24 + k] = tc[k];
    A.clk[(size_t)pb * 24 + 7] = tc[6];
  }
#endif
#undef MKH_QTICK

  // ------------------------------------------------------------------ out
  if (live) {
    if (dv) {
      const double vd = LOOP ? vlast : x / A.dt;                                              // v = Δq / dt (solve_ik.py:104)
      A.v_out[(size_t)pb * nv + l] = (status_all & 14) ? __builtin_nan("") : vd;
      if (LOOP && A.q_out) A.q_out[(size_t)pb * nq + qadr] = qd;
    }
    if (l == 0) {
      if (A.status_out) A.status_out[pb] = status_all;
      if (LOOP && until) {
        if (A.iters_out) A.iters_out[pb] = it_done;
        if (A.converged_out) A.converged_out[pb] = conv_flag;
      }
    }
  }
}

}  // namespace mkh
