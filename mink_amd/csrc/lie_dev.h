// SO3/SE3 device math for the frame-task error and Jacobian.
//
// What it must reproduce (reference, float64):
//   SO3.log        mink/lie/so3.py:176-191      SE3.log      mink/lie/se3.py:159-185
//   SO3.ljacinv    mink/lie/so3.py:214-226      SE3.ljacinv  mink/lie/se3.py:210-218
//   _getQ          mink/lie/se3.py:222-249      jlog         mink/lie/base.py:150-156
//   inverse / multiply / apply of both groups   so3.py:136-151, se3.py:136-157
// including the reference's branch thresholds (1e-10 on θ² resp. θ) so that the
// same inputs take the same branch.  Quaternions are (w,x,y,z).
#pragma once
#include <hip/hip_runtime.h>

namespace mkh {

// 1/d without the IEEE division sequence (v_div_scale ×2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup ≈ 11
// instructions and ≈145 cycles of dependent latency): hardware reciprocal + two Newton steps, ≤ 1 ulp for
// normal d.  The kernel is VALU-issue bound, so every IEEE division that is not part of the reference's
// bit pattern (the final v = Δq/dt is) goes through this.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

struct V3 { double x, y, z; };
struct Q4 { double w, x, y, z; };
struct M3 { double m[9]; };  // row-major

__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// mju_mulQuat (mink/lie/so3.py:150)
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qconj(Q4 a) { return {a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 1e-15) return {1.0, 0.0, 0.0, 0.0};
  double inv = fast_rcp(n);
  return {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
// mju_quat2Mat (mink/lie/so3.py:113)
__device__ __forceinline__ M3 qmat(Q4 q) {
  M3 r;
  double q00 = q.w * q.w, q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z;
  double q11 = q.x * q.x, q12 = q.x * q.y, q13 = q.x * q.z;
  double q22 = q.y * q.y, q23 = q.y * q.z, q33 = q.z * q.z;
  r.m[0] = q00 + q11 - q22 - q33; r.m[4] = q00 - q11 + q22 - q33; r.m[8] = q00 - q11 - q22 + q33;
  r.m[1] = 2 * (q12 - q03); r.m[2] = 2 * (q13 + q02); r.m[3] = 2 * (q12 + q03);
  r.m[5] = 2 * (q23 - q01); r.m[6] = 2 * (q13 - q02); r.m[7] = 2 * (q23 + q01);
  return r;
}
__device__ __forceinline__ V3 mul(const M3& R, V3 v) {
  return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z,
          R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
          R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3 mulT(const M3& R, V3 v) {
  return {R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z,
          R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
          R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}
// rotate v by unit quaternion q:  v + 2w(u×v) + 2u×(u×v)
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
  V3 u{q.x, q.y, q.z};
  V3 t = 2.0 * cross(u, v);
  return v + q.w * t + cross(u, t);
}
// sin/cos for joint-angle-sized arguments (|x| < ~1e5): Cody–Waite reduction by π/2 with
// fma (three-term split of π/2) + the fdlibm minimax kernels on [−π/4, π/4]; < 2 ulp.
// OCML's sincos carries a Payne–Hanek path that costs ~100 VGPRs and scratch in this kernel.
__device__ __forceinline__ void sincos_cw(double x, double* sn, double* cs) {
  const double n = rint(x * 0.63661977236758134308);         // 2/π
  double r = fma(-n, 1.57079632679489655800e+00, x);
  r = fma(-n, 6.12323399573676603587e-17, r);
  r = fma(-n, -1.49738490485916983e-33, r);
  const double z = r * r;
  // __kernel_sin
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                               2.75573137070700676789e-06), -1.98412698298579493134e-04),
                              8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double s0 = fma(r * z, ps, r);
  // __kernel_cos
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                               -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                              -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = ((int)n) & 3;
  const double sv = (q & 1) ? c0 : s0;
  const double cv = (q & 1) ? s0 : c0;
  *sn = (q & 2) ? -sv : sv;
  *cs = ((q + 1) & 2) ? -cv : cv;
}

// mju_axisAngle2Quat
__device__ __forceinline__ Q4 axis_angle(V3 axis, double angle) {
  double s, c;
  sincos_cw(0.5 * angle, &s, &c);
  return {c, axis.x * s, axis.y * s, axis.z * s};
}

// SO3.log (mink/lie/so3.py:176-191), sign-invariant in q.
__device__ __forceinline__ V3 so3_log(Q4 q) {
  const double w = q.w;
  const double norm_sq = q.x * q.x + q.y * q.y + q.z * q.z;
  double factor;
  if (norm_sq < 1e-10) {
    const double iw = fast_rcp(w);
    factor = 2.0 * iw - 2.0 / 3.0 * norm_sq * (iw * iw * iw);
  } else {
    const double nrm = sqrt(norm_sq);
    if (fabs(w) < 1e-10) {
      factor = (w > 0.0 ? 1.0 : -1.0) * M_PI * fast_rcp(nrm);
    } else {
      factor = 2.0 * atan2(w < 0 ? -nrm : nrm, fabs(w)) * fast_rcp(nrm);
    }
  }
  return {factor * q.x, factor * q.y, factor * q.z};
}

// mju_quat2Vel(quat, dt = 1): rotation vector of a (not necessarily unit) quaternion — axis = normalised vector part
// ((1,0,0) when it vanishes), angle = 2·atan2(|vector part|, w) wrapped to (−π, π]
__device__ __forceinline__ V3 quat2vel(Q4 q) {
  double ax[3] = {q.x, q.y, q.z};
  const double sn = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (sn < 1e-15) { ax[0] = 1; ax[1] = 0; ax[2] = 0; } else { ax[0] /= sn; ax[1] /= sn; ax[2] /= sn; }
  double sp = 2.0 * atan2(sn, q.w);
  if (sp > M_PI) sp -= 2.0 * M_PI;
  return V3{ax[0] * sp, ax[1] * sp, ax[2] * sp};
}

struct SE3 { Q4 q; V3 p; };
__device__ __forceinline__ SE3 se3_mul(SE3 a, SE3 b) { return {qmul(a.q, b.q), qrot(a.q, b.p) + a.p}; }
__device__ __forceinline__ SE3 se3_inv(SE3 a) { Q4 qi = qconj(a.q); return {qi, -1.0 * qrot(qi, a.p)}; }

// SE3.log (mink/lie/se3.py:159-185): tangent (v, ω).
__device__ __forceinline__ void se3_log(SE3 T, V3& v, V3& omega) {
  omega = so3_log(T.q);
  const double th2 = dot(omega, omega);
  double k;  // coefficient of [ω]²
  if (th2 < 1e-10) {
    k = 1.0 / 12.0;
  } else {
    const double th = sqrt(th2);
    double s, c;
    sincos_cw(0.5 * th, &s, &c);
    k = (1.0 - th * c * fast_rcp(2.0 * s)) * fast_rcp(th2);
  }
  // V⁻¹ t = t − ½ ω×t + k ω×(ω×t)
  V3 wt = cross(omega, T.p);
  v = T.p - 0.5 * wt + k * cross(omega, wt);
}

// SE3.ljacinv(ξ) (mink/lie/se3.py:210-218) = [[J, −J·Q·J],[0, J]] with J = SO3.ljacinv(ω)
// (so3.py:214-226) and Q = _getQ(ξ) (se3.py:222-249).  Returns J and Q (row-major 3x3);
// `ident` is set when θ² < 1e-10, where the reference returns the 6x6 identity.
//
// Q is evaluated through the vector identities of the skew products in _getQ
//   [w][v][w] = −(w·v)[w],  [v][w] + [w][v] = v wᵀ + w vᵀ − 2(w·v)I,
//   [v][w]² − ([v][w]²)ᵀ = [w×(v×w)] − 2θ²[v],   [w]² = w wᵀ − θ²I,
// i.e. the same matrix with O(10) instead of O(100) live registers.
__device__ __forceinline__ void se3_ljacinv(V3 v, V3 w, double* J, double* Q, bool& ident) {
  const double th2 = dot(w, w);
  ident = th2 < 1e-10;
  if (ident) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { J[i] = (i % 4 == 0) ? 1.0 : 0.0; Q[i] = 0.0; }
    return;
  }
  const double th = sqrt(th2);
  double s, c;
  sincos_cw(th, &s, &c);
  // SO3.ljacinv: I − ½[ω] + A[ω]²   (θ ≥ 1e-5 here, never the θ < 1e-10 Taylor branch)
  const double ith2 = fast_rcp(th2), ith = th * ith2, ith4 = ith2 * ith2;
  const double A = ith2 * (1.0 - (th * s * fast_rcp(2.0 * (1.0 - c))));
  const double Bc = (th - s) * (ith2 * ith);
  const double Cc = (1.0 - th2 * 0.5 - c) * ith4;
  const double Dc = (2.0 * th - 3.0 * s + th * c) * (0.5 * ith4 * ith);
  const double wv[3] = {w.x, w.y, w.z}, vv[3] = {v.x, v.y, v.z};
  const double sv = dot(w, v);
  const V3 u = cross(v, w);
  const V3 wu = cross(w, u);
  // skew-part coefficient vectors:  ½[v] − B·sv[w] − C([wu] − 2θ²[v] + 3sv[w])
  const V3 sk = (0.5 + 2.0 * Cc * th2) * v + (-(Bc + 3.0 * Cc) * sv) * w + (-Cc) * wu;
  const double skv[3] = {sk.x, sk.y, sk.z};
  const double dgq = -2.0 * Bc * sv + 2.0 * Dc * sv * th2;   // coefficient of I
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sym = Bc * (vv[i] * wv[j] + wv[i] * vv[j]) - 2.0 * Dc * sv * (wv[i] * wv[j]);
      double w2 = wv[i] * wv[j];
      double skw = 0.0, skq = 0.0;
      if (i != j) {
        const int k = 3 - i - j;
        const double sg = ((j - i + 3) % 3 == 1) ? -1.0 : 1.0;  // [a]_{ij} = −ε_{ijk} a_k
        skw = sg * wv[k];
        skq = sg * skv[k];
      } else {
        sym += dgq;
        w2 -= th2;
      }
      J[3 * i + j] = ((i == j) ? 1.0 : 0.0) - 0.5 * skw + A * w2;
      Q[3 * i + j] = sym + skq;
    }
}

}  // namespace mkh
