/* See mink_oracle.h.  ORACLE — test infrastructure only.  Operation order follows oracle/{mjmath,lie,ik,qp_gi}.py. */
#include "mink_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
static const double mjMINVAL = 1e-15, mjMAXVAL = 1e10, mjPI = 3.14159265358979323846;
static const double EPS64 = 1e-10; /* mink/lie/utils.py:4-8 */

/* ------------------------------------------------------------------ mju_* (engine_util_spatial.c, restated) */
static double normalize3(double *v) {
  double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n < mjMINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { double inv = 1.0 / n; v[0] *= inv; v[1] *= inv; v[2] *= inv; }
  return n;
}
static double normalize4(double *q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < mjMINVAL) { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; }
  else if (fabs(n - 1.0) > mjMINVAL) { double inv = 1.0 / n; q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv; }
  return n;
}
static void mulQuat(double *res, const double *a, const double *b) {
  double r0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double r1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double r2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double r3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  res[0] = r0; res[1] = r1; res[2] = r2; res[3] = r3;
}
static void negQuat(double *res, const double *q) { res[0] = q[0]; res[1] = -q[1]; res[2] = -q[2]; res[3] = -q[3]; }
static void quat2Mat(double *res, const double *q) {
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  res[0] = q00 + q11 - q22 - q33; res[4] = q00 - q11 + q22 - q33; res[8] = q00 - q11 - q22 + q33;
  res[1] = 2 * (q12 - q03); res[2] = 2 * (q13 + q02); res[3] = 2 * (q12 + q03);
  res[5] = 2 * (q23 - q01); res[6] = 2 * (q13 - q02); res[7] = 2 * (q23 + q01);
}
static void mat2Quat(double *quat, const double *m) {
  if (m[0] + m[4] + m[8] > 0) {
    quat[0] = 0.5 * sqrt(1 + m[0] + m[4] + m[8]);
    quat[1] = 0.25 * (m[7] - m[5]) / quat[0]; quat[2] = 0.25 * (m[2] - m[6]) / quat[0]; quat[3] = 0.25 * (m[3] - m[1]) / quat[0];
  } else if (m[0] > m[4] && m[0] > m[8]) {
    quat[1] = 0.5 * sqrt(1 + m[0] - m[4] - m[8]);
    quat[0] = 0.25 * (m[7] - m[5]) / quat[1]; quat[2] = 0.25 * (m[1] + m[3]) / quat[1]; quat[3] = 0.25 * (m[2] + m[6]) / quat[1];
  } else if (m[4] > m[8]) {
    quat[2] = 0.5 * sqrt(1 - m[0] + m[4] - m[8]);
    quat[0] = 0.25 * (m[2] - m[6]) / quat[2]; quat[1] = 0.25 * (m[1] + m[3]) / quat[2]; quat[3] = 0.25 * (m[5] + m[7]) / quat[2];
  } else {
    quat[3] = 0.5 * sqrt(1 - m[0] - m[4] + m[8]);
    quat[0] = 0.25 * (m[3] - m[1]) / quat[3]; quat[1] = 0.25 * (m[2] + m[6]) / quat[3]; quat[2] = 0.25 * (m[5] + m[7]) / quat[3];
  }
  normalize4(quat);
}
static void axisAngle2Quat(double *res, const double *axis, double angle) {
  if (angle == 0) { res[0] = 1; res[1] = 0; res[2] = 0; res[3] = 0; return; }
  double s = sin(angle * 0.5);
  res[0] = cos(angle * 0.5); res[1] = axis[0] * s; res[2] = axis[1] * s; res[3] = axis[2] * s;
}
static void mulMatVec3(double *res, const double *mat, const double *v) {
  double a = mat[0] * v[0] + mat[1] * v[1] + mat[2] * v[2];
  double b = mat[3] * v[0] + mat[4] * v[1] + mat[5] * v[2];
  double c = mat[6] * v[0] + mat[7] * v[1] + mat[8] * v[2];
  res[0] = a; res[1] = b; res[2] = c;
}
static void rotVecQuat(double *res, const double *vec, const double *quat) {
  if (vec[0] == 0 && vec[1] == 0 && vec[2] == 0) { res[0] = res[1] = res[2] = 0; return; }
  if (quat[0] == 1 && quat[1] == 0 && quat[2] == 0 && quat[3] == 0) { res[0] = vec[0]; res[1] = vec[1]; res[2] = vec[2]; return; }
  double mat[9];
  quat2Mat(mat, quat);
  mulMatVec3(res, mat, vec);
}
static void quat2Vel(double *res, const double *quat, double dt) {
  double axis[3] = {quat[1], quat[2], quat[3]};
  double sin_a_2 = normalize3(axis);
  double speed = 2 * atan2(sin_a_2, quat[0]);
  if (speed > mjPI) speed -= 2 * mjPI;
  speed /= dt;
  res[0] = axis[0] * speed; res[1] = axis[1] * speed; res[2] = axis[2] * speed;
}
static void cross3(double *r, const double *a, const double *b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}

/* ------------------------------------------------------------------ the subset of mjData mink reads */
typedef struct {
  double *xpos, *xquat, *xmat, *xipos, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat,
      *subtree_com, *cdof;
  double *H, *c, *G, *h, *J, *e, *jacp, *jacr, *tmpv, *tmpq;
  double *qp;     /* QP workspace */
  double *dq;     /* (nv) */
  int *idx;       /* (nv) */
  double *pool;
} Work;

static Work *work_new(const MkoModel *m, int mrows) {
  const int nv = m->nv, nb = m->nbody, nj = m->njnt, ng = m->ngeom, ns = m->nsite, nq = m->nq;
  size_t n = 0;
  n += (size_t)nb * (3 + 4 + 9 + 3 + 3) + (size_t)nj * 6 + (size_t)ng * 12 + (size_t)ns * 12 + (size_t)nv * 6;
  n += (size_t)nv * nv + nv + (size_t)mrows * nv + mrows + (size_t)6 * nv * 2 + (size_t)nv * nv + nv + (size_t)6 * nv + 4 * (size_t)nv + 4 * (size_t)nq;
  n += (size_t)4 * nv * nv + (size_t)8 * nv + (size_t)4 * mrows + 64 + nv + (size_t)12 * nv;
  Work *w = (Work *)malloc(sizeof(Work));
  w->pool = (double *)calloc(n, sizeof(double));
  double *p = w->pool;
#define TAKE(field, count) do { w->field = p; p += (count); } while (0)
  TAKE(xpos, nb * 3); TAKE(xquat, nb * 4); TAKE(xmat, nb * 9); TAKE(xipos, nb * 3); TAKE(subtree_com, nb * 3);
  TAKE(xanchor, nj * 3); TAKE(xaxis, nj * 3); TAKE(geom_xpos, ng * 3); TAKE(geom_xmat, ng * 9);
  TAKE(site_xpos, ns * 3); TAKE(site_xmat, ns * 9); TAKE(cdof, nv * 6);
  TAKE(H, nv * nv); TAKE(c, nv); TAKE(G, (size_t)mrows * nv); TAKE(h, mrows);
  TAKE(jacp, 3 * nv); TAKE(jacr, 3 * nv); TAKE(J, (size_t)nv * nv + 18 * nv); TAKE(e, nv + 6);
  TAKE(tmpv, 4 * nv); TAKE(tmpq, 4 * nq);
  TAKE(qp, (size_t)4 * nv * nv + 8 * nv + 4 * mrows + 64);
  TAKE(dq, nv);
#undef TAKE
  w->idx = (int *)calloc((size_t)nv + 1, sizeof(int));
  return w;
}
static void work_free(Work *w) { free(w->idx); free(w->pool); free(w); }

static void local2global(const Work *d, double *xpos, double *xmat, const double *pos, const double *quat, int body) {
  double t[3], q[4];
  mulMatVec3(t, d->xmat + 9 * body, pos);
  xpos[0] = t[0] + d->xpos[3 * body]; xpos[1] = t[1] + d->xpos[3 * body + 1]; xpos[2] = t[2] + d->xpos[3 * body + 2];
  mulQuat(q, d->xquat + 4 * body, quat);
  quat2Mat(xmat, q);
}

/* mj_kinematics (engine_core_smooth.c; mink/configuration.py:63) */
static void kinematics(const MkoModel *m, Work *d, const double *qpos) {
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  memset(d->xmat, 0, 9 * sizeof(double)); d->xmat[0] = d->xmat[4] = d->xmat[8] = 1;
  for (int i = 1; i < m->nbody; ++i) {
    const int jntadr = m->body_jntadr[i], jntnum = m->body_jntnum[i];
    double xpos[3], xquat[4];
    if (jntnum == 1 && m->jnt_type[jntadr] == JNT_FREE) {
      const int qadr = m->jnt_qposadr[jntadr];
      memcpy(xpos, qpos + qadr, 3 * sizeof(double));
      memcpy(xquat, qpos + qadr + 3, 4 * sizeof(double));
      normalize4(xquat);
      memcpy(d->xanchor + 3 * jntadr, xpos, 3 * sizeof(double));
      memcpy(d->xaxis + 3 * jntadr, m->jnt_axis + 3 * jntadr, 3 * sizeof(double));
    } else {
      const int pid = m->body_parentid[i], mid = m->body_mocapid[i];
      double bodyquat[4];
      const double *bodypos;
      if (mid >= 0) {
        bodypos = m->mocap_pos + 3 * mid;
        memcpy(bodyquat, m->mocap_quat + 4 * mid, 4 * sizeof(double));
        normalize4(bodyquat);
      } else {
        bodypos = m->body_pos + 3 * i;
        memcpy(bodyquat, m->body_quat + 4 * i, 4 * sizeof(double));
      }
      if (pid) {
        double t[3];
        mulMatVec3(t, d->xmat + 9 * pid, bodypos);
        for (int k = 0; k < 3; ++k) xpos[k] = t[k] + d->xpos[3 * pid + k];
        mulQuat(xquat, d->xquat + 4 * pid, bodyquat);
      } else {
        memcpy(xpos, bodypos, 3 * sizeof(double));
        memcpy(xquat, bodyquat, 4 * sizeof(double));
      }
      for (int jid = jntadr; jid < jntadr + jntnum; ++jid) {
        const int qadr = m->jnt_qposadr[jid], jtype = m->jnt_type[jid];
        double xaxis[3], xanchor[3], t[3];
        rotVecQuat(xaxis, m->jnt_axis + 3 * jid, xquat);
        rotVecQuat(t, m->jnt_pos + 3 * jid, xquat);
        for (int k = 0; k < 3; ++k) xanchor[k] = t[k] + xpos[k];
        memcpy(d->xaxis + 3 * jid, xaxis, sizeof xaxis);
        memcpy(d->xanchor + 3 * jid, xanchor, sizeof xanchor);
        if (jtype == JNT_SLIDE) {
          const double s = qpos[qadr] - m->qpos0[qadr];
          for (int k = 0; k < 3; ++k) xpos[k] = xpos[k] + xaxis[k] * s;
        } else if (jtype == JNT_BALL || jtype == JNT_HINGE) {
          double qloc[4], vec[3];
          if (jtype == JNT_BALL) { memcpy(qloc, qpos + qadr, 4 * sizeof(double)); normalize4(qloc); }
          else axisAngle2Quat(qloc, m->jnt_axis + 3 * jid, qpos[qadr] - m->qpos0[qadr]);
          mulQuat(xquat, xquat, qloc);
          rotVecQuat(vec, m->jnt_pos + 3 * jid, xquat);
          for (int k = 0; k < 3; ++k) xpos[k] = xanchor[k] - vec[k];
        }
      }
    }
    normalize4(xquat);
    memcpy(d->xquat + 4 * i, xquat, sizeof xquat);
    memcpy(d->xpos + 3 * i, xpos, sizeof xpos);
    quat2Mat(d->xmat + 9 * i, xquat);
  }
  for (int i = 0; i < m->nbody; ++i) {
    double t[3];
    mulMatVec3(t, d->xmat + 9 * i, m->body_ipos + 3 * i);
    for (int k = 0; k < 3; ++k) d->xipos[3 * i + k] = t[k] + d->xpos[3 * i + k];
  }
  for (int g = 0; g < m->ngeom; ++g)
    local2global(d, d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_pos + 3 * g, m->geom_quat + 4 * g, m->geom_bodyid[g]);
  for (int s = 0; s < m->nsite; ++s)
    local2global(d, d->site_xpos + 3 * s, d->site_xmat + 9 * s, m->site_pos + 3 * s, m->site_quat + 4 * s, m->site_bodyid[s]);
}

static void dofCom(double *res, const double *axis, const double *offset) {
  if (offset) { res[0] = axis[0]; res[1] = axis[1]; res[2] = axis[2]; cross3(res + 3, axis, offset); }
  else { res[0] = res[1] = res[2] = 0; res[3] = axis[0]; res[4] = axis[1]; res[5] = axis[2]; }
}

/* mj_comPos: subtree_com and cdof */
static void comPos(const MkoModel *m, Work *d) {
  memset(d->subtree_com, 0, (size_t)m->nbody * 3 * sizeof(double));
  for (int i = m->nbody - 1; i >= 0; --i) {
    for (int k = 0; k < 3; ++k) d->subtree_com[3 * i + k] += d->xipos[3 * i + k] * m->body_mass[i];
    if (i) for (int k = 0; k < 3; ++k) d->subtree_com[3 * m->body_parentid[i] + k] += d->subtree_com[3 * i + k];
    if (m->body_subtreemass[i] < mjMINVAL) {
      for (int k = 0; k < 3; ++k) d->subtree_com[3 * i + k] = d->xipos[3 * i + k];
    } else {
      const double inv = 1.0 / fmax(mjMINVAL, m->body_subtreemass[i]);
      for (int k = 0; k < 3; ++k) d->subtree_com[3 * i + k] = d->subtree_com[3 * i + k] * inv;
    }
  }
  for (int j = 0; j < m->njnt; ++j) {
    const int da = m->jnt_dofadr[j], bi = m->jnt_bodyid[j], jt = m->jnt_type[j];
    double offset[3];
    for (int k = 0; k < 3; ++k) offset[k] = d->subtree_com[3 * m->body_rootid[bi] + k] - d->xanchor[3 * j + k];
    int skip = 0;
    if (jt == JNT_FREE) {
      memset(d->cdof + 6 * da, 0, 18 * sizeof(double));
      for (int i = 0; i < 3; ++i) d->cdof[6 * (da + i) + 3 + i] = 1;
      skip = 3;
    }
    if (jt == JNT_FREE || jt == JNT_BALL) {
      for (int i = 0; i < 3; ++i) {
        const double axis[3] = {d->xmat[9 * bi + i], d->xmat[9 * bi + i + 3], d->xmat[9 * bi + i + 6]};
        dofCom(d->cdof + 6 * (da + skip + i), axis, offset);
      }
    } else if (jt == JNT_SLIDE) {
      dofCom(d->cdof + 6 * da, d->xaxis + 3 * j, NULL);
    } else {
      dofCom(d->cdof + 6 * da, d->xaxis + 3 * j, offset);
    }
  }
}

/* mj_jac: world-aligned point Jacobian (3×nv each, row-major) */
static void jac(const MkoModel *m, const Work *d, double *jacp, double *jacr, const double *point, int body) {
  const int nv = m->nv;
  if (jacp) memset(jacp, 0, (size_t)3 * nv * sizeof(double));
  if (jacr) memset(jacr, 0, (size_t)3 * nv * sizeof(double));
  double offset[3];
  for (int k = 0; k < 3; ++k) offset[k] = point[k] - d->subtree_com[3 * m->body_rootid[body] + k];
  while (body && !m->body_dofnum[body]) body = m->body_parentid[body];
  if (!body) return;
  int i = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  while (i >= 0) {
    const double *cd = d->cdof + 6 * i;
    if (jacr) for (int k = 0; k < 3; ++k) jacr[k * nv + i] = cd[k];
    if (jacp) {
      double t[3];
      cross3(t, cd, offset);
      for (int k = 0; k < 3; ++k) jacp[k * nv + i] = cd[3 + k] + t[k];
    }
    i = m->dof_parentid[i];
  }
}

/* mj_jacSubtreeCom */
static void jacSubtreeCom(const MkoModel *m, const Work *d, double *jacp, double *tmp, int body) {
  const int nv = m->nv;
  memset(jacp, 0, (size_t)3 * nv * sizeof(double));
  for (int b = body; b < m->nbody; ++b) {
    if (b > body && m->body_parentid[b] < body) break;
    jac(m, d, tmp, NULL, d->xipos + 3 * b, b);
    for (int k = 0; k < 3 * nv; ++k) jacp[k] += tmp[k] * m->body_mass[b];
  }
  const double inv = 1.0 / m->body_subtreemass[body];
  for (int k = 0; k < 3 * nv; ++k) jacp[k] *= inv;
}

/* mj_differentiatePos: qvel = (qpos2 ⊖ qpos1)/dt */
static void differentiatePos(const MkoModel *m, double *qvel, double dt, const double *qpos1, const double *qpos2) {
  for (int j = 0; j < m->njnt; ++j) {
    int padr = m->jnt_qposadr[j], vadr = m->jnt_dofadr[j];
    const int jt = m->jnt_type[j];
    if (jt == JNT_FREE) {
      for (int i = 0; i < 3; ++i) qvel[vadr + i] = (qpos2[padr + i] - qpos1[padr + i]) / dt;
      vadr += 3; padr += 3;
    }
    if (jt == JNT_FREE || jt == JNT_BALL) {
      double neg[4], dif[4];
      negQuat(neg, qpos1 + padr);
      mulQuat(dif, neg, qpos2 + padr);
      quat2Vel(qvel + vadr, dif, dt);
    } else {
      qvel[vadr] = (qpos2[padr] - qpos1[padr]) / dt;
    }
  }
}

/* ------------------------------------------------------------------ SO3 / SE3 (mink/lie) on wxyz_xyz[7] */
static void so3_log(double *out, const double *q) { /* so3.py:176-191 */
  const double w = q[0];
  const double norm_sq = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const int use_taylor = norm_sq < EPS64;
  const double norm_safe = use_taylor ? 1.0 : sqrt(norm_sq);
  const double w_safe = use_taylor ? w : 1.0;
  const double atan_n_over_w = atan2(w < 0 ? -norm_safe : norm_safe, fabs(w));
  double f;
  if (use_taylor) f = 2.0 / w_safe - 2.0 / 3.0 * norm_sq / (w_safe * w_safe * w_safe);
  else if (fabs(w) < EPS64) f = (w > 0.0 ? 1.0 : -1.0) * mjPI / norm_safe;
  else f = 2.0 * atan_n_over_w / norm_safe;
  out[0] = f * q[1]; out[1] = f * q[2]; out[2] = f * q[3];
}
static void so3_apply(double *out, const double *q, const double *v) { /* so3.py:143-146: q ⊗ (0,v) ⊗ q⁻¹ */
  const double padded[4] = {0.0, v[0], v[1], v[2]};
  double t[4], qi[4], r[4];
  mulQuat(t, q, padded);
  negQuat(qi, q);
  mulQuat(r, t, qi);
  out[0] = r[1]; out[1] = r[2]; out[2] = r[3];
}
static void se3_inverse(double *out, const double *T) { /* se3.py:136-141 */
  double qi[4], t[3];
  negQuat(qi, T);
  so3_apply(t, qi, T + 4);
  memcpy(out, qi, sizeof qi);
  out[4] = -t[0]; out[5] = -t[1]; out[6] = -t[2];
}
static void se3_multiply(double *out, const double *a, const double *b) { /* se3.py:153-157 */
  double q[4], t[3];
  mulQuat(q, a, b);
  so3_apply(t, a, b + 4);
  memcpy(out, q, sizeof q);
  out[4] = t[0] + a[4]; out[5] = t[1] + a[5]; out[6] = t[2] + a[6];
}
static void skew(double *S, const double *x) {
  S[0] = 0; S[1] = -x[2]; S[2] = x[1]; S[3] = x[2]; S[4] = 0; S[5] = -x[0]; S[6] = -x[1]; S[7] = x[0]; S[8] = 0;
}
static void mat3mul(double *C, const double *A, const double *B) {
  double r[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, r, sizeof r);
}
static void se3_log(double *out, const double *T) { /* se3.py:159-185 → (v, ω) */
  double omega[3], S[9], SS[9], Vinv[9];
  so3_log(omega, T);
  const double th2 = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  const int use_taylor = th2 < EPS64;
  skew(S, omega);
  const double th2s = use_taylor ? 1.0 : th2, ths = sqrt(th2s), half = 0.5 * ths;
  mat3mul(SS, S, S);
  const double k = use_taylor ? 1.0 / 12.0 : (1.0 - ths * cos(half) / (2.0 * sin(half))) / th2s;
  for (int i = 0; i < 9; ++i) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * S[i] + (use_taylor ? SS[i] / 12.0 : k * SS[i]);
  mulMatVec3(out, Vinv, T + 4);
  out[3] = omega[0]; out[4] = omega[1]; out[5] = omega[2];
}
static void so3_ljacinv(double *J, const double *w) { /* so3.py:214-226 */
  const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double A;
  if (theta < EPS64) {
    const double t2 = theta * theta;
    A = (1.0 / 12.0) * (1.0 + t2 / 60.0 * (1.0 + t2 / 42.0 * (1.0 + t2 / 40.0)));
  } else {
    A = (1.0 / (theta * theta)) * (1.0 - (theta * sin(theta) / (2.0 * (1.0 - cos(theta)))));
  }
  double S[9], SS[9];
  skew(S, w);
  mat3mul(SS, S, S);
  for (int i = 0; i < 9; ++i) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * S[i] + A * SS[i];
}
static void getQ(double *Q, const double *c) { /* se3.py:222-249 */
  const double th2 = c[3] * c[3] + c[4] * c[4] + c[5] * c[5];
  const double A = 0.5;
  double B, C, D;
  if (th2 < EPS64) {
    B = (1.0 / 6.0) + (1.0 / 120.0) * th2; C = -(1.0 / 24.0) + (1.0 / 720.0) * th2; D = -(1.0 / 60.0);
  } else {
    const double th = sqrt(th2), s = sin(th), co = cos(th);
    B = (th - s) / (th2 * th);
    C = (1.0 - th2 / 2.0 - co) / (th2 * th2);
    D = (2 * th - 3 * s + th * co) / (2 * th2 * th2 * th);
  }
  double V[9], W[9], VW[9], WV[9], WVW[9], VWW[9], t1[9], t2[9];
  skew(V, c); skew(W, c + 3);
  mat3mul(VW, V, W);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) WV[3 * i + j] = VW[3 * j + i];
  mat3mul(WVW, WV, W);
  mat3mul(VWW, VW, W);
  mat3mul(t1, WVW, W);
  mat3mul(t2, W, WVW);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int k = 3 * i + j, kt = 3 * j + i;
      Q[k] = A * V[k] + B * (WV[k] + VW[k] + WVW[k]) - C * (VWW[k] - VWW[kt] - 3 * WVW[k]) + D * (t1[k] + t2[k]);
    }
}
static void se3_ljacinv(double *J6, const double *xi) { /* se3.py:210-218, 6×6 row-major */
  memset(J6, 0, 36 * sizeof(double));
  if (xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5] < EPS64) { for (int i = 0; i < 6; ++i) J6[7 * i] = 1.0; return; }
  double Q[9], Ji[9], JQ[9], JQJ[9];
  getQ(Q, xi);
  so3_ljacinv(Ji, xi + 3);
  mat3mul(JQ, Ji, Q);
  mat3mul(JQJ, JQ, Ji);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      J6[6 * i + j] = Ji[3 * i + j];
      J6[6 * i + 3 + j] = -JQJ[3 * i + j];
      J6[6 * (i + 3) + 3 + j] = Ji[3 * i + j];
    }
}
static void se3_jlog(double *J6, const double *T) { /* base.py:150-156: ljacinv(−log T) */
  double xi[6];
  se3_log(xi, T);
  for (int i = 0; i < 6; ++i) xi[i] = -xi[i];
  se3_ljacinv(J6, xi);
}

/* ------------------------------------------------------------------ Configuration (mink/configuration.py) */
static void frame_pose(const MkoModel *m, const Work *d, int type, int id, const double **xpos, const double **xmat, int *body) {
  if (type == MKO_FRAME_BODY) { *xpos = d->xpos + 3 * id; *xmat = d->xmat + 9 * id; *body = id; }
  else if (type == MKO_FRAME_GEOM) { *xpos = d->geom_xpos + 3 * id; *xmat = d->geom_xmat + 9 * id; *body = m->geom_bodyid[id]; }
  else { *xpos = d->site_xpos + 3 * id; *xmat = d->site_xmat + 9 * id; *body = m->site_bodyid[id]; }
}
/* get_transform_frame_to_world (:157-185) and get_frame_jacobian (:112-155; 6×nv, rows: linear then angular) */
static void frame_transform_jacobian(const MkoModel *m, Work *d, int type, int id, double *T, double *J6) {
  const double *xpos, *xmat;
  int body;
  const int nv = m->nv;
  frame_pose(m, d, type, id, &xpos, &xmat, &body);
  mat2Quat(T, xmat);
  T[4] = xpos[0]; T[5] = xpos[1]; T[6] = xpos[2];
  jac(m, d, d->jacp, d->jacr, xpos, body);
  double qi[4], R[9];
  negQuat(qi, T);
  quat2Mat(R, qi);                                   /* adjoint of (R_wf⁻¹, 0) = blkdiag(R, R) */
  for (int k = 0; k < nv; ++k) {
    const double p[3] = {d->jacp[k], d->jacp[nv + k], d->jacp[2 * nv + k]};
    const double r[3] = {d->jacr[k], d->jacr[nv + k], d->jacr[2 * nv + k]};
    double a[3], b[3];
    mulMatVec3(a, R, p);
    mulMatVec3(b, R, r);
    for (int i = 0; i < 3; ++i) { J6[i * nv + k] = a[i]; J6[(3 + i) * nv + k] = b[i]; }
  }
}

/* Task.compute_qp_objective (mink/tasks/task.py:105-138): H += JwᵀJw + μI, c += −weᵀJw */
static void add_objective(int nv, int k, const double *J, const double *e, const double *cost, double gain, double lm,
                          double *H, double *c) {
  double we[k > 0 ? k : 1];
  double mu = 0.0;
  for (int r = 0; r < k; ++r) { we[r] = cost[r] * (-gain * e[r]); mu += we[r] * we[r]; }
  mu *= lm;
  for (int i = 0; i < nv; ++i) {
    for (int j = 0; j < nv; ++j) {
      double s = 0.0;
      for (int r = 0; r < k; ++r) s += (cost[r] * J[r * nv + i]) * (cost[r] * J[r * nv + j]);
      H[i * nv + j] += s + (i == j ? mu : 0.0);
    }
    double s = 0.0;
    for (int r = 0; r < k; ++r) s += we[r] * (cost[r] * J[r * nv + i]);
    c[i] += -s;
  }
}

/* ------------------------------------------------------------------ Goldfarb–Idnani (oracle/qp_gi.py) */
static void givens(double a, double b, double *c, double *s, double *h) {
  if (b == 0.0) { *c = 1.0; *s = 0.0; *h = a; return; }
  *h = hypot(a, b); *c = a / *h; *s = b / *h;
}
static int32_t solve_qp_ws(int n, int m, const double *P, const double *q, const double *G, const double *hvec,
                           double *x, double *ws) {
  /* workspace: L[n*n], J[n*n], R[n*n], d[n], z[n], r[n], u[n+1], tmp[n], nrm[m], A[m as double], act flag */
  double *L = ws, *J = L + n * n, *R = J + n * n, *dv = R + n * n, *z = dv + n, *r = z + n, *u = r + n, *tmp = u + n + 1,
         *nrm = tmp + n, *Aidx = nrm + m, *inA = Aidx + m;
  memset(L, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; ++j) {
    double s = P[j * n + j];
    for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0.0)) return 4;
    L[j * n + j] = sqrt(s);
    for (int i = j + 1; i < n; ++i) {
      double t = P[i * n + j];
      for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / L[j * n + j];
    }
  }
  /* J = L^{-T}: column j of L^{-1} by forward substitution, stored as row j of J */
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < n; ++i) tmp[i] = (i == j) ? 1.0 : 0.0;
    for (int i = j; i < n; ++i) {
      double t = tmp[i];
      for (int k = j; k < i; ++k) t -= L[i * n + k] * tmp[k];
      tmp[i] = t / L[i * n + i];
    }
    for (int i = 0; i < n; ++i) J[j * n + i] = tmp[i];      /* Linv[:, j] = tmp ⇒ J = Linvᵀ ⇒ J[j][i] = tmp[i] */
  }
  /* x = −J (Jᵀ q) */
  for (int i = 0; i < n; ++i) { double s = 0; for (int k = 0; k < n; ++k) s += J[k * n + i] * q[k]; tmp[i] = s; }
  for (int i = 0; i < n; ++i) { double s = 0; for (int k = 0; k < n; ++k) s += J[i * n + k] * tmp[k]; x[i] = -s; }
  if (m == 0) return 0;
  for (int i = 0; i < m; ++i) {
    double s = 0;
    for (int k = 0; k < n; ++k) s += G[i * n + k] * G[i * n + k];
    nrm[i] = sqrt(s);
    inA[i] = 0.0;
  }
  memset(R, 0, sizeof(double) * n * n);
  int nact = 0, it = 0;
  const int max_iter = 50 * (n + m);
  const double tol = 1e-12;
  for (;;) {
    /* step 1: most violated constraint, n_i = −G_i, b_i = −h_i */
    double best = 0.0;
    int p = -1;
    for (int i = 0; i < m; ++i) {
      if (inA[i] != 0.0 || !isfinite(hvec[i]) || nrm[i] == 0.0) continue;
      double s = 0;
      for (int k = 0; k < n; ++k) s += -G[i * n + k] * x[k];
      s -= -hvec[i];
      const double v = s / nrm[i];
      if (v < -tol * fmax(1.0, fabs(hvec[i]) / nrm[i]) && v < best) { best = v; p = i; }
    }
    if (p < 0) break;
    u[nact] = 0.0;
    for (;;) {
      if (++it > max_iter) return 8;
      double dd = 0.0, dd2 = 0.0;
      for (int i = 0; i < n; ++i) {      /* d = Jᵀ n_p */
        double s = 0;
        for (int k = 0; k < n; ++k) s += J[k * n + i] * (-G[p * n + k]);
        dv[i] = s; dd += s * s;
        if (i >= nact) dd2 += s * s;
      }
      for (int i = 0; i < n; ++i) {      /* z = J[:, nact:] d[nact:] */
        double s = 0;
        for (int k = nact; k < n; ++k) s += J[i * n + k] * dv[k];
        z[i] = s;
      }
      for (int i = nact - 1; i >= 0; --i) {   /* R r = d1 */
        double s = dv[i];
        for (int k = i + 1; k < nact; ++k) s -= R[i * n + k] * r[k];
        r[i] = s / R[i * n + i];
      }
      double t1 = INFINITY, t2 = INFINITY;
      int l = -1;
      for (int k = 0; k < nact; ++k)
        if (r[k] > 0.0) { const double tk = u[k] / r[k]; if (tk < t1) { t1 = tk; l = k; } }
      if (dd2 > 1e-24 * dd) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += -G[p * n + k] * x[k];
        t2 = -(s - (-hvec[p])) / dd2;
      }
      const double t = fmin(t1, t2);
      if (!isfinite(t)) return 2;
      const int dual_only = !isfinite(t2);
      if (!dual_only) for (int i = 0; i < n; ++i) x[i] = x[i] + t * z[i];
      for (int k = 0; k < nact; ++k) u[k] -= t * r[k];
      u[nact] += t;
      if (!dual_only && t2 <= t1) {
        /* add p: rotate d[nact:] onto its first component, same rotations on the columns of J */
        for (int k = n - 1; k > nact; --k) {
          double c, s, hh;
          givens(dv[k - 1], dv[k], &c, &s, &hh);
          if (s == 0.0) continue;
          dv[k - 1] = hh; dv[k] = 0.0;
          for (int i = 0; i < n; ++i) {
            const double a = J[i * n + k - 1], b = J[i * n + k];
            J[i * n + k - 1] = c * a + s * b;
            J[i * n + k] = -s * a + c * b;
          }
        }
        for (int i = 0; i <= nact; ++i) R[i * n + nact] = dv[i];
        Aidx[nact] = (double)p; inA[p] = 1.0;
        ++nact;
        break;
      }
      /* drop blocking constraint at position l */
      inA[(int)Aidx[l]] = 0.0;
      for (int k = l; k < nact - 1; ++k) {
        for (int i = 0; i < n; ++i) R[i * n + k] = R[i * n + k + 1];
        u[k] = u[k + 1];
        Aidx[k] = Aidx[k + 1];
      }
      u[nact - 1] = u[nact];
      for (int i = 0; i < n; ++i) R[i * n + nact - 1] = 0.0;
      --nact;
      for (int k = l; k < nact; ++k) {
        double c, s, hh;
        givens(R[k * n + k], R[(k + 1) * n + k], &c, &s, &hh);
        if (s == 0.0) continue;
        for (int j = k; j < nact; ++j) {
          const double a = R[k * n + j], b = R[(k + 1) * n + j];
          R[k * n + j] = c * a + s * b;
          R[(k + 1) * n + j] = -s * a + c * b;
        }
        R[(k + 1) * n + k] = 0.0;
        for (int i = 0; i < n; ++i) {
          const double a = J[i * n + k], b = J[i * n + k + 1];
          J[i * n + k] = c * a + s * b;
          J[i * n + k + 1] = -s * a + c * b;
        }
      }
    }
  }
  return 0;
}

int32_t mko_solve_qp(int32_t n, int32_t m, const double *P, const double *q, const double *G, const double *h,
                     double *x_out) {
  double *ws = (double *)calloc((size_t)3 * n * n + 6 * n + 3 * m + 16, sizeof(double));
  const int32_t rc = solve_qp_ws(n, m, P, q, G, h, x_out, ws);
  free(ws);
  return rc;
}


/* ------------------------------------------------------------------ CollisionAvoidanceLimit rows
 * mink/limits/collision_avoidance_limit.py:187-229 on top of mj_geomDistance (third-party, restated from the published
 * pair routines — engine_collision_primitive.c: mjraw_SphereSphere, mjc_CapsuleCapsule, mjraw_SphereCapsule,
 * mjc_PlaneSphere, mjc_PlaneCapsule — in the operation order of oracle/mjmath.py:375-462, 768-833; PARITY UNPINNED
 * against the wheel).  Plane / sphere / capsule pairs, and (round 5) box against plane / sphere / capsule and cylinder against
 * plane / sphere / capsule — mjc_PlaneBox, mjc_PlaneCylinder, mjc_SphereBox, mjc_SphereCylinder, mjc_CapsuleBox, in the
 * operation order of oracle/mjmath.py:469-633, 738-765 (which carries the tie rules). */
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_CYLINDER = 5, GEOM_BOX = 6 };
typedef struct { double dist, pos[3], n[3]; } Con;

static int sphere_sphere(Con *out, const double *p1, double r1, const double *p2, double r2, double margin) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const double cdist = sqrt(dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2]);
  const double dist = cdist - r1 - r2;
  if (dist > margin) return 0;
  double n[3];
  if (cdist < mjMINVAL) { n[0] = 1; n[1] = 0; n[2] = 0; }
  else for (int k = 0; k < 3; ++k) n[k] = dif[k] / cdist;
  out->dist = dist;
  for (int k = 0; k < 3; ++k) { out->n[k] = n[k]; out->pos[k] = p1[k] + n[k] * (r1 + 0.5 * dist); }
  return 1;
}

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static int capsule_capsule(Con *out, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                           const double *mat2, const double *size2, double margin) {
  const double axis1[3] = {mat1[2], mat1[5], mat1[8]}, axis2[3] = {mat2[2], mat2[5], mat2[8]};
  const double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  const double ma = axis1[0] * axis1[0] + axis1[1] * axis1[1] + axis1[2] * axis1[2];
  const double mb = -(axis1[0] * axis2[0] + axis1[1] * axis2[1] + axis1[2] * axis2[2]);
  const double mc = axis2[0] * axis2[0] + axis2[1] * axis2[1] + axis2[2] * axis2[2];
  const double u = -(axis1[0] * dif[0] + axis1[1] * dif[1] + axis1[2] * dif[2]);
  const double v = axis2[0] * dif[0] + axis2[1] * dif[1] + axis2[2] * dif[2];
  const double det = ma * mc - mb * mb;
  int n = 0;
  if (fabs(det) >= mjMINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > size1[1]) { x1 = size1[1]; x2 = (v - mb * size1[1]) / mc; }
    else if (x1 < -size1[1]) { x1 = -size1[1]; x2 = (v + mb * size1[1]) / mc; }
    if (x2 > size2[1]) { x2 = size2[1]; x1 = (u - mb * size2[1]) / ma; }
    else if (x2 < -size2[1]) { x2 = -size2[1]; x1 = (u + mb * size2[1]) / ma; }
    x1 = clampd(x1, -size1[1], size1[1]);
    x2 = clampd(x2, -size2[1], size2[1]);
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) { a[k] = pos1[k] + axis1[k] * x1; b[k] = pos2[k] + axis2[k] * x2; }
    n += sphere_sphere(out + n, a, size1[0], b, size2[0], margin);
  } else {
    /* parallel axes: both ends of each capsule against the other segment, the first two that project inside */
    double cx1[4], cx2[4];
    int nc = 0;
    for (int s = 0; s < 2; ++s) {
      const double x1 = (s ? -1.0 : 1.0) * size1[1], x2 = (v - mb * x1) / mc;
      if (-size2[1] <= x2 && x2 <= size2[1]) { cx1[nc] = x1; cx2[nc] = x2; ++nc; }
    }
    for (int s = 0; s < 2; ++s) {
      const double x2 = (s ? -1.0 : 1.0) * size2[1], x1 = (u - mb * x2) / ma;
      if (-size1[1] <= x1 && x1 <= size1[1]) { cx1[nc] = x1; cx2[nc] = x2; ++nc; }
    }
    for (int i = 0; i < nc && i < 2; ++i) {
      double a[3], b[3];
      for (int k = 0; k < 3; ++k) { a[k] = pos1[k] + axis1[k] * cx1[i]; b[k] = pos2[k] + axis2[k] * cx2[i]; }
      n += sphere_sphere(out + n, a, size1[0], b, size2[0], margin);
    }
  }
  return n;
}

static int sphere_capsule(Con *out, const double *pos1, const double *size1, const double *pos2, const double *mat2,
                          const double *size2, double margin) {
  const double axis[3] = {mat2[2], mat2[5], mat2[8]};
  double x = axis[0] * (pos1[0] - pos2[0]) + axis[1] * (pos1[1] - pos2[1]) + axis[2] * (pos1[2] - pos2[2]);
  x = clampd(x, -size2[1], size2[1]);
  const double b[3] = {pos2[0] + axis[0] * x, pos2[1] + axis[1] * x, pos2[2] + axis[2] * x};
  return sphere_sphere(out, pos1, size1[0], b, size2[0], margin);
}

static int plane_sphere(Con *out, const double *pos1, const double *mat1, const double *pos2, double r2, double margin) {
  const double n[3] = {mat1[2], mat1[5], mat1[8]};
  const double cdist = n[0] * (pos2[0] - pos1[0]) + n[1] * (pos2[1] - pos1[1]) + n[2] * (pos2[2] - pos1[2]);
  const double dist = cdist - r2;
  if (dist > margin) return 0;
  out->dist = dist;
  for (int k = 0; k < 3; ++k) { out->n[k] = n[k]; out->pos[k] = pos2[k] - n[k] * (r2 + 0.5 * dist); }
  return 1;
}

static int plane_capsule(Con *out, const double *pos1, const double *mat1, const double *pos2, const double *mat2,
                         const double *size2, double margin) {
  const double axis[3] = {mat2[2], mat2[5], mat2[8]};
  int n = 0;
  for (int s = 0; s < 2; ++s) {
    const double sg = s ? -1.0 : 1.0;
    const double e[3] = {pos2[0] + axis[0] * sg * size2[1], pos2[1] + axis[1] * sg * size2[1], pos2[2] + axis[2] * sg * size2[1]};
    n += plane_sphere(out + n, pos1, mat1, e, size2[0], margin);
  }
  return n;
}

/* R (row-major 3×3) and its transpose applied to a vector */
static void matT_vec(double *r, const double *R, const double *v) {
  for (int k = 0; k < 3; ++k) r[k] = R[k] * v[0] + R[3 + k] * v[1] + R[6 + k] * v[2];
}
static void mat_vec(double *r, const double *R, const double *v) {
  for (int k = 0; k < 3; ++k) r[k] = R[3 * k] * v[0] + R[3 * k + 1] * v[1] + R[3 * k + 2] * v[2];
}
/* contacts found in the frame of geom 2 → world (oracle/mjmath.py::_to_world) */
static void con_to_world(Con *c, int n, const double *R, const double *o) {
  for (int i = 0; i < n; ++i) {
    double p[3], nn[3];
    mat_vec(p, R, c[i].pos); mat_vec(nn, R, c[i].n);
    for (int k = 0; k < 3; ++k) { c[i].pos[k] = o[k] + p[k]; c[i].n[k] = nn[k]; }
  }
}

/* mjc_PlaneBox: the lowest corner; ties towards the −size corner (oracle/mjmath.py:469-479) */
static int plane_box(Con *out, const double *pos1, const double *mat1, const double *pos2, const double *mat2, const double *size2,
                     double margin) {
  const double n[3] = {mat1[2], mat1[5], mat1[8]};
  double nb[3], vec[3], rv[3];
  matT_vec(nb, mat2, n);
  for (int k = 0; k < 3; ++k) vec[k] = nb[k] < 0.0 ? size2[k] : -size2[k];
  const double dist = (n[0] * (pos2[0] - pos1[0]) + n[1] * (pos2[1] - pos1[1]) + n[2] * (pos2[2] - pos1[2])) +
                      (nb[0] * vec[0] + nb[1] * vec[1] + nb[2] * vec[2]);
  if (dist > margin) return 0;
  mat_vec(rv, mat2, vec);
  out->dist = dist;
  for (int k = 0; k < 3; ++k) { out->n[k] = n[k]; out->pos[k] = pos2[k] + rv[k] - n[k] * (0.5 * dist); }
  return 1;
}

/* mjc_PlaneCylinder: the lowest rim point; a cap parallel to the plane ties to its centre (oracle/mjmath.py:482-495) */
static int plane_cylinder(Con *out, const double *pos1, const double *mat1, const double *pos2, const double *mat2,
                          const double *size2, double margin) {
  const double n[3] = {mat1[2], mat1[5], mat1[8]}, axis[3] = {mat2[2], mat2[5], mat2[8]};
  const double c = n[0] * axis[0] + n[1] * axis[1] + n[2] * axis[2];
  double radial[3], pt[3];
  for (int k = 0; k < 3; ++k) radial[k] = n[k] - c * axis[k];
  const double rl = sqrt(radial[0] * radial[0] + radial[1] * radial[1] + radial[2] * radial[2]);
  for (int k = 0; k < 3; ++k) pt[k] = pos2[k] - axis[k] * (c < 0.0 ? -size2[1] : size2[1]);
  if (rl > mjMINVAL) for (int k = 0; k < 3; ++k) pt[k] = pt[k] - radial[k] * (size2[0] / rl);
  const double dist = n[0] * (pt[0] - pos1[0]) + n[1] * (pt[1] - pos1[1]) + n[2] * (pt[2] - pos1[2]);
  if (dist > margin) return 0;
  out->dist = dist;
  for (int k = 0; k < 3; ++k) { out->n[k] = n[k]; out->pos[k] = pt[k] - n[k] * (0.5 * dist); }
  return 1;
}

/* ball of radius r centred at p (box frame) against the box ±s (mjc_SphereBox; oracle/mjmath.py:527-548) */
static int ball_box_local(Con *out, const double *p, double r, const double *s, double margin) {
  double d[3];
  for (int k = 0; k < 3; ++k) d[k] = clampd(p[k], -s[k], s[k]) - p[k];
  const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (dl - r > margin) return 0;
  if (dl > mjMINVAL) {
    const double dist = dl - r;
    out->dist = dist;
    for (int k = 0; k < 3; ++k) { out->n[k] = d[k] / dl; out->pos[k] = p[k] + out->n[k] * (r + 0.5 * dist); }
    return 1;
  }
  const double face[3] = {s[0] - fabs(p[0]), s[1] - fabs(p[1]), s[2] - fabs(p[2])};
  int k = 0;
  for (int i = 1; i < 3; ++i) if (face[i] < face[k]) k = i;
  double n[3] = {0, 0, 0};
  n[k] = p[k] <= 0.0 ? 1.0 : -1.0;
  const double closest = face[k];
  out->dist = -closest - r;
  for (int i = 0; i < 3; ++i) { out->n[i] = n[i]; out->pos[i] = p[i] + n[i] * (0.5 * (r - closest)); }
  return 1;
}

/* ball against a cylinder in its own frame (mjc_SphereCylinder; oracle/mjmath.py:591-611) */
static int ball_cylinder_local(Con *out, const double *p, double r, double rad, double half, double margin) {
  const double rho = hypot(p[0], p[1]);
  const double sc = rho > rad ? rad / rho : 1.0;
  const double cl[3] = {p[0] * sc, p[1] * sc, clampd(p[2], -half, half)};
  const double d[3] = {cl[0] - p[0], cl[1] - p[1], cl[2] - p[2]};
  const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (dl - r > margin) return 0;
  if (dl > mjMINVAL) {
    const double dist = dl - r;
    out->dist = dist;
    for (int k = 0; k < 3; ++k) { out->n[k] = d[k] / dl; out->pos[k] = p[k] + out->n[k] * (r + 0.5 * dist); }
    return 1;
  }
  const double fr = rad - rho, fz = half - fabs(p[2]);
  double closest, n[3];
  if (fz < fr) { closest = fz; n[0] = 0; n[1] = 0; n[2] = p[2] <= 0.0 ? 1.0 : -1.0; }
  else {
    closest = fr;
    if (rho > mjMINVAL) { n[0] = -p[0] / rho; n[1] = -p[1] / rho; n[2] = 0; } else { n[0] = -1; n[1] = 0; n[2] = 0; }
  }
  out->dist = -closest - r;
  for (int k = 0; k < 3; ++k) { out->n[k] = n[k]; out->pos[k] = p[k] + n[k] * (0.5 * (r - closest)); }
  return 1;
}

static int sphere_box(Con *out, const double *pos1, const double *size1, const double *pos2, const double *mat2,
                      const double *size2, double margin) {
  const double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double p[3];
  matT_vec(p, mat2, dif);
  const int n = ball_box_local(out, p, size1[0], size2, margin);
  con_to_world(out, n, mat2, pos2);
  return n;
}

static int sphere_cylinder(Con *out, const double *pos1, const double *size1, const double *pos2, const double *mat2,
                           const double *size2, double margin) {
  const double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double p[3];
  matT_vec(p, mat2, dif);
  const int n = ball_cylinder_local(out, p, size1[0], size2[0], size2[1], margin);
  con_to_world(out, n, mat2, pos2);
  return n;
}

/* argmin over |t| ≤ l of dist(c + t·a, box ±s): root of the piecewise-linear derivative over its sorted breakpoints
 * (oracle/mjmath.py:551-588, including its flat-stretch and rounding rules) */
static double seg_box_g(const double *c, const double *a, const double *s, double t, int on, double tol) {
  double v = 0.0;
  for (int i = 0; i < 3; ++i) {
    const double p = c[i] + t * a[i];
    double e = p - clampd(p, -s[i], s[i]);
    if (i == on) e = 0.0;
    v += a[i] * e;
  }
  return fabs(v) <= tol ? 0.0 : v;
}
static double seg_box_param(const double *c, const double *a, double l, const double *s) {
  double cm = fmax(fabs(c[0]), fmax(fabs(c[1]), fabs(c[2]))), sm = fmax(s[0], fmax(s[1], s[2]));
  const double tol = 1e-13 * (l + cm + sm);
  double ts[8], gs[8];
  int n = 0;
  ts[n] = -l; gs[n] = seg_box_g(c, a, s, -l, -1, tol); ++n;
  ts[n] = l; gs[n] = seg_box_g(c, a, s, l, -1, tol); ++n;
  for (int i = 0; i < 3; ++i)
    if (fabs(a[i]) >= mjMINVAL)
      for (int sg = 0; sg < 2; ++sg) {
        const double e = sg ? s[i] : -s[i], tb = (e - c[i]) / a[i];
        if (-l < tb && tb < l) { ts[n] = tb; gs[n] = seg_box_g(c, a, s, tb, i, tol); ++n; }
      }
  for (int i = 1; i < n; ++i) {                     /* stable insertion sort by t (Python's list.sort is stable) */
    const double t = ts[i], g = gs[i];
    int j = i - 1;
    while (j >= 0 && ts[j] > t) { ts[j + 1] = ts[j]; gs[j + 1] = gs[j]; --j; }
    ts[j + 1] = t; gs[j + 1] = g;
  }
  if (gs[0] > 0.0) return -l;
  if (gs[n - 1] < 0.0) return l;
  int lo = 0, hi = n - 1;
  for (int i = 0; i < n; ++i) if (gs[i] <= 0.0) lo = i;              /* last point with g ≤ 0 */
  for (int i = n - 1; i >= 0; --i) if (gs[i] >= 0.0) hi = i;         /* first point with g ≥ 0 */
  const double tL = ts[lo], gL = gs[lo], tR = ts[hi], gR = gs[hi];
  if (gR - gL > 0.0) return tL + (tR - tL) * (-gL / (gR - gL));
  return 0.5 * (tL + tR);
}

static int capsule_box(Con *out, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                       const double *mat2, const double *size2, double margin) {
  const double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]}, ax[3] = {mat1[2], mat1[5], mat1[8]};
  double c[3], a[3], p[3];
  matT_vec(c, mat2, dif); matT_vec(a, mat2, ax);
  const double t = seg_box_param(c, a, size1[1], size2);
  for (int k = 0; k < 3; ++k) p[k] = c[k] + t * a[k];
  const int n = ball_box_local(out, p, size1[0], size2, margin);
  con_to_world(out, n, mat2, pos2);
  return n;
}

/* capsule against cylinder: 64 bisection steps on the derivative of the squared distance along the capsule axis, then
 * ball-against-cylinder (oracle/mjmath.py:738-765) */
static double cap_cyl_g(const double *c, const double *a, double t, double rad, double half) {
  const double p[3] = {c[0] + t * a[0], c[1] + t * a[1], c[2] + t * a[2]};
  const double rho = hypot(p[0], p[1]), sc = rho > rad ? rad / rho : 1.0;
  const double cl[3] = {p[0] * sc, p[1] * sc, clampd(p[2], -half, half)};
  return a[0] * (p[0] - cl[0]) + a[1] * (p[1] - cl[1]) + a[2] * (p[2] - cl[2]);
}
static int capsule_cylinder(Con *out, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                            const double *mat2, const double *size2, double margin) {
  const double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]}, ax[3] = {mat1[2], mat1[5], mat1[8]};
  double c[3], a[3], p[3];
  matT_vec(c, mat2, dif); matT_vec(a, mat2, ax);
  const double l = size1[1], rad = size2[0], half = size2[1];
  double lo = -l, hi = l, t;
  if (cap_cyl_g(c, a, lo, rad, half) >= 0.0) t = lo;
  else if (cap_cyl_g(c, a, hi, rad, half) <= 0.0) t = hi;
  else {
    for (int it = 0; it < 64; ++it) {
      const double mid = 0.5 * (lo + hi);
      if (cap_cyl_g(c, a, mid, rad, half) < 0.0) lo = mid; else hi = mid;
    }
    t = hi;
  }
  for (int k = 0; k < 3; ++k) p[k] = c[k] + t * a[k];
  const int n = ball_cylinder_local(out, p, size1[0], rad, half, margin);
  con_to_world(out, n, mat2, pos2);
  return n;
}

/* mj_geomDistance (collision_avoidance_limit.py:219): smallest signed distance, fromto = the connecting segment;
 * returns distmax (fromto zeroed) when nothing is closer.  *err set for a pair type outside the restated set. */
static double geom_distance(const MkoModel *m, const Work *d, int g1, int g2, double distmax, double *fromto, int *err) {
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  const int flip = t1 > t2;
  if (flip) { int t = g1; g1 = g2; g2 = t; t = t1; t1 = t2; t2 = t; }
  const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
  const double *R1 = d->geom_xmat + 9 * g1, *R2 = d->geom_xmat + 9 * g2;
  const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
  Con cons[2];
  int n;
  if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) n = capsule_capsule(cons, p1, R1, s1, p2, R2, s2, distmax);
  else if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) n = sphere_sphere(cons, p1, s1[0], p2, s2[0], distmax);
  else if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) n = sphere_capsule(cons, p1, s1, p2, R2, s2, distmax);
  else if (t1 == GEOM_PLANE && t2 == GEOM_SPHERE) n = plane_sphere(cons, p1, R1, p2, s2[0], distmax);
  else if (t1 == GEOM_PLANE && t2 == GEOM_CAPSULE) n = plane_capsule(cons, p1, R1, p2, R2, s2, distmax);
  else if (t1 == GEOM_PLANE && t2 == GEOM_BOX) n = plane_box(cons, p1, R1, p2, R2, s2, distmax);
  else if (t1 == GEOM_PLANE && t2 == GEOM_CYLINDER) n = plane_cylinder(cons, p1, R1, p2, R2, s2, distmax);
  else if (t1 == GEOM_SPHERE && t2 == GEOM_BOX) n = sphere_box(cons, p1, s1, p2, R2, s2, distmax);
  else if (t1 == GEOM_SPHERE && t2 == GEOM_CYLINDER) n = sphere_cylinder(cons, p1, s1, p2, R2, s2, distmax);
  else if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX) n = capsule_box(cons, p1, R1, s1, p2, R2, s2, distmax);
  else if (t1 == GEOM_CAPSULE && t2 == GEOM_CYLINDER) n = capsule_cylinder(cons, p1, R1, s1, p2, R2, s2, distmax);
  else { *err = 1; n = 0; }
  for (int k = 0; k < 6; ++k) fromto[k] = 0.0;
  if (!n) return distmax;
  const Con *c = (n == 2 && cons[1].dist < cons[0].dist) ? cons + 1 : cons;      /* first minimum */
  const double sg = flip ? -1.0 : 1.0;
  for (int k = 0; k < 3; ++k) {
    fromto[k] = c->pos[k] - c->n[k] * (0.5 * sg * c->dist);
    fromto[3 + k] = c->pos[k] + c->n[k] * (0.5 * sg * c->dist);
  }
  return c->dist;
}

/* one limit's rows after kinematics + comPos: G (n_pairs, nv) zero-initialised by the caller, h (n_pairs) */
static int collision_rows(const MkoModel *m, Work *w, const MkoCollisionLimit *c, double dt, double *G, double *h) {
  const int nv = m->nv;
  int err = 0;
  for (int k = 0; k < c->n_pairs; ++k) {
    const int g1 = c->pairs[2 * k], g2 = c->pairs[2 * k + 1];
    double fromto[6];
    h[k] = INFINITY;
    const double dist = geom_distance(m, w, g1, g2, c->detection_distance, fromto, &err);
    if (dist == c->detection_distance) continue;                       /* Contact.inactive (:53-56) */
    if (dist > c->minimum_distance) h[k] = c->gain * (dist - c->minimum_distance) / dt + c->bound_relaxation;
    else h[k] = c->bound_relaxation;
    double normal[3] = {fromto[3] - fromto[0], fromto[4] - fromto[1], fromto[5] - fromto[2]};
    normalize3(normal);                                                /* Contact.normal (:44-50) */
    jac(m, w, w->jacp, NULL, fromto + 3, m->geom_bodyid[g2]);          /* compute_contact_normal_jacobian (:59-72) */
    jac(m, w, w->jacr, NULL, fromto, m->geom_bodyid[g1]);
    for (int j = 0; j < nv; ++j) {
      double s = 0;
      for (int r = 0; r < 3; ++r) s += normal[r] * (w->jacp[r * nv + j] - w->jacr[r * nv + j]);
      G[(size_t)k * nv + j] = -s;
    }
  }
  return err ? -2 : 0;
}

int32_t mko_collision_rows(const MkoModel *m, const MkoCollisionLimit *c, const double *q, double dt, double *G_out,
                           double *h_out) {
  Work *w = work_new(m, 1);
  kinematics(m, w, q);
  comPos(m, w);
  memset(G_out, 0, sizeof(double) * (size_t)c->n_pairs * m->nv);
  const int rc = collision_rows(m, w, c, dt, G_out, h_out);
  work_free(w);
  return rc;
}

/* ------------------------------------------------------------------ solve_ik */
static int count_rows(const MkoModel *m, const MkoProblem *p) {
  int rows = 2 * p->n_vel;
  if (p->has_cfg_limit)
    for (int j = 0; j < m->njnt; ++j)
      if (m->jnt_type[j] != JNT_FREE && m->jnt_limited[j]) rows += 2 * (m->jnt_type[j] == JNT_BALL ? 3 : 1);
  for (int l = 0; l < p->n_coll; ++l) rows += p->coll[l].n_pairs;
  rows += p->n_dense_limit_rows;
  return rows;
}

static int32_t solve_one(const MkoModel *m, const MkoProblem *p, Work *w, int mrows, const double *q,
                         const double *frame_targets, const double *posture_targets, const double *com_targets,
                         const MkoDenseRows *dense, double dt, double damping, double *v_out, double *H_out, double *c_out) {
  const int nv = m->nv, nq = m->nq;
  kinematics(m, w, q);
  comPos(m, w);
  /* _compute_qp_objective (solve_ik.py:13-22) */
  memset(w->H, 0, sizeof(double) * nv * nv);
  memset(w->c, 0, sizeof(double) * nv);
  for (int i = 0; i < nv; ++i) w->H[i * nv + i] = damping;
  double *J6 = w->J, *Jt = w->J + 6 * nv;
  for (int t = 0; t < p->n_frame; ++t) {
    const MkoFrameTask *ft = p->frame + t;
    const double *target = frame_targets + 7 * t;
    double Tf[7], Tfi[7], Tti[7], Tbt[7], Ttb[7], e[6], JL[36];
    frame_transform_jacobian(m, w, ft->frame_type, ft->frame_id, Tf, J6);
    if (ft->root_type >= 0) {
      /* RelativeFrameTask (relative_frame_task.py:106-142): T_fr = T_root⁻¹·T_frame, e = T_fr.rminus(target) = log(target⁻¹·T_fr),
       * J = jlog(T_tf)·(ᶠJ − Ad(T_fr⁻¹)·ʳJ) with Ad(T) = [[R, [t]×R],[0, R]] (se3.py:187-194) */
      double *Jr = w->J + 12 * nv;
      double Tr[7], Tri[7], Tfr[7], Trf[7], Ttf[7], R[9], S[9], SR[9];
      frame_transform_jacobian(m, w, ft->root_type, ft->root_id, Tr, Jr);
      se3_inverse(Tri, Tr);
      se3_multiply(Tfr, Tri, Tf);
      se3_inverse(Tti, target);
      se3_multiply(Ttf, Tti, Tfr);
      se3_log(e, Ttf);
      se3_jlog(JL, Ttf);
      se3_inverse(Trf, Tfr);
      quat2Mat(R, Trf);
      skew(S, Trf + 4);
      mat3mul(SR, S, R);
      for (int k = 0; k < nv; ++k) {
        double d6[6];
        for (int i = 0; i < 3; ++i) {
          double a = 0, b = 0;
          for (int j = 0; j < 3; ++j) { a += R[3 * i + j] * Jr[j * nv + k] + SR[3 * i + j] * Jr[(3 + j) * nv + k]; b += R[3 * i + j] * Jr[(3 + j) * nv + k]; }
          d6[i] = J6[i * nv + k] - a; d6[3 + i] = J6[(3 + i) * nv + k] - b;
        }
        for (int r = 0; r < 6; ++r) {
          double sacc = 0;
          for (int i = 0; i < 6; ++i) sacc += JL[6 * r + i] * d6[i];
          Jt[r * nv + k] = sacc;
        }
      }
      add_objective(nv, 6, Jt, e, ft->cost, ft->gain, ft->lm_damping, w->H, w->c);
      continue;
    }
    se3_inverse(Tfi, Tf);
    se3_multiply(Tbt, Tfi, target);
    se3_log(e, Tbt);                                   /* target.minus(frame) (frame_task.py:119-122) */
    se3_inverse(Tti, target);
    se3_multiply(Ttb, Tti, Tf);
    se3_jlog(JL, Ttb);                                 /* J = −jlog(T_tb)·ᴮJ (frame_task.py:144-146) */
    for (int r = 0; r < 6; ++r)
      for (int k = 0; k < nv; ++k) {
        double s = 0;
        for (int i = 0; i < 6; ++i) s += -JL[6 * r + i] * J6[i * nv + k];
        Jt[r * nv + k] = s;
      }
    add_objective(nv, 6, Jt, e, ft->cost, ft->gain, ft->lm_damping, w->H, w->c);
  }
  for (int t = 0; t < p->n_posture; ++t) {
    const MkoPostureTask *pt = p->posture + t;
    double *qvel = w->e, *Jp = w->J;
    differentiatePos(m, qvel, 1.0, q, posture_targets + (size_t)t * nq);     /* posture_task.py:107 */
    memset(Jp, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; ++i) Jp[i * nv + i] = -1.0;
    for (int j = 0; j < m->njnt; ++j)
      if (m->jnt_type[j] == JNT_FREE) {
        const int va = m->jnt_dofadr[j];
        for (int k = 0; k < 6; ++k) { qvel[va + k] = 0.0; for (int i = 0; i < nv; ++i) Jp[i * nv + va + k] = 0.0; }
      }
    /* k = nv rows; diagonal J keeps this O(nv²) */
    double mu = 0.0;
    for (int r = 0; r < nv; ++r) { const double we = pt->cost[r] * (-pt->gain * qvel[r]); mu += we * we; }
    mu *= pt->lm_damping;
    for (int i = 0; i < nv; ++i) {
      const double jw = pt->cost[i] * Jp[i * nv + i], we = pt->cost[i] * (-pt->gain * qvel[i]);
      w->H[i * nv + i] += jw * jw + mu;
      w->c[i] += -(we * jw);
    }
  }
  for (int t = 0; t < p->n_com; ++t) {
    const MkoComTask *ct = p->com + t;
    double e[3];
    for (int k = 0; k < 3; ++k) e[k] = w->subtree_com[3 + k] - com_targets[3 * t + k];   /* com_task.py:81 */
    jacSubtreeCom(m, w, w->J, w->J + 3 * nv, 1);
    add_objective(nv, 3, w->J, e, ct->cost, ct->gain, ct->lm_damping, w->H, w->c);
  }
  if (dense)
    for (int t = 0, r0 = 0; t < p->n_dense; r0 += p->dense[t].k, ++t) {
      const MkoDenseTask *dt_ = p->dense + t;
      add_objective(nv, dt_->k, dense->task_J + (size_t)r0 * nv, dense->task_e + r0, dt_->cost, dt_->gain, dt_->lm_damping,
                    w->H, w->c);
    }
  if (H_out) memcpy(H_out, w->H, sizeof(double) * nv * nv);
  if (c_out) memcpy(c_out, w->c, sizeof(double) * nv);
  /* _compute_qp_inequalities (solve_ik.py:25-40) */
  int rows = 0;
  memset(w->G, 0, sizeof(double) * (size_t)mrows * nv);
  if (p->has_cfg_limit) {
    /* ConfigurationLimit (configuration_limit.py:41-124) */
    double *lower = w->tmpq, *upper = w->tmpq + nq, *dqmax = w->tmpv, *dqmin = w->tmpv + nv;
    for (int i = 0; i < nq; ++i) { lower[i] = -mjMAXVAL; upper[i] = mjMAXVAL; }
    int nidx = 0;
    int *idx = w->idx;
    for (int j = 0; j < m->njnt; ++j) {
      const int jt = m->jnt_type[j];
      if (jt == JNT_FREE || !m->jnt_limited[j]) continue;
      const int padr = m->jnt_qposadr[j], qw = (jt == JNT_BALL) ? 4 : 1, dw = (jt == JNT_BALL) ? 3 : 1;
      for (int k = 0; k < qw; ++k) {
        lower[padr + k] = m->jnt_range[2 * j] + p->cfg_min_distance;
        upper[padr + k] = m->jnt_range[2 * j + 1] - p->cfg_min_distance;
      }
      for (int k = 0; k < dw; ++k) idx[nidx++] = m->jnt_dofadr[j] + k;
    }
    if (nidx) {
      memset(dqmax, 0, sizeof(double) * nv); memset(dqmin, 0, sizeof(double) * nv);
      differentiatePos(m, dqmax, 1.0, q, upper);
      differentiatePos(m, dqmin, 1.0, lower, q);
      for (int k = 0; k < nidx; ++k) { w->G[(size_t)(rows + k) * nv + idx[k]] = 1.0; w->h[rows + k] = p->cfg_gain * dqmax[idx[k]]; }
      rows += nidx;
      for (int k = 0; k < nidx; ++k) { w->G[(size_t)(rows + k) * nv + idx[k]] = -1.0; w->h[rows + k] = p->cfg_gain * dqmin[idx[k]]; }
      rows += nidx;
    }
  }
  if (p->n_vel) {  /* VelocityLimit (velocity_limit.py:71-101) */
    for (int k = 0; k < p->n_vel; ++k) { w->G[(size_t)(rows + k) * nv + p->vel_idx[k]] = 1.0; w->h[rows + k] = dt * p->vel_limit[k]; }
    rows += p->n_vel;
    for (int k = 0; k < p->n_vel; ++k) { w->G[(size_t)(rows + k) * nv + p->vel_idx[k]] = -1.0; w->h[rows + k] = dt * p->vel_limit[k]; }
    rows += p->n_vel;
  }
  /* CollisionAvoidanceLimit rows; an inactive pair's row (h = +inf, G = 0: collision_avoidance_limit.py:191-197) can never
   * bind and is left out of the QP — same optimum */
  for (int l = 0; l < p->n_coll; ++l) {
    const MkoCollisionLimit *cl = p->coll + l;
    double *Gc = w->G + (size_t)rows * nv, *hc = w->h + rows;
    if (collision_rows(m, w, cl, dt, Gc, hc)) { for (int i = 0; i < nv; ++i) v_out[i] = NAN; return -2; }
    int keep = 0;
    for (int k = 0; k < cl->n_pairs; ++k) {
      if (!isfinite(hc[k])) continue;
      if (keep != k) { memcpy(Gc + (size_t)keep * nv, Gc + (size_t)k * nv, sizeof(double) * nv); hc[keep] = hc[k]; }
      ++keep;
    }
    rows += keep;
  }
  if (dense)
    for (int k = 0; k < p->n_dense_limit_rows; ++k) {
      if (!isfinite(dense->limit_h[k])) continue;                       /* an inactive row of a caller's limit */
      memcpy(w->G + (size_t)rows * nv, dense->limit_G + (size_t)k * nv, sizeof(double) * nv);
      w->h[rows++] = dense->limit_h[k];
    }
  double *dq = w->dq;
  const int32_t rc = solve_qp_ws(nv, rows, w->H, w->c, w->G, w->h, dq, w->qp);
  if (rc) { for (int i = 0; i < nv; ++i) v_out[i] = NAN; return rc; }
  for (int i = 0; i < nv; ++i) v_out[i] = dq[i] / dt;     /* solve_ik.py:104 */
  return 0;
}

int32_t mko_solve_ik(const MkoModel *m, const MkoProblem *p, const double *q, const double *frame_targets,
                     const double *posture_targets, const double *com_targets, double dt, double damping,
                     double *v_out, double *H_out, double *c_out) {
  const int mrows = count_rows(m, p);
  Work *w = work_new(m, mrows > 0 ? mrows : 1);
  const int32_t rc = solve_one(m, p, w, mrows > 0 ? mrows : 1, q, frame_targets, posture_targets, com_targets, NULL, dt,
                               damping, v_out, H_out, c_out);
  work_free(w);
  return rc;
}

int32_t mko_solve_ik_batch(const MkoModel *m, const MkoProblem *p, int32_t B, const double *q,
                           const double *frame_targets, const double *posture_targets, int32_t posture_batched,
                           const double *com_targets, int32_t com_batched, double dt, double damping, int32_t nthreads,
                           double *v_out, int32_t *status_out) {
  return mko_solve_ik_batch_dense(m, p, B, q, frame_targets, posture_targets, posture_batched, com_targets, com_batched,
                                  NULL, dt, damping, nthreads, v_out, status_out);
}

int32_t mko_solve_ik_batch_dense(const MkoModel *m, const MkoProblem *p, int32_t B, const double *q,
                                 const double *frame_targets, const double *posture_targets, int32_t posture_batched,
                                 const double *com_targets, int32_t com_batched, const MkoDenseRows *rows, double dt,
                                 double damping, int32_t nthreads, double *v_out, int32_t *status_out) {
  int K = 0;
  for (int t = 0; t < p->n_dense; ++t) K += p->dense[t].k;
  const int M = p->n_dense_limit_rows;
  if ((K || M) && !rows) return -3;
  const int mrows = count_rows(m, p) > 0 ? count_rows(m, p) : 1;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    Work *w = work_new(m, mrows);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int b = 0; b < B; ++b) {
      const double *pt = posture_targets ? posture_targets + (posture_batched ? (size_t)b * p->n_posture * m->nq : 0) : NULL;
      const double *ct = com_targets ? com_targets + (com_batched ? (size_t)b * p->n_com * 3 : 0) : NULL;
      MkoDenseRows rb;
      if (rows) {
        rb.task_e = K ? rows->task_e + (size_t)b * K : NULL;
        rb.task_J = K ? rows->task_J + (size_t)b * K * m->nv : NULL;
        rb.limit_G = M ? rows->limit_G + (size_t)b * M * m->nv : NULL;
        rb.limit_h = M ? rows->limit_h + (size_t)b * M : NULL;
      }
      const int32_t rc = solve_one(m, p, w, mrows, q + (size_t)b * m->nq, frame_targets + (size_t)b * p->n_frame * 7, pt,
                                   ct, rows ? &rb : NULL, dt, damping, v_out + (size_t)b * m->nv, NULL, NULL);
      if (status_out) status_out[b] = rc;
    }
    work_free(w);
  }
  return 0;
}
