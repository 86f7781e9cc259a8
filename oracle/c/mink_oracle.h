/*
 * ORACLE (test infrastructure, never shipped, never imported by mink_amd/) — plain-C restatement of the
 * reference's CPU path for one solve_ik call:
 *
 *   mink.solve_ik                      mink/solve_ik.py:68-105
 *   build_ik / objective / inequalities mink/solve_ik.py:13-65
 *   Configuration.update / frames      mink/configuration.py:53-64,112-185
 *   FrameTask / PostureTask / ComTask   mink/tasks/{frame_task,posture_task,com_task,task}.py
 *   ConfigurationLimit / VelocityLimit  mink/limits/{configuration_limit,velocity_limit}.py
 *   SO3 / SE3                           mink/lie/{so3,se3,base}.py
 *   MuJoCo arithmetic (mj_kinematics, mj_comPos, mj_jac*, mj_differentiatePos): third-party, restated from
 *   the published algorithm — PARITY UNPINNED against the mujoco wheel itself
 *   quadprog (Goldfarb–Idnani, dense):  third-party, restated — PARITY UNPINNED
 *
 * Same operation order as the numpy restatement (the .py files next to this directory), which is pinned against the real mink Python
 * (tests/golden/make_golden.py); tests/test_oracle_c.py pins this file against both.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 */
#ifndef MINK_ORACLE_H
#define MINK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t nq, nv, nbody, njnt, ngeom, nsite;
  const int32_t *body_parentid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_mocapid;
  const double *body_pos, *body_quat, *body_ipos, *body_mass, *body_subtreemass;
  const int32_t *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range;
  const int32_t *dof_parentid;
  const double *qpos0;
  const int32_t *site_bodyid;
  const double *site_pos, *site_quat;
  const int32_t *geom_bodyid;
  const double *geom_pos, *geom_quat;
  const double *mocap_pos, *mocap_quat;
  const int32_t *geom_type;     /* mjtGeom: 0 plane, 2 sphere, 3 capsule, 5 cylinder, 6 box (the pair types restated here) */
  const double *geom_size;      /* (ngeom, 3) */
} MkoModel;

enum { MKO_FRAME_BODY = 0, MKO_FRAME_GEOM = 1, MKO_FRAME_SITE = 2 };

/* root_type < 0: FrameTask (mink/tasks/frame_task.py); else RelativeFrameTask(frame, root) — mink/tasks/relative_frame_task.py:28-142,
 * the task's target slot holds transform_target_to_root */
typedef struct { int32_t frame_type, frame_id; double cost[6], gain, lm_damping; int32_t root_type, root_id; } MkoFrameTask;
typedef struct { const double *cost; double gain, lm_damping; } MkoPostureTask;       /* cost: (nv,) */
typedef struct { double cost[3], gain, lm_damping; } MkoComTask;

/* CollisionAvoidanceLimit(model, geom_pairs, gain, minimum_distance_from_collisions, collision_detection_distance,
 * bound_relaxation) — mink/limits/collision_avoidance_limit.py:78-114; pairs: (n_pairs, 2) geom ids */
typedef struct {
  int32_t n_pairs; const int32_t *pairs;
  double gain, minimum_distance, detection_distance, bound_relaxation;
} MkoCollisionLimit;

/* a caller-defined mink.Task reduced to what Task.compute_qp_objective consumes (mink/tasks/task.py:81-138): k rows with
 * cost (k,), gain, lm_damping; its e (k,) and J (k, nv) arrive per instance (MkoDenseRows) */
typedef struct { int32_t k; const double *cost; double gain, lm_damping; } MkoDenseTask;

/* per-instance values of the caller-defined tasks / limits of a batch: task_e (B, K), task_J (B, K, nv) with K = Σ k in
 * task order; limit_G (B, M, nv), limit_h (B, M) — the (G, h) of mink.Limit.compute_qp_inequalities
 * (mink/limits/limit.py:34-57), rows with h = +inf inactive */
typedef struct { const double *task_e, *task_J, *limit_G, *limit_h; } MkoDenseRows;

typedef struct {
  int32_t n_frame; const MkoFrameTask *frame;
  int32_t n_posture; const MkoPostureTask *posture;
  int32_t n_com; const MkoComTask *com;
  int32_t has_cfg_limit; double cfg_gain, cfg_min_distance;    /* ConfigurationLimit(model, gain, min_distance) */
  int32_t n_vel; const int32_t *vel_idx; const double *vel_limit;   /* VelocityLimit: dof indices, max |v| */
  int32_t n_coll; const MkoCollisionLimit *coll;               /* CollisionAvoidanceLimits (plane / sphere / capsule pairs) */
  int32_t n_dense; const MkoDenseTask *dense;                  /* caller-defined tasks (added after the built-in ones) */
  int32_t n_dense_limit_rows;                                  /* M: rows of the caller-defined limits (stacked last) */
} MkoProblem;

/* status: 0 ok, 2 constraints inconsistent, 4 H not positive definite, 8 iteration limit, -2 a geom pair of a
 * CollisionAvoidanceLimit is outside the restated set (plane / sphere / capsule) */
int32_t mko_solve_ik(const MkoModel *m, const MkoProblem *p, const double *q, const double *frame_targets,
                     const double *posture_targets, const double *com_targets, double dt, double damping,
                     double *v_out, double *H_out, double *c_out);

/* B independent problems; posture_batched: posture_targets is (B, n_posture, nq) instead of (n_posture, nq);
 * com_batched: com_targets is (B, n_com, 3) instead of (n_com, 3).  nthreads > 1 uses OpenMP. */
int32_t mko_solve_ik_batch(const MkoModel *m, const MkoProblem *p, int32_t B, const double *q,
                           const double *frame_targets, const double *posture_targets, int32_t posture_batched,
                           const double *com_targets, int32_t com_batched, double dt, double damping, int32_t nthreads,
                           double *v_out, int32_t *status_out);

/* the same with caller-defined task / limit rows (rows may be NULL when the problem has none) */
int32_t mko_solve_ik_batch_dense(const MkoModel *m, const MkoProblem *p, int32_t B, const double *q,
                                 const double *frame_targets, const double *posture_targets, int32_t posture_batched,
                                 const double *com_targets, int32_t com_batched, const MkoDenseRows *rows, double dt,
                                 double damping, int32_t nthreads, double *v_out, int32_t *status_out);

/* rows of one CollisionAvoidanceLimit at configuration q: G_out (n_pairs, nv), h_out (n_pairs; +inf = inactive pair).
 * Returns 0, or -2 for an unsupported pair type. */
int32_t mko_collision_rows(const MkoModel *m, const MkoCollisionLimit *c, const double *q, double dt, double *G_out,
                           double *h_out);

/* dense strictly convex QP  min ½xᵀPx + qᵀx  s.t. Gx ≤ h  (Goldfarb–Idnani); P is n×n, G is m×n row-major */
int32_t mko_solve_qp(int32_t n, int32_t m, const double *P, const double *q, const double *G, const double *h,
                     double *x_out);

#ifdef __cplusplus
}
#endif
#endif
