/*
 * minkhip.h — C ABI of libminkhip.so: batched differential IK on MI355X (gfx950).
 *
 * Drop-in boundary for the ONE hot path of kevinzakka/mink: `solve_ik` over a batch
 * of independent (q, target) instances.  The reference has no FFI of its own on
 * this path — it is a pure-Python plugin API that crosses into native code through
 * the third-party wheels `mujoco` (pybind11) and `quadprog` (Cython).  Every entry
 * point below therefore cites the reference *Python* interface it replaces
 * (file:line in /root/reference) and is what a ctypes binding of that interface
 * binds (see INTEGRATION.md for the binding a mink maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch/numpy types.
 *   - every function returns MKH_OK (0) or a negative MKH_E_* code and never
 *     throws; mkh_last_error() gives a thread-local message.
 *   - arrays are float64 / int32, row-major, batch-major: q is (B, nq), v is (B, nv).
 *   - data pointers of the per-call functions are DEVICE pointers when
 *     MKH_FLAG_DEVICE_PTRS is set (e.g. torch.Tensor.data_ptr() on ROCm; the call
 *     is then asynchronous on `stream`), otherwise HOST pointers (the library
 *     stages through its own device buffers and returns after completion; from 32 MB
 *     of staged data on in up to four chunks whose copies run beside the kernels of their
 *     neighbours, on streams of the handle ordered by events against `stream`).
 *   - the library owns everything it allocates; the caller owns every buffer it
 *     passes.  One in-flight call per MkhProblem: asynchronous calls on one handle must be
 *     ordered (same stream, or events) — a handle carries the staging buffers and the ticket
 *     counter that hands the tail of a batch to idle wavefronts, so two of its launches running
 *     at the same time would corrupt each other.  Distinct problems are independent.  An MkhModel is shared by the
 *     problems created on it and by every caller of mkh_integrate: that entry point is re-entrant (its small host-pointer
 *     path stages through one buffer of the model under a mutex; every other path holds no state of the model).  A device-pointer
 *     call is one plain kernel launch with no host-side state: it can be captured into a hipGraph and
 *     replayed (tests/test_gpu_scale.py::test_launches_replay_inside_a_hip_graph).
 *   - per-instance `status` (int32): bit flags MKH_ST_*.
 */
#ifndef MINKHIP_H_
#define MINKHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 106 (round 4): + mkh_geom_distance_eval; models beyond 64 bodies / dofs and instances with up to 448 active half-space rows
 * (the workgroup-per-problem kernel) where 105 returned MKH_E_LIMIT / MKH_ST_ROW_OVERFLOW.  No struct changed.
 * 107 (round 5): no signature or struct changed; calls that 106 refused now run — on models beyond 64 bodies / dofs the per-task
 * (e, J) and iteration taps (mkh_eval) and the fused loops mkh_solve_steps / mkh_solve_until; the fused loops and calls with taps
 * no longer report MKH_ST_ROW_OVERFLOW below 448 rows per instance (the flagged instances run again with every row).
 * 108 (round 6): + mkh_problem_create_diag and the MKH_DIAG_* bits (per-handle parity / measurement switches that used to be
 * MKH_DEBUG_* environment variables: the product library no longer reads its environment); MKH_ST_DEGENERATE documented.  No
 * struct and no existing signature changed. */
#define MKH_VERSION 108

/* return codes */
#define MKH_OK 0
#define MKH_E_INVALID (-1)   /* bad argument / unsupported model feature   */
#define MKH_E_HIP (-2)       /* HIP runtime error (message has the detail)  */
#define MKH_E_NOGPU (-3)     /* no gfx950 device visible                    */
#define MKH_E_LIMIT (-4)     /* exceeds a size limit (16 frame tasks, ...).  Models beyond 64 bodies or 64 dofs run on the
                                workgroup-per-problem kernel, which keeps a problem's kinematic state in the 160 KB of LDS of one
                                CU: ≈ 8·(nq + 7·nbody + 6·njnt + 14·nv + 5.5·(nv + rows) + 64·frame tasks) bytes ≤ 158 KB — a
                                serial chain fits up to ≈ 550 dofs; mkh_problem_create reports the figure.  (mkh_model_create
                                itself refuses only beyond 4096 bodies / 1024 dofs.) */

/* per-instance status bits written to status_out */
#define MKH_ST_OK 0
#define MKH_ST_OUTSIDE_LIMITS 1  /* q violates a joint range by > 1e-6 (mink/configuration.py:77-110); still solved */
#define MKH_ST_INFEASIBLE 2      /* constraints inconsistent (quadprog "no solution" → mink/solve_ik.py:103 assert) */
#define MKH_ST_NOT_PD 4          /* H not positive definite (quadprog "matrix G is not positive definite") */
#define MKH_ST_ITER_LIMIT 8      /* active-set iteration cap hit */
#define MKH_ST_ROW_OVERFLOW 16   /* more half-space rows active at once than the solve could hold AND a row that found no place is
                                    violated at the solution.  No entry point returns it below 448 rows per instance: a wavefront
                                    kernel holds 64 - nv rows (the tightest contacts get them, the rest are checked at the
                                    solution), and the instances it flags are solved again with EVERY row by the
                                    workgroup-per-problem kernel — plain solves, calls with taps and (round 5) the fused loops
                                    mkh_solve_steps / mkh_solve_until, whose flagged instances run their whole loop again.  Beyond
                                    448 contacts in range the same rule applies one level up: the 448 tightest are rows, the bit
                                    is set only if a dropped one is violated at the solution; caller-defined limit rows that find
                                    no place set it unconditionally. */
#define MKH_ST_DEGENERATE 32     /* the active half-space rows of this instance were almost linearly dependent (many geom pairs of one
                                    body pair) and the answer comes from the tableau iteration, which loses digits there (errors up
                                    to ~1e-6 relative).  Transient: the workgroup-per-problem kernel behind every problem with rows
                                    re-solves such instances with orthogonal factors (quadprog's own algorithm) and clears the bit.
                                    A caller sees it only where that launch does not run: handles created with
                                    MKH_DIAG_NO_WIDE_REDO and calls that tap the cycle counters. */

/* flags */
#define MKH_FLAG_DEVICE_PTRS 1   /* data pointers are device pointers; async on stream */
#define MKH_FLAG_POSTURE_BATCHED 2  /* posture_target is (B, nq) instead of (nq,)       */
#define MKH_FLAG_COM_BATCHED 4      /* com_target is (B, 3) instead of (3,)             */
#define MKH_FLAG_DIRECT_QP 8        /* never use the low-rank start of the QP (parity/diagnostic switch) */
#define MKH_FLAG_WAVE_KERNEL 16     /* never use the row- / lane-per-problem kernels of small robots (parity/diagnostic switch) */
#define MKH_FLAG_LANE_KERNEL 32     /* use the lane-per-problem kernel whenever the problem qualifies, whatever the batch
                                     * size (default: plain solves from 73728 instances, fused loops from 28672;
                                     * parity/diagnostic switch) */
#define MKH_FLAG_TWO_WAVES 64       /* never use the 3-waves-per-SIMD kernel variants (parity/diagnostic switch) */
#define MKH_FLAG_WARM_START 128     /* closed-loop callers: start the QP's active-set phase from where the previous solve of
                                     * THIS problem handle (same batch size, instance i = instance i) ended.  The state lives
                                     * in the handle; it is used from its third solve on and reset when the batch size
                                     * changes.  Same optimum as a cold solve (the QP is strictly convex), fewer pivots:
                                     * along an IK loop the active set changes by a few dofs per step.  Every kernel
                                     * family honours it: the wavefront kernels, the row kernel and (round 6) the lane kernel of
                                     * small robots; all keep their partition inside mkh_solve_steps / _until. */
#define MKH_FLAG_FULL_ROWS 512      /* collision problems: never launch the tight-rows variant first (fewer half-space rows than
                                     * geom pairs, the tightest contacts get them, flagged instances re-solved on the full-row
                                     * variant behind it) — parity/diagnostic switch */
#define MKH_FLAG_QUAD_KERNEL 256    /* use the row-per-problem kernel of small robots (16 lanes per problem, nv <= 16) whenever the problem
                                     * qualifies, whatever the batch size (default: plain solves below 73728 instances, fused
                                     * loops below 28672; parity/diagnostic switch) */

/* frame types (mink/constants.py:3 SUPPORTED_FRAMES) */
#define MKH_FRAME_BODY 0
#define MKH_FRAME_GEOM 1
#define MKH_FRAME_SITE 2

typedef struct MkhModel MkhModel;
typedef struct MkhProblem MkhProblem;

/*
 * One-time flattened copy of the mjModel kinematic tree: the mjModel fields the
 * reference hot path reads (mink/configuration.py:53-155, limits/ constructors,
 * tasks/posture_task.py:44).  Field names/semantics are MuJoCo's.  Host pointers;
 * copied at mkh_model_create.
 */
typedef struct MkhFlatModel {
  int32_t nq, nv, nbody, njnt, ngeom, nsite;
  const int32_t *body_parentid, *body_rootid, *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr;
  const double *body_pos /*nbody*3*/, *body_quat /*nbody*4*/, *body_ipos /*nbody*3*/;
  const double *body_mass, *body_subtreemass;
  const int32_t *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos /*njnt*3*/, *jnt_axis /*njnt*3*/, *jnt_range /*njnt*2*/, *qpos0 /*nq*/;
  const int32_t *dof_bodyid, *dof_jntid, *dof_parentid;
  const int32_t *site_bodyid;
  const double *site_pos /*nsite*3*/, *site_quat /*nsite*4*/;
  const int32_t *geom_bodyid, *geom_type;
  const double *geom_size /*ngeom*3*/, *geom_pos /*ngeom*3*/, *geom_quat /*ngeom*4*/;
  /* Mesh geoms (mjGEOM_MESH = 7) as mj_geomDistance sees them: their CONVEX HULL.  geom_dataid[g] = mesh of geom g (−1: not
   * a mesh geom); the hull vertices of mesh k, in the geom frame (the compiler re-expresses a mesh in its inertial frame and
   * folds that frame into geom_pos / geom_quat), are mesh_vert[3·mesh_vertadr[k] …), mesh_vertnum[k] of them — MuJoCo's
   * field names; from a real MjModel: the vertices mesh_graph lists (FlatModel.from_mjmodel).  nmesh = 0 / NULL pointers:
   * a model without mesh geoms.  A primitive FITTED to a mesh (<geom type="capsule" mesh=…>) is an ordinary primitive
   * here: its compiled geom_size / geom_pos / geom_quat say everything. */
  int32_t nmesh, nmeshvert;
  const int32_t *geom_dataid /*ngeom*/, *mesh_vertadr /*nmesh*/, *mesh_vertnum /*nmesh*/;
  const double *mesh_vert /*nmeshvert*3*/;
} MkhFlatModel;

/* mink.FrameTask(frame_name, frame_type, position_cost, orientation_cost, gain, lm_damping)
 * — mink/tasks/frame_task.py:29-46; cost = [position x3, orientation x3] (:60,:75). */
typedef struct MkhFrameTaskDesc {
  int32_t frame_type, frame_id;
  double cost[6];
  double gain, lm_damping;
  /* mink.RelativeFrameTask(frame, root, ...) — mink/tasks/relative_frame_task.py:28-48: pose of the
   * frame expressed in `root`; root_type < 0 selects the plain FrameTask (pose in the world).  The
   * task's target slot then holds transform_target_to_root. */
  int32_t root_type, root_id;
} MkhFrameTaskDesc;

/* mink.PostureTask(model, cost, gain, lm_damping) — mink/tasks/posture_task.py:29-52.
 * mink.DampingTask is the gain=0 special case (mink/tasks/damping_task.py:11-20). */
typedef struct MkhPostureTaskDesc {
  const double *cost /*nv*/;
  double gain, lm_damping;
} MkhPostureTaskDesc;

/* mink.ComTask(cost, gain, lm_damping) — mink/tasks/com_task.py:25-35 (subtree of body 1). */
typedef struct MkhComTaskDesc {
  double cost[3];
  double gain, lm_damping;
} MkhComTaskDesc;

/* mink.ConfigurationLimit(model, gain, min_distance_from_limits) —
 * mink/limits/configuration_limit.py:18-67.  lower/upper are the constructor's
 * per-qpos arrays (±mjMAXVAL where unlimited); indices are its dof `indices`. */
typedef struct MkhConfigurationLimitDesc {
  double gain;
  const double *lower /*nq*/, *upper /*nq*/;
  int32_t n_indices;
  const int32_t *indices;
} MkhConfigurationLimitDesc;

/* mink.VelocityLimit(model, velocities) — mink/limits/velocity_limit.py:33-69:
 * `indices` (dof ids) and `limit` (max |v|). */
typedef struct MkhVelocityLimitDesc {
  int32_t n_indices;
  const int32_t *indices;
  const double *limit;
} MkhVelocityLimitDesc;

/* mink.CollisionAvoidanceLimit(model, geom_pairs, gain, minimum_distance_from_collisions,
 * collision_detection_distance, bound_relaxation) — mink/limits/collision_avoidance_limit.py:145-185;
 * geom_id_pairs is the constructor's filtered (min,max) id list (:253-278).
 * Distance routines behind mj_geomDistance (:219): analytic for plane/sphere/capsule among themselves, box against
 * plane/sphere/capsule/box, cylinder against plane/sphere/capsule, plane–ellipsoid; every other pair of the convex
 * primitives sphere / capsule / ellipsoid / cylinder / box (cylinder–box, cylinder–cylinder, ellipsoid–*) through a
 * general convex distance routine (GJK on support mappings — MuJoCo uses libccd there).  A MESH geom takes part through its
 * convex hull (MkhFlatModel.mesh_vert): plane–mesh analytically (the hull's lowest vertex), mesh against any primitive or
 * mesh through the general convex routine with the hull's vertices as the support mapping.  Height fields — and mesh geoms
 * of a model that carries no hull for them — fail mkh_problem_create with MKH_E_INVALID.
 *
 * Two places where the contact differs from mujoco 3.1.6's by construction (no test against the wheel can exist here):
 *  (i) TIES.  Where the closest pair of points is not unique — a capsule parallel to a box face, face-to-face boxes —
 *      MuJoCo returns several contacts of equal distance and mj_geomDistance keeps the first of ITS enumeration.  The routines
 *      here: capsule ∥ box face — the midpoint of the stretch of the capsule's axis that lies over the face; parallel capsules —
 *      the two ends of the overlapping stretch as two contacts, the first kept (mjc_CapsuleCapsule's shape); boxes face to
 *      face — the first closest (vertex, face) pair in the routine's own vertex order.  The distance h is the same as MuJoCo's;
 *      the witness points feed mj_jac, so the row of G may differ by the lever arm between two equally close points.
 *      tests/test_gpu_collision_shapes.py::test_tie_rule_is_pinned holds these three rules in place.
 *  (ii) GENERAL CONVEX PAIRS.  The nine primitive pair types without an analytic routine, and every mesh pair, get the exact
 *      Euclidean distance of the two convex sets (GJK, ~1e-13) where MuJoCo answers with libccd's MPR on shapes inflated by
 *      half the margin each, to a tolerance of 1e-6: h agrees to that tolerance, not to 1e-9. */
typedef struct MkhCollisionLimitDesc {
  int32_t n_pairs;
  const int32_t *geom_id_pairs /*n_pairs*2*/;
  double gain, minimum_distance_from_collisions, collision_detection_distance, bound_relaxation;
} MkhCollisionLimitDesc;

/*
 * Plugin route: a caller-defined mink.Task subclass — anything that implements the reference's extension point
 * Task.compute_error / Task.compute_jacobian (mink/tasks/task.py:81-103) — reaches the device as DENSE ROWS: the
 * descriptor holds its constructor state (cost per row, gain, lm_damping: task.py:48-62), the per-call arrays hold
 * what its two methods return for every instance (MkhDenseRows).  The kernel folds them into the objective exactly
 * like Task.compute_qp_objective does (task.py:105-138): weighted rows W·J, weighted error W·(−gain·e),
 * H += (WJ)ᵀ(WJ) + lm_damping·‖W·(−gain·e)‖²·I, c −= (W·(−gain·e))ᵀ·WJ.
 */
typedef struct MkhDenseTaskDesc {
  int32_t k;            /* rows of compute_error / compute_jacobian */
  const double *cost;   /* (k,) */
  double gain, lm_damping;
} MkhDenseTaskDesc;

/* The task/limit lists solve_ik receives (mink/solve_ik.py:68-77).  The QP objective is a
 * sum, so list order only affects rounding; limits=[] disables limits, the Python layer
 * materialises mink's limits=None default (a fresh ConfigurationLimit, solve_ik.py:28-29). */
typedef struct MkhProblemDesc {
  int32_t n_frame_tasks;
  const MkhFrameTaskDesc *frame_tasks;
  int32_t n_posture_tasks;
  const MkhPostureTaskDesc *posture_tasks; /* each has its own target slot */
  int32_t n_com_tasks;
  const MkhComTaskDesc *com_tasks;
  int32_t n_configuration_limits;
  const MkhConfigurationLimitDesc *configuration_limits;
  int32_t n_velocity_limits;
  const MkhVelocityLimitDesc *velocity_limits;
  int32_t n_collision_limits;
  const MkhCollisionLimitDesc *collision_limits;
  /* plugin route (mkh_solve_dense): user Task subclasses, and the total number of rows G·Δq ≤ h that user Limit
   * subclasses return from Limit.compute_qp_inequalities (mink/limits/limit.py:34-57) */
  int32_t n_dense_tasks;
  const MkhDenseTaskDesc *dense_tasks;
  int32_t n_dense_limit_rows;
  /* 1: user Limit subclasses also contribute per-instance BOX rows (rows of G with a single nonzero entry — e.g. an
   * acceleration limit [I; −I] — folded by the caller into lo ≤ Δq ≤ hi): MkhDenseRows.limit_lo / limit_hi.  Box rows cost no
   * tableau row, so a limit with 2·nv of them fits any robot (general rows are capped at 64 − nv per instance). */
  int32_t dense_limit_box;
} MkhProblemDesc;

/* Per-call arrays of the plugin route (same host/device pointer convention as q).  K = Σ k over the dense tasks (in
 * descriptor order), M = n_dense_limit_rows.  A limit row with h = +inf is inactive (mink's Constraint.inactive,
 * limit.py:19-23, per row); at most 64 − nv rows (collision + dense) can be active in one instance. */
typedef struct MkhDenseRows {
  const double *task_e; /* (B, K)      compute_error per instance                    */
  const double *task_J; /* (B, K, nv)  compute_jacobian per instance                 */
  const double *limit_G;/* (B, M, nv)  compute_qp_inequalities(...).G per instance   */
  const double *limit_h;/* (B, M)      compute_qp_inequalities(...).h per instance   */
  const double *limit_lo;/* (B, nv) or NULL: per-instance lower bounds on Δq from single-entry rows (−inf = none);   */
  const double *limit_hi;/* (B, nv) or NULL: upper bounds; both need MkhProblemDesc.dense_limit_box = 1             */
} MkhDenseRows;

/* Optional debug/parity taps: any non-NULL pointer receives that intermediate for the
 * whole batch (same host/device convention as the other data pointers). */
typedef struct MkhTaps {
  double *xpos;      /* (B, nbody, 3)  data.xpos          (mink/configuration.py:63)  */
  double *xquat;     /* (B, nbody, 4)  data.xquat                                     */
  double *frame_pose;/* (B, n_frame_tasks, 7) wxyz_xyz, get_transform_frame_to_world (:157-185) */
  double *subtree_com;/*(B, 3)        data.subtree_com[1] (mink/tasks/com_task.py:82) */
  double *task_e;    /* (B, n_rows)   compute_error of frame tasks (6 each), posture (nv each), com (3 each) */
  double *task_J;    /* (B, n_rows, nv) compute_jacobian, same row order             */
  double *H;         /* (B, nv, nv)   build_ik(...).P   (mink/solve_ik.py:63)         */
  double *c;         /* (B, nv)       build_ik(...).q                                 */
  double *box_lo;    /* (B, nv)  merged box  lo <= dq <= hi  from Configuration/VelocityLimit rows */
  double *box_hi;    /* (B, nv)                                                        */
  double *coll_G;    /* (B, n_pairs, nv) CollisionAvoidanceLimit G rows (0 when inactive) */
  double *coll_h;    /* (B, n_pairs)     and h (+inf when inactive)                   */
  int32_t *qp_iters; /* (B,)  packed counters of the active-set phase after the unconstrained solve:
                      * bits 0-9 pivots (Goldfarb-Idnani selections when rows exist), 10-19 outer loop
                      * iterations / block steps, 20-29 rank-1 pivots */
  int64_t *cycles;   /* (B, 16): [0,8) shader-clock stamps at the kernel's phase boundaries, [8,16) cycles summed
                      * over the QP iterations: phase-0 publish, phase-0 pivot, GI select, GI publish, GI ratio
                      * test, GI pivot; [14] stamp at the end of the Jacobian-column loop (profiling) */
} MkhTaps;

int32_t mkh_version(void);
const char *mkh_last_error(void);
int32_t mkh_device_count(void);

/* Upload the flattened kinematic tree once: the mujoco.MjModel fields mink reads per solve through
 * Configuration.update / get_frame_jacobian (mink/configuration.py:53-64,112-155), the limits' constructors
 * (limits/configuration_limit.py:41-67, limits/velocity_limit.py:45-69) and the collision pair filter
 * (limits/collision_avoidance_limit.py:75-115). */
int32_t mkh_model_create(const MkhFlatModel *host_model, int32_t device, MkhModel **out);
void mkh_model_destroy(MkhModel *model);

/* Snapshot the Task/Limit plugin objects of one solve_ik call site (the `tasks` and `limits` arguments of
 * mink/solve_ik.py:68-77; constructor state of tasks/frame_task.py:29-46, relative_frame_task.py:28-52,
 * posture_task.py:29-52, com_task.py:25-35 and of the three limits) into a device descriptor. */
/* max_batch: the largest B any later call on this handle may pass — with host OR device pointers: it sizes everything the
 * handle owns per instance (staging buffers, the active sets kept for MKH_FLAG_WARM_START).  A call with B > max_batch
 * returns MKH_E_INVALID before anything is launched.
 * A handle owns device state its launches share (ticket counters, the workspace slices and the redo queue of the
 * workgroup-per-problem kernel, warm-start sets): calls on ONE handle must not overlap in time — one stream at a time, or
 * streams ordered by events.  Handles of the same model are independent of each other. */
int32_t mkh_problem_create(MkhModel *model, const MkhProblemDesc *desc, int32_t max_batch, MkhProblem **out);
/* The same with per-handle diagnostic switches (MKH_DIAG_* bits, 0 = mkh_problem_create).  They select among paths that return
 * the same optimum and exist for parity tests ("what does the first launch alone leave flagged?") and measurements (bench.py's
 * redo_instances); no counterpart in the reference (mink/solve_ik.py:68-105 has one path).  Unknown bits: MKH_E_INVALID. */
#define MKH_DIAG_NO_WIDE_REDO 1    /* problems with half-space rows: do not build / launch the workgroup-per-problem kernel behind the
                                      wavefront kernel — instances it would re-solve keep MKH_ST_ROW_OVERFLOW / _DEGENERATE / failures */
#define MKH_DIAG_NO_TIGHT_REDO 2   /* collision problems on the tight-rows build: do not launch the full-row build behind it */
#define MKH_DIAG_NO_COLD_REFINE 4  /* low-rank QP start: no second elimination for the bounds the unconstrained minimiser violates */
#define MKH_DIAG_NO_PAIR_CULL 8    /* more than 64 collision pairs: no bounding-sphere cull in front of the distance routines */
int32_t mkh_problem_create_diag(MkhModel *model, const MkhProblemDesc *desc, int32_t max_batch, int32_t diag,
                                MkhProblem **out);
void mkh_problem_destroy(MkhProblem *problem);
int32_t mkh_problem_num_task_rows(const MkhProblem *problem);
int32_t mkh_problem_num_collision_pairs(const MkhProblem *problem);
/* Name of the kernel variant the last solve/eval on this handle launched ("" before the first call):
 * "ik_solve_kernel_<rows>_<features>[_r<dof rows>]".  Diagnostic (benchmarks, profiles). */
const char *mkh_problem_last_kernel(const MkhProblem *problem);

/*
 * Batched mink.solve_ik (mink/solve_ik.py:68-105):
 *   v[b] = solve_ik(Configuration(model, q[b]), tasks(targets[b]), dt, "quadprog", damping, limits=limits)
 *   q              (B, nq)
 *   frame_targets  (B, n_frame_tasks, 7)  transform_target_to_world as wxyz_xyz (frame_task.py:77-83)
 *   posture_target (n_posture_tasks, nq) or (B, n_posture_tasks, nq) with MKH_FLAG_POSTURE_BATCHED
 *   com_target     (n_com_tasks, 3) or (B, n_com_tasks, 3) with MKH_FLAG_COM_BATCHED
 *   v_out          (B, nv)   velocity dq/dt
 *   status_out     (B,)      MKH_ST_* bits (may be NULL)
 */
int32_t mkh_solve(MkhProblem *problem, int32_t B, const double *q, const double *frame_targets,
                  const double *posture_target, const double *com_target, double dt, double damping,
                  double *v_out, int32_t *status_out, int32_t flags, void *hip_stream);

/*
 * Fused outer IK loop on the device — what mink's callers write around solve_ik
 * (examples/arm_ur5e_actuators.py:88-97, examples/arm_aloha.py:146-169):
 *   for _ in range(n_steps): v = solve_ik(cfg, ...); cfg.integrate_inplace(v, dt)
 * q stays on chip between steps.  q_out (B, nq) receives the final configuration (may alias q),
 * v_out the last velocity, status_out the OR of the per-step status bits; an instance stops at the
 * first step whose QP fails.
 */
int32_t mkh_solve_steps(MkhProblem *problem, int32_t B, const double *q, const double *frame_targets,
                        const double *posture_target, const double *com_target, double dt, double damping,
                        int32_t n_steps, double *q_out, double *v_out, int32_t *status_out, int32_t flags,
                        void *hip_stream);

/*
 * mkh_solve / mkh_eval for a problem with dense (plugin) rows: same arguments plus the per-call dense arrays.
 * `taps` may be NULL (plain solve); task_e / task_J tap rows of the dense tasks follow the built-in ones.
 */
int32_t mkh_solve_dense(MkhProblem *problem, int32_t B, const double *q, const double *frame_targets,
                        const double *posture_target, const double *com_target, const MkhDenseRows *dense, double dt,
                        double damping, double *v_out, int32_t *status_out, const MkhTaps *taps, int32_t flags,
                        void *hip_stream);

/*
 * The same loop as the reference's callers REALLY write it (examples/arm_ur5e_actuators.py:88-97,
 * examples/arm_aloha.py:146-169):
 *   for i in range(max_iters):
 *       v = solve_ik(cfg, ...); cfg.integrate_inplace(v, dt)
 *       err = task.compute_error(cfg)
 *       if norm(err[:3]) <= pos_threshold and norm(err[3:]) <= ori_threshold: break      (for every frame task)
 * per instance, in one launch.  The position (orientation) test of a frame task only counts when it has a nonzero
 * position (orientation) cost.  iters_out (B,) = iterations performed (1..max_iters), converged_out (B,) = 1 when the
 * loop ended on the thresholds; q_out / v_out / status_out as in mkh_solve_steps.  iters_out / converged_out may be NULL.
 */
int32_t mkh_solve_until(MkhProblem *problem, int32_t B, const double *q, const double *frame_targets,
                        const double *posture_target, const double *com_target, double dt, double damping,
                        int32_t max_iters, double pos_threshold, double ori_threshold, double *q_out, double *v_out,
                        int32_t *status_out, int32_t *iters_out, int32_t *converged_out, int32_t flags,
                        void *hip_stream);

/* Same inputs; additionally writes the requested intermediates (build_ik / compute_error /
 * compute_jacobian / get_transform_frame_to_world parity taps).  v_out/status_out may be NULL
 * to skip the QP. */
int32_t mkh_eval(MkhProblem *problem, int32_t B, const double *q, const double *frame_targets,
                 const double *posture_target, const double *com_target, double dt, double damping,
                 double *v_out, int32_t *status_out, const MkhTaps *taps, int32_t flags, void *hip_stream);

/* Configuration.integrate (mink/configuration.py:214-226): q_out[b] = q[b] (+) v[b]*dt. */
int32_t mkh_integrate(MkhModel *model, int32_t B, const double *q, const double *v, double dt,
                      double *q_out, int32_t flags, void *hip_stream);

/*
 * The SO3/SE3 device functions of the hot path (mkh lie_dev.h), evaluated element-wise over n inputs — the
 * same code the frame-task lanes run, exposed so that it can be held directly against the reference's
 * known-answer vectors (mink/lie/so3.py, mink/lie/se3.py, mink/lie/base.py:107-156).  Poses are wxyz_xyz (7),
 * rotations wxyz (4), tangents (v, ω) (6); matrices row-major.  `b` may be NULL for unary ops.
 *   MKH_LIE_SE3_LOG      a (n,7)          -> out (n,6)    SE3.log            se3.py:159-185
 *   MKH_LIE_SE3_JLOG     a (n,7)          -> out (n,36)   jlog = rjacinv(log) base.py:150-156
 *   MKH_LIE_SE3_LJACINV  a (n,6)          -> out (n,36)   SE3.ljacinv        se3.py:210-218
 *   MKH_LIE_SE3_MULTIPLY a (n,7), b (n,7) -> out (n,7)    a @ b              se3.py:144-151
 *   MKH_LIE_SE3_INVERSE  a (n,7)          -> out (n,7)    se3.py:136-142
 *   MKH_LIE_SE3_RMINUS   a (n,7), b (n,7) -> out (n,6)    a.rminus(b) = log(b^-1 a)  base.py:111-112
 *   MKH_LIE_SO3_LOG      a (n,4)          -> out (n,3)    SO3.log            so3.py:176-191
 *   MKH_LIE_SO3_MATRIX   a (n,4)          -> out (n,9)    SO3.as_matrix      so3.py:111-114
 *   MKH_LIE_SE3_APPLY    a (n,7), b (n,3) -> out (n,3)    SE3.apply          se3.py:153-157
 */
#define MKH_LIE_SE3_LOG 0
#define MKH_LIE_SE3_JLOG 1
#define MKH_LIE_SE3_LJACINV 2
#define MKH_LIE_SE3_MULTIPLY 3
#define MKH_LIE_SE3_INVERSE 4
#define MKH_LIE_SE3_RMINUS 5
#define MKH_LIE_SO3_LOG 6
#define MKH_LIE_SO3_MATRIX 7
#define MKH_LIE_SE3_APPLY 8
int32_t mkh_lie_eval(int32_t device, int32_t op, int32_t n, const double *a, const double *b, double *out,
                     int32_t flags, void *hip_stream);

/* mujoco.mj_geomDistance(model, data, geom1, geom2, distmax, fromto) — the third-party routine behind every row of
 * CollisionAvoidanceLimit (mink/limits/collision_avoidance_limit.py:214-229) — element-wise on the device routines of the
 * collision phase (primitive geoms; parity tests only, host pointers).  pairs (n, 22): per pair two records of
 * (mjtGeom type, size[3], world position[3], world quaternion wxyz[4]); dist_out (n): the signed distance, distmax when
 * nothing is closer, NaN for a pair of types without a routine; fromto_out (n, 6): the connecting segment, geom1 → geom2. */
int32_t mkh_geom_distance_eval(int32_t device, int32_t n, const double *pairs, double distmax, double *dist_out,
                               double *fromto_out, void *hip_stream);

/* Launch geometry of the most recent solve / eval on this problem (the kernel variant depends on the call; see
 * mkh_problem_last_kernel); before any launch, that of the lean direct variant for a batch of B.  For benchmarks
 * and occupancy reports. */
int32_t mkh_problem_launch_info(const MkhProblem *problem, int32_t B, int32_t *grid, int32_t *block,
                                int32_t *lds_bytes, int32_t *tableau_rows);

#ifdef __cplusplus
}
#endif
#endif /* MINKHIP_H_ */
