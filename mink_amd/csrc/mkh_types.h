// Device-side problem descriptor: the flattened model + task/limit snapshot laid
// out as lane tables (structure-of-arrays with stride 64 so that lane l reads
// table[field*64 + l] coalesced).  Built once on the host in minkhip.hip.
#pragma once
#include <stdint.h>

namespace mkh {

constexpr int kWave = 64;
constexpr int kMaxRounds = 6;      // pointer-jumping rounds: tree depth <= 64
constexpr int kMaxFrameTasks = 16;
constexpr int kMaxPostureTasks = 4;
constexpr int kMaxComTasks = 2;
constexpr int kMaxDenseTasks = 8;  // caller-defined Task subclasses (dense rows)
constexpr int kMaxBoxTerms = 4;    // ConfigurationLimit / VelocityLimit instances each

// dof kinds
enum { DOF_FREE_LIN = 0, DOF_BALL = 1, DOF_SLIDE = 2, DOF_HINGE = 3, DOF_FREE_ANG = 4 };
// joint types (MuJoCo mjtJoint)
enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };

// body lane table fields (double)
enum { BF_POS = 0, BF_QUAT = 3, BF_IPOS = 7, BF_MASS = 10, BF_SUBTREEMASS = 11, BF_COUNT = 12 };
// body lane table fields (int)
enum { BI_PARENT = 0, BI_JNTADR = 1, BI_JNTNUM = 2, BI_SUBTREE_LAST = 3, BI_IN_ROBOT = 4, BI_ANC0 = 5,
       // the pointer-jumping ancestors once more, 6 bits each (bodies < 64): rounds 0-4 in one word, round 5 in the next —
       // one load at the start of FK instead of one dependent L2 round trip per round
       BI_ANCPACK0 = BI_ANC0 + kMaxRounds, BI_ANCPACK1 = BI_ANCPACK0 + 1, BI_COUNT = BI_ANCPACK1 + 1 };
// joint arrays (indexed by joint id, double)
enum { JF_AXIS = 0, JF_POS = 3, JF_QPOS0 = 6, JF_COUNT = 7 };
enum { JI_TYPE = 0, JI_QADR = 1, JI_DADR = 2, JI_COUNT = 3 };
// dof lane table (int)
enum { DI_JNT = 0, DI_KIND = 1, DI_K = 2, DI_BODY = 3, DI_QADR = 4, DI_JIDX = 5, DI_COUNT = 6 };
// dof lane table (double)
enum { DF_RANGE_LO = 0, DF_RANGE_HI = 1, DF_COUNT = 2 };

struct FrameTaskDev {
  int32_t body;          // body the frame is attached to
  int32_t pad;
  double lpos[3];        // frame pose in the body frame
  double lquat[4];
  double cost[6];
  double gain, lm_damping;
  uint64_t dof_mask;     // dofs on the chain world -> body (mj_jac's ancestor walk)
  int32_t row0;          // first row in the (e, J) tap layout
  int32_t any_ori;       // any orientation cost > 0
  int32_t rowmask;       // bit r set iff cost[r] > 0: only those rows are staged in LDS / enter H
  int32_t jrow0;         // first compact LDS row of this task
  int32_t relative;      // RelativeFrameTask: pose of the frame in the root frame
  int32_t root_body;
  double root_lpos[3];
  double root_lquat[4];
  uint64_t root_mask;
};

struct CollisionPairDev {
  int32_t type1, type2, body1, body2;
  double size1[3], size2[3];
  double lpos1[3], lpos2[3];
  double lquat1[4], lquat2[4];
  uint64_t mask1, mask2;
  double gain, dmin, ddetect, relax;
  // mesh geoms: hull vertices in the geom frame (device memory of the model, 3 doubles each); nullptr / 0 for primitives
  const double* vert1;
  const double* vert2;
  int32_t nvert1, nvert2;
  // ≥ 0: a pair WITHOUT an analytic routine whose contact of a plain solve is evaluated by the kernel in front of the solve
  // (convex_pre.hip) — slot of its (dist, from, to) record in DeviceProblem::cv_contacts; −1: analytic pair
  int32_t cv_slot, cv_pad;
};

// Bounding-sphere record of a collision pair (problems with more than one wavefront of pairs: collision_phase's cull pass): the two
// geom centres in their body frames and reach² = ((rbound1 + rbound2 + detection distance)·(1 + 1e-9))² — a pair whose centres are
// farther apart than that is beyond the detection distance, which is all mj_geomDistance says about it (returns distmax:
// collision_avoidance_limit.py:214-229, Contact.inactive :52-56).  +inf with a plane: never culled.
struct PairCull {
  int32_t body1, body2;
  double lpos1[3], lpos2[3];
  double reach2;
};

struct DeviceProblem {
  // sizes
  int32_t nq, nv, nbody, njnt, nrounds;
  int32_t n_frame, n_posture, n_com, n_cfg, n_vel, n_pairs, n_rows_tap;
  int32_t max_rows;      // tableau rows reserved for half-spaces (ntab = nv + max_rows)
  int32_t n_jrows;       // weighted Jacobian rows staged in LDS (Σ nonzero-cost rows of frame + CoM tasks)
  // low-rank start: lane l computes rows [wood_row0[l], +wood_rpc) of column wood_col[l] of Jh·Jhᵀ
  // (row n_jrows = the right-hand side); −1 = idle lane
  int32_t wood_rpc;
  int32_t wood_col[64], wood_row0[64];
  uint64_t wood_mask[64];  // dofs on the kinematic chain of the task that owns column wood_col[l] (the only nonzeros of that Jh row)
  // low-rank start: Jacobian columns by (task, dof) pair lanes — pair i = column jpair_dof[i] of frame task
  // jpair_task[i] (only dofs on the task's kinematic chain); mu_src[r] = offset in the task LDS block of the
  // weighted error of compact row r
  int32_t n_jpairs;
  int16_t jpair_task[256], jpair_dof[256];
  int16_t mu_src[64];
  // direct start: (task, dof) pair lanes of every frame task's Jacobian (dofs on the chain of the frame or of the root
  // frame of a RelativeFrameTask); 0 when there are more than 64 pairs (then the dof lanes compute task after task)
  int32_t n_dpairs;
  int16_t dpair_task[64], dpair_dof[64];
  int32_t nt;            // tableau rows per lane of the compiled kernel variant (row stride of the J rows)
  int32_t prefetch_w3;   // the same decision for the 3-waves-per-SIMD variants (compact LDS layout), direct start
  int32_t prefetch_w3w;  // ... and low-rank start
  int32_t wood_refine;   // low-rank cold start: second elimination on the bounds the unconstrained minimiser violates
  int32_t wood_compact;  // low-rank start on the 2-waves map: use the compact LDS layout too (the plain one would cost a resident wave)
  int32_t prefetch_wc;   // ... and the prefetch decision for that layout
  int32_t prefetch;      // the next problem's q / targets are fetched into second LDS buffers (host: only if that costs no residency)
  int32_t robot_root;    // body 1 (ComTask subtree root)
  // model lane tables
  const double* body_f;  // [BF_COUNT][64]
  const int32_t* body_i; // [BI_COUNT][64]
  const double* jnt_f;   // [njnt][JF_COUNT]
  const int32_t* jnt_i;  // [njnt][JI_COUNT]
  const int32_t* dof_i;  // [DI_COUNT][64]
  const double* dof_f;   // [DF_COUNT][64]
  // tasks
  const FrameTaskDev* frame;       // [n_frame]
  const double* posture_cost;      // [n_posture][64]
  double posture_gain[kMaxPostureTasks], posture_lm[kMaxPostureTasks];
  int32_t posture_row0[kMaxPostureTasks];
  double com_cost[kMaxComTasks][3], com_gain[kMaxComTasks], com_lm[kMaxComTasks];
  int32_t com_row0[kMaxComTasks];
  int32_t com_rowmask[kMaxComTasks], com_jrow0[kMaxComTasks];
  // limits
  const double* cfg_lower;         // [n_cfg][64] per dof (value at the dof's qpos address), ±inf when absent
  const double* cfg_upper;         // [n_cfg][64]
  double cfg_gain[kMaxBoxTerms];
  const double* vel_limit;         // [n_vel][64], +inf when absent
  const CollisionPairDev* pairs;   // [n_pairs]
  // plugin route: caller-defined tasks as dense rows (e, J per instance in SolveArgs) and caller-defined limit rows
  int32_t n_dense_tasks, n_dense_rows, n_dense_limit_rows, dense_tap_row0;
  int32_t dense_box;               // caller-defined limits also hand over per-instance box rows (SolveArgs::dense_lo / dense_hi)
  int32_t dense_row0[kMaxDenseTasks], dense_k[kMaxDenseTasks];
  double dense_lm[kMaxDenseTasks];
  const double* dense_cost;        // [n_dense_rows] cost of each row
  const double* dense_wgain;       // [n_dense_rows] cost·(−gain) of each row: weighted error = wgain·e
  // (appended last: the offsets of the fields above are what the register allocation of the W3 builds was tuned on —
  //  one int32 in the middle cost the headline kernel 25 spilled VGPRs)
  int32_t n_hsel;        // doubles of LDS behind the per-problem ranges: n_pairs (even) when there are more pairs than rows (h of every pair: row selection) + the expanding polytope's workspace when a pair needs the general convex routine
  // general convex pairs of plain solves: (dist, from[3], to[3]) per (instance, convex pair), written by convex_contacts_kernel
  // in front of the solve and read by the analytic collision build (round 5: the two-kernel split)
  int32_t n_cv, n_cv_pad;
  const double* cv_contacts;       // [max_batch][n_cv][7], owned by the problem handle
  const PairCull* cull;  // one record per pair when n_pairs > 64, else nullptr (ik_kernel.h collision_phase)
};

struct SolveArgs {
  int32_t B;
  int32_t posture_batched, com_batched, do_qp;
  const double* q;                 // (B, nq)
  const double* frame_targets;     // (B, n_frame, 7)
  const double* posture_target;    // (n_posture, nq) or (B, n_posture, nq)
  const double* com_target;        // (n_com, 3) or (B, n_com, 3)
  double dt, damping;
  double* v_out;                   // (B, nv)
  int32_t* status_out;             // (B,)
  int32_t n_steps;                 // fused outer loop: (solve, q ← q ⊕ v·dt) repeated n_steps times
  double* q_out;                   // (B, nq) configuration after the last step (nullable)
  // problem distribution (ik_kernel.h): the first static_rounds·gridDim.x rows are split into one contiguous range
  // per XCD and taken statically; the rest of the batch is handed out one problem at a time through the ticket
  // counter — problem static_rounds·gridDim.x + atomicAdd(work_counter, 1) until that is ≥ B.  A launch draws
  // exactly (B − static_rounds·gridDim.x) + gridDim.x tickets (every wave ends on one rejected draw); whoever
  // draws the last of them knows that no draw is outstanding and zeroes the counter, so every launch starts from
  // zero with no host-side state: a launch can be replayed (hipGraph) or fail without consequences.
  // static_rounds = INT32_MAX: no tickets at all (every XCD owns a contiguous eighth of the batch).
  uint32_t* work_counter;
  int32_t static_rounds;
  // threshold-terminated fused loop (mkh_solve_until): pos_threshold < 0 disables it (fixed n_steps)
  double pos_threshold, ori_threshold;
  int32_t* iters_out;              // (B,) solve + integrate iterations performed
  int32_t* converged_out;          // (B,) 1 when every frame task ended within the thresholds
  // plugin route (variants with every feature): dense task rows and dense limit rows of every instance
  const double* dense_e;           // (B, n_dense_rows)
  const double* dense_J;           // (B, n_dense_rows, nv)
  const double* dense_G;           // (B, n_dense_limit_rows, nv)
  const double* dense_h;           // (B, n_dense_limit_rows)
  const double* dense_lo;          // (B, nv) or nullptr: box rows of caller-defined limits (single-entry rows of G, folded by the caller)
  const double* dense_hi;          // (B, nv) or nullptr
  // warm start across CALLS (MKH_FLAG_WARM_START): where every dof of every instance ended the previous solve of this
  // problem handle (0 free, 1 at its lower bound, 2 at its upper), read at the start of the active-set phase when the
  // state is at least two solves old and written at its end; nullptr = cold
  int8_t* warm;                    // (B, nv)
  int32_t warm_age;
  // != 0: a REDO launch — only the problems whose status_out has one of these bits are solved (again), the others are
  // skipped where the wave draws its next problem (collision builds: the full-row launch behind a tight-rows one)
  int32_t redo_mask;
  // cycle stamps of every problem, (B, 24) — 16 as the `cycles` tap + 8 finer ones — read only by kernels compiled with -DMKH_CLOCKS (experiment
  // builds, tools/phase_clocks.py); nullptr otherwise
  long long* clk;
};

// Debug/parity taps (nullable pointers).  Lives in device memory and is passed by pointer so
// that a production launch (taps == nullptr) does not pin 28 SGPRs of null pointers.
struct TapArgs {
  double *t_xpos, *t_xquat, *t_frame_pose, *t_subtree_com, *t_task_e, *t_task_J, *t_H, *t_c, *t_box_lo,
      *t_box_hi, *t_coll_G, *t_coll_h;
  int32_t* t_qp_iters;
  long long* t_cycles;             // (B, 16) shader-clock stamps at phase boundaries + QP sub-phase sums (profiling)
};

}  // namespace mkh
