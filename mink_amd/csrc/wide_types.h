// Descriptor of the workgroup-per-problem kernel (wide_kernel.h), shared with the host side (minkhip.hip build_wide_problem).
#pragma once
#include <stdint.h>

#include "mkh_types.h"

namespace mkh {

constexpr int kWideThreads = 256;
constexpr int kWideMaxRows = 448;          // half-space rows of one instance (contacts in range + caller's rows); more: MKH_ST_ROW_OVERFLOW

struct WideProblem {
  int32_t nq, nv, nbody, njnt, nlevels, chain_words;
  int32_t n_frame, n_posture, n_com, n_cfg, n_vel, n_pairs, n_jrows;
  int32_t n_dense_tasks, n_dense_rows, n_dense_limit_rows, dense_box, robot_root;
  int32_t max_rows;                        // row capacity of the tableau workspace (≤ kWideMaxRows)
  int32_t tableau_in_lds;                  // the tableau of the LARGEST instance (nv + max_rows) fits in LDS
  int32_t t_lds_doubles;                   // LDS reserved for the tableau: an instance uses it when its own (nv + rows)² fits (round 5:
                                           // ALOHA's 1 104 pairs reserve 448 rows, an instance has 8 on average)
  int32_t o_gi, gi_in_lds;                 // J, R, u, active list of the dense Goldfarb–Idnani fallback: in LDS when they fit
  // rows of the (e, J) tap layout (MkhTaps::task_e / task_J): frame tasks 6 each (FrameTaskDev::row0), then per posture task
  // nv, per CoM task 3, then the caller-defined tasks
  int32_t n_rows_tap, dense_tap_row0;
  int32_t posture_row0[kMaxPostureTasks], com_row0[kMaxComTasks];
  // LDS offsets (doubles)
  int32_t o_q, o_X, o_jnt, o_dof, o_task, o_com, o_we, o_c, o_hd, o_z, o_w, o_lo, o_hi, o_rown, o_ref, o_col, o_red, o_state, o_cws, o_T,
      lds_doubles;
  // eight staging vectors of `blk_stride` doubles for the block pivots of phase 0 and the rank-4 accumulation of H (wide_rank4): on
  // top of the body poses / joint axes when they fit there (dead once the contacts are evaluated), else a region of their own;
  // −1: no room — single pivots, rank-2 accumulation
  int32_t o_blk, blk_stride;
  // model (plain arrays)
  const int32_t *level_start, *level_body, *body_parent, *body_jntadr, *body_jntnum, *body_last, *body_inrobot;
  const double *body_pos, *body_quat, *body_ipos, *body_mass, *body_stmass;
  const int32_t *jnt_type, *jnt_qadr, *jnt_dadr;
  const double *jnt_axis, *jnt_pos, *jnt_qpos0;
  const int32_t *dof_jnt, *dof_kind, *dof_k, *dof_body, *dof_qadr;
  const double *dof_lo, *dof_hi;           // joint range seen by check_limits (±inf: none)
  const uint64_t* chain;                   // [nbody][chain_words]: dofs that move body b (mj_jac's ancestor walk)
  // tasks
  const FrameTaskDev* frame;
  const double* posture_cost;              // [n_posture][nv]
  double posture_gain[kMaxPostureTasks], posture_lm[kMaxPostureTasks];
  double com_cost[kMaxComTasks][3], com_gain[kMaxComTasks], com_lm[kMaxComTasks];
  int32_t com_rowmask[kMaxComTasks], com_jrow0[kMaxComTasks];
  int32_t dense_row0[kMaxDenseTasks], dense_k[kMaxDenseTasks];
  double dense_lm[kMaxDenseTasks];
  const double *dense_cost, *dense_wgain;
  // limits
  const double *cfg_lower, *cfg_upper;     // [n_cfg][nv]
  double cfg_gain[kMaxBoxTerms];
  const double* vel_limit;                 // [n_vel][nv]
  const CollisionPairDev* pairs;
  const PairCull* cull;                    // bounding-sphere records (more than 64 pairs: wide_contacts' cull pass), else nullptr
  // redo launches: the flagged instances of all workgroups go through ONE queue (ring of redo_cap entries, −1 = empty;
  // redo_ctr[0] = head, [1] = tail, never reset — the ring is at least as long as the largest batch)
  int32_t* redo_queue;
  uint32_t* redo_ctr;
  uint32_t redo_cap;                       // power of two ≥ max_batch
  uint32_t redo_pad;
  // per-workgroup slice of device memory: weighted Jacobian rows, pair records, row → pair map, (the tableau)
  double* ws;
  long long ws_stride;                     // doubles per workgroup
  long long ws_gi;                         // dense Goldfarb–Idnani fallback (wide_qp_dense): J, R (nv² each), u, active list — only for problems with rows
  long long ws_jw, ws_rec, ws_rowpair, ws_rank, ws_T;   // (ws_rank: per pair, its rank by h when more contacts are in range than rows — int32)
};

}  // namespace mkh
