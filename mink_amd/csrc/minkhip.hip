// libminkhip.so — host side of the C ABI declared in include/minkhip.h.
//
// mkh_model_create / mkh_problem_create flatten the mjModel fields and the Task/Limit
// plugin objects of one mink.solve_ik call site into lane tables on the device
// (mkh_types.h); mkh_solve launches the one-wavefront-per-problem kernel (ik_kernel.h).
#include "../../include/minkhip.h"

#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "ik_kernel.h"
#include "lane_kernel.h"
#include "wide_types.h"

namespace mkh {
struct CvPre { int32_t n_cv; const int32_t* pair; const int32_t* chain_adr; const int32_t* chain; };     // (convex_pre.hip)
}
using namespace mkh;

// Experiment switches (A/B runs, sweeps, phase censuses): environment variables MKH_DEBUG_* that change kernel selection, launch
// shapes or caps.  They exist only in builds made with -DMKH_DEBUG_SWITCHES (MKH_EXTRA_FLAGS=-DMKH_DEBUG_SWITCHES python -m
// mink_amd.csrc.build, normally under an MKH_BUILD_TAG): the product library never reads the environment, so a stray variable
// cannot change a production handle's results or speed (round-5 review).  The four switches the parity tests and bench.py need
// are per-handle diagnostic options of the ABI instead: mkh_problem_create_diag, MKH_DIAG_*.
#if defined(MKH_CLOCKS) && !defined(MKH_DEBUG_SWITCHES)
#define MKH_DEBUG_SWITCHES 1     // (a clock build is an experiment build: MKH_DEBUG_CLOCKS / MKH_DEBUG_PHASE_STOP drive its stamps)
#endif
#ifdef MKH_DEBUG_SWITCHES
static inline const char* dbg_env(const char* name) { return getenv(name); }
#else
static inline const char* dbg_env(const char*) { return nullptr; }
#endif
static thread_local std::string g_err;
static int32_t fail(int32_t code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_OK(expr)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(MKH_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                       \
  } while (0)

template <class T>
static hipError_t upload(const std::vector<T>& h, T** d) {
  *d = nullptr;
  const size_t n = h.empty() ? 1 : h.size();
  hipError_t e = hipMalloc((void**)d, n * sizeof(T));
  if (e != hipSuccess) return e;
  if (!h.empty()) e = hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

// Small host-pointer calls (a control loop's single configuration: the reference's everyday use) go through ONE pinned,
// device-visible host buffer: the kernel reads its inputs and writes its outputs across the bus itself — a launch and a
// synchronize instead of five staged copies of ≈8 µs host time each, whatever their size.
constexpr size_t kSmallCallBytes = 64 << 10;
struct PinnedScratch {
  char* host = nullptr;
  char* dev = nullptr;
  hipError_t ensure() {
    if (host) return hipSuccess;
    if (hipError_t e = hipHostMalloc((void**)&host, kSmallCallBytes, hipHostMallocDefault)) { host = nullptr; return e; }
    if (hipError_t e = hipHostGetDevicePointer((void**)&dev, host, 0)) { (void)hipHostFree(host); host = nullptr; return e; }
    return hipSuccess;
  }
  void release() { if (host) (void)hipHostFree(host); host = dev = nullptr; }
};

struct MkhModel {
  int device = 0;
  PinnedScratch small;             // mkh_integrate, host pointers
  std::mutex small_mutex;          // one model serves many callers (every Configuration of a FlatModel shares it): the small
                                   // host-pointer path of mkh_integrate is serialised, so the entry point stays re-entrant
  int nq = 0, nv = 0, nbody = 0, njnt = 0, ngeom = 0, nsite = 0, nrounds = 0;
  // host copies needed when problems are created
  std::vector<int32_t> body_parentid, body_rootid, body_jntnum, body_jntadr, body_dofnum, body_dofadr;
  std::vector<int32_t> jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited;
  std::vector<int32_t> dof_bodyid, dof_jntid, dof_parentid, site_bodyid, geom_bodyid, geom_type;
  std::vector<double> body_pos, body_quat, site_pos, site_quat, geom_size, geom_pos, geom_quat, jnt_range;
  std::vector<double> jnt_pos, jnt_axis, jnt_qpos0;   // (per joint; qpos0 at the joint's first qpos address)
  std::vector<double> body_ipos, body_mass, body_subtreemass;
  bool big = false;                // more than 64 bodies or dofs: no lane tables — every problem runs on the workgroup-per-problem kernel
  std::vector<int32_t> geom_dataid, mesh_vertadr, mesh_vertnum;   // mesh geoms: hull vertices (geom frame) in d_mesh_vert
  double* d_mesh_vert = nullptr;
  std::vector<double> h_mesh_vert;   // (host copy: bounding radii of mesh geoms, mkh_problem_create)
  // device tables
  double* d_body_f = nullptr;
  int32_t* d_body_i = nullptr;
  double* d_jnt_f = nullptr;
  int32_t* d_jnt_i = nullptr;
  int32_t* d_dof_i = nullptr;
  double* d_dof_f = nullptr;
  int num_cus = 0;
};

struct MkhProblem {
  PinnedScratch small;             // run(), host pointers
  MkhModel* model = nullptr;
  int device = 0;                  // (copied: the destructor must not depend on the model still being alive)
  DeviceProblem dev{};
  int nt = 8;  // tableau rows per lane (compiled variants: multiples of 8)
  int max_batch = 0;
  int lds_bytes = 0;
  int blocks_per_cu = 1;
  // feature-rich variants (taps / ComTask / collisions / RelativeFrameTask) need the compiler's full VGPR
  // budget: they never use the high-occupancy register maps of NT ≤ 24 (ik_kernel.h MKH_WAVES)
  int nt_full = 8, lds_bytes_full = 0;
  // tight rows (capsule-only collision sets on a small tableau): the same problem with fewer half-space rows than pairs — the
  // tightest contacts get them, dropped ones are checked at the solution — launched first; the full-row variant then re-solves
  // what it flagged (SolveArgs::redo_mask)
  DeviceProblem* d_dev_tight = nullptr;
  int32_t* d_status_tight = nullptr;   // status of a call with a redo launch whose caller passed no status_out (the redo launch reads it)
  // the workgroup-per-problem kernel (wide_kernel.h): every call of a model beyond one wavefront (wide_only), or the redo
  // launch of the instances a wavefront kernel flagged MKH_ST_ROW_OVERFLOW
  WideProblem wide{};
  WideProblem* d_wide = nullptr;
  bool wide_only = false;
  int wide_grid = 0, wide_lds = 0;
  std::vector<void*> wide_allocs;
  int nt_tight = 0, lds_tight = 0;
  // 3-waves-per-SIMD register map + compact LDS layout (ik_kernel.h MKH_W3; variants without collision rows):
  // LDS bytes per wavefront, 0 when no such variant is compiled for this tableau size or it would not reach 12 waves per CU
  int lds_bytes_w3 = 0;
  bool has_relative = false;
  bool simple_pairs = false;       // every collision pair is plane / sphere / capsule (F_SIMPLE_COLL variants)
  bool convex_pairs = false;       // some pair has no analytic routine: general convex distance (F_CONVEX_COLL variants)
  // low-rank ("Woodbury") start of the QP (ik_kernel.h F_WOOD): compiled (NT, NR) pair or 0 when the
  // problem does not qualify; lower bound of the diagonal part of H without the damping argument, and
  // the largest squared task cost (conditioning gate, evaluated per call because damping is a call argument)
  int wood_nt = 0, wood_nr = 0, wood_lds_bytes = 0, wood_lds_bytes_w3 = 0;
  // low-rank start WITH half-space rows (round 6; `48_40_r48`): the lane tables are in the descriptor(s), LDS bytes of that layout
  // for the tight-rows descriptor and for the main one (0: not available there)
  bool woodr_tables = false;
  int woodr_lds_tight = 0, woodr_lds = 0;
  bool wood_big = false;           // more than kMu task rows, or S columns in a second register set: the F_COM builds carry those
  double wood_min_diag = 0.0, wood_max_cost2 = 0.0;
  // lane-per-problem kernel for small arms (lane_kernel.h): template size (0 = the problem does not qualify)
  int lane_nv = 0, lane_lds = 0;   // lane kernel's template size (0: not eligible)
  int quad_nt = 0;                 // row kernel's column registers, 8 or 16 (0: not eligible)
  LaneProblem* d_lane = nullptr;
  LaneDims lane_dims{};
  char last_kernel[64] = "";
  // device descriptor storage
  FrameTaskDev* d_frame = nullptr;
  double* d_posture_cost = nullptr;
  double *d_cfg_lower = nullptr, *d_cfg_upper = nullptr, *d_vel = nullptr;
  CollisionPairDev* d_pairs = nullptr;
  PairCull* d_cull = nullptr;          // bounding-sphere records of the pairs (problems with more than 64 of them)
  bool use_cull = false;
  int32_t diag = 0;                // MKH_DIAG_* of mkh_problem_create_diag (parity / measurement switches of this handle)
  double *d_dense_cost = nullptr, *d_dense_wgain = nullptr;
  DeviceProblem* d_dev = nullptr;   // device copy of `dev` (the kernel reads the descriptor from memory)
  TapArgs* d_taps = nullptr;
  int last_grid = 0, last_lds = 0, last_nt = 0, last_block = kWave;   // geometry of the most recent launch (mkh_problem_launch_info)
  long long* d_clk = nullptr;      // MKH_DEBUG_CLOCKS (experiment builds): cycle stamps of the last launch
  // general convex pairs of plain solves: evaluated by convex_contacts_kernel in front of the analytic collision build
  mkh::CvPre cv{};
  double* d_cv = nullptr;          // [max_batch][n_cv][7]
  int32_t *d_cv_pair = nullptr, *d_cv_adr = nullptr, *d_cv_chain = nullptr;
  double* d_qkeep = nullptr;       // fused loops with a redo launch behind them: the call's q as it came (q_out may alias it)
  int8_t* d_warm = nullptr;        // MKH_FLAG_WARM_START: active set of every instance after the previous solve (max_batch × nv)
  int warm_age = 0, warm_B = 0;    // solves since the state was reset / the batch size it belongs to
  uint32_t* d_work = nullptr;      // ticket counter of the dynamic problem distribution (zeroed by the kernel's last draw)
  // staging buffers for host-pointer calls
  double *s_q = nullptr, *s_ft = nullptr, *s_pt = nullptr, *s_ct = nullptr, *s_v = nullptr;
  // host-pointer calls on large batches: copies of chunk c + 1 / c − 1 run beside the kernel of chunk c
  hipStream_t st_in = nullptr, st_out = nullptr;
  hipEvent_t ev_start = nullptr, ev_in[4] = {nullptr, nullptr, nullptr, nullptr}, ev_k[4] = {nullptr, nullptr, nullptr, nullptr};
  int32_t* s_status = nullptr;
  int32_t* s_iters = nullptr;      // [2][max_batch]: iterations, converged (mkh_solve_until, host-pointer calls)
  size_t s_pt_cap = 0, s_ct_cap = 0;
  double *s_de = nullptr, *s_dJ = nullptr, *s_dG = nullptr, *s_dh = nullptr, *s_dbox = nullptr;   // dense (plugin) rows
};

// Kernel variants live in their own translation units (mink_amd/csrc/build.py generates one
// variant_<NT>_<FEAT>.hip per compiled combination so that they build in parallel); this is the
// generated dispatcher.
namespace mkh {
int launch_variant(int nt, int nr, int feat, bool w3, int grid, int lds_bytes, hipStream_t stream, const DeviceProblem* P,
                   const SolveArgs& a, const TapArgs* taps, bool one_shot = false);
int launch_lane(int nv_max, bool loop, int grid, int lds_bytes, hipStream_t stream, const LaneProblem* P, const SolveArgs& a);
int launch_quad(int nt, bool loop, int grid, hipStream_t stream, const void* P, const LaneDims& dims, const SolveArgs& a);   // returns its LDS bytes per wavefront
int launch_wide(int grid, int lds_bytes, hipStream_t stream, const WideProblem* P, const SolveArgs& a, const TapArgs* taps, bool convex);
int launch_convex_pre(hipStream_t stream, const WideProblem* P, const CvPre& C, int B, const double* q, double* out);
constexpr int kLaneMinBatchLoop = 28672;  // fused loops of a small arm: row kernel below, lane kernel from here (M targets/s at 16 384: 39.7 vs 24.1, at 32 768: 42.4 vs 48.2)
constexpr int kLaneMinBatch = 73728;  // plain solves of a small arm: row kernel below, lane kernel from here (launch())
}

// Descriptor of the row- and lane-per-problem kernels (quad_kernel.h, lane_kernel.h) of a problem that qualifies: nv ≤ 16
// (row kernel; the lane kernel: nv ≤ 8), at most 16 links on the frames' chains, hinge / slide joints only, plain
// FrameTasks (≤ 8) + PostureTasks + box limits.  Returns the size class (4 / 6 / 7 / 8: both kernels, 16: the row
// kernel only), 0 when the problem stays on the wavefront kernel.
static int build_lane_problem(const MkhModel* m, const MkhProblemDesc* d, const DeviceProblem& P,
                              const std::vector<FrameTaskDev>& ft, const std::vector<double>& pcost,
                              const std::vector<double>& clo, const std::vector<double>& chi,
                              const std::vector<double>& vlim, bool has_relative, LaneProblem2& L) {
  const double inf = std::numeric_limits<double>::infinity();
  if (m->nv > kLaneDescDofs2) return 0;
  if (P.n_frame < 1 || P.n_frame > kLaneMaxFrames || P.n_com > 1 || P.n_pairs || P.n_dense_rows ||
      P.n_dense_limit_rows || P.dense_box)       // (dense_box: per-instance box rows of a plugin limit, wavefront kernels only)
    return 0;
  const bool com = P.n_com == 1;                 // ComTask: the two-row build of the row kernel, every body of the robot a link
  if (com && (m->big || m->body_mass.empty())) return 0;
  // hinge / slide joints; the two-row build of the row kernel (17 … 32 dofs or links) also takes free joints — a floating base
  bool has_free = false;
  for (int j = 0; j < m->njnt; ++j) {
    if (m->jnt_type[j] == JNT_FREE && m->body_jntnum[m->jnt_bodyid[j]] == 1) { has_free = true; continue; }
    if (m->jnt_type[j] != JNT_HINGE && m->jnt_type[j] != JNT_SLIDE) return 0;
  }
  if (m->nq != m->nv + (has_free ? 1 : 0)) return 0;      // (one free joint at most: its quaternion is the extra coordinate)
  memset(&L, 0, sizeof L);
  L.nq = m->nq; L.nv = m->nv; L.n_frame = P.n_frame; L.n_posture = P.n_posture; L.n_cfg = P.n_cfg; L.n_vel = P.n_vel;
  // links = the bodies on the chains world → frame bodies, in body-id order (parents first)
  std::vector<int> need(m->nbody, 0), link_of(m->nbody, -1);
  for (int t = 0; t < P.n_frame; ++t) {
    for (int b = ft[t].body; b > 0; b = m->body_parentid[b]) need[b] = 1;
    if (ft[t].relative)                              // RelativeFrameTask: the root frame's chain as well (two-row build)
      for (int b = ft[t].root_body; b > 0; b = m->body_parentid[b]) need[b] = 1;
  }
  std::vector<char> in_robot(m->nbody, 0);           // subtree of body 1 (mj_jacSubtreeCom(m, d, jac, 1): com_task.py:71-97)
  if (com)
    for (int b = 1; b < m->nbody; ++b) {
      in_robot[b] = b == 1 || in_robot[m->body_parentid[b]];
      if (in_robot[b]) need[b] = 1;
    }
  int nl = 0;
  int free_link[4] = {-1, -1, -1, -1};
  std::vector<int> link_of_jnt(m->njnt, -1);
  // Pose of a jointless body relative to its nearest ancestor that is a link (or the world): folded into the local
  // transforms of its children and of the frames attached to it, so that it costs neither a link nor LDS.
  struct Xf { double p[3], q[4]; };
  auto compose = [](const Xf& a, const double* p, const double* q) {     // a ∘ (p, q)
    Xf r;
    const double w = a.q[0], x = a.q[1], y = a.q[2], z = a.q[3];
    const double tx = 2 * (y * p[2] - z * p[1]), ty = 2 * (z * p[0] - x * p[2]), tz = 2 * (x * p[1] - y * p[0]);
    r.p[0] = a.p[0] + p[0] + w * tx + (y * tz - z * ty);
    r.p[1] = a.p[1] + p[1] + w * ty + (z * tx - x * tz);
    r.p[2] = a.p[2] + p[2] + w * tz + (x * ty - y * tx);
    r.q[0] = w * q[0] - x * q[1] - y * q[2] - z * q[3];
    r.q[1] = w * q[1] + x * q[0] + y * q[3] - z * q[2];
    r.q[2] = w * q[2] - x * q[3] + y * q[0] + z * q[1];
    r.q[3] = w * q[3] + x * q[2] - y * q[1] + z * q[0];
    return r;
  };
  const Xf ident{{0, 0, 0}, {1, 0, 0, 0}};
  std::vector<Xf> fold(m->nbody, ident);
  std::vector<char> folded(m->nbody, 0);
  folded[0] = 1;                                      // the world: identity, link −1
  for (int b = 1; b < m->nbody; ++b) {
    if (!need[b]) continue;
    const int pb = m->body_parentid[b];
    const Xf local = folded[pb] ? compose(fold[pb], &m->body_pos[3 * b], &m->body_quat[4 * b])
                                : compose(ident, &m->body_pos[3 * b], &m->body_quat[4 * b]);
    if (m->body_jntnum[b] == 0) {                     // jointless: fold
      fold[b] = local; folded[b] = 1; link_of[b] = link_of[pb];
      continue;
    }
    // A body with k > 1 joints becomes a chain of k links with identity offsets: mj_kinematics applies a body's
    // joints one after the other in the moving body frame, which is exactly a serial chain of coincident frames.
    if (m->jnt_type[m->body_jntadr[b]] == JNT_FREE) {
      // mj_kinematics: a free body's pose IS its qpos (body_pos / body_quat are not used): three slide links along the world
      // axes from the origin, then the rotation
      const int jj = m->body_jntadr[b];
      if (nl + 4 > kLaneMaxLinks2) return 0;
      for (int i = 0; i < 4; ++i) {
        LaneLink& k = L.link[nl];
        k.parent = i == 0 ? -1 : nl - 1;
        k.quat[0] = 1.0;
        k.jtype = i < 3 ? JNT_SLIDE : JNT_BALL;
        k.dof = m->jnt_dofadr[jj] + (i < 3 ? i : 3);
        k.qadr = m->jnt_qposadr[jj] + 3;
        if (i < 3) k.axis[i] = 1.0;
        free_link[i] = nl++;
      }
      link_of_jnt[jj] = nl - 1;
      link_of[b] = nl - 1;
      continue;
    }
    for (int i = 0; i < m->body_jntnum[b]; ++i) {
      if (nl >= kLaneMaxLinks2) return 0;
      LaneLink& k = L.link[nl];
      k.parent = i == 0 ? link_of[pb] : nl - 1;
      k.quat[0] = 1.0;
      if (i == 0) {
        for (int c = 0; c < 3; ++c) k.pos[c] = local.p[c];
        for (int c = 0; c < 4; ++c) k.quat[c] = local.q[c];
      }
      const int jj = m->body_jntadr[b] + i;
      k.jtype = m->jnt_type[jj]; k.dof = m->jnt_dofadr[jj]; k.qpos0 = m->jnt_qpos0[jj];
      for (int c = 0; c < 3; ++c) { k.axis[c] = m->jnt_axis[3 * jj + c]; k.jpos[c] = m->jnt_pos[3 * jj + c]; }
      link_of_jnt[jj] = nl;
      link_of[b] = nl++;
    }
  }
  L.nlink = nl;
  for (int dd = 0; dd < kLaneDescDofs2; ++dd) { L.dof_link[dd] = -1; L.dof_qadr[dd] = 0; L.range_lo[dd] = -inf; L.range_hi[dd] = inf; }
  for (int dd = 0; dd < m->nv; ++dd) {
    const int j = m->dof_jntid[dd];
    if (m->jnt_type[j] == JNT_FREE) {
      // dofs 0-2: translations along the world axes (the slide links); 3-5: rotations about the body's own axes through its origin
      const int k = dd - m->jnt_dofadr[j];
      L.dof_link[dd] = link_of_jnt[j] < 0 ? -1 : (k < 3 ? free_link[k] : free_link[3]);
      L.dof_qadr[dd] = m->jnt_qposadr[j] + (k < 3 ? k : k + 1);      // (rotations: any finite entry — their posture cost is zero, mink/tasks/posture_task.py:95-99)
      L.dof_axis[dd][k % 3] = 1.0;
      L.dof_slide[dd] = k < 3;
      continue;
    }
    L.dof_link[dd] = link_of_jnt[j];
    L.dof_qadr[dd] = m->jnt_qposadr[j];
    for (int c = 0; c < 3; ++c) { L.dof_axis[dd][c] = m->jnt_axis[3 * j + c]; L.dof_jpos[dd][c] = m->jnt_pos[3 * j + c]; }
    L.dof_slide[dd] = m->jnt_type[j] == JNT_SLIDE;
    if (m->jnt_limited[j]) { L.range_lo[dd] = m->jnt_range[2 * j]; L.range_hi[dd] = m->jnt_range[2 * j + 1]; }
  }
  if (com) {
    L.n_com = 1;
    L.com_rowmask = P.com_rowmask[0];
    for (int k = 0; k < 3; ++k) L.com_cost[k] = P.com_cost[0][k];
    L.com_gain = P.com_gain[0]; L.com_lm = P.com_lm[0];
    double mtot = 0.0;
    std::vector<double> wsum(3 * (nl > 0 ? nl : 1), 0.0);        // Σ m·(centre of mass in the link frame) per link
    for (int b = 1; b < m->nbody; ++b) {
      if (!in_robot[b]) continue;
      const double mb = m->body_mass[b];
      mtot += mb;
      // the body's centre of mass in the frame of its link (its own frame, or through the folded jointless chain) — or in the world
      Xf at = ident;
      if (m->body_jntnum[b] == 0) at = fold[b];
      const Xf cw = compose(at, &m->body_ipos[3 * b], ident.q);
      const int a = link_of[b];
      if (a < 0) { for (int k = 0; k < 3; ++k) L.com_static[k] += mb * cw.p[k]; continue; }
      L.link_mass[a] += mb;
      for (int k = 0; k < 3; ++k) wsum[3 * a + k] += mb * cw.p[k];
    }
    if (!(mtot > 0.0)) return 0;
    L.com_minv = 1.0 / mtot;
    for (int a = 0; a < nl; ++a) {
      for (int k = 0; k < 3; ++k) L.link_ipos[a][k] = L.link_mass[a] > 0.0 ? wsum[3 * a + k] / L.link_mass[a] : 0.0;
      // links are in body order (depth first): the subtree of a link is the range up to the next link that is not below it
      int last = a;
      for (int c = a + 1; c < nl; ++c) {
        int up = L.link[c].parent;
        while (up > a) up = L.link[up].parent;
        if (up != a) break;
        last = c;
      }
      L.link_last[a] = last;
      for (int c = a; c <= last; ++c) L.link_stmass[a] += L.link_mass[c];
    }
    for (int dd = 0; dd < m->nv; ++dd) if (L.dof_link[dd] < 0) return 0;     // (every dof of the robot moves mass: all need their link)
  }
  for (int t = 0; t < P.n_frame; ++t) {
    LaneFrame& f = L.frame[t];
    f.link = link_of[ft[t].body];
    f.chain = (uint32_t)ft[t].dof_mask;
    f.rowmask = ft[t].rowmask;
    const Xf fl = folded[ft[t].body] ? compose(fold[ft[t].body], ft[t].lpos, ft[t].lquat) : compose(ident, ft[t].lpos, ft[t].lquat);
    for (int i = 0; i < 3; ++i) f.lpos[i] = fl.p[i];
    for (int i = 0; i < 4; ++i) f.lquat[i] = fl.q[i];
    for (int i = 0; i < 6; ++i) f.cost[i] = ft[t].cost[i];
    f.gain = ft[t].gain; f.lm_damping = ft[t].lm_damping;
    if (ft[t].relative) {
      auto& r = L.rel[t];
      r.relative = 1;
      r.root_link = link_of[ft[t].root_body];
      r.rchain = (uint32_t)ft[t].root_mask;
      const Xf rl = folded[ft[t].root_body] ? compose(fold[ft[t].root_body], ft[t].root_lpos, ft[t].root_lquat)
                                             : compose(ident, ft[t].root_lpos, ft[t].root_lquat);
      for (int i = 0; i < 3; ++i) r.rlpos[i] = rl.p[i];
      for (int i = 0; i < 4; ++i) r.rlquat[i] = rl.q[i];
    }
  }
  for (int t = 0; t < P.n_posture; ++t) {
    for (int dd = 0; dd < m->nv; ++dd)      // (free-joint dofs: error and Jacobian column are zero, posture_task.py:115-116,139-141)
      L.posture_cost[t][dd] = m->jnt_type[m->dof_jntid[dd]] == JNT_FREE ? 0.0 : pcost[t * 64 + dd];
    L.posture_gain[t] = P.posture_gain[t]; L.posture_lm[t] = P.posture_lm[t];
  }
  for (int t = 0; t < kMaxBoxTerms; ++t)
    for (int dd = 0; dd < kLaneDescDofs2; ++dd) { L.cfg_lower[t][dd] = -inf; L.cfg_upper[t][dd] = inf; L.vel_limit[t][dd] = inf; }
  for (int t = 0; t < P.n_cfg; ++t) {
    L.cfg_gain[t] = P.cfg_gain[t];
    for (int dd = 0; dd < m->nv; ++dd) { L.cfg_lower[t][dd] = clo[t * 64 + dd]; L.cfg_upper[t][dd] = chi[t * 64 + dd]; }
  }
  for (int t = 0; t < P.n_vel; ++t)
    for (int dd = 0; dd < m->nv; ++dd) L.vel_limit[t][dd] = vlim[t * 64 + dd];
  (void)d;
  if (has_free || com || has_relative || m->nv > 16 || nl > 16) return 32;                                     // (32: the row kernel on two DPP rows per problem)
  return m->nv <= 4 ? 4 : (m->nv <= 6 ? 6 : (m->nv == 7 ? 7 : (m->nv == 8 ? 8 : 16)));   // (16: the row kernel only)
}

// Resident wavefronts per CU of a kernel variant: bounded by LDS (160 KiB/CU) and by the register map the
// variant was built for (4 waves/SIMD for NT ≤ 8, 3 for NT ≤ 24, else 2 — ik_kernel.h MKH_WAVES).
static int waves_per_cu(int nt, int lds_bytes, bool w3 = false) {
  int by_lds = (160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1);
  // gfx950 hands LDS out in granules of 320 dwords: 128 of them per CU.  A persistent kernel must not be launched with more
  // workgroups than are resident at once — the ones that wait for a slot start when the first ones EXIT, i.e. after the whole
  // batch, and then walk their static share alone (round 5: `44_36_r44_w3` asked for 16 064 B, 10 per CU by division, 9 by
  // granules: 2.12 ms instead of 1.44 with a third of the CUs idle; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES gave it away).  Applied
  // where a layout lands between the two figures: the F_COM builds on the one-more-wave map.
  if (w3 && nt == 44 && lds_bytes > (160 * 1024) / 12) by_lds = 128 / ((lds_bytes + 1279) / 1280);
  // (w3: the high-occupancy build of the variant — one more resident wave per SIMD than its plain register map)
  const int by_regs = 4 * (nt <= 8 ? 4 : (nt <= 24 ? (w3 ? 4 : 3) : (w3 ? 3 : 2)));
  const int w = by_lds < by_regs ? by_lds : by_regs;
  return w < 1 ? 1 : w;
}

template <class T>
static hipError_t ensure(T** buf, size_t n) {
  if (*buf) return hipSuccess;
  return hipMalloc((void**)buf, (n ? n : 1) * sizeof(T));
}

// The debug / plugin taps of a call: `alloc(host pointer, bytes, zero first)` returns the device-visible buffer the kernel
// writes (nullptr for a tap the caller did not ask for).
template <typename Alloc>
static void assign_taps(const MkhTaps* taps, const DeviceProblem& P, size_t Bz, TapArgs& t, Alloc&& alloc) {
  const size_t nv = P.nv;
  t.t_xpos = (double*)alloc(taps->xpos, Bz * P.nbody * 3 * 8, false);
  t.t_xquat = (double*)alloc(taps->xquat, Bz * P.nbody * 4 * 8, false);
  t.t_frame_pose = (double*)alloc(taps->frame_pose, Bz * P.n_frame * 7 * 8, false);
  t.t_subtree_com = (double*)alloc(taps->subtree_com, Bz * 3 * 8, true);
  t.t_task_e = (double*)alloc(taps->task_e, Bz * P.n_rows_tap * 8, true);
  t.t_task_J = (double*)alloc(taps->task_J, Bz * P.n_rows_tap * nv * 8, true);
  t.t_H = (double*)alloc(taps->H, Bz * nv * nv * 8, false);
  t.t_c = (double*)alloc(taps->c, Bz * nv * 8, false);
  t.t_box_lo = (double*)alloc(taps->box_lo, Bz * nv * 8, false);
  t.t_box_hi = (double*)alloc(taps->box_hi, Bz * nv * 8, false);
  t.t_coll_G = (double*)alloc(taps->coll_G, Bz * P.n_pairs * nv * 8, true);
  t.t_coll_h = (double*)alloc(taps->coll_h, Bz * P.n_pairs * 8, false);
  t.t_qp_iters = (int32_t*)alloc(taps->qp_iters, Bz * 4, true);
  t.t_cycles = (long long*)alloc(taps->cycles, Bz * 16 * 8, true);
}


// The descriptor of the workgroup-per-problem kernel (wide_kernel.h): plain arrays over bodies / joints / dofs (no 64-wide
// lane tables), the kinematic tree by levels, the dof chain of every body as a bit set, LDS offsets, and the per-workgroup
// slice of device memory (weighted Jacobian rows, contact records, the tableau when it does not fit in LDS).
static int32_t build_wide_problem(MkhProblem* p, const MkhModel* m, const MkhProblemDesc* d, const std::vector<FrameTaskDev>& ft,
                                  const std::vector<CollisionPairDev>& pairs, const std::vector<PairCull>& culls, const std::vector<double>& dcost,
                                  const std::vector<double>& dwgain) {
  const double inf = std::numeric_limits<double>::infinity();
  WideProblem& W = p->wide;
  memset(&W, 0, sizeof W);
  const int nv = m->nv, nb = m->nbody, nj = m->njnt;
  W.nq = m->nq; W.nv = nv; W.nbody = nb; W.njnt = nj; W.robot_root = 1;
  W.n_frame = d->n_frame_tasks; W.n_posture = d->n_posture_tasks; W.n_com = d->n_com_tasks;
  W.n_cfg = d->n_configuration_limits; W.n_vel = d->n_velocity_limits; W.n_pairs = (int)pairs.size();
  W.n_dense_tasks = d->n_dense_tasks; W.n_dense_rows = (int)dcost.size(); W.n_dense_limit_rows = d->n_dense_limit_rows;
  W.dense_box = d->dense_limit_box ? 1 : 0;
  auto up = [&](const auto& v, auto** out) -> hipError_t {
    const hipError_t e = upload(v, out);
    if (e == hipSuccess) p->wide_allocs.push_back((void*)*out);
    return e;
  };
  // ---- the tree by levels, subtree ranges, dof chains
  std::vector<int32_t> depth(nb, 0), level_start, level_body, last(nb), inrobot(nb), jadr(nb);
  int maxd = 0;
  for (int b = 1; b < nb; ++b) { depth[b] = depth[m->body_parentid[b]] + 1; if (depth[b] > maxd) maxd = depth[b]; }
  for (int lv = 0; lv <= maxd; ++lv) {
    level_start.push_back((int32_t)level_body.size());
    for (int b = 0; b < nb; ++b) if (depth[b] == lv) level_body.push_back(b);
  }
  level_start.push_back((int32_t)level_body.size());
  W.nlevels = maxd + 1;
  for (int b = nb - 1; b >= 0; --b) {
    last[b] = b;
    for (int c = b + 1; c < nb; ++c) if (m->body_parentid[c] == b && last[c] > last[b]) last[b] = last[c];
  }
  for (int b = 0; b < nb; ++b) { inrobot[b] = (b >= 1 && m->body_rootid[b] == 1) ? 1 : 0; jadr[b] = m->body_jntadr[b] < 0 ? 0 : m->body_jntadr[b]; }
  W.chain_words = (nv + 63) / 64;
  std::vector<uint64_t> chain((size_t)nb * W.chain_words, 0ull);
  for (int b = 1; b < nb; ++b) {
    int x = b;
    while (x > 0 && m->body_dofnum[x] == 0) x = m->body_parentid[x];
    if (x == 0) continue;
    for (int i = m->body_dofadr[x] + m->body_dofnum[x] - 1; i >= 0; i = m->dof_parentid[i])
      chain[(size_t)b * W.chain_words + (i >> 6)] |= 1ull << (i & 63);
  }
  // ---- dofs
  std::vector<int32_t> dkind(nv), dk(nv, 0), dqadr(nv, -1);
  std::vector<double> dlo(nv, -inf), dhi(nv, inf);
  for (int dd = 0; dd < nv; ++dd) {
    const int j = m->dof_jntid[dd], jt = m->jnt_type[j], k = dd - m->jnt_dofadr[j];
    if (jt == JNT_HINGE) { dkind[dd] = DOF_HINGE; dqadr[dd] = m->jnt_qposadr[j]; }
    else if (jt == JNT_SLIDE) { dkind[dd] = DOF_SLIDE; dqadr[dd] = m->jnt_qposadr[j]; }
    else if (jt == JNT_BALL) { dkind[dd] = DOF_BALL; dk[dd] = k; dqadr[dd] = m->jnt_qposadr[j]; }
    else if (k < 3) { dkind[dd] = DOF_FREE_LIN; dk[dd] = k; }
    else { dkind[dd] = DOF_FREE_ANG; dk[dd] = k - 3; }
    if (((jt == JNT_HINGE || jt == JNT_SLIDE) || (jt == JNT_BALL && k == 0)) && m->jnt_limited[j]) {
      dlo[dd] = m->jnt_range[2 * j]; dhi[dd] = m->jnt_range[2 * j + 1];
    }
  }
  // ---- tasks and limits, one value per dof
  std::vector<double> pcost((size_t)(W.n_posture ? W.n_posture : 1) * nv, 0.0);
  for (int t = 0; t < W.n_posture; ++t) {
    for (int i = 0; i < nv; ++i) pcost[(size_t)t * nv + i] = d->posture_tasks[t].cost[i];
    W.posture_gain[t] = d->posture_tasks[t].gain; W.posture_lm[t] = d->posture_tasks[t].lm_damping;
  }
  int jrows = 0;
  for (const auto& f : ft) jrows += __builtin_popcount(f.rowmask);      // (FrameTaskDev::jrow0 counts the same way)
  for (int t = 0; t < W.n_com; ++t) {
    W.com_rowmask[t] = 0;
    for (int k = 0; k < 3; ++k) { W.com_cost[t][k] = d->com_tasks[t].cost[k]; if (d->com_tasks[t].cost[k] != 0.0) W.com_rowmask[t] |= 1 << k; }
    W.com_gain[t] = d->com_tasks[t].gain; W.com_lm[t] = d->com_tasks[t].lm_damping;
    W.com_jrow0[t] = jrows; jrows += __builtin_popcount(W.com_rowmask[t]);
  }
  W.n_jrows = jrows;
  // rows of the (e, J) tap layout — the same order mkh_problem_create counts for the wavefront kernels (DeviceProblem::n_rows_tap)
  {
    int row = 6 * W.n_frame;
    for (int t = 0; t < W.n_posture; ++t) { W.posture_row0[t] = row; row += nv; }
    for (int t = 0; t < W.n_com; ++t) { W.com_row0[t] = row; row += 3; }
    W.dense_tap_row0 = row;
    for (int t = 0; t < W.n_dense_tasks; ++t) row += d->dense_tasks[t].k;
    W.n_rows_tap = row;
  }
  for (int t = 0, r0 = 0; t < W.n_dense_tasks; ++t) {
    W.dense_row0[t] = r0; W.dense_k[t] = d->dense_tasks[t].k; W.dense_lm[t] = d->dense_tasks[t].lm_damping; r0 += d->dense_tasks[t].k;
  }
  std::vector<double> clo((size_t)(W.n_cfg ? W.n_cfg : 1) * nv, -inf), chi((size_t)(W.n_cfg ? W.n_cfg : 1) * nv, inf);
  for (int t = 0; t < W.n_cfg; ++t) {
    const MkhConfigurationLimitDesc& c = d->configuration_limits[t];
    W.cfg_gain[t] = c.gain;
    for (int k = 0; k < c.n_indices; ++k) {
      const int dof = c.indices[k], qa = m->jnt_qposadr[m->dof_jntid[dof]];     // (validated by the caller)
      clo[(size_t)t * nv + dof] = c.lower[qa]; chi[(size_t)t * nv + dof] = c.upper[qa];
    }
  }
  std::vector<double> vlim((size_t)(W.n_vel ? W.n_vel : 1) * nv, inf);
  for (int t = 0; t < W.n_vel; ++t) {
    const MkhVelocityLimitDesc& v = d->velocity_limits[t];
    for (int k = 0; k < v.n_indices; ++k) vlim[(size_t)t * nv + v.indices[k]] = std::fmin(vlim[(size_t)t * nv + v.indices[k]], v.limit[k]);
  }
  hipError_t e = hipSuccess;
#define MKH_UP(vec, field) if (e == hipSuccess) { std::remove_const_t<std::remove_pointer_t<decltype(W.field)>>* dp = nullptr; e = up(vec, &dp); W.field = dp; }
  MKH_UP(level_start, level_start) MKH_UP(level_body, level_body) MKH_UP(m->body_parentid, body_parent) MKH_UP(jadr, body_jntadr)
  MKH_UP(m->body_jntnum, body_jntnum) MKH_UP(last, body_last) MKH_UP(inrobot, body_inrobot)
  MKH_UP(m->body_pos, body_pos) MKH_UP(m->body_quat, body_quat) MKH_UP(m->body_ipos, body_ipos) MKH_UP(m->body_mass, body_mass)
  MKH_UP(m->body_subtreemass, body_stmass) MKH_UP(m->jnt_type, jnt_type) MKH_UP(m->jnt_qposadr, jnt_qadr) MKH_UP(m->jnt_dofadr, jnt_dadr) MKH_UP(m->jnt_axis, jnt_axis)
  MKH_UP(m->jnt_pos, jnt_pos) MKH_UP(m->jnt_qpos0, jnt_qpos0) MKH_UP(m->dof_jntid, dof_jnt) MKH_UP(dkind, dof_kind) MKH_UP(dk, dof_k)
  MKH_UP(m->dof_bodyid, dof_body) MKH_UP(dqadr, dof_qadr) MKH_UP(dlo, dof_lo) MKH_UP(dhi, dof_hi) MKH_UP(chain, chain)
  MKH_UP(ft, frame) MKH_UP(pcost, posture_cost) MKH_UP(dcost, dense_cost) MKH_UP(dwgain, dense_wgain) MKH_UP(clo, cfg_lower)
  MKH_UP(chi, cfg_upper) MKH_UP(vlim, vel_limit) MKH_UP(pairs, pairs)
  // ---- LDS layout and the per-workgroup slice of device memory
  const int rows_max = W.n_pairs + W.n_dense_limit_rows;
  W.max_rows = rows_max < kWideMaxRows ? rows_max : kWideMaxRows;
  const int Ncap = nv + W.max_rows, R_all = W.n_jrows + W.n_dense_rows;
  auto ev = [](int x) { return (x + 1) & ~1; };
  // more than a wavefront of pairs: bounding-sphere cull in front of the distance routines (wide_contacts; the candidate list —
  // 16-bit pair indices — lives in the three QP vectors behind o_rown, which are dead until the tableau is built)
  W.cull = nullptr;
  if (W.n_pairs > kWave && W.n_pairs < 65536 && W.n_pairs <= 12 * ev(Ncap) && culls.size() == pairs.size() && !(p->diag & MKH_DIAG_NO_PAIR_CULL)) { MKH_UP(culls, cull) }
#undef MKH_UP
  if (e != hipSuccess) return fail(MKH_E_HIP, "wide problem upload: %s", hipGetErrorString(e));
  int o = 0;
  W.o_q = o; o += ev(W.nq);
  W.o_X = o; o += ev(7 * nb);
  W.o_jnt = o; o += ev(6 * (nj > 0 ? nj : 1));
  W.o_dof = o; o += 10 * nv;
  W.o_task = o; o += 64 * (W.n_frame > 0 ? W.n_frame : 1);
  W.o_com = o; o += W.n_com > 0 ? 4 * nb : 0;
  W.o_we = o; o += ev(R_all > 0 ? R_all : 1);
  W.o_c = o; o += ev(nv);
  W.o_hd = o; o += ev(nv);
  W.o_z = o; o += ev(Ncap); W.o_w = o; o += ev(Ncap); W.o_lo = o; o += ev(nv); W.o_hi = o; o += ev(nv);
  W.o_rown = o; o += ev(Ncap); W.o_ref = o; o += ev(Ncap); W.o_col = o; o += ev(Ncap);
  W.o_red = o; o += kWideThreads + kWideThreads / 2;
  W.o_state = o; o += ev((Ncap + 1) / 2);
  W.o_cws = o; o += p->convex_pairs ? 4 * (kEpaWsDoubles > kGjkWsDoubles ? kEpaWsDoubles : kGjkWsDoubles) : 0;
  const int lds_cap = (160 * 1024 - 2048) / 8;                              // doubles of LDS one workgroup may take
  // staging vectors of the block pivots (wide_kernel.h wide_rank4): over the poses / joint axes when they fit, else appended — unless
  // that would push the tableau out of LDS or cost the second resident workgroup
  W.blk_stride = ev(Ncap);
  W.o_blk = -1;
  if (8 * W.blk_stride <= W.o_dof - W.o_X) W.o_blk = W.o_X;
  else {
    const long long with = (long long)o + 8 * W.blk_stride, t = (long long)Ncap * Ncap;
    const bool keeps_t = (with + t <= lds_cap) == ((long long)o + t <= lds_cap);
    const long long tot0 = (long long)o + ((long long)o + t <= lds_cap ? t : 0), tot1 = with + (with + t <= lds_cap ? t : 0);
    const bool keeps_two = (tot1 * 8 <= 80 * 1024) == (tot0 * 8 <= 80 * 1024);
    if (with <= lds_cap && keeps_t && keeps_two) { W.o_blk = o; o += 8 * W.blk_stride; }
  }
  W.tableau_in_lds = (long long)o + (long long)Ncap * Ncap <= lds_cap ? 1 : 0;
  // the dense Goldfarb–Idnani fallback's factors (2·nv² + 2·nv + 8 doubles) in LDS when that costs neither the tableau's place nor
  // the second resident workgroup
  const long long gi_sz = rows_max > 0 ? 2ll * nv * (nv | 1) + 4ll * (nv + 2) : 0;
  const int two_wg = 80 * 1024 / 8;
  W.o_gi = o; W.gi_in_lds = 0;
  if (gi_sz > 0 && (long long)o + gi_sz + (long long)(nv + 8) * (nv + 8) <= two_wg) { W.gi_in_lds = 1; o += (int)gi_sz; }
  W.o_T = o;
  if (W.tableau_in_lds) { W.t_lds_doubles = Ncap * Ncap; o += Ncap * Ncap; }
  else {
    // the largest instance does not fit: reserve what keeps two workgroups resident (or, failing that, what is left) for the
    // instances that do — most have far fewer rows in range than the limit lists pairs
    long long room = (long long)two_wg - o;
    if (room < (long long)(nv + 8) * (nv + 8)) room = (long long)lds_cap - o;
    W.t_lds_doubles = room > 0 ? (int)room : 0;
    o += W.t_lds_doubles;
  }
  W.lds_doubles = o;
  if (o > lds_cap)
    return fail(MKH_E_LIMIT, "model too large for the workgroup-per-problem kernel: %d KB of LDS per problem, 158 KB available (per problem "
                "≈ 8·(nq + 7·nbody + 6·njnt + 14·nv + 5.5·(nv + rows) + 64·frame tasks) bytes: a serial chain fits up to ≈ 550 dofs)", o / 128);
  long long w = 0;
  W.ws_jw = w; w += (long long)(R_all > 0 ? R_all : 1) * nv;
  W.ws_rec = w; w += (long long)(W.n_pairs > 0 ? W.n_pairs : 1) * 10;
  W.ws_rowpair = w; w += ev((W.max_rows + 2) / 2);
  W.ws_rank = w; w += ev((W.n_pairs + 2) / 2);
  W.ws_T = w; w += W.tableau_in_lds ? 0 : (long long)Ncap * Ncap;
  W.ws_gi = w; w += rows_max > 0 ? 2ll * nv * (nv | 1) + 4ll * (nv + 2) : 0;
  W.ws_stride = (w + 15) & ~15ll;
  const int per_cu = (160 * 1024) / (o * 8) < 2 ? ((160 * 1024) / (o * 8) < 1 ? 1 : (160 * 1024) / (o * 8)) : 2;
  int grid = m->num_cus * per_cu;
  if (grid > p->max_batch) grid = p->max_batch;
  while (grid > 1 && (long long)grid * W.ws_stride * 8 > (1ll << 30)) grid /= 2;      // (≤ 1 GB of slices)
  double* ws = nullptr;
  if (hipMalloc((void**)&ws, (size_t)grid * W.ws_stride * 8) != hipSuccess) return fail(MKH_E_HIP, "wide workspace (%lld MB)", (long long)grid * W.ws_stride * 8 >> 20);
  p->wide_allocs.push_back(ws);
  W.ws = ws;
  p->wide_grid = grid; p->wide_lds = o * 8;

  {
    uint32_t cap = 256;
    while (cap < (uint32_t)p->max_batch) cap *= 2;
    int32_t* rq = nullptr;
    if (hipMalloc((void**)&rq, ((size_t)cap + 4) * sizeof(int32_t)) != hipSuccess) return fail(MKH_E_HIP, "redo queue");
    p->wide_allocs.push_back(rq);
    if (hipMemset(rq, 0xff, (size_t)cap * sizeof(int32_t)) != hipSuccess || hipMemset(rq + cap, 0, 4 * sizeof(int32_t)) != hipSuccess)
      return fail(MKH_E_HIP, "redo queue");
    W.redo_queue = rq; W.redo_ctr = reinterpret_cast<uint32_t*>(rq + cap); W.redo_cap = cap; W.redo_pad = 0;
  }
  if (hipMalloc((void**)&p->d_wide, sizeof(WideProblem)) != hipSuccess ||
      hipMemcpy(p->d_wide, &W, sizeof(WideProblem), hipMemcpyHostToDevice) != hipSuccess)
    return fail(MKH_E_HIP, "wide descriptor upload failed");
  if (!p->d_status_tight && hipMalloc((void**)&p->d_status_tight, (size_t)p->max_batch * sizeof(int32_t)) != hipSuccess)
    return fail(MKH_E_HIP, "status buffer");
  return MKH_OK;
}

// one launch of the workgroup-per-problem kernel: every instance (redo_mask = 0) or the ones a wavefront kernel flagged
static int32_t launch_wide_kernel(MkhProblem* p, const SolveArgs& a, hipStream_t stream, int32_t redo_mask, const TapArgs* dtaps = nullptr) {
  SolveArgs aw = a;
  aw.redo_mask = redo_mask;
  aw.work_counter = p->d_work;        // (redo_mask = 0: the kernel as THE path of a model draws its problems from it)
  int grid = p->wide_grid < a.B ? p->wide_grid : a.B;
  if (grid < 1) grid = 1;
  const int rc = mkh::launch_wide(grid, p->wide_lds, stream, p->d_wide, aw, dtaps, p->convex_pairs);
  if (rc != 0) return fail(MKH_E_HIP, "wide kernel: %s", hipGetErrorString((hipError_t)rc));
  HIP_OK(hipGetLastError());
  return MKH_OK;
}

extern "C" {

int32_t mkh_version(void) { return MKH_VERSION; }
const char* mkh_last_error(void) { return g_err.c_str(); }

int32_t mkh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static uint64_t dof_chain_mask(const MkhModel* m, int body) {
  uint64_t mask = 0;
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body == 0) return 0;
  int i = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  while (i >= 0) {
    mask |= 1ull << i;
    i = m->dof_parentid[i];
  }
  return mask;
}

int32_t mkh_model_create(const MkhFlatModel* h, int32_t device, MkhModel** out) {
  if (!h || !out) return fail(MKH_E_INVALID, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MKH_E_NOGPU, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MKH_E_INVALID, "device %d out of range (%d visible)", device, ndev);
  // (more than 64 bodies or dofs: beyond the wavefront kernels — the workgroup-per-problem kernel takes every call, wide_kernel.h)
  const bool big = h->nbody > kWave || h->nv > kWave;
  if (h->nbody > 4096 || h->nv > 1024) return fail(MKH_E_LIMIT, "nbody=%d / nv=%d: at most 4096 bodies and 1024 dofs", h->nbody, h->nv);
  if (h->nv < 1 || h->nbody < 2) return fail(MKH_E_INVALID, "model has no degrees of freedom");
  HIP_OK(hipSetDevice(device));
  MkhModel* m = new MkhModel();
  m->device = device;
  m->big = big;
  m->nq = h->nq; m->nv = h->nv; m->nbody = h->nbody; m->njnt = h->njnt; m->ngeom = h->ngeom; m->nsite = h->nsite;
  m->body_ipos.assign(h->body_ipos, h->body_ipos + 3 * h->nbody);
  m->body_mass.assign(h->body_mass, h->body_mass + h->nbody);
  m->body_subtreemass.assign(h->body_subtreemass, h->body_subtreemass + h->nbody);
  auto cpi = [](const int32_t* p, int n) { return std::vector<int32_t>(p, p + n); };
  auto cpd = [](const double* p, int n) { return std::vector<double>(p, p + n); };
  m->body_parentid = cpi(h->body_parentid, h->nbody); m->body_rootid = cpi(h->body_rootid, h->nbody);
  m->body_jntnum = cpi(h->body_jntnum, h->nbody); m->body_jntadr = cpi(h->body_jntadr, h->nbody);
  m->body_dofnum = cpi(h->body_dofnum, h->nbody); m->body_dofadr = cpi(h->body_dofadr, h->nbody);
  m->jnt_type = cpi(h->jnt_type, h->njnt); m->jnt_qposadr = cpi(h->jnt_qposadr, h->njnt);
  m->jnt_dofadr = cpi(h->jnt_dofadr, h->njnt); m->jnt_bodyid = cpi(h->jnt_bodyid, h->njnt);
  m->jnt_limited = cpi(h->jnt_limited, h->njnt); m->jnt_range = cpd(h->jnt_range, h->njnt * 2);
  m->jnt_pos = cpd(h->jnt_pos, h->njnt * 3); m->jnt_axis = cpd(h->jnt_axis, h->njnt * 3);
  for (int j = 0; j < h->njnt; ++j) m->jnt_qpos0.push_back(h->qpos0[h->jnt_qposadr[j]]);
  m->dof_bodyid = cpi(h->dof_bodyid, h->nv); m->dof_jntid = cpi(h->dof_jntid, h->nv);
  m->dof_parentid = cpi(h->dof_parentid, h->nv);
  m->site_bodyid = cpi(h->site_bodyid, h->nsite); m->geom_bodyid = cpi(h->geom_bodyid, h->ngeom);
  m->geom_type = cpi(h->geom_type, h->ngeom);
  m->body_pos = cpd(h->body_pos, h->nbody * 3); m->body_quat = cpd(h->body_quat, h->nbody * 4);
  m->site_pos = cpd(h->site_pos, h->nsite * 3); m->site_quat = cpd(h->site_quat, h->nsite * 4);
  m->geom_size = cpd(h->geom_size, h->ngeom * 3); m->geom_pos = cpd(h->geom_pos, h->ngeom * 3);
  m->geom_quat = cpd(h->geom_quat, h->ngeom * 4);
  m->geom_dataid.assign(h->ngeom, -1);
  if (h->nmesh > 0) {
    if (!h->geom_dataid || !h->mesh_vertadr || !h->mesh_vertnum || !h->mesh_vert || h->nmeshvert < 1) {
      delete m; return fail(MKH_E_INVALID, "nmesh = %d but a mesh array is null", h->nmesh);
    }
    m->geom_dataid = cpi(h->geom_dataid, h->ngeom);
    m->mesh_vertadr = cpi(h->mesh_vertadr, h->nmesh); m->mesh_vertnum = cpi(h->mesh_vertnum, h->nmesh);
    for (int k = 0; k < h->nmesh; ++k)
      if (m->mesh_vertadr[k] < 0 || m->mesh_vertnum[k] < 1 || m->mesh_vertadr[k] + m->mesh_vertnum[k] > h->nmeshvert) {
        delete m; return fail(MKH_E_INVALID, "mesh %d: vertex range out of bounds", k);
      }
    for (int g = 0; g < h->ngeom; ++g)
      if (m->geom_dataid[g] >= h->nmesh) { delete m; return fail(MKH_E_INVALID, "geom %d: mesh id out of range", g); }
  }

  const double inf = std::numeric_limits<double>::infinity();
  // ---- body tables
  std::vector<int> depth(h->nbody, 0);
  int maxdepth = 0;
  for (int b = 1; b < h->nbody; ++b) {
    if (h->body_parentid[b] >= b) { delete m; return fail(MKH_E_INVALID, "body %d has parent id >= its own id", b); }
    depth[b] = depth[h->body_parentid[b]] + 1;
    if (depth[b] > maxdepth) maxdepth = depth[b];
  }
  int nrounds = 0;
  while ((1 << nrounds) < maxdepth) ++nrounds;
  if (!big && nrounds > kMaxRounds) { delete m; return fail(MKH_E_LIMIT, "kinematic tree too deep"); }
  m->nrounds = nrounds;
  std::vector<double> body_f(BF_COUNT * 64, 0.0);
  std::vector<int32_t> body_i(BI_COUNT * 64, 0);
  if (!big) {                                      // (the lane tables of the wavefront kernels: ≤ 64 bodies / dofs)
  std::vector<int> subtree_last(h->nbody);
  for (int b = h->nbody - 1; b >= 0; --b) {
    subtree_last[b] = b;
    for (int c = b + 1; c < h->nbody; ++c)
      if (h->body_parentid[c] == b && subtree_last[c] > subtree_last[b]) subtree_last[b] = subtree_last[c];
  }
  for (int b = 0; b < h->nbody; ++b) {
    for (int k = 0; k < 3; ++k) body_f[(BF_POS + k) * 64 + b] = h->body_pos[3 * b + k];
    for (int k = 0; k < 4; ++k) body_f[(BF_QUAT + k) * 64 + b] = h->body_quat[4 * b + k];
    for (int k = 0; k < 3; ++k) body_f[(BF_IPOS + k) * 64 + b] = h->body_ipos[3 * b + k];
    body_f[BF_MASS * 64 + b] = h->body_mass[b];
    body_f[BF_SUBTREEMASS * 64 + b] = h->body_subtreemass[b];
    body_i[BI_PARENT * 64 + b] = h->body_parentid[b];
    body_i[BI_JNTADR * 64 + b] = h->body_jntadr[b] < 0 ? 0 : h->body_jntadr[b];
    body_i[BI_JNTNUM * 64 + b] = h->body_jntnum[b];
    body_i[BI_SUBTREE_LAST * 64 + b] = subtree_last[b];
    body_i[BI_IN_ROBOT * 64 + b] = (b >= 1 && h->body_rootid[b] == 1) ? 1 : 0;  // subtree of body 1 (ComTask)
    // pointer-jumping ancestors: anc_0 = parent, anc_{r+1} = anc_r(anc_r)
    int a = h->body_parentid[b];
    for (int r = 0; r < kMaxRounds; ++r) {
      body_i[(BI_ANC0 + r) * 64 + b] = a;
      // advance 2^r more steps
      int steps = 1 << r, x = a;
      for (int k = 0; k < steps && x > 0; ++k) x = h->body_parentid[x];
      a = x;
    }
  }
  for (int b = 0; b < h->nbody; ++b) {
    int lo = 0;
    for (int r = 0; r < 5 && r < kMaxRounds; ++r) lo |= (body_i[(BI_ANC0 + r) * 64 + b] & 63) << (6 * r);
    body_i[BI_ANCPACK0 * 64 + b] = lo;
    body_i[BI_ANCPACK1 * 64 + b] = kMaxRounds > 5 ? (body_i[(BI_ANC0 + 5) * 64 + b] & 63) : 0;
  }
  // the anc recurrence above must satisfy anc_{r+1}(b) = anc_r(anc_r(b)); verify
  for (int b = 0; b < h->nbody; ++b)
    for (int r = 0; r + 1 < kMaxRounds; ++r) {
      int a = body_i[(BI_ANC0 + r) * 64 + b];
      int aa = body_i[(BI_ANC0 + r) * 64 + a];
      if (body_i[(BI_ANC0 + r + 1) * 64 + b] != aa) { delete m; return fail(MKH_E_INVALID, "internal: ancestor table"); }
    }
  }
  // ---- joint arrays
  std::vector<double> jnt_f(h->njnt * JF_COUNT, 0.0);
  std::vector<int32_t> jnt_i(h->njnt * JI_COUNT, 0);
  for (int j = 0; j < h->njnt; ++j) {
    for (int k = 0; k < 3; ++k) jnt_f[j * JF_COUNT + JF_AXIS + k] = h->jnt_axis[3 * j + k];
    for (int k = 0; k < 3; ++k) jnt_f[j * JF_COUNT + JF_POS + k] = h->jnt_pos[3 * j + k];
    jnt_f[j * JF_COUNT + JF_QPOS0] = h->qpos0[h->jnt_qposadr[j]];
    jnt_i[j * JI_COUNT + JI_TYPE] = h->jnt_type[j];
    jnt_i[j * JI_COUNT + JI_QADR] = h->jnt_qposadr[j];
    jnt_i[j * JI_COUNT + JI_DADR] = h->jnt_dofadr[j];
    if (h->jnt_type[j] == JNT_FREE && (h->body_jntnum[h->jnt_bodyid[j]] != 1)) {
      delete m; return fail(MKH_E_INVALID, "free joint must be the only joint of its body");
    }
  }
  // ---- dof tables
  std::vector<int32_t> dof_i(DI_COUNT * 64, 0);
  std::vector<double> dof_f(DF_COUNT * 64, 0.0);
  for (int d = 0; d < 64; ++d) { dof_f[DF_RANGE_LO * 64 + d] = -inf; dof_f[DF_RANGE_HI * 64 + d] = inf; dof_i[DI_QADR * 64 + d] = -1; }
  for (int d = 0; d < (big ? 0 : h->nv); ++d) {
    const int j = h->dof_jntid[d];
    const int jt = h->jnt_type[j];
    const int k = d - h->jnt_dofadr[j];
    int kind, kk = 0, qadr = -1;
    if (jt == JNT_HINGE) { kind = DOF_HINGE; qadr = h->jnt_qposadr[j]; }
    else if (jt == JNT_SLIDE) { kind = DOF_SLIDE; qadr = h->jnt_qposadr[j]; }
    else if (jt == JNT_BALL) { kind = DOF_BALL; kk = k; qadr = h->jnt_qposadr[j]; }
    else { if (k < 3) { kind = DOF_FREE_LIN; kk = k; } else { kind = DOF_FREE_ANG; kk = k - 3; } }
    dof_i[DI_JNT * 64 + d] = j;
    dof_i[DI_KIND * 64 + d] = kind;
    dof_i[DI_K * 64 + d] = kk;
    dof_i[DI_BODY * 64 + d] = h->dof_bodyid[d];
    dof_i[DI_QADR * 64 + d] = qadr;
    dof_i[DI_JIDX * 64 + d] = j - h->body_jntadr[h->jnt_bodyid[j]];
    // (a limited ball joint: the reference's check_limits compares the quaternion's w with the range — first dof lane)
    if (((jt == JNT_HINGE || jt == JNT_SLIDE) || (jt == JNT_BALL && k == 0)) && h->jnt_limited[j]) {
      dof_f[DF_RANGE_LO * 64 + d] = h->jnt_range[2 * j];
      dof_f[DF_RANGE_HI * 64 + d] = h->jnt_range[2 * j + 1];
    }
  }
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = upload(body_f, &m->d_body_f);
  if (e == hipSuccess) e = upload(body_i, &m->d_body_i);
  if (e == hipSuccess) e = upload(jnt_f, &m->d_jnt_f);
  if (e == hipSuccess) e = upload(jnt_i, &m->d_jnt_i);
  if (e == hipSuccess) e = upload(dof_i, &m->d_dof_i);
  if (e == hipSuccess) e = upload(dof_f, &m->d_dof_f);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
  if (h->nmesh > 0) m->h_mesh_vert.assign(h->mesh_vert, h->mesh_vert + (size_t)h->nmeshvert * 3);
  if (e == hipSuccess && h->nmesh > 0) e = upload(m->h_mesh_vert, &m->d_mesh_vert);
  if (e != hipSuccess) { mkh_model_destroy(m); return fail(MKH_E_HIP, "model upload: %s", hipGetErrorString(e)); }
  m->num_cus = prop.multiProcessorCount;
  *out = m;
  return MKH_OK;
}

void mkh_model_destroy(MkhModel* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipFree(m->d_body_f); (void)hipFree(m->d_body_i); (void)hipFree(m->d_jnt_f); (void)hipFree(m->d_jnt_i);
  (void)hipFree(m->d_dof_i); (void)hipFree(m->d_dof_f); (void)hipFree(m->d_mesh_vert);
  m->small.release();
  delete m;
}

static void fill_base(const MkhModel* m, DeviceProblem& P) {
  P.nq = m->nq; P.nv = m->nv; P.nbody = m->nbody; P.njnt = m->njnt; P.nrounds = m->nrounds;
  P.robot_root = 1;
  P.body_f = m->d_body_f; P.body_i = m->d_body_i; P.jnt_f = m->d_jnt_f; P.jnt_i = m->d_jnt_i;
  P.dof_i = m->d_dof_i; P.dof_f = m->d_dof_f;
}

int32_t mkh_problem_create(MkhModel* m, const MkhProblemDesc* d, int32_t max_batch, MkhProblem** out) {
  return mkh_problem_create_diag(m, d, max_batch, 0, out);
}

int32_t mkh_problem_create_diag(MkhModel* m, const MkhProblemDesc* d, int32_t max_batch, int32_t diag, MkhProblem** out) {
  if (!m || !d || !out) return fail(MKH_E_INVALID, "null argument");
  *out = nullptr;
  if (diag & ~(MKH_DIAG_NO_WIDE_REDO | MKH_DIAG_NO_TIGHT_REDO | MKH_DIAG_NO_COLD_REFINE | MKH_DIAG_NO_PAIR_CULL))
    return fail(MKH_E_INVALID, "unknown MKH_DIAG_* bits 0x%x", diag);
  if (max_batch < 1) return fail(MKH_E_INVALID, "max_batch must be >= 1");
  if (d->n_frame_tasks > kMaxFrameTasks) return fail(MKH_E_LIMIT, "at most %d frame tasks", kMaxFrameTasks);
  if (d->n_posture_tasks > kMaxPostureTasks) return fail(MKH_E_LIMIT, "at most %d posture tasks", kMaxPostureTasks);
  if (d->n_com_tasks > kMaxComTasks) return fail(MKH_E_LIMIT, "at most %d CoM tasks", kMaxComTasks);
  if (d->n_dense_tasks < 0 || d->n_dense_tasks > kMaxDenseTasks) return fail(MKH_E_LIMIT, "at most %d dense (plugin) tasks", kMaxDenseTasks);
  if (d->n_dense_limit_rows < 0) return fail(MKH_E_INVALID, "n_dense_limit_rows must be >= 0");
  if (d->n_configuration_limits > kMaxBoxTerms || d->n_velocity_limits > kMaxBoxTerms)
    return fail(MKH_E_LIMIT, "at most %d configuration and %d velocity limits", kMaxBoxTerms, kMaxBoxTerms);
  HIP_OK(hipSetDevice(m->device));
  const double inf = std::numeric_limits<double>::infinity();
  MkhProblem* p = new MkhProblem();
  p->model = m;
  p->device = m->device;
  p->max_batch = max_batch;
  p->diag = diag;
  DeviceProblem& P = p->dev;
  fill_base(m, P);
  P.n_frame = d->n_frame_tasks; P.n_posture = d->n_posture_tasks; P.n_com = d->n_com_tasks;
  P.n_cfg = d->n_configuration_limits; P.n_vel = d->n_velocity_limits;
  auto bail = [&](int32_t code) { mkh_problem_destroy(p); return code; };

  // ---- frame tasks: resolve the frame to (body, local pose); rows of the tap layout
  int row = 0, jrows = 0;
  std::vector<FrameTaskDev> ft(d->n_frame_tasks);
  for (int t = 0; t < d->n_frame_tasks; ++t) {
    const MkhFrameTaskDesc& s = d->frame_tasks[t];
    FrameTaskDev& f = ft[t];
    memset(&f, 0, sizeof f);
    static const double zero3[3] = {0, 0, 0}, ident4[4] = {1, 0, 0, 0};
    auto resolve = [&](int ftype, int fid, int32_t& body, const double*& lp, const double*& lq) -> int32_t {
      if (ftype == MKH_FRAME_BODY) {
        if (fid < 0 || fid >= m->nbody) return fail(MKH_E_INVALID, "frame task %d: body id %d out of range", t, fid);
        body = fid; lp = zero3; lq = ident4;
      } else if (ftype == MKH_FRAME_SITE) {
        if (fid < 0 || fid >= m->nsite) return fail(MKH_E_INVALID, "frame task %d: site id %d out of range", t, fid);
        body = m->site_bodyid[fid]; lp = &m->site_pos[3 * fid]; lq = &m->site_quat[4 * fid];
      } else if (ftype == MKH_FRAME_GEOM) {
        if (fid < 0 || fid >= m->ngeom) return fail(MKH_E_INVALID, "frame task %d: geom id %d out of range", t, fid);
        body = m->geom_bodyid[fid]; lp = &m->geom_pos[3 * fid]; lq = &m->geom_quat[4 * fid];
      } else {
        return fail(MKH_E_INVALID, "frame task %d: unsupported frame type %d", t, ftype);
      }
      return MKH_OK;
    };
    const double *lp = nullptr, *lq = nullptr;
    if (resolve(s.frame_type, s.frame_id, f.body, lp, lq) != MKH_OK) return bail(MKH_E_INVALID);
    for (int k = 0; k < 3; ++k) f.lpos[k] = lp[k];
    for (int k = 0; k < 4; ++k) f.lquat[k] = lq[k];
    if (s.root_type >= 0) {
      f.relative = 1;
      p->has_relative = true;
      if (resolve(s.root_type, s.root_id, f.root_body, lp, lq) != MKH_OK) return bail(MKH_E_INVALID);
      for (int k = 0; k < 3; ++k) f.root_lpos[k] = lp[k];
      for (int k = 0; k < 4; ++k) f.root_lquat[k] = lq[k];
      f.root_mask = m->big ? 0ull : dof_chain_mask(m, f.root_body);
    }
    for (int k = 0; k < 6; ++k) {
      if (!(s.cost[k] >= 0.0)) return bail(fail(MKH_E_INVALID, "frame task %d: cost must be >= 0", t));
      f.cost[k] = s.cost[k];
    }
    f.gain = s.gain; f.lm_damping = s.lm_damping;
    f.dof_mask = m->big ? 0ull : dof_chain_mask(m, f.body);
    f.row0 = row; row += 6;
    f.any_ori = (s.cost[3] != 0.0 || s.cost[4] != 0.0 || s.cost[5] != 0.0) ? 1 : 0;
    f.rowmask = 0;
    for (int k = 0; k < 6; ++k) if (s.cost[k] != 0.0) f.rowmask |= 1 << k;
    f.jrow0 = jrows;
    jrows += __builtin_popcount(f.rowmask);
  }
  std::vector<double> pcost((size_t)(d->n_posture_tasks ? d->n_posture_tasks : 1) * 64, 0.0);
  for (int t = 0; t < d->n_posture_tasks; ++t) {
    for (int i = 0; i < (m->big ? 0 : m->nv); ++i) pcost[t * 64 + i] = d->posture_tasks[t].cost[i];
    P.posture_gain[t] = d->posture_tasks[t].gain; P.posture_lm[t] = d->posture_tasks[t].lm_damping;
    P.posture_row0[t] = row; row += m->nv;
  }
  for (int t = 0; t < d->n_com_tasks; ++t) {
    for (int k = 0; k < 3; ++k) P.com_cost[t][k] = d->com_tasks[t].cost[k];
    P.com_gain[t] = d->com_tasks[t].gain; P.com_lm[t] = d->com_tasks[t].lm_damping;
    P.com_row0[t] = row; row += 3;
    P.com_rowmask[t] = 0;
    for (int k = 0; k < 3; ++k) if (d->com_tasks[t].cost[k] != 0.0) P.com_rowmask[t] |= 1 << k;
    P.com_jrow0[t] = jrows;
    jrows += __builtin_popcount(P.com_rowmask[t]);
  }
  // ---- plugin route: caller-defined tasks as dense rows (tap rows follow the built-in ones)
  std::vector<double> dcost, dwgain;
  P.n_dense_tasks = d->n_dense_tasks;
  P.dense_tap_row0 = row;
  for (int t = 0; t < d->n_dense_tasks; ++t) {
    const MkhDenseTaskDesc& s = d->dense_tasks[t];
    if (s.k < 1 || !s.cost) return bail(fail(MKH_E_INVALID, "dense task %d: k must be >= 1 and cost non-null", t));
    if (!(s.gain >= 0.0 && s.gain <= 1.0) || !(s.lm_damping >= 0.0))
      return bail(fail(MKH_E_INVALID, "dense task %d: gain must be in [0, 1] and lm_damping >= 0", t));
    P.dense_row0[t] = (int)dcost.size(); P.dense_k[t] = s.k; P.dense_lm[t] = s.lm_damping;
    for (int r = 0; r < s.k; ++r) {
      if (!(s.cost[r] >= 0.0)) return bail(fail(MKH_E_INVALID, "dense task %d: cost must be >= 0", t));
      dcost.push_back(s.cost[r]);
      dwgain.push_back(s.cost[r] * -s.gain);
    }
    row += s.k;
  }
  P.n_dense_rows = (int)dcost.size();
  P.n_dense_limit_rows = d->n_dense_limit_rows;
  P.dense_box = d->dense_limit_box ? 1 : 0;
  P.n_rows_tap = row;
  P.n_jrows = jrows;

  // ---- box limits: per-dof lower/upper in joint coordinates (±inf = absent)
  std::vector<double> clo((size_t)(P.n_cfg ? P.n_cfg : 1) * 64, -inf), chi((size_t)(P.n_cfg ? P.n_cfg : 1) * 64, inf);
  for (int t = 0; t < P.n_cfg; ++t) {
    const MkhConfigurationLimitDesc& c = d->configuration_limits[t];
    if (!(c.gain > 0.0 && c.gain <= 1.0)) return bail(fail(MKH_E_INVALID, "configuration limit gain must be in (0, 1]"));
    P.cfg_gain[t] = c.gain;
    for (int k = 0; k < c.n_indices; ++k) {
      const int dof = c.indices[k];
      if (dof < 0 || dof >= m->nv) return bail(fail(MKH_E_INVALID, "configuration limit: dof %d out of range", dof));
      const int jt = m->jnt_type[m->dof_jntid[dof]];
      if (jt == JNT_FREE) return bail(fail(MKH_E_INVALID, "configuration limit on free-joint dof %d (the reference skips free joints)", dof));
      const int qa = m->jnt_qposadr[m->dof_jntid[dof]];
      if (m->big) continue;                                  // (beyond 64 dofs: build_wide_problem keeps its own arrays)
      clo[t * 64 + dof] = c.lower[qa];
      chi[t * 64 + dof] = c.upper[qa];
    }
  }
  std::vector<double> vlim((size_t)(P.n_vel ? P.n_vel : 1) * 64, inf);
  for (int t = 0; t < P.n_vel; ++t) {
    const MkhVelocityLimitDesc& v = d->velocity_limits[t];
    for (int k = 0; k < v.n_indices; ++k) {
      const int dof = v.indices[k];
      if (dof < 0 || dof >= m->nv) return bail(fail(MKH_E_INVALID, "velocity limit: dof %d out of range", dof));
      if (!m->big) vlim[t * 64 + dof] = std::fmin(vlim[t * 64 + dof], v.limit[k]);
    }
  }
  // ---- collision pairs
  std::vector<CollisionPairDev> pairs;
  std::vector<PairCull> culls;
  int n_cv = 0;
  for (int t = 0; t < d->n_collision_limits; ++t) {
    const MkhCollisionLimitDesc& c = d->collision_limits[t];
    for (int k = 0; k < c.n_pairs; ++k) {
      const int g1 = c.geom_id_pairs[2 * k], g2 = c.geom_id_pairs[2 * k + 1];
      if (g1 < 0 || g2 < 0 || g1 >= m->ngeom || g2 >= m->ngeom) return bail(fail(MKH_E_INVALID, "collision pair (%d,%d): geom id out of range", g1, g2));
      CollisionPairDev cp;
      memset(&cp, 0, sizeof cp);
      int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
      // pairs with an analytic routine (collide_dev.h) ...
      auto analytic = [](int a, int b) {
        if (a > b) { int x = a; a = b; b = x; }
        return (a == GEOM_CAPSULE && b == GEOM_CAPSULE) || (a == GEOM_SPHERE && b == GEOM_SPHERE) ||
               (a == GEOM_SPHERE && b == GEOM_CAPSULE) || (a == GEOM_PLANE && (b == GEOM_SPHERE || b == GEOM_CAPSULE)) ||
               (b == GEOM_BOX && (a == GEOM_PLANE || a == GEOM_SPHERE || a == GEOM_CAPSULE || a == GEOM_BOX)) ||
               (b == GEOM_CYLINDER && (a == GEOM_PLANE || a == GEOM_SPHERE || a == GEOM_CAPSULE));
      };
      // ... and the ones that go through the general convex routine (convex_dev.h): cylinder–box, cylinder–cylinder,
      // ellipsoid against any primitive
      // ellipsoid against any primitive, and every pair with a mesh geom (its hull: plane–mesh analytically)
      auto convex = [](int a, int b) {
        if (a > b) { int x = a; a = b; b = x; }
        const bool cb = b == GEOM_SPHERE || b == GEOM_CAPSULE || b == GEOM_ELLIPSOID || b == GEOM_CYLINDER || b == GEOM_BOX || b == GEOM_MESH;
        return cb && (a == GEOM_PLANE ? (b == GEOM_ELLIPSOID || b == GEOM_MESH) : (a >= GEOM_SPHERE && a <= GEOM_MESH));
      };
      auto supported = [&](int a, int b) { return analytic(a, b) || convex(a, b); };
      cp.cv_slot = -1;
      if (!analytic(t1, t2) && convex(t1, t2)) { p->convex_pairs = true; cp.cv_slot = n_cv++; }
      if (!supported(t1, t2)) return bail(fail(MKH_E_INVALID, "collision pair (%d,%d): geom types (%d,%d) are not supported (height fields)", g1, g2, t1, t2));
      for (int side = 0; side < 2; ++side) {
        const int g = side ? g2 : g1;
        if (m->geom_type[g] != GEOM_MESH) continue;
        const int k = m->geom_dataid[g];
        if (k < 0 || !m->d_mesh_vert)
          return bail(fail(MKH_E_INVALID, "collision pair (%d,%d): mesh geom %d has no hull in the model (geom_dataid / mesh_vert)", g1, g2, g));
        (side ? cp.vert2 : cp.vert1) = m->d_mesh_vert + 3 * (size_t)m->mesh_vertadr[k];
        (side ? cp.nvert2 : cp.nvert1) = m->mesh_vertnum[k];
      }
      cp.type1 = t1; cp.type2 = t2; cp.body1 = m->geom_bodyid[g1]; cp.body2 = m->geom_bodyid[g2];
      for (int i = 0; i < 3; ++i) { cp.size1[i] = m->geom_size[3 * g1 + i]; cp.size2[i] = m->geom_size[3 * g2 + i];
                                    cp.lpos1[i] = m->geom_pos[3 * g1 + i]; cp.lpos2[i] = m->geom_pos[3 * g2 + i]; }
      for (int i = 0; i < 4; ++i) { cp.lquat1[i] = m->geom_quat[4 * g1 + i]; cp.lquat2[i] = m->geom_quat[4 * g2 + i]; }
      if (!m->big) { cp.mask1 = dof_chain_mask(m, cp.body1); cp.mask2 = dof_chain_mask(m, cp.body2); }
      cp.gain = c.gain; cp.dmin = c.minimum_distance_from_collisions; cp.ddetect = c.collision_detection_distance;
      cp.relax = c.bound_relaxation;
      pairs.push_back(cp);
      // bounding spheres for the cull pass of problems with more than one wavefront of pairs (mkh_types.h PairCull)
      auto rbound = [&](int g) -> double {
        const double* sz = &m->geom_size[3 * g];
        switch (m->geom_type[g]) {
          case GEOM_SPHERE: return sz[0];
          case GEOM_CAPSULE: return sz[0] + sz[1];
          case GEOM_ELLIPSOID: return std::fmax(sz[0], std::fmax(sz[1], sz[2]));
          case GEOM_CYLINDER: return std::hypot(sz[0], sz[1]);
          case GEOM_BOX: return std::sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
          case GEOM_MESH: {
            const int k = m->geom_dataid[g];
            double r2 = 0.0;
            if (k < 0 || m->h_mesh_vert.empty()) return inf;
            for (int v = 0; v < m->mesh_vertnum[k]; ++v) {
              const double* x = &m->h_mesh_vert[3 * ((size_t)m->mesh_vertadr[k] + v)];
              r2 = std::fmax(r2, x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            }
            return std::sqrt(r2);
          }
          default: return inf;                                 // planes (and anything unbounded): never culled
        }
      };
      PairCull pc;
      memset(&pc, 0, sizeof pc);
      pc.body1 = cp.body1; pc.body2 = cp.body2;
      for (int i = 0; i < 3; ++i) { pc.lpos1[i] = cp.lpos1[i]; pc.lpos2[i] = cp.lpos2[i]; }
      const double reach = (rbound(g1) + rbound(g2) + std::fmax(cp.ddetect, 0.0)) * (1.0 + 1e-9) + 1e-12;
      pc.reach2 = reach * reach;
      culls.push_back(pc);
    }
  }
  P.n_pairs = (int)pairs.size();
  p->simple_pairs = !pairs.empty();
  for (const auto& cp : pairs)
    for (int ty : {cp.type1, cp.type2})
      if (ty != GEOM_PLANE && ty != GEOM_SPHERE && ty != GEOM_CAPSULE) p->simple_pairs = false;
  if (m->big) {
    // beyond one wavefront: the workgroup-per-problem kernel takes every call of this problem (wide_kernel.h)
    const int32_t rc = build_wide_problem(p, m, d, ft, pairs, culls, dcost, dwgain);
    if (rc != MKH_OK) return bail(rc);
    p->wide_only = true;
    if (hipMalloc((void**)&p->d_taps, sizeof(TapArgs)) != hipSuccess || hipMalloc((void**)&p->d_work, 128) != hipSuccess ||
        hipMemset(p->d_work, 0, 128) != hipSuccess)
      return bail(fail(MKH_E_HIP, "descriptor upload failed"));
    snprintf(p->last_kernel, sizeof(p->last_kernel), "ik_wide_kernel");
    p->dev.n_rows_tap = p->wide.n_rows_tap;        // (what mkh_problem_num_task_rows reports; the wavefront descriptor is not built)
    p->last_grid = p->wide_grid; p->last_lds = p->wide_lds; p->last_nt = p->wide.nv + p->wide.max_rows; p->last_block = kWideThreads;
    *out = p;
    return MKH_OK;
  }
  {
    const int want = P.n_pairs + P.n_dense_limit_rows;          // half-space rows that can be active at once
    P.max_rows = want < (kWave - m->nv) ? want : (kWave - m->nv);
    if (const char* cap = dbg_env("MKH_DEBUG_MAX_ROWS")) { const int c = atoi(cap); if (c > 0 && c < P.max_rows) P.max_rows = c; }   // (experiments)
    // LDS behind the per-problem ranges: h of every pair when pairs outnumber rows (row selection), then the expanding polytope's
    // workspace when some pair goes through the general convex routine (collision_phase: the same two terms)
    P.n_hsel = (P.n_pairs > P.max_rows ? lds_even(P.n_pairs) : 0) + (p->convex_pairs ? (kEpaWsDoubles > kGjkWsDoubles ? kEpaWsDoubles : kGjkWsDoubles) : 0);
    // the cull pass's candidate list (16-bit pair indices) when the pairs do not fit one trip of the wavefront
    p->use_cull = P.n_pairs > kWave && P.n_pairs < 65536 && !p->simple_pairs && !(p->diag & MKH_DIAG_NO_PAIR_CULL);
    if (p->use_cull) P.n_hsel += lds_even((P.n_pairs + 3) / 4);
  }
  const int ntab = m->nv + P.max_rows;
  {
    static const int kVariants[] = {8, 16, 24, 32, 44, 48, 64};
    p->nt = 64;
    for (int v : kVariants) if (ntab <= v) { p->nt = v; break; }
  }

  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = upload(ft, &p->d_frame);
  if (e == hipSuccess) e = upload(pcost, &p->d_posture_cost);
  if (e == hipSuccess) e = upload(clo, &p->d_cfg_lower);
  if (e == hipSuccess) e = upload(chi, &p->d_cfg_upper);
  if (e == hipSuccess) e = upload(vlim, &p->d_vel);
  if (e == hipSuccess) e = upload(pairs, &p->d_pairs);
  if (e == hipSuccess && p->use_cull) e = upload(culls, &p->d_cull);
  if (e == hipSuccess) e = upload(dcost, &p->d_dense_cost);
  if (e == hipSuccess) e = upload(dwgain, &p->d_dense_wgain);
  if (e != hipSuccess) return bail(fail(MKH_E_HIP, "problem upload: %s", hipGetErrorString(e)));
  P.frame = p->d_frame; P.posture_cost = p->d_posture_cost; P.cfg_lower = p->d_cfg_lower; P.cfg_upper = p->d_cfg_upper;
  P.vel_limit = p->d_vel; P.pairs = p->d_pairs; P.cull = p->use_cull ? p->d_cull : nullptr;
  P.dense_cost = p->d_dense_cost; P.dense_wgain = p->d_dense_wgain;

  P.nt = p->nt;
  p->nt_full = p->nt < 32 ? 32 : p->nt;
  // Second LDS buffers for the next problem's inputs (ik_kernel.h "load inputs") only where they do not cost a
  // resident wave in the lean or the all-feature variant of this problem (the low-rank variant is checked below).
  auto lds_of = [&](int nt, bool pre) {
    return lds_layout(P.nq, P.nv, P.nbody, P.njnt, P.n_frame, P.n_posture, P.n_com, P.max_rows, 6, j_stride_direct(P.nv, nt), 0, pre, false, false, P.n_hsel).total *
           (int)sizeof(double);
  };
  P.prefetch = (waves_per_cu(p->nt, lds_of(p->nt, true)) == waves_per_cu(p->nt, lds_of(p->nt, false)) &&
                waves_per_cu(p->nt_full, lds_of(p->nt_full, true)) == waves_per_cu(p->nt_full, lds_of(p->nt_full, false))) ? 1 : 0;
  p->lds_bytes = lds_of(p->nt, P.prefetch != 0);
  if (p->lds_bytes > 64 * 1024) return bail(fail(MKH_E_LIMIT, "problem needs %d bytes of LDS per wavefront (> 64 KiB)", p->lds_bytes));
  p->blocks_per_cu = waves_per_cu(p->nt, p->lds_bytes);
  p->lds_bytes_full = lds_of(p->nt_full, P.prefetch != 0);
  if (p->blocks_per_cu < 1) p->blocks_per_cu = 1;
  // 3 waves per SIMD: tableau sizes with a TabW3 map (build.py W3), no half-space rows (the compact layout lets the
  // Jacobian rows reuse the body poses, which the collision phase still reads), and 12 wavefronts' LDS must fit the CU
  P.prefetch_w3 = 0;
  if (p->nt == 44 && P.max_rows == 0 && P.n_dense_rows == 0) {
    auto lds_w3 = [&](bool pre) {
      return lds_layout(P.nq, P.nv, P.nbody, P.njnt, P.n_frame, P.n_posture, P.n_com, P.max_rows, 6, j_stride_direct(P.nv, p->nt), 0,
                        pre, true).total * (int)sizeof(double);
    };
    if (waves_per_cu(p->nt, lds_w3(false), true) == 12) {
      P.prefetch_w3 = waves_per_cu(p->nt, lds_w3(true), true) == 12 ? 1 : 0;
      p->lds_bytes_w3 = lds_w3(P.prefetch_w3 != 0);
    }
  }
  // ---- direct start: Jacobian columns by (task, dof) pair lanes when they fit one wavefront
  P.n_dpairs = 0;
  {
    int n = 0;
    bool fits = true;
    for (size_t t = 0; t < ft.size() && fits; ++t) {
      const uint64_t cm = ft[t].dof_mask | (ft[t].relative ? ft[t].root_mask : 0ull);
      for (int k = 0; k < m->nv && fits; ++k)
        if ((cm >> k) & 1) {
          if (n >= kWave) { fits = false; break; }
          P.dpair_task[n] = (int16_t)t; P.dpair_dof[n] = (int16_t)k; ++n;
        }
    }
    if (fits) P.n_dpairs = n;
  }
  // The low-rank start's lane tables (ik_kernel.h wood_start): which column / row chunk of Jh·Jhᵀ a lane computes and over which
  // dofs, the (task, dof) pair lanes of the Jacobian rows, the source of each residual's weighted error.  They depend on the
  // tasks only — shared by the builds without rows and, round 6, the ones with half-space rows.  false: more than 256 pairs.
  auto wood_tables = [&]() -> bool {
    // (column, row-chunk) lanes of the Jh·Jhᵀ product: rows 0..n_jrows (the last one is the rhs)
    const int groups = kWave / P.n_jrows;
    P.wood_rpc = (P.n_jrows + 1 + groups - 1) / groups;
    for (int l = 0; l < kWave; ++l) {
      const int ch = l / P.n_jrows;
      P.wood_col[l] = ch < groups ? l % P.n_jrows : -1;
      P.wood_row0[l] = ch < groups ? ch * P.wood_rpc : 0;
      P.wood_mask[l] = 0;
      if (ch < groups)
        for (size_t t = 0; t < ft.size(); ++t)
          if (l % P.n_jrows >= ft[t].jrow0 && l % P.n_jrows < ft[t].jrow0 + __builtin_popcount(ft[t].rowmask & 63))
            P.wood_mask[l] = ft[t].dof_mask;
    }
    // Jacobian columns by (task, dof) pair lanes; source of each residual's weighted error
    P.n_jpairs = 0;
    for (size_t t = 0; t < ft.size(); ++t) {
      for (int k = 0; k < m->nv; ++k)
        if ((ft[t].dof_mask >> k) & 1) {
          if (P.n_jpairs >= 256) return false;
          P.jpair_task[P.n_jpairs] = (int16_t)t; P.jpair_dof[P.n_jpairs] = (int16_t)k; ++P.n_jpairs;
        }
      int c = 0;
      for (int r = 0; r < 6; ++r)
        if ((ft[t].rowmask >> r) & 1) { P.mu_src[ft[t].jrow0 + c] = (int16_t)(t * 64 + 30 + r); ++c; }
    }
    // ComTask rows: dense over the robot's dofs; their weighted error is computed on the device (mu_src < 0)
    for (int t = 0; t < P.n_com; ++t) {
      int c = 0;
      for (int r = 0; r < 3; ++r)
        if ((P.com_rowmask[t] >> r) & 1) { P.mu_src[P.com_jrow0[t] + c] = (int16_t)(-1 - 3 * t - r); ++c; }
      for (int l = 0; l < kWave; ++l)
        if (P.wood_col[l] >= P.com_jrow0[t] && P.wood_col[l] < P.com_jrow0[t] + c)
          P.wood_mask[l] = m->nv >= 64 ? ~0ull : ((1ull << m->nv) - 1ull);
    }
    return true;
  };
  // what the low-rank start's stability criterion compares (launch(): damping + the smallest posture diagonal against the largest
  // task cost²)
  auto wood_scales = [&]() {
    double mn = __builtin_huge_val();
    for (int i = 0; i < m->nv; ++i) {
      double dsum = 0.0;
      for (int t = 0; t < d->n_posture_tasks; ++t) {
        // free-joint dofs carry no posture term (posture_task.py:115-116,139-141)
        const int jt = m->jnt_type[m->dof_jntid[i]];
        if (jt != 0) dsum += pcost[t * 64 + i] * pcost[t * 64 + i];
      }
      mn = dsum < mn ? dsum : mn;
    }
    p->wood_min_diag = mn;
    p->wood_max_cost2 = 0.0;
    for (const auto& f : ft) for (int k = 0; k < 6; ++k) p->wood_max_cost2 = fmax(p->wood_max_cost2, f.cost[k] * f.cost[k]);
    for (int t = 0; t < P.n_com; ++t) for (int k = 0; k < 3; ++k) p->wood_max_cost2 = fmax(p->wood_max_cost2, P.com_cost[t][k] * P.com_cost[t][k]);
  };
  if (P.n_jrows > 0 && P.n_pairs == 0 && !p->has_relative && P.n_dense_rows == 0 &&
      P.n_dense_limit_rows == 0 && P.n_jrows <= kMuBig) {
    // (NT = NR: the task residuals are eliminated outside the tableau, one column of [S | Jh] per lane — wood_start; the
    //  S columns sit on lanes [NR, NR + n_μ) or, when those do not exist, on lanes [0, n_μ) in a second register set)
    static const int kWoodVariants[] = {16, 24, 32, 44, 48};
    int cand = 0;
    for (int v : kWoodVariants)
      if (m->nv <= v && (v + P.n_jrows <= kWave || P.n_jrows <= v)) { cand = v; break; }
    const bool big = P.n_jrows > kMu || cand + P.n_jrows > kWave;
    // When it pays.  Fewer task rows than dofs by a margin — half, three quarters for humanoid-size tableaus — measured in
    // rounds 1-2 on arms, hands and the G1; round 4 (tools/bench_wood_criterion.py, H1 / Go1 / Spot / Allegro / Shadow with
    // 12-18 rows on 18-25 dofs): whenever the small elimination applies (≤ kMu rows, S columns on lanes of their own) the
    // low-rank start wins by 1.15-1.38 x up to rows = dofs; the large one (> kMu rows) loses beyond three quarters
    // (Shadow 21 / 24: 0.57 x, H1 24 / 25: 0.93 x) and keeps the old margin.
    const bool pays = 2 * P.n_jrows <= m->nv || (m->nv >= 32 && 4 * P.n_jrows <= 3 * m->nv) ||
                      (!big && m->nv >= 16 && P.n_jrows <= m->nv) || dbg_env("MKH_DEBUG_WOOD_ALWAYS");
    if (cand && pays) { p->wood_nt = cand; p->wood_nr = cand; }
    if (p->wood_nt) {
      p->wood_big = P.n_jrows > kMu || p->wood_nr + P.n_jrows > kWave;
      const int sp = lds_even(P.n_jrows);
      auto lds_wood = [&](bool pre, bool compact, bool piv_small = false) {
        return lds_layout(P.nq, P.nv, P.nbody, P.njnt, P.n_frame, P.n_posture, P.n_com, P.max_rows, P.n_jrows + 1, p->wood_nr,
                          wood_s_aliases_dof(P.nv, P.n_jrows, sp, P.n_com > 0 ? P.nbody : 0) ? 0 : P.n_jrows * (sp + 1),
                          pre, compact, true, 0, piv_small);
      };
      // 2-waves map: the plain layout, or — when that would cost a resident wave and the pair lanes need one pass only (the
      // compact layout lets the Jacobian rows overwrite the task blocks) — the compact one
      P.wood_compact = 0; P.prefetch_wc = 0;
      P.wood_refine = (p->diag & MKH_DIAG_NO_COLD_REFINE) ? 0 : 1;
      {
        auto bytes = [&](bool pre, bool compact) { return lds_wood(pre, compact).total * (int)sizeof(double); };
        if (waves_per_cu(p->wood_nt, bytes(false, false)) < waves_per_cu(p->wood_nt, 1) && P.n_jrows > 0) {
          int n_jp = 0;
          for (size_t t = 0; t < ft.size(); ++t) n_jp += __builtin_popcountll(ft[t].dof_mask);
          if (n_jp <= kWave && waves_per_cu(p->wood_nt, bytes(false, true)) > waves_per_cu(p->wood_nt, bytes(false, false))) {
            P.wood_compact = 1;
            P.prefetch_wc = waves_per_cu(p->wood_nt, bytes(true, true)) == waves_per_cu(p->wood_nt, bytes(false, true)) ? 1 : 0;
          }
        }
      }
      const LdsLayout Lw = P.wood_compact ? lds_wood(P.prefetch_wc != 0, true) : lds_wood(P.prefetch != 0, false);
      if (!wood_tables()) p->wood_nt = 0;
      p->wood_lds_bytes = Lw.total * (int)sizeof(double);
      if (p->wood_lds_bytes * 8 > 160 * 1024) p->wood_nt = 0;      // would cost residency
      // 3 waves per SIMD (compact layout: the Jacobian rows overwrite the task blocks, so the pair lanes need one pass)
      P.prefetch_w3w = 0;
      if (p->wood_nt != 48 && P.n_jpairs <= kWave && P.n_com == 0 && !p->wood_big) {   // (build.py W3_WOOD: 16, 24, 32, 44)
        const int full = waves_per_cu(p->wood_nt, 1, true);     // 16 waves per CU for NT ≤ 24, else 12
        if (waves_per_cu(p->wood_nt, lds_wood(false, true).total * (int)sizeof(double), true) == full) {
          P.prefetch_w3w = waves_per_cu(p->wood_nt, lds_wood(true, true).total * (int)sizeof(double), true) == full ? 1 : 0;
          p->wood_lds_bytes_w3 = lds_wood(P.prefetch_w3w != 0, true).total * (int)sizeof(double);
        }
      }
      // ... and the F_COM builds of a humanoid-size robot (ComTask rows and / or up to 24 task rows: `44_36_r44_w3`, round 5).
      // Their 25 rows of Jh take 8.8 KB of the compact layout — 15.4 KB for the G1 full example, 10 wavefronts per CU instead
      // of 8 — so the bar is "more resident wavefronts than the two-waves map", not all twelve.
      static const bool no_com_w3 = dbg_env("MKH_DEBUG_NO_COM_W3") != nullptr;       // (A/B switch)
      if (!no_com_w3 && p->wood_nt == 44 && P.n_jpairs <= kWave && (P.n_com > 0 || p->wood_big)) {
        const int two = waves_per_cu(p->wood_nt, 1, false);     // 8
        const int w = waves_per_cu(p->wood_nt, lds_wood(false, true, true).total * (int)sizeof(double), true);
        if (w >= two + 2) {                                   // (nine per CU — 3 + 2 + 2 + 2 — measured SLOWER than eight: 1.51 vs 1.44 ms)
          P.prefetch_w3w = waves_per_cu(p->wood_nt, lds_wood(true, true, true).total * (int)sizeof(double), true) == w ? 1 : 0;
          p->wood_lds_bytes_w3 = lds_wood(P.prefetch_w3w != 0, true, true).total * (int)sizeof(double);
        }
      }
      wood_scales();
    }
  }
  // ---- low-rank start WITH half-space rows (round 6; ik_kernel.h wood_start "half-space rows"): a collision problem whose
  // tasks qualify for the low-rank start keeps it — every contact row is one more column of the n_μ-step elimination instead of
  // the reason for nv single pivots on the whole tableau (47 % of `g1_coll`).  Analytic pair sets (FEAT 8 → 40), frame + posture
  // tasks, at most kMu task rows, one pass of (task, dof) pair lanes (the Jacobian rows overwrite the task blocks in LDS) and at most
  // 8 half-space rows per instance (the rows' Gram entries wait in the contact table); a 48-row tableau (mkh_problem_create below).
  if (P.n_jrows > 0 && P.n_jrows <= kMu && P.n_pairs > 0 && !p->has_relative && P.n_com == 0 && P.n_dense_rows == 0 &&
      P.n_dense_limit_rows == 0 && !P.dense_box && !p->simple_pairs && !p->convex_pairs) {
    if (wood_tables() && P.n_jpairs <= kWave) { p->woodr_tables = true; wood_scales(); }
  }
  auto woodr_bytes = [&](const DeviceProblem& D, bool pre) {
    return lds_layout(D.nq, D.nv, D.nbody, D.njnt, D.n_frame, D.n_posture, D.n_com, D.max_rows, D.n_jrows + 1, 48,
                      D.n_jrows * (lds_even(D.n_jrows) + 1), pre, false, true, D.n_hsel, true, true).total * (int)sizeof(double);
  };
  if (p->woodr_tables && p->nt == 48 && P.max_rows <= 8) {
    P.prefetch_wc = waves_per_cu(48, woodr_bytes(P, true)) == waves_per_cu(48, woodr_bytes(P, false)) ? 1 : 0;
    p->woodr_lds = woodr_bytes(P, P.prefetch_wc != 0);
  }
  {
    LaneProblem2 lp;
    const int small_cls = build_lane_problem(m, d, P, ft, pcost, clo, chi, vlim, p->has_relative, lp);
    p->lane_nv = small_cls <= 8 ? small_cls : 0;
    p->quad_nt = small_cls ? (small_cls <= 8 ? 8 : (small_cls == 32 ? 32 : 16)) : 0;
    if (dbg_env("MKH_DEBUG_NO_PAIR_ROWS") && p->quad_nt == 32) p->quad_nt = 0;   // (A/B switch: 17 … 32-dof robots back on the wavefront kernel)
    if (small_cls) {
      p->lane_lds = lane_lds_bytes(lp.nlink);
      bool ident = true;
      for (int dd = 0; dd < lp.nv; ++dd) ident = ident && lp.dof_qadr[dd] == dd;
      p->lane_dims = LaneDims{lp.nq, lp.nv, lp.nlink, lp.n_frame, lp.n_posture, lp.n_cfg, lp.n_vel, ident ? 1 : 0};
      // (the lane kernel and the one-row builds of the row kernel read the small descriptor, the two-row build the large one)
      LaneProblem lp1;
      if (small_cls != 32) lane_problem_narrow(lp, lp1);
      const void* src = small_cls == 32 ? (const void*)&lp : (const void*)&lp1;
      const size_t bytes = small_cls == 32 ? sizeof(LaneProblem2) : sizeof(LaneProblem);
      if (hipMalloc((void**)&p->d_lane, bytes) != hipSuccess || hipMemcpy(p->d_lane, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(MKH_E_HIP, "lane descriptor upload failed"));
    }
  }
  // Tight rows: a hand with 40 capsule pairs has ≈17 contacts inside the detection distance at a time, but 40 rows put it
  // on the 64-row tableau (128 pinned VGPRs, 23 KB of LDS: 7 wavefronts per CU).  With 48 − nv rows it runs on the 48-row
  // build — Shadow config 4: 0.398 → 0.343 ms — and the rare instance with a violated dropped contact is solved again on
  // the full-row build (launch()).
  {
    const DeviceProblem& P0 = p->dev;
    const int cap = 48 - m->nv;
    static const bool no_tight = dbg_env("MKH_DEBUG_NO_TIGHT_ROWS") != nullptr;      // (A/B of this path)
    // Round 5: the same for pair sets with boxes, cylinders, ... (the analytic collision build) — a humanoid-size robot reserves
    // 64 − nv rows, which costs the 64-row build a resident wavefront or two in LDS: G1 with 46 pairs (21 rows, 24.6 KB: 6
    // wavefronts per CU) 0.757 → ≈ 0.60 ms with 5 rows on the 48-row build.  Few contacts are in range at a time when
    // collision avoidance works; when many are, the cost is the two launches (MKH_FLAG_FULL_ROWS pins the full-row build).
    //  Humanoid-size robots only (32 dofs and up): ALOHA (16 dofs, 48 rows reserved, 1 104 pairs) measures 2.26 → 2.53 ms this
    //  way — its full build is not short of wavefronts, and the instances that overflow 32 rows walk the pair list twice.
    const bool generic = !p->simple_pairs && !p->convex_pairs && P0.n_pairs > 0 && m->nv >= 32 && cap >= 4 && !dbg_env("MKH_DEBUG_NO_TIGHT_GENERIC");
    if (!no_tight && (p->simple_pairs ? cap >= 16 : generic) && P0.n_dense_limit_rows == 0 && P0.n_dense_rows == 0 && !P0.dense_box && p->nt == 64 &&
        P0.n_pairs > cap) {
      DeviceProblem T = P0;
      T.max_rows = cap;
      T.n_hsel = lds_even(T.n_pairs) + (p->use_cull ? lds_even((T.n_pairs + 3) / 4) : 0);
      T.nt = 48;
      auto lds_t = [&](bool pre) {
        return lds_layout(T.nq, T.nv, T.nbody, T.njnt, T.n_frame, T.n_posture, T.n_com, T.max_rows, 6, j_stride_direct(T.nv, 48), 0, pre,
                          false, false, T.n_hsel).total * (int)sizeof(double);
      };
      T.prefetch = waves_per_cu(48, lds_t(true)) == waves_per_cu(48, lds_t(false)) ? 1 : 0;
      p->nt_tight = 48;
      p->lds_tight = lds_t(T.prefetch != 0);
      if (p->woodr_tables && T.max_rows <= 8) {               // the tight launch with the low-rank start (round 6)
        T.prefetch_wc = waves_per_cu(48, woodr_bytes(T, true)) == waves_per_cu(48, woodr_bytes(T, false)) ? 1 : 0;
        p->woodr_lds_tight = woodr_bytes(T, T.prefetch_wc != 0);
      }
      if (hipMalloc((void**)&p->d_dev_tight, sizeof(DeviceProblem)) != hipSuccess ||
          hipMemcpy(p->d_dev_tight, &T, sizeof(DeviceProblem), hipMemcpyHostToDevice) != hipSuccess ||
          hipMalloc((void**)&p->d_status_tight, (size_t)p->max_batch * sizeof(int32_t)) != hipSuccess)
        return bail(fail(MKH_E_HIP, "descriptor upload failed"));
    }
  }
  if (hipMalloc((void**)&p->d_dev, sizeof(DeviceProblem)) != hipSuccess ||
      hipMemcpy(p->d_dev, &p->dev, sizeof(DeviceProblem), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc((void**)&p->d_taps, sizeof(TapArgs)) != hipSuccess ||
      hipMalloc((void**)&p->d_work, 128) != hipSuccess || hipMemset(p->d_work, 0, 128) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess)   // (hipMemset may return before the fill has run; launches may come on any stream)
    return bail(fail(MKH_E_HIP, "descriptor upload failed"));
  // More rows can be active at once than the 64 − nv the wavefront kernels hold (the reference stacks them all:
  // mink/solve_ik.py:25-40): the instances a launch flags MKH_ST_ROW_OVERFLOW are solved again by the workgroup-per-problem
  // kernel with every row (launch(): the redo launch behind plain solves)
  const bool no_wide = (p->diag & MKH_DIAG_NO_WIDE_REDO) != 0;                  // (tests: what the wavefront kernel alone leaves flagged)
  // (round 5: EVERY problem with half-space rows gets this twin — its dense Goldfarb–Idnani iteration also re-solves the instances
  //  whose active rows are almost dependent, wherever they occur)
  if (!no_wide && P.n_pairs + P.n_dense_limit_rows > 0) {
    const int32_t rc = build_wide_problem(p, m, d, ft, pairs, culls, dcost, dwgain);
    if (rc != MKH_OK) return bail(rc);
  }
  // The two-kernel split of general convex pairs (convex_pre.hip): per convex pair the chains root → body of its two geoms, and
  // the buffer of pre-evaluated contacts the analytic build reads.  It uses the plain model arrays of the twin above.
  static const bool no_split = dbg_env("MKH_DEBUG_NO_CONVEX_SPLIT") != nullptr;      // (A/B: the in-kernel routine, round 4's path)
  if (p->d_wide && n_cv > 0 && !no_split) {
    std::vector<int32_t> cv_pair, adr{0}, chain;
    for (size_t k = 0; k < pairs.size(); ++k) {
      if (pairs[k].cv_slot < 0) continue;
      cv_pair.push_back((int32_t)k);
      for (int body : {pairs[k].body1, pairs[k].body2}) {
        std::vector<int32_t> up;
        for (int b = body; b > 0; b = m->body_parentid[b]) up.push_back(b);
        chain.insert(chain.end(), up.rbegin(), up.rend());
        adr.push_back((int32_t)chain.size());
      }
    }
    if (upload(cv_pair, &p->d_cv_pair) != hipSuccess || upload(adr, &p->d_cv_adr) != hipSuccess || upload(chain, &p->d_cv_chain) != hipSuccess ||
        hipMalloc((void**)&p->d_cv, (size_t)p->max_batch * n_cv * 7 * sizeof(double)) != hipSuccess)
      return bail(fail(MKH_E_HIP, "convex pre-pass buffers"));
    p->cv = mkh::CvPre{n_cv, p->d_cv_pair, p->d_cv_adr, p->d_cv_chain};
    p->dev.n_cv = n_cv; p->dev.cv_contacts = p->d_cv;
    if (hipMemcpy(p->d_dev, &p->dev, sizeof(DeviceProblem), hipMemcpyHostToDevice) != hipSuccess) return bail(fail(MKH_E_HIP, "descriptor upload failed"));
  }
  *out = p;
  return MKH_OK;
}

void mkh_problem_destroy(MkhProblem* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipFree(p->d_frame); (void)hipFree(p->d_posture_cost); (void)hipFree(p->d_cfg_lower); (void)hipFree(p->d_cfg_upper);
  (void)hipFree(p->d_vel); (void)hipFree(p->d_pairs); (void)hipFree(p->d_cull); (void)hipFree(p->d_dev); (void)hipFree(p->d_taps); (void)hipFree(p->d_work);
  (void)hipFree(p->d_lane); (void)hipFree(p->d_warm); (void)hipFree(p->d_qkeep);
  (void)hipFree(p->d_cv); (void)hipFree(p->d_cv_pair); (void)hipFree(p->d_cv_adr); (void)hipFree(p->d_cv_chain); (void)hipFree(p->d_dev_tight); (void)hipFree(p->d_status_tight);
  (void)hipFree(p->d_dense_cost); (void)hipFree(p->d_dense_wgain); (void)hipFree(p->s_iters);
  (void)hipFree(p->s_de); (void)hipFree(p->s_dJ); (void)hipFree(p->s_dG); (void)hipFree(p->s_dh); (void)hipFree(p->s_dbox); (void)hipFree(p->d_clk);
  for (void* w : p->wide_allocs) (void)hipFree(w);
  (void)hipFree(p->d_wide);
  (void)hipFree(p->s_q); (void)hipFree(p->s_ft); (void)hipFree(p->s_pt); (void)hipFree(p->s_ct); (void)hipFree(p->s_v); (void)hipFree(p->s_status);
  p->small.release();
  if (p->st_in) (void)hipStreamDestroy(p->st_in);
  if (p->st_out) (void)hipStreamDestroy(p->st_out);
  if (p->ev_start) (void)hipEventDestroy(p->ev_start);
  for (int i = 0; i < 4; ++i) { if (p->ev_in[i]) (void)hipEventDestroy(p->ev_in[i]); if (p->ev_k[i]) (void)hipEventDestroy(p->ev_k[i]); }
  delete p;
}

const char* mkh_problem_last_kernel(const MkhProblem* p) { return p ? p->last_kernel : ""; }
int32_t mkh_problem_num_task_rows(const MkhProblem* p) { return p ? p->dev.n_rows_tap : 0; }
int32_t mkh_problem_num_collision_pairs(const MkhProblem* p) { return p ? p->dev.n_pairs : 0; }

static int grid_for(const MkhProblem* p, int B) {
  int g = p->model->num_cus * p->blocks_per_cu;
  return B < g ? B : g;
}
// LDS bytes of the plain (direct-start) layout of this problem with an nt-row tableau (the kernel computes the same layout
// from its own NT: ik_kernel.h kernel_lds_layout)
static int lds_for_nt(const MkhProblem* p, int nt) {
  const DeviceProblem& P = p->dev;
  return lds_layout(P.nq, P.nv, P.nbody, P.njnt, P.n_frame, P.n_posture, P.n_com, P.max_rows, 6, j_stride_direct(P.nv, nt), 0,
                    P.prefetch != 0, false, false, P.n_hsel).total * (int)sizeof(double);
}

// Static rounds of a wavefront kernel's problem distribution (ik_kernel.h): each wavefront first walks `static` problems of its
// XCD's contiguous row range, the rest of the batch goes through the ticket counter.  Static rows share cache lines inside one
// L2 and cost no atomic; the ticket tail evens out what the static part left uneven — QP work varies by ±9 % per problem,
// CUs differ by a few per cent — and that grows with the number of rounds.  Round 1 measured the 1.18-ms kernel of its day
// flat between 4/16 and 14/16 static and took 7/8; on today's kernels (round 5, sweep by sixteenths at 65 536 / 32 768 / 16 384
// G1 instances, the plugin workload and the G1 full example): 21 rounds per wavefront want 6 of them dynamic (0.779 -> 0.752 ms
// on the headline), 10 want 2, 5 want 1, 32 of the two-waves plugin build 6 or more (1.681 -> 1.632 ms).
// `uneven`: resident wavefronts per CU that do not divide by its four SIMDs (ten: 3 + 3 + 2 + 2) — a wavefront that shares its
// SIMD with two others is slower than one that shares it with one, so nearly half of the batch goes through the counter there
// (G1 full example 1.37 -> 1.24 ms).  MKH_DEBUG_STATIC_78=1: the old 7/8 rule; MKH_DEBUG_STATIC_16THS=n: n sixteenths (sweeps).
// Below this many rounds per wavefront the batch is dealt statically as a whole (round 5, kernel ms on G1 config 3, ticket tail /
// one strided share per wavefront: 12 288 instances = 4 rounds 0.186 / 0.177, 16 384 = 5.3 rounds 0.243 / 0.224, 20 480 = 6.7 rounds
// 0.282 / 0.270, 24 576 = 8 rounds 0.321 / 0.335, 32 768 0.409 / 0.422, 65 536 0.754 / 0.827): over a few rounds the work of
// the wavefronts has not drifted apart yet, and a ticket tail only adds its own ragged last round.  Round 1's threshold was 4.
constexpr int kMinRoundsForTickets = 8;
static int static_rounds_for(int per_wave, bool uneven, bool loops = false) {
  static const bool static78 = dbg_env("MKH_DEBUG_STATIC_78") != nullptr;
  static const int dbg_16ths = dbg_env("MKH_DEBUG_STATIC_16THS") ? atoi(dbg_env("MKH_DEBUG_STATIC_16THS")) : -1;
  if (dbg_16ths >= 0 && dbg_16ths <= 16) return (per_wave * dbg_16ths) / 16;
  if (dbg_16ths == 99) return INT32_MAX;                     // (no ticket counter at all: one strided share per wavefront)
  if (static78) return (per_wave * 7) / 8;
  if (uneven) return (per_wave * 9) / 16;
  // (fused loops: a problem is 3 … 40 solves long and the threshold-terminated ones differ by that much — half of the batch
  //  dynamic: converged targets of the headline's loop leg 5.45 -> 5.65 M/s; the fixed-count loops do not care)
  if (loops) return per_wave / 2;
  const int dyn = (3 * per_wave - 2) / 10;
  return per_wave - (dyn < 1 ? 1 : dyn);
}

static int grid_for_variant(const MkhProblem* p, int B, int nt, int lds, bool w3 = false) {
  int wpc = waves_per_cu(nt, lds, w3);
  // diagnostic: cap the resident waves per CU (occupancy experiments, docs/HISTORY.md §7b); never raises it
  static const int dbg = dbg_env("MKH_DEBUG_WAVES_PER_CU") ? atoi(dbg_env("MKH_DEBUG_WAVES_PER_CU")) : 0;
  if (dbg > 0 && dbg < wpc) wpc = dbg;
  int g = p->model->num_cus * wpc;
  return B < g ? B : g;
}

int32_t mkh_problem_launch_info(const MkhProblem* p, int32_t B, int32_t* grid, int32_t* block, int32_t* lds_bytes,
                                int32_t* tableau_rows) {
  if (!p) return fail(MKH_E_INVALID, "null problem");
  // what the most recent launch of this problem used (the variant depends on the call: taps, fused steps, the
  // low-rank QP start); before any launch, the lean direct variant
  if (grid) *grid = p->wide_only ? (p->wide_grid < B ? p->wide_grid : B) : (p->last_nt ? p->last_grid : grid_for(p, B));
  if (block) *block = p->last_block;        // (64: one wavefront per workgroup; 256 on the workgroup-per-problem kernel)
  if (lds_bytes) *lds_bytes = p->last_nt ? p->last_lds : p->lds_bytes;
  if (tableau_rows) *tableau_rows = p->last_nt ? p->last_nt : p->nt;
  return MKH_OK;
}

// Experiment builds with -DMKH_CLOCKS (tools/phase_clocks.py): MKH_DEBUG_CLOCKS=<file> makes every launch of this process
// synchronous and writes its (B, 24) cycle stamps to <file> — phase profile of kernels that cannot be tapped
static const char* clk_path() { static const char* const s = dbg_env("MKH_DEBUG_CLOCKS"); return s; }
static hipError_t clk_begin(MkhProblem* p, SolveArgs& a, hipStream_t stream) {
  if (!clk_path()) return hipSuccess;
  if (!p->d_clk)
    if (hipError_t e = hipMalloc((void**)&p->d_clk, (size_t)p->max_batch * 24 * sizeof(long long))) return e;
  a.clk = p->d_clk;
  if (hipError_t e = hipMemsetAsync(p->d_clk, 0, (size_t)a.B * 24 * sizeof(long long), stream)) return e;
  // MKH_DEBUG_PHASE_STOP=k (tools/phase_census.sh): slot 15 of every row = the phase boundary after which the kernel abandons the solve
  static const int stop = dbg_env("MKH_DEBUG_PHASE_STOP") ? atoi(dbg_env("MKH_DEBUG_PHASE_STOP")) : 0;
  if (stop) {
    std::vector<long long> h((size_t)a.B, (long long)stop);
    if (hipError_t e = hipMemcpy2DAsync(p->d_clk + 15, 24 * sizeof(long long), h.data(), sizeof(long long), sizeof(long long), (size_t)a.B,
                                        hipMemcpyHostToDevice, stream)) return e;
    return hipStreamSynchronize(stream);
  }
  return hipSuccess;
}
static hipError_t clk_end(MkhProblem* p, int B, hipStream_t stream) {
  if (!clk_path()) return hipSuccess;
  std::vector<long long> h((size_t)B * 24);
  if (hipError_t e = hipMemcpyAsync(h.data(), p->d_clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost, stream)) return e;
  if (hipError_t e = hipStreamSynchronize(stream)) return e;
  if (FILE* f = fopen(clk_path(), "wb")) { fwrite(h.data(), sizeof(long long), h.size(), f); fclose(f); }
  return hipSuccess;
}

static int32_t launch(MkhProblem* p, const SolveArgs& a_in, const TapArgs* taps, hipStream_t stream, int32_t flags) {
  // (the kernel choice must not depend on whether the caller wants the status: a call without status_out on a handle with a
  //  tight-rows build keeps it in a buffer of the handle — round-3 advisor finding)
  SolveArgs a = a_in;
  if (!a.status_out && p->d_status_tight && a.do_qp) a.status_out = p->d_status_tight;
  if (p->wide_only) {
    if (taps && taps->t_cycles)
      return fail(MKH_E_INVALID, "models beyond one wavefront (more than 64 bodies or dofs) run on the workgroup-per-problem kernel, "
                                 "which has no cycle-counter tap");
    const TapArgs* dt_ = nullptr;
    if (taps) {
      HIP_OK(hipMemcpyAsync(p->d_taps, taps, sizeof(TapArgs), hipMemcpyHostToDevice, stream));
      HIP_OK(hipStreamSynchronize(stream));
      dt_ = p->d_taps;
    }
    snprintf(p->last_kernel, sizeof(p->last_kernel), "ik_wide_kernel");
    p->last_grid = p->wide_grid < a.B ? p->wide_grid : a.B; p->last_lds = p->wide_lds; p->last_nt = p->wide.nv + p->wide.max_rows;
    p->last_block = kWideThreads;
    HIP_OK(clk_begin(p, a, stream));                         // (MKH_DEBUG_CLOCKS=<file>: phase stamps of every problem, tools/wide_phase_clocks.py)
    const int32_t rcw = launch_wide_kernel(p, a, stream, 0, dt_);
    if (rcw == MKH_OK) HIP_OK(clk_end(p, a.B, stream));
    return rcw;
  }
  // A problem whose rows can outnumber the tableau's: the flagged instances once more, with every row — plain solves, calls
  // with taps (the workgroup-per-problem kernel writes every tap but the cycle counters) and, round 5, the fused loops: an
  // instance that overflows at some step runs its whole loop again from its ORIGINAL q, of which the handle keeps a copy for
  // the duration of the call (q_out may alias q) — a device-to-device copy of B·nq doubles in front of the launch
  const bool wide_redo = p->d_wide && a.do_qp && a.status_out && !(taps && taps->t_cycles);
  const double* q_redo = a.q;
  // (only when q_out really aliases q: otherwise the redo reads the caller's q, which the first launch did not touch — the copy
  //  was a latency cost of every small-batch control loop with collision limits, round-5 advisor finding)
  if (wide_redo && a.q_out && a.q_out == a.q) {
    HIP_OK(ensure(&p->d_qkeep, (size_t)p->max_batch * p->dev.nq));
    HIP_OK(hipMemcpyAsync(p->d_qkeep, a.q, (size_t)a.B * p->dev.nq * sizeof(double), hipMemcpyDefault, stream));
    q_redo = p->d_qkeep;
  }
  p->last_block = kWave;
  (void)hipGetLastError();          // a stale error of an unrelated earlier runtime call must not be blamed on this launch
  const TapArgs* dtaps = nullptr;
  if (taps) {
    HIP_OK(hipMemcpyAsync(p->d_taps, taps, sizeof(TapArgs), hipMemcpyHostToDevice, stream));
    HIP_OK(hipStreamSynchronize(stream));   // taps are a debug path: keep the host struct's lifetime simple
    dtaps = p->d_taps;
  }
  // Small arms (nv ≤ 8, hinge / slide joints, box limits) have two kernels of their own (plain solves without taps):
  //   * a 16-lane ROW per problem (quad_kernel.h): four problems per wavefront, so 4 096 problems put one wavefront on
  //     every SIMD and each problem still spreads its phases over its lanes.  Its fused loop below
  //     kLaneMinBatchLoop instances (threshold-terminated UR5e loop at 4 096: 22.4 M targets/s against 8.4 on the
  //     wavefront kernel and 6.0 on the lane kernel);
  //   * one LANE per problem (lane_kernel.h): 64 problems per wavefront with every lane busy, one ≈48 µs dependent
  //     instruction stream per problem whatever the batch — the best use of the machine once every SIMD has several
  //     wavefronts to interleave.  Also the fused caller loop (steps / until) from kLaneMinBatchLoop instances.
  // Measured on MI355X, UR5e config 2, M solves/s (tools/bench_small_arm.py; wavefront / row / lane kernel):
  //   B = 256: 12 / 25 / 7.6    4 096: 119 / 219 / 90    8 192: 130 / 449 / 152    32 768: 150 / 839 / 528
  //   65 536: 155 / 900 / 855    131 072: 158 / 1 169 / 1 609    1 048 576: 160 / 1 331 / 3 510
  // — the row kernel up to kLaneMinBatch, the lane kernel beyond.
  // MKH_FLAG_WAVE_KERNEL / _QUAD_KERNEL / _LANE_KERNEL force one of the three (parity switches).
  const bool small_ok = !taps && a.do_qp && !(flags & MKH_FLAG_WAVE_KERNEL);
  const bool small_arm = p->lane_nv && small_ok;                               // (nv ≤ 8: both kernels)
  const bool loop = a.n_steps > 1 || a.q_out || a.pos_threshold >= 0.0;       // fused caller loop (steps / until)
  // (9 … 16 dofs — hands, mobile arms: the row kernel with sixteen column registers, whatever the batch; there is no lane
  //  kernel of that size to hand over to)
  // (MKH_FLAG_WARM_START: the row kernel keeps its partition in the handle's warm-start buffer like the wavefront kernels)
  // (17 … 32 dofs or links, floating bases — H1, Go1, Spot, Allegro: the row kernel on TWO DPP rows per problem, two problems per
  //  wavefront; single solves only — the fused loops of such robots integrate a quaternion and stay on the wavefront kernel)
  if (p->quad_nt && small_ok && !(p->quad_nt == 32 && loop) && !((flags & MKH_FLAG_LANE_KERNEL) && p->lane_nv) &&
      (!p->lane_nv || a.B < (loop ? mkh::kLaneMinBatchLoop : mkh::kLaneMinBatch) || (flags & MKH_FLAG_QUAD_KERNEL))) {
    const int per_wave = p->quad_nt == 32 ? 2 : 4;
    const int grid = (a.B + per_wave - 1) / per_wave;
    p->last_grid = grid; p->last_nt = p->quad_nt;
    snprintf(p->last_kernel, sizeof(p->last_kernel), p->quad_nt == 8 ? (loop ? "ik_quad_kernel_loop" : "ik_quad_kernel")
                                                     : (p->quad_nt == 32 ? "ik_quad_kernel_32" : (loop ? "ik_quad_kernel_16_loop" : "ik_quad_kernel_16")));
    SolveArgs aq = a;
    HIP_OK(clk_begin(p, aq, stream));
    p->last_lds = mkh::launch_quad(p->quad_nt, loop, grid, stream, p->d_lane, p->lane_dims, aq);
    HIP_OK(hipGetLastError());
    HIP_OK(clk_end(p, a.B, stream));
    return MKH_OK;
  }
  if (small_arm && (a.B >= (loop ? mkh::kLaneMinBatchLoop : mkh::kLaneMinBatch) || (flags & MKH_FLAG_LANE_KERNEL))) {
    const int grid = (a.B + kWave - 1) / kWave;
    p->last_grid = grid; p->last_lds = p->lane_lds; p->last_nt = p->lane_nv;
    snprintf(p->last_kernel, sizeof(p->last_kernel), loop ? "ik_lane_kernel_%d_loop" : "ik_lane_kernel_%d", p->lane_nv);
    if (mkh::launch_lane(p->lane_nv, loop, grid, p->lane_lds, stream, p->d_lane, a) != 0)
      return fail(MKH_E_INVALID, "no kernel variant %s", p->last_kernel);
    HIP_OK(hipGetLastError());
    return MKH_OK;
  }
  // lean production variant unless the call needs a feature it leaves out
  int need = 0;
  if (taps) need |= F_TAPS;
  if (p->has_relative) need |= F_REL;
  if (p->dev.n_com > 0) need |= F_COM;
  if (p->dev.n_pairs > 0) need |= F_COLL;
  if (a.n_steps > 1 || a.q_out || a.pos_threshold >= 0.0) need |= F_STEPS;
  const bool dense = p->dev.n_dense_rows > 0 || p->dev.n_dense_limit_rows > 0 || p->dev.dense_box;
  if (dense) need |= 64;                                              // plugin rows: only the all-feature variants have them
  // (general convex pairs of a plain solve: their contacts come from convex_contacts_kernel, launched below in front of the
  //  ANALYTIC build — round 5; without the pre-pass buffers, the build with the routine inside)
  // Measured (ur5e_convex, one cylinder–box pair; in-kernel routine / split): 4 096 instances 0.165 / 0.207 ms, 16 384 0.463 / 0.505,
  // 65 536 1.270 / 0.970: GJK is one long dependent chain per lane — in front of a small batch its latency adds to the solve's, on
  // a large one 64 busy lanes per wavefront beat one busy lane per problem.  The split from 32 768 (instance, pair) items on.
  const bool cv_split = need == F_COLL && p->convex_pairs && p->d_cv != nullptr && (long long)a.B * p->cv.n_cv >= 32768;
  int feat;
  if (need == 0) feat = 0;
  else if (need == 64) feat = F_DENSE;                                // plugin rows next to frame / posture tasks and box limits: lean build
  else if (need == F_STEPS) feat = F_STEPS;
  else if (need == F_COLL) feat = p->simple_pairs ? (F_COLL | F_SIMPLE_COLL) : ((p->convex_pairs && !cv_split) ? (F_COLL | F_CONVEX_COLL) : F_COLL);
  // (fused loops over capsule-only collision sets — the Shadow hand's closed loop — have a build of their own: the
  //  all-feature one spills 534 VGPRs at this tableau size)
  else if (need == (F_COLL | F_STEPS) && p->simple_pairs) feat = F_COLL | F_SIMPLE_COLL | F_STEPS;
  else if ((need & ~(F_REL | F_COM)) == 0) feat = F_REL | F_COM;      // box limits only: keeps the block-pivoting active set
  else feat = (need & F_TAPS) ? F_ALL : (F_ALL & ~F_TAPS);
  const bool rich = (feat & (F_ALL & ~F_STEPS)) != 0;
  int nt = rich ? p->nt_full : p->nt, nr = 0, lds = rich ? p->lds_bytes_full : p->lds_bytes;
  // Builds whose register-hungry phases are real calls (ik_kernel.h MKH_CALLS: FEAT 8 / 136 / 30) do not need the 256-register
  // map of a 32-row tableau for the compiler's sake: a small arm keeps a small tableau (UR5e + 2 collision rows: phase 0 on
  // 32 rows was a third of the solve) — at least `calls_nt_min` rows, so that the callees still find a usable register file
  // Measured (UR5e + 2 collision rows, 4 096 instances): analytic pairs 0.060 ms on 32 rows, 0.049 on 16 and on 8; with a
  // cylinder–box pair on the general convex routine (round 4: simplex in LDS, 143 VGPRs) 0.186 ms on 32 rows, 0.156 on 16
  // and on 24 — round 3's register-resident GJK (196 VGPRs) wanted the 2-waves file and kept 32 rows.
  static const int dbg_min = dbg_env("MKH_DEBUG_CALLS_NT") ? atoi(dbg_env("MKH_DEBUG_CALLS_NT")) : 0;
  const int calls_nt_min = dbg_min ? dbg_min : 16;
  const bool calls = feat == F_COLL || feat == (F_ALL & ~F_TAPS) || feat == (F_COLL | F_CONVEX_COLL);
  if (calls && p->nt < 32) {
    static const int kV[] = {8, 16, 24, 32};
    for (int v : kV) if (v >= p->nt && v >= calls_nt_min) { nt = v; break; }
    lds = nt == p->nt ? p->lds_bytes : lds_for_nt(p, nt);
  }
  // Low-rank start when the problem qualifies and the diagonal part of H is not tiny against JwᵀJw
  // (error amplification of the quasi-definite elimination ≈ eps·max cost²/min Dg ≤ 1e-9).
  const double dg_min = a.damping + p->wood_min_diag;
  // (taps: only the cycle counters / pivot counts exist in the low-rank variant — profiling)
  const bool prof_only = taps && !taps->t_xpos && !taps->t_xquat && !taps->t_frame_pose && !taps->t_subtree_com &&
                         !taps->t_task_e && !taps->t_task_J && !taps->t_H && !taps->t_c && !taps->t_box_lo &&
                         !taps->t_box_hi && !taps->t_coll_G && !taps->t_coll_h;
  if (p->wood_nt && ((need & ~(F_STEPS | F_COM)) == 0 || (need == F_TAPS && prof_only)) && a.do_qp &&
      !(flags & MKH_FLAG_DIRECT_QP) && dg_min > 0.0 && dg_min >= 1e-7 * p->wood_max_cost2 &&
      !((need & F_TAPS) && ((need & F_COM) || p->wood_big))) {   // (no profiling build of the F_COM low-rank start: direct variant)
    nt = p->wood_nt; nr = p->wood_nr; lds = p->wood_lds_bytes;
    feat = F_WOOD | (need & (F_STEPS | F_TAPS | F_COM)) | (p->wood_big ? F_COM : 0);
  }
  // ... and, round 6, with half-space rows (`48_40_r48`: analytic collision pairs; mkh_problem_create "low-rank start WITH half-space
  // rows"): the launch that does the work of a plain collision solve — the tight-rows one where it exists, else the main one
  const bool woodr_ok = a.do_qp && !taps && feat == F_COLL && !(flags & MKH_FLAG_DIRECT_QP) && dg_min > 0.0 && dg_min >= 1e-7 * p->wood_max_cost2;
  if (woodr_ok && p->woodr_lds && !p->d_dev_tight && nt == 48) { nr = 48; lds = p->woodr_lds; feat = F_WOOD | F_COLL; }
  // three resident waves per SIMD where a variant exists (FrameTask / PostureTask / RelativeFrameTask / ComTask, box limits)
  static const int kW3Variants[][2] = {{44, 0}};   // (44_6 and 44_16 still spill 34–76 VGPRs at 74 registers: scratch traffic makes them slower than their 2-waves builds)
  bool w3 = false;
  if (!nr && p->lds_bytes_w3 && !(flags & MKH_FLAG_TWO_WAVES))
    for (const auto& v : kW3Variants) w3 = w3 || (v[0] == nt && v[1] == feat);
  if (w3) lds = p->lds_bytes_w3;
  if (nr && p->wood_lds_bytes_w3 && !(flags & MKH_FLAG_TWO_WAVES) &&
      (feat == F_WOOD || feat == (F_WOOD | F_STEPS) || ((feat == (F_WOOD | F_COM) || feat == (F_WOOD | F_COM | F_STEPS)) && nt == 44))) {
    w3 = true; lds = p->wood_lds_bytes_w3;
  }
  // tight rows first (see mkh_problem_create): plain solves on the capsule-only collision build whose caller takes the status
  const bool tight = p->d_dev_tight && (feat == (F_COLL | F_SIMPLE_COLL) || feat == F_COLL) && !nr && !w3 && a.do_qp && a.status_out && !taps &&
                     !(flags & MKH_FLAG_FULL_ROWS);
  // (the tight launch itself on the low-rank start: same descriptor, its own LDS layout)
  const bool tight_wood = tight && woodr_ok && p->woodr_lds_tight != 0;
  const int t_feat = tight_wood ? (feat | F_WOOD) : feat, t_nr = tight_wood ? 48 : 0, t_lds = tight_wood ? p->woodr_lds_tight : p->lds_tight;
  const bool no_redo = (p->diag & MKH_DIAG_NO_TIGHT_REDO) != 0;              // (tests: what the tight launch alone leaves flagged)
  char t_name[48];
  snprintf(t_name, sizeof t_name, t_nr ? "ik_solve_kernel_%d_%d_r%d" : "ik_solve_kernel_%d_%d", p->nt_tight, t_feat, t_nr);
  if (tight && no_redo) {
    const int gt = grid_for_variant(p, a.B, p->nt_tight, t_lds, false);
    SolveArgs at = a;
    at.work_counter = p->d_work;
    at.static_rounds = INT32_MAX;
    if (mkh::launch_variant(p->nt_tight, t_nr, t_feat, false, gt, t_lds, stream, p->d_dev_tight, at, nullptr) != 0)
      return fail(MKH_E_INVALID, "no kernel variant %s", t_name);
    HIP_OK(hipGetLastError());
    snprintf(p->last_kernel, sizeof(p->last_kernel), "%s", t_name);
    return MKH_OK;
  }
  if (tight) {
    const int gt = grid_for_variant(p, a.B, p->nt_tight, t_lds, false);
    SolveArgs at = a;
    at.work_counter = p->d_work;
    const int pw = a.B / gt;
    at.static_rounds = (pw >= kMinRoundsForTickets) ? static_rounds_for(pw, gt % p->model->num_cus == 0 && ((gt / p->model->num_cus) & 3) != 0) : INT32_MAX;
    HIP_OK(clk_begin(p, at, stream));                 // (clock builds: the stamps of the launch that does the work)
    if (mkh::launch_variant(p->nt_tight, t_nr, t_feat, false, gt, t_lds, stream, p->d_dev_tight, at, nullptr) != 0)
      return fail(MKH_E_INVALID, "no kernel variant %s", t_name);
    HIP_OK(hipGetLastError());
  }
  int grid = grid_for_variant(p, a.B, nt, lds, w3);
  // One problem per workgroup instead of persistent wavefronts, for the humanoid-size builds of the one-more-wave map between
  // 3.5 and 22 rounds (round 5): the hardware's workgroup dispatcher is a ticket counter that costs nothing, and a workgroup of
  // these builds starts cheaply (the phases that need registers are callees).  Kernel ms on G1 config 3, persistent /
  // one per workgroup: 12 288 instances 0.176 / 0.163, 16 384 0.224 / 0.209, 24 576 0.323 / 0.301, 32 768 0.411 / 0.390,
  // 49 152 0.588 / 0.572, 65 536 0.751 / 0.743; outside the range it loses (8 192: 0.111 / 0.117, 131 072: 1.414 / 1.424), and so
  // it does on every two-waves build (plugin workload 1.627 / 1.689, Shadow at 65 536 instances 1.186 / 1.226) and on the G1 full
  // example's 25.6 rounds (1.244 / 1.265; at 16 384 instances 0.392 / 0.349).  Two, three, four problems per workgroup lie
  // between the two shapes (65 536: 0.739 / 0.772 / 0.766).  The XCD still owns one contiguous row range
  // (workgroup g: XCD g % 8, row g / 8 of its range).  MKH_DEBUG_PERSISTENT=1: persistent wavefronts everywhere (A/B).
  static const bool persistent_only = dbg_env("MKH_DEBUG_PERSISTENT") != nullptr;
  // (fused loops, measured: no difference — 13.55 / 13.58 ms for the headline's 20-step loop, converged targets 5.64 / 5.58 M/s — they stay persistent)
  // Round 6: where the build has a twin compiled for this shape (`44_32_r44_w3o`, below) there is no upper end — kernel ms, persistent /
  // one per workgroup on the twin: 12 288 instances 0.166 / 0.151, 65 536 0.737 / 0.701, 73 728 0.817 / 0.786, 131 072 1.400 / 1.379,
  // 262 144 2.750 / 2.728 (below 3.5 rounds the persistent shape stays: 10 240 instances 0.145 on both).
  static const int os_max = dbg_env("MKH_DEBUG_ONE_SHOT_MAX") ? atoi(dbg_env("MKH_DEBUG_ONE_SHOT_MAX")) : 22;    // (A/B switch: upper end of the range, in rounds)
  const bool has_twin = w3 && nt == 44 && nr == 44 && feat == F_WOOD && !dtaps;
  if (!persistent_only && !tight && w3 && nt == 44 && a.n_steps <= 1 && a.B > grid && 2 * (long long)a.B >= 7LL * grid &&
      (has_twin || a.B <= (long long)os_max * grid))
    grid = a.B;
  // ... and, round 6, on a build WITHOUT the persistent loop's machinery where one exists (`44_32_r44_w3o`: build.py W3_WOOD_ONE_SHOT,
  // ik_kernel.h MKH_ONE_SHOT — 78 → 32 spilled SGPRs in the kernel body, headline 0.713 → 0.703 ms; the F_COM twin measured no gain)
  const bool one_shot = grid == a.B && has_twin;
  p->last_grid = grid; p->last_lds = lds; p->last_nt = nt;
  snprintf(p->last_kernel, sizeof(p->last_kernel), nr ? (w3 ? (one_shot ? "ik_solve_kernel_%d_%d_r%d_w3o" : "ik_solve_kernel_%d_%d_r%d_w3") : "ik_solve_kernel_%d_%d_r%d") : (w3 ? "ik_solve_kernel_%d_%d_w3" : "ik_solve_kernel_%d_%d"), nt, feat, nr);
  SolveArgs al = a;
  al.work_counter = p->d_work;
  if (tight) {
    al.redo_mask = MKH_ST_ROW_OVERFLOW;              // the full-row build: only what the tight launch flagged
    snprintf(p->last_kernel, sizeof(p->last_kernel), "%s+redo_%d", t_name, nt);
    p->last_nt = p->nt_tight; p->last_lds = t_lds; p->last_grid = grid_for_variant(p, a.B, p->nt_tight, t_lds, false);
  }
  // Distribution (ik_kernel.h): most of each wave's share is static — one contiguous row range per XCD — and the tail
  // of the batch goes through the ticket counter (how much: static_rounds_for above).  Round 1, on G1 (kernel ms by static
  // sixteenths): 16 → 1.283, 15 → 1.233, 14 → 1.183, 12 → 1.185, 8 → 1.193, 4 → 1.195, 0 → 1.264: every wave opening with an
  // atomic costs more than the balance returns.  Short problems (the 8-row variants) and thin batches stay static.
  const int per_wave = a.B / grid;
  // (fused loops: problems of very different length — the ticket tail from four rounds on, as before)
  const bool dynamic = nt > 8 && per_wave >= (a.n_steps > 1 ? 4 : kMinRoundsForTickets) && !tight;   // (a redo launch walks its static share, ik_kernel.h)
  al.static_rounds = dynamic ? static_rounds_for(per_wave, grid % p->model->num_cus == 0 && ((grid / p->model->num_cus) & 3) != 0, a.n_steps > 1) : INT32_MAX;
  if (!tight) HIP_OK(clk_begin(p, al, stream));
  if (cv_split) {
    const int rc = mkh::launch_convex_pre(stream, p->d_wide, p->cv, a.B, a.q, p->d_cv);
    if (rc != 0) return fail(MKH_E_HIP, "convex pre-pass: %s", hipGetErrorString((hipError_t)rc));
  }
  if (mkh::launch_variant(nt, nr, feat, w3, grid, lds, stream, p->d_dev, al, dtaps, one_shot) != 0)
    return fail(MKH_E_INVALID, "no kernel variant %s", p->last_kernel);
  HIP_OK(hipGetLastError());
  HIP_OK(clk_end(p, a.B, stream));
  if (cv_split) {                                          // ("convex_pre+": the pre-pass launch in front)
    char tmp[64];
    snprintf(tmp, sizeof tmp, "convex_pre+%s", p->last_kernel);
    snprintf(p->last_kernel, sizeof(p->last_kernel), "%s", tmp);
  }
  if (wide_redo) {
    const size_t len = strlen(p->last_kernel);
    snprintf(p->last_kernel + len, sizeof(p->last_kernel) - len, "+wide");
    SolveArgs ar = a;
    ar.q = q_redo;
    HIP_OK(clk_begin(p, ar, stream));                      // (MKH_DEBUG_CLOCKS: the stamps of the redo launch overwrite the main kernel's)
    // (… and the instances whose active rows the sweep tableau found almost dependent, or on which it failed: MKH_ST_DEGENERATE is
    //  internal — the dense Goldfarb–Idnani iteration of the workgroup-per-problem kernel answers for them)
    const int32_t rcr = launch_wide_kernel(p, ar, stream, MKH_ST_ROW_OVERFLOW | MKH_ST_INFEASIBLE | MKH_ST_ITER_LIMIT | 32, dtaps);
    if (rcr == MKH_OK) HIP_OK(clk_end(p, a.B, stream));
    return rcr;
  }
  return MKH_OK;
}

struct TapBuf {
  void* host; void* dev; size_t bytes;
};

static int32_t run(MkhProblem* p, int32_t B, const double* q, const double* frame_targets, const double* posture_target,
                   const double* com_target, double dt, double damping, double* v_out, int32_t* status_out,
                   const MkhTaps* taps, int32_t flags, void* hip_stream, int32_t n_steps, double* q_out,
                   const MkhDenseRows* dense = nullptr, double pos_threshold = -1.0, double ori_threshold = -1.0,
                   int32_t* iters_out = nullptr, int32_t* converged_out = nullptr) {
  if (!p) return fail(MKH_E_INVALID, "null problem");
  if (B < 1) return fail(MKH_E_INVALID, "B must be >= 1");
  const DeviceProblem& P = p->dev;
  if (!q) return fail(MKH_E_INVALID, "q is null");
  if (P.n_frame > 0 && !frame_targets) return fail(MKH_E_INVALID, "frame_targets is null (TargetNotSet)");
  if (P.n_posture > 0 && !posture_target) return fail(MKH_E_INVALID, "posture_target is null (TargetNotSet)");
  if (P.n_com > 0 && !com_target) return fail(MKH_E_INVALID, "com_target is null (TargetNotSet)");
  if (!(dt > 0.0)) return fail(MKH_E_INVALID, "dt must be > 0");
  if (n_steps < 1) return fail(MKH_E_INVALID, "n_steps must be >= 1");
  const size_t Kd = P.n_dense_rows, Md = P.n_dense_limit_rows;
  const bool Bd = P.dense_box != 0;
  if (!Bd && dense && (dense->limit_lo || dense->limit_hi))
    return fail(MKH_E_INVALID, "limit_lo / limit_hi need a problem created with dense_limit_box = 1");
  if (Kd || Md || Bd) {
    if (!dense) return fail(MKH_E_INVALID, "this problem has dense (plugin) rows: call mkh_solve_dense");
    if (Kd && (!dense->task_e || !dense->task_J)) return fail(MKH_E_INVALID, "dense task_e / task_J is null");
    if (Md && (!dense->limit_G || !dense->limit_h)) return fail(MKH_E_INVALID, "dense limit_G / limit_h is null");
    if (n_steps > 1 || q_out) return fail(MKH_E_INVALID, "dense (plugin) rows are evaluated by the caller at q: no fused steps");
  }
  // max_batch bounds everything the handle owns per instance — staging buffers, the warm-start active sets — whichever
  // kind of pointer the caller passes (a device-pointer warm start used to write past d_warm here)
  if (B > p->max_batch) return fail(MKH_E_INVALID, "B=%d exceeds max_batch=%d of this problem", B, p->max_batch);
  HIP_OK(hipSetDevice(p->model->device));
  hipStream_t stream = (hipStream_t)hip_stream;
  const bool devp = (flags & MKH_FLAG_DEVICE_PTRS) != 0;
  const bool pbat = (flags & MKH_FLAG_POSTURE_BATCHED) != 0, cbat = (flags & MKH_FLAG_COM_BATCHED) != 0;
  SolveArgs a;
  memset(&a, 0, sizeof a);
  TapArgs t;
  memset(&t, 0, sizeof t);
  bool any_tap = false;
  a.B = B; a.posture_batched = pbat; a.com_batched = cbat; a.do_qp = (v_out != nullptr);
  a.dt = dt; a.damping = damping; a.n_steps = n_steps;
  a.pos_threshold = pos_threshold; a.ori_threshold = ori_threshold;
  if (flags & MKH_FLAG_WARM_START) {
    // closed-loop callers solve the same instances again and again: keep each instance's active set on the device
    if (!p->d_warm) HIP_OK(hipMalloc((void**)&p->d_warm, (size_t)p->max_batch * P.nv));
    if (p->warm_B != B) { HIP_OK(hipMemsetAsync(p->d_warm, 0, (size_t)p->max_batch * P.nv, stream)); p->warm_B = B; p->warm_age = 0; }
    a.warm = p->d_warm; a.warm_age = p->warm_age;
    if (v_out) ++p->warm_age;
  }
  const bool until = pos_threshold >= 0.0;
  if (until && P.n_frame < 1) return fail(MKH_E_INVALID, "mkh_solve_until needs at least one frame task to test the thresholds on");
  const size_t nq = P.nq, nv = P.nv;
  const size_t n_pt = (size_t)P.n_posture * nq * (pbat ? B : 1), n_ct = (size_t)P.n_com * 3 * (cbat ? B : 1);
  std::vector<TapBuf> tb;
  if (devp) {
    a.q = q; a.frame_targets = frame_targets; a.posture_target = posture_target; a.com_target = com_target;
    a.v_out = v_out; a.status_out = status_out; a.q_out = q_out;
    a.iters_out = iters_out; a.converged_out = converged_out;
    if (Kd) { a.dense_e = dense->task_e; a.dense_J = dense->task_J; }
    if (Md) { a.dense_G = dense->limit_G; a.dense_h = dense->limit_h; }
    if (Bd) { a.dense_lo = dense->limit_lo; a.dense_hi = dense->limit_hi; }
    if (taps) {
      t.t_xpos = taps->xpos; t.t_xquat = taps->xquat; t.t_frame_pose = taps->frame_pose;
      t.t_subtree_com = taps->subtree_com; t.t_task_e = taps->task_e; t.t_task_J = taps->task_J; t.t_H = taps->H;
      t.t_c = taps->c; t.t_box_lo = taps->box_lo; t.t_box_hi = taps->box_hi; t.t_coll_G = taps->coll_G;
      t.t_coll_h = taps->coll_h; t.t_qp_iters = taps->qp_iters; t.t_cycles = (long long*)taps->cycles;
      if (t.t_coll_G) HIP_OK(hipMemsetAsync(t.t_coll_G, 0, (size_t)B * P.n_pairs * nv * sizeof(double), stream));
    }
    return launch(p, a, taps ? &t : nullptr, stream, flags);
  }
  // ---- host pointers, small call: everything through the pinned scratch (see PinnedScratch) — inputs, outputs and taps
  {
    const size_t Bz = B;
    const size_t d_q = Bz * nq, d_ft = Bz * P.n_frame * 7, d_v = v_out ? Bz * nv : 0;
    size_t bytes = (d_q + d_ft + n_pt + n_ct + d_v) * sizeof(double) + ((Bz * (until ? 3 : 1) * sizeof(int32_t) + 7) & ~(size_t)7);
    if (taps) {
      TapArgs sizes;
      assign_taps(taps, P, Bz, sizes, [&](void* host, size_t n, bool) -> void* { if (host) bytes += (n + 7) & ~(size_t)7; return nullptr; });
    }
    if (!Kd && !Md && !Bd && bytes <= kSmallCallBytes) {
      HIP_OK(p->small.ensure());
      char* const h = p->small.host;
      char* const d = p->small.dev;
      size_t o = 0;                                    // (bytes; every block a multiple of 8)
      auto put = [&](const double* src, size_t n) -> const double* {
        if (!n) return nullptr;
        memcpy(h + o, src, n * sizeof(double));
        const double* r = (const double*)(d + o);
        o += n * sizeof(double);
        return r;
      };
      a.q = put(q, d_q);
      a.frame_targets = put(frame_targets, d_ft);
      a.posture_target = put(posture_target, n_pt);
      a.com_target = put(com_target, n_ct);
      const size_t o_v = o;
      a.v_out = v_out ? (double*)(d + o_v) : nullptr;
      o += d_v * sizeof(double);
      const size_t o_i = o;
      int32_t* const di32 = (int32_t*)(d + o_i);
      a.status_out = di32;
      a.q_out = q_out ? (double*)a.q : nullptr;        // in place, as in the staged path
      if (until) { a.iters_out = di32 + Bz; a.converged_out = di32 + 2 * Bz; }
      o += (Bz * (until ? 3 : 1) * sizeof(int32_t) + 7) & ~(size_t)7;
      struct Back { void* host; size_t off, n; };
      std::vector<Back> back;
      if (taps)
        assign_taps(taps, P, Bz, t, [&](void* host, size_t n, bool zero) -> void* {
          if (!host) return nullptr;
          if (zero) memset(h + o, 0, n);
          back.push_back({host, o, n});
          void* r = d + o;
          o += (n + 7) & ~(size_t)7;
          return r;
        });
      const int32_t rc = launch(p, a, taps ? &t : nullptr, stream, flags);
      if (rc != MKH_OK) return rc;
      HIP_OK(hipStreamSynchronize(stream));
      const int32_t* const hi32 = (const int32_t*)(h + o_i);
      if (v_out) memcpy(v_out, h + o_v, d_v * sizeof(double));
      if (q_out) memcpy(q_out, h, d_q * sizeof(double));
      if (status_out && v_out) memcpy(status_out, hi32, Bz * sizeof(int32_t));
      if (until && iters_out) memcpy(iters_out, hi32 + Bz, Bz * sizeof(int32_t));
      if (until && converged_out) memcpy(converged_out, hi32 + 2 * Bz, Bz * sizeof(int32_t));
      for (const Back& b : back) memcpy(b.host, h + b.off, b.n);
      return MKH_OK;
    }
  }
  // ---- host pointers: stage through library-owned device buffers
  const size_t mb = p->max_batch;
  HIP_OK(ensure(&p->s_q, mb * nq));
  HIP_OK(ensure(&p->s_ft, mb * P.n_frame * 7));
  HIP_OK(ensure(&p->s_v, mb * nv));
  HIP_OK(ensure(&p->s_status, mb));
  if (n_pt > p->s_pt_cap) { (void)hipFree(p->s_pt); p->s_pt = nullptr; HIP_OK(hipMalloc((void**)&p->s_pt, n_pt * sizeof(double))); p->s_pt_cap = n_pt; }
  if (n_ct > p->s_ct_cap) { (void)hipFree(p->s_ct); p->s_ct = nullptr; HIP_OK(hipMalloc((void**)&p->s_ct, n_ct * sizeof(double))); p->s_ct_cap = n_ct; }
  // Large plain solves in chunks: the host → device copy of chunk c + 1 and the device → host copy of chunk c − 1 run on
  // streams of their own beside the kernel of chunk c (the single-shot sequence copy in, solve, copy out left the GPU idle
  // for more than half of a 65 536-instance G1 call).  Chunks of at least 8 192 instances; each is dispatched by its own
  // size (a small arm's chunk may run the row kernel where the whole batch would have taken the lane kernel: same optimum).
  // Only where the copies are worth overlapping (≥ 32 MB staged: tools/bench_host_path.py — G1 at 65 536 instances
  // 2.02 → 1.72 ms per call, the G1 full example 2.79 → 2.19 ms; a 65 536-instance UR5e call moves 10 MB around a 0.08 ms
  // kernel and LOSES 0.15 ms to the events and the three extra launches).
  static const bool no_chunks = dbg_env("MKH_DEBUG_NO_CHUNKS") != nullptr;      // (A/B of this path)
  const size_t staged_bytes = (size_t)B * sizeof(double) *
      (nq + (size_t)P.n_frame * 7 + nv + (pbat ? (size_t)P.n_posture * nq : 0) + (cbat ? (size_t)P.n_com * 3 : 0) + (q_out ? nq : 0));
  const int n_chunks = (!no_chunks && !taps && !Kd && !Md && !Bd && v_out && B >= 2 * 8192 && staged_bytes >= ((size_t)32 << 20))
                           ? (B / 8192 < 4 ? B / 8192 : 4) : 1;
  if (n_chunks > 1) {
    if (!p->st_in) {
      HIP_OK(hipStreamCreateWithFlags(&p->st_in, hipStreamNonBlocking));
      HIP_OK(hipStreamCreateWithFlags(&p->st_out, hipStreamNonBlocking));
      HIP_OK(hipEventCreateWithFlags(&p->ev_start, hipEventDisableTiming));
      for (int i = 0; i < 4; ++i) {
        HIP_OK(hipEventCreateWithFlags(&p->ev_in[i], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&p->ev_k[i], hipEventDisableTiming));
      }
    }
    if (until) HIP_OK(ensure(&p->s_iters, mb * 2));
    // (shared targets on the caller's stream, ahead of the first kernel; the side streams start behind the caller's work)
    if (n_pt && !pbat) HIP_OK(hipMemcpyAsync(p->s_pt, posture_target, n_pt * sizeof(double), hipMemcpyHostToDevice, stream));
    if (n_ct && !cbat) HIP_OK(hipMemcpyAsync(p->s_ct, com_target, n_ct * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_OK(hipEventRecord(p->ev_start, stream));
    HIP_OK(hipStreamWaitEvent(p->st_in, p->ev_start, 0));
    const size_t chunk = ((size_t)B / n_chunks) / 64 * 64;          // ≥ 8 192; the last chunk takes the remainder
    const size_t ft_w = (size_t)P.n_frame * 7, pt_w = (size_t)P.n_posture * nq, ct_w = (size_t)P.n_com * 3;
    int32_t rc = MKH_OK;
    hipError_t e = hipSuccess;
    // (an error inside the loop must not return: copies that read the caller's q / targets may be in flight on st_in)
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    for (int c = 0; c < n_chunks && rc == MKH_OK && e == hipSuccess; ++c) {
      const size_t off = c * chunk, Bc = (c + 1 < n_chunks) ? chunk : (size_t)B - off;
      ok(hipMemcpyAsync(p->s_q + off * nq, q + off * nq, Bc * nq * sizeof(double), hipMemcpyHostToDevice, p->st_in));
      if (ft_w) ok(hipMemcpyAsync(p->s_ft + off * ft_w, frame_targets + off * ft_w, Bc * ft_w * sizeof(double), hipMemcpyHostToDevice, p->st_in));
      if (n_pt && pbat) ok(hipMemcpyAsync(p->s_pt + off * pt_w, posture_target + off * pt_w, Bc * pt_w * sizeof(double), hipMemcpyHostToDevice, p->st_in));
      if (n_ct && cbat) ok(hipMemcpyAsync(p->s_ct + off * ct_w, com_target + off * ct_w, Bc * ct_w * sizeof(double), hipMemcpyHostToDevice, p->st_in));
      ok(hipEventRecord(p->ev_in[c], p->st_in));
      if (!ok(hipStreamWaitEvent(stream, p->ev_in[c], 0))) break;
      SolveArgs ac = a;
      ac.B = (int32_t)Bc;
      ac.q = p->s_q + off * nq; ac.frame_targets = p->s_ft + off * ft_w;
      ac.posture_target = p->s_pt + (pbat ? off * pt_w : 0); ac.com_target = p->s_ct + (cbat ? off * ct_w : 0);
      ac.v_out = p->s_v + off * nv; ac.status_out = p->s_status + off;
      ac.q_out = q_out ? p->s_q + off * nq : nullptr;
      if (until) { ac.iters_out = p->s_iters + off; ac.converged_out = p->s_iters + mb + off; }
      if (a.warm) ac.warm = a.warm + off * nv;
      rc = launch(p, ac, nullptr, stream, flags);
      if (rc == MKH_OK) ok(hipEventRecord(p->ev_k[c], stream));
    }
    for (int c = 0; c < n_chunks && rc == MKH_OK && e == hipSuccess; ++c) {
      const size_t off = c * chunk, Bc = (c + 1 < n_chunks) ? chunk : (size_t)B - off;
      e = hipStreamWaitEvent(p->st_out, p->ev_k[c], 0);
      if (e == hipSuccess) e = hipMemcpyAsync(v_out + off * nv, p->s_v + off * nv, Bc * nv * sizeof(double), hipMemcpyDeviceToHost, p->st_out);
      if (e == hipSuccess && q_out) e = hipMemcpyAsync(q_out + off * nq, p->s_q + off * nq, Bc * nq * sizeof(double), hipMemcpyDeviceToHost, p->st_out);
      if (e == hipSuccess && status_out) e = hipMemcpyAsync(status_out + off, p->s_status + off, Bc * sizeof(int32_t), hipMemcpyDeviceToHost, p->st_out);
      if (e == hipSuccess && until && iters_out) e = hipMemcpyAsync(iters_out + off, p->s_iters + off, Bc * sizeof(int32_t), hipMemcpyDeviceToHost, p->st_out);
      if (e == hipSuccess && until && converged_out) e = hipMemcpyAsync(converged_out + off, p->s_iters + mb + off, Bc * sizeof(int32_t), hipMemcpyDeviceToHost, p->st_out);
    }
    // (a failed call still drains what it started: the staging buffers belong to the handle)
    const hipError_t e1 = hipStreamSynchronize(p->st_in), e2 = hipStreamSynchronize(stream), e3 = hipStreamSynchronize(p->st_out);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3);
    if (rc == MKH_OK && e != hipSuccess) rc = fail(MKH_E_HIP, "solve: %s", hipGetErrorString(e));
    return rc;
  }
  HIP_OK(hipMemcpyAsync(p->s_q, q, (size_t)B * nq * sizeof(double), hipMemcpyHostToDevice, stream));
  if (P.n_frame) HIP_OK(hipMemcpyAsync(p->s_ft, frame_targets, (size_t)B * P.n_frame * 7 * sizeof(double), hipMemcpyHostToDevice, stream));
  if (n_pt) HIP_OK(hipMemcpyAsync(p->s_pt, posture_target, n_pt * sizeof(double), hipMemcpyHostToDevice, stream));
  if (n_ct) HIP_OK(hipMemcpyAsync(p->s_ct, com_target, n_ct * sizeof(double), hipMemcpyHostToDevice, stream));
  if (Kd) {
    HIP_OK(ensure(&p->s_de, mb * Kd));
    HIP_OK(ensure(&p->s_dJ, mb * Kd * nv));
    HIP_OK(hipMemcpyAsync(p->s_de, dense->task_e, (size_t)B * Kd * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(p->s_dJ, dense->task_J, (size_t)B * Kd * nv * sizeof(double), hipMemcpyHostToDevice, stream));
    a.dense_e = p->s_de; a.dense_J = p->s_dJ;
  }
  if (Md) {
    HIP_OK(ensure(&p->s_dG, mb * Md * nv));
    HIP_OK(ensure(&p->s_dh, mb * Md));
    HIP_OK(hipMemcpyAsync(p->s_dG, dense->limit_G, (size_t)B * Md * nv * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(p->s_dh, dense->limit_h, (size_t)B * Md * sizeof(double), hipMemcpyHostToDevice, stream));
    a.dense_G = p->s_dG; a.dense_h = p->s_dh;
  }
  if (Bd && (dense->limit_lo || dense->limit_hi)) {
    HIP_OK(ensure(&p->s_dbox, mb * nv * 2));
    if (dense->limit_lo) { HIP_OK(hipMemcpyAsync(p->s_dbox, dense->limit_lo, (size_t)B * nv * sizeof(double), hipMemcpyHostToDevice, stream)); a.dense_lo = p->s_dbox; }
    if (dense->limit_hi) { HIP_OK(hipMemcpyAsync(p->s_dbox + mb * nv, dense->limit_hi, (size_t)B * nv * sizeof(double), hipMemcpyHostToDevice, stream)); a.dense_hi = p->s_dbox + mb * nv; }
  }
  a.q = p->s_q; a.frame_targets = p->s_ft; a.posture_target = p->s_pt; a.com_target = p->s_ct;
  a.v_out = v_out ? p->s_v : nullptr;
  a.status_out = p->s_status;
  a.q_out = q_out ? p->s_q : nullptr;               // in place in the staging buffer
  if (until) {
    HIP_OK(ensure(&p->s_iters, mb * 2));
    a.iters_out = p->s_iters; a.converged_out = p->s_iters + mb;
  }
  int32_t rc = MKH_OK;
  auto tap = [&](void* host, size_t bytes, bool zero) -> void* {
    if (!host || rc != MKH_OK) return nullptr;
    void* dptr = nullptr;
    if (hipMalloc(&dptr, bytes ? bytes : 1) != hipSuccess) { rc = fail(MKH_E_HIP, "tap buffer allocation failed"); return nullptr; }
    tb.push_back({host, dptr, bytes});               // (freed at the end of the call whatever happens next)
    if (zero && hipMemsetAsync(dptr, 0, bytes, stream) != hipSuccess) { rc = fail(MKH_E_HIP, "tap buffer clear failed"); return nullptr; }
    return dptr;
  };
  if (taps) assign_taps(taps, P, (size_t)B, t, tap);
  if (rc == MKH_OK) rc = launch(p, a, taps ? &t : nullptr, stream, flags);
  if (rc == MKH_OK) {
    hipError_t e = hipSuccess;
    if (v_out) e = hipMemcpyAsync(v_out, p->s_v, (size_t)B * nv * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && q_out) e = hipMemcpyAsync(q_out, p->s_q, (size_t)B * nq * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && status_out && v_out)
      e = hipMemcpyAsync(status_out, p->s_status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && until && iters_out)
      e = hipMemcpyAsync(iters_out, p->s_iters, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && until && converged_out)
      e = hipMemcpyAsync(converged_out, p->s_iters + mb, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, stream);
    for (auto& t : tb)
      if (e == hipSuccess) e = hipMemcpyAsync(t.host, t.dev, t.bytes, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) rc = fail(MKH_E_HIP, "solve: %s", hipGetErrorString(e));
  }
  for (auto& t : tb) (void)hipFree(t.dev);
  return rc;
}

int32_t mkh_eval(MkhProblem* p, int32_t B, const double* q, const double* frame_targets, const double* posture_target,
                 const double* com_target, double dt, double damping, double* v_out, int32_t* status_out,
                 const MkhTaps* taps, int32_t flags, void* hip_stream) {
  return run(p, B, q, frame_targets, posture_target, com_target, dt, damping, v_out, status_out, taps, flags, hip_stream,
             1, nullptr);
}

int32_t mkh_solve(MkhProblem* p, int32_t B, const double* q, const double* frame_targets, const double* posture_target,
                  const double* com_target, double dt, double damping, double* v_out, int32_t* status_out,
                  int32_t flags, void* hip_stream) {
  if (!v_out) return fail(MKH_E_INVALID, "v_out is null");
  return run(p, B, q, frame_targets, posture_target, com_target, dt, damping, v_out, status_out, nullptr, flags,
             hip_stream, 1, nullptr);
}

int32_t mkh_solve_dense(MkhProblem* p, int32_t B, const double* q, const double* frame_targets,
                        const double* posture_target, const double* com_target, const MkhDenseRows* dense, double dt,
                        double damping, double* v_out, int32_t* status_out, const MkhTaps* taps, int32_t flags,
                        void* hip_stream) {
  return run(p, B, q, frame_targets, posture_target, com_target, dt, damping, v_out, status_out, taps, flags, hip_stream,
             1, nullptr, dense);
}

int32_t mkh_solve_steps(MkhProblem* p, int32_t B, const double* q, const double* frame_targets,
                        const double* posture_target, const double* com_target, double dt, double damping,
                        int32_t n_steps, double* q_out, double* v_out, int32_t* status_out, int32_t flags,
                        void* hip_stream) {
  if (!v_out || !q_out) return fail(MKH_E_INVALID, "v_out / q_out is null");
  return run(p, B, q, frame_targets, posture_target, com_target, dt, damping, v_out, status_out, nullptr, flags,
             hip_stream, n_steps, q_out);
}

int32_t mkh_solve_until(MkhProblem* p, int32_t B, const double* q, const double* frame_targets,
                        const double* posture_target, const double* com_target, double dt, double damping,
                        int32_t max_iters, double pos_threshold, double ori_threshold, double* q_out, double* v_out,
                        int32_t* status_out, int32_t* iters_out, int32_t* converged_out, int32_t flags, void* hip_stream) {
  if (!v_out || !q_out) return fail(MKH_E_INVALID, "v_out / q_out is null");
  if (!(pos_threshold >= 0.0) || !(ori_threshold >= 0.0)) return fail(MKH_E_INVALID, "thresholds must be >= 0");
  return run(p, B, q, frame_targets, posture_target, com_target, dt, damping, v_out, status_out, nullptr, flags,
             hip_stream, max_iters, q_out, nullptr, pos_threshold, ori_threshold, iters_out, converged_out);
}

int32_t mkh_integrate(MkhModel* m, int32_t B, const double* q, const double* v, double dt, double* q_out,
                      int32_t flags, void* hip_stream) {
  if (!m || !q || !v || !q_out) return fail(MKH_E_INVALID, "null argument");
  if (B < 1) return fail(MKH_E_INVALID, "B must be >= 1");
  HIP_OK(hipSetDevice(m->device));
  hipStream_t stream = (hipStream_t)hip_stream;
  DeviceProblem P;
  memset(&P, 0, sizeof P);
  fill_base(m, P);
  const long long total = (long long)B * m->njnt;
  const int block = 256;
  const int grid = (int)((total + block - 1) / block);
  if (flags & MKH_FLAG_DEVICE_PTRS) {
    hipLaunchKernelGGL(integrate_kernel, dim3(grid), dim3(block), 0, stream, P, B, q, v, dt, q_out);
    HIP_OK(hipGetLastError());
    return MKH_OK;
  }
  const size_t bq = (size_t)B * m->nq * 8, bv = (size_t)B * m->nv * 8;
  if (2 * bq + bv <= kSmallCallBytes) {               // small call: through the pinned scratch (see PinnedScratch)
    std::lock_guard<std::mutex> lock(m->small_mutex);   // (threads of one process share the model: round-3 advisor finding)
    HIP_OK(m->small.ensure());
    char* const h = m->small.host;
    char* const d = m->small.dev;
    memcpy(h, q, bq);
    memcpy(h + bq, v, bv);
    hipLaunchKernelGGL(integrate_kernel, dim3(grid), dim3(block), 0, stream, P, B, (const double*)d, (const double*)(d + bq), dt,
                       (double*)(d + bq + bv));
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(stream));
    memcpy(q_out, h + bq + bv, bq);
    return MKH_OK;
  }
  double *dq = nullptr, *dv = nullptr, *dout = nullptr;
  HIP_OK(hipMalloc((void**)&dq, bq));
  hipError_t e = hipMalloc((void**)&dv, bv);
  if (e == hipSuccess) e = hipMalloc((void**)&dout, bq);
  if (e == hipSuccess) e = hipMemcpyAsync(dq, q, bq, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dv, v, bv, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(integrate_kernel, dim3(grid), dim3(block), 0, stream, P, B, dq, dv, dt, dout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(q_out, dout, bq, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dq); (void)hipFree(dv); (void)hipFree(dout);
  if (e != hipSuccess) return fail(MKH_E_HIP, "integrate: %s", hipGetErrorString(e));
  return MKH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ mkh_lie_eval
namespace {
__device__ void assemble_ljacinv(V3 v, V3 w, double* o) {
  double J[9], Q[9];
  bool ident;
  se3_ljacinv(v, w, J, Q, ident);
  // [[J, −J·Q·J],[0, J]]  (se3.py:217-218; identity when θ² < 1e-10)
  double JQ[9], B[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) JQ[3 * i + j] = J[3 * i] * Q[j] + J[3 * i + 1] * Q[3 + j] + J[3 * i + 2] * Q[6 + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) B[3 * i + j] = -(JQ[3 * i] * J[j] + JQ[3 * i + 1] * J[3 + j] + JQ[3 * i + 2] * J[6 + j]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      o[6 * i + j] = J[3 * i + j];
      o[6 * i + 3 + j] = ident ? 0.0 : B[3 * i + j];
      o[6 * (i + 3) + j] = 0.0;
      o[6 * (i + 3) + 3 + j] = J[3 * i + j];
    }
}
__device__ SE3 load_se3(const double* p) { return SE3{Q4{p[0], p[1], p[2], p[3]}, V3{p[4], p[5], p[6]}}; }
__device__ void store_se3(double* o, SE3 T) {
  o[0] = T.q.w; o[1] = T.q.x; o[2] = T.q.y; o[3] = T.q.z; o[4] = T.p.x; o[5] = T.p.y; o[6] = T.p.z;
}
__global__ __launch_bounds__(64) void lie_eval_kernel(int op, int n, const double* __restrict__ a,
                                                      const double* __restrict__ b, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 v, w;
  switch (op) {
    case MKH_LIE_SE3_LOG: {
      se3_log(load_se3(a + 7 * i), v, w);
      double* o = out + 6 * i;
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = w.x; o[4] = w.y; o[5] = w.z;
    } break;
    case MKH_LIE_SE3_JLOG: {
      se3_log(load_se3(a + 7 * i), v, w);
      assemble_ljacinv(-1.0 * v, -1.0 * w, out + 36 * i);      // jlog(T) = ljacinv(−log T)
    } break;
    case MKH_LIE_SE3_LJACINV: {
      const double* t = a + 6 * i;
      assemble_ljacinv(V3{t[0], t[1], t[2]}, V3{t[3], t[4], t[5]}, out + 36 * i);
    } break;
    case MKH_LIE_SE3_MULTIPLY: store_se3(out + 7 * i, se3_mul(load_se3(a + 7 * i), load_se3(b + 7 * i))); break;
    case MKH_LIE_SE3_INVERSE: store_se3(out + 7 * i, se3_inv(load_se3(a + 7 * i))); break;
    case MKH_LIE_SE3_RMINUS: {
      se3_log(se3_mul(se3_inv(load_se3(b + 7 * i)), load_se3(a + 7 * i)), v, w);
      double* o = out + 6 * i;
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = w.x; o[4] = w.y; o[5] = w.z;
    } break;
    case MKH_LIE_SO3_LOG: {
      const double* q = a + 4 * i;
      w = so3_log(Q4{q[0], q[1], q[2], q[3]});
      out[3 * i] = w.x; out[3 * i + 1] = w.y; out[3 * i + 2] = w.z;
    } break;
    case MKH_LIE_SO3_MATRIX: {
      const double* q = a + 4 * i;
      const M3 R = qmat(Q4{q[0], q[1], q[2], q[3]});
      for (int k = 0; k < 9; ++k) out[9 * i + k] = R.m[k];
    } break;
    case MKH_LIE_SE3_APPLY: {
      const SE3 T = load_se3(a + 7 * i);
      const V3 r = qrot(T.q, V3{b[3 * i], b[3 * i + 1], b[3 * i + 2]}) + T.p;
      out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    } break;
  }
}
}  // namespace

extern "C" int32_t mkh_lie_eval(int32_t device, int32_t op, int32_t n, const double* a, const double* b, double* out,
                                int32_t flags, void* hip_stream) {
  static const int in_a[] = {7, 7, 6, 7, 7, 7, 4, 4, 7}, in_b[] = {0, 0, 0, 7, 0, 7, 0, 0, 3},
                   n_out[] = {6, 36, 36, 7, 7, 6, 3, 9, 3};
  if (op < 0 || op > MKH_LIE_SE3_APPLY) return fail(MKH_E_INVALID, "mkh_lie_eval: unknown op %d", op);
  if (n < 1 || !a || !out || (in_b[op] && !b)) return fail(MKH_E_INVALID, "mkh_lie_eval: null argument or n < 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MKH_E_NOGPU, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MKH_E_INVALID, "device %d out of range (%d visible)", device, ndev);
  HIP_OK(hipSetDevice(device));
  hipStream_t stream = (hipStream_t)hip_stream;
  const int grid = (n + 63) / 64;
  if (flags & MKH_FLAG_DEVICE_PTRS) {
    hipLaunchKernelGGL(lie_eval_kernel, dim3(grid), dim3(64), 0, stream, op, n, a, b, out);
    HIP_OK(hipGetLastError());
    return MKH_OK;
  }
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t ba = (size_t)n * in_a[op] * 8, bb = (size_t)n * in_b[op] * 8, bo = (size_t)n * n_out[op] * 8;
  hipError_t e = hipMalloc((void**)&da, ba);
  if (e == hipSuccess) e = hipMalloc((void**)&db, bb ? bb : 8);
  if (e == hipSuccess) e = hipMalloc((void**)&dout, bo);
  if (e == hipSuccess) e = hipMemcpyAsync(da, a, ba, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess && bb) e = hipMemcpyAsync(db, b, bb, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lie_eval_kernel, dim3(grid), dim3(64), 0, stream, op, n, da, db, dout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, bo, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  if (e != hipSuccess) return fail(MKH_E_HIP, "lie_eval: %s", hipGetErrorString(e));
  return MKH_OK;
}

// ------------------------------------------------------------------ mkh_geom_distance_eval
namespace {
// One geom pair per lane through the device's geom_distance (collide_dev.h / convex_dev.h) — the routine behind the
// collision rows — with the wave-level second half for general convex pairs whose cores overlap.  A pair is 22 doubles: two
// records of (type, size[3], pos[3], quat[4] wxyz).
__global__ __launch_bounds__(64) void geom_distance_eval_kernel(int n, const double* __restrict__ g, double distmax,
                                                                double* __restrict__ dist_out, double* __restrict__ fromto_out) {
  __shared__ __attribute__((aligned(16))) double ws[kEpaWsDoubles > 2 * kGjkWsDoubles ? kEpaWsDoubles : 2 * kGjkWsDoubles];
  const int lane = lane_id();
  const int base = blockIdx.x * 64;
  auto load = [&](int i, int& t1, V3& s1, V3& p1, Q4& q1, int& t2, V3& s2, V3& p2, Q4& q2) {
    const double* r = g + 22 * (size_t)i;
    t1 = (int)r[0]; s1 = {r[1], r[2], r[3]}; p1 = {r[4], r[5], r[6]}; q1 = {r[7], r[8], r[9], r[10]};
    t2 = (int)r[11]; s2 = {r[12], r[13], r[14]}; p2 = {r[15], r[16], r[17]}; q2 = {r[18], r[19], r[20], r[21]};
  };
  const int i = base + lane;
  const bool want = i < n;
  double dist = distmax;
  V3 from{0, 0, 0}, to{0, 0, 0};
  bool need_epa = false, known = true;
  if (want) {
    int t1, t2; V3 s1, p1, s2, p2; Q4 q1, q2;
    load(i, t1, s1, p1, q1, t2, s2, p2, q2);
    // (two GJK banks: every lane of this kernel may hold a convex pair)
    known = geom_distance<false, true>(t1, s1, p1, q1, t2, s2, p2, q2, distmax, dist, from, to, nullptr, 0, nullptr, 0, &need_epa,
                                       ws + (lane >> 5) * kGjkWsDoubles + (lane & 31));
  }
  for (unsigned long long em = __ballot(want && need_epa); em; em &= em - 1) {
    const int l = (int)__builtin_ctzll(em);
    int t1, t2; V3 s1, p1, s2, p2; Q4 q1, q2;
    load(base + l, t1, s1, p1, q1, t2, s2, p2, q2);
    // (loose polytope + polish on the pair's lane; without a certificate the tight polytope — collide_dev.h geom_overlap_polish)
    bool certified = false;
#pragma nounroll
    for (int pass = geom_overlap_loose(t1, t2) ? 0 : 1; pass < 2 && !certified; ++pass) {
      double d_e; V3 f_e, t_e;
      geom_overlap_distance(t1, s1, p1, q1, t2, s2, p2, q2, d_e, f_e, t_e, nullptr, 0, nullptr, 0, ws, pass ? mkh::kEpaTol : mkh::kLooseEpa);
      bool ok = false;
      if (lane == l) { dist = d_e; from = f_e; to = t_e; ok = geom_overlap_polish(t1, s1, p1, q1, t2, s2, p2, q2, dist, from, to, pass != 0); }
      certified = __ballot(ok) != 0;
    }
  }
  if (want) {
    dist_out[i] = known ? dist : __builtin_nan("");
    double* o = fromto_out + 6 * (size_t)i;
    o[0] = from.x; o[1] = from.y; o[2] = from.z; o[3] = to.x; o[4] = to.y; o[5] = to.z;
  }
}
}  // namespace

extern "C" int32_t mkh_geom_distance_eval(int32_t device, int32_t n, const double* pairs, double distmax, double* dist_out,
                                          double* fromto_out, void* hip_stream) {
  if (n < 1 || !pairs || !dist_out || !fromto_out) return fail(MKH_E_INVALID, "mkh_geom_distance_eval: null argument or n < 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MKH_E_NOGPU, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MKH_E_INVALID, "device %d out of range (%d visible)", device, ndev);
  HIP_OK(hipSetDevice(device));
  hipStream_t stream = (hipStream_t)hip_stream;
  double *dg = nullptr, *dd = nullptr, *df = nullptr;
  hipError_t e = hipMalloc((void**)&dg, (size_t)n * 22 * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&dd, (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&df, (size_t)n * 48);
  if (e == hipSuccess) e = hipMemcpyAsync(dg, pairs, (size_t)n * 22 * 8, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(geom_distance_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, n, dg, distmax, dd, df);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(dist_out, dd, (size_t)n * 8, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(fromto_out, df, (size_t)n * 48, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dg); (void)hipFree(dd); (void)hipFree(df);
  if (e != hipSuccess) return fail(MKH_E_HIP, "geom_distance_eval: %s", hipGetErrorString(e));
  return MKH_OK;
}
