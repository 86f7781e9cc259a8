// Cross-lane primitives for one-wavefront-per-problem kernels on gfx950 (wave64).
//
// Everything here stays in the VALU/SGPR path: v_readlane for broadcasts (an SGPR
// pair feeds v_fma_f64 directly), DPP row permutes for reductions.  `__shfl` is
// deliberately avoided: with a wave-uniform source lane hipcc still lowers it to
// ds_bpermute_b32 (LDS pipe, ~50+ cycles) instead of v_readlane_b32.
#pragma once
#include <hip/hip_runtime.h>

namespace mkh {

#ifdef MKH_ONE_SHOT
// (mbcnt, not threadIdx.x & 63: a callee that reads the work-item id makes its caller keep v0 alive for the v31 argument of every
//  call — in the one-problem-per-workgroup twin of the headline's kernel that was one of three spilled VGPRs; round 6.  That build
//  only: everywhere else the change moves spill counts by ± a dozen and measures nothing or worse.)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
#else
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
#endif

// Broadcast lane `src` (wave-uniform) of x to all lanes through SGPRs.
__device__ __forceinline__ double readlane_f64(double x, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}
// Lane `src` (per lane) of x through the LDS crossbar (ds_bpermute_b32 × 2): no store → wait → load round trip.
__device__ __forceinline__ double bperm_f64(double x, int src) {
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int readlane_i32(int x, int src) { return __builtin_amdgcn_readlane(x, src); }

// Tell the compiler a value is wave-uniform (it cannot prove that for loads through pointers that were
// themselves loaded from memory, and then keeps descriptor fields in VGPRs): first lane → SGPR.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double uni(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ unsigned long long uni(unsigned long long x) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// DPP controls (gfx9): quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E,
// row_half_mirror=0x141 (i -> 7-i in each 8), row_mirror=0x140 (i -> 15-i in each 16).
// After the four steps every 16-lane row holds its own reduction; rows are then
// combined through v_readlane, which also makes the result wave-uniform.
#define MKH_WAVE_REDUCE(NAME, OP)                                     \
  __device__ __forceinline__ double NAME(double x) {                  \
    x = OP(x, dpp_f64<0xB1>(x));                                      \
    x = OP(x, dpp_f64<0x4E>(x));                                      \
    x = OP(x, dpp_f64<0x141>(x));                                     \
    x = OP(x, dpp_f64<0x140>(x));                                     \
    double a = readlane_f64(x, 0), b = readlane_f64(x, 16);           \
    double c = readlane_f64(x, 32), d = readlane_f64(x, 48);          \
    return OP(OP(a, b), OP(c, d));                                    \
  }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
MKH_WAVE_REDUCE(wave_max, op_max)
MKH_WAVE_REDUCE(wave_sum, op_add)
#undef MKH_WAVE_REDUCE

// 32-bit unsigned reductions: one v_max/min_u32 with a DPP operand per step (half the cost of the
// 64-bit float versions above, which need two DPP moves + a 64-bit op per step).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
#define MKH_WAVE_REDUCE_U32(NAME, OP)                                                     \
  __device__ __forceinline__ unsigned NAME(unsigned x) {                                  \
    x = OP(x, dpp_u32<0xB1>(x));                                                          \
    x = OP(x, dpp_u32<0x4E>(x));                                                          \
    x = OP(x, dpp_u32<0x141>(x));                                                         \
    x = OP(x, dpp_u32<0x140>(x));                                                         \
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)x, 0);                    \
    const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)x, 16);                   \
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)x, 32);                   \
    const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)x, 48);                   \
    return OP(OP(a, b), OP(c, d));                                                        \
  }
MKH_WAVE_REDUCE_U32(wave_max_u32, umax32)
MKH_WAVE_REDUCE_U32(wave_min_u32, umin32)
#undef MKH_WAVE_REDUCE_U32

// Exact minimum of non-negative doubles (+inf allowed, no NaN) over the lanes in `cand` (the others
// must hold +inf): IEEE order = unsigned integer order of the bit patterns, so reduce the high words,
// then — only if several lanes tie on the high word — the low words among them.
__device__ __forceinline__ double wave_min_nonneg(double x, unsigned long long cand) {
  if (!cand) return __builtin_huge_val();
  const unsigned h = (unsigned)__double2hiint(x), l = (unsigned)__double2loint(x);
  const unsigned mh = wave_min_u32(h);
  const unsigned long long tie = __ballot(h == mh);
  unsigned ml;
  if (__builtin_popcountll(tie) == 1) ml = (unsigned)__builtin_amdgcn_readlane((int)l, (int)__builtin_ctzll(tie));
  else ml = wave_min_u32(h == mh ? l : 0xffffffffu);
  return __hiloint2double((int)mh, (int)ml);
}

// Index of the first lane where pred holds (wave-uniform), or -1.
__device__ __forceinline__ int first_lane(bool pred) {
  unsigned long long m = __ballot(pred);
  return m ? (int)__builtin_ctzll(m) : -1;
}

// Make LDS writes of this wave visible to its other lanes.  The block is exactly
// one wavefront (launch_bounds 64), DS ops of a wave execute in order, so this is
// a compiler-level fence plus an lgkmcnt wait; the s_barrier is elided by LLVM.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

}  // namespace mkh
