// Instantiations of the lane-per-problem kernel (lane_kernel.h), the row-per-problem kernel (quad_kernel.h) and their launchers.
#include <hip/hip_runtime.h>

#include "lane_kernel.h"
#include "quad_kernel.h"

namespace mkh {

int launch_lane(int nv_max, bool loop, int grid, int lds_bytes, hipStream_t stream, const LaneProblem* P, const SolveArgs& a) {
#define MKH_LANE_CASE(N)                                                                                          \
  case N:                                                                                                         \
    if (loop) hipLaunchKernelGGL((ik_lane_kernel<N, true>), dim3(grid), dim3(kWave), lds_bytes, stream, P, a);    \
    else hipLaunchKernelGGL((ik_lane_kernel<N, false>), dim3(grid), dim3(kWave), lds_bytes, stream, P, a);        \
    return 0;
  switch (nv_max) {
    MKH_LANE_CASE(4)
    MKH_LANE_CASE(6)
    MKH_LANE_CASE(7)
    MKH_LANE_CASE(8)
  }
#undef MKH_LANE_CASE
  return -1;
}

int launch_quad(int nt, bool loop, int grid, hipStream_t stream, const void* Pv, const LaneDims& dims, const SolveArgs& a) {
  const LaneProblem* P = (const LaneProblem*)Pv;          // (nt == 32: a LaneProblem2)
#define MKH_QUAD_LAUNCH(NT, LOOP, LP) hipLaunchKernelGGL((ik_quad_kernel<NT, LOOP, LP>), dim3(grid), dim3(kWave), quad_lds_bytes(LP), stream, P, dims, a)
  if (nt == 32) {
    if (loop) return -1;
    const LaneProblem2* P = (const LaneProblem2*)Pv;
    MKH_QUAD_LAUNCH(32, false, 32);
    return quad_lds_bytes(32);
  }   // two DPP rows per problem: single solves
  if (nt == 8) { if (loop) MKH_QUAD_LAUNCH(8, true, 16); else MKH_QUAD_LAUNCH(8, false, 16); }
  else { if (loop) MKH_QUAD_LAUNCH(16, true, 16); else MKH_QUAD_LAUNCH(16, false, 16); }
#undef MKH_QUAD_LAUNCH
  return quad_lds_bytes(16);
}

}  // namespace mkh
