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

int launch_quad(int nt, bool loop, int grid, hipStream_t stream, const LaneProblem* P, const LaneDims& dims, const SolveArgs& a) {
#define MKH_QUAD_LAUNCH(NT, LOOP) hipLaunchKernelGGL((ik_quad_kernel<NT, LOOP>), dim3(grid), dim3(kWave), quad_lds_bytes(), stream, P, dims, a)
  if (nt == 8) { if (loop) MKH_QUAD_LAUNCH(8, true); else MKH_QUAD_LAUNCH(8, false); }
  else { if (loop) MKH_QUAD_LAUNCH(16, true); else MKH_QUAD_LAUNCH(16, false); }
#undef MKH_QUAD_LAUNCH
  return quad_lds_bytes();
}

}  // namespace mkh
