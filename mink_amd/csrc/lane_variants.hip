// Instantiations of the lane-per-problem kernel (lane_kernel.h) and their launcher.
#include <hip/hip_runtime.h>

#include "lane_kernel.h"

namespace mkh {

int launch_lane(int nv_max, int grid, int lds_bytes, hipStream_t stream, const LaneProblem* P, const SolveArgs& a) {
  switch (nv_max) {
    case 4: hipLaunchKernelGGL(ik_lane_kernel<4>, dim3(grid), dim3(kWave), lds_bytes, stream, P, a); return 0;
    case 6: hipLaunchKernelGGL(ik_lane_kernel<6>, dim3(grid), dim3(kWave), lds_bytes, stream, P, a); return 0;
    case 7: hipLaunchKernelGGL(ik_lane_kernel<7>, dim3(grid), dim3(kWave), lds_bytes, stream, P, a); return 0;
    case 8: hipLaunchKernelGGL(ik_lane_kernel<8>, dim3(grid), dim3(kWave), lds_bytes, stream, P, a); return 0;
  }
  return -1;
}

}  // namespace mkh
