// The workgroup-per-problem kernel (wide_kernel.h) and its launcher, in a translation unit of its own.
#include <hip/hip_runtime.h>

#include "wide_kernel.h"

namespace mkh {

// returns 0, or the HIP error of raising the kernel's dynamic-LDS ceiling (a tableau in LDS can want more than 64 KB)
int launch_wide(int grid, int lds_bytes, hipStream_t stream, const WideProblem* P, const SolveArgs& a, const TapArgs* taps) {
  static int raised = 0;
  if (lds_bytes > raised) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ik_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    raised = lds_bytes;
  }
  hipLaunchKernelGGL(ik_wide_kernel, dim3(grid), dim3(kWideThreads), lds_bytes, stream, P, a, taps);
  return 0;
}

}  // namespace mkh
