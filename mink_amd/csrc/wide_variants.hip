// The workgroup-per-problem kernel (wide_kernel.h) and its launcher, in a translation unit of its own.
#include <hip/hip_runtime.h>

#include <mutex>

#include "wide_kernel.h"

namespace mkh {

// returns 0, or the HIP error of raising the kernel's dynamic-LDS ceiling (a tableau in LDS can want more than 64 KB)
int launch_wide(int grid, int lds_bytes, hipStream_t stream, const WideProblem* P, const SolveArgs& a, const TapArgs* taps, bool convex) {
  // (the attribute belongs to the CURRENT device's copy of the kernel: one ceiling per device, under a mutex — several devices
  //  or threads may create problems in one process; round-4 advisor finding)
  // (convex: the build with the general convex routine — wide_kernel.h, "the two builds")
  static std::mutex mu;
  static int raised[2][64] = {{0}, {0}};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  const void* fn = convex ? reinterpret_cast<const void*>(ik_wide_kernel_cvx) : reinterpret_cast<const void*>(ik_wide_kernel);
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || lds_bytes > raised[convex ? 1 : 0][dev]) {
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      if (e != hipSuccess) return (int)e;
      if (dev >= 0) raised[convex ? 1 : 0][dev] = lds_bytes;
    }
  }
  if (convex) hipLaunchKernelGGL(ik_wide_kernel_cvx, dim3(grid), dim3(kWideThreads), lds_bytes, stream, P, a, taps);
  else hipLaunchKernelGGL(ik_wide_kernel, dim3(grid), dim3(kWideThreads), lds_bytes, stream, P, a, taps);
  return 0;
}

}  // namespace mkh
