// Micro-benchmarks of the QP building blocks (cycles per call, one wave / block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mink_amd/csrc/ik_kernel.h"
using namespace mkh;
constexpr int NT = 48;

template <int WHICH>
__global__ __launch_bounds__(64, 2) __attribute__((amdgpu_num_vgpr(Tab<NT>::kCompilerVgprs)))
void k(double* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 128; i += 64) sm[i] = 1e-3 * (i + 1);
  __syncthreads();
  Tab<NT>::zero();
  double acc = 1.0 + lane * 1e-3, x = 0.5 + lane;
  int col = 3;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (WHICH == 0) { Tab<NT>::rank1(lds_addr(sm), acc); }
    if (WHICH == 1) { __syncthreads(); if (lane == col) { Tab<NT>::publish(lds_addr(sm)); sm[col] = 0.0; } __syncthreads(); col = (col + 7) % 43; }
    if (WHICH == 2) { acc = 1.0 / (acc + 1.5); }
    if (WHICH == 3) { acc += wave_max(acc * x); }
    if (WHICH == 4) { acc += readlane_f64(acc, col); col = (col + 7) % 43; }
    if (WHICH == 5) { __syncthreads(); sm[64 + lane] = acc; __syncthreads(); acc += sm[64 + ((lane + 1) & 63)]; }
    if (WHICH == 6) { acc += (double)first_lane(acc * x > 3.0 + it); }
    if (WHICH == 7) { acc = acc / (x + it); }
    if (WHICH == 8) { QpLane q; q.D = acc; q.sg = x; q.w = acc; q.z = x; PivotScalars ps; double own = publish_column<NT>(q, col, lane, sm, ps); acc += own + ps.d * 1e-9; col = (col + 7) % 43; }
    if (WHICH == 9) { QpLane q; q.D = acc; q.sg = 1.0; q.w = acc; q.z = x; PivotScalars ps; double own = publish_column<NT>(q, col, lane, sm, ps); double inv = fast_rcp(ps.d + 2.0); pivot<NT>(q, col, false, lane, sm, own * 1e-3, ps, inv); acc = q.D * 1e-3 + 1.0; col = (col + 7) % 43; }
    if (WHICH == 10) { acc = fast_rcp(acc + 1.5); }
    if (WHICH == 11) { asm volatile("v_fma_f64 v[160:161], %0, %0, v[160:161]\n v_fma_f64 v[162:163], %0, %0, v[162:163]\n v_fma_f64 v[164:165], %0, %0, v[164:165]\n v_fma_f64 v[166:167], %0, %0, v[166:167]\n v_fma_f64 v[168:169], %0, %0, v[168:169]\n v_fma_f64 v[170:171], %0, %0, v[170:171]\n v_fma_f64 v[172:173], %0, %0, v[172:173]\n v_fma_f64 v[174:175], %0, %0, v[174:175]" :: "v"(acc) : "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175"); }
    if (WHICH == 12) { acc += Tab<NT>::get_dyn(col); col = (col + 7) % 43; }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = acc + Tab<NT>::get<5>();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W> void run(const char* name, int grid) {
  double* out; long long* cyc;
  hipMalloc(&out, grid * 64 * 8); hipMalloc(&cyc, grid * 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k<W>, dim3(grid), dim3(64), 4096, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<W>, dim3(grid), dim3(64), 4096, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long* h = new long long[grid];
  hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
  printf("%-28s grid %5d : %8.1f cycles/iter\n", name, grid, s / grid / iters);
  hipFree(out); hipFree(cyc); delete[] h;
}
int main() {
  for (int grid : {256, 1536}) {   // 1 wave/CU and 8 waves/CU (2 per SIMD)
    run<0>("rank1 (24 b128 + 48 fma)", grid);
    run<1>("publish (48 ds_write_b64)", grid);
    run<2>("fp64 reciprocal", grid);
    run<3>("wave_max", grid);
    run<4>("readlane_f64", grid);
    run<5>("sync+lds write+sync+read", grid);
    run<6>("ballot first_lane", grid);
    run<7>("fp64 divide", grid);
    run<8>("publish_column (new)", grid);
    run<9>("publish+rcp+pivot", grid);
    run<10>("fast_rcp", grid);
    run<11>("8 independent v_fma_f64", grid);
    run<12>("get_dyn", grid);
  }
  return 0;
}
