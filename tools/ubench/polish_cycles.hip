// Round 6 micro-benchmark: cycles of cvx_gjk and of cvx_polish for one cylinder–box pair on ONE lane of a wavefront (how the pair
// lanes of the collision phase run them), poses drawn like `ur5e_convex`'s.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mink_amd/csrc
//   tools/ubench/polish_cycles.hip -o /tmp/polish_cycles && /tmp/polish_cycles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "convex_dev.h"
using namespace mkh;

__global__ void bench(const double* in, long long* out, int n, int waves_active) {
  __shared__ double ws[kGjkWsDoubles];
  const int i = blockIdx.x;
  if (i >= n) return;
  const double* p = in + 16 * i;
  if (threadIdx.x != 0) return;
  const ConvexRel g{kGeomCylinder, V3{0.04, 0.05, 0.0}, nullptr, 0, kGeomBox, V3{0.1, 0.1, 0.1}, nullptr, 0, Q4{p[0], p[1], p[2], p[3]}, V3{p[4], p[5], p[6]}};
  double dc = 0.0;
  V3 pa{0, 0, 0}, pb{0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  const bool apart = cvx_gjk(g, 0.3, dc, pa, pb, ws);
  const long long t1 = __builtin_readcyclecounter();
  long long t2 = t1;
  double h = 0.0;
  if (apart && dc > 1e-9 && dc < 0.3) {
    const V3 nrm = (1.0 / dc) * (pb - pa);
    const CvxPolish pl = cvx_polish(g.t1, g.s1, g.t2, g.s2, g.q21, g.p21, nrm);
    t2 = __builtin_readcyclecounter();
    h = pl.ok ? pl.h : 1.0;
  }
  out[4 * i] = t1 - t0; out[4 * i + 1] = t2 - t1; out[4 * i + 2] = (long long)(h * 1e15); out[4 * i + 3] = apart;
}

int main() {
  const int n = 4096;
  std::vector<double> in(16 * n);
  srand(1);
  auto rnd = []() { return rand() / (double)RAND_MAX * 2.0 - 1.0; };
  for (int i = 0; i < n; ++i) {
    double q[4] = {rnd(), rnd(), rnd(), rnd()};
    const double l = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) in[16 * i + k] = q[k] / l;
    double d[3] = {rnd(), rnd(), rnd()};
    const double dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double r = 0.2 + 0.15 * (rand() / (double)RAND_MAX);
    for (int k = 0; k < 3; ++k) in[16 * i + 4 + k] = d[k] / dl * r;
  }
  double* din; long long* dout;
  hipMalloc(&din, in.size() * 8); hipMalloc(&dout, 4 * n * 8);
  hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench, dim3(n), dim3(64), 0, 0, din, dout, n, 0);
  hipDeviceSynchronize();
  std::vector<long long> out(4 * n);
  hipMemcpy(out.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
  double g = 0, p = 0; int np = 0, ng = 0;
  for (int i = 0; i < n; ++i) { g += out[4 * i]; ++ng; if (out[4 * i + 1] > 0) { p += out[4 * i + 1]; ++np; } }
  printf("pairs %d: GJK mean %.0f cycles; polished %d: polish mean %.0f cycles\n", ng, g / ng, np, p / np);
  return 0;
}
