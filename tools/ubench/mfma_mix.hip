// Micro-benchmark behind profiles/r04_ubench_mfma_mix.txt: the rank-18 update of a 44-row x 64-column fp64 tile
//     T -= U (44 x 18) . V (18 x 64)
// (what the low-rank start of the G1 kernel applies to the dof block per solve) done
//   dpp   : the way the production kernel does it — column per lane, rows pinned in VGPRs, 18 x 44 v_fmac_f64_dpp row_newbcast
//   mfma  : as 3 x 4 tiles of v_mfma_f64_16x16x4_f64, 5 k-steps each (18 -> 20): 60 matrix instructions, tile layout of the instruction
//   valu  : a stand-in for the rest of a solve — 792 independent v_fma_f64 on other registers
//   ialu  : the same count of 32-bit integer instructions (v_mad_u32_u24) instead
//   valu+mfma : both instruction streams interleaved in ONE wave (does the matrix pipe run beside the wave's own VALU work?)
//   valu|mfma : half the waves of every SIMD run `valu`, the others `mfma` (does it run beside OTHER waves' VALU work?)
// at the kernel's residency (3 single-wave workgroups per SIMD, 3 072 waves) and alone (1 per SIMD).
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_mix tools/ubench/mfma_mix.hip && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kRows = 44, kRank = 18, kTilesR = 3, kTilesC = 4, kSteps = 5;

// one v_fmac_f64_dpp with the multiplier broadcast from lane `I` of every 16-lane row
template <int I> __device__ __forceinline__ void fmac_bcast(double& acc, double u, double p) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(p), "n"(I));
}

template <int R> struct DppRows {
  static __device__ __forceinline__ void run(double (&T)[kRows], const double (&u)[3], double p) {
    fmac_bcast<R % 16>(T[R], u[R / 16], p);
    DppRows<R + 1>::run(T, u, p);
  }
};
template <> struct DppRows<kRows> { static __device__ __forceinline__ void run(double (&)[kRows], const double (&)[3], double) {} };

enum { MODE_DPP = 0, MODE_MFMA = 1, MODE_VALU = 2, MODE_BOTH = 3, MODE_SPLIT = 4, MODE_INT = 5, MODE_INT_BOTH = 6, MODE_INT_SPLIT = 7 };

template <bool MFMA_WAVE, bool VALU_WAVE, bool INT_VALU = false> __device__ __forceinline__ double tile_loop(int lane, int iters) {
  d4 C[kTilesR * kTilesC];
  double W[24];
#pragma unroll
  for (int t = 0; t < kTilesR * kTilesC; ++t) C[t] = d4{1e-3 * lane, 2e-3, 3e-3, 1e-3 * t};
#pragma unroll
  for (int i = 0; i < 24; ++i) W[i] = 1e-3 * (i + lane);
  double a[kTilesR], b[kTilesC], m = 1.0 + 1e-9 * lane;
#pragma unroll
  for (int i = 0; i < kTilesR; ++i) a[i] = 1e-6 * (lane + i);
#pragma unroll
  for (int j = 0; j < kTilesC; ++j) b[j] = 1.0 + 1e-6 * (lane + j);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      if (MFMA_WAVE) {
#pragma unroll
        for (int i = 0; i < kTilesR; ++i)
#pragma unroll
          for (int j = 0; j < kTilesC; ++j) C[i * kTilesC + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], C[i * kTilesC + j], 0, 0, 0);
      }
      if (VALU_WAVE) {                          // 792 / 5 = 158.4 -> 6.6 passes over 24 independent accumulators per k-step
#pragma unroll
        for (int rep = 0; rep < 7; ++rep)
#pragma unroll
          for (int i = 0; i < 24; ++i) {
            if (rep == 6 && i >= 14) break;     // 6 * 24 + 14 = 158 (x 5 = 790)
            if (INT_VALU) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(((int*)&W[i])[0]) : "v"(lane));
            else asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(W[i]) : "v"(m));
          }
      }
      a[0] += 1e-9;
    }
  }
  double acc = 0;
#pragma unroll
  for (int t = 0; t < kTilesR * kTilesC; ++t) acc += C[t].x + C[t].y + C[t].z + C[t].w;
#pragma unroll
  for (int i = 0; i < 24; ++i) acc += W[i];
  return acc;
}

template <int MODE> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k(double* out, int iters) {
  const int lane = threadIdx.x;
  double acc = 0;
  if (MODE == MODE_DPP) {
    double T[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) T[r] = 1e-3 * (r + lane);
    double u[3] = {1e-6 * lane, 2e-6 * lane, 3e-6 * lane}, p = 1.0 + 1e-3 * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < kRank; ++c) {          // 18 rank-1 updates: the column planes u and the pivot row p change per update
        DppRows<0>::run(T, u, p);
        u[0] += 1e-9; p += 1e-9;
      }
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r) acc += T[r];
  } else if (MODE == MODE_MFMA) acc = tile_loop<true, false>(lane, iters);
  else if (MODE == MODE_VALU) acc = tile_loop<false, true>(lane, iters);
  else if (MODE == MODE_BOTH) acc = tile_loop<true, true>(lane, iters);
  else if (MODE == MODE_INT) acc = tile_loop<false, true, true>(lane, iters);
  else if (MODE == MODE_INT_BOTH) acc = tile_loop<true, true, true>(lane, iters);
  else if ((blockIdx.x / 4) & 1) acc = tile_loop<true, false>(lane, iters);   // workgroups go round the 4 SIMDs of a CU: every SIMD gets both kinds
  else if (MODE == MODE_INT_SPLIT) acc = tile_loop<false, true, true>(lane, iters);
  else acc = tile_loop<false, true>(lane, iters);
  out[blockIdx.x * 64 + lane] = acc;
}

// placement-controlled mix: ONE 768-thread workgroup per CU = 3 waves per SIMD; the wave says which SIMD it runs on (HW_ID) and the
// host checks that every SIMD holds exactly one matrix wave and two vector waves before the time is believed
template <bool INT_VALU> __global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
void ksplit(double* out, int* where, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const bool matrix_wave = (w / 4) == 1;
  if (lane == 0) where[blockIdx.x * 12 + w] = (int)(((hw >> 4) & 3) | (matrix_wave ? 4 : 0) | (((hw >> 8) & 15) << 4) | (((hw >> 13) & 7) << 8));
  double acc;
  if (matrix_wave) acc = tile_loop<true, false>(lane, iters);
  else acc = tile_loop<false, true, INT_VALU>(lane, iters);
  out[blockIdx.x * 768 + threadIdx.x] = acc;
}

template <bool INT_VALU> static void run_split(const char* name, int iters, double* d_out) {
  int* d_where; CK(hipMalloc(&d_where, 4 * 256 * 12));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(ksplit<INT_VALU>, dim3(256), dim3(768), 0, 0, d_out, d_where, rep ? iters : 10);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  std::vector<int> where(256 * 12);
  CK(hipMemcpy(where.data(), d_where, 4 * 256 * 12, hipMemcpyDeviceToHost));
  int good = 0;
  for (int b = 0; b < 256; ++b) {
    int mat[4] = {0, 0, 0, 0}, vec[4] = {0, 0, 0, 0};
    for (int w = 0; w < 12; ++w) ((where[b * 12 + w] & 4) ? mat : vec)[where[b * 12 + w] & 3]++;
    bool ok = true;
    for (int sd = 0; sd < 4; ++sd) ok = ok && mat[sd] == 1 && vec[sd] == 2;
    good += ok;
  }
  printf("%-10s 256 workgroups of 12 waves : %9.1f ns per iteration; %d of 256 workgroups have 1 matrix + 2 vector waves on every SIMD\n", name,
         best * 1e6 / iters, good);
}

// raw probe: lane l feeds a[l], b[l] and stores its four results; main() finds the element mapping that explains them
__global__ void layout_probe(const double* a, const double* b, double* d) {
  const int lane = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[lane * 4 + v] = c[v];
}

template <int MODE> static double run(const char* name, int grid, int iters, double* d_out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d_out, 10);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d_out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // ns per iteration and SIMD: waves of one SIMD run concurrently, so this is what one "rank-18 update per resident wave" costs the SIMD
  const double ns = best * 1e6 / iters;
  printf("%-10s grid %5d : %9.1f ns per iteration (all resident waves one update each)\n", name, grid, ns);
  return ns;
}

int main() {
  double* d_out; CK(hipMalloc(&d_out, 8 * 64 * 4096));
  // layout check: A[i][k] on lane 16k + i, B[k][j] on lane 16k + j; where is D[i][j]?
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 64; ++i) { A[i] = 0.25 * (i % 7) - 0.5 + 0.01 * i; B[i] = 0.125 * (i % 5) + 0.1 * i; }
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  auto ref = [&](int i, int j) { double r = 0; for (int kk = 0; kk < 4; ++kk) r += A[16 * kk + i] * B[16 * kk + j]; return r; };
  const char* names[2] = {"D[4 (lane / 16) + v][lane % 16]", "D[4 v + lane / 16][lane % 16]"};
  for (int m = 0; m < 2; ++m) {
    double err = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v)
      err = fmax(err, fabs(D[l * 4 + v] - ref(m == 0 ? 4 * (l / 16) + v : 4 * v + l / 16, l % 16)));
    printf("v_mfma_f64_16x16x4_f64, A[i][k] on lane 16k+i, B[k][j] on lane 16k+j, element v of a lane = %s: max |err| %.2e\n", names[m], err);
  }
  const int iters = 2000;
  for (int grid : {1024, 3072}) {
    printf("# %d single-wave workgroups = %d per SIMD\n", grid, grid / 1024);
    run<MODE_DPP>("dpp", grid, iters, d_out);
    run<MODE_MFMA>("mfma", grid, iters, d_out);
    run<MODE_VALU>("valu", grid, iters, d_out);
    run<MODE_BOTH>("valu+mfma", grid, iters, d_out);
    if (grid > 1024) run<MODE_SPLIT>("valu|mfma", grid, iters, d_out);
    run<MODE_INT>("ialu", grid, iters, d_out);
    run<MODE_INT_BOTH>("ialu+mfma", grid, iters, d_out);
    if (grid > 1024) run<MODE_INT_SPLIT>("ialu|mfma", grid, iters, d_out);
  }
  printf("# per SIMD: one wave of `mfma` beside two waves of `valu` / `ialu` (no overlap: 1 x mfma + 2 x valu of the 1-per-SIMD lines; full overlap: 2 x valu)\n");
  run_split<false>("2valu|mfma", iters, d_out);
  run_split<true>("2ialu|mfma", iters, d_out);
  return 0;
}
