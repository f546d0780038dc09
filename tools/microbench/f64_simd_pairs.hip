// Microbenchmark: does float64 VALU work on one wavefront of a workgroup slow another wavefront's float64 stream on the same CU?
// Four wavefronts per workgroup (one per SIMD, as k_rollout_coop runs them); a mask says which of them execute a chain of
// independent v_fma_f64 (8 accumulators per lane) and which a chain of v_fma_f32; the rest exit at once.  Reported: cycles
// (s_memtime) per wave-instruction as seen by wave 0.  The loop body is unrolled 64 times (512 FMAs of straight-line code, 4 KB), so
// the taken branch (≈ 25 cycles) does not hide the issue rate and the instruction fetch is exercised as in the real kernel.   Build: hipcc -O3 --offload-arch=gfx950 f64_simd_pairs.hip -o f64_simd_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename T>
__device__ __forceinline__ T chain(T x, int iters) {
  T a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  const T m = (T)1.0000001, c = (T)1e-9;
#pragma unroll 64
  for (int i = 0; i < iters; i++) {
    a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
    a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
  }
  return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// the same f64 stream as FOUR different pieces of code (one per wave: distinct addresses, as the roles of k_rollout_coop are):
// does the instruction fetch of four separate streams slow them down?  (Measured: no -- but the copies themselves run at 4.1 or
// 5.1 cycles per instruction depending on their placement: functions are 4-byte aligned, 8-byte instructions 4 bytes off the fetch
// window straddle it.)
template <int ID>
__device__ __attribute__((noinline)) double chain_copy(double x, int iters) {
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  const double m = 1.0000001 + 1e-9 * ID, c = 1e-9 * (ID + 1);
#pragma unroll 64
  for (int i = 0; i < iters; i++) {
    a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
    a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
  }
  return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void k_copies(int mask, int iters, unsigned long long* cycles, double* sink) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  double out = 0.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if ((mask >> wave) & 1) {
    if (wave == 0) out = chain_copy<0>((double)threadIdx.x, iters);
    else if (wave == 1) out = chain_copy<1>((double)threadIdx.x, iters);
    else if (wave == 2) out = chain_copy<2>((double)threadIdx.x, iters);
    else out = chain_copy<3>((double)threadIdx.x, iters);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
  if (out == 12345.678) sink[0] = out;
}

__global__ __launch_bounds__(256) void k(int mask64, int mask32, int iters, unsigned long long* cycles, double* sink) {
  const int wave = threadIdx.x >> 6;
  double out = 0.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if ((mask64 >> wave) & 1) out = chain<double>((double)threadIdx.x, iters);
  else if ((mask32 >> wave) & 1) out = (double)chain<float>((float)threadIdx.x, iters);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
  if (out == 12345.678) sink[0] = out;
}

int main() {
  const int blocks = 256, iters = 4096;
  unsigned long long* d_c; double* d_s;
  hipMalloc(&d_c, blocks * 4 * sizeof(unsigned long long));
  hipMalloc(&d_s, 8);
  std::vector<unsigned long long> h(blocks * 4);
  struct Case { const char* name; int m64, m32; };
  const Case cases[] = {
      {"wave0 f64 alone", 1, 0},          {"wave0 f32 alone", 0, 1},          {"waves 0,1 f64", 3, 0},
      {"waves 0,2 f64", 5, 0},            {"waves 0,3 f64", 9, 0},            {"all four f64", 15, 0},
      {"wave0 f64, wave1 f32", 1, 2},     {"wave0 f64, waves 1-3 f32", 1, 14}, {"all four f32", 0, 15},
      {"waves 0,1 f64, waves 2,3 f32", 3, 12},
  };
  for (const Case& c : cases) {
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, c.m64, c.m32, iters, d_c, d_s);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_c, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < blocks; b++) for (int w = 0; w < 4; w++) s[w] += (double)h[b * 4 + w];
    printf("%-32s s_memtime ticks per FMA wave-instruction, waves 0..3:", c.name);
    for (int w = 0; w < 4; w++) printf(" %7.3f", s[w] / blocks / (8.0 * iters));
    printf("\n");
  }
  const Case copies[] = {{"four code copies: wave0 alone", 1, 0}, {"four code copies: waves 0,1", 3, 0}, {"four code copies: all four waves", 15, 0}};
  for (const Case& c : copies) {
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_copies, dim3(blocks), dim3(256), 0, 0, c.m64, iters, d_c, d_s);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_c, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < blocks; b++) for (int w = 0; w < 4; w++) s[w] += (double)h[b * 4 + w];
    printf("%-32s s_memtime ticks per FMA wave-instruction, waves 0..3:", c.name);
    for (int w = 0; w < 4; w++) printf(" %7.3f", s[w] / blocks / (8.0 * iters));
    printf("\n");
  }
  return 0;
}
