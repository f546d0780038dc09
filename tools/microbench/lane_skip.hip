// Microbenchmark: does a wave64 VALU instruction get cheaper when only the first 32 / 16 lanes are active (EXEC mask)?
// If the SIMD skipped the 16-lane passes whose lanes are all masked off, a 16-envs-per-wave layout could spread 4096 envs
// over all 256 CUs at four times the issue rate per env.  One wavefront per workgroup, 8 independent accumulators per lane,
// straight-line unrolled FMAs; reported: s_memtime ticks per wave-instruction.
// Also: issue cost of the cross-lane moves a lane-group-per-env layout would use (DPP quad_perm, row_shr, ds_swizzle, readlane).
// Build: hipcc -O3 --offload-arch=gfx950 lane_skip.hip -o lane_skip.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__constant__ double g_m = 1.0000001, g_c = 1e-9;   // run-time values: nothing folds
template <typename T>
__device__ __forceinline__ T chain(T x, int iters) {
  T a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  const T m = (T)g_m, c = (T)g_c;
#pragma unroll 32
  for (int i = 0; i < iters; i++) {
    a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
    a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
  }
  return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename T>
__global__ __launch_bounds__(64) void k_lanes(int active, int iters, unsigned long long* cycles, double* sink) {
  const int lane = threadIdx.x;
  double out = 0.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (lane < active) out = (double)chain<T>((T)lane, iters);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
  if (out == 12345.0) sink[0] = out;
}

// dependent chain of cross-lane moves + add (kind: 0 quad_perm dpp, 1 row_shr dpp, 2 ds_swizzle, 3 plain add as the baseline)
template <int KIND>
__global__ __launch_bounds__(64) void k_xlane(int iters, unsigned long long* cycles, float* sink) {
  float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {
      a0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0xB1, 0xF, 0xF, true));
      a1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0xB1, 0xF, 0xF, true));
      a2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x4E, 0xF, 0xF, true));
      a3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a3), 0x4E, 0xF, 0xF, true));
    } else if (KIND == 1) {
      a0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x111, 0xF, 0xF, true));
      a1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x111, 0xF, 0xF, true));
      a2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x112, 0xF, 0xF, true));
      a3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a3), 0x112, 0xF, 0xF, true));
    } else if (KIND == 2) {
      a0 += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a0), 0x041F));
      a1 += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a1), 0x041F));
      a2 += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a2), 0x081F));
      a3 += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a3), 0x081F));
    } else {
      a0 += a1 * 1.0000001f; a1 += a2 * 1.0000001f; a2 += a3 * 1.0000001f; a3 += a0 * 1.0000001f;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  const float out = a0 + a1 + a2 + a3;
  if (out == 12345.0f) sink[0] = out;
}

// one dependent chain per lane: latency of a single dependent instruction stream
template <typename T>
__global__ __launch_bounds__(64) void k_dep(int iters, unsigned long long* cycles, double* sink) {
  T a = (T)threadIdx.x;
  const T m = (T)g_m, c = (T)g_c;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 64
  for (int i = 0; i < iters; i++) a = a * m + c;
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if ((double)a == 12345.0) sink[0] = (double)a;
}
// two / three interleaved dependent chains
template <typename T, int NCH>
__global__ __launch_bounds__(64) void k_dep_n(int iters, unsigned long long* cycles, double* sink) {
  T a[NCH];
#pragma unroll
  for (int k = 0; k < NCH; k++) a[k] = (T)(threadIdx.x + k);
  const T m = (T)g_m, c = (T)g_c;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < NCH; k++) a[k] = a[k] * m + c;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  T s = 0;
#pragma unroll
  for (int k = 0; k < NCH; k++) s += a[k];
  if ((double)s == 12345.0) sink[0] = (double)s;
}

static double run_avg(std::vector<unsigned long long>& h, unsigned long long* d_c, int blocks) {
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d_c, blocks * sizeof(h[0]), hipMemcpyDeviceToHost);
  double s = 0;
  for (int b = 0; b < blocks; b++) s += (double)h[b];
  return s / blocks;
}

int main() {
  const int blocks = 64, iters = 2048;
  unsigned long long* d_c; double* d_s; float* d_f;
  hipMalloc(&d_c, blocks * sizeof(unsigned long long));
  hipMalloc(&d_s, 8); hipMalloc(&d_f, 4);
  std::vector<unsigned long long> h(blocks);
  // s_memtime ticks vs shader clock: report the tick rate by timing a known-length kernel with events
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_lanes<double>, dim3(blocks), dim3(64), 0, 0, 64, iters * 16, d_c, d_s);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_lanes<double>, dim3(blocks), dim3(64), 0, 0, 64, iters * 16, d_c, d_s);
    hipEventRecord(e1);
    const double ticks = run_avg(h, d_c, blocks);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("s_memtime: %.0f ticks in a %.1f us kernel -> %.1f MHz tick rate\n", ticks, ms * 1e3, ticks / (ms * 1e3));
  }
  for (int active : {64, 48, 32, 16, 8, 1}) {
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_lanes<float>, dim3(blocks), dim3(64), 0, 0, active, iters, d_c, d_s);
    const double f32 = run_avg(h, d_c, blocks) / (8.0 * iters);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_lanes<double>, dim3(blocks), dim3(64), 0, 0, active, iters, d_c, d_s);
    const double f64 = run_avg(h, d_c, blocks) / (8.0 * iters);
    printf("lanes active %2d: v_fma_f32 %.3f  v_fma_f64 %.3f ticks per wave-instruction (8 independent chains)\n", active, f32, f64);
  }
  {
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_dep<float>, dim3(blocks), dim3(64), 0, 0, iters, d_c, d_s);
    const double f32 = run_avg(h, d_c, blocks) / iters;
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_dep<double>, dim3(blocks), dim3(64), 0, 0, iters, d_c, d_s);
    const double f64 = run_avg(h, d_c, blocks) / iters;
    printf("one dependent chain: v_fma_f32 %.3f  v_fma_f64 %.3f ticks per instruction\n", f32, f64);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_dep_n<float, 2>), dim3(blocks), dim3(64), 0, 0, iters, d_c, d_s);
    const double a2 = run_avg(h, d_c, blocks) / (2.0 * iters);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_dep_n<double, 2>), dim3(blocks), dim3(64), 0, 0, iters, d_c, d_s);
    const double b2 = run_avg(h, d_c, blocks) / (2.0 * iters);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_dep_n<float, 3>), dim3(blocks), dim3(64), 0, 0, iters, d_c, d_s);
    const double a3 = run_avg(h, d_c, blocks) / (3.0 * iters);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_dep_n<double, 3>), dim3(blocks), dim3(64), 0, 0, iters, d_c, d_s);
    const double b3 = run_avg(h, d_c, blocks) / (3.0 * iters);
    printf("two chains: f32 %.3f f64 %.3f   three chains: f32 %.3f f64 %.3f ticks per instruction\n", a2, b2, a3, b3);
  }
  {
    const char* nm[4] = {"dpp quad_perm + add", "dpp row_shr + add", "ds_swizzle + add", "plain fma (baseline)"};
    double r[4];
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_xlane<0>, dim3(blocks), dim3(64), 0, 0, iters, d_c, d_f);
    r[0] = run_avg(h, d_c, blocks) / (4.0 * iters);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_xlane<1>, dim3(blocks), dim3(64), 0, 0, iters, d_c, d_f);
    r[1] = run_avg(h, d_c, blocks) / (4.0 * iters);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_xlane<2>, dim3(blocks), dim3(64), 0, 0, iters, d_c, d_f);
    r[2] = run_avg(h, d_c, blocks) / (4.0 * iters);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_xlane<3>, dim3(blocks), dim3(64), 0, 0, iters, d_c, d_f);
    r[3] = run_avg(h, d_c, blocks) / (4.0 * iters);
    for (int k = 0; k < 4; k++) printf("%-24s %.3f ticks per (move+add) pair, 4 interleaved chains\n", nm[k], r[k]);
  }
  return 0;
}
