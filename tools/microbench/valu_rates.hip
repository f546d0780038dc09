// Microbenchmark: issue cost (shader-clock cycles per wave64 instruction, one wavefront alone on its SIMD) of what the policy
// kernels' layer epilogue is made of -- v_exp_f32, v_rcp_f32, the float16 conversions, packed float32 arithmetic -- and of
// v_mfma_f32_16x16x32_f16 alone and with independent VALU instructions between the MFMAs (do they hide under it?).
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__constant__ float g_m = 1.0000001f, g_c = 1e-9f;

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(int iters, unsigned long long* cycles, float* sink) {
  float a[8];
  for (int k = 0; k < 8; k++) a[k] = 0.5f + 0.001f * (threadIdx.x + k);
  const float m = g_m, c = g_c;
  f2 p[4] = {{a[0], a[1]}, {a[2], a[3]}, {a[4], a[5]}, {a[6], a[7]}};
  const f2 pm = {m, m}, pc = {c, c};
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (KIND == 0) a[k] = a[k] * m + c;                                         // v_fma_f32 (baseline)
      if (KIND == 1) a[k] = __builtin_amdgcn_exp2f(a[k]) * 0.5f;                  // v_exp_f32 + v_mul
      if (KIND == 2) a[k] = __builtin_amdgcn_rcpf(a[k]) * 0.5f;                   // v_rcp_f32 + v_mul
      if (KIND == 3) a[k] = (float)(_Float16)a[k] + c;                            // v_cvt_f16_f32 + v_cvt_f32_f16 + v_add
      if (KIND == 5) a[k] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a[k]) & 0xFFFFE000u) * m;  // v_and + v_mul
    }
    if (KIND == 4) {
#pragma unroll
      for (int k = 0; k < 4; k++) p[k] = p[k] * pm + pc;                           // v_pk_fma_f32 (8 values in 4 instructions)
    }
    if (KIND == 6) {
#pragma unroll
      for (int k = 0; k < 4; k++) {                                                // v_cvt_pkrtz_f16_f32 + 2 x v_cvt_f32_f16 (sdwa) + v_pk_add
        const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(p[k][0], p[k][1]));
        p[k] = f2{(float)h[0], (float)h[1]} + pc;
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float out = 0.f;
  for (int k = 0; k < 8; k++) out += a[k];
  for (int k = 0; k < 4; k++) out += p[k][0] + p[k][1];
  if (out == 12345.0f) sink[0] = out;
}

// NV independent VALU instructions after every MFMA (4 independent accumulators)
template <int NV>
__global__ __launch_bounds__(64) void k_mfma(int iters, unsigned long long* cycles, float* sink) {
  h8 x, y;
  for (int j = 0; j < 8; j++) { x[j] = (_Float16)(0.001f * (threadIdx.x + j)); y[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a[8];
  for (int k = 0; k < 8; k++) a[k] = 0.5f + 0.001f * (threadIdx.x + k);
  const float m = g_m, c = g_c;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[u], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; v++) a[(u * NV + v) & 7] = a[(u * NV + v) & 7] * m + c;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float out = 0.f;
  for (int k = 0; k < 8; k++) out += a[k];
  for (int u = 0; u < 4; u++) out += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  if (out == 12345.0f) sink[0] = out;
}

int main() {
  unsigned long long* cyc; float* sink;
  hipMalloc(&cyc, 8); hipMalloc(&sink, 4);
  const int iters = 4096;
  auto report = [&](const char* name, double per_iter_instr) {
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.2f cycles per wave-instruction (%.0f per iteration)\n", name, (double)h / iters / per_iter_instr, (double)h / iters);
  };
#define RUN(K, NAME, N) for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, iters, cyc, sink); hipDeviceSynchronize(); } report(NAME, N)
  RUN(k_rate<0>, "v_fma_f32 x8 independent", 8);
  RUN(k_rate<1>, "v_exp_f32 + v_mul_f32 x8 (pairs)", 8);
  RUN(k_rate<2>, "v_rcp_f32 + v_mul_f32 x8 (pairs)", 8);
  RUN(k_rate<3>, "cvt f32->f16->f32 + add x8 (triples)", 8);
  RUN(k_rate<4>, "v_pk_fma_f32 x4 (8 values)", 4);
  RUN(k_rate<5>, "v_and + v_mul x8 (pairs)", 8);
  RUN(k_rate<6>, "cvt_pkrtz + 2 cvt back + pk_add x4 (8 values)", 4);
  RUN(k_mfma<0>, "v_mfma_f32_16x16x32_f16 x4 independent", 4);
  RUN(k_mfma<1>, "mfma + 1 v_fma_f32, x4", 4);
  RUN(k_mfma<2>, "mfma + 2 v_fma_f32, x4", 4);
  RUN(k_mfma<3>, "mfma + 3 v_fma_f32, x4", 4);
  RUN(k_mfma<4>, "mfma + 4 v_fma_f32, x4", 4);
  return 0;
}
