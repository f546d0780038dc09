// qd_micro.hip -- measurements that size the step kernel's design on gfx950 (diagnostic, not product):
//   A  issue cost of one wavefront alone on its SIMD, per instruction kind (dependent chain / 4 independent chains)
//   B  cost of handing a value from one wave of a workgroup to another through LDS (s_barrier and flag forms)
//   C  per-launch period of graph-replayed dependent launches shaped like k_step: empty, memory-only, memory + N FMAs,
//      with 64 / 128 workgroups of 64 threads and 64 workgroups of 256 threads
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tests/_build/qd_micro tools/microbench/qd_micro.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

enum Op { FMA32_DEP, FMA32_IND, FMA64_DEP, FMA64_IND, PKFMA32_DEP, PKFMA32_IND, MUL64_DEP, ADD64_DEP, RCP64_DEP, RCP32_DEP, SQRT32_DEP, CVT_RT, DPP_ADD, OPS };
static const char* op_name[OPS] = {"v_fma_f32 dependent", "v_fma_f32 4 chains", "v_fma_f64 dependent", "v_fma_f64 4 chains", "v_pk_fma_f32 dependent",
                                   "v_pk_fma_f32 4 chains", "v_mul_f64 dependent", "v_add_f64 dependent", "v_rcp_f64 dependent", "v_rcp_f32 dependent",
                                   "v_sqrt_f32 dependent", "cvt f32->f64->f32 dependent (2 instr)", "v_add_f32 row_shr dpp dependent"};

template <int OP>
__global__ __launch_bounds__(64) void k_issue(unsigned long long* out, float seed) {
  float a = seed + threadIdx.x * 1e-3f, b = 0.999f, c = 1e-4f;
  float x0 = a, x1 = a + 1.f, x2 = a + 2.f, x3 = a + 3.f;
  double d0 = a, d1 = a + 1.0, d2 = a + 2.0, d3 = a + 3.0, db = 0.999, dc = 1e-4;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a, a}, p1 = {a + 1.f, a}, p2 = {a + 2.f, a}, p3 = {a + 3.f, a}, pb = {0.999f, 0.999f}, pc = {1e-4f, 1e-4f};
  const unsigned long long t0 = now();
  for (int it = 0; it < 16; it++) {
    if (OP == FMA32_DEP) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(b), "v"(c));) }
    if (OP == FMA32_IND) {
      REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));)
    }
    if (OP == FMA64_DEP) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(db), "v"(dc));) }
    if (OP == FMA64_IND) {
      REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)
    }
    if (OP == PKFMA32_DEP) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc));) }
    if (OP == PKFMA32_IND) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
    }
    if (OP == MUL64_DEP) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d0) : "v"(db));) }
    if (OP == ADD64_DEP) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(dc));) }
    if (OP == RCP64_DEP) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(d0));) }
    if (OP == RCP32_DEP) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(x0));) }
    if (OP == SQRT32_DEP) { REP64(asm volatile("v_sqrt_f32 %0, %0" : "+v"(x0));) }
    if (OP == CVT_RT) { REP64(asm volatile("v_cvt_f64_f32 %1, %0\n\tv_cvt_f32_f64 %0, %1" : "+v"(x0), "+v"(d0));) }
    if (OP == DPP_ADD) { REP64(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(x0));) }
  }
  const unsigned long long t1 = now();
  float sink = x0 + x1 + x2 + x3 + (float)(d0 + d1 + d2 + d3) + p0.x + p1.x + p2.x + p3.x + p0.y;
  if (sink == 12345.678f) out[1] = 1;  // keep everything alive
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

// B: ping-pong of one dword between wave 0 and wave 1 of a 128-thread workgroup
template <bool FLAG>
__global__ __launch_bounds__(128) void k_handoff(unsigned long long* out, int rounds) {
  __shared__ volatile float box[2][64];
  __shared__ volatile int flag[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 2) flag[threadIdx.x] = 0;
  float v = (float)lane;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int r = 1; r <= rounds; r++) {
    if (FLAG) {
      if (wave == 0) {
        box[0][lane] = v;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        if (lane == 0) flag[0] = r;
        while (flag[1] != r) {}
        v = box[1][lane] + 1.f;
      } else {
        while (flag[0] != r) {}
        const float w = box[0][lane] + 1.f;
        box[1][lane] = w;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) flag[1] = r;
      }
    } else {
      if (wave == 0) box[0][lane] = v;
      __syncthreads();
      if (wave == 1) box[1][lane] = box[0][lane] + 1.f;
      __syncthreads();
      if (wave == 0) v = box[1][lane] + 1.f;
    }
  }
  const unsigned long long t1 = now();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)v; }
}

// C: launches shaped like k_step.  planes: float4[npad] each; reads R planes, writes W planes (the first W of the ones read),
// NF dependent FMAs in between; obs-like streaming store of 22 floats per lane through LDS is left out (memory-only shape).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_shape(float4* g, int npad, int n, int R, int W, int nf, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) {  // "sampler"-like extra workgroups: two loads, exit
    const int j = ((int)blockIdx.x - main_blocks) * BLOCK + threadIdx.x;
    if (j < n) {
      const float4 a = g[20 * npad + j], b = g[5 * npad + j];
      if (a.x == 123.f && b.y == 7.f) g[21 * npad + j] = a;
    }
    return;
  }
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = k < R ? g[k * npad + i] : make_float4(0.f, 0.f, 0.f, 0.f);  // all loads in flight together
#pragma unroll
  for (int k = 0; k < 16; k++) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
  float x = acc.x;
  for (int k = 0; k < nf; k++) x = fmaf(x, 0.999f, 1e-4f);
  acc.x = x;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (k < W) g[k * npad + i] = make_float4(acc.x * 1e-3f, acc.y * 1e-3f, acc.z * 1e-3f, acc.w * 1e-3f + k);
}

__global__ void k_empty(int) {}

static double graph_period_us(hipStream_t s, int T, int reps, const std::function<void(hipStream_t)>& launch) {
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int t = 0; t < T; t++) launch(s);
  CK(hipStreamEndCapture(s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; r++) CK(hipGraphLaunch(exec, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; r++) CK(hipGraphLaunch(exec, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(exec));
  CK(hipGraphDestroy(graph));
  return ms * 1e3 / ((double)T * reps);
}

#include <functional>

template <int OP>
static void run_issue(unsigned long long* d_out) {
  unsigned long long h[2];
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_issue<OP>, dim3(1), dim3(64), 0, 0, d_out, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
  const int per_iter = (OP == CVT_RT) ? 128 : 64;
  printf("A  %-42s %6.2f cycles / instruction\n", op_name[OP], (double)h[0] / (16.0 * per_iter));
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long* d_out;
  CK(hipMalloc(&d_out, 64));
  CK(hipMemset(d_out, 0, 64));
  run_issue<FMA32_DEP>(d_out); run_issue<FMA32_IND>(d_out); run_issue<FMA64_DEP>(d_out); run_issue<FMA64_IND>(d_out);
  run_issue<PKFMA32_DEP>(d_out); run_issue<PKFMA32_IND>(d_out); run_issue<MUL64_DEP>(d_out); run_issue<ADD64_DEP>(d_out);
  run_issue<RCP64_DEP>(d_out); run_issue<RCP32_DEP>(d_out); run_issue<SQRT32_DEP>(d_out); run_issue<CVT_RT>(d_out); run_issue<DPP_ADD>(d_out);

  for (int flag = 0; flag < 2; flag++) {
    unsigned long long h[2];
    const int rounds = 2000;
    for (int r = 0; r < 2; r++) {
      if (flag) hipLaunchKernelGGL(k_handoff<true>, dim3(1), dim3(128), 0, 0, d_out, rounds);
      else hipLaunchKernelGGL(k_handoff<false>, dim3(1), dim3(128), 0, 0, d_out, rounds);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
    printf("B  LDS hand-off wave0 -> wave1 -> wave0 (%s): %7.1f cycles per round trip (2 hops)\n", flag ? "LDS flag poll" : "s_barrier x2", (double)h[0] / rounds);
  }

  const int n = 4096, npad = 4096, planes = 24;
  float4* g;
  CK(hipMalloc(&g, sizeof(float4) * npad * planes));
  CK(hipMemset(g, 0, sizeof(float4) * npad * planes));
  const int T = 512, reps = 8;
  printf("C  empty kernel, 1 block:                          %6.3f us / launch\n", graph_period_us(s, T, reps, [&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, 0); }));
  printf("C  empty kernel, 128 blocks x 64:                  %6.3f us / launch\n", graph_period_us(s, T, reps, [&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(128), dim3(64), 0, st, 0); }));
  struct Shape { int block, blocks_mult, R, W, nf; const char* what; };
  const Shape shapes[] = {
      {64, 1, 1, 1, 0, "64 wg x 64, 1 load 1 store"},
      {64, 1, 14, 7, 0, "64 wg x 64, 14 loads 7 stores"},
      {64, 2, 14, 7, 0, "64+64 wg x 64, 14 loads 7 stores (+sampler-like wgs)"},
      {64, 1, 14, 7, 250, "64 wg x 64, 14/7 + 250 dependent FMAs"},
      {64, 1, 14, 7, 500, "64 wg x 64, 14/7 + 500 dependent FMAs"},
      {64, 1, 14, 7, 1000, "64 wg x 64, 14/7 + 1000 dependent FMAs"},
      {64, 2, 14, 7, 1000, "64+64 wg x 64, 14/7 + 1000 dependent FMAs"},
      {256, 1, 14, 7, 1000, "16 wg x 256, 14/7 + 1000 dependent FMAs"},
      {64, 1, 14, 7, 1500, "64 wg x 64, 14/7 + 1500 dependent FMAs"},
  };
  for (const Shape& sh : shapes) {
    const int mb = (n + sh.block - 1) / sh.block;
    const double us = graph_period_us(s, T, reps, [&](hipStream_t st) {
      if (sh.block == 64) hipLaunchKernelGGL(k_shape<64>, dim3(mb * sh.blocks_mult), dim3(64), 0, st, g, npad, n, sh.R, sh.W, sh.nf, mb);
      else hipLaunchKernelGGL(k_shape<256>, dim3(mb * sh.blocks_mult), dim3(256), 0, st, g, npad, n, sh.R, sh.W, sh.nf, mb);
    });
    printf("C  %-58s %6.3f us / launch\n", sh.what, us);
  }
  CK(hipFree(g));
  CK(hipFree(d_out));
  return 0;
}
