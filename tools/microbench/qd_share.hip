// qd_share.hip -- should the three waves of k_step_coop fetch the env's 17 planes once and share them through LDS?
// 64 workgroups x 192 threads, dependent launches from a replayed graph.  Variant A: every wave loads all 17 planes (what the
// kernel does).  Variant B: wave w loads planes w, w+3, ... (6 / 6 / 5), writes them to LDS, barrier, every wave reads all 17
// from LDS.  Stamps: cycles from the wave's first instruction until all 17 planes are in its registers.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 -o tests/_build/qd_share tools/microbench/qd_share.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NP = 17;
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return t;
}
template <bool SHARE>
__global__ __launch_bounds__(192) void k(float4* g, int npad, unsigned long long* out) {
  __shared__ float4 L[NP][64];
  const unsigned long long t0 = now();
  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63, i = blockIdx.x * 64 + lane;
  float4 v[NP];
  if (SHARE) {
    float4 mine[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { const int p = role + 3 * k; mine[k] = p < NP ? g[p * npad + i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int k = 0; k < 6; k++) { const int p = role + 3 * k; if (p < NP) L[p][lane] = mine[k]; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int p = 0; p < NP; p++) v[p] = L[p][lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
#pragma unroll
    for (int p = 0; p < NP; p++) v[p] = g[p * npad + i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < NP; p++) { acc.x += v[p].x; acc.y += v[p].y; acc.z += v[p].z; acc.w += v[p].w; }
  asm volatile("" ::"v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));
  const unsigned long long t1 = now();
  if (role == 0) {
#pragma unroll
    for (int p = 0; p < 7; p++) g[p * npad + i] = make_float4(acc.x * 1e-3f, acc.y * 1e-3f, acc.z * 1e-3f, acc.w * 1e-3f + p);
  }
  if (lane == 0) out[blockIdx.x * 4 + role] = t1 - t0;
}
int main() {
  const int n = 4096, npad = 4096, T = 256, reps = 8;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float4* g;
  unsigned long long* out;
  CK(hipMalloc(&g, sizeof(float4) * npad * NP));
  CK(hipMemset(g, 0, sizeof(float4) * npad * NP));
  CK(hipMalloc(&out, 64 * 4 * 8));
  for (int share = 0; share < 2; share++) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < T; t++) {
      if (share) hipLaunchKernelGGL(k<true>, dim3(n / 64), dim3(192), 0, s, g, npad, out);
      else hipLaunchKernelGGL(k<false>, dim3(n / 64), dim3(192), 0, s, g, npad, out);
    }
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; r++) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(64 * 4), c;
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < 64; b++) for (int r = 0; r < 3; r++) c.push_back(h[b * 4 + r]);
    std::sort(c.begin(), c.end());
    printf("%-46s %.3f us per launch; cycles until all 17 planes are in registers: median %llu  min %llu  max %llu\n",
           share ? "B  each plane fetched once, shared through LDS:" : "A  every wave fetches all 17 planes:", ms * 1e3 / (T * reps), c[c.size() / 2], c.front(), c.back());
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
  return 0;
}
