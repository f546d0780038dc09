// qd_latency.hip -- where a step-shaped launch at 4096 envs spends its memory waits (diagnostic, not product).
// 64 workgroups x 192 threads, dependent launches replayed from a graph like qd_step_fragment's.  Each wave stamps s_memtime
// around five rounds of memory operations, each waited for before the next is issued:
//   R0  one dword from a small read-only buffer: the first memory access of the wave
//   R1  13 planes the PREVIOUS launch wrote (the env state)
//   R2  13 planes nobody writes (read by every launch: do clean lines survive the kernel boundary in L2?)
//   R3  the 13 planes of R1 again (a cache hit, for scale)
//   R4  7 planes written back + (optionally) 88 bytes per lane of non-temporal stores to a place that moves with every launch
//       (the observation rows of a fragment), waited for with s_waitcnt vmcnt(0): the drain
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-mllvm -amdgpu-kernarg-preload-count=8] -o tests/_build/qd_latency tools/microbench/qd_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                                  \
    }                                                                                           \
  } while (0)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return t;
}
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(192) void k_lat(float4* g, int npad, const unsigned* probe, float* stream, unsigned long long* out,
                                             int stream_on) {
  const unsigned long long t0 = now();
  const int lane = threadIdx.x & 63, i = blockIdx.x * 64 + lane;
  const unsigned p = probe[blockIdx.x];
  drain();
  const unsigned long long t1 = now();
  float4 v[13];
#pragma unroll
  for (int k = 0; k < 13; k++) v[k] = g[k * npad + i];
  drain();
  const unsigned long long t2 = now();
  float4 w[13];
#pragma unroll
  for (int k = 0; k < 13; k++) w[k] = g[(13 + k) * npad + i];
  drain();
  const unsigned long long t3 = now();
  float4 u[13];
#pragma unroll
  for (int k = 0; k < 13; k++) { const f4 q = reinterpret_cast<const volatile f4*>(g)[k * npad + i]; u[k] = make_float4(q.x, q.y, q.z, q.w); }
  drain();
  const unsigned long long t4 = now();
  float4 acc = make_float4((float)p, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 13; k++) {
    acc.x += v[k].x + w[k].x + u[k].x; acc.y += v[k].y + w[k].y + u[k].y;
    acc.z += v[k].z + w[k].z + u[k].z; acc.w += v[k].w + w[k].w + u[k].w;
  }
  if (threadIdx.x < 64) {
#pragma unroll
    for (int k = 0; k < 7; k++) g[k * npad + i] = make_float4(acc.x * 1e-3f, acc.y * 1e-3f, acc.z * 1e-3f, acc.w * 1e-3f + k);
  }
  if (stream_on) {
    float* row = stream + (size_t)blockIdx.x * 64 * 24 + threadIdx.x * 8;   // 192 threads x 32 bytes = the group's 64 rows of 96 bytes
    { f4 q = {acc.x, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(q, reinterpret_cast<f4*>(row)); }
    { f4 q = {acc.w, acc.z, acc.y, acc.x}; __builtin_nontemporal_store(q, reinterpret_cast<f4*>(row) + 1); }
  }
  const unsigned long long t5 = now();
  drain();
  const unsigned long long t6 = now();
  if (threadIdx.x == 0) {
    unsigned long long* o = out + blockIdx.x * 8;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = t6 - t5; o[6] = t6 - t0;
  }
}

int main(int argc, char** argv) {
  const int n = 4096, npad = 4096, planes = 26, T = argc > 1 ? atoi(argv[1]) : 1024, reps = 4;
  const int threads = argc > 2 ? atoi(argv[2]) : 192;   // 192: three waves fetch the same planes (k_step_coop); 64: one wave
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float4* g;
  unsigned* probe;
  float* stream;
  unsigned long long* out;
  const size_t row_bytes = (size_t)n * 24 * 4;
  CK(hipMalloc(&g, sizeof(float4) * npad * planes));
  CK(hipMemset(g, 0, sizeof(float4) * npad * planes));
  CK(hipMalloc(&probe, 4096));
  CK(hipMemset(probe, 0, 4096));
  CK(hipMalloc(&stream, row_bytes * T));
  CK(hipMalloc(&out, 64 * 8 * 8));
  for (int stream_on = 0; stream_on < 2; stream_on++) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < T; t++)
      hipLaunchKernelGGL(k_lat, dim3(n / 64), dim3(threads), 0, s, g, npad, probe, stream + (size_t)t * n * 24, out, stream_on);
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
    std::vector<unsigned long long> h(64 * 8);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%d threads per workgroup, fragment of %d launches, observation-like stream %s (%.1f MB per fragment): %.3f us per launch\n", threads, T, stream_on ? "ON" : "off",
           stream_on ? row_bytes * T / 1e6 : 0.0, ms * 1e3 / ((double)T * reps));
    const char* names[7] = {"R0 first access (1 dword)", "R1 13 planes the previous launch wrote", "R2 13 planes nobody writes",
                            "R3 the planes of R1 again", "R4 stores issued", "R4 stores drained", "whole wave"};
    for (int j = 0; j < 7; j++) {
      std::vector<unsigned long long> c(64);
      for (int b = 0; b < 64; b++) c[b] = h[b * 8 + j];
      std::sort(c.begin(), c.end());
      printf("   %-42s median %6llu   min %6llu   max %6llu cycles (wave 0 of each of the 64 workgroups, last launch)\n", names[j], c[32], c[0], c[63]);
    }
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
  return 0;
}
