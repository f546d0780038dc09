// qd_policy.h -- on-device policy inference: the reference's actor / critic MLPs evaluated for a whole
// env batch in one launch, so that a rollout never leaves the GPU.
//
// What it computes: a small "layer program" over per-env activation buffers -- dense layers y = act(W x + b)
// (nn.Linear layout, W[out][in]), eval-mode BatchNorm as a per-feature affine map, and the input gathers the
// reference's forward() methods do (slices of the observation row, the previous action).  The host side
// (mujoco_drone_amd/policy.py) compiles RMA_full / RMA_model / SimpleMLPmodel (models/PPO/RMA/RMA_model.py:77-110,
// :262-292; models/PPO/SimpleMLP/SimpleMLP.py:72-98) into such programs; the kernel is architecture-agnostic.
// The epilogue applies MyBetaDist's deterministic action (distributions.py:8-26): softplus(clamp(logits)) + 1 ->
// (alpha, beta) -> alpha / (alpha + beta).
//
// How it maps to CDNA4: this is the one GEMM-shaped piece of the path, so it runs on the matrix cores in exact f32
// (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, the reference computes in float32).  One 256-thread workgroup
// owns a tile of 16 envs (M = 16); activations never leave LDS; for each layer the four waves split the output
// features into 16-wide tiles, up to four tiles per wave at a time sharing one A operand read.  Weights are pre-packed
// on the host into the order the lanes consume them (one coalesced 1 KiB load per wave per 16x16 k-block) and streamed
// from L2 through a register ring several k-blocks ahead of the MFMAs, because at 16 envs per workgroup the kernel is
// an L2-latency chain, not a FLOP problem (57.8k MAC per env for RMA_full's actor).
#pragma once

#include "qd_math.h"

namespace qd {

constexpr int POL_MAX_OPS = 24, POL_MAX_BUFS = 4, POL_TILE = 16, POL_THREADS = 256, POL_RING = 4;
enum { POL_DENSE = 0, POL_AFFINE = 1, POL_COPY_OBS = 2, POL_COPY_PREV = 3 };
enum { POL_ACT_NONE = 0, POL_ACT_TANH = 1, POL_ACT_RELU = 2 };

struct PolOp {
  int kind, in_buf, in_off, in_dim, out_buf, out_off, out_dim, act;
  int k16, ntiles;          // DENSE: k-blocks of 16 inputs, 16-wide output tiles (both padded, the padding holds zeros)
  long long w_off, b_off;   // float offsets into the packed device blob (AFFINE: scale / shift)
};

struct PolArgs {
  PolOp ops[POL_MAX_OPS];
  int n_ops, n_bufs;
  int ld[POL_MAX_BUFS], base[POL_MAX_BUFS];  // row stride / first float of each activation buffer in LDS
  int lds_floats;
  int obs_dim, act_dim;
  int logits_buf, logits_off, n_logits;
  int value_buf, value_off;                  // value_buf < 0: the program has no value head
  const float* packed;
};

typedef float pol_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pol_act(float x, int act) {
  if (act == POL_ACT_TANH) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); exp overflow / underflow give the right limits
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  }
  if (act == POL_ACT_RELU) return fmaxf(x, 0.f);
  return x;
}

// U output tiles (tile, tile + 4, ...) of one dense layer for this wave; A operand shared by the U tiles
template <int U>
__device__ __forceinline__ void pol_dense_tiles(const PolArgs& p, const PolOp& op, float* lds, int tile0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const float* a_ptr = lds + p.base[op.in_buf] + i * p.ld[op.in_buf] + op.in_off + g * 4;
  const float4* w_ptr[U];
  pol_f32x4 acc[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int tile = tile0 + 4 * u;
    w_ptr[u] = reinterpret_cast<const float4*>(p.packed + op.w_off) + ((size_t)tile * op.k16) * 64 + lane;
    const float b = p.packed[op.b_off + tile * 16 + i];
    acc[u] = pol_f32x4{b, b, b, b};
  }
  // register ring: the weights of k-block kb + POL_RING are in flight while k-block kb is multiplied
  float4 ring[POL_RING][U];
  const int k16 = op.k16;
#pragma unroll
  for (int d = 0; d < POL_RING; d++) {
    const int kb = min(d, k16 - 1);  // clamped: short layers re-read their last block instead of branching
#pragma unroll
    for (int u = 0; u < U; u++) ring[d][u] = w_ptr[u][(size_t)kb * 64];
  }
  for (int kb0 = 0; kb0 < k16; kb0 += POL_RING) {
#pragma unroll
    for (int d = 0; d < POL_RING; d++) {
      const int kb = kb0 + d;
      if (kb < k16) {  // wave-uniform
        const float4 a4 = *reinterpret_cast<const float4*>(a_ptr + kb * 16);
        float4 w[U];
#pragma unroll
        for (int u = 0; u < U; u++) w[u] = ring[d][u];
        const int kn = min(kb + POL_RING, k16 - 1);
#pragma unroll
        for (int u = 0; u < U; u++) ring[d][u] = w_ptr[u][(size_t)kn * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w[u].x, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w[u].y, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w[u].z, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w[u].w, acc[u], 0, 0, 0);
      }
    }
  }
  // D layout: lane holds rows 4g..4g+3 of column i of each tile
  float* o_ptr = lds + p.base[op.out_buf] + (4 * g) * p.ld[op.out_buf] + op.out_off;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int col = (tile0 + 4 * u) * 16 + i;
    if (col < op.out_dim) {
#pragma unroll
      for (int v = 0; v < 4; v++) o_ptr[v * p.ld[op.out_buf] + col] = pol_act(acc[u][v], op.act);
    }
  }
}

__global__ __launch_bounds__(POL_THREADS) void k_policy(PolArgs p, int n_envs, const float* __restrict__ obs,
                                                        const float* __restrict__ prev_actions,
                                                        const uint8_t* __restrict__ prev_truncated, float* __restrict__ actions,
                                                        float* __restrict__ logits, float* __restrict__ value) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int env0 = blockIdx.x * POL_TILE;
  for (int k = tid; k < p.lds_floats; k += POL_THREADS) lds[k] = 0.f;  // padding columns must hold zeros, not NaNs
  __syncthreads();
  for (int o = 0; o < p.n_ops; o++) {
    const PolOp& op = p.ops[o];
    if (op.kind == POL_DENSE) {
      // this wave's tiles: wave, wave + 4, ...; up to four at a time
      for (int t = wave; t < op.ntiles; t += 16) {
        const int left = (op.ntiles - t + 3) >> 2;
        if (left >= 4) pol_dense_tiles<4>(p, op, lds, t, lane);
        else if (left == 3) { pol_dense_tiles<2>(p, op, lds, t, lane); pol_dense_tiles<1>(p, op, lds, t + 8, lane); }
        else if (left == 2) pol_dense_tiles<2>(p, op, lds, t, lane);
        else pol_dense_tiles<1>(p, op, lds, t, lane);
      }
    } else if (op.kind == POL_AFFINE) {
      float* b = lds + p.base[op.out_buf] + op.out_off;
      const int ld = p.ld[op.out_buf];
      for (int k = tid; k < POL_TILE * op.out_dim; k += POL_THREADS) {
        const int r = k / op.out_dim, c = k - r * op.out_dim;
        b[r * ld + c] = fmaf(b[r * ld + c], p.packed[op.w_off + c], p.packed[op.b_off + c]);
      }
    } else {
      float* b = lds + p.base[op.out_buf] + op.out_off;
      const int ld = p.ld[op.out_buf];
      for (int k = tid; k < POL_TILE * op.in_dim; k += POL_THREADS) {
        const int r = k / op.in_dim, c = k - r * op.in_dim;
        const int e = env0 + r;
        float v = 0.f;
        if (e < n_envs) {
          if (op.kind == POL_COPY_OBS) v = obs[(size_t)e * p.obs_dim + op.in_off + c];
          else if (prev_actions && !(prev_truncated && prev_truncated[e])) v = prev_actions[(size_t)e * p.act_dim + op.in_off + c];
        }
        b[r * ld + c] = v;
      }
    }
    __syncthreads();
  }
  // outputs: logits, MyBetaDist.deterministic_sample (distributions.py:8-26), value
  const float* lg = lds + p.base[p.logits_buf] + p.logits_off;
  const int ldl = p.ld[p.logits_buf];
  if (logits)
    for (int k = tid; k < POL_TILE * p.n_logits; k += POL_THREADS) {
      const int r = k / p.n_logits, c = k - r * p.n_logits;
      if (env0 + r < n_envs) logits[(size_t)(env0 + r) * p.n_logits + c] = lg[r * ldl + c];
    }
  if (actions) {
    const int h = p.n_logits >> 1;
    for (int k = tid; k < POL_TILE * h; k += POL_THREADS) {
      const int r = k / h, c = k - r * h;
      if (env0 + r < n_envs) {
        const float la = qclamp(lg[r * ldl + c], -50.f, 50.f), lb = qclamp(lg[r * ldl + h + c], -50.f, 50.f);
        const float al = log1pf(__expf(la)) + 1.0f, be = log1pf(__expf(lb)) + 1.0f;
        actions[(size_t)(env0 + r) * p.act_dim + c] = al * __builtin_amdgcn_rcpf(al + be);
      }
    }
  }
  if (value && p.value_buf >= 0 && tid < POL_TILE && env0 + tid < n_envs)
    value[env0 + tid] = lds[p.base[p.value_buf] + tid * p.ld[p.value_buf] + p.value_off];
}

}  // namespace qd
