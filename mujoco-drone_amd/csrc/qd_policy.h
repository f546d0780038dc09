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
// How it maps to CDNA4: this is the one GEMM-shaped piece of the path, so it runs on the matrix cores: this file's interpreter
// in exact f32 (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, the reference computes in float32), the specialised kernels
// of qd_policy_static.h on float16 pairs (three v_mfma_f32_16x16x32_f16 per float32 product, same accuracy, 5.3x the rate).  One 256-thread workgroup
// owns a tile of 16 envs (M = 16); activations never leave LDS; for each layer the four waves split the output
// features into 16-wide tiles, up to four tiles per wave at a time sharing one A operand read.  At 16 envs per
// workgroup the kernel is a latency chain, not a FLOP problem (57.8k MAC per env for RMA_full's actor), so:
//   - weights are pre-packed on the host in the order the lanes consume them (one coalesced 1 KiB load per wave per
//     16x16 k-block) and each wave streams them from L2 one step ahead of its MFMAs, across layer boundaries;
//   - the program is pre-digested on the host into per-wave step descriptors that the kernel mirrors in LDS together
//     with the biases, so that interpreting it costs LDS reads, not chains of dependent scalar loads.
#pragma once

#include "qd_math.h"
#include "qd_policy_dist.h"

namespace qd {

constexpr int POL_MAX_OPS = 32, POL_MAX_BUFS = 8, POL_MAX_RINGS = 4, POL_WAVES = POL_THREADS / 64;  // POL_TILE, POL_THREADS: qd_policy_dist.h
constexpr int POL_KC = 4;       // k-blocks (of 16 inputs) per step
constexpr int POL_DESC = 16;    // ints per op descriptor
constexpr int POL_SDESC = 32;   // ints per step descriptor
enum { POL_DENSE = 0, POL_AFFINE = 1, POL_COPY_OBS = 2, POL_COPY_PREV = 3, POL_RING_LOAD = 4, POL_RING_PUSH = 5, POL_LSTM_CELL = 6 };
enum { POL_ACT_NONE = 0, POL_ACT_TANH = 1, POL_ACT_RELU = 2 };
enum { POL_FLAG_VALUE_ONLY = 1 };  // the op only feeds the value head: skipped when no value output is requested
enum { POL_STEP_FIRST = 1, POL_STEP_LAST = 2, POL_STEP_VALUE_ONLY = 4 };

// op descriptor (POL_DESC ints, written by pol_compile)
enum { OD_KIND = 0, OD_FLAGS, OD_SRC_OFF, OD_COUNT, OD_OUT, OD_LD_OUT, OD_SCALE, OD_SHIFT, OD_RING, OD_IN, OD_LD_IN };
// step descriptor (POL_SDESC ints): one wave, POL_KC k-blocks of up to four output tiles of one dense op.
// SD_LANE + u: 1 if tile slot u is used (lanes read consecutive float4s) else 0 (all lanes read one word);
// SD_OFF + 4 u + d: float4 offset of k-block d of tile slot u in the weight region (0 for unused slots, the last live
// block repeated past klive): the kernel's loads are pure address arithmetic on these, no conditions
enum { SD_OP = 0, SD_U, SD_FLAGS, SD_KLIVE, SD_A, SD_LD_IN, SD_BIAS, SD_OUT, SD_LD_OUT, SD_COLS, SD_ACT, SD_LANE = 12, SD_OFF = 16 };

struct PolArgs {
  const float* packed;        // [program ints | small floats | packed weights]
  int prog_ints;              // op descriptors, then the four waves' step lists
  int small_floats;           // biases, affine scale / shift
  int n_ops;
  int step_base[POL_WAVES];   // int offset of each wave's step list in the program
  int act_floats;             // activation buffers + POL_SCRATCH floats (LDS), zero-initialised
  int obs_dim, act_dim;
  int logits_lds, ld_logits, n_logits;   // LDS float offset of env row 0's logits, row stride
  int value_lds, ld_value;               // value_lds < 0: the program has no value head
  int aux_lds, ld_aux, n_aux;            // auxiliary slice read back by qd_policy_aux (n_aux 0: none)
  int dist;                              // POL_DIST_*: which distribution of distributions.py reads the logits
  long long weights_off;      // float offset of the packed weights in the blob
  long long wsplit_off;       // float offset of the same weights split into float16 pairs (the specialised kernels, qd_policy_static.h)
  // per-env history rings (see qd_policy_ring in include/qd.h)
  int n_rings, state_floats;                     // floats of history per env
  int ring_rows[POL_MAX_RINGS], ring_width[POL_MAX_RINGS], ring_period[POL_MAX_RINGS];
  int ring_off[POL_MAX_RINGS], ring_fill[POL_MAX_RINGS];  // offset in the env's history block; fill values in the small region
};

typedef float pol_f32x4 __attribute__((ext_vector_type(4)));

#if defined(QD_STAMPS) && !defined(QD_POL_SECOND_UNIT)
// diagnostic build only: s_memrealtime (100 MHz) stamps of wave 0 of workgroup 0 after every op's barrier.  Not in the pipelined
// rollout's unit: under its 256-register cap the per-op stamps change the allocation enough to slow the network by a third.
__device__ unsigned long long qd_pstamps[64];
#define POL_STAMP(k)                                                                                                  \
  do {                                                                                                                \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (k) < 64) qd_pstamps[(k)] = __builtin_amdgcn_s_memrealtime();              \
  } while (0)
#else
#define POL_STAMP(k)
#endif

__device__ __forceinline__ float pol_act(float x, int act) {
  if (act == POL_ACT_TANH) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); exp overflow / underflow give the right limits
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  }
  if (act == POL_ACT_RELU) return fmaxf(x, 0.f);
  return x;
}

struct PolStep {
  int op, U, flags, klive, a, ld_in, bias, out, ld_out, cols, act;
  int lane_mul[4];
  int off[4][POL_KC];
};
__device__ __forceinline__ int pol_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ PolStep pol_read_step(const int* prog, int idx) {
  const int4* q = reinterpret_cast<const int4*>(prog + idx);
  const int4 a = q[0], b = q[1], c = q[2], d = q[3];
  PolStep s;
  // the descriptor is the same for all lanes: keep it in scalar registers
  s.op = pol_sgpr(a.x); s.U = pol_sgpr(a.y); s.flags = pol_sgpr(a.z); s.klive = pol_sgpr(a.w);
  s.a = pol_sgpr(b.x); s.ld_in = pol_sgpr(b.y); s.bias = pol_sgpr(b.z); s.out = pol_sgpr(b.w);
  s.ld_out = pol_sgpr(c.x); s.cols = pol_sgpr(c.y); s.act = pol_sgpr(c.z);
  s.lane_mul[0] = pol_sgpr(d.x); s.lane_mul[1] = pol_sgpr(d.y); s.lane_mul[2] = pol_sgpr(d.z); s.lane_mul[3] = pol_sgpr(d.w);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int4 o = q[4 + u];
    s.off[u][0] = pol_sgpr(o.x); s.off[u][1] = pol_sgpr(o.y); s.off[u][2] = pol_sgpr(o.z); s.off[u][3] = pol_sgpr(o.w);
  }
  return s;
}
// next step of this wave's list that runs (value-only steps are skipped when no value is wanted); lists end with a
// sentinel step (op = n_ops, U = 0) that is returned forever
__device__ __forceinline__ PolStep pol_fetch(const int* prog, int& idx, bool want_value, int n_ops) {
  PolStep s = pol_read_step(prog, idx);
  while (s.op < n_ops && !want_value && (s.flags & POL_STEP_VALUE_ONLY)) {
    idx += POL_SDESC;
    s = pol_read_step(prog, idx);
  }
  if (s.op < n_ops) idx += POL_SDESC;
  return s;
}

typedef float4 PolBuf[POL_KC][4];

// the POL_KC * 4 weight loads of one step: always that many instructions and no conditions (the descriptor carries
// every offset), so the compiler can count them and wait for exactly the loads it needs
__device__ __forceinline__ void pol_issue(const float4* weights, const PolStep& s, PolBuf& buf, int lane) {
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const float4* src = weights + lane * s.lane_mul[u];
#pragma unroll
    for (int d = 0; d < POL_KC; d++) {
      buf[d][u] = src[s.off[u][d]];
    }
  }
}

template <int U>
__device__ __forceinline__ void pol_mac(const PolStep& s, const PolBuf& w, const float* a_ptr, pol_f32x4 (&acc)[4]) {
#pragma unroll
  for (int d = 0; d < POL_KC; d++) {
    if (d < s.klive) {  // wave-uniform; no global memory operation inside
      const float4 a = *reinterpret_cast<const float4*>(a_ptr + d * 16);
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[d][u].x, acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[d][u].y, acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[d][u].z, acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[d][u].w, acc[u], 0, 0, 0);
    }
  }
}

#ifndef QD_POL_SECOND_UNIT   // the two non-template kernels live in ONE translation unit (qd_kernels.hip)
// qd_policy_reset_state: every ring slot of the selected envs <- the ring's episode-start values (read from the blob)
__global__ __launch_bounds__(256) void k_policy_reset_state(PolArgs p, float* __restrict__ state, int n_envs, const uint8_t* __restrict__ mask) {
  const float* small = p.packed + p.prog_ints;
  const size_t total = (size_t)n_envs * p.state_floats;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (size_t)gridDim.x * 256) {
    const int e = (int)(k / p.state_floats), o = (int)(k - (size_t)e * p.state_floats);
    if (mask && !mask[e]) continue;
    int rg = 0;
    while (rg + 1 < p.n_rings && o >= p.ring_off[rg + 1]) rg++;
    state[k] = small[p.ring_fill[rg] + (o - p.ring_off[rg]) % p.ring_width[rg]];
  }
}

__global__ __launch_bounds__(POL_THREADS) void k_policy(PolArgs p, int n_envs, const float* __restrict__ obs,
                                                        const float* __restrict__ prev_actions,
                                                        const uint8_t* __restrict__ prev_truncated, PolSample smp,
                                                        float* __restrict__ state, float* __restrict__ actions,
                                                        float* __restrict__ logp, float* __restrict__ logits,
                                                        float* __restrict__ value, float* __restrict__ aux) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int env0 = blockIdx.x * POL_TILE;
  const bool want_value = value != nullptr;
  POL_STAMP(0);
  // LDS: [activations | program | small]
  int* prog = reinterpret_cast<int*>(lds + p.act_floats);
  float* small = lds + p.act_floats + p.prog_ints;
  {
    const float4* src = reinterpret_cast<const float4*>(p.packed);
    float4* dst = reinterpret_cast<float4*>(lds + p.act_floats);
    const int n4 = (p.prog_ints + p.small_floats) >> 2;
    for (int k = tid; k < n4; k += POL_THREADS) dst[k] = src[k];
    for (int k = tid; k < p.act_floats; k += POL_THREADS) lds[k] = 0.f;  // padding columns must hold zeros, not NaNs
  }
  __syncthreads();
  POL_STAMP(1);
  const float4* weights = reinterpret_cast<const float4*>(p.packed + p.weights_off);
  // the weight stream starts before the inputs are gathered
  int sidx = p.step_base[wave];
  PolBuf wnext, wcur;
  PolStep cur = pol_fetch(prog, sidx, want_value, p.n_ops);
  pol_issue(weights, cur, wnext, lane);
  pol_f32x4 acc[4];
  for (int o = 0; o < p.n_ops; o++) {
    const int* od = prog + o * POL_DESC;
    const int kind = od[OD_KIND];
    if (!want_value && (od[OD_FLAGS] & POL_FLAG_VALUE_ONLY)) continue;  // uniform over the workgroup
    if (kind == POL_DENSE) {
      while (cur.op == o) {
        const PolStep nx = pol_fetch(prog, sidx, want_value, p.n_ops);
#pragma unroll
        for (int d = 0; d < POL_KC; d++)
#pragma unroll
          for (int u = 0; u < 4; u++) wcur[d][u] = wnext[d][u];  // this step's weights, requested one step ago
        pol_issue(weights, nx, wnext, lane);
        if (cur.flags & POL_STEP_FIRST) {
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const float b = u < cur.U ? small[cur.bias + u * (POL_WAVES * 16) + li] : 0.f;
            acc[u] = pol_f32x4{b, b, b, b};
          }
        }
        const float* a_ptr = lds + cur.a + li * cur.ld_in + lg * 4;
        if (cur.U == 4) pol_mac<4>(cur, wcur, a_ptr, acc);
        else if (cur.U == 2) pol_mac<2>(cur, wcur, a_ptr, acc);
        else pol_mac<1>(cur, wcur, a_ptr, acc);
        if (cur.flags & POL_STEP_LAST) {
          // D layout: lane holds rows 4 lg .. 4 lg + 3 of column li of each tile
          float* o_ptr = lds + cur.out + (4 * lg) * cur.ld_out + li;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (u < cur.U && u * (POL_WAVES * 16) + li < cur.cols) {
#pragma unroll
              for (int v = 0; v < 4; v++) o_ptr[v * cur.ld_out + u * (POL_WAVES * 16)] = pol_act(acc[u][v], cur.act);
            }
          }
        }
        cur = nx;
      }
    } else if (kind == POL_AFFINE) {
      float* b = lds + od[OD_OUT];
      const int ld = od[OD_LD_OUT], n = od[OD_COUNT];
      const float* sc = small + od[OD_SCALE];
      const float* sh = small + od[OD_SHIFT];
      for (int k = tid; k < POL_TILE * n; k += POL_THREADS) {
        const int r = k / n, c = k - r * n;
        b[r * ld + c] = fmaf(b[r * ld + c], sc[c], sh[c]);
      }
    } else if (kind == POL_LSTM_CELL) {
      // torch.nn.LSTM cell, gate order (i, f, g, o); c sits right after h in the output buffer
      const float* gt = lds + od[OD_IN];
      float* hc = lds + od[OD_OUT];
      const int ldg = od[OD_LD_IN], ldh = od[OD_LD_OUT], H = od[OD_COUNT];
      for (int k = tid; k < POL_TILE * H; k += POL_THREADS) {
        const int r = k / H, j = k - r * H;
        const float* g4 = gt + r * ldg + j;
        const float si = __builtin_amdgcn_rcpf(1.0f + __expf(-g4[0])), sf = __builtin_amdgcn_rcpf(1.0f + __expf(-g4[H]));
        const float so = __builtin_amdgcn_rcpf(1.0f + __expf(-g4[3 * H]));
        const float cn = fmaf(sf, hc[r * ldh + H + j], si * pol_act(g4[2 * H], POL_ACT_TANH));
        hc[r * ldh + H + j] = cn;
        hc[r * ldh + j] = so * pol_act(cn, POL_ACT_TANH);
      }
    } else if (kind == POL_RING_LOAD || kind == POL_RING_PUSH) {
      // history rings: slot written at step t is t mod rows (period 2: bank t & 1, slot (t >> 1) mod rows)
      const int rg = od[OD_RING];
      const int R = p.ring_rows[rg], W = p.ring_width[rg], per = p.ring_period[rg];
      const unsigned tt = per == 2 ? smp.counter >> 1 : smp.counter;
      const int bank = per == 2 ? (int)(smp.counter & 1u) : 0;
      const float* fill = small + p.ring_fill[rg];
      if (kind == POL_RING_LOAD) {
        float* b = lds + od[OD_OUT];
        const int ld = od[OD_LD_OUT];
        for (int k = tid; k < POL_TILE * R * W; k += POL_THREADS) {
          const int r = k / (R * W), rem = k - r * (R * W), j = rem / W, c = rem - j * W;
          const int e = env0 + r;
          float v = 0.f;
          if (e < n_envs) {
            const bool fresh = prev_truncated && prev_truncated[e];  // first observation of a new episode
            const int slot = (int)((tt + (unsigned)j) % (unsigned)R);
            v = fresh ? fill[c] : state[(size_t)e * p.state_floats + p.ring_off[rg] + (bank * R + slot) * W + c];
          }
          b[r * ld + j * W + c] = v;
        }
      } else {
        const float* b = lds + od[OD_IN];
        const int ld = od[OD_LD_IN];
        const int slot_new = (int)(tt % (unsigned)R);
        for (int k = tid; k < POL_TILE * W; k += POL_THREADS) {
          const int r = k / W, c = k - r * W;
          const int e = env0 + r;
          if (e < n_envs) {
            float* ring = state + (size_t)e * p.state_floats + p.ring_off[rg];
            if (prev_truncated && prev_truncated[e])
              for (int q = 0; q < per * R; q++) ring[q * W + c] = fill[c];
            ring[(bank * R + slot_new) * W + c] = b[r * ld + c];
          }
        }
      }
    } else {
      float* b = lds + od[OD_OUT];
      const int ld = od[OD_LD_OUT], n = od[OD_COUNT], so = od[OD_SRC_OFF];
      for (int k = tid; k < POL_TILE * n; k += POL_THREADS) {
        const int r = k / n, c = k - r * n;
        const int e = env0 + r;
        float v = 0.f;
        if (e < n_envs) {
          if (kind == POL_COPY_OBS) v = obs[(size_t)e * p.obs_dim + so + c];
          else if (prev_actions && !(prev_truncated && prev_truncated[e])) v = prev_actions[(size_t)e * p.act_dim + so + c];
        }
        b[r * ld + c] = v;
      }
    }
    __syncthreads();
    POL_STAMP(2 + o);
  }
  // outputs: value, then logits / action / log-prob (MyBetaDist, qd_policy_dist.h)
  if (want_value && p.value_lds >= 0 && tid < POL_TILE && env0 + tid < n_envs) value[env0 + tid] = lds[p.value_lds + tid * p.ld_value];
  if (aux)
    for (int k = tid; k < POL_TILE * p.n_aux; k += POL_THREADS) {
      const int r = k / p.n_aux, c = k - r * p.n_aux;
      if (env0 + r < n_envs) aux[(size_t)(env0 + r) * p.n_aux + c] = lds[p.aux_lds + r * p.ld_aux + c];
    }
  pol_outputs(lds + p.logits_lds, p.ld_logits, p.n_logits, p.act_dim, env0, n_envs, tid, lds + p.act_floats - POL_SCRATCH, smp, actions, logp, logits, nullptr, p.dist);
  POL_STAMP(2 + p.n_ops);
}
#endif  // QD_POL_SECOND_UNIT

}  // namespace qd
