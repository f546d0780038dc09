// qd_policy_dist.h -- the output stage shared by the policy kernels: the action distributions of distributions.py on the logits.
// MyBetaDist (distributions.py:6-38), what every training script configures:
//   inputs  = softplus(clamp(logits, -50, 50)) + 1, first half alpha (concentration1), second half beta (:12-17)
//   deterministic_sample = alpha / (alpha + beta)                                                     (:24-26)
//   sample               = Beta(alpha, beta) draw (TorchBeta.sample -> torch.distributions.Beta.sample, no squashing)
//   logp(x)              = sum_d log Beta(clamp(x_d, 0.01, 0.99); alpha_d, beta_d)                    (:19-22)
// The reference draws from torch's global generator; here every (env, step counter, action dimension) has its own
// Philox4x32-10 stream keyed by the caller's seed (as for resets, qd_rng.h), so a rollout is reproducible whatever the
// scheduling.  Beta(a, b) = Ga / (Ga + Gb) with Marsaglia-Tsang gamma variates; both shapes are >= 1 by construction
// (softplus + 1), so the shape-boost step for a < 1 is never needed.
// MySquashedGaussian (distributions.py:41-119), the alternative the scripts import: logits = mean | log_std,
//   std = exp(clamp(log_std, -5, 5));  _squash = sigmoid (clamped to [0, 1]);  deterministic_sample = sigmoid(mean)   (:54-66, :103-106)
//   sample  = sigmoid(mean + std * N(0, 1))                                                                         (:68-71)
//   logp(x) = sum_d clamp(log N(u_d; mean_d, std_d), -100, 100) - sum_d log(1 - tanh(u_d)^2 + 1e-4),
//             u = atanh(clamp(2 x - 1, -1 + 1e-4, 1 - 1e-4))                                                        (:73-85, :108-112)
//   QUIRK reproduced: the class squashes with a sigmoid but un-squashes with atanh(2 x - 1) = z / 2, so the log-probability of
//   its own sample is evaluated at half the pre-squash value.
#pragma once

#include "qd_rng.h"

namespace qd {

constexpr uint32_t STREAM_POLICY = 3u;
constexpr int POL_DIST_BETA = 0, POL_DIST_SQUASHED_GAUSSIAN = 1;  // = QD_DIST_* of include/qd.h
#ifndef QD_POL_THREADS
#define QD_POL_THREADS 256
#endif
#ifndef QD_POL_TILE
#define QD_POL_TILE 16   // 32 in qd_rollout_fused32.hip only: two 16-env MFMA blocks per workgroup share every weight they load
#endif
constexpr int POL_TILE = QD_POL_TILE, POL_THREADS = QD_POL_THREADS;  // envs per workgroup (M tiles of 16), threads per workgroup
constexpr int POL_SCRATCH = 4 * POL_TILE;  // floats reserved behind the activation buffers for the per-env log-prob reduction

struct PolSample {
  int explore;             // 0: deterministic_sample, 1: sample
  unsigned int counter;    // the caller's step counter: one stream per (env, counter, action dimension)
  unsigned long long seed;
};

// log Gamma(z) for z >= 1: recurrence up to z >= 8, then Stirling's series (float32: abs error ~2e-6 for z <= 60)
__device__ __forceinline__ float pol_lgamma(float z) {
  float p = 1.0f;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const bool low = z < 8.0f;
    p = low ? p * z : p;
    z = low ? z + 1.0f : z;
  }
  const float r = __builtin_amdgcn_rcpf(z), r2 = r * r;
  const float series = r * fmaf(r2, fmaf(r2, 7.9365079365e-4f, -2.7777777778e-3f), 8.3333333333e-2f);
  return fmaf(z - 0.5f, __logf(z), -z) + 0.91893853320467274f + series - __logf(p);
}

// Gamma(a, 1), a >= 1 (Marsaglia & Tsang 2000); `sub` separates the streams of the two variates of one Beta draw
__device__ __forceinline__ float pol_gamma(float a, const PolSample& s, uint32_t env, uint32_t sub) {
  const float d = a - 0.33333333333f, c = __builtin_amdgcn_rsqf(9.0f * d);
  const uint32_t k0 = (uint32_t)s.seed, k1 = (uint32_t)(s.seed >> 32);
  for (uint32_t attempt = 0; attempt < 16; attempt++) {  // acceptance > 95 % per attempt
    uint32_t w[4];
    philox4x32_10(env, s.counter, sub * 16u + attempt, STREAM_POLICY, k0, k1, w);
    const float u1 = u32_to_unit(w[0]), u2 = u32_to_unit(w[1]), u = u32_to_unit(w[2]);
    const float z = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
    float v = fmaf(c, z, 1.0f);
    if (v <= 0.f) continue;
    v = v * v * v;
    if (__logf(u) < fmaf(0.5f * z, z, d - d * v + d * __logf(v))) return d * v;
  }
  return d;
}

// logits (LDS rows lgt[r * ldl + c]) -> actions / logp / logits in global memory for the workgroup's POL_TILE envs.
// `scratch` = POL_SCRATCH floats of LDS nobody else uses; every thread of the workgroup must call this (it has a barrier
// when logp is requested).
// one N(0, 1) variate per (env, step counter, action dimension): Box-Muller on the policy stream
__device__ __forceinline__ float pol_normal(const PolSample& s, uint32_t env, uint32_t sub) {
  uint32_t w[4];
  philox4x32_10(env, s.counter, sub * 16u, STREAM_POLICY, (uint32_t)s.seed, (uint32_t)(s.seed >> 32), w);
  const float u1 = u32_to_unit(w[0]), u2 = u32_to_unit(w[1]);
  return __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
}

__device__ __forceinline__ void pol_outputs(const float* lgt, int ldl, int NL, int AD, int env0, int n_envs, int tid, float* scratch,
                                            const PolSample& smp, float* __restrict__ actions, float* __restrict__ logp,
                                            float* __restrict__ logits, float* act_lds = nullptr, int dist = POL_DIST_BETA,
                                            float* prev_dst = nullptr, int prev_ld = 0, const uint8_t* prev_trunc = nullptr, int prev_rows = 0) {
  if (logits)
    for (int k = tid; k < POL_TILE * NL; k += POL_THREADS) {
      const int r = k / NL, c = k - r * NL;
      if (env0 + r < n_envs) logits[(size_t)(env0 + r) * NL + c] = lgt[r * ldl + c];
    }
  const int H = NL >> 1;
  if (actions || logp || act_lds || prev_dst) {
    for (int k = tid; k < POL_TILE * H; k += POL_THREADS) {
      const int r = k / H, c = k - r * H;
      float x, lp = 0.f;
      if (dist == POL_DIST_SQUASHED_GAUSSIAN) {  // uniform over the launch
        const float mean = lgt[r * ldl + c], ls = qclamp(lgt[r * ldl + H + c], -5.f, 5.f), sd = __expf(ls);
        const float z = smp.explore ? fmaf(sd, pol_normal(smp, (uint32_t)(env0 + r), (uint32_t)c), mean) : mean;
        x = qclamp(__builtin_amdgcn_rcpf(1.0f + __expf(-z)), 0.f, 1.f);
        if (logp) {
          const float th = qclamp(2.0f * x - 1.0f, -1.0f + 1e-4f, 1.0f - 1e-4f);
          const float u = 0.5f * __logf((1.0f + th) / (1.0f - th));  // atanh
          const float q = (u - mean) / sd;
          lp = qclamp(-0.5f * q * q - ls - 0.91893853320467274f, -100.f, 100.f) - __logf(1.0f - th * th + 1e-4f);
        }
      } else {
        const float la = qclamp(lgt[r * ldl + c], -50.f, 50.f), lb = qclamp(lgt[r * ldl + H + c], -50.f, 50.f);
        const float al = __logf(1.0f + __expf(la)) + 1.0f, be = __logf(1.0f + __expf(lb)) + 1.0f;
        if (smp.explore) {
          const float ga = pol_gamma(al, smp, (uint32_t)(env0 + r), 2u * c), gb = pol_gamma(be, smp, (uint32_t)(env0 + r), 2u * c + 1u);
          x = fminf(ga * __builtin_amdgcn_rcpf(ga + gb), 1.0f);   // (v_rcp_f32 is good to 1 ulp: without the bound 1 + 2^-23 turned up once in 1.3e9 draws)
        } else {
          x = al * __builtin_amdgcn_rcpf(al + be);
        }
        if (logp) {
          const float xc = qclamp(x, 0.01f, 0.99f);
          lp = (al - 1.0f) * __logf(xc) + (be - 1.0f) * __logf(1.0f - xc) - (pol_lgamma(al) + pol_lgamma(be) - pol_lgamma(al + be));
        }
      }
      if (actions && env0 + r < n_envs) actions[(size_t)(env0 + r) * AD + c] = x;
      if (act_lds) act_lds[r * AD + c] = x;  // fused rollouts: the env step of the same workgroup consumes it
      // pipelined rollouts: also as the next pass's previous-action input (the COPY_PREV gather: zero behind a truncation)
      if (prev_dst) prev_dst[r * prev_ld + c] = (r < prev_rows && !prev_trunc[r]) ? x : 0.f;
      if (logp && H <= POL_SCRATCH / POL_TILE) scratch[r * H + c] = lp;
    }
  }
  if (logp) {  // uniform over the workgroup
    __syncthreads();
    if (tid < POL_TILE && env0 + tid < n_envs) {
      float sum = 0.f;
      for (int c = 0; c < H; c++) sum += scratch[tid * H + c];
      logp[env0 + tid] = sum;
    }
  }
}

}  // namespace qd
