// qd_rng.h -- counter-based per-env random numbers and the reset / domain
// randomisation transforms.
//
// The reference draws every random number from ONE sequential numpy PCG64 stream
// (BaseDroneEnv.py:113), drone after drone, so env i's values depend on how many
// numbers all earlier drones consumed (the ziggurat normal sampler consumes a
// variable amount).  That order cannot be kept when 4096+ envs reset independently
// in parallel, so the device uses Philox4x32-10 keyed by the env seed with the
// counter (env index, episode / regen count, block, stream id): every env and every
// episode has its own reproducible stream, independent of scheduling.  What IS kept
// identical to the reference is the transform from raw draws to values:
//   sample_state           BaseDroneEnv.py:218-257   (15 normals + 2 uniforms)
//   generate_drone_params  BaseDroneEnv.py:180-216   (6 uniforms)
#pragma once
#include "qd_dynamics.h"
#include "qd_math.h"
#include "qd_model.h"

namespace qd {

QD_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 23-bit uniform strictly inside (0,1), exactly representable in float32
QD_HD float u32_to_unit(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

constexpr uint32_t STREAM_STATE = 1u, STREAM_PARAMS = 2u;

// reset-state sampling configuration (already scaled by state_difficulty, BaseDroneEnv.py:101-106)
struct SampleCfg {
  float start_pos[4];
  float max_pos_offset;
  float angle_var[2], vel_var[3], ang_vel_var[3], pend_rp_var[2], pend_vel_var[2];
  int   random_start;   // 0 fixed, 1 BaseDroneEnv.sample_state, 2 SimpleDrone.reset_model
};

QD_HD float clipf(float x, float lim) { return x < -lim ? -lim : (x > lim ? lim : x); }

// the 15 standard normals and 2 uniforms one reset consumes, in the reference's draw order -- in two stages, because the
// persistent fragment kernel (qd_rollout_coop.hip) spreads one sample over the idle slots of several steps:
//   sample_words<B0, B1>  Philox blocks B0 .. B1-1 of the (env, episode) stream -> w[4 B0 .. 4 B1 - 1]   (~100 instructions per block)
//   draws_from_words      Box-Muller pairs r = sqrt(-2 ln u1), angle = 2 pi u2 over w[0..15] -> z[16]; w[16], w[17] -> u[2]
template <int B0, int B1>
QD_HD void sample_words(uint64_t seed, uint32_t env, uint32_t episode, uint32_t* w /* [20] */) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (uint32_t b = B0; b < B1; b++) philox4x32_10(env, episode, b, STREAM_STATE, k0, k1, w + 4 * b);
}
QD_HD void draws_from_words(const uint32_t w[20], float z[16], float u[2]) {
  // No multiply-add fusion in the two float stages of a sample: a sample must be the same bits whoever computes it -- a reset
  // kernel, a sampler wave, the chunked job of the persistent fragment kernel or a truncating lane itself -- and which products
  // the compiler fuses depends on the code around the inlined body.  (Found by the cut-invariance test of k_rollout_coop: rows
  // changed in the last bit with WHERE an entry had been sampled.)
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const float u1 = u32_to_unit(w[2 * i]), u2 = u32_to_unit(w[2 * i + 1]);
#if defined(__HIP_DEVICE_COMPILE__)
    // v_log_f32 is log2; v_sin_f32 / v_cos_f32 take their argument in revolutions, i.e. u2 itself
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));
    const float sn = __builtin_amdgcn_sinf(u2), cs = __builtin_amdgcn_cosf(u2);
#else
    const float r = sqrtf(-2.0f * logf(u1)), a = 6.28318530717958647692f * u2;
    const float sn = sinf(a), cs = cosf(a);
#endif
    z[2 * i] = r * cs;
    z[2 * i + 1] = r * sn;
  }
  u[0] = u32_to_unit(w[16]);
  u[1] = u32_to_unit(w[17]);
}
QD_HD void sample_draws(uint64_t seed, uint32_t env, uint32_t episode, float z[16], float u[2]) {
  uint32_t w[20];
  sample_words<0, 5>(seed, env, episode, w);
  draws_from_words(w, z, u);
}

// BaseDroneEnv.sample_state: raw draws -> qpos / qvel (activations are NOT touched: QUIRK C-2)
template <bool LOAD>
QD_HD void sample_state(const SampleCfg& c, const float z[16], const float u[2], State<float>& s) {
#pragma clang fp contract(off)   // see draws_from_words
  float roll = 0.f, pitch = 0.f, yaw = c.start_pos[3];
  if (c.random_start == 1) {
    const float inv = frsq(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    const float r = c.max_pos_offset * cbrtf(u[0]);
    s.px = c.start_pos[0] + r * (z[0] * inv);
    s.py = c.start_pos[1] + r * (z[1] * inv);
    s.pz = c.start_pos[2] + r * (z[2] * inv);
    roll = clipf(z[3] * c.angle_var[0], 2.f * c.angle_var[0]);
    pitch = clipf(z[4] * c.angle_var[1], 2.f * c.angle_var[1]);
    yaw = 3.14159265358979323846f - 6.28318530717958647692f * u[1];
    s.vx = clipf(z[5] * c.vel_var[0], 2.f * c.vel_var[0]);
    s.vy = clipf(z[6] * c.vel_var[1], 2.f * c.vel_var[1]);
    s.vz = clipf(z[7] * c.vel_var[2], 2.f * c.vel_var[2]);
    s.wx = clipf(z[8] * c.ang_vel_var[0], 2.f * c.ang_vel_var[0]);
    s.wy = clipf(z[9] * c.ang_vel_var[1], 2.f * c.ang_vel_var[1]);
    s.wz = clipf(z[10] * c.ang_vel_var[2], 2.f * c.ang_vel_var[2]);
    if (LOAD) {
      s.th1 = clipf(z[11] * c.pend_rp_var[0], 2.f * c.pend_rp_var[0]);
      s.th2 = clipf(z[12] * c.pend_rp_var[1], 2.f * c.pend_rp_var[1]);
      s.thd1 = clipf(z[13] * c.pend_vel_var[0], 2.f * c.pend_vel_var[0]);
      s.thd2 = clipf(z[14] * c.pend_vel_var[1], 2.f * c.pend_vel_var[1]);
    } else {
      s.th1 = s.th2 = s.thd1 = s.thd2 = 0.f;
    }
  } else {
    s.px = c.start_pos[0]; s.py = c.start_pos[1]; s.pz = c.start_pos[2];
    s.vx = s.vy = s.vz = s.wx = s.wy = s.wz = 0.f;
    s.th1 = s.th2 = s.thd1 = s.thd2 = 0.f;
  }
  // mujoco_rpy2quat (transformation.py:21-24)
  float sr, cr, sp, cp, sy, cy;
  qsincos(0.5f * roll, &sr, &cr);
  qsincos(0.5f * pitch, &sp, &cp);
  qsincos(0.5f * yaw, &sy, &cy);
  s.qw = cr * cp * cy + sr * sp * sy;
  s.qx = sr * cp * cy - cr * sp * sy;
  s.qy = cr * sp * cy + sr * cp * sy;
  s.qz = cr * cp * sy - sr * sp * cy;
}

// SimpleDrone.reset_model (SimpleDrone.py:63-72): qpos = qpos0 + U(-0.03, 0.03) on every
// coordinate (QUIRK C-9: the quaternion is perturbed and left unnormalised), then
// qpos[:3] = start_pos, which moves ONLY drone 0: the others stay on the spawn grid
// (env_gen.py:116-124) plus their noise.  qvel = U(-0.01, 0.01).
QD_HD void sample_simple(const SampleCfg& c, uint64_t seed, uint32_t env, uint32_t n, uint32_t episode, State<float>& s) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t w[16];
#pragma unroll
  for (uint32_t b = 0; b < 4; b++) philox4x32_10(env, episode, b, STREAM_STATE, k0, k1, w + 4 * b);
  float u[13];
#pragma unroll
  for (int i = 0; i < 13; i++) u[i] = u32_to_unit(w[i]);
  if (env == 0) {
    s.px = c.start_pos[0]; s.py = c.start_pos[1]; s.pz = c.start_pos[2];
  } else {
    const int sz = (int)ceil(sqrt((double)n));
    s.px = (float)round5(((double)(env % sz) - (sz - 1) * 0.5) * 0.5) + (-0.03f + 0.06f * u[10]);
    s.py = (float)round5(((double)(env / sz) - (sz - 1) * 0.5) * 0.5) + (-0.03f + 0.06f * u[11]);
    s.pz = 0.15f + (-0.03f + 0.06f * u[12]);
  }
  s.qw = 1.0f + (-0.03f + 0.06f * u[0]); s.qx = -0.03f + 0.06f * u[1];
  s.qy = -0.03f + 0.06f * u[2];          s.qz = -0.03f + 0.06f * u[3];
  s.vx = -0.01f + 0.02f * u[4]; s.vy = -0.01f + 0.02f * u[5]; s.vz = -0.01f + 0.02f * u[6];
  s.wx = -0.01f + 0.02f * u[7]; s.wy = -0.01f + 0.02f * u[8]; s.wz = -0.01f + 0.02f * u[9];
  s.th1 = s.th2 = s.thd1 = s.thd2 = 0.f;
}

// generate_drone_params for one env: c + uniform(-w, w) * difficulty, float64
struct ParamCfg {
  double center[6], width[6];
  double difficulty;
  int    random_params;
  int    load;
};
QD_HD void gen_params(const ParamCfg& c, uint64_t seed, uint32_t env, uint32_t regen, double raw[6]) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t w[12];
#pragma unroll
  for (uint32_t b = 0; b < 3; b++) philox4x32_10(env, regen, b, STREAM_PARAMS, k0, k1, w + 4 * b);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (c.random_params) {
      const uint64_t x = ((uint64_t)w[2 * k] << 32) | w[2 * k + 1];
      const double uu = (double)(x >> 11) * (1.0 / 9007199254740992.0);
      const double un = -c.width[k] + (c.width[k] - (-c.width[k])) * uu;
      raw[k] = c.center[k] + un * c.difficulty;
    } else {
      raw[k] = c.center[k];
    }
  }
  if (!c.load) { raw[4] = 0.0; raw[5] = 0.0; }  // BaseDroneEnv.py:212-213: `self.pendulum * value`
}

}  // namespace qd
